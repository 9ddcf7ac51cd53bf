"""Seeded random sweep over shapes, band counts, likelihood modes, masks, thresholds and KDE
routes: BruteForce.fit_predict (fused) and fit + predict (planes) against the oracle.  Small
problems, many combinations -- aimed at boundary handling (partial waves / tiles, padded bands,
masked-out objects, objects with no usable band)."""
import os

import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import EVID

pytestmark = pytest.mark.gpu


def dicts():
    from frankenz_amd import PDFDict
    grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
    return PDFDict(grid, sg), fo.KernelDict(grid, sg)


def rows_defined(a, b):
    """rows where both sides are fully finite"""
    return np.isfinite(a).all(axis=1) & np.isfinite(b).all(axis=1)


@pytest.mark.parametrize('seed', range(int(os.environ.get('FZ_FUZZ_SEEDS', 160))))
def test_random_configuration(seed):
    from frankenz_amd import BruteForce
    d, od = dicts()
    rs = np.random.RandomState(5000 + seed)
    N = int(rs.choice([1, 2, 3, 7, 33, 64, 65, 70]))
    M = int(rs.choice([1, 2, 5, 63, 64, 65, 130, 255, 256, 257, 600]))
    B = int(rs.choice([1, 2, 3, 5, 5, 5, 6, 8, 9, 12, 17]))
    mode = seed % 4
    kw = [{}, {'ignore_model_err': True}, {'free_scale': True, 'ignore_model_err': True}, {'free_scale': True}][mode]
    if rs.rand() < 0.3:
        kw = dict(kw, dim_prior=False)
    sig = rs.uniform(0.2, 2.0, B)
    Y = rs.lognormal(1., 1., size=(M, 1)) * rs.lognormal(0., .6, size=(M, B))
    Ye = Y * rs.uniform(0.01, 0.08, size=(M, B))
    Ym = np.ones((M, B)); Xm = np.ones((N, B))
    if rs.rand() < 0.5:
        Ym[rs.rand(M, B) < 0.1] = 0
    if rs.rand() < 0.5:
        Xm[rs.rand(N, B) < 0.15] = 0
    X = Y[rs.choice(M, N)] * rs.lognormal(0, .3, N)[:, None] + sig * rs.randn(N, B)
    Xe = np.tile(sig, (N, 1))
    if rs.rand() < 0.3:
        X[rs.randint(N), rs.randint(B)] = np.nan           # cleaned in place like pdf.py:310-311
    z = rs.uniform(0, 6, M); ze = rs.uniform(0.01, 0.3, M) if rs.rand() < 0.5 else np.full(M, 0.05)
    kde = {'wt_thresh': float(rs.choice([1e-3, 1e-2, 1e-6]))}
    route = dict(label_dict=d) if rs.rand() < 0.7 else dict(label_grid=d.grid)
    oroute = dict(label_dict=od) if 'label_dict' in route else dict(label_grid=od.grid)
    bf = BruteForce(Y, Ye, Ym)
    xa, xea, xma = X.copy(), Xe.copy(), Xm.copy()
    p, (lm, le) = bf.fit_predict(xa, xea, xma, z, ze, lprob_kwargs=kw, kde_kwargs=kde, return_gof=True, verbose=False,
                                 save_fits=False, **route)
    xb, xeb, xmb = X.copy(), Xe.copy(), Xm.copy()
    rp, rlm, rle = fo.bruteforce_fit_predict(xb, xeb, xmb, Y, Ye, Ym, z, ze, kde_kwargs=kde, **oroute, **kw)
    np.testing.assert_array_equal(xa, xb); np.testing.assert_array_equal(xea, xeb); np.testing.assert_array_equal(xma, xmb)
    rf = fo.bruteforce_fit(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, **kw)
    # free scale + dim prior with exactly one usable band is undefined in the reference itself (nan or -inf)
    undefined = (rf['Ndim'] == 1).any(axis=1) if (kw.get('free_scale') and kw.get('dim_prior', True)) else np.zeros(N, bool)
    ok = rows_defined(p, rp) & ~undefined
    assert ok.sum() >= (~undefined).sum() - np.isnan(rp).all(axis=1).sum() - 1 or N <= 3
    np.testing.assert_allclose(p[ok], rp[ok], rtol=2e-7, atol=1e-13)
    np.testing.assert_allclose(lm[ok], rlm[ok], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(le[ok], rle[ok], **EVID)
    both_nan = np.isnan(rp).all(axis=1) & ~undefined
    assert np.isnan(p[both_nan]).all()
    # materialised planes and predict() from them
    bf.fit(X.copy(), Xe.copy(), Xm.copy(), lprob_kwargs=kw, verbose=False)
    fin = np.isfinite(rf['lnlike']) & ~undefined[:, None]
    np.testing.assert_allclose(bf.fit_lnlike[fin], rf['lnlike'][fin], rtol=1e-8, atol=1e-8)
    np.testing.assert_array_equal(bf.fit_Ndim, rf['Ndim'])
    np.testing.assert_allclose(bf.fit_chi2[fin], rf['chi2'][fin], rtol=1e-8, atol=1e-8)
    p2 = bf.predict(z, ze, kde_kwargs=kde, verbose=False, **route)
    ok2 = rows_defined(p2, rp) & ~undefined
    np.testing.assert_allclose(p2[ok2], rp[ok2], rtol=2e-7, atol=1e-13)


@pytest.mark.parametrize('seed', range(int(os.environ.get('FZ_FUZZ_KNN_SEEDS', 40))))
def test_random_knn_configuration(seed):
    """NearestNeighbors: random sizes, K, k, feature maps, likelihood modes, optional ln-prior."""
    from frankenz_amd import NearestNeighbors
    from frankenz_amd.pdf import logprob_prior
    d, od = dicts()
    rs = np.random.RandomState(9000 + seed)
    N = int(rs.choice([1, 5, 40, 70]))
    M = int(rs.choice([30, 64, 200, 1000]))
    B = int(rs.choice([3, 5, 5, 8]))
    K = int(rs.choice([1, 3, 6])); k = int(rs.choice([1, 4, 9]))
    k = min(k, M)
    fmap = str(rs.choice(['identity', 'magnitude', 'luptitude']))
    mode = seed % 4
    kw = [{}, {'ignore_model_err': True}, {'free_scale': True, 'ignore_model_err': True}, {'free_scale': True}][mode]
    sig = rs.uniform(0.2, 1.0, B)
    Y = (rs.lognormal(1., 1., size=(M, 1)) * rs.lognormal(0., .5, size=(M, B)) + 3.0) * 10
    Ye = Y * rs.uniform(0.01, 0.05, size=(M, B)); Ym = np.ones((M, B))
    X = Y[rs.choice(M, N)] + sig * rs.randn(N, B); Xe = np.tile(sig, (N, 1)); Xm = np.ones((N, B))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
    fk = {} if fmap == 'identity' else (dict(zeropoints=10 ** (0.4 * 23.9)) if fmap == 'magnitude'
                                        else dict(skynoise=sig, zeropoints=10 ** (0.4 * 23.9)))
    prior = None; lp = None
    if rs.rand() < 0.4:
        tab = np.log(rs.dirichlet(np.full(M, 0.5), size=3)); rows = rs.randint(0, 3, N)
        prior = logprob_prior(tab, rows); lp = tab[rows]
    nn = NearestNeighbors(Y, Ye, Ym, K=K, feature_map=fmap, fmap_kwargs=fk, rstate=np.random.RandomState(seed), verbose=False)
    p, (lm, le) = nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, lprob_func=prior, lprob_kwargs=kw,
                                 rstate=np.random.RandomState(seed + 1), k=k, label_dict=d, return_gof=True, verbose=False)
    feats = fo.knn_train(Y, Ye, K, fmap, np.random.RandomState(seed), **fk)
    q = fo.knn_query_features(X, Xe, fmap, np.random.RandomState(seed + 1), **fk)
    tab_nb = fo.knn_neighbors_exact(feats, q, k)
    rp, rlm, rle, rn, rnn, rlnp = fo.knn_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, tab_nb, z, ze, label_dict=od,
                                                     lnprior=lp, **kw)
    # exact search: identical unless two float32 feature distances tie -- compare rows whose tables agree
    same = (nn.neighbors == rn).all(axis=1)
    assert same.mean() >= 0.9 or N < 10
    ok = same & np.isfinite(rp).all(axis=1) & np.isfinite(p).all(axis=1)
    np.testing.assert_array_equal(nn.Nneighbors[same], rnn[same])
    np.testing.assert_allclose(p[ok], rp[ok], rtol=2e-7, atol=1e-13)
    np.testing.assert_allclose(le[ok], rle[ok], **EVID)
    fin = np.isfinite(rlnp) & same[:, None]
    np.testing.assert_allclose(nn.fit_lnprob[fin], rlnp[fin], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize('seed', range(int(os.environ.get('FZ_FUZZ_BIG_SEEDS', 12))))
def test_random_configuration_full_chip_geometries(seed):
    """enough objects (>= 64 per CU) for the 4x8 / 2x16 launch geometries, model counts off the
    tile size; the oracle checks a sample of objects (objects are independent)."""
    from frankenz_amd import BruteForce
    d, od = dicts()
    rs = np.random.RandomState(7000 + seed)
    N = 16384 + int(rs.randint(1, 700))
    M = int(rs.choice([300, 1000, 2049, 511]))
    B = 5 if seed % 3 else int(rs.choice([4, 7]))
    kw = [{}, {'ignore_model_err': True}, {'free_scale': True, 'ignore_model_err': True}, {'dim_prior': False}][seed % 4]
    sig = rs.uniform(0.2, 2.0, B)
    Y = rs.lognormal(1., 1., size=(M, 1)) * rs.lognormal(0., .6, size=(M, B)); Ye = Y * rs.uniform(0.01, 0.08, size=(M, B))
    Ym = np.ones((M, B)); Xm = np.ones((N, B))
    if seed % 2:
        Xm[rs.rand(N, B) < 0.05] = 0
    X = Y[rs.choice(M, N)] * rs.lognormal(0, .3, N)[:, None] + sig * rs.randn(N, B); Xe = np.tile(sig, (N, 1))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
    p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, lprob_kwargs=kw,
                                                    return_gof=True, verbose=False, save_fits=False)
    pick = np.concatenate([[0, 1, N - 1, N - 2], rs.choice(N, 40, replace=False)])
    rp, rlm, rle = fo.bruteforce_fit_predict(X[pick].copy(), Xe[pick].copy(), Xm[pick].copy(), Y, Ye, Ym, z, ze, label_dict=od, **kw)
    ok = np.isfinite(rp).all(axis=1)
    np.testing.assert_allclose(p[pick][ok], rp[ok], rtol=2e-7, atol=1e-13)
    np.testing.assert_allclose(lm[pick][ok], rlm[ok], rtol=1e-9); np.testing.assert_allclose(le[pick][ok], rle[ok], **EVID)
    fin = np.isfinite(p).all(axis=1)
    assert fin.mean() > 0.99 and np.abs(p[fin].sum(axis=1) - 1).max() < 1e-9
