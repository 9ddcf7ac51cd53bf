"""Properties that do not need an oracle pass over the whole problem, checked at
BASELINE-sized inputs on the GPU (configs[1] full size, configs[2] at 1e5 x 1e5):
normalisation, shard invariance, agreement between the fused and the materialised
routes, idempotence of the in-place clean, and oracle parity on a random sample."""
import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import EVID

pytestmark = pytest.mark.gpu
EVID64 = dict(rtol=1e-12, atol=1e-12)      # every form of k_hist forms and sums its weights in fp64 (DESIGN: precision)


def evid_tol():
    """the ln-evidence tolerance of the route the LAST call took: 1e-12 on the one-pass kernel (fp64 throughout), the looser
    EVID only where round 2's k_fused (fp32 remainder of the evidence) served the call"""
    from frankenz_amd.engine import get_engine
    return EVID64 if get_engine().last_form().startswith('k_hist') else EVID
SDSS_SIGMA = np.array([0.873, 0.348, 0.418, 0.873, 3.476])


def problem(n, m, seed=20260101, varying_errors=False, mask_frac=0.0):
    rs = np.random.RandomState(seed)
    Y = rs.lognormal(1., 1., size=(m, 5)); Ye = np.tile(SDSS_SIGMA, (m, 1)); Ym = np.ones((m, 5))
    if varying_errors:                      # per-model errors: the general mode A kernels
        Ye = Ye * np.random.RandomState(77).uniform(0.5, 1.5, size=Ye.shape)
    X = Y[rs.randint(0, m, n)] + SDSS_SIGMA * rs.standard_normal((n, 5))
    Xe = np.tile(SDSS_SIGMA, (n, 1)); Xm = np.ones((n, 5))
    if mask_frac > 0:                       # mixed chunks: split between mask-free and masked kernels
        Xm[np.random.RandomState(5).rand(n, 5) < mask_frac] = 0.0
    return Y, Ye, Ym, X, Xe, Xm, rs.uniform(0, 6, m), np.full(m, 0.05)


def dicts():
    from frankenz_amd import PDFDict
    grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
    return PDFDict(grid, sg), fo.KernelDict(grid, sg)


@pytest.mark.parametrize('kw,prob,form', [({}, {}, 'k_hist<screen>'), ({'free_scale': True, 'ignore_model_err': True}, {}, 'k_hist<exact>'),
                                          ({}, {'varying_errors': True}, 'k_hist<screen>'),
                                          ({}, {'mask_frac': 0.02}, 'k_hist<screen> (per-object band counts)'),
                                          ({}, {'varying_errors': True, 'mask_frac': 0.02}, 'k_hist<screen> (segmented models)'),
                                          ({'dim_prior': False}, {'varying_errors': True, 'mask_frac': 0.02}, 'k_fused')])
def test_config2_fused_properties_and_sample_parity(kw, prob, form):
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    n, m = 100000, 100000
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(n, m, **prob)
    d, od = dicts()
    bf = BruteForce(Y, Ye, Ym)
    Xc, Xec, Xmc = X.copy(), Xe.copy(), Xm.copy()
    p, (lm, le) = bf.fit_predict(Xc, Xec, Xmc, z, ze, label_dict=d, lprob_kwargs=kw, return_gof=True,
                                 save_fits=False, verbose=False)
    assert get_engine().last_form() == form                                      # (a silent fall-back to another kernel fails here)
    tol = evid_tol()
    assert p.shape == (n, 701) and np.isfinite(p).all() and (p >= 0).all()
    np.testing.assert_allclose(p.sum(axis=1), 1.0, rtol=0, atol=1e-12)          # normalised
    assert np.all(le >= lm - 1e-12) and np.all(le <= lm + np.log(m) + 1e-9)    # max <= logsumexp <= max + ln M
    np.testing.assert_array_equal(Xc, X)                                         # clean data untouched
    # shard invariance: any block of objects alone gives the same rows, bit for bit
    sl = slice(31337, 31337 + 5000)
    p2, (lm2, le2) = bf.fit_predict(X[sl].copy(), Xe[sl].copy(), Xm[sl].copy(), z, ze, label_dict=d,
                                    lprob_kwargs=kw, return_gof=True, save_fits=False, verbose=False)
    np.testing.assert_array_equal(lm2, lm[sl]); np.testing.assert_allclose(le2, le[sl], **tol)    # (another launch geometry: the sums run in another order)
    np.testing.assert_allclose(p2, p[sl], rtol=1e-12, atol=1e-15)               # LDS float atomics: order may differ
    # oracle on a random sample of objects against the FULL model set
    pick = np.random.RandomState(1).choice(n, 100, replace=False)
    rp, rlm, rle = fo.bruteforce_fit_predict(X[pick].copy(), Xe[pick].copy(), Xm[pick].copy(), Y, Ye, Ym, z, ze,
                                             label_dict=od, **kw)
    np.testing.assert_allclose(lm[pick], rlm, rtol=1e-9)
    np.testing.assert_allclose(le[pick], rle, **tol)
    np.testing.assert_allclose(p[pick], rp, rtol=1e-7, atol=1e-14)


def test_config1_materialised_planes_and_predict_route():
    """configs[1] at full size (1e5 x 1e4): planes, then predict() from the planes must
    reproduce the fused route; checksum-of-rows and sample parity against the oracle."""
    from frankenz_amd import BruteForce
    n, m = 100000, 10000
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(n, m, seed=7)
    d, od = dicts()
    bf = BruteForce(Y, Ye, Ym)
    bf.fit(X.copy(), Xe.copy(), Xm.copy(), verbose=False)
    assert bf.fit_lnlike.shape == (n, m) and bf.fit_Ndim.dtype == np.int64 and (bf.fit_Ndim == 5).all()
    assert np.all(bf.fit_scale == 1.0) and np.all(bf.fit_scale_err == 0.0) and np.all(bf.fit_lnprior == 0.0)
    pick = np.random.RandomState(2).choice(n, 16, replace=False)
    ref = fo.bruteforce_fit(X[pick].copy(), Xe[pick].copy(), Xm[pick].copy(), Y, Ye, Ym)
    np.testing.assert_allclose(bf.fit_lnlike[pick], ref['lnlike'], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(bf.fit_chi2[pick], ref['chi2'], rtol=1e-10, atol=1e-10)
    # checksum of checksums: row sums of chi2 against an independent float64 evaluation of
    # sum_j chi2_ij = sum_b sum_j (x_ib - y_jb)^2 / (xe_ib^2 + ye_jb^2)
    rows = pick[:4]
    want = [np.sum((X[i][None, :] - Y) ** 2 / (Xe[i][None, :] ** 2 + Ye ** 2)) for i in rows]
    np.testing.assert_allclose(bf.fit_chi2[rows].sum(axis=1), want, rtol=1e-11)
    from frankenz_amd.engine import get_engine
    p_pred, (lm1, le1) = bf.predict(z, ze, label_dict=d, return_gof=True, verbose=False)
    assert get_engine().last_form() == 'k_plane_rows'
    p_fused, (lm2, le2) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d,
                                                            return_gof=True, save_fits=False, verbose=False)
    assert get_engine().last_form() == 'k_hist<screen>'
    # fused: ln L(mode) + ln of the best weight, from a chi2 formed as sum (xs - y s)^2 (fz_hist.h, C2OP: one rounding of x s,
    # 1.1e-16 S/N on each difference) -- the stored row's maximum comes from the exact-difference form: 1e-13 apart at most here
    # (observed 8e-15; until the middle of round 4 both used the exact difference and agreed to two ulps)
    np.testing.assert_allclose(lm1, lm2, rtol=1e-13, atol=0)
    np.testing.assert_allclose(le1, le2, rtol=1e-12, atol=1e-12)                # both routes sum every weight in fp64
    np.testing.assert_allclose(p_pred, p_fused, rtol=1e-10, atol=1e-15)


def test_clean_is_idempotent_and_matches_reference_rule():
    from frankenz_amd.engine import get_engine
    rs = np.random.RandomState(4)
    n = 200000
    x = rs.randn(n, 5); xe = rs.randn(n, 5); xm = np.ones((n, 5))
    x[rs.rand(n, 5) < 0.01] = np.nan; xe[rs.rand(n, 5) < 0.01] = np.inf; x[rs.rand(n, 5) < 0.01] = -np.inf
    bad = ~(np.isfinite(x) & np.isfinite(xe) & (xe > 0))
    eng = get_engine()
    a, b, c = x.copy(), xe.copy(), xm.copy()
    eng.clean(a, b, c)
    assert np.all(a[bad] == 0) and np.all(b[bad] == 1) and np.all(c[bad] == 0)
    np.testing.assert_array_equal(a[~bad], x[~bad]); np.testing.assert_array_equal(b[~bad], xe[~bad])
    a2, b2, c2 = a.copy(), b.copy(), c.copy()
    eng.clean(a2, b2, c2)
    np.testing.assert_array_equal(a2, a); np.testing.assert_array_equal(b2, b); np.testing.assert_array_equal(c2, c)


def test_config3_full_size_one_million_objects():
    """BASELINE configs[2] at its full size, 1e6 objects x 1e5 models (the bench workload) through
    the drop-in class with host arrays: normalisation, max <= evidence <= max + ln M, shard
    invariance of a slice, oracle parity on a sample."""
    from frankenz_amd import BruteForce
    n, m = 1000000, 100000
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(n, m)
    d, od = dicts()
    bf = BruteForce(Y, Ye, Ym)
    from frankenz_amd.engine import get_engine
    p, (lm, le) = bf.fit_predict(X, Xe, Xm, z, ze, label_dict=d, return_gof=True, save_fits=False, verbose=False)
    assert get_engine().last_form() == 'k_hist<screen>'                          # the bench's kernel, fp64 throughout
    assert p.shape == (n, 701)
    s = p.sum(axis=1)
    assert np.isfinite(s).all() and np.abs(s - 1.0).max() < 1e-12 and p.min() >= 0.0
    assert np.all(le >= lm - 1e-12) and np.all(le <= lm + np.log(m) + 1e-9)
    sl = slice(777777, 777777 + 3000)
    p2, (lm2, le2) = bf.fit_predict(X[sl].copy(), Xe[sl].copy(), Xm[sl].copy(), z, ze, label_dict=d, return_gof=True,
                                    save_fits=False, verbose=False)
    np.testing.assert_array_equal(lm2, lm[sl]); np.testing.assert_allclose(le2, le[sl], **EVID64)  # (another launch geometry: the fp64 sums run in another order)
    np.testing.assert_allclose(p2, p[sl], rtol=1e-12, atol=1e-15)
    pick = np.random.RandomState(2).choice(n, 40, replace=False)
    rp, rlm, rle = fo.bruteforce_fit_predict(X[pick].copy(), Xe[pick].copy(), Xm[pick].copy(), Y, Ye, Ym, z, ze, label_dict=od)
    np.testing.assert_allclose(lm[pick], rlm, rtol=1e-9); np.testing.assert_allclose(le[pick], rle, **EVID64)
    np.testing.assert_allclose(p[pick], rp, rtol=1e-7, atol=1e-14)


def test_config4_knn_at_its_stated_shape(monkeypatch):
    """BASELINE configs[3]: KMCkNN with the reference defaults K = 25, k = 20, luptitude features, on the
    1e5-model set and 1e5 objects (knn.py:722-874) -- the regime where the screened fp32 search's admission
    bound and the LDS hash de-duplication matter.  Properties on every row; the neighbour table, fits and
    PDFs of a 60-object sample against the oracle's exact float64 search; the screened search against the
    all-fp64 search on 20 000 objects."""
    from frankenz_amd import NearestNeighbors
    n, m = 100000, 100000
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(n, m)
    Ye = 0.03 * Y + 0.1 * SDSS_SIGMA                       # a training set with its own photometric errors (demos/2 cell 73)
    d, od = dicts()
    fk = dict(skynoise=SDSS_SIGMA, zeropoints=10 ** (0.4 * 23.9))
    nn = NearestNeighbors(Y, Ye, Ym, K=25, feature_map='luptitude', fmap_kwargs=fk, rstate=np.random.RandomState(1), verbose=False)
    p, (lm, le) = nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(2), k=20, label_dict=d,
                                 return_gof=True, verbose=False)
    assert p.shape == (n, 701) and np.isfinite(p).all() and p.min() >= 0
    np.testing.assert_allclose(p.sum(axis=1), 1.0, rtol=0, atol=1e-12)
    W = 500
    assert nn.neighbors.shape == (n, W) and nn.Nneighbors.min() >= 20 and nn.Nneighbors.max() <= W
    col = np.arange(W)[None, :]
    valid = col < nn.Nneighbors[:, None]
    assert np.all(nn.neighbors[~valid] == -99) and nn.neighbors[valid].min() >= 0 and nn.neighbors[valid].max() < m
    srt = np.sort(np.where(valid, nn.neighbors, -1 - col), axis=1)
    assert np.all(np.diff(srt, axis=1) != 0)                                  # first-appearance de-duplication: no model twice in a row
    assert np.all(np.isneginf(nn.fit_lnprob[~valid])) and np.all(le >= lm - 1e-12) and np.all(le <= lm + np.log(W) + 1e-9)
    # oracle: exact float64 search over the same Monte-Carlo feature sets and query draws
    pick = np.random.RandomState(3).choice(n, 60, replace=False)
    feats = fo.knn_train(Y, Ye, 25, 'luptitude', np.random.RandomState(1), **fk)
    q = fo.knn_query_features(X, Xe, 'luptitude', np.random.RandomState(2), **fk)
    tab = fo.knn_neighbors_exact(feats, q[pick], 20)
    rp, rlm, rle, rn, rnn, rlnp = fo.knn_fit_predict(X[pick].copy(), Xe[pick].copy(), Xm[pick].copy(), Y, Ye, Ym, tab, z, ze, label_dict=od)
    np.testing.assert_array_equal(nn.Nneighbors[pick], rnn)
    np.testing.assert_array_equal(nn.neighbors[pick], rn)
    np.testing.assert_allclose(nn.fit_lnprob[pick], rlnp, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(p[pick], rp, rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(lm[pick], rlm, rtol=1e-10); np.testing.assert_allclose(le[pick], rle, rtol=1e-10)
    # screened (packed fp32 + exact re-check) search == all-fp64 search, bit for bit
    sub = slice(40000, 60000)
    nn2 = NearestNeighbors(Y, Ye, Ym, K=25, feature_map='luptitude', fmap_kwargs=fk, rstate=np.random.RandomState(1), verbose=False)
    monkeypatch.setenv('FZ_KNN_FP64', '1')
    nn2.fit(X[sub].copy(), Xe[sub].copy(), Xm[sub].copy(), rstate=_Slice(np.random.RandomState(2), X, Xe, sub), k=20, verbose=False)
    monkeypatch.delenv('FZ_KNN_FP64')
    np.testing.assert_array_equal(nn2.neighbors, nn.neighbors[sub]); np.testing.assert_array_equal(nn2.Nneighbors, nn.Nneighbors[sub])


class _Slice(object):
    """RandomState stand-in: draws the Monte-Carlo realisation of the WHOLE catalogue (the stream of the
    full call) and hands back the rows of a slice"""

    def __init__(self, rs, X, Xe, sl):
        self.draws = rs.normal(X, Xe)[sl]

    def normal(self, loc, scale):
        assert np.shape(loc) == self.draws.shape
        return self.draws


def test_config4_knn_at_its_stated_million_objects():
    """BASELINE configs[3] at its stated size: 1e6 objects x 1e5 models, K = 25, k = 20 (knn.py:722-874), ``save_fits=False``
    (no fit_* / neighbour attributes are kept, as in the reference).  Size-independent properties on every row --
    normalised finite non-negative PDFs, ln-max <= ln-evidence <= ln-max + ln(K k) -- and the first and last 2 000 objects
    against separate 2 000-object calls fed the same Monte-Carlo draws (a k-NN object's result does not depend on its
    batch: same neighbours, so the same PDFs to rounding and bit-identical ln-max) -- and 40 objects picked across the million
    against the oracle's exact float64 search."""
    from frankenz_amd import NearestNeighbors
    n, m = 1000000, 100000
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(n, m)
    Ye = 0.03 * Y + 0.1 * SDSS_SIGMA
    d, od = dicts()
    fk = dict(skynoise=SDSS_SIGMA, zeropoints=10 ** (0.4 * 23.9))
    nn = NearestNeighbors(Y, Ye, Ym, K=25, feature_map='luptitude', fmap_kwargs=fk, rstate=np.random.RandomState(1), verbose=False)
    p, (lm, le) = nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(2), k=20, label_dict=d,
                                 return_gof=True, verbose=False, save_fits=False)
    W = 500
    assert p.shape == (n, 701)
    for lo in range(0, n, 100000):                                            # blockwise: keeps the temporaries small
        sl = slice(lo, lo + 100000)
        pb = p[sl]
        assert np.isfinite(pb).all() and pb.min() >= 0
        np.testing.assert_allclose(pb.sum(axis=1), 1.0, rtol=0, atol=1e-12)
        assert np.all(le[sl] >= lm[sl] - 1e-12) and np.all(le[sl] <= lm[sl] + np.log(W) + 1e-9)
    # oracle: exact float64 search over the same Monte-Carlo feature sets and query draws, 40 objects of the million
    pick = np.random.RandomState(3).choice(n, 40, replace=False)
    feats = fo.knn_train(Y, Ye, 25, 'luptitude', np.random.RandomState(1), **fk)
    q = fo.knn_query_features(X, Xe, 'luptitude', np.random.RandomState(2), **fk)
    tab = fo.knn_neighbors_exact(feats, q[pick], 20)
    rp, rlm, rle, rn, rnn, rlnp = fo.knn_fit_predict(X[pick].copy(), Xe[pick].copy(), Xm[pick].copy(), Y, Ye, Ym, tab, z, ze, label_dict=od)
    np.testing.assert_allclose(p[pick], rp, rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(lm[pick], rlm, rtol=1e-10); np.testing.assert_allclose(le[pick], rle, rtol=1e-10)
    for sub in (slice(0, 2000), slice(n - 2000, n)):
        nn2 = NearestNeighbors(Y, Ye, Ym, K=25, feature_map='luptitude', fmap_kwargs=fk, rstate=np.random.RandomState(1), verbose=False)
        p2, (lm2, le2) = nn2.fit_predict(X[sub].copy(), Xe[sub].copy(), Xm[sub].copy(), z, ze, rstate=_Slice(np.random.RandomState(2), X, Xe, sub),
                                         k=20, label_dict=d, return_gof=True, verbose=False, save_fits=True)
        assert nn2.Nneighbors.min() >= 20 and nn2.Nneighbors.max() <= W
        np.testing.assert_allclose(p2, p[sub], rtol=1e-12, atol=1e-16); np.testing.assert_array_equal(lm2, lm[sub])
        np.testing.assert_allclose(le2, le[sub], rtol=1e-13, atol=1e-13)


def test_config5_substitute_stack_and_population_likelihood_at_catalogue_scale():
    """BASELINE configs[4] beyond the 2 000-object golden (g12): a catalogue-sized run of the same chain -- a TRAINING-SET fit
    (per-model errors, 2 % of the objects' bands unobserved), fused PDFs, the stacked n(z) = sum_i pdf_i, the population
    ln-likelihood and its per-object overlaps (samplers.py:60-80), one Gibbs assignment step (samplers.py:498-499) -- checked
    through what does not need the reference at this size: the stack and the overlaps against NumPy on the GPU's own PDFs,
    every object assigned exactly once and only where its posterior has support, shard invariance of the stack, and oracle
    parity of the PDFs on a random sample against the full training set."""
    from frankenz_amd import BruteForce, samplers
    n, m = 200000, 20000
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(n, m, seed=20261004, varying_errors=True, mask_frac=0.02)
    d, od = dicts()
    bf = BruteForce(Y, Ye, Ym)
    from frankenz_amd.engine import get_engine
    p, (lm, le) = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, return_gof=True, save_fits=False,
                                 verbose=False)
    assert get_engine().last_form() == 'k_hist<screen> (segmented models)'       # per-model errors x unobserved bands: the one-pass kernel
    assert p.shape == (n, 701) and np.isfinite(p).all()
    np.testing.assert_allclose(p.sum(axis=1), 1.0, rtol=0, atol=1e-12)
    stack = p.sum(axis=0)
    nz = stack / stack.sum()
    ll, ov = samplers.loglike_nz(nz, p, return_overlap=True)
    want = p @ nz
    np.testing.assert_allclose(ov, want, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(ll, np.sum(np.log(want)), rtol=1e-12)
    # the stack of two halves computed alone (another launch geometry) is the stack
    h = n // 2
    pa = bf.fit_predict(X[:h].copy(), Xe[:h].copy(), Xm[:h].copy(), z, ze, label_dict=d, save_fits=False, verbose=False)
    pb = bf.fit_predict(X[h:].copy(), Xe[h:].copy(), Xm[h:].copy(), z, ze, label_dict=d, save_fits=False, verbose=False)
    np.testing.assert_allclose(pa.sum(axis=0) + pb.sum(axis=0), stack, rtol=1e-11, atol=1e-12)
    # one assignment step of the hierarchical sampler: one draw per object, inside the support of pdf_i * nz
    u = np.random.RandomState(9).rand(n)
    counts, bins = samplers.nz_assign(nz, p, u=u, return_bins=True)
    assert counts.sum() == n and counts.shape == (701,)
    np.testing.assert_array_equal(np.bincount(bins, minlength=701), counts)
    assert np.all(p[np.arange(n), bins] * nz[bins] > 0)
    # oracle parity on a sample, against the FULL training set
    pick = np.random.RandomState(2).choice(n, 60, replace=False)
    rp, rlm, rle = fo.bruteforce_fit_predict(X[pick].copy(), Xe[pick].copy(), Xm[pick].copy(), Y, Ye, Ym, z, ze, label_dict=od)
    np.testing.assert_allclose(lm[pick], rlm, rtol=1e-9)
    np.testing.assert_allclose(le[pick], rle, **EVID64)
    np.testing.assert_allclose(p[pick], rp, rtol=1e-7, atol=1e-14)


@pytest.mark.parametrize('widths', [False, True])
def test_catalogue_shape_at_full_size(widths):
    """the bench's ``roofline_catalogue`` / ``roofline_catalogue_widths`` configurations at 1e5 x 1e5: per-model errors, 2 % of the object
    bands and 2 % of the model bands missing, and (widths) per-model label errors -- the segmented form of k_hist, with the class
    flushes in the second case.  Normalisation, max <= logsumexp <= max + ln M, shard invariance, and the oracle on a sample of
    objects against the full model set; rows the reference leaves undefined (a pair without a common band) must be nan here too."""
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    n, m = 100000, 100000
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(n, m, varying_errors=True, mask_frac=0.02)
    Ym[np.random.RandomState(6).rand(m, 5) < 0.02] = 0.0
    if widths:
        ze = np.random.RandomState(78).uniform(0.01, 0.1, size=m)
    d, od = dicts()
    bf = BruteForce(Y, Ye, Ym)
    with np.errstate(all='ignore'):
        p, (lm, le) = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, return_gof=True, save_fits=False, verbose=False)
        assert get_engine().last_form() == 'k_hist<screen> (segmented models' + (', many widths)' if widths else ')')
        fin = np.isfinite(p).all(axis=1)
        assert fin.mean() > 0.99 and np.isnan(p[~fin]).all()
        assert (p[fin] >= 0).all()
        np.testing.assert_allclose(p[fin].sum(axis=1), 1.0, rtol=0, atol=1e-12)
        assert np.all(le[fin] >= lm[fin] - 1e-12) and np.all(le[fin] <= lm[fin] + np.log(m) + 1e-9)
        sl = slice(4242, 4242 + 5000)
        p2, (lm2, le2) = bf.fit_predict(X[sl].copy(), Xe[sl].copy(), Xm[sl].copy(), z, ze, label_dict=d, return_gof=True, save_fits=False,
                                        verbose=False)
        f2 = fin[sl]
        np.testing.assert_array_equal(np.isfinite(p2).all(axis=1), f2)
        np.testing.assert_allclose(lm2[f2], lm[sl][f2], rtol=1e-13); np.testing.assert_allclose(le2[f2], le[sl][f2], **EVID64)
        np.testing.assert_allclose(p2[f2], p[sl][f2], rtol=1e-11, atol=1e-15)
        # the oracle on a sample: objects with and without missing bands, and up to five of the undefined ones
        rs = np.random.RandomState(2)
        masked = np.flatnonzero((Xm == 0).any(axis=1) & fin)
        pick = np.concatenate([rs.choice(n, 30, replace=False), rs.choice(masked, 25, replace=False), np.flatnonzero(~fin)[:5]])
        rp, rlm, rle = fo.bruteforce_fit_predict(X[pick].copy(), Xe[pick].copy(), Xm[pick].copy(), Y, Ye, Ym, z, ze, label_dict=od)
    ok = np.isfinite(rp).all(axis=1)
    np.testing.assert_array_equal(ok, fin[pick])
    np.testing.assert_allclose(lm[pick][ok], rlm[ok], rtol=1e-9)
    np.testing.assert_allclose(le[pick][ok], rle[ok], **EVID64)
    np.testing.assert_allclose(p[pick][ok], rp[ok], rtol=1e-7, atol=1e-14)
