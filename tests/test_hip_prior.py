"""GPU parity of the additive ln-prior extension: ``lprob_func =
pdf.logprob_prior(table, rows)`` against the reference's own lprob_func hook (golden g8)
and against the oracle on seeded problems."""
import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import load_golden

pytestmark = pytest.mark.gpu
SDSS5 = np.array([0.873, 0.348, 0.418, 0.873, 3.476])


def close(a, b, rtol=1e-9, atol=1e-11):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def dicts():
    from frankenz_amd import PDFDict
    grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
    return PDFDict(grid, sg), fo.KernelDict(grid, sg)


def problem(seed, N, M, B=5):
    rs = np.random.RandomState(seed)
    Y = rs.lognormal(1., 1., size=(M, B)) * 4; Ye = 0.05 * Y; Ym = np.ones((M, B))
    X = Y[rs.choice(M, N)] + SDSS5[:B] * rs.randn(N, B); Xe = np.tile(SDSS5[:B], (N, 1)); Xm = np.ones((N, B))
    z = rs.uniform(0, 6, M); ze = rs.uniform(0.02, 0.1, M)
    return rs, Y, Ye, Ym, X, Xe, Xm, z, ze


@pytest.mark.parametrize('tag,kw', [('A', {}), ('B', {'free_scale': True, 'ignore_model_err': True})])
def test_g8_prior_hook_golden(tag, kw):
    from frankenz_amd import BruteForce
    from frankenz_amd.pdf import logprob_prior
    g = load_golden('g8_prior_hook')
    d, _ = dicts()
    hook = logprob_prior(g['table'], g['rows'])
    bf = BruteForce(g['Y'], g['Ye'], g['Ym'])
    bf.fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), lprob_func=hook, lprob_kwargs=kw, verbose=False)
    np.testing.assert_array_equal(bf.fit_lnprior, g[tag + '_lnprior'])
    # one usable band + free scale: nan or -inf by rounding luck in the reference itself
    # (see test_hip_parity.test_oracle_parity_band_counts); such rows are compared as "undefined"
    fin = np.isfinite(g[tag + '_lnlike'])
    assert not np.isfinite(bf.fit_lnlike[~fin]).any()
    close(bf.fit_lnlike[fin], g[tag + '_lnlike'][fin])
    finp = np.isfinite(g[tag + '_lnprob'])
    assert not np.isfinite(bf.fit_lnprob[~finp]).any()
    close(bf.fit_lnprob[finp], g[tag + '_lnprob'][finp])
    ok = ~(np.isnan(bf.fit_lnprob).any(axis=1) | np.isnan(g[tag + '_lnprob']).any(axis=1))
    assert ok.sum() >= 15
    p, (lm, le) = bf.predict(g['z'], g['ze'], label_dict=d, return_gof=True, verbose=False)
    close(p[ok], g[tag + '_pred'][ok], rtol=1e-8, atol=1e-13); close(lm[ok], g[tag + '_lmap'][ok]); close(le[ok], g[tag + '_levid'][ok])
    close(bf.predict(g['z'], g['ze'], label_dict=d, logwt=bf.fit_lnlike, verbose=False)[ok], g[tag + '_pred_like'][ok],
          rtol=1e-8, atol=1e-13)
    for save_fits in (False, True):
        p = BruteForce(g['Y'], g['Ye'], g['Ym']).fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'],
                                                             lprob_func=hook, lprob_kwargs=kw, label_dict=d, verbose=False,
                                                             save_fits=save_fits)
        close(p[ok], g[tag + '_fp'][ok], rtol=1e-8, atol=1e-13)
    p = BruteForce(g['Y'], g['Ye'], g['Ym']).fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'],
                                                         lprob_func=hook, lprob_kwargs=kw, label_grid=d.grid, verbose=False,
                                                         save_fits=False)
    close(p[ok], g[tag + '_fp_grid'][ok], rtol=1e-8, atol=1e-13)
    # generator twins and the one-object call
    gen = list(BruteForce(g['Y'], g['Ye'], g['Ym'])._fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'],
                                                                 lprob_func=hook, lprob_kwargs=kw, label_dict=d, save_fits=False))
    close(np.array([r[0] for r in gen])[ok], g[tag + '_fp'][ok], rtol=1e-8, atol=1e-13)
    rows = list(BruteForce(g['Y'], g['Ye'], g['Ym'])._fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), lprob_func=hook,
                                                          lprob_kwargs=kw, save_fits=False))
    close(np.array([r[2] for r in rows])[finp], g[tag + '_lnprob'][finp])
    one = hook(g['X'][4].copy(), g['Xe'][4].copy(), g['Xm'][4].copy(), g['Y'], g['Ye'], g['Ym'], row=int(g['rows'][4]), **kw)
    close(one[2][finp[4]], g[tag + '_lnprob'][4][finp[4]]); np.testing.assert_array_equal(one[0], g[tag + '_lnprior'][4])


@pytest.mark.parametrize('kind', ['dense', 'shared', 'rows'])
@pytest.mark.parametrize('kw', [{}, {'dim_prior': False}, {'free_scale': True, 'ignore_model_err': True},
                                {'free_scale': True}])
def test_prior_vs_oracle(kind, kw):
    """dense (P = N), shared (P = 1) and indexed tables; all likelihood modes incl. the iterative one;
    -inf (prior 0) entries."""
    from frankenz_amd import BruteForce
    from frankenz_amd.pdf import logprob_prior
    d, od = dicts()
    N, M = 33, 800
    rs, Y, Ye, Ym, X, Xe, Xm, z, ze = problem(61, N, M)
    if kind == 'dense':
        tab = np.log(rs.dirichlet(np.full(M, 0.5), size=N)); rows = None; lp = tab
    elif kind == 'shared':
        tab = np.log(rs.dirichlet(np.full(M, 0.5), size=1)); rows = None; lp = np.repeat(tab, N, axis=0)
    else:
        tab = np.log(rs.dirichlet(np.full(M, 0.5), size=7)); rows = rs.randint(0, 7, N); lp = tab[rows]
    tab[0, 5:40] = -np.inf
    lp = tab if kind == 'dense' else (np.repeat(tab, N, axis=0) if kind == 'shared' else tab[rows])
    hook = logprob_prior(tab if kind != 'shared' else tab[0], rows)
    p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, lprob_func=hook, lprob_kwargs=kw,
                                                    label_dict=d, return_gof=True, verbose=False, save_fits=False)
    rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, lnprior=lp, **kw)
    close(p, rp, rtol=1e-7, atol=1e-13); close(lm, rlm); close(le, rle)
    kk = {'wt_thresh': None, 'cdf_thresh': 0.01}                      # CDF rule on the posterior
    p = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, lprob_func=hook, lprob_kwargs=kw,
                                          kde_kwargs=kk, label_dict=d, verbose=False, save_fits=False)
    rp, _, _ = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, lnprior=lp,
                                         kde_kwargs=kk, **kw)
    close(p, rp, rtol=1e-7, atol=1e-13)


def test_prior_large_chunk_geometries_and_fallback():
    """enough objects for the 4-objects-per-wave geometry, a device-resident table, and the
    two-pass fallback; nan / +inf prior entries poison their rows like lnlike + lnprior does."""
    from conftest import DevArray
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    from frankenz_amd.pdf import logprob_prior
    d, od = dicts()
    N, M, P = 16700, 600, 11
    rs, Y, Ye, Ym, X, Xe, Xm, z, ze = problem(62, N, M)
    tab = np.log(rs.dirichlet(np.full(M, 0.5), size=P)); rows = rs.randint(0, P, N)
    tab[3, 17] = np.nan; tab[4, 0] = np.nan; tab[5, 9] = np.inf
    lp = tab[rows]
    rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, lnprior=lp)
    assert np.isnan(rp).all(axis=1).sum() > 100 and np.isfinite(rp).all(axis=1).sum() > 100
    dev_tab = DevArray(tab); dev_rows = DevArray(rows.astype(np.int64))
    eng = get_engine()
    for hook, lim in ((logprob_prior(tab, rows), 32 << 30), (logprob_prior(dev_tab, dev_rows), 32 << 30),
                      (logprob_prior(tab, rows), 1 << 20)):
        eng.set_workspace_limit(lim)
        try:
            p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, lprob_func=hook,
                                                            label_dict=d, return_gof=True, verbose=False, save_fits=False)
        finally:
            eng.set_workspace_limit(32 << 30)
        close(p, rp, rtol=1e-7, atol=1e-13); close(lm, rlm); close(le, rle)


def test_prior_argument_errors():
    from frankenz_amd import BruteForce
    from frankenz_amd.pdf import logprob_prior
    d, _ = dicts()
    rs, Y, Ye, Ym, X, Xe, Xm, z, ze = problem(63, 12, 100)
    bf = BruteForce(Y, Ye, Ym)
    with pytest.raises(ValueError):                      # wrong model count
        bf.fit(X.copy(), Xe.copy(), Xm.copy(), lprob_func=logprob_prior(np.zeros((1, 99))), verbose=False)
    with pytest.raises(ValueError):                      # P rows, no index, P != N
        bf.fit(X.copy(), Xe.copy(), Xm.copy(), lprob_func=logprob_prior(np.zeros((5, 100))), verbose=False)
    with pytest.raises(IndexError):                      # row index out of range
        bf.fit(X.copy(), Xe.copy(), Xm.copy(), lprob_func=logprob_prior(np.zeros((5, 100)), np.full(12, 5)), verbose=False)
    with pytest.raises(NotImplementedError):             # a data-form prior takes keyword options only
        bf.fit(X.copy(), Xe.copy(), Xm.copy(), lprob_func=logprob_prior(np.zeros(100)), lprob_args=[True], verbose=False)
    # a zero prior is the plain likelihood
    p0 = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, verbose=False, save_fits=False)
    p1 = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, lprob_func=logprob_prior(np.zeros(100)), label_dict=d,
                        verbose=False, save_fits=False)
    close(p0, p1, rtol=1e-10, atol=1e-15)


@pytest.mark.parametrize('tag,kw', [('A', {}), ('B', {'free_scale': True, 'ignore_model_err': True})])
def test_g9_knn_prior_hook_golden(tag, kw):
    """NearestNeighbors + logprob_prior against the reference's own hook on the neighbour subset."""
    from frankenz_amd import NearestNeighbors
    from frankenz_amd.pdf import logprob_prior
    g = load_golden('g9_knn_prior_hook')
    d, _ = dicts()
    hook = logprob_prior(g['row'])
    ts = bool(kw.get('free_scale'))
    nn = NearestNeighbors(g['Y'], g['Ye'], g['Ym'], K=5, feature_map='identity', rstate=np.random.RandomState(1), verbose=False)
    p, (lm, le) = nn.fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'], lprob_func=hook,
                                 lprob_kwargs=dict(kw, return_scale=True) if ts else kw, rstate=np.random.RandomState(2), k=4,
                                 label_dict=d, return_gof=True, track_scale=ts, verbose=False)
    np.testing.assert_array_equal(nn.neighbors, g[tag + '_neighbors'])
    np.testing.assert_array_equal(nn.Nneighbors, g[tag + '_Nneighbors'])
    np.testing.assert_array_equal(nn.fit_lnprior, g[tag + '_lnprior'])           # padding -inf included
    for nm in ('lnlike', 'lnprob', 'chi2', 'scale'):
        close(getattr(nn, 'fit_' + nm), g[tag + '_' + nm], rtol=1e-9, atol=1e-9)
    close(p, g[tag + '_pdfs'], rtol=1e-8, atol=1e-13); close(lm, g[tag + '_lmap']); close(le, g[tag + '_levid'])
    close(nn.predict(g['z'], g['ze'], label_dict=d, verbose=False), g[tag + '_pred'], rtol=1e-8, atol=1e-13)
    # streaming (no fit arrays) and the CDF rule / iterative likelihood against the oracle
    p2 = NearestNeighbors(g['Y'], g['Ye'], g['Ym'], K=5, feature_map='identity', rstate=np.random.RandomState(1),
                          verbose=False).fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'],
                                                     lprob_func=hook, lprob_kwargs=kw, rstate=np.random.RandomState(2), k=4,
                                                     label_dict=d, verbose=False, save_fits=False)
    close(p2, g[tag + '_pdfs'], rtol=1e-8, atol=1e-13)


@pytest.mark.parametrize('kw,kk', [({'free_scale': True}, {}), ({}, {'wt_thresh': None, 'cdf_thresh': 0.02})])
def test_knn_prior_other_routes_vs_oracle(kw, kk):
    from frankenz_amd import NearestNeighbors
    from frankenz_amd.pdf import logprob_prior
    d, od = dicts()
    N, M = 40, 900
    rs, Y, Ye, Ym, X, Xe, Xm, z, ze = problem(64, N, M)
    tab = np.log(rs.dirichlet(np.full(M, 0.5), size=5)); rows = rs.randint(0, 5, N)
    nn = NearestNeighbors(Y, Ye, Ym, K=4, feature_map='identity', rstate=np.random.RandomState(5), verbose=False)
    p, (lm, le) = nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, lprob_func=logprob_prior(tab, rows), lprob_kwargs=kw,
                                 kde_kwargs=kk, rstate=np.random.RandomState(6), k=8, label_dict=d, return_gof=True, verbose=False)
    feats = fo.knn_train(Y, Ye, 4, 'identity', np.random.RandomState(5))
    q = fo.knn_query_features(X, Xe, 'identity', np.random.RandomState(6))
    nt = fo.knn_neighbors_exact(feats, q, 8)
    rp, rlm, rle, rn, rnn, rlnp = fo.knn_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, nt, z, ze, label_dict=od,
                                                     kde_kwargs=kk, lnprior=tab[rows], **kw)
    np.testing.assert_array_equal(nn.neighbors, rn)
    close(nn.fit_lnprob, rlnp, rtol=1e-8, atol=1e-8)
    close(p, rp, rtol=1e-7, atol=1e-13); close(lm, rlm); close(le, rle)
