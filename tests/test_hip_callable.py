"""GPU parity of the plugin hook with a USER callable (bruteforce.py:193-194, knn.py:375-377;
demos/2 cell 69-71's ``lprob_bpz``): the callable runs on the host per object exactly as the
reference calls it, the softmax / KDE half runs on the GPU.  Golden g8 / g9 hold the outputs of
the reference's own BruteForce / NearestNeighbors loops driven by the same hooks."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-9, atol=1e-11):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def dicts():
    from frankenz_amd import PDFDict
    return PDFDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))


def bpz_like_hook(table, edges, **likekw):
    """the hook of tests/golden/make_golden.py g8, with this package's loglike in place of the
    reference's: ln-prior row picked by the object's magnitude in a reference band"""
    from frankenz_amd import pdf as hp

    def lprob(x, xe, xm, ys, yes, yms):
        res = hp.loglike(x, xe, xm, ys, yes, yms, **likekw)
        lnlike, ndim, chi2 = res[:3]
        mag = -2.5 * np.log10(max(x[1], 1e-3))          # x is already cleaned in place
        lnprior = table[int(np.searchsorted(edges, mag))]
        return lnprior, lnlike, lnlike + lnprior, ndim, chi2
    return lprob


@pytest.mark.parametrize('tag,kw', [('A', {}), ('B', {'free_scale': True, 'ignore_model_err': True})])
def test_bruteforce_user_callable_matches_reference_hook(tag, kw):
    from frankenz_amd import BruteForce
    g = load_golden('g8_prior_hook')
    d = dicts()
    edges = np.array([-1.5, -0.5, 0.3, 1.0, 2.0])
    hook = bpz_like_hook(g['table'], edges, **kw)
    bf = BruteForce(g['Y'], g['Ye'], g['Ym'])
    bf.fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), lprob_func=hook, verbose=False)
    np.testing.assert_array_equal(bf.fit_lnprior, g[tag + '_lnprior'])
    fin = np.isfinite(g[tag + '_lnprob'])
    assert not np.isfinite(bf.fit_lnprob[~fin]).any()
    close(bf.fit_lnprob[fin], g[tag + '_lnprob'][fin])
    ok = ~(np.isnan(bf.fit_lnprob).any(axis=1) | np.isnan(g[tag + '_lnprob']).any(axis=1))
    p, (lm, le) = bf.predict(g['z'], g['ze'], label_dict=d, return_gof=True, verbose=False)
    close(p[ok], g[tag + '_pred'][ok], rtol=1e-8, atol=1e-13); close(lm[ok], g[tag + '_lmap'][ok]); close(le[ok], g[tag + '_levid'][ok])
    # fused call, dictionary and direct KDE, with and without stored fits; generator twin
    for save_fits in (False, True):
        p = BruteForce(g['Y'], g['Ye'], g['Ym']).fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'],
                                                             lprob_func=hook, label_dict=d, verbose=False, save_fits=save_fits)
        close(p[ok], g[tag + '_fp'][ok], rtol=1e-8, atol=1e-13)
    p = BruteForce(g['Y'], g['Ye'], g['Ym']).fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'],
                                                         lprob_func=hook, label_grid=d.grid, verbose=False, save_fits=False)
    close(p[ok], g[tag + '_fp_grid'][ok], rtol=1e-8, atol=1e-13)
    rows = list(BruteForce(g['Y'], g['Ye'], g['Ym'])._fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'],
                                                                   lprob_func=hook, label_dict=d, save_fits=False))
    close(np.array([r[0] for r in rows])[ok], g[tag + '_fp'][ok], rtol=1e-8, atol=1e-13)
    res = list(BruteForce(g['Y'], g['Ye'], g['Ym'])._fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), lprob_func=hook))
    assert len(res) == len(g['X']) and len(res[0]) == 5


def test_positional_lprob_args_go_through_the_host_loop():
    """``lprob_args`` are positional arguments of the callable (bruteforce.py:193-194): with the default
    ``logprob`` they are (free_scale, ignore_model_err, dim_prior, ...) -- same planes as the keyword form."""
    from frankenz_amd import BruteForce
    g = load_golden('g8_prior_hook')
    bf1 = BruteForce(g['Y'], g['Ye'], g['Ym']); bf2 = BruteForce(g['Y'], g['Ye'], g['Ym'])
    bf1.fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), lprob_args=[True, True], verbose=False)
    bf2.fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), lprob_kwargs={'free_scale': True, 'ignore_model_err': True}, verbose=False)
    close(bf1.fit_lnprob, bf2.fit_lnprob, rtol=1e-12, atol=0); close(bf1.fit_chi2, bf2.fit_chi2, rtol=1e-12, atol=0)


@pytest.mark.parametrize('tag,kw,ts', [('A', {}, False),
                                       ('B', {'free_scale': True, 'ignore_model_err': True, 'return_scale': True}, True)])
def test_knn_user_callable_matches_reference_hook(tag, kw, ts):
    from frankenz_amd import NearestNeighbors
    from frankenz_amd import pdf as hp
    g = load_golden('g9_knn_prior_hook')
    d = dicts()

    def lnp_of(ys):
        return -0.5 * np.square((np.log(ys[:, 2]) - 1.0) / 0.7) - 0.3 * np.log(ys[:, 0])

    def hook(x, xe, xm, ys, yes, yms, **likekw):
        res = hp.loglike(x, xe, xm, ys, yes, yms, **likekw)
        lnlike, ndim, chi2 = res[:3]
        lnprior = lnp_of(ys)
        return (lnprior, lnlike, lnlike + lnprior, ndim, chi2) + tuple(res[3:])

    nn = NearestNeighbors(g['Y'], g['Ye'], g['Ym'], K=5, feature_map='identity', rstate=np.random.RandomState(1), verbose=False)
    p, (lm, le) = nn.fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'], lprob_func=hook,
                                 lprob_kwargs=kw, rstate=np.random.RandomState(2), k=4, eps=0.0, label_dict=d,
                                 return_gof=True, track_scale=ts, verbose=False)
    np.testing.assert_array_equal(nn.neighbors, g[tag + '_neighbors'])
    np.testing.assert_array_equal(nn.Nneighbors, g[tag + '_Nneighbors'])
    for nm in ('lnprior', 'lnlike', 'lnprob', 'chi2', 'scale'):
        a, b = getattr(nn, 'fit_' + nm), g[tag + '_' + nm]
        fin = np.isfinite(b)
        assert not np.isfinite(a[~fin]).any()
        close(a[fin], b[fin], rtol=1e-8, atol=1e-9)
    ok = ~np.isnan(g[tag + '_pdfs']).any(axis=1)
    close(p[ok], g[tag + '_pdfs'][ok], rtol=1e-8, atol=1e-13); close(lm[ok], g[tag + '_lmap'][ok]); close(le[ok], g[tag + '_levid'][ok])
