"""GPU parity of the PDF-stack consumers: pdf.pdfs_summarize with its
fp64-MFMA risk product, and samplers.loglike_nz -- against the reference (golden g10) and the
oracle on larger seeded stacks."""
import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import load_golden

pytestmark = pytest.mark.gpu


def flat(res):
    return np.array([a for grp in res[:5] for a in grp] + [res[5]])


ROWS = ['mean', 'mean_std', 'mean_conf', 'mean_risk', 'med', 'med_std', 'med_conf', 'med_risk', 'mode', 'mode_std',
        'mode_conf', 'mode_risk', 'best', 'best_std', 'best_conf', 'best_risk', 'low95', 'low68', 'high68', 'high95', 'mc']


def check(got, want, rtol=1e-10, atol=1e-12):
    for r, name in enumerate(ROWS):
        np.testing.assert_allclose(got[r], want[r], rtol=rtol, atol=atol, err_msg=name, equal_nan=True)


@pytest.mark.parametrize('kern', ['lorentz', 'gaussian', 'tophat'])
def test_g10_summarize_golden(kern):
    from frankenz_amd.pdf import pdfs_summarize
    g = load_golden('g10_summarize')
    work = g['pdfs_in'].copy()
    res = pdfs_summarize(work, g['grid'], rstate=np.random.RandomState(10), pkern=kern)
    check(flat(res), g[kern + '_stats'])
    np.testing.assert_allclose(work, g[kern + '_pdfs_after'], rtol=1e-14, atol=0)       # renormalised in place
    work = g['pdfs_in'].copy()
    res = pdfs_summarize(work, g['grid'], renormalize=False, rstate=np.random.RandomState(10))
    check(flat(res), g['noren_stats'])
    np.testing.assert_array_equal(work, g['pdfs_in'])


def test_summarize_larger_stack_custom_kernel_and_window():
    """N not a multiple of the GEMM tile, a grid that is not the demo's, a callable kernel, a
    custom wconf_func."""
    from frankenz_amd.pdf import pdfs_summarize
    rs = np.random.RandomState(4)
    N, G = 777, 333
    grid = np.linspace(0.0, 4.0, G)
    mu = rs.uniform(0.1, 3.9, N)[:, None]; sg = rs.uniform(0.02, 0.6, N)[:, None]
    pd = np.exp(-0.5 * ((grid[None, :] - mu) / sg) ** 2) + 0.3 * np.exp(-0.5 * ((grid[None, :] - (4 - mu)) / (0.5 * sg)) ** 2)
    pd[rs.rand(N, G) < 0.2] = 0.0                                     # plateaus in the CDFs
    pd[5] = 0.0; pd[5, 17] = 2.0                                       # single-bin PDF
    kern = lambda x: np.exp(-np.abs(x))
    wfun = lambda p: 0.02 + 0.05 * p
    u = np.random.RandomState(9).rand(N)
    a = pd.copy(); b = pd.copy()
    got = flat(pdfs_summarize(a, grid, rstate=np.random.RandomState(9), pkern=kern, wconf_func=wfun))
    # oracle with the same kernel and window (wconf passed through explicit widths)
    res = fo.pdfs_summarize(b, grid, urand=u, pkern=kern)
    want = flat(res)
    cdfs = b.cumsum(axis=1)
    for e in range(4):
        est = want[4 * e]
        w = wfun(est)
        want[4 * e + 2] = [fo.interp_rows([est[i] + w[i]], grid, cdfs[i])[0] - fo.interp_rows([est[i] - w[i]], grid, cdfs[i])[0]
                           for i in range(N)]
    # "best" is an arg-min over a GEMM row: equal up to summation order, so compare where it is decisive
    risk = np.dot(b, fo.loss_matrix(grid, kern))
    srt = np.sort(risk, axis=1)
    decisive = (srt[:, 1] - srt[:, 0]) > 1e-9 * np.abs(srt[:, 0])
    assert decisive.mean() > 0.9
    check(got[:, decisive], want[:, decisive], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(a, b, rtol=1e-14, atol=0)


@pytest.mark.parametrize('G', [4800, 6000, 11000])
def test_summarize_on_grids_beyond_4096_points(G):
    """pdf.py:899-1074 takes any grid; rounds 1-3 stopped at 4 096 points (four CDF rows per block in LDS).  Two / one rows per
    block serve up to 9 600 / 19 200 points: the statistics of sharp and of broad PDFs against the oracle."""
    from frankenz_amd.pdf import pdfs_summarize
    rs = np.random.RandomState(G)
    N = 37
    grid = np.linspace(0.0, 6.0, G)
    mu = rs.uniform(0.3, 5.7, N)[:, None]; sg = rs.uniform(0.01, 0.8, N)[:, None]
    pd = np.exp(-0.5 * ((grid[None, :] - mu) / sg) ** 2) + 0.2 * np.exp(-0.5 * ((grid[None, :] - (6 - mu)) / (0.3 * sg)) ** 2)
    u = np.random.RandomState(9).rand(N)
    a, b = pd.copy(), pd.copy()
    got = flat(pdfs_summarize(a, grid, rstate=np.random.RandomState(9)))
    want = flat(fo.pdfs_summarize(b, grid, urand=u))
    risk = np.dot(b, fo.loss_matrix(grid, 'lorentz'))
    srt = np.sort(risk, axis=1)
    decisive = (srt[:, 1] - srt[:, 0]) > 1e-9 * np.abs(srt[:, 0])
    assert decisive.mean() > 0.8
    check(got[:, decisive], want[:, decisive], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(a, b, rtol=1e-14, atol=0)
    with pytest.raises(NotImplementedError):
        pdfs_summarize(np.ones((2, 20000)), np.linspace(0, 1, 20000))


def test_loglike_nz_golden_and_device_stack():
    from conftest import DevArray
    from frankenz_amd.samplers import loglike_nz
    g = load_golden('g10_summarize')
    norm = np.ascontiguousarray(g['pdfs_in'] / g['pdfs_in'].sum(axis=1)[:, None])
    ll, ov = loglike_nz(g['nz'], norm, return_overlap=True)
    np.testing.assert_allclose(ll, g['nz_lnlike'], rtol=1e-13); np.testing.assert_allclose(ov, g['nz_overlap'], rtol=1e-13)
    ll, ov = loglike_nz(g['nz'], norm, return_overlap=True, pair=(120, 300), pair_step=1e-4)
    np.testing.assert_allclose(ll, g['nz_pair_lnlike'], rtol=1e-13); np.testing.assert_allclose(ov, g['nz_pair_overlap'], rtol=1e-13)
    assert loglike_nz(-g['nz'], norm) == -np.inf
    # a larger stack that stays in device memory
    rs = np.random.RandomState(6)
    big = rs.dirichlet(np.full(701, 0.2), size=5000)
    nz = big.sum(axis=0) / big.sum()
    ll = loglike_nz(nz, DevArray(big))
    np.testing.assert_allclose(ll, fo.loglike_nz(nz, big)[0], rtol=1e-13)


def test_pdfs_resample_golden():
    from frankenz_amd.pdf import pdfs_resample
    g = load_golden('g10_summarize')
    np.testing.assert_allclose(pdfs_resample(g['pdfs_in'].copy(), g['grid'], g['new_grid']), g['resampled'], rtol=1e-13, atol=0,
                               equal_nan=True)
    np.testing.assert_allclose(pdfs_resample(g['pdfs_in'].copy(), g['grid'], g['new_grid'], renormalize=False, left=-1., right=2.),
                               g['resampled_lr'], rtol=1e-14, atol=0, equal_nan=True)


@pytest.mark.parametrize('tag,kw,lk', [('wt', dict(wt_thresh=1e-3), None), ('cdf', dict(wt_thresh=None, cdf_thresh=0.05), None),
                                       ('fixed', dict(wt_thresh=1e-2, track_scale=False),
                                        {'free_scale': False, 'ignore_model_err': True})])
def test_g11_network_map_golden(tag, kw, lk):
    """networks.populate_network: node likelihoods on the GPU, lists as the reference builds them."""
    from frankenz_amd.networks import populate_network
    g = load_golden('g11_network_map')
    r = populate_network(g['nodes'], g['models'].copy(), g['models_err'].copy(), g['models_mask'].copy(), lpnet_kwargs=lk, **kw)
    np.testing.assert_array_equal(r.nodes_Nmatch, g[tag + '_Nmatch'])
    np.testing.assert_allclose(r.models_lmap, g[tag + '_lmap'], rtol=1e-10); np.testing.assert_allclose(r.models_levid, g[tag + '_levid'], rtol=1e-10)
    np.testing.assert_array_equal(np.concatenate([np.array(v, dtype='int') for v in r.nodes_idxs]), g[tag + '_idxs'])
    cat = lambda L: np.concatenate([np.array(v, dtype='float') for v in L])
    np.testing.assert_allclose(cat(r.nodes_logwts), g[tag + '_logwts'], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(cat(r.nodes_scales), g[tag + '_scales'], rtol=1e-10)
    np.testing.assert_allclose(cat(r.nodes_scales_err), g[tag + '_scales_err'], rtol=1e-10)
    bmu = np.array([[j for j in range(len(g['nodes'])) if i in r.nodes_bmus[j]][0] for i in range(len(g['models']))])
    np.testing.assert_array_equal(bmu, g[tag + '_bmu_of_model'])


@pytest.mark.gpu
def test_nz_assign_categorical_draw():
    """SURVEY 8f-2, samplers.py:498-499 / 519-520: the per-object redshift assignment of the Gibbs sweeps.
    (i) exact: bins and counts equal the oracle's NumPy inverse CDF on the same uniforms (a bin may differ
    only where the target sits within rounding of a CDF edge -- across a plateau of empty bins, then); (ii) distribution: with many uniforms per
    object the bin frequencies follow p * nz / dot(p, nz), the pvals of the reference's multinomial(1, .);
    edge rows: all mass in one bin, a zero row (-1), u = 0 and u -> 1."""
    from frankenz_amd import samplers
    g = load_golden('g10_summarize')
    pdfs = np.ascontiguousarray(g['pdfs_in'], dtype=np.float64)
    pdfs = pdfs / pdfs.sum(axis=1)[:, None]
    N, G = pdfs.shape
    rs = np.random.RandomState(8)
    nz = rs.dirichlet(np.full(G, 0.7))
    reps = 400
    big = np.tile(pdfs, (reps, 1))
    u = rs.rand(len(big))
    u[:N] = 0.0; u[N:2 * N] = 1.0 - 2.0 ** -53
    big[-1] = 0.0                                            # a row without mass
    counts, bins = samplers.nz_assign(nz, big, u=u, return_bins=True)
    rc, rb, cdf = fo.nz_assign(nz, big, u)
    diff = np.flatnonzero(bins != rb)
    for i in diff:                                           # only at a CDF edge within rounding
        t = u[i] * cdf[i, -1]
        lo_, hi_ = min(bins[i], rb[i]), max(bins[i], rb[i])
        assert abs(cdf[i, lo_] - t) <= 1e-12 * cdf[i, -1] and cdf[i, hi_ - 1] - cdf[i, lo_] <= 1e-12 * cdf[i, -1]
    assert len(diff) <= 2 * N + 2 and bins[-1] == -1 and counts.sum() == len(big) - 1
    np.testing.assert_array_equal(counts, np.bincount(bins[bins >= 0], minlength=G))
    assert np.abs(counts - rc).sum() <= 2 * len(diff)
    w = pdfs * nz
    assert np.all(w[np.arange(N), bins[:N]] > 0) and np.all(w[np.arange(N), bins[N:2 * N]] > 0)      # u = 0 / u -> 1 land on mass
    # distribution level: frequencies of object k's draws against its pvals (chi-square, 5-sigma bound)
    pv = w / w.sum(axis=1)[:, None]
    for k in (0, 3, N - 2):
        b = bins[2 * N + k::N][:reps - 2]
        obs = np.bincount(b, minlength=G).astype(float)
        exp = pv[k] * len(b)
        keep = exp > 5
        chi2 = np.sum((obs[keep] - exp[keep]) ** 2 / exp[keep])
        dof = keep.sum()
        assert obs[pv[k] == 0].sum() == 0 and (dof == 0 or chi2 < dof + 5 * np.sqrt(2 * dof) + 10)
    single = np.zeros((3, G)); single[:, 350] = 1.0
    c1 = samplers.nz_assign(np.full(G, 1. / G), single, u=np.array([0., 0.5, 0.999999]))
    assert c1[350] == 3 and c1.sum() == 3
    with pytest.raises(ValueError):
        samplers.nz_assign(nz, pdfs, u=np.full(N, 1.0))


def _two_sample_chi2(a, b):
    """chi-square statistic and degrees of freedom of two equal-size samples of one multinomial law"""
    keep = (a + b) >= 10
    stat = np.sum((a[keep] - b[keep]) ** 2 / (a[keep] + b[keep]).astype(float))
    return stat, max(int(keep.sum()) - 1, 1), (a[~keep].sum(), b[~keep].sum())


def test_nz_assign_law_against_the_reference_draws():
    """golden g13: the reference's own sampler (samplers.py:498-499) ran 12 500 ``multinomial(1, pvals)`` draws per object
    for 8 objects.  (i) the weights the device draws from, p * nz / dot(p, nz), equal the reference's ``pvals`` rows;
    (ii) 12 500 device draws per object (one uniform each) follow the same law: two-sample chi-square against the
    reference's bin counts inside a 5-sigma bound, no draw in a bin the reference gives no mass."""
    from frankenz_amd import samplers
    g = load_golden('g13_nz_assign_law')
    pd, nz, pv, rc, reps = g['pdfs'], g['nz'], g['pvals'], g['counts'].astype(np.int64), int(g['reps'])
    w = pd * nz
    np.testing.assert_allclose(w / w.sum(axis=1)[:, None], pv, rtol=1e-13, atol=1e-300)
    K, G = pd.shape
    big = np.repeat(pd, reps, axis=0)
    u = np.random.RandomState(1313).rand(len(big))
    counts, bins = samplers.nz_assign(nz, big, u=u, return_bins=True)
    assert counts.sum() == len(big) and (bins >= 0).all()
    for k in range(K):
        dc = np.bincount(bins[k * reps:(k + 1) * reps], minlength=G).astype(np.int64)
        assert dc[pv[k] == 0].sum() == 0
        stat, dof, rare = _two_sample_chi2(dc, rc[k])
        assert stat < dof + 5 * np.sqrt(2 * dof) + 10, (k, stat, dof)
        assert abs(int(rare[0]) - int(rare[1])) < 10 + 5 * np.sqrt(rare[0] + rare[1] + 1)


@pytest.mark.gpu
@pytest.mark.parametrize('G', [701, 1536, 1537, 2500])
def test_nz_assign_row_staging_and_long_grids(G):
    """k_nz_assign keeps the weighted row in LDS for grids up to 1 536 points and reads the lanes' runs from memory beyond: the same
    bins as the oracle's inverse CDF on the same uniforms either way (a bin may differ only at a CDF edge within rounding)"""
    from frankenz_amd import samplers
    rs = np.random.RandomState(G)
    N = 3000
    cen = rs.uniform(0.05, 0.95, N)[:, None] * G
    pdfs = np.exp(-0.5 * ((np.arange(G)[None, :] - cen) / rs.uniform(3, 40, N)[:, None]) ** 2)
    pdfs[pdfs < 1e-12] = 0.0
    pdfs /= pdfs.sum(axis=1)[:, None]
    nz = rs.dirichlet(np.full(G, 0.9))
    u = rs.rand(N)
    counts, bins = samplers.nz_assign(nz, pdfs, u=u, return_bins=True)
    rc, rb, cdf = fo.nz_assign(nz, pdfs, u)
    diff = np.flatnonzero(bins != rb)
    for i in diff:
        t = u[i] * cdf[i, -1]
        lo_, hi_ = min(bins[i], rb[i]), max(bins[i], rb[i])
        assert abs(cdf[i, lo_] - t) <= 1e-12 * cdf[i, -1] and cdf[i, hi_ - 1] - cdf[i, lo_] <= 1e-12 * cdf[i, -1]
    assert len(diff) <= 3 and counts.sum() == N
    np.testing.assert_array_equal(counts, np.bincount(bins, minlength=G))
