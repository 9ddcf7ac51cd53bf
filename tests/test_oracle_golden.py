"""The oracle (oracle/frankenz_oracle.py) against EVERY golden vector generated
from the reference (tests/golden/make_golden.py).  CPU only.  This is what pins
the oracle; the HIP parity tests then compare the kernels with the oracle."""
import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import load_golden, SDSS_SIGMA

MODES = [(fs, ime, dp) for fs in (False, True) for ime in (False, True)
         for dp in (False, True)]
TIGHT = dict(rtol=1e-12, atol=1e-12)


def eq(a, b, **kw):
    kw = kw or TIGHT
    np.testing.assert_allclose(a, b, equal_nan=True, **kw)


def demo_dict():
    return fo.KernelDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))


@pytest.mark.parametrize('mi', range(8))
def test_g1_loglike_all_modes(mi):
    g = load_golden('g1_loglike')
    fs, ime, dp = MODES[mi]
    Y, Ye, Ym = g['Y'], g['Ye'], g['Ym']
    for oi in range(len(g['X'])):
        for mk, mdt in (('f', float), ('b', bool)):
            x, xe, xm = g['X'][oi].copy(), g['Xe'][oi].copy(), g['Xm'][oi].astype(mdt)
            res = fo.loglike(x, xe, xm, Y, Ye, Ym.astype(mdt), free_scale=fs,
                             ignore_model_err=ime, dim_prior=dp, return_scale=fs)
            key = 'm%d_o%d_%s' % (mi, oi, mk)
            tol = dict(rtol=1e-9, atol=1e-9) if (fs and not ime) else TIGHT
            eq(res[0], g[key + '_lnl'], **tol)
            eq(res[1], g[key + '_ndim'])
            assert np.asarray(res[1]).dtype == g[key + '_ndim'].dtype
            eq(res[2], g[key + '_chi2'], **tol)
            if fs:
                eq(res[3], g[key + '_scale'], **tol)
                eq(res[4], g[key + '_scale_err'], **tol)
            if mk == 'f':   # in-place clean is caller-visible
                eq(x, g['clean_o%d_x' % oi]); eq(xe, g['clean_o%d_xe' % oi])
                eq(xm, g['clean_o%d_xm' % oi])


def test_g1_special_values_present():
    """the fixture really exercises -inf / nan / +inf rows (not just finite)."""
    g = load_golden('g1_loglike')
    allv = np.concatenate([g[k] for k in g.files if k.endswith('_lnl')])
    assert np.isneginf(allv).any() and np.isnan(allv).any()


def test_g1_logprob_adapter():
    g = load_golden('g1_loglike')
    r = fo.logprob(g['X'][0].copy(), g['Xe'][0].copy(), g['Xm'][0].copy(),
                   g['Y'], g['Ye'], g['Ym'])
    eq(r[0], g['logprob_lnprior']); eq(r[2], g['logprob_lnprob'])
    assert len(r) == 5


def test_g2b_mode_c_iteration_counts_of_the_reference_at_the_benchmarked_size():
    """G2b: M = 10 000, the reference's outputs and the number of passes its loop at pdf.py:199 took (counted on the reference
    itself by make_golden.py): the oracle's ``return_niter`` is the reference's count, its rows the reference's rows."""
    g = load_golden('g2b_modec_10k')
    Y, Ye = g['Y'], g['Ye']
    for oi in range(2):                      # (object 2 takes 2104 / 4794 passes: the GPU test covers it)
        for tname, ltol in (('t4', 1e-4), ('t8', 1e-8)):
            r = fo.lnlike_scaled(g['X'][oi].copy(), g['Xe'][oi].copy(), np.ones(5), Y, Ye, np.ones_like(Y), ltol=ltol,
                                 return_scale=True, return_niter=True)
            k = 'o%d_%s' % (oi, tname)
            assert r[5] == int(g[k + '_niter'])
            eq(r[0], g[k + '_lnl'], rtol=0, atol=0)
            eq(r[3], g[k + '_scale'], rtol=0, atol=0)


def test_g2_mode_c_global_stop_rule():
    g = load_golden('g2_modec')
    Y, Ye, Ym = g['Y'], g['Ye'], g['Ym']
    for oi in range(3):
        for dp in (False, True):
            for tname, ltol in (('t4', 1e-4), ('t8', 1e-8)):
                r = fo.lnlike_scaled(g['X'][oi].copy(), g['Xe'][oi].copy(),
                                     g['Xm'][oi].copy(), Y, Ye, Ym,
                                     dim_prior=dp, ltol=ltol, return_scale=True,
                                     return_niter=True)
                k = 'o%d_dp%d_%s' % (oi, int(dp), tname)
                eq(r[0], g[k + '_lnl'], rtol=1e-10, atol=1e-10)
                eq(r[2], g[k + '_chi2'], rtol=1e-10, atol=1e-10)
                eq(r[3], g[k + '_scale'], rtol=1e-10, atol=1e-12)
                eq(r[4], g[k + '_scale_err'], rtol=1e-10, atol=1e-12)
                assert r[5] >= 1
    # the stop rule is global: a tile evaluated alone stops earlier and differs
    r_all = fo.lnlike_scaled(g['X'][0].copy(), g['Xe'][0].copy(), g['Xm'][0].copy(),
                             Y, Ye, Ym, dim_prior=False, return_niter=True)
    r_sub = fo.lnlike_scaled(g['X'][0].copy(), g['Xe'][0].copy(), g['Xm'][0].copy(),
                             Y[:100], Ye[:100], Ym[:100], dim_prior=False,
                             return_niter=True)
    assert r_sub[3] <= r_all[3]


def test_g3_pdfdict_tables_and_fit():
    g = load_golden('g3_pdfdict')
    d = demo_dict()
    assert d.Ngrid == int(g['Ngrid']) and d.Ndict == int(g['Ndict'])
    eq(d.grid, g['grid']); eq(d.sigma_grid, g['sigma_grid'])
    eq(d.delta, g['delta']); eq(d.dsigma, g['dsigma'])
    np.testing.assert_array_equal(d.sigma_width, g['sigma_width'])
    np.testing.assert_array_equal([len(k) for k in d.sigma_dict], g['lens'])
    nf = int(g['nfull'])
    eq(np.concatenate(d.sigma_dict[:nf]), g['kern'], rtol=1e-14, atol=0)
    eq(np.concatenate(d.sigma_dict_cdf[:nf]), g['kcdf'], rtol=1e-14, atol=0)
    eq([k.sum() for k in d.sigma_dict], g['kern_sum'], rtol=1e-13, atol=0)
    eq([k[0] for k in d.sigma_dict], g['kern_first'], rtol=1e-14, atol=0)
    eq([c[-1] for c in d.sigma_dict_cdf], g['kcdf_last'], rtol=1e-13, atol=0)
    xi, si = d.fit(g['fit_X'], g['fit_Xe'])
    np.testing.assert_array_equal(xi, g['fit_xi'])
    np.testing.assert_array_equal(si, g['fit_si'])


def test_g4_kde_variants():
    g = load_golden('g4_kde')
    d = demo_dict()
    y, ys, wt, grid = g['y'], g['ys'], g['wt'], g['grid']
    eq(fo.gauss_kde_dict(d, y=y, y_std=ys, y_wt=wt), g['dict_default'])
    eq(fo.gauss_kde_dict(d, y=y, y_std=ys, y_wt=wt, wt_thresh=0.25), g['dict_thresh25'])
    eq(fo.gauss_kde_dict(d, y=y, y_std=ys, y_wt=wt, wt_thresh=None, cdf_thresh=None),
       g['dict_nothresh'])
    eq(fo.gauss_kde_dict(d, y=y, y_std=ys, y_wt=wt, wt_thresh=None), g['dict_cdf'])
    eq(fo.gauss_kde_dict(d, y=y, y_std=ys), g['dict_unit'])
    yi, ysi = d.fit(y, ys)
    np.testing.assert_array_equal(yi, g['yi']); np.testing.assert_array_equal(ysi, g['ysi'])
    eq(fo.gauss_kde_dict(d, y_idx=yi, y_std_idx=ysi, y_wt=wt), g['dict_idx'])
    eq(fo.gauss_kde(y, ys, grid, y_wt=wt), g['kde_default'])
    eq(fo.gauss_kde(y, ys, grid, y_wt=wt, wt_thresh=0.25), g['kde_thresh25'])
    eq(fo.gauss_kde(y, ys, grid, y_wt=wt, wt_thresh=None, cdf_thresh=None), g['kde_nothresh'])
    eq(fo.gauss_kde(y, ys, grid, y_wt=wt, wt_thresh=None), g['kde_cdf'])
    eq(fo.gauss_kde(y, ys, grid, y_wt=wt, sig_thresh=3.), g['kde_sig3'])
    eq(fo.gauss_kde(g['y2'], g['ys2'], grid, y_wt=g['w2']), g['kde_tiny'])
    # strictness really is exercised: thresh25 differs from a >= rule
    sel_gt = wt > 0.25 * wt.max()
    sel_ge = wt >= 0.25 * wt.max()
    assert sel_ge.sum() == sel_gt.sum() + 2


def test_g5_bruteforce_fit_predict():
    g = load_golden('g5_bruteforce')
    d = demo_dict()
    Y, Ye, Ym, z, ze = g['Y'], g['Ye'], g['Ym'], g['z'], g['ze']
    X, Xe, Xm = g['X'].copy(), g['Xe'].copy(), g['Xm'].copy()
    fit = fo.bruteforce_fit(X, Xe, Xm, Y, Ye, Ym)
    eq(X, g['clean_X']); eq(Xe, g['clean_Xe']); eq(Xm, g['clean_Xm'])
    for nm in ('lnprior', 'lnlike', 'lnprob', 'Ndim', 'chi2', 'scale', 'scale_err'):
        eq(fit[nm], g['fitA_' + nm])
        assert fit[nm].dtype == g['fitA_' + nm].dtype
    p, lm, le = fo.bruteforce_predict(fit['lnprob'], z, ze, label_dict=d)
    eq(p, g['predA_dict']); eq(lm, g['predA_lmap']); eq(le, g['predA_levid'])
    p, _, _ = fo.bruteforce_predict(fit['lnprob'], z, ze, label_grid=d.grid)
    eq(p, g['predA_grid'])
    p, _, _ = fo.bruteforce_predict(-0.5 * fit['chi2'], z, ze, label_dict=d)
    eq(p, g['predA_logwt_chi2'])
    p, _, _ = fo.bruteforce_predict(fit['lnprob'], z, ze, label_dict=d, wt_thresh=1e-2)
    eq(p, g['predA_thresh'])
    with pytest.raises(ValueError):
        fo.bruteforce_predict(fit['lnprob'], z, ze)


FUSED = [('A', {}), ('An', {'dim_prior': False}), ('Ai', {'ignore_model_err': True}),
         ('B', {'free_scale': True, 'ignore_model_err': True}),
         ('Bn', {'free_scale': True, 'ignore_model_err': True, 'dim_prior': False}),
         ('C', {'free_scale': True, 'ignore_model_err': False}),
         ('Cn', {'free_scale': True, 'ignore_model_err': False, 'dim_prior': False})]


@pytest.mark.parametrize('tag,kw', FUSED)
def test_g5_bruteforce_fused(tag, kw):
    g = load_golden('g5_bruteforce')
    d = demo_dict()
    p, lm, le = fo.bruteforce_fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(),
                                          g['Y'], g['Ye'], g['Ym'], g['z'], g['ze'],
                                          label_dict=d, **kw)
    tol = dict(rtol=1e-9, atol=1e-12) if tag.startswith('C') else dict(rtol=1e-11, atol=1e-13)
    eq(p, g['fp%s_pdfs' % tag], **tol)
    eq(lm, g['fp%s_lmap' % tag], **tol); eq(le, g['fp%s_levid' % tag], **tol)
    if kw.get('free_scale'):
        fit = fo.bruteforce_fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(),
                                g['Y'], g['Ye'], g['Ym'], track_scale=True,
                                return_scale=True, **kw)
        eq(fit['scale'], g['fp%s_scale' % tag], **tol)
        eq(fit['scale_err'], g['fp%s_scale_err' % tag], **tol)
        eq(fit['chi2'], g['fp%s_chi2' % tag], **tol)
        eq(fit['lnprob'], g['fp%s_lnprob' % tag], **tol)


def test_g5_fused_grid_kde():
    g = load_golden('g5_bruteforce')
    d = demo_dict()
    p, _, _ = fo.bruteforce_fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(),
                                        g['Y'], g['Ye'], g['Ym'], g['z'], g['ze'],
                                        label_grid=d.grid)
    eq(p, g['fpA_grid_pdfs'], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize('fmap', ['luptitude', 'identity'])
def test_g6_knn(fmap):
    g = load_golden('g6_knn')
    d = demo_dict()
    fk = dict(skynoise=SDSS_SIGMA, zeropoints=10 ** (0.4 * 23.9)) if fmap == 'luptitude' else {}
    Y, Ye, Ym = g['Y'], g['Ye'], g['Ym']
    feats = fo.knn_train(Y, Ye, 5, fmap, np.random.RandomState(1), **fk)
    assert feats.dtype == np.float32
    np.testing.assert_array_equal(feats, g[fmap + '_feats'])     # bit-exact MC sets
    q = fo.knn_query_features(g['X'], g['Xe'], fmap, np.random.RandomState(2), **fk)
    # the same stream as ONE (N,B) draw
    q2 = fo._fmap(fmap)(np.random.RandomState(2).normal(g['X'], g['Xe']), g['Xe'], **fk)[0]
    eq(q, q2, rtol=0, atol=0)
    tab = fo.knn_neighbors_exact(feats, q, 4)
    p, lm, le, nbrs, nn, lnp = fo.knn_fit_predict(
        g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), Y, Ye, Ym, tab, g['z'], g['ze'],
        label_dict=d)
    # exact search == scipy KDTree with eps=0 (no distance ties in this fixture)
    np.testing.assert_array_equal(nbrs, g[fmap + '_neighbors_eps0'])
    eq(p, g[fmap + '_pdfs_eps0'], rtol=1e-11, atol=1e-13)
    # eps=1e-3 (reference default) may legitimately differ on near-ties: where
    # the neighbour SETS agree, everything downstream must agree
    same = [set(nbrs[i, :nn[i]]) == set(g[fmap + '_neighbors'][i, :g[fmap + '_Nneighbors'][i]])
            for i in range(len(nn))]
    same = np.array(same)
    assert same.mean() > 0.8
    rows = np.where(same & (nbrs == g[fmap + '_neighbors']).all(axis=1))[0]
    assert len(rows) > 0
    eq(p[rows], g[fmap + '_pdfs'][rows], rtol=1e-11, atol=1e-13)
    eq(lm[rows], g[fmap + '_lmap'][rows]); eq(le[rows], g[fmap + '_levid'][rows])
    eq(lnp[rows], g[fmap + '_lnprob'][rows])


def test_g7_config1_reference_mock():
    g = load_golden('g7_config1')
    d = demo_dict()
    obs, err = g['obs'], g['err']
    n = 48   # the first 48 objects (16 full PDFs + gof for all 48)
    kw = {'free_scale': True, 'ignore_model_err': True}
    mphot = g['mphot']
    p, lm, le = fo.bruteforce_fit_predict(obs[:n].copy(), err[:n].copy(), np.ones((n, 5)),
                                          mphot, np.zeros_like(mphot), np.ones_like(mphot),
                                          g['mz'], np.full(len(g['mz']), 0.03),
                                          label_dict=d, **kw)
    eq(p[:16], g['grid_pdfs16'], rtol=1e-10, atol=1e-14)
    eq(lm, g['grid_lmap'][:n], rtol=1e-11, atol=0); eq(le, g['grid_levid'][:n], rtol=1e-11, atol=0)
    fit = fo.bruteforce_fit(obs[:4].copy(), err[:4].copy(), np.ones((4, 5)), mphot,
                            np.zeros_like(mphot), np.ones_like(mphot), track_scale=True,
                            return_scale=True, **kw)
    eq(fit['lnprob'], g['grid_lnprob_rows'], rtol=1e-11, atol=1e-11)
    eq(fit['scale'], g['grid_scale_rows'], rtol=1e-11, atol=0)
    # training-set mode (default likelihood; the self-match gives chi2=0 -> -inf)
    p, lm, le = fo.bruteforce_fit_predict(obs[:n].copy(), err[:n].copy(), np.ones((n, 5)),
                                          obs, err, np.ones_like(obs), g['redshifts'],
                                          np.full(len(obs), 0.03), label_dict=d)
    eq(p[:16], g['train_pdfs16'], rtol=1e-10, atol=1e-14)
    eq(lm, g['train_lmap'][:n], rtol=1e-11, atol=0); eq(le, g['train_levid'][:n], rtol=1e-11, atol=0)


@pytest.mark.parametrize('tag,kw', [('A', {}), ('B', {'free_scale': True, 'ignore_model_err': True})])
def test_g8_prior_hook(tag, kw):
    """the additive ln-prior against the REFERENCE's lprob_func hook (demos/2 cell 69 shape)."""
    g = load_golden('g8_prior_hook')
    lp = g['table'][g['rows']]
    rf = fo.bruteforce_fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['Y'], g['Ye'], g['Ym'], lnprior=lp, **kw)
    for k in ('lnprior', 'lnlike', 'lnprob'):
        eq(rf[k], g[tag + '_' + k])
    d = demo_dict()
    p, lm, le = fo.bruteforce_predict(rf['lnprob'], g['z'], g['ze'], label_dict=d)
    eq(p, g[tag + '_pred']); eq(lm, g[tag + '_lmap']); eq(le, g[tag + '_levid'])
    p, lm, le = fo.bruteforce_fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['Y'], g['Ye'], g['Ym'],
                                          g['z'], g['ze'], label_dict=d, lnprior=lp, **kw)
    eq(p, g[tag + '_fp'])
    p, lm, le = fo.bruteforce_fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['Y'], g['Ye'], g['Ym'],
                                          g['z'], g['ze'], label_grid=d.grid, lnprior=lp, **kw)
    eq(p, g[tag + '_fp_grid'])


@pytest.mark.parametrize('tag,kw', [('A', {}), ('B', {'free_scale': True, 'ignore_model_err': True})])
def test_g9_knn_prior_hook(tag, kw):
    """k-NN variant of the lprob_func hook (prior read at each object's neighbours)."""
    g = load_golden('g9_knn_prior_hook')
    d = demo_dict()
    feats = fo.knn_train(g['Y'], g['Ye'], 5, 'identity', np.random.RandomState(1))
    q = fo.knn_query_features(g['X'], g['Xe'], 'identity', np.random.RandomState(2))
    tab = fo.knn_neighbors_exact(feats, q, 4)
    lp = np.repeat(g['row'][None, :], len(g['X']), axis=0)
    rp, rlm, rle, rn, rnn, rlnp = fo.knn_fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['Y'], g['Ye'], g['Ym'],
                                                     tab, g['z'], g['ze'], label_dict=d, lnprior=lp, **kw)
    np.testing.assert_array_equal(rn, g[tag + '_neighbors'])
    np.testing.assert_array_equal(rnn, g[tag + '_Nneighbors'])
    eq(rlnp, g[tag + '_lnprob']); eq(rp, g[tag + '_pdfs']); eq(rlm, g[tag + '_lmap']); eq(rle, g[tag + '_levid'])


@pytest.mark.parametrize('kern', ['lorentz', 'gaussian', 'tophat'])
def test_g10_pdfs_summarize(kern):
    """pdf.pdfs_summarize (pdf.py:899-1074) incl. the in-place renormalisation."""
    g = load_golden('g10_summarize')
    work = g['pdfs_in'].copy()
    res = fo.pdfs_summarize(work, g['grid'], urand=g['urand'], pkern=kern)
    flat = np.array([a for grp in res[:5] for a in grp] + [res[5]])
    eq(flat, g[kern + '_stats'])
    np.testing.assert_array_equal(work, g[kern + '_pdfs_after'])
    # the oracle's own interp restatement against numpy's
    rs = np.random.RandomState(3)
    xp = np.cumsum(rs.rand(50) * (rs.rand(50) > 0.3)); fp = rs.randn(50)
    x = np.concatenate([rs.uniform(-1, xp[-1] + 1, 200), xp[::7], [np.nan]])
    np.testing.assert_array_equal(fo.interp_rows(x, xp, fp), np.interp(x, xp, fp))


def test_g10_loglike_nz():
    g = load_golden('g10_summarize')
    norm = g['pdfs_in'] / g['pdfs_in'].sum(axis=1)[:, None]
    ll, ov = fo.loglike_nz(g['nz'], norm)
    eq(ll, g['nz_lnlike']); eq(ov, g['nz_overlap'])
    ll, ov = fo.loglike_nz(g['nz'], norm, (120, 300), 1e-4)
    eq(ll, g['nz_pair_lnlike']); eq(ov, g['nz_pair_overlap'])
    assert fo.loglike_nz(-g['nz'], norm)[0] == -np.inf


def test_g10_pdfs_resample():
    g = load_golden('g10_summarize')
    eq(fo.pdfs_resample(g['pdfs_in'].copy(), g['grid'], g['new_grid']), g['resampled'])
    eq(fo.pdfs_resample(g['pdfs_in'].copy(), g['grid'], g['new_grid'], renormalize=False, left=-1., right=2.), g['resampled_lr'])


NET_CASES = [('wt', dict(wt_thresh=1e-3), {}), ('cdf', dict(wt_thresh=None, cdf_thresh=0.05), {}),
             ('fixed', dict(wt_thresh=1e-2, track_scale=False), {'free_scale': False, 'ignore_model_err': True})]


@pytest.mark.parametrize('tag,kw,lk', NET_CASES)
def test_g11_network_map(tag, kw, lk):
    """_Network.populate_network (networks.py:244-354)."""
    g = load_golden('g11_network_map')
    r = fo.populate_network(g['nodes'], g['models'].copy(), g['models_err'].copy(), g['models_mask'].copy(), **kw, **lk)
    np.testing.assert_array_equal(r['Nmatch'], g[tag + '_Nmatch'])
    eq(r['lmap'], g[tag + '_lmap']); eq(r['levid'], g[tag + '_levid'])
    np.testing.assert_array_equal(np.concatenate([np.array(v, dtype='int') for v in r['idxs']]), g[tag + '_idxs'])
    eq(np.concatenate([np.array(v, dtype='float') for v in r['logwts']]), g[tag + '_logwts'])
    eq(np.concatenate([np.array(v, dtype='float') for v in r['scales']]), g[tag + '_scales'])
    eq(np.concatenate([np.array(v, dtype='float') for v in r['scales_err']]), g[tag + '_scales_err'])
    bmu = np.array([[j for j in range(len(g['nodes'])) if i in r['bmus'][j]][0] for i in range(len(g['models']))])
    np.testing.assert_array_equal(bmu, g[tag + '_bmu_of_model'])


@pytest.mark.parametrize('tag', ['grid', 'train'])
def test_g12_catalogue_stack(tag):
    """configs[4] substitute: the oracle on the reference-simulated SDSS-like catalogue -- sampled PDFs,
    ln-max / ln-evidence and the population ln-likelihood of the stack against the reference's outputs."""
    g = load_golden('g12_catalogue_stack')
    d = demo_dict()
    obs, err = g['obs'], g['err']
    pick = np.arange(0, 2000, 10)[:40]
    if tag == 'grid':
        Y, Ye, lab, lerr, kw = g['mphot'], np.zeros_like(g['mphot']), g['mz'], np.full(len(g['mz']), 0.03), dict(free_scale=True, ignore_model_err=True)
    else:
        Y, Ye, lab, lerr, kw = g['tr_obs'], g['tr_err'], g['tr_z'], np.full(len(g['tr_z']), 0.05), {}
    p, lm, le = fo.bruteforce_fit_predict(obs[pick].copy(), err[pick].copy(), np.ones_like(obs[pick]), Y, Ye, np.ones_like(Y), lab, lerr,
                                          label_dict=d, **kw)
    eq(p, g[tag + '_pdfs_every10'][:40], rtol=1e-9, atol=1e-14)
    eq(lm, g[tag + '_lmap'][pick], rtol=1e-11, atol=0); eq(le, g[tag + '_levid'][pick], rtol=1e-11, atol=0)
    pd = g[tag + '_pdfs_every10']
    stack = g[tag + '_stack']
    ll, ov = fo.loglike_nz(stack / stack.sum(), pd)
    eq(ov, g[tag + '_overlap_stack'][::10], rtol=1e-12, atol=0)


def test_g13_nz_assign_law_of_the_oracle():
    """the oracle's inverse-CDF draw against the reference sampler's own pvals and multinomial(1, .) counts (g13)"""
    g = load_golden('g13_nz_assign_law')
    pd, nz, pv, rc, reps = g['pdfs'], g['nz'], g['pvals'], g['counts'].astype(np.int64), int(g['reps'])
    w = pd * nz
    np.testing.assert_allclose(w / w.sum(axis=1)[:, None], pv, rtol=1e-13, atol=1e-300)
    rs = np.random.RandomState(77)
    for k in range(len(pd)):
        _, bins, _ = fo.nz_assign(nz, np.repeat(pd[k:k + 1], reps, axis=0), rs.rand(reps))
        dc = np.bincount(bins, minlength=pd.shape[1]).astype(np.int64)
        keep = (dc + rc[k]) >= 10
        stat = np.sum((dc[keep] - rc[k][keep]) ** 2 / (dc[keep] + rc[k][keep]).astype(float))
        dof = max(int(keep.sum()) - 1, 1)
        assert dc[pv[k] == 0].sum() == 0 and stat < dof + 5 * np.sqrt(2 * dof) + 10, (k, stat, dof)


def test_knn_limits_are_refused_loudly_where_the_reference_would_run():
    """knn.py:190-193 takes any k; the GPU search keeps k <= 256 (sorted lists in 64-entry segments; until the middle of round 4: 64)
    and K k <= 4096 (LDS of the subset kernel; round 3: 512): beyond that the call must fail before any work, with the limits in
    the message -- never a silent truncation."""
    from frankenz_amd import NearestNeighbors
    rs = np.random.RandomState(3)
    Y = rs.lognormal(1, 1, (300, 5)); Ye = 0.05 * Y; Ym = np.ones_like(Y)
    X = Y[:4] + 0.1
    for K, k in ((9, 257), (65, 64)):       # k > 256; K k = 4160 > 4096
        nn = NearestNeighbors(Y, Ye, Ym, K=K, feature_map='identity', rstate=np.random.RandomState(1), verbose=False)
        with pytest.raises(NotImplementedError, match='k <= 256 and K\\*k <= 4096'):
            nn.fit(X, 0.1 * np.ones_like(X), np.ones_like(X), k=k, verbose=False)
        with pytest.raises(NotImplementedError, match='k <= 256 and K\\*k <= 4096'):
            nn.fit_predict(X, 0.1 * np.ones_like(X), np.ones_like(X), np.zeros(300), np.ones(300), label_grid=np.arange(5.), k=k, verbose=False)


G14_CASES = [(r, n, d) for r in ('wt', 'cdf') for n in (0, 1) for d in (0, 1)]


def _g14_net(g, fixed=False):
    kw = dict(track_scale=False, free_scale=False, ignore_model_err=True) if fixed else {}
    return fo.populate_network(g['nodes'], g['models'].copy(), g['models_err'].copy(), g['models_mask'].copy(), **kw)


@pytest.mark.parametrize('rule,nodes_only,disc', G14_CASES)
def test_g14_network_inference(rule, nodes_only, disc):
    """_Network.fit / fit_predict / predict through a populated network (networks.py:782-936, 1130-1473): neighbour lists in the
    reference's order, the fitted ln-probabilities and the PDFs, for both thresholding rules x nodes_only x discrete."""
    g = load_golden('g14_network_inference')
    d = demo_dict()
    net = _g14_net(g)
    np.testing.assert_array_equal(net['Nmatch'], g['Nmatch'])
    rk = dict(wt_thresh=1e-3) if rule == 'wt' else dict(wt_thresh=None, cdf_thresh=0.05)
    tag = '%s_n%d_d%d' % (rule, nodes_only, disc)
    p, lm, le, lists = fo.network_fit_predict(net, g['nodes'], g['data'].copy(), g['data_err'].copy(), g['data_mask'].copy(), g['models'],
                                              g['models_err'], g['models_mask'], g['labels'], g['label_errs'], label_dict=d,
                                              nodes_only=bool(nodes_only), discrete=bool(disc), **rk)
    np.testing.assert_array_equal(np.array([len(v) for v in lists['neighbors']]), g[tag + '_Nneighbors'])
    np.testing.assert_array_equal(np.concatenate(lists['neighbors']), g[tag + '_neighbors'])
    eq(np.concatenate(lists['lnprob']), g[tag + '_lnprob']); eq(np.concatenate(lists['chi2']), g[tag + '_chi2'])
    np.testing.assert_array_equal(np.concatenate(lists['Ndim']), g[tag + '_Ndim'])
    if nodes_only:
        eq(np.concatenate(lists['scale']), g[tag + '_scale'])
    eq(lm, g[tag + '_lmap']); eq(le, g[tag + '_levid'])
    eq(p, g[tag + '_pdfs'], rtol=1e-10, atol=1e-15)
    eq(g[tag + '_pdfs_predict'], g[tag + '_pdfs'], rtol=1e-12, atol=1e-15)      # (the reference's predict() from its stored fits: the same rows)


@pytest.mark.parametrize('disc', [0, 1])
def test_g14_node_pdfs(disc):
    """_Network.get_pdfs (networks.py:413-560)"""
    g = load_golden('g14_network_inference')
    p, lm, le = fo.network_node_pdfs(_g14_net(g), g['labels'], g['label_errs'], label_dict=demo_dict(), discrete=bool(disc))
    eq(p, g['nodepdfs_d%d' % disc], rtol=1e-10, atol=1e-15); eq(lm, g['nodelmap_d%d' % disc]); eq(le, g['nodelevid_d%d' % disc])


@pytest.mark.parametrize('nodes_only', [0, 1])
def test_g14_nodes_without_models_are_left_out(nodes_only):
    """a network mapped with the fixed-scale node likelihood: node 13 matches no model and is dropped from the node fits (networks.py:873)"""
    g = load_golden('g14_network_inference')
    net = _g14_net(g, fixed=True)
    np.testing.assert_array_equal(net['Nmatch'], g['fx_Nmatch'])
    assert net['Nmatch'][13] == 0
    tag = 'fx_n%d' % nodes_only
    p, lm, le, lists = fo.network_fit_predict(net, g['nodes'], g['data'].copy(), g['data_err'].copy(), g['data_mask'].copy(), g['models'],
                                              g['models_err'], g['models_mask'], g['labels'], g['label_errs'], label_dict=demo_dict(),
                                              nodes_only=bool(nodes_only), lpnet_kwargs={'free_scale': False, 'ignore_model_err': True})
    np.testing.assert_array_equal(np.concatenate(lists['neighbors']), g[tag + '_neighbors'])
    eq(np.concatenate(lists['lnprob']), g[tag + '_lnprob']); eq(lm, g[tag + '_lmap']); eq(le, g[tag + '_levid'])
    eq(p, g[tag + '_pdfs'], rtol=1e-10, atol=1e-15)


def test_g14_grid_kde():
    g = load_golden('g14_network_inference')
    d = demo_dict()
    p, lm, le, _ = fo.network_fit_predict(_g14_net(g), g['nodes'], g['data'].copy(), g['data_err'].copy(), g['data_mask'].copy(), g['models'],
                                          g['models_err'], g['models_mask'], g['labels'], g['label_errs'], label_grid=d.grid)
    eq(p, g['grid_pdfs'], rtol=1e-10, atol=1e-15); eq(lm, g['grid_lmap']); eq(le, g['grid_levid'])
