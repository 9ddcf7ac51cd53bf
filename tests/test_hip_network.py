"""Inference through a trained network: frankenz_amd.networks.Network against golden G14, generated from the reference's
_Network with hand-set nodes (networks.py:244-356, 413-560, 782-936, 938-1128, 1130-1473), and against the oracle on a larger case."""
import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import load_golden

pytestmark = pytest.mark.gpu


def eq(a, b, rtol=1e-9, atol=1e-11):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def dicts():
    from frankenz_amd import PDFDict
    grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
    return PDFDict(grid, sg), fo.KernelDict(grid, sg)


def make_net(g, **kw):
    from frankenz_amd.networks import Network
    net = Network(g['models'].copy(), g['models_err'].copy(), g['models_mask'].copy())
    net.set_nodes(g['nodes'])
    net.populate_network(verbose=False, **kw)
    return net


def cat(lst, dt):
    return np.concatenate([np.asarray(v, dtype=dt) for v in lst])


@pytest.mark.parametrize('rule', ['wt', 'cdf'])
@pytest.mark.parametrize('nodes_only', [0, 1])
@pytest.mark.parametrize('disc', [0, 1])
def test_g14_fit_predict_through_the_network(rule, nodes_only, disc):
    g = load_golden('g14_network_inference')
    d, _ = dicts()
    net = make_net(g)
    np.testing.assert_array_equal(net.nodes_Nmatch, g['Nmatch'])
    rk = dict(wt_thresh=1e-3) if rule == 'wt' else dict(wt_thresh=None, cdf_thresh=0.05)
    tag = '%s_n%d_d%d' % (rule, nodes_only, disc)
    X, Xe, Xm = g['data'].copy(), g['data_err'].copy(), g['data_mask'].copy()
    with np.errstate(all='ignore'):
        pdfs, (lm, le) = net.fit_predict(X, Xe, Xm, g['labels'], g['label_errs'], label_dict=d, nodes_only=bool(nodes_only), discrete=bool(disc),
                                         return_gof=True, verbose=False, save_fits=True, track_scale=bool(nodes_only), **rk)
    np.testing.assert_array_equal(net.Nneighbors, g[tag + '_Nneighbors'])
    np.testing.assert_array_equal(cat(net.neighbors, 'int'), g[tag + '_neighbors'])       # the reference's order (pandas.unique / argsort)
    eq(cat(net.fit_lnprob, 'float'), g[tag + '_lnprob']); eq(cat(net.fit_chi2, 'float'), g[tag + '_chi2'])
    np.testing.assert_array_equal(cat(net.fit_Ndim, 'int'), g[tag + '_Ndim'])
    if nodes_only:
        eq(cat(net.fit_scale, 'float'), g[tag + '_scale'])
    eq(lm, g[tag + '_lmap']); eq(le, g[tag + '_levid'])
    eq(pdfs, g[tag + '_pdfs'], rtol=1e-8, atol=1e-14)
    # predict() from the stored fits; fit() alone stores the same
    with np.errstate(all='ignore'):
        p2, (lm2, le2) = net.predict(g['labels'], g['label_errs'], label_dict=d, return_gof=True, discrete=bool(disc), verbose=False)
    eq(p2, g[tag + '_pdfs_predict'], rtol=1e-8, atol=1e-14); eq(lm2, g[tag + '_lmap']); eq(le2, g[tag + '_levid'])
    net.fit(g['data'].copy(), g['data_err'].copy(), g['data_mask'].copy(), nodes_only=bool(nodes_only), discrete=bool(disc), verbose=False,
            track_scale=bool(nodes_only), **rk)
    np.testing.assert_array_equal(cat(net.neighbors, 'int'), g[tag + '_neighbors'])
    eq(cat(net.fit_lnprob, 'float'), g[tag + '_lnprob'])
    # the in-place clean of the objects reached the caller's arrays (pdf.py:309-311)
    assert X[2, 3] == 0.0 and Xe[2, 3] == 1.0 and Xm[2, 3] == 0.0


@pytest.mark.parametrize('disc', [0, 1])
def test_g14_node_pdfs(disc):
    g = load_golden('g14_network_inference')
    d, _ = dicts()
    net = make_net(g)
    p, (lm, le) = net.get_pdfs(g['labels'], g['label_errs'], label_dict=d, return_gof=True, discrete=bool(disc), verbose=False)
    eq(p, g['nodepdfs_d%d' % disc], rtol=1e-8, atol=1e-14); eq(lm, g['nodelmap_d%d' % disc]); eq(le, g['nodelevid_d%d' % disc])
    p1, gof = net.get_pdf(3, g['labels'], g['label_errs'], label_dict=d, return_gof=True, discrete=bool(disc))
    eq(p1, g['nodepdfs_d%d' % disc][3], rtol=1e-8, atol=1e-14)


@pytest.mark.parametrize('nodes_only', [0, 1])
def test_g14_nodes_without_models_are_left_out(nodes_only):
    g = load_golden('g14_network_inference')
    d, _ = dicts()
    net = make_net(g, track_scale=False, lpnet_kwargs={'free_scale': False, 'ignore_model_err': True})
    np.testing.assert_array_equal(net.nodes_Nmatch, g['fx_Nmatch'])
    tag = 'fx_n%d' % nodes_only
    with np.errstate(all='ignore'):
        pdfs, (lm, le) = net.fit_predict(g['data'].copy(), g['data_err'].copy(), g['data_mask'].copy(), g['labels'], g['label_errs'],
                                         label_dict=d, nodes_only=bool(nodes_only), return_gof=True, verbose=False)
    np.testing.assert_array_equal(cat(net.neighbors, 'int'), g[tag + '_neighbors'])
    eq(cat(net.fit_lnprob, 'float'), g[tag + '_lnprob']); eq(lm, g[tag + '_lmap']); eq(le, g[tag + '_levid'])
    eq(pdfs, g[tag + '_pdfs'], rtol=1e-8, atol=1e-14)


def test_g14_grid_kde_and_generators():
    g = load_golden('g14_network_inference')
    d, _ = dicts()
    net = make_net(g)
    with np.errstate(all='ignore'):
        pdfs, (lm, le) = net.fit_predict(g['data'].copy(), g['data_err'].copy(), g['data_mask'].copy(), g['labels'], g['label_errs'],
                                         label_grid=d.grid, return_gof=True, verbose=False)
        gen = list(net._fit_predict(g['data'].copy(), g['data_err'].copy(), g['data_mask'].copy(), g['labels'], g['label_errs'], label_grid=d.grid))
    eq(pdfs, g['grid_pdfs'], rtol=1e-8, atol=1e-14); eq(lm, g['grid_lmap']); eq(le, g['grid_levid'])
    eq(np.array([r[0] for r in gen]), g['grid_pdfs'], rtol=1e-8, atol=1e-14)
    with pytest.raises(ValueError):
        net.fit_predict(g['data'], g['data_err'], g['data_mask'], g['labels'], g['label_errs'])
    with pytest.raises(ValueError):
        net.get_node()


def test_user_callables_run_on_the_host_and_match_the_device_path():
    """a foreign lpnet_func / lprob_func (here: thin wrappers of the package's own logprob) goes through the host loops"""
    from frankenz_amd import pdf as fpdf
    g = load_golden('g14_network_inference')
    d, _ = dicts()
    mine = lambda *a, **k: fpdf.logprob(*a, **k)
    net = make_net(g)
    net2 = make_net(g, lpnet_func=mine)
    np.testing.assert_array_equal(net.nodes_Nmatch, net2.nodes_Nmatch)
    for a, b in zip(net.nodes_idxs, net2.nodes_idxs):
        assert a == b
    args = (g['data'].copy(), g['data_err'].copy(), g['data_mask'].copy(), g['labels'], g['label_errs'])
    with np.errstate(all='ignore'):
        p0 = net.fit_predict(*[np.copy(a) for a in args], label_dict=d, verbose=False)
        p1 = net2.fit_predict(*[np.copy(a) for a in args], label_dict=d, lprob_func=mine, verbose=False)
    eq(p0, p1, rtol=1e-9, atol=1e-14)
    np.testing.assert_array_equal(cat(net.neighbors, 'int'), cat(net2.neighbors, 'int'))


def test_a_larger_network_against_the_oracle():
    """2 000 models on 60 nodes, 300 objects (more than one wave's worth of nodes, unions of hundreds of models)"""
    from frankenz_amd.networks import Network
    d, od = dicts()
    rs = np.random.RandomState(8)
    M, N, Nn, B = 2000, 300, 60, 5
    sig = np.array([0.873, 0.348, 0.418, 0.873, 3.476])
    Y = rs.lognormal(1., 1., size=(M, 1)) * rs.lognormal(0., .4, size=(M, B)) * 5; Ye = 0.05 * Y; Ym = np.ones((M, B))
    nodes = Y[rs.choice(M, Nn, replace=False)] * rs.lognormal(0, 0.05, size=(Nn, B))
    X = Y[rs.choice(M, N)] * rs.lognormal(0, .3, N)[:, None] + sig * rs.randn(N, B); Xe = np.tile(sig, (N, 1)); Xm = np.ones((N, B))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
    net = Network(Y, Ye, Ym); net.set_nodes(nodes); net.populate_network(verbose=False)
    onet = fo.populate_network(nodes, Y.copy(), Ye.copy(), Ym.copy())
    np.testing.assert_array_equal(net.nodes_Nmatch, onet['Nmatch'])
    for kw in (dict(), dict(nodes_only=True), dict(discrete=True)):
        with np.errstate(all='ignore'):
            p, (lm, le) = net.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, return_gof=True, verbose=False, **kw)
            rp, rlm, rle, lists = fo.network_fit_predict(onet, nodes, X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, **kw)
        np.testing.assert_array_equal(net.Nneighbors, [len(v) for v in lists['neighbors']])
        np.testing.assert_array_equal(cat(net.neighbors, 'int'), np.concatenate(lists['neighbors']))
        eq(lm, rlm); eq(le, rle); eq(p, rp, rtol=1e-8, atol=1e-14)
