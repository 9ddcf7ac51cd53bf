"""NearestNeighbors (knn.py) on the GPU vs the golden vectors (reference + SciPy KDTree)
and the oracle.  Search parity is defined as in SURVEY 8c: the exact k-NN table must
equal KDTree.query(eps=0); against the reference default eps=1e-3 everything
downstream must agree wherever the neighbour sets agree."""
import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import load_golden, SDSS_SIGMA


def dicts():
    from frankenz_amd import PDFDict
    grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
    return PDFDict(grid, sg), fo.KernelDict(grid, sg)


def fkw(fmap):
    return dict(skynoise=SDSS_SIGMA, zeropoints=10 ** (0.4 * 23.9)) if fmap == 'luptitude' else {}


@pytest.mark.parametrize('fmap', ['luptitude', 'identity'])
def test_mc_feature_sets_bit_exact(fmap):
    """host side only (no GPU): same RNG stream, same float32 roundings as knn.py:177-184."""
    from frankenz_amd import NearestNeighbors
    g = load_golden('g6_knn')
    nn = NearestNeighbors(g['Y'], g['Ye'], g['Ym'], K=5, feature_map=fmap, fmap_kwargs=fkw(fmap),
                          rstate=np.random.RandomState(1), verbose=False)
    assert len(nn.KDTrees) == 5 and nn.KDTrees[0].data.dtype == np.float32
    np.testing.assert_array_equal(np.stack([t.data for t in nn.KDTrees]), g[fmap + '_feats'])
    with pytest.raises(ValueError):
        NearestNeighbors(g['Y'], g['Ye'], g['Ym'], K=1, feature_map=lambda x, xe: (x, xe), verbose=False)


@pytest.mark.gpu
@pytest.mark.parametrize('fmap', ['luptitude', 'identity'])
def test_g6_knn_golden(fmap):
    from frankenz_amd import NearestNeighbors
    g = load_golden('g6_knn')
    d, _ = dicts()
    nn = NearestNeighbors(g['Y'], g['Ye'], g['Ym'], K=5, feature_map=fmap, fmap_kwargs=fkw(fmap),
                          rstate=np.random.RandomState(1), verbose=False)
    p, (lm, le) = nn.fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'],
                                 rstate=np.random.RandomState(2), k=4, label_dict=d, return_gof=True,
                                 verbose=False)
    # exact search == KDTree.query(eps=0), bit for bit, padding included
    np.testing.assert_array_equal(nn.neighbors, g[fmap + '_neighbors_eps0'])
    assert nn.neighbors.dtype == g[fmap + '_neighbors_eps0'].dtype
    np.testing.assert_allclose(p, g[fmap + '_pdfs_eps0'], rtol=1e-9, atol=1e-13)
    # reference default (eps=1e-3): rows whose neighbour table is identical must agree everywhere
    rows = np.where((nn.neighbors == g[fmap + '_neighbors']).all(axis=1))[0]
    assert len(rows) >= len(p) // 2
    np.testing.assert_array_equal(nn.Nneighbors[rows], g[fmap + '_Nneighbors'][rows])
    np.testing.assert_allclose(p[rows], g[fmap + '_pdfs'][rows], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(lm[rows], g[fmap + '_lmap'][rows], rtol=1e-10)
    np.testing.assert_allclose(le[rows], g[fmap + '_levid'][rows], rtol=1e-10)
    for nm in ('lnprob', 'chi2', 'Ndim'):
        a, b = getattr(nn, 'fit_' + nm)[rows], g[fmap + '_' + nm][rows]
        assert a.dtype == b.dtype
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)
    # fit() then predict() from the stored table reproduces fit_predict()
    nn.fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), rstate=np.random.RandomState(2), k=4, eps=0.0,
           verbose=False)
    np.testing.assert_array_equal(nn.neighbors, g[fmap + '_neighbors_eps0'])
    p2 = nn.predict(g['z'], g['ze'], label_dict=d, verbose=False)
    np.testing.assert_allclose(p2, g[fmap + '_pdfs_eps0'], rtol=1e-9, atol=1e-13)
    rows2 = list(nn._predict(g['z'], g['ze'], label_dict=d))
    np.testing.assert_allclose(np.array([r[0] for r in rows2]), p2, rtol=0, atol=0)


@pytest.mark.gpu
def test_knn_vs_oracle_larger():
    """K=25, k=20 (reference defaults) on a 3000-model set; oracle = exact float64 brute force."""
    from frankenz_amd import NearestNeighbors
    d, od = dicts()
    rs = np.random.RandomState(77)
    M, N, B = 3000, 50, 5
    Y = rs.lognormal(1., 1., size=(M, B)) * 30; Ye = 0.03 * Y; Ym = np.ones((M, B))
    X = Y[rs.choice(M, N)] + SDSS_SIGMA * rs.randn(N, B); Xe = np.tile(SDSS_SIGMA, (N, 1)); Xm = np.ones((N, B))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.04)
    kw = fkw('luptitude')
    nn = NearestNeighbors(Y, Ye, Ym, feature_map='luptitude', fmap_kwargs=kw, rstate=np.random.RandomState(5),
                          verbose=False)
    lk = {'free_scale': True, 'ignore_model_err': True}
    p, (lm, le) = nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(6),
                                 label_dict=d, lprob_kwargs=lk, return_gof=True, track_scale=True, verbose=False)
    feats = fo.knn_train(Y, Ye, 25, 'luptitude', np.random.RandomState(5), **kw)
    q = fo.knn_query_features(X, Xe, 'luptitude', np.random.RandomState(6), **kw)
    tab = fo.knn_neighbors_exact(feats, q, 20)
    rp, rlm, rle, rn, rnn, rlnp = fo.knn_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, tab, z, ze,
                                                     label_dict=od, **lk)
    np.testing.assert_array_equal(nn.Nneighbors, rnn)
    np.testing.assert_array_equal(nn.neighbors, rn)
    np.testing.assert_allclose(nn.fit_lnprob, rlnp, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(p, rp, rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(le, rle, rtol=1e-10)
    assert np.all(nn.fit_scale[nn.neighbors >= 0] != 1.0)


@pytest.mark.gpu
def test_knn_upper_bound_raises_like_reference():
    from frankenz_amd import NearestNeighbors
    g = load_golden('g6_knn')
    d, _ = dicts()
    nn = NearestNeighbors(g['Y'], g['Ye'], g['Ym'], K=3, feature_map='identity', rstate=np.random.RandomState(1),
                          verbose=False)
    with pytest.raises(IndexError):      # KDTree returns index Nmodel -> models[idxs] raises (knn.py:847)
        nn.fit_predict(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), g['z'], g['ze'],
                       rstate=np.random.RandomState(2), k=4, distance_upper_bound=1e-9, label_dict=d,
                       verbose=False)


@pytest.mark.gpu
@pytest.mark.parametrize('dp', [True, False])
def test_knn_mode_c_subset(dp):
    """free scale WITH model errors on the neighbour subset: the global stop rule now
    spans the object's <= K*k neighbours (knn.py:847 -> pdf.py:196-223)."""
    from frankenz_amd import NearestNeighbors
    d, od = dicts()
    rs = np.random.RandomState(31)
    M, N, B = 600, 40, 5
    Y = rs.lognormal(1., 1., size=(M, B)) * 10; Ye = Y * rs.uniform(0.01, 0.08, size=(M, B))
    Ym = np.ones((M, B)); Ym[rs.rand(M) < 0.1, 2] = 0
    X = Y[rs.choice(M, N)] * rs.lognormal(0, .4, N)[:, None] + SDSS_SIGMA * rs.randn(N, B)
    Xe = np.tile(SDSS_SIGMA, (N, 1)); Xm = np.ones((N, B))
    z = rs.uniform(0, 6, M); ze = rs.uniform(0.02, 0.1, M)
    nn = NearestNeighbors(Y, Ye, Ym, K=5, feature_map='identity', rstate=np.random.RandomState(5), verbose=False)
    lk = {'free_scale': True, 'ignore_model_err': False, 'dim_prior': dp}
    p, (lm, le) = nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(6), k=6,
                                 label_dict=d, lprob_kwargs=dict(lk, return_scale=True), return_gof=True,
                                 track_scale=True, verbose=False)
    feats = fo.knn_train(Y, Ye, 5, 'identity', np.random.RandomState(5))
    q = fo.knn_query_features(X, Xe, 'identity', np.random.RandomState(6))
    tab = fo.knn_neighbors_exact(feats, q, 6)
    rp, rlm, rle, rn, rnn, rlnp = fo.knn_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, tab, z, ze,
                                                     label_dict=od, **lk)
    np.testing.assert_array_equal(nn.neighbors, rn)
    np.testing.assert_allclose(nn.fit_lnprob, rlnp, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(p, rp, rtol=1e-7, atol=1e-13)
    np.testing.assert_allclose(le, rle, rtol=1e-9)
    assert np.all(nn.fit_chi2[nn.neighbors < 0] == np.inf) and np.all(nn.fit_scale[nn.neighbors < 0] == 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize('p', [1, np.inf])
def test_knn_other_norms_match_scipy(p):
    """lp_norm = 1 and inf against scipy's KDTree (exact, eps=0) on the same float32 feature sets."""
    from scipy.spatial import KDTree
    from frankenz_amd import NearestNeighbors
    g = load_golden('g6_knn')
    nn = NearestNeighbors(g['Y'], g['Ye'], g['Ym'], K=3, feature_map='identity', rstate=np.random.RandomState(1),
                          verbose=False)
    nn.fit(g['X'].copy(), g['Xe'].copy(), g['Xm'].copy(), rstate=np.random.RandomState(2), k=5, lp_norm=p,
           verbose=False)
    q = np.random.RandomState(2).normal(g['X'], g['Xe'])
    want = np.concatenate([KDTree(t.data, leafsize=50).query(q, k=5, eps=0, p=p)[1] for t in nn.KDTrees], axis=1)
    for i in range(len(q)):
        ids = fo.first_unique(want[i])
        np.testing.assert_array_equal(nn.neighbors[i, :len(ids)], ids)
        assert nn.Nneighbors[i] == len(ids)


@pytest.mark.gpu
def test_knn_wide_band_set():
    """24 bands -> the 32-band instantiation of the query / subset kernels."""
    from frankenz_amd import NearestNeighbors
    d, od = dicts()
    rs = np.random.RandomState(78)
    M, N, B = 1500, 30, 24
    sig = rs.uniform(0.2, 1.0, B)
    Y = rs.lognormal(1., 1., size=(M, 1)) * rs.lognormal(0., .5, size=(M, B)) * 10; Ye = 0.03 * Y; Ym = np.ones((M, B))
    X = Y[rs.choice(M, N)] + sig * rs.randn(N, B); Xe = np.tile(sig, (N, 1)); Xm = np.ones((N, B))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.04)
    nn = NearestNeighbors(Y, Ye, Ym, K=4, feature_map='identity', rstate=np.random.RandomState(5), verbose=False)
    p, (lm, le) = nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(6), k=8,
                                 label_dict=d, return_gof=True, verbose=False)
    feats = fo.knn_train(Y, Ye, 4, 'identity', np.random.RandomState(5))
    q = fo.knn_query_features(X, Xe, 'identity', np.random.RandomState(6))
    tab = fo.knn_neighbors_exact(feats, q, 8)
    rp, rlm, rle, rn, rnn, rlnp = fo.knn_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, tab, z, ze,
                                                     label_dict=od)
    np.testing.assert_array_equal(nn.neighbors, rn)
    np.testing.assert_allclose(nn.fit_lnprob, rlnp, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(p, rp, rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(le, rle, rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize('scale', [1.0, 1e4, 1e-3])
def test_screened_search_is_the_exact_search(scale, monkeypatch):
    """the packed-fp32 screen + fp64 re-check returns the same neighbour table as the all-fp64
    search (FZ_KNN_FP64=1), also with near-duplicate models, features far from the origin (where
    rounding the query to fp32 costs the most) and tiny features."""
    from frankenz_amd.engine import get_engine
    rs = np.random.RandomState(int(scale * 7) % 1000 + 3)
    K, M, F, N, k = 3, 5000, 5, 300, 20
    base = rs.normal(20.0, 1.0, size=(M, F))
    base[1000:2000] = base[:1000] + rs.normal(0, 1e-6, size=(1000, F))       # near ties at the fp32 resolution
    feats = np.stack([(base + rs.normal(0, 0.01, size=(M, F))) * scale for _ in range(K)]).astype(np.float32)
    q = (base[rs.choice(M, N)] + rs.normal(0, 0.02, size=(N, F))) * scale
    eng = get_engine()
    eng.upload_models(np.ones((M, F)), np.zeros((M, F)), np.ones((M, F)))       # the search only needs their count
    eng.knn_upload_trees(feats)
    out = {}
    for name in ('screen', 'fp64'):
        if name == 'fp64':
            monkeypatch.setenv('FZ_KNN_FP64', '1')
        idx = np.empty((N, K * k), dtype=np.int64)
        eng.knn_query(np.ascontiguousarray(q), k, np.inf, idx)
        out[name] = idx
    monkeypatch.delenv('FZ_KNN_FP64')
    np.testing.assert_array_equal(out['screen'], out['fp64'])
    # and both equal a float64 brute force on the host for a few queries
    for i in (0, 7, 123):
        for t in range(K):
            d2 = ((q[i][None, :] - feats[t].astype(np.float64)) ** 2).sum(axis=1)
            want = np.argsort(d2, kind='stable')[:k]
            got = out['screen'][i, t * k:(t + 1) * k]
            np.testing.assert_array_equal(np.sort(d2[got]), np.sort(d2[want]))


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['ties', 'bound', 'single_set', 'F3', 'F6', 'k40', 'unrelated_sets', 'ragged', 'tiny', 'F1', 'F2',
                                  'flat_feature', 'nan_query', 'clustered', 'k1', 'k7', 'k25', 'k32', 'k32_ties', 'few_models'])
def test_matrix_pipe_search_is_the_exact_search(case, monkeypatch):
    """fz_knn_mfma.h (fp32 MFMA screen, outward scan over the reachable tiles of the k-d order, (distance, index) ordered
    lists) returns the neighbour table of the all-fp64 ascending scan (FZ_KNN_FP64=1) bit for bit: exact duplicates among
    the models (ties resolved by model index, also at the k-th place), a finite distance bound (no first-tile selection
    then), one feature set, other feature counts, k > 32, feature sets that are NOT realisations of the same models,
    sizes that fill neither a tile nor a wave (the own leaf holds padding), NaN queries, clumpy data."""
    from frankenz_amd.engine import get_engine
    rs = np.random.RandomState(len(case) * 13 + 5)
    K, M, F, N, k, bound = 4, 3000, 5, 203, 20, np.inf
    if case == 'single_set': K = 1
    if case == 'F3': F = 3
    if case == 'F6': F = 6
    if case == 'k40': k = 40                                            # wave-serial LDS lists
    if case == 'k1': k = 1
    if case == 'k7': k = 7                                              # register lists of 4 x 5, partly filled
    if case == 'k25': k = 25                                            # register lists of 4 x 8
    if case in ('k32', 'k32_ties'): k = 32
    if case == 'few_models': M, k = 25, 20                              # one tile, mostly padding: fewer than k under any bar of the first-tile selection
    if case == 'ragged': M, N = 1000 + 37, 17
    if case == 'tiny': M, N, k = 10, 3, 4
    if case == 'F1': F = 1
    if case == 'F2': F = 2
    base = rs.normal(22.0, 1.0, size=(M, F))
    noise = 0.05
    if case in ('ties', 'k32_ties'):
        base[500:1500] = base[:1000]                                   # exact duplicates in every set
        noise = 0.0                                                    # ... and identical sets
    feats = np.stack([base + rs.normal(0, noise, size=(M, F)) if noise else base for _ in range(K)])
    if case == 'unrelated_sets':
        feats = np.stack([rs.permutation(base) for _ in range(K)])
    if case == 'flat_feature':
        feats[:, :, 2] = 21.5                                           # a feature without any spread (bounding box of width 0)
    if case == 'clustered':                                             # tight far-apart clumps: most tiles lie beyond every bar
        cen = rs.normal(22.0, 3.0, size=(30, F))
        base = cen[rs.randint(0, 30, M)] + rs.normal(0, 0.01, size=(M, F))
        feats = np.stack([base + rs.normal(0, 0.003, size=(M, F)) for _ in range(K)])
    feats = feats.astype(np.float32)
    q = base[rs.choice(M, N)] + rs.normal(0, 0.05, size=(N, F))
    if case == 'nan_query':
        q[3, 1] = np.nan; q[100] = np.inf
    if case in ('ties', 'k32_ties'):
        q[:50] = feats[0][rs.choice(1000, 50)].astype(np.float64)        # distance exactly 0 to two models each
    if case == 'bound':
        bound = 0.25
    eng = get_engine()
    eng.upload_models(np.ones((M, F)), np.zeros((M, F)), np.ones((M, F)))
    eng.knn_upload_trees(feats)
    out = {}
    for name in ('mfma', 'fp64'):
        if name == 'fp64':
            monkeypatch.setenv('FZ_KNN_FP64', '1')
        idx = np.empty((N, K * k), dtype=np.int64)
        eng.knn_query(np.ascontiguousarray(q), k, bound, idx)
        out[name] = idx
    monkeypatch.delenv('FZ_KNN_FP64')
    np.testing.assert_array_equal(out['mfma'], out['fp64'])
    # the host's own float64 brute force: stable argsort = ascending (distance, index)
    for i in range(0, N, 29):
        if not np.isfinite(q[i]).all():
            continue
        for t in range(K):
            d2 = ((q[i][None, :] - feats[t].astype(np.float64)) ** 2).sum(axis=1)
            want = np.argsort(d2, kind='stable')[:k]
            want = np.where(d2[want] < bound ** 2, want, M)
            got = out['mfma'][i, t * k:(t + 1) * k]
            if case in ('ties', 'k32_ties'):
                np.testing.assert_array_equal(got, want)
            else:
                np.testing.assert_array_equal(np.sort(d2[got[got < M]]), np.sort(d2[want[want < M]]))
                assert (got == M).sum() == (want == M).sum()


@pytest.mark.gpu
@pytest.mark.parametrize('M,bound', [(40000, np.inf), (200000, np.inf), (400000, np.inf), (600000, np.inf), (200000, 0.3), (600000, 0.2)])
def test_reachability_mask_at_every_group_size(M, bound, monkeypatch):
    """The scan's bit mask of reachable tiles (fz_knn_mfma.h, build_mask) is built along three routes by the size of the tile groups:
    word by word under the group mask (groups of 1-8 tiles, M <= 65 k: 40 000 here; and groups of 128+ tiles, M > 524 k: 600 000),
    and 4 / 2 / 1 reachable groups of 16 / 32 / 64 tiles per pass (1e5 in test_config4..., 200 000 and 400 000 here).  Clumpy data
    (most tiles beyond every bar), queries in and between the clumps, with and without a distance bound (a bounded search takes
    no first-tile selection): the table must be the all-fp64 scan's, bit for bit, and the host's own stable argsort on a sample."""
    from frankenz_amd.engine import get_engine
    rs = np.random.RandomState(M // 1000 + (0 if np.isinf(bound) else 7))
    K, F, N, k = 2, 5, 96, 20
    cen = rs.normal(22.0, 2.0, size=(60, F))
    base = cen[rs.randint(0, 60, M)] + rs.normal(0, 0.08, size=(M, F))
    base[1000:1200] = base[:200]                                          # exact duplicates
    feats = np.stack([base + rs.normal(0, 0.02, size=(M, F)) for _ in range(K)]).astype(np.float32)
    q = base[rs.choice(M, N)] + rs.normal(0, 0.05, size=(N, F))
    q[N // 2:] = 0.5 * (cen[rs.randint(0, 60, N - N // 2)] + cen[rs.randint(0, 60, N - N // 2)])      # between two clumps: wide balls
    eng = get_engine()
    eng.upload_models(np.ones((M, F)), np.zeros((M, F)), np.ones((M, F)))
    eng.knn_upload_trees(feats)
    out = {}
    for name in ('mfma', 'fp64'):
        with monkeypatch.context() as mp:
            if name == 'fp64':
                mp.setenv('FZ_KNN_FP64', '1')
            idx = np.empty((N, K * k), dtype=np.int64)
            eng.knn_query(np.ascontiguousarray(q), k, bound, idx)
            out[name] = idx
    np.testing.assert_array_equal(out['mfma'], out['fp64'])
    for i in (0, N // 2 - 1, N // 2, N - 1):
        for t in range(K):
            d2 = ((q[i][None, :] - feats[t].astype(np.float64)) ** 2).sum(axis=1)
            want = np.argsort(d2, kind='stable')[:k]
            want = np.where(d2[want] < bound ** 2, want, M)
            np.testing.assert_array_equal(out['mfma'][i, t * k:(t + 1) * k], want)


@pytest.mark.gpu
@pytest.mark.parametrize('switch', ['FZ_KNN_NOBOX', 'FZ_KNN_NOSORT', 'FZ_KNN_SERIAL', 'FZ_KNN_NOMFMA'])
def test_search_switches_leave_the_neighbour_table_unchanged(switch, monkeypatch):
    """The diagnostic switches of the search -- no tile / group skipping, models in storage order
    instead of k-d order (NOSORT is read at upload), wave-serial list insertion, the vector-ALU search instead of the matrix
    pipe -- change how the table is found, never the table: ordered by (distance, model index) it is the all-fp64 scan's, bit
    for bit, on clumpy data with exact duplicates (ties at the k-th place) and enough models for 3 tile groups."""
    from frankenz_amd.engine import get_engine
    rs = np.random.RandomState(2718)
    K, M, F, N, k = 3, 9000, 5, 333, 20
    cen = rs.normal(22.0, 2.0, size=(40, F))
    base = cen[rs.randint(0, 40, M)] + rs.normal(0, 0.05, size=(M, F))
    base[4000:4400] = base[:400]                                         # exact duplicates
    feats = np.stack([base + rs.normal(0, 0.02, size=(M, F)) for _ in range(K)]).astype(np.float32)
    feats[1] = feats[0]                                                   # one set identical to set 0
    q = base[rs.choice(M, N)] + rs.normal(0, 0.05, size=(N, F))
    q[:20] = feats[0][:20].astype(np.float64)                             # distance exactly 0 to two models each
    eng = get_engine()
    eng.upload_models(np.ones((M, F)), np.zeros((M, F)), np.ones((M, F)))

    def search(env):
        with monkeypatch.context() as mp:
            for e in env:
                mp.setenv(e, '1')
            eng._trees_key = None                                         # (the upload cache does not know about the environment)
            eng.knn_upload_trees(feats)
            idx = np.empty((N, K * k), dtype=np.int64)
            eng.knn_query(np.ascontiguousarray(q), k, np.inf, idx)
        return idx
    want = search(['FZ_KNN_FP64'])
    np.testing.assert_array_equal(search([]), want)
    np.testing.assert_array_equal(search([switch]), want)
    eng._trees_key = None


@pytest.mark.gpu
@pytest.mark.parametrize('K,k,mode', [(12, 50, 'B'), (33, 64, 'A'), (12, 50, 'C'),
                                      (8, 65, 'A'), (5, 128, 'B'), (4, 129, 'A'), (16, 256, 'A'), (3, 200, 'C')])
def test_knn_more_than_512_neighbours_per_object(K, k, mode):
    """knn.py:190-193 takes any k and K: K k = 600 and 2 112 neighbour slots per object (round 3 stopped at 512) -- the subset
    kernel's de-dup table and lists are sized per launch -- against the oracle: neighbour lists in first-appearance order, padded
    fit rows, PDFs; mode C goes through the de-dup kernel + the fixed point on the subset.  k = 65 ... 256 (rounds 1-3 and the
    first half of round 4 stopped at 64 = one list entry per lane): the matrix-pipe search keeps such lists in 64-entry segments
    (rank over all segments, shift from the top segment down); K k = 4 096 is the largest table."""
    from frankenz_amd import NearestNeighbors
    d, od = dicts()
    rs = np.random.RandomState(K * 100 + k)
    M, N, B = 2500, 24, 5
    Y = rs.lognormal(1., 1., size=(M, B)) * 20; Ye = Y * rs.uniform(0.01, 0.03, size=(M, B)); Ym = np.ones((M, B))
    X = Y[rs.choice(M, N)] * rs.choice([0.5, 1.0], N)[:, None] + SDSS_SIGMA * rs.randn(N, B)
    Xe = np.tile(SDSS_SIGMA, (N, 1)); Xm = np.ones((N, B))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.04)
    lk = {'A': {}, 'B': {'free_scale': True, 'ignore_model_err': True}, 'C': {'free_scale': True}}[mode]
    nn = NearestNeighbors(Y, Ye, Ym, K=K, feature_map='identity', rstate=np.random.RandomState(5), verbose=False)
    p, (lm, le) = nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(6), k=k, label_dict=d,
                                 lprob_kwargs=lk, return_gof=True, verbose=False)
    assert nn.neighbors.shape == (N, K * k)
    feats = fo.knn_train(Y, Ye, K, 'identity', np.random.RandomState(5))
    q = fo.knn_query_features(X, Xe, 'identity', np.random.RandomState(6))
    tab = fo.knn_neighbors_exact(feats, q, k)
    rp, rlm, rle, rn, rnn, rlnp = fo.knn_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, tab, z, ze, label_dict=od, **lk)
    np.testing.assert_array_equal(nn.Nneighbors, rnn)
    np.testing.assert_array_equal(nn.neighbors, rn)
    np.testing.assert_allclose(nn.fit_lnprob, rlnp, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(p, rp, rtol=1e-7, atol=1e-13)
    np.testing.assert_allclose(le, rle, rtol=1e-9)
    np.testing.assert_allclose(lm, rlm, rtol=1e-9)


@pytest.mark.gpu
def test_knn_long_lists_are_refused_off_the_matrix_pipe():
    """k > 64 is served by the Euclidean matrix-pipe search; the other norms keep one list entry per lane and refuse loudly."""
    from frankenz_amd import NearestNeighbors
    d, _ = dicts()
    rs = np.random.RandomState(3)
    M, N, B = 600, 8, 5
    Y = rs.lognormal(1., 1., size=(M, B)) * 20; Ye = 0.02 * Y; Ym = np.ones((M, B))
    X = Y[rs.choice(M, N)] + SDSS_SIGMA * rs.randn(N, B); Xe = np.tile(SDSS_SIGMA, (N, 1)); Xm = np.ones((N, B))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.04)
    nn = NearestNeighbors(Y, Ye, Ym, K=3, feature_map='identity', rstate=np.random.RandomState(5), verbose=False)
    with pytest.raises(NotImplementedError):
        nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(6), k=100, lp_norm=1, label_dict=d,
                       verbose=False)
    with pytest.raises(NotImplementedError):
        nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(6), k=300, label_dict=d, verbose=False)
    p = nn.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(6), k=100, label_dict=d, verbose=False)
    assert np.all(np.isfinite(p)) and nn.neighbors.shape == (N, 300)
