"""Round-4 surface: the stream-ordering contract (fz_set_producer_stream), device-resident NearestNeighbors, prepared handles on a
shared engine, results in page-locked memory, and label uploads that alternate between a dictionary and a direct grid."""
import os
import subprocess
import sys

import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import EVID

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SIG = np.array([0.873, 0.348, 0.418, 0.873, 3.476])


def problem(N, M, seed):
    rs = np.random.RandomState(seed)
    Y = rs.lognormal(1., 1., size=(M, 5)); Ye = np.tile(SIG, (M, 1)); Ym = np.ones((M, 5))
    X = Y[rs.choice(M, N)] + SIG * rs.randn(N, 5); Xe = np.tile(SIG, (N, 1)); Xm = np.ones((N, 5))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
    return Y, Ye, Ym, X, Xe, Xm, z, ze


def test_torch_side_checks_in_a_fresh_process():
    """stream contract (a call under fz_set_producer_stream returns while another stream still spins; the default drains the
    device), NearestNeighbors with device tensors / out= / query_features, prepared handles on the shared engine, and the whole
    overlapped sharded call (BruteForce and NearestNeighbors) through RCCL in a one-rank group"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(HERE, '_gpu_torch_checks.py')], capture_output=True, text=True, timeout=900, env=env)
    for marker in ('STREAM_CONTRACT_OK', 'KNN_DEVICE_RESIDENT_OK', 'PREPARED_HANDLE_OK', 'OVERLAPPED_RCCL_ONE_RANK_OK'):
        assert marker in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_dictionary_then_grid_then_dictionary_labels_on_one_engine():
    """fz_labels_upload_grid leaves ITS grid length in the context; the dictionary upload is skipped when the dictionary is
    unchanged (content key), so the dictionary labels that follow must bring the dictionary's grid length back themselves
    (701 -> 300 -> 701 and 701 -> 900 -> 701: a wrong stride would corrupt or overflow the (N, 701) result)"""
    from frankenz_amd import BruteForce, PDFDict
    grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
    pd, od = PDFDict(grid, sg), fo.KernelDict(grid, sg)
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(64, 1500, 31)
    rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od)
    bf = BruteForce(Y, Ye, Ym)
    for other in (300, 900):
        g2 = np.linspace(0., 6., other)
        p1 = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=pd, save_fits=False, verbose=False)
        pg = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_grid=g2, save_fits=False, verbose=False)
        p2, (lm, le) = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=pd, save_fits=False, verbose=False, return_gof=True)
        assert p1.shape == p2.shape == (64, 701) and pg.shape == (64, other)
        np.testing.assert_array_equal(p1, p2)
        np.testing.assert_allclose(p2, rp, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(le, rle, **EVID)
        rg, _, _ = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_grid=g2)
        np.testing.assert_allclose(pg, rg, rtol=1e-9, atol=1e-13)


def test_results_come_back_in_page_locked_memory_and_blocks_are_reused():
    """the drop-in classes return large PDF arrays in library-owned page-locked memory (fz_host_alloc): same numbers as a
    caller-owned pageable ``out=``, ordinary ndarray semantics, and a freed block serves the next call"""
    from frankenz_amd import BruteForce, PDFDict
    from frankenz_amd import engine
    pd = PDFDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(6000, 800, 32)       # 6000 x 701 x 8 B = 34 MB: above the page-locking threshold
    bf = BruteForce(Y, Ye, Ym)
    p = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=pd, save_fits=False, verbose=False)
    assert isinstance(p, np.ndarray) and p.flags.c_contiguous and p.flags.writeable and p.shape == (6000, 701)
    assert isinstance(p.base, engine._PinnedBlock)
    addr = p.ctypes.data
    out = (np.zeros((6000, 701)), np.zeros(6000), np.zeros(6000))
    bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=pd, save_fits=False, verbose=False, out=out)
    np.testing.assert_array_equal(p, out[0])
    np.testing.assert_allclose(p.sum(axis=1), 1.0, rtol=1e-12)
    view = p[10:20]                                           # a view keeps the block alive
    del p
    assert not any(b[1] == addr for b in engine._PinnedBlock._pool)
    del view
    assert any(b[1] == addr for b in engine._PinnedBlock._pool)
    p2 = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=pd, save_fits=False, verbose=False)
    assert p2.ctypes.data == addr                             # the block was reused
    np.testing.assert_array_equal(p2, out[0])


def test_large_host_call_stages_its_objects_once_and_writes_the_clean_back():
    """a host call of >= 4096 objects sends the whole object set to the device once and copies it back only if pdf.py:310-311
    rewrote something: dirty rows must still reach the caller's arrays, clean calls must leave them untouched"""
    from frankenz_amd import BruteForce, PDFDict
    grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
    pd, od = PDFDict(grid, sg), fo.KernelDict(grid, sg)
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(5000, 600, 33)
    X[7, 1] = np.nan; Xe[4090, 3] = -1.0; Xm[4999, 0] = 0.0
    Xc, Xec, Xmc = X.copy(), Xe.copy(), Xm.copy()
    p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(Xc, Xec, Xmc, z, ze, label_dict=pd, save_fits=False, verbose=False, return_gof=True)
    assert Xc[7, 1] == 0.0 and Xec[7, 1] == 1.0 and Xmc[7, 1] == 0.0
    assert Xc[4090, 3] == 0.0 and Xec[4090, 3] == 1.0 and Xmc[4090, 3] == 0.0
    rows = [0, 7, 4090, 4999]
    rx, rxe, rxm = X[rows].copy(), Xe[rows].copy(), Xm[rows].copy()
    rp, rlm, rle = fo.bruteforce_fit_predict(rx, rxe, rxm, Y, Ye, Ym, z, ze, label_dict=od)
    np.testing.assert_allclose(p[rows], rp, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(lm[rows], rlm, rtol=1e-10)
    np.testing.assert_array_equal(Xc[rows], rx); np.testing.assert_array_equal(Xec[rows], rxe); np.testing.assert_array_equal(Xmc[rows], rxm)


def test_grid_kde_with_a_very_wide_window_keeps_the_per_point_exponential():
    """direct gauss_kde on an even grid with sig_thresh = 40 and narrow label errors: the window recurrence's seed would sit beyond
    the exponential's clamp (z^2 / 2 > 700); such uploads take the per-point form (pdf.py:519-524 values, as the oracle's)"""
    from frankenz_amd import BruteForce
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(48, 900, 34)
    ze = np.full(len(z), 0.012)
    grid = np.arange(0, 7 + 1e-5, .01)
    for st in (5.0, 40.0):
        p = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_grid=grid, kde_kwargs={'sig_thresh': st},
                                              save_fits=False, verbose=False)
        rp, _, _ = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_grid=grid, kde_kwargs={'sig_thresh': st})
        np.testing.assert_allclose(p, rp, rtol=1e-9, atol=1e-13)
