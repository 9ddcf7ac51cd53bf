"""Mode C (free scale WITH model errors, pdf.py:196-223) on every launch shape the library picks by model count:
``k_modec_rounds<256 / 512 / 1024>`` (mask-free tame data, M <= 16 384: several iterations per record read, round 5),
``k_modec_persist<1024,4>`` (M <= 4096), ``<768,14>`` (<= 10 752), ``<512,32>`` (<= 16 384: masked data, IEEE re-runs, ``FZ_MODEC_ROUNDS=0``)
and the state-plane kernels ``k_modec_step / _check`` beyond -- against the oracle (whose iteration counter is pinned to the
reference's by golden G2b), with per-model errors, with and without a masked band, at ltol 1e-4 and 1e-8; the switches
``FZ_MODEC_IEEE`` / ``FZ_MODEC_PLANES`` / ``FZ_MODEC_FINAL`` / ``FZ_MODEC_BURST`` / ``FZ_MODEC_ROUNDS`` must leave iteration counts and ln-likes
unchanged; and an object whose max |dlnl| lands within rounding of ltol must take the IEEE re-run (fz_inst.hip, run_modec)."""
import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import EVID64, load_golden

pytestmark = pytest.mark.gpu

SDSS5 = np.array([0.873, 0.348, 0.418, 0.873, 3.476])
KW = {'free_scale': True, 'ignore_model_err': False}
SHAPES = {6000: (1, 768), 10000: (1, 768), 14000: (1, 512), 20000: (2, 0)}      # M -> (path, threads per block) of fz_modec_info
ROUNDS = {6000: (3, 1024), 10000: (3, 1024), 14000: (3, 1024), 20000: (3, 512)}    # ... for mask-free data: k_modec_rounds (round 5)


def problem(M, N, seed, masked=False, amps=(1.0, 0.5)):
    rs = np.random.RandomState(seed)
    Y = rs.lognormal(1.0, 1.0, size=(M, 5))
    Ye = Y * rs.uniform(0.005, 0.03, size=(M, 5))         # per-model errors (heterogeneous; at G2's 1-12 % some objects need thousands of passes or never stop)
    Ym = np.ones((M, 5))
    pick = rs.choice(M, N)
    amp = np.resize(amps, N)                              # (five to several hundred passes per object)
    X = Y[pick] * amp[:, None] + SDSS5 * rs.randn(N, 5)
    Xe = np.tile(SDSS5, (N, 1)); Xm = np.ones((N, 5))
    if masked:
        Xm[::3, 4] = 0.0                                  # a masked band on every third object
        Ym[rs.choice(M, M // 50, replace=False), 1] = 0.0    # and on 2 % of the models
    return Y, Ye, Ym, X, Xe, Xm


def hip_fit(Y, Ye, Ym, X, Xe, Xm, ltol, dim_prior=True):
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    bf = BruteForce(Y, Ye, Ym)
    eng = get_engine()
    eng.timing_reset()
    bf.fit(X.copy(), Xe.copy(), Xm.copy(), lprob_kwargs=dict(KW, ltol=ltol, dim_prior=dim_prior), track_scale=True, verbose=False)
    return bf, eng.modec_niter(len(X)), eng.modec_info()


def oracle_rows(Y, Ye, Ym, X, Xe, Xm, ltol, rows, dim_prior=True):
    out = []
    for i in rows:
        out.append(fo.lnlike_scaled(X[i].copy(), Xe[i].copy(), Xm[i].copy(), Y, Ye, Ym, ltol=ltol, dim_prior=dim_prior,
                                    return_scale=True, return_niter=True, max_iter=9000))
        assert out[-1][5] < 9000
    return out


@pytest.mark.parametrize('M', sorted(SHAPES))
@pytest.mark.parametrize('masked', [False, True])
def test_mode_c_every_launch_shape_against_the_oracle(M, masked, monkeypatch):
    N = 10
    Y, Ye, Ym, X, Xe, Xm = problem(M, N, 900 + M // 1000, masked)
    for ltol in (1e-4, 1e-8):
        rows = [0, 9] if ltol == 1e-4 else [1]             # (object 9 is the slow one at M = 20 000: 521 / 805 passes)
        bf, niter, info = hip_fit(Y, Ye, Ym, X, Xe, Xm, ltol)
        assert (info[2], info[3]) == (SHAPES if masked else ROUNDS)[M], info   # the shape this test is about did run
        want = oracle_rows(Y, Ye, Ym, X, Xe, Xm, ltol, rows)
        for i, w in zip(rows, want):
            assert niter[i] == w[5], (M, ltol, i, niter[i], w[5])      # the reference's iteration count (oracle pinned by G2b)
            fin = np.isfinite(w[0])
            np.testing.assert_allclose(bf.fit_lnlike[i][fin], w[0][fin], rtol=1e-9, atol=1e-9)
            np.testing.assert_array_equal(np.isfinite(bf.fit_lnlike[i]), fin)
            np.testing.assert_allclose(bf.fit_chi2[i], w[2], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(bf.fit_scale[i], w[3], rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(bf.fit_scale_err[i], w[4], rtol=1e-9, atol=1e-11)
        assert info[1] == niter.max()
        # the switches: IEEE divisions throughout / the state-plane kernels (/ one host round trip per iteration) -- same iteration
        # count for EVERY object, ln-likes to 1e-10
        for env in ({'FZ_MODEC_IEEE': '1'}, {'FZ_MODEC_PLANES': '1'}, {'FZ_MODEC_PLANES': '1', 'FZ_MODEC_IEEE': '1', 'FZ_MODEC_BURST': '1'},
                    {'FZ_MODEC_ROUNDS': '0'}):
            if ltol == 1e-8 and 'FZ_MODEC_BURST' in env:
                continue
            with monkeypatch.context() as mp:
                for k, v in env.items():
                    mp.setenv(k, v)
                bf2, niter2, info2 = hip_fit(Y, Ye, Ym, X, Xe, Xm, ltol)
            np.testing.assert_array_equal(niter2, niter)
            if 'FZ_MODEC_PLANES' in env:
                assert info2[2] == 2
            if 'FZ_MODEC_ROUNDS' in env or 'FZ_MODEC_IEEE' in env and 'FZ_MODEC_PLANES' not in env:
                assert (info2[2], info2[3]) == SHAPES[M], info2          # one iteration per record read: k_modec_persist
            fin = np.isfinite(bf.fit_lnlike)
            np.testing.assert_array_equal(np.isfinite(bf2.fit_lnlike), fin)
            np.testing.assert_allclose(bf2.fit_lnlike[fin], bf.fit_lnlike[fin], rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(bf2.fit_scale, bf.fit_scale, rtol=1e-10, atol=1e-12)


def test_g2b_reference_rows_and_iteration_counts_at_ten_thousand_models():
    """golden G2b: the REFERENCE's rows and its own iteration counts at M = 10 000 (2104 / 4794 passes for the third object)"""
    g = load_golden('g2b_modec_10k')
    Y, Ye = g['Y'], g['Ye']
    for tname, ltol in (('t4', 1e-4), ('t8', 1e-8)):
        bf, niter, info = hip_fit(Y, Ye, np.ones_like(Y), g['X'], g['Xe'], np.ones((3, 5)), ltol)
        assert info[2:] == (3, 1024)
        for oi in range(3):
            k = 'o%d_%s' % (oi, tname)
            assert niter[oi] == int(g[k + '_niter'])
            np.testing.assert_allclose(bf.fit_lnlike[oi], g[k + '_lnl'], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(bf.fit_scale[oi], g[k + '_scale'], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize('M', [3000, 10000, 20000])
def test_an_error_within_rounding_of_ltol_takes_the_ieee_rerun(M):
    """The reciprocal-based solve may only differ from the IEEE one in the last ulps; an object whose max |dlnl| lands that
    close to ltol is detected (interval guard) and re-run with IEEE divisions.  ltol is placed 1e-10 (relative) above / below
    the error the ORACLE sees at some pass t*: the reference then stops at t* / goes on, the guard must fire (fz_modec_info
    counts the re-run) and the iteration count must be the oracle's on both sides."""
    Y, Ye, Ym, X, Xe, Xm = problem(M, 6, 4242 + M, amps=(2.0,))
    i = cand = trace = None
    for i in range(6):
        # a pass whose error is the smallest so far (so the reference has not stopped before it) and comfortably above the 1e-7 the trace ends at
        base = fo.lnlike_scaled(X[i].copy(), Xe[i].copy(), Xm[i].copy(), Y, Ye, Ym, ltol=1e-7, return_niter=True, return_trace=True, max_iter=1500)
        trace = base[-1]
        errs = np.array([t[0] for t in trace])
        cand = [t for t in range(3, len(errs) - 2) if errs[t] < errs[:t].min() and errs[t] > 1e-6] if base[-2] < 1500 else []
        if len(cand) >= 3:
            break
    assert cand and len(cand) >= 3
    tstar = cand[len(cand) // 2]
    e, mag = trace[tstar]
    assert 1e-10 * e < 0.2 * 2.9e-14 * mag            # ltol sits inside the guard's interval (fz_modec.h: d = 2.9e-14 (|l| + |l_old|))
    for ltol, stops_at in ((e * (1 + 1e-10), tstar + 1), (e * (1 - 1e-10), None)):
        want = fo.lnlike_scaled(X[i].copy(), Xe[i].copy(), Xm[i].copy(), Y, Ye, Ym, ltol=ltol, return_scale=True, return_niter=True)
        if stops_at is not None:
            assert want[5] == stops_at
        else:
            assert want[5] > tstar + 1
        bf, niter, info = hip_fit(Y, Ye, Ym, X[i:i + 1], Xe[i:i + 1], Xm[i:i + 1], ltol)
        assert info[0] >= 1, info                      # the guard fired: the object was re-run by the IEEE instantiation
        assert niter[0] == want[5]
        np.testing.assert_allclose(bf.fit_lnlike[0], want[0], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(bf.fit_scale[0], want[3], rtol=1e-9, atol=1e-11)
    # and an ordinary ltol does not fire it
    _, _, info = hip_fit(Y, Ye, Ym, X[i:i + 1], Xe[i:i + 1], Xm[i:i + 1], 1e-4)
    assert info[0] == 0


@pytest.mark.parametrize('M', [10000, 20000])
def test_mode_c_fused_pdfs_with_and_without_the_in_kernel_final_pass(M, monkeypatch):
    """fit_predict(save_fits=False) in mode C: the iteration kernel writes the final ln-like plane itself (one plane instead of
    four), ``FZ_MODEC_FINAL=1`` keeps the separate k_modec_final pass; both against the oracle's PDFs"""
    from frankenz_amd import BruteForce, PDFDict
    Y, Ye, Ym, X, Xe, Xm = problem(M, 6, 77 + M)
    rs = np.random.RandomState(5)
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
    grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
    pd, od = PDFDict(grid, sg), fo.KernelDict(grid, sg)
    rp, rlm, rle = fo.bruteforce_fit_predict(X[:3].copy(), Xe[:3].copy(), Xm[:3].copy(), Y, Ye, Ym, z, ze, label_dict=od, **KW)
    outs = []
    for final in (None, '1'):
        with monkeypatch.context() as mp:
            if final:
                mp.setenv('FZ_MODEC_FINAL', final)
            p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=pd, lprob_kwargs=KW,
                                                            return_gof=True, save_fits=False, verbose=False)
        np.testing.assert_allclose(p[:3], rp, rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(lm[:3], rlm, rtol=1e-9)
        np.testing.assert_allclose(le[:3], rle, **EVID64)
        outs.append((p, lm, le))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('zero_model', [0, 7])
def test_mode_c_with_a_model_whose_ln_like_is_not_a_number(zero_model, monkeypatch):
    """a model of all-zero fluxes has shape 0 and scale 0 / 0: its ln-like is nan in every pass.  pdf.py:199 takes the builtin ``max``
    over the errors, which skips a nan unless it is element 0 -- then the loop ends at once.  Such a model set is not "tame": the
    library keeps it off the reciprocal-based solve (and with it off k_modec_rounds, whose stop rule assumes finite errors) and runs
    the IEEE kernel -- asserted here -- which must give the oracle's pass counts and rows for both placements."""
    Y, Ye, Ym, X, Xe, Xm = problem(3000, 8, 31 + zero_model)
    Y[zero_model] = 0.0
    with np.errstate(all='ignore'):
        want = oracle_rows(Y, Ye, Ym, X, Xe, Xm, 1e-4, range(8))
        bf, niter, info = hip_fit(Y, Ye, Ym, X, Xe, Xm, 1e-4)
    assert info[2] == 1 and info[0] == 0                       # one block per object, one iteration per record read, IEEE: no re-run
    for i, w in enumerate(want):
        assert niter[i] == w[5], (zero_model, i, niter[i], w[5])
        fin = np.isfinite(w[0])
        np.testing.assert_array_equal(np.isfinite(bf.fit_lnlike[i]), fin)
        np.testing.assert_allclose(bf.fit_lnlike[i][fin], w[0][fin], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize('M', [3000, 10000, 20000])
def test_mode_c_rounds_that_run_past_the_stop_are_redone(M, monkeypatch):
    """FZ_MODEC_RFIXED=7: every round of k_modec_rounds is seven iterations long instead of sized from the error's decay, so most
    objects reach the reference's stop in the MIDDLE of a round.  Their models have then moved past the state the reference returns:
    the kernel must notice (the first iteration without a model above ltol is not the round's last) and hand the object to the IEEE
    kernel -- same pass counts and rows as the oracle, and the hand-over count says it happened."""
    Y, Ye, Ym, X, Xe, Xm = problem(M, 10, 77 + M // 1000)
    want = oracle_rows(Y, Ye, Ym, X, Xe, Xm, 1e-4, range(10))
    monkeypatch.setenv('FZ_MODEC_RFIXED', '7')
    bf, niter, info = hip_fit(Y, Ye, Ym, X, Xe, Xm, 1e-4)
    assert info[2] == 3 and info[0] >= 5, info                  # ((T - 2) mod 7 == 0 for about one object in seven only)
    for i, w in enumerate(want):
        assert niter[i] == w[5], (M, i, niter[i], w[5])
        np.testing.assert_allclose(bf.fit_lnlike[i], w[0], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(bf.fit_scale[i], w[3], rtol=1e-9, atol=1e-11)
    monkeypatch.delenv('FZ_MODEC_RFIXED')
    _, niter2, info2 = hip_fit(Y, Ye, Ym, X, Xe, Xm, 1e-4)
    np.testing.assert_array_equal(niter2, niter)
    assert info2[0] == 0                                        # predicted rounds: nothing handed over
