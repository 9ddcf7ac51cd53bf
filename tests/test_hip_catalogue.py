"""Real-catalogue inputs on the one-pass kernel (fz_hist.h, SEG): masked MODELS and objects with unobserved bands against per-model
errors -- pdf.py:76-87 / 181-189 with models_mask and per-model models_err, where N_dim is a property of the (object, model) PAIR.
The kernel walks a copy of the model records sorted by mask pattern; every case is held against the oracle AND against the masked
kernels of round 2 (FZ_HIST_SEG=0), and the route taken is asserted so that a silent fall-back fails the test."""
import numpy as np
import pytest

import frankenz_oracle as fo
from conftest import EVID

pytestmark = pytest.mark.gpu
EVID64 = dict(rtol=1e-12, atol=1e-12)
SDSS5 = np.array([0.873, 0.348, 0.418, 0.873, 3.476])
MODES = {'A': {}, 'Ai': {'ignore_model_err': True}, 'B': {'free_scale': True, 'ignore_model_err': True}}


def close(a, b, rtol=1e-9, atol=1e-11):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def dicts():
    from frankenz_amd import PDFDict
    grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
    return PDFDict(grid, sg), fo.KernelDict(grid, sg)


def catalogue(rs, M, N, B, model_err, model_mask, obj_mask, sig=None):
    sig = np.resize(SDSS5, B) if sig is None else sig
    Y = rs.lognormal(1., 1., size=(M, 1)) * rs.lognormal(0., .5, size=(M, B)) * 3
    Ye = np.tile(0.3 * sig, (M, 1)) if model_err == 'const' else 0.3 * sig * rs.uniform(0.5, 1.5, size=(M, B))
    Ym = (rs.uniform(size=(M, B)) >= model_mask).astype(float)
    X = Y[rs.choice(M, N)] * rs.lognormal(0, .1, N)[:, None] + sig * rs.randn(N, B)
    Xe = np.tile(sig, (N, 1)) * rs.uniform(0.8, 1.2, size=(N, B))
    Xm = (rs.uniform(size=(N, B)) >= obj_mask).astype(float)
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
    return Y, Ye, Ym, X, Xe, Xm, z, ze


def undefined_rows(mode, Xm, Ym):
    """objects for which some pair has too few common bands for a defined ln-like under the dimensionality prior (pdf.py:88-93,
    226-229: N_dim = 0 gives nan - inf; the free scale with one band gammaln(0)): the reference's own row is nan / -inf there"""
    nd = Xm.astype(int) @ Ym.astype(int).T
    return (nd <= (1 if mode == 'B' else 0)).any(axis=1)


@pytest.mark.parametrize('mode', ['A', 'Ai', 'B'])
@pytest.mark.parametrize('model_err', ['const', 'varying'])
@pytest.mark.parametrize('obj_mask', [0.0, 0.1])
def test_masked_models_on_the_one_pass_kernel(mode, model_err, obj_mask, monkeypatch):
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    d, od = dicts()
    kw = MODES[mode]
    rs = np.random.RandomState(811 + len(mode) + (model_err == 'const') + int(10 * obj_mask))
    M, N, B = 3100, 460, 5
    Y, Ye, Ym, X, Xe, Xm, z, ze = catalogue(rs, M, N, B, model_err, 0.06, obj_mask)
    Ym[Ym.sum(axis=1) < 4] = 1.0                                  # (one model with a single band would leave EVERY object's row undefined with the free scale)
    X[Xm == 0] = 1e6                                              # garbage in the unobserved bands must not matter
    Y[Ym == 0] = -3.0                                             # ... nor a placeholder in a masked model band
    bf = BruteForce(Y, Ye, Ym)
    run = lambda: bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, lprob_kwargs=kw, return_gof=True, save_fits=False, verbose=False)
    with np.errstate(all='ignore'):
        p0, (lm0, le0) = run()
        assert get_engine().last_form() == ('k_hist<exact> (segmented models)' if mode == 'B' else 'k_hist<screen> (segmented models)')
        monkeypatch.setenv('FZ_HIST_SEG', '0')
        p1, (lm1, le1) = run()
        assert 'segmented' not in get_engine().last_form()
        monkeypatch.delenv('FZ_HIST_SEG')
        rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, **kw)
    ok = ~undefined_rows(mode, Xm, Ym) & np.isfinite(rp).all(axis=1)
    assert ok.mean() > 0.8
    close(p0[ok], rp[ok], rtol=1e-7, atol=1e-13); close(lm0[ok], rlm[ok], rtol=1e-9); close(le0[ok], rle[ok], **EVID64)
    if mode != 'B':        # (free scale with ONE common band: zero degrees of freedom, nan or -inf by the rounding of a perfect fit's residual)
        assert np.isnan(p0[np.isnan(rp).all(axis=1)]).all()
    fin = np.isfinite(p1).all(axis=1) & np.isfinite(p0).all(axis=1)
    assert (fin | ~ok).all()
    close(p0[fin], p1[fin], rtol=1e-7, atol=1e-13); close(lm0[fin], lm1[fin], rtol=1e-9); close(le0[fin], le1[fin], **EVID)


@pytest.mark.parametrize('B', [4, 5, 6, 8])
def test_unobserved_object_bands_against_per_model_errors(B, monkeypatch):
    """unmasked models WITH their own errors, objects with missing bands (the masked half of a training-set fit, which the chunk
    split used to leave on round 2's kernel): one segment, the multiplier form of the difference."""
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    d, od = dicts()
    rs = np.random.RandomState(900 + B)
    M, N = 2500, 400
    Y, Ye, Ym, X, Xe, Xm, z, ze = catalogue(rs, M, N, B, 'varying', 0.0, 0.2, sig=rs.uniform(0.3, 2.0, B))
    for i, nobs in enumerate(range(B + 1)):                       # 0 ... B observed bands, explicitly
        Xm[i] = 0.0; Xm[i, :nobs] = 1.0
    X[Xm == 0] = -7e5
    bf = BruteForce(Y, Ye, Ym)
    run = lambda: bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, return_gof=True, save_fits=False, verbose=False)
    with np.errstate(all='ignore'):
        p0, (lm0, le0) = run()
        assert get_engine().last_form() == 'k_hist<screen> (segmented models)'
        monkeypatch.setenv('FZ_NOLIST', '1')                      # the form that weighs every pair directly
        p2, (lm2, le2) = run()
        assert get_engine().last_form() == 'k_hist<exact> (segmented models)'
        monkeypatch.delenv('FZ_NOLIST')
        rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od)
    ok = np.isfinite(rp).all(axis=1)
    assert ok.sum() >= N - 3
    for p, lm, le in ((p0, lm0, le0), (p2, lm2, le2)):
        close(p[ok], rp[ok], rtol=1e-7, atol=1e-13); close(lm[ok], rlm[ok], rtol=1e-9); close(le[ok], rle[ok], **EVID64)
        assert np.isnan(p[~ok]).all()


def test_self_match_through_a_masked_band():
    """a training object fitted against its own (masked) record: chi2 is exactly 0 over the common bands in the reference (weight 0
    under the dimensionality prior); the segmented kernel must not turn the masked band's garbage into a tiny chi2 that wins."""
    from frankenz_amd import BruteForce
    d, od = dicts()
    rs = np.random.RandomState(31)
    M, B = 3000, 5
    Y, Ye, Ym, _, _, _, z, ze = catalogue(rs, M, 8, B, 'varying', 0.1, 0.0)
    js = rs.choice(M, 24, replace=False)
    X = Y[js].copy(); Xe = np.tile(SDSS5 * 0.05, (len(js), 1)); Xm = np.ones_like(X)
    X[Ym[js] == 0] += 50.0                                        # differs from the model only where the model is masked
    for mode in ('A', 'Ai'):
        p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, lprob_kwargs=MODES[mode],
                                                        return_gof=True, save_fits=False, verbose=False)
        rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, **MODES[mode])
        ok = np.isfinite(rp).all(axis=1)
        assert ok.sum() >= len(js) - 2
        close(lm[ok], rlm[ok], rtol=1e-10); close(le[ok], rle[ok], rtol=1e-10); close(p[ok], rp[ok], rtol=1e-7, atol=1e-13)


def test_many_mask_patterns_and_short_segments():
    """8 bands at 25 % missing: ~200 patterns, most of them a handful of models (segments of one group, mostly pad slots), and a
    model count that is not a multiple of anything."""
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    d, od = dicts()
    rs = np.random.RandomState(77)
    M, N, B = 1777, 150, 8
    Y, Ye, Ym, X, Xe, Xm, z, ze = catalogue(rs, M, N, B, 'varying', 0.25, 0.1, sig=rs.uniform(0.3, 2.0, B))
    Ym[Ym.sum(axis=1) < 3] = 1.0                                  # every pair keeps a defined likelihood ... mostly
    for kw in ({}, {'free_scale': True, 'ignore_model_err': True}):
        with np.errstate(all='ignore'):
            p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, lprob_kwargs=kw,
                                                            return_gof=True, save_fits=False, verbose=False)
            form = get_engine().last_form()
            rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, **kw)
        assert 'segmented' in form or form == 'k_fused'           # (padding beyond 25 % of the set: the form declines)
        ok = np.isfinite(rp).all(axis=1) & ~undefined_rows('B' if kw else 'A', Xm, Ym)
        close(p[ok], rp[ok], rtol=1e-7, atol=1e-13); close(lm[ok], rlm[ok], rtol=1e-9); close(le[ok], rle[ok], **EVID64)


def test_full_chip_launch_with_masked_models():
    """enough objects for every CU (16 waves x 256 blocks) and more than one round per wave; the oracle checks a sample."""
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    d, od = dicts()
    rs = np.random.RandomState(5)
    M, N, B = 5000, 9000, 5
    Y, Ye, Ym, X, Xe, Xm, z, ze = catalogue(rs, M, N, B, 'varying', 0.02, 0.02)
    with np.errstate(all='ignore'):
        p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, return_gof=True, save_fits=False,
                                                        verbose=False)
        assert get_engine().last_form() == 'k_hist<screen> (segmented models)'
        pick = np.concatenate([[0, 1, N - 1], rs.choice(N, 60, replace=False)])
        rp, rlm, rle = fo.bruteforce_fit_predict(X[pick].copy(), Xe[pick].copy(), Xm[pick].copy(), Y, Ye, Ym, z, ze, label_dict=od)
    ok = np.isfinite(rp).all(axis=1)
    close(p[pick][ok], rp[ok], rtol=1e-7, atol=1e-13); close(lm[pick][ok], rlm[ok], rtol=1e-9); close(le[pick][ok], rle[ok], **EVID64)
    fin = np.isfinite(p).all(axis=1)
    assert fin.mean() > 0.99 and np.abs(p[fin].sum(axis=1) - 1).max() < 1e-9


@pytest.mark.parametrize('mode', ['A', 'Ai', 'B'])
@pytest.mark.parametrize('masks', ['none', 'objects', 'models+objects'])
def test_many_dictionary_widths_on_the_one_pass_kernel(mode, masks, monkeypatch):
    """per-model label errors (gauss_kde_dict with many kernel widths, pdf.py:599-620) on the segmented kernel: the model stream is
    ordered by width class, the histogram is convolved with its class's kernel and added to the PDF row at every change of class
    (fz_hist.h, class_flush).  With and without masks; labels piled at both grid edges (truncated kernel masses), a class of one
    model at either end of the width range; against the oracle, against the class-sorted stack of k_fused (FZ_HIST_SEG_MC=0), with
    the ambiguous lists forced to overflow (the hand-back to the sweep), un-normalised, and in the form that weighs every pair."""
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    d, od = dicts()
    kw = MODES[mode]
    rs = np.random.RandomState(1200 + len(mode) + len(masks))
    M, N, B = 12000 if masks == 'models+objects' else 3300, 420, 5
    Y, Ye, Ym, X, Xe, Xm, z, ze = catalogue(rs, M, N, B, 'varying', 0.0, 0.1 if masks != 'none' else 0.0)
    if masks == 'models+objects':                                  # four mask patterns x ~30 classes: 120 segments (every pattern of every class is
        Ym[rs.uniform(size=M) < 0.15, 1] = 0.0                     # padded to whole 64-model groups; the layout is declined when pads would dominate)
        Ym[rs.uniform(size=M) < 0.10, 3] = 0.0
    z = np.clip(rs.uniform(-0.3, 7.3, M), 0.0, 7.0)
    ze = rs.uniform(0.01, 0.12, M)                                 # ~28 classes, half-widths 5..60
    ze[7] = 0.125; ze[8] = 0.006
    X[Xm == 0] = 1e6; Y[Ym == 0] = -3.0
    bf = BruteForce(Y, Ye, Ym)
    run = lambda **kk: bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, lprob_kwargs=kw, return_gof=True, save_fits=False,
                                      verbose=False, **kk)
    want = 'k_hist<exact> (segmented models, many widths)' if mode == 'B' else 'k_hist<screen> (segmented models, many widths)'
    with np.errstate(all='ignore'):
        if masks == 'none':
            monkeypatch.setenv('FZ_HIST_SEG_MC', '1')              # (mask-free data keep k_fused's class-sorted stack by default: faster there)
        p0, (lm0, le0) = run()
        assert get_engine().last_form() == want
        monkeypatch.setenv('FZ_HIST_SEG_MC', '0')
        p1, (lm1, le1) = run()
        assert 'many widths' not in get_engine().last_form()
        monkeypatch.delenv('FZ_HIST_SEG_MC')
        if masks == 'none':
            monkeypatch.setenv('FZ_HIST_SEG_MC', '1')
        monkeypatch.setenv('FZ_HIST_AMBCAP', '2')                  # nearly every object overflows its list: the exact sweep redoes it
        p2, (lm2, le2) = run()
        monkeypatch.delenv('FZ_HIST_AMBCAP')
        monkeypatch.setenv('FZ_NOLIST', '1')
        p3, (lm3, le3) = run()
        assert get_engine().last_form() == 'k_hist<exact> (segmented models, many widths)'
        monkeypatch.delenv('FZ_NOLIST')
        rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, **kw)
    ok = ~undefined_rows(mode, Xm, Ym) & np.isfinite(rp).all(axis=1)
    assert ok.mean() > 0.8
    for p, lm, le in ((p0, lm0, le0), (p2, lm2, le2), (p3, lm3, le3)):
        close(p[ok], rp[ok], rtol=1e-7, atol=1e-13); close(lm[ok], rlm[ok], rtol=1e-9); close(le[ok], rle[ok], **EVID64)
        np.testing.assert_allclose(p[ok].sum(axis=1), 1.0, rtol=1e-12)
    fin = np.isfinite(p1).all(axis=1) & ok
    close(p0[fin], p1[fin], rtol=1e-7, atol=1e-13); close(le0[fin], le1[fin], **EVID)


@pytest.mark.parametrize('mode', ['A', 'Ai', 'B'])
def test_segmented_form_on_mask_free_data(mode, monkeypatch):
    """FZ_HIST_SEG_FORCE=1: the segmented kernel on data that does not need it (one segment, one pattern) -- the same PDFs, ln-max
    and ln-evidence as the plain one-pass kernel, to rounding"""
    from frankenz_amd import BruteForce
    from frankenz_amd.engine import get_engine
    d, od = dicts()
    rs = np.random.RandomState(77 + len(mode))
    Y, Ye, Ym, X, Xe, Xm, z, ze = catalogue(rs, 4000, 700, 5, 'varying', 0.0, 0.0)
    bf = BruteForce(Y, Ye, Ym)
    run = lambda: bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, lprob_kwargs=MODES[mode], return_gof=True, save_fits=False,
                                 verbose=False)
    p0, (lm0, le0) = run()
    assert 'segmented' not in get_engine().last_form()
    monkeypatch.setenv('FZ_HIST_SEG_FORCE', '1')
    p1, (lm1, le1) = run()
    assert 'segmented' in get_engine().last_form()
    close(p1, p0, rtol=1e-9, atol=1e-15); close(lm1, lm0, rtol=1e-13); close(le1, le0, **EVID64)
