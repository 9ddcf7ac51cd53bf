"""quick parity of the fused path against the oracle (dev aid): small problem, modes A / A-varying / Ai / B."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import frankenz_oracle as fo
from frankenz_amd import BruteForce, PDFDict
rs = np.random.RandomState(5)
M, N, B = int(os.environ.get("PQ_M", 5000)), int(os.environ.get("PQ_N", 300)), 5
sig = np.array([0.873, 0.348, 0.418, 0.873, 3.476])
Y = rs.lognormal(1., 1., size=(M, B)); Ym = np.ones((M, B))
X = Y[rs.choice(M, N)] + sig * rs.randn(N, B); Xe = np.tile(sig, (N, 1)); Xm = np.ones((N, B))
X[3] = Y[7]                                  # a self match
X[5] = 50 * Y[9] + sig * rs.randn(B)    # an object that matches nothing well (free scale: a good match at scale 50)
z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
grid, sgrid = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
pd, od = PDFDict(grid, sgrid), fo.KernelDict(grid, sgrid)
worst = 0
for name, Ye, kw in (("A const", np.tile(sig, (M, 1)), {}), ("A varying", np.tile(sig, (M, 1)) * rs.uniform(.5, 1.5, (M, B)), {}),
                     ("Ai", np.tile(sig, (M, 1)), {'ignore_model_err': True}), ("B", np.tile(sig, (M, 1)), {'free_scale': True, 'ignore_model_err': True})):
    if os.environ.get("PQ_KDE") == "grid":               # the direct gauss_kde (no dictionary), a few wide kernels among the labels
        zz = ze.copy(); zz[::97] = 0.4
        p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, zz, label_grid=grid, lprob_kwargs=kw,
                                                        return_gof=True, save_fits=False, verbose=False)
        rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, zz, label_grid=grid, **kw)
    else:
        p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=pd, lprob_kwargs=kw,
                                                        return_gof=True, save_fits=False, verbose=False)
        rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, **kw)
    dl, de = np.nanmax(np.abs(lm - rlm) / np.maximum(1, np.abs(rlm))), np.nanmax(np.abs(le - rle) / np.maximum(1, np.abs(rle)))
    dp = np.nanmax(np.abs(p - rp))
    from frankenz_amd.engine import get_engine
    name = name + ' [' + get_engine().last_form() + ']'
    nanmis = int((np.isnan(p) != np.isnan(rp)).sum())
    print("%-28s lmap %.2e  levid %.2e  pdf abs %.2e  nan-mismatch %d" % (name, dl, de, dp, nanmis), flush=True)
    worst = max(worst, dl, dp * 1e3, nanmis)
    if not (dl < 1e-11 and de < 1e-7 and dp < 1e-11 and nanmis == 0):
        bad = np.argsort(-np.nanmax(np.abs(p - rp), axis=1))[:5]
        print("   worst objects", bad, np.nanmax(np.abs(p - rp), axis=1)[bad], "levid diff", (le - rle)[bad], "lmap diff", (lm - rlm)[bad])
print("PARITY", "OK" if worst < 1e-8 else "FAIL")
