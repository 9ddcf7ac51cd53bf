"""debug aid: fused fit_predict vs the oracle on a small problem; prints the worst rows."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import frankenz_oracle as fo
from frankenz_amd import BruteForce, PDFDict
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
kw = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
noise = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
rs = np.random.RandomState(5)
B = 5
sig = np.array([0.873, 0.348, 0.418, 0.873, 3.476]) * noise
Y = rs.lognormal(1., 1., size=(M, B)); Ye = np.tile(sig, (M, 1)); Ym = np.ones((M, B))
X = Y[rs.choice(M, N)] + sig * rs.randn(N, B); Xe = np.tile(sig, (N, 1)); Xm = np.ones((N, B))
z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
grid, sgrid = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
pd, od = PDFDict(grid, sgrid), fo.KernelDict(grid, sgrid)
p, (lm, le) = BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=pd, lprob_kwargs=kw,
                                                return_gof=True, save_fits=False, verbose=False)
S = min(N, 200)
rp, rlm, rle = fo.bruteforce_fit_predict(X[:S].copy(), Xe[:S].copy(), Xm[:S].copy(), Y, Ye, Ym, z, ze, label_dict=od, **kw)
print('bad le>=lm:', np.sum(~(le >= lm - 1e-12)), 'nonfinite le', np.sum(~np.isfinite(le)), 'nonfinite p', np.sum(~np.isfinite(p).all(axis=1)))
dl = np.abs(le[:S] - rle); dm = np.abs(lm[:S] - rlm)
print('max |dle| %.3e  max |dlm| %.3e  max rel dp %.3e' % (dl.max(), dm.max(), (np.abs(p[:S] - rp) / (np.abs(rp) + 1e-14)).max()))
k = np.argsort(dl)[-5:]
for i in k: print(i, 'le', le[i], rle[i], 'lm', lm[i], rlm[i])
bad = np.where(~(le >= lm - 1e-12) | ~np.isfinite(le))[0][:10]
for i in bad: print('BAD', i, le[i], lm[i])
