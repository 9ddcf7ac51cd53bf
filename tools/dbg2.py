import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import frankenz_oracle as fo
from frankenz_amd import BruteForce, PDFDict
SDSS5 = np.array([0.873, 0.348, 0.418, 0.873, 3.476])
grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
d, od = PDFDict(grid, sg), fo.KernelDict(grid, sg)
kw = {'free_scale': True, 'ignore_model_err': True}
rs = np.random.RandomState(909)
M, N, B = 2100, 260, 5
Y = rs.lognormal(1., 1., size=(M, B)) * 3; Ye = np.tile(0.5 * SDSS5, (M, 1)); Ym = np.ones((M, B))
X = Y[rs.choice(M, N)] + SDSS5 * rs.randn(N, B); Xe = np.tile(SDSS5, (N, 1)); Xm = np.ones((N, B))
X[:8] = Y[:8]
Y[-1] = 1e5 * SDSS5; X[8] = Y[-1] + SDSS5 * rs.randn(B)
X[9] = Y[M // 2 + 7] * (1 + 1e-9)
z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
run = lambda: BruteForce(Y, Ye, Ym).fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=d, lprob_kwargs=kw,
                                                return_gof=True, save_fits=False, verbose=False)
rp, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=od, **kw)
for tag, env in (('default', {}), ('nowspace', {'FZ_NO_WSPACE': '1'})):
    os.environ.update(env)
    p, (lm, le) = run()
    for k in env: del os.environ[k]
    dl = np.abs(le - rle); dm = np.abs(lm - rlm); dp = np.abs(p - rp).max(axis=1)
    print(tag, 'worst le', np.argsort(dl)[-3:], np.sort(dl)[-3:], 'worst lm', np.argsort(dm)[-3:], np.sort(dm)[-3:], 'worst p', np.argsort(dp)[-3:], np.sort(dp)[-3:])
    for i in np.argsort(dl)[-2:]: print('   obj', i, 'le', le[i], rle[i], 'lm', lm[i], rlm[i])
# g7 train mode
from conftest import load_golden
g = load_golden('g7_config1')
print([k for k in g.keys()][:40])
