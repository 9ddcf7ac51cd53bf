import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import frankenz_oracle as fo
from frankenz_amd import BruteForce, PDFDict
from conftest import load_golden
grid, sg = np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500)
d, od = PDFDict(grid, sg), fo.KernelDict(grid, sg)
g = load_golden('g7_config1')
obs, err = g['obs'], g['err']
z = g['redshifts']; ze = np.full(len(obs), 0.03)
rp, rlm, rle = fo.bruteforce_fit_predict(obs.copy(), err.copy(), np.ones_like(obs), obs, err, np.ones_like(obs), z, ze, label_dict=od)
for tag, env in (('default', {}), ('nowspace', {'FZ_NO_WSPACE': '1'})):
    os.environ.update(env)
    p, (lm, le) = BruteForce(obs, err, np.ones_like(obs)).fit_predict(obs.copy(), err.copy(), np.ones_like(obs), z, ze, label_dict=d, return_gof=True, verbose=False, save_fits=False)
    for k in env: del os.environ[k]
    dl = np.abs(le - rle); dm = np.abs(lm - rlm); dp = np.abs(p - rp).max(axis=1)
    print(tag, 'worst le', np.argsort(dl)[-3:], np.sort(dl)[-3:], 'worst lm', np.argsort(dm)[-3:], np.sort(dm)[-3:], 'worst p', np.argsort(dp)[-3:], np.sort(dp)[-3:])
    for i in np.argsort(dp)[-3:]:
        print('   obj', i, 'le', le[i], rle[i], 'lm', lm[i], rlm[i], 'psum', p[i].sum(), 'nan', np.isnan(p[i]).sum(), 'snr', (obs[i] / err[i]).max())
