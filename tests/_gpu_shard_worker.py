"""One rank of the multi-process GPU tests (tests/test_hip_sharded.py): a fresh process per rank,
gloo rendezvous on 127.0.0.1, every rank on GPU 0 of the box.  Runs sharded_fit_predict with the
REAL HIP BruteForce / NearestNeighbors on this rank's object block and saves what it got.
    python tests/_gpu_shard_worker.py <scenario> <rank> <world> <port> <outdir>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def scenario(name):
    """-> (fitter factory, data, err, mask, labels, label_errs, fit_predict kwargs)"""
    from frankenz_amd import BruteForce, NearestNeighbors, PDFDict
    d = PDFDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    if name.startswith('g12'):
        from conftest import load_golden
        g = load_golden('g12_catalogue_stack')
        obs, err = g['obs'], g['err']
        if name == 'g12_grid':
            mk = lambda: BruteForce(g['mphot'], np.zeros_like(g['mphot']), np.ones_like(g['mphot']))
            return mk, obs, err, np.ones_like(obs), g['mz'], np.full(len(g['mz']), 0.03), dict(
                label_dict=d, lprob_kwargs={'free_scale': True, 'ignore_model_err': True}, save_fits=False)
        mk = lambda: BruteForce(g['tr_obs'], g['tr_err'], np.ones_like(g['tr_obs']))
        return mk, obs, err, np.ones_like(obs), g['tr_z'], np.full(len(g['tr_z']), 0.05), dict(label_dict=d, save_fits=False)
    rs = np.random.RandomState(2026)
    sig = np.array([0.873, 0.348, 0.418, 0.873, 3.476])
    if name in ('bf_big', 'bf_ragged'):
        N, M = (40000, 3000) if name == 'bf_big' else (100003, 2000)       # ragged: no rank count divides it
        Y = rs.lognormal(1., 1., size=(M, 5)); Ye = np.tile(sig, (M, 1)); Ym = np.ones((M, 5))
        X = Y[rs.choice(M, N)] + sig * rs.randn(N, 5); Xe = np.tile(sig, (N, 1)); Xm = np.ones((N, 5))
        X[7, 2] = np.nan; Xm[11, 0] = 0
        z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
        return (lambda: BruteForce(Y, Ye, Ym)), X, Xe, Xm, z, ze, dict(label_dict=d, save_fits=False)
    if name == 'knn':
        N, M = 3001, 4000
        Y = rs.lognormal(1., 1., size=(M, 5)) * 3; Ye = 0.05 * Y; Ym = np.ones((M, 5))
        X = Y[rs.choice(M, N)] + sig * rs.randn(N, 5); Xe = np.tile(sig, (N, 1)); Xm = np.ones((N, 5))
        z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
        mk = lambda: NearestNeighbors(Y, Ye, Ym, K=7, feature_map='luptitude', fmap_kwargs=dict(skynoise=sig, zeropoints=10 ** (0.4 * 23.9)),
                                      rstate=np.random.RandomState(1), verbose=False)
        return mk, X, Xe, Xm, z, ze, dict(label_dict=d, k=6, save_fits=False)
    raise ValueError(name)


def main():
    name, rank, world, port, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    from frankenz_amd import sharded
    if world > 1:
        import torch.distributed as dist
        os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    mk, X, Xe, Xm, z, ze, kw = scenario(name)
    rst = (lambda: np.random.RandomState(2)) if name == 'knn' else (lambda: None)
    Xc, Xec, Xmc = X.copy(), Xe.copy(), Xm.copy()
    full, (lm, le) = sharded.sharded_fit_predict(mk(), Xc, Xec, Xmc, z, ze, gather='pdfs', rstate=rst(), **kw)
    stack, _ = sharded.sharded_fit_predict(mk(), X.copy(), Xe.copy(), Xm.copy(), z, ze, gather='stack', rstate=rst(), **kw)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), pdfs=full, lmap=lm, levid=le, stack=stack, x=Xc, xm=Xmc)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
