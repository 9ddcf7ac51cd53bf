"""N>1 path on CPU: two gloo ranks shard the object axis, compute their block with the
ORACLE standing in for the GPU kernels, and exchange the results with the same
collective code the GPU path uses (all-gather of PDF rows / all-reduce of the stack)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


class OracleBF(object):
    """BruteForce-shaped adapter around the oracle (CPU stand-in for the HIP path)."""

    def __init__(self, Y, Ye, Ym, kd):
        self.Y, self.Ye, self.Ym, self.kd = Y, Ye, Ym, kd

    def fit_predict(self, x, xe, xm, z, ze, return_gof=True, verbose=False, out=None, **kw):
        import frankenz_oracle as fo
        if out is not None:                       # the device-resident form of the product (tensors in, rows written in place)
            x, xe, xm = x.numpy(), xe.numpy(), xm.numpy()
        p, lm, le = fo.bruteforce_fit_predict(x, xe, xm, self.Y, self.Ye, self.Ym, z, ze, label_dict=self.kd)
        if out is not None:
            import torch
            out[0].copy_(torch.from_numpy(p)); out[1].copy_(torch.from_numpy(lm)); out[2].copy_(torch.from_numpy(le))
            return out[0], (out[1], out[2])
        return p, (lm, le)


class OracleBFTensors(OracleBF):
    """... that also takes the overlapped path of sharded_fit_predict (block-cyclic rounds, in-place all-gather)"""
    accepts_tensors = True

    def prepare_fit_predict(self, z, ze, label_dict=None, **kw):
        bf = self

        class _P(object):
            @staticmethod
            def run(x, xe, xm, out=None):
                return bf.fit_predict(x, xe, xm, z, ze, out=out)
        return _P()

    def _engine(self):
        class _E(object):
            @staticmethod
            def clean(x, xe, xm):                 # pdf.py:309-311 on the tensors' memory
                import frankenz_oracle as fo
                fo.clean_object(x.numpy(), xe.numpy(), xm.numpy())      # element-wise: works on the whole (N, B) arrays
        return _E()


class OracleKNN(object):
    """NearestNeighbors-shaped adapter around the oracle: the k-NN variant on the overlapped path of sharded_fit_predict
    (``KDTrees`` marks it; the Monte-Carlo query features are drawn for every object before the rounds)"""
    accepts_tensors = True
    NDIM = 5

    def __init__(self, Y, Ye, Ym, kd, K=3, k=4):
        import frankenz_oracle as fo
        self.Y, self.Ye, self.Ym, self.kd, self.K, self.k = Y, Ye, Ym, kd, K, k
        self.feats = fo.knn_train(Y, Ye, K, 'luptitude', np.random.RandomState(1))
        self.KDTrees = [None] * K

    def _query_features(self, x, xe, rstate):
        import frankenz_oracle as fo
        return fo.knn_query_features(x, xe, 'luptitude', rstate)

    def reference(self, x, xe, xm, q, z, ze):
        import frankenz_oracle as fo
        nb = fo.knn_neighbors_exact(self.feats, q, self.k)
        return fo.knn_fit_predict(x, xe, xm, self.Y, self.Ye, self.Ym, nb, z, ze, label_dict=self.kd)[:3]

    def prepare_fit_predict(self, z, ze, label_dict=None, k=None, **kw):
        nn = self
        assert k == self.k

        class _P(object):
            @staticmethod
            def run(x, xe, xm, out=None, query_features=None):
                import torch
                p, lm, le = nn.reference(x.numpy(), xe.numpy(), xm.numpy(), query_features.numpy(), z, ze)
                out[0].copy_(torch.from_numpy(p)); out[1].copy_(torch.from_numpy(lm)); out[2].copy_(torch.from_numpy(le))
                return out
        return _P()

    _engine = OracleBFTensors._engine


def _knn_worker(rank, world, port, n, q, chunks):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import torch.distributed as dist
    import frankenz_oracle as fo
    from frankenz_amd import sharded
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rs = np.random.RandomState(4)
    M, B = 90, 5
    Y = rs.lognormal(1, 1, (M, B)) * 5; Ye = 0.05 * Y; Ym = np.ones((M, B))
    X = Y[rs.choice(M, n)] + 0.3 * rs.randn(n, B); Xe = np.full((n, B), 0.3); Xm = np.ones((n, B))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
    kd = fo.KernelDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    nn = OracleKNN(Y, Ye, Ym, kd)
    full, (lm, le) = sharded.sharded_fit_predict(nn, X.copy(), Xe.copy(), Xm.copy(), z, ze, gather='pdfs', label_dict=kd, chunks=chunks,
                                                 save_fits=False, rstate=np.random.RandomState(2), k=nn.k)
    assert sharded.last_stats['world'] == world
    # the single-process answer: one (N, B) draw from the same seed, every object
    qq = nn._query_features(X, Xe, np.random.RandomState(2))
    ref, rlm, rle = nn.reference(X, Xe, Xm, qq, z, ze)
    ok = np.array_equal(full, ref) and np.array_equal(lm, rlm) and np.array_equal(le, rle)
    q.put((rank, bool(ok), full.shape))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n,chunks', [(2, 13, 3), (3, 20, 2)])
def test_gloo_knn_variant_on_the_overlapped_path(world, n, chunks):
    """configs[3]'s sharded form: block-cyclic rounds + in-place gather for NearestNeighbors, Monte-Carlo draws made for all
    objects first (the result does not depend on the rank count)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_knn_worker, args=(r, world, port, n, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n, 701) for _, _, shape in res)


def _worker(rank, world, port, n, q, tensors=False, chunks=4):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import torch.distributed as dist
    import frankenz_oracle as fo
    from frankenz_amd import sharded
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rs = np.random.RandomState(3)
    M, B = 120, 5
    Y = rs.lognormal(1, 1, (M, B)); Ye = 0.05 * Y; Ym = np.ones((M, B))
    X = Y[rs.choice(M, n)] + rs.randn(n, B); Xe = np.ones((n, B)); Xm = np.ones((n, B))
    X[1, 2] = np.nan                                  # exercises the in-place clean on rank 0's block
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
    kd = fo.KernelDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    bf = (OracleBFTensors if tensors else OracleBF)(Y, Ye, Ym, kd)
    Xc = X.copy()
    full, (lm, le) = sharded.sharded_fit_predict(bf, Xc, Xe.copy(), Xm.copy(), z, ze, gather='pdfs', label_dict=kd, chunks=chunks,
                                                 save_fits=False)
    if tensors:
        assert sharded.last_stats['world'] == world and np.isfinite(Xc).all()       # overlapped path taken; every rank's copy is cleaned
    stack, _ = sharded.sharded_fit_predict(bf, X.copy(), Xe.copy(), Xm.copy(), z, ze, gather='stack')
    local, _ = sharded.sharded_fit_predict(bf, X.copy(), Xe.copy(), Xm.copy(), z, ze, gather=None)
    ref, rlm, rle = fo.bruteforce_fit_predict(X.copy(), Xe.copy(), Xm.copy(), Y, Ye, Ym, z, ze, label_dict=kd)
    sl = sharded.shard_slice(n, world, rank)
    ok = (np.array_equal(full, ref) and np.array_equal(lm, rlm) and np.array_equal(le, rle)
          and np.allclose(stack, ref.sum(axis=0), rtol=1e-13, atol=0) and np.array_equal(local, ref[sl]))
    q.put((rank, bool(ok), full.shape))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n,tensors,chunks', [(2, 10, False, 4), (2, 11, False, 4),      # even and ragged split
                                                    (8, 11, False, 4),                          # more ranks than some blocks hold rows: the pad path of allgather_rows
                                                    (2, 11, True, 4), (2, 37, True, 3), (8, 37, True, 2), (3, 2, True, 4)])   # overlapped, block-cyclic rounds
def test_gloo_shard_and_gather(world, n, tensors, chunks):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, tensors, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n, 701) for _, _, shape in res)


def test_shard_bounds_cover_and_order():
    from frankenz_amd.sharded import shard_bounds, shard_slice
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0) and np.diff(b).max() - np.diff(b).min() <= 1
            assert sum(shard_slice(n, w, r).stop - shard_slice(n, w, r).start for r in range(w)) == n


def test_knn_draws_are_rank_independent():
    """the replayed per-rank draws concatenate to the single-process stream"""
    from frankenz_amd.sharded import _Replay, shard_slice
    rs = np.random.RandomState(9)
    X = rs.randn(13, 5); Xe = np.abs(rs.randn(13, 5)) + .1
    one = np.random.RandomState(2).normal(X, Xe)
    draws = np.random.RandomState(2).normal(X, Xe)
    parts = [_Replay(draws[shard_slice(13, 3, r)]).normal(X[shard_slice(13, 3, r)], Xe[shard_slice(13, 3, r)])
             for r in range(3)]
    assert np.array_equal(np.concatenate(parts), one)
