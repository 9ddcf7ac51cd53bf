"""
Multi-GPU driver: one process per GPU (``torch.distributed``; backend ``nccl`` is RCCL
over xGMI on ROCm, ``gloo`` on CPU for tests).

The path shards on the OBJECT axis only -- every object's likelihood row, weights and
PDF depend on that object and the (replicated, ~12 MB) model set alone
(bruteforce.py:602-631 has no cross-object state) -- so there is no collective inside
the compute.  What can be exchanged afterwards:

  * ``gather='pdfs'``  : all-gather of the (N/P, Nx) PDF shards -> the full (N, Nx)
                         array on every rank;
  * ``gather='stack'`` : all-reduce(sum) of the Nx-vector stacked PDF  sum_i pdf_i
                         (the population n(z) estimate), 5.6 KB.

For the k-NN variant the per-object Monte-Carlo draws are generated for ALL objects
before sharding, so the random stream -- and therefore the result -- does not depend on
the number of ranks.
"""
import numpy as np

__all__ = ["shard_bounds", "shard_slice", "allgather_rows", "allreduce_sum", "sharded_fit_predict"]


def shard_bounds(n, world):
    """Contiguous near-equal blocks: bounds[r]..bounds[r+1] is rank r's share."""
    base, rem = divmod(int(n), int(world))
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def shard_slice(n, world, rank):
    b = shard_bounds(n, world)
    return slice(int(b[rank]), int(b[rank + 1]))


def _dist():
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist


def _device_for(dist, group):
    import torch
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allgather_rows(local, n_total, group=None):
    """All-gather row blocks of unequal length (rank r holds shard_slice(n_total, P, r)).
    ``local`` is a NumPy array or a torch tensor; returns the same kind, full length."""
    import torch
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_total, world)
    sizes = np.diff(bounds)
    was_numpy = isinstance(local, np.ndarray)
    dev = _device_for(dist, group)
    t = torch.from_numpy(np.ascontiguousarray(local)) if was_numpy else local
    t = t.to(dev).contiguous()
    maxn = int(sizes.max())
    tail = tuple(t.shape[1:])
    if t.shape[0] == maxn and all(s == maxn for s in sizes):
        out = torch.empty((world * maxn,) + tail, dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(out, t, group=group)
    else:
        pad = torch.zeros((maxn,) + tail, dtype=t.dtype, device=dev)
        pad[: t.shape[0]] = t
        buf = torch.empty((world * maxn,) + tail, dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(buf, pad, group=group)
        out = torch.cat([buf[r * maxn: r * maxn + int(sizes[r])] for r in range(world)], dim=0)
    return out.cpu().numpy() if was_numpy else out


def allreduce_sum(vec, group=None):
    import torch
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return vec
    was_numpy = isinstance(vec, np.ndarray)
    dev = _device_for(dist, group)
    t = (torch.from_numpy(np.ascontiguousarray(vec)) if was_numpy else vec).to(dev).contiguous().clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy() if was_numpy else t


def sharded_fit_predict(fitter, data, data_err, data_mask, model_labels, model_label_errs, gather='pdfs',
                        group=None, rstate=None, **kwargs):
    """``fitter.fit_predict`` (BruteForce or NearestNeighbors) on this rank's block of objects.

    Returns ``(pdfs, (lmap, levid))``: the FULL arrays when ``gather='pdfs'``; the local
    block when ``gather`` is ``None``; ``(stack, (lmap_local, levid_local))`` with the
    all-reduced  sum_i pdf_i  when ``gather='stack'``.  ``fit_*`` attributes (if kept) cover
    the local block only.  The in-place clean of the reference reaches the caller's
    arrays for the local block."""
    dist = _dist()
    world = dist.get_world_size(group) if dist is not None else 1
    rank = dist.get_rank(group) if dist is not None else 0
    n = len(data)
    sl = shard_slice(n, world, rank)
    kwargs = dict(kwargs, return_gof=True)
    kwargs.setdefault('verbose', False)
    if hasattr(fitter, 'KDTrees'):
        # k-NN: draw the Monte-Carlo realisation of EVERY object, then hand this rank a
        # generator that replays only its rows (knn.py:830 consumes B normals per object)
        if rstate is None:
            rstate = np.random
        draws = rstate.normal(np.asarray(data), np.asarray(data_err))
        kwargs['rstate'] = _Replay(draws[sl])
    pdfs, (lmap, levid) = fitter.fit_predict(data[sl], data_err[sl], data_mask[sl], model_labels,
                                             model_label_errs, **kwargs)
    if gather == 'pdfs':
        return (allgather_rows(pdfs, n, group), (allgather_rows(lmap, n, group), allgather_rows(levid, n, group)))
    if gather == 'stack':
        return allreduce_sum(np.nansum(pdfs, axis=0), group), (lmap, levid)
    return pdfs, (lmap, levid)


class _Replay(object):
    """A stand-in RandomState whose ``normal`` returns pre-drawn values."""

    def __init__(self, draws):
        self.draws = draws

    def normal(self, loc, scale):
        if np.shape(loc) != self.draws.shape:
            raise ValueError("replayed draws do not match the request")
        return self.draws
