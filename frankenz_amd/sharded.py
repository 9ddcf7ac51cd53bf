"""
Multi-GPU driver: one process per GPU (``torch.distributed``; backend ``nccl`` is RCCL
over xGMI on ROCm, ``gloo`` on CPU for tests).

The path shards on the OBJECT axis only -- every object's likelihood row, weights and
PDF depend on that object and the (replicated, ~12 MB) model set alone
(bruteforce.py:602-631 has no cross-object state) -- so there is no collective inside
the compute.  What can be exchanged afterwards:

  * ``gather='pdfs'``  : all-gather of the (N/P, Nx) PDF shards -> the full (N, Nx)
                         array on every rank.  With ``save_fits=False`` (BruteForce and NearestNeighbors) the
                         gather is OVERLAPPED with the compute (``_overlapped``): a rank's objects are dealt out block-cyclically
                         in ``chunks`` rounds, round c's PDFs are written by the kernel straight into
                         their rows of the full device array and leave on RCCL's stream (in-place
                         ``all_gather_into_tensor``, ``async_op``) while round c + 1 is computed; one
                         fence at the end.  Nothing crosses PCIe except the final result when the
                         caller handed in NumPy arrays;
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


def _gpu_ok():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


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


last_stats = {}        # timing of the last overlapped call on this rank (ms_compute, ms_total, ms_gather_exposed, bytes, ...)


def _overlapped(fitter, data, data_err, data_mask, model_labels, model_label_errs, group, chunks, kwargs, rstate=None):
    """``fit_predict`` (BruteForce, or NearestNeighbors) over all ranks with the PDF all-gather hidden behind the compute.

    Object i belongs to round c = i // (P cs) and rank r = (i // cs) % P (cs = ceil(N / (P chunks))): the rows a
    round produces are one contiguous (P cs, Nx) slab of the result, rank r's part at offset r cs of it -- which is
    exactly the in-place form of an all-gather, so the kernel's output buffer IS the collective's buffer.
    (bruteforce.py:602-631, knn.py:826-874: no cross-object state, any assignment of objects to ranks gives the same rows;
    the k-NN variant's per-object Monte-Carlo draws, knn.py:830, are made for ALL objects before the rounds start.)

    Stream contract: the library is told that its inputs come from torch's current stream (``fz_set_producer_stream``), so
    round c + 1's kernels wait for that stream alone -- NOT for the device, which would drain round c's gather in flight on
    RCCL's stream and serialise the two (the default wait of a naive caller does exactly that)."""
    import time
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nccl = dist.get_backend(group) == "nccl"
    on_gpu = torch.cuda.is_available()
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")   # (cpu: the gloo tests' stand-in fitter)
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    was_numpy = isinstance(data, np.ndarray)
    is_knn = hasattr(fitter, 'KDTrees')
    N = int(data.shape[0])
    label_dict, label_grid = kwargs.get("label_dict"), kwargs.get("label_grid")
    if label_dict is None and label_grid is None:
        raise ValueError("`label_dict` or `label_grid` must be specified.")
    G = int(label_dict.Ngrid) if label_dict is not None else len(label_grid)

    def dv(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)) if isinstance(a, np.ndarray) else a
        return t.to(dev).contiguous()
    known = {"label_dict", "label_grid", "kde_kwargs", "lprob_kwargs", "lprob_func", "lprob_args", "kde_args",
             "return_gof", "save_fits", "verbose", "track_scale", "prepared"}
    search = {}
    if is_knn:
        known |= {"k", "eps", "lp_norm", "distance_upper_bound", "query_features"}
        search = {kk: kwargs[kk] for kk in ("k", "eps", "lp_norm", "distance_upper_bound") if kk in kwargs}
    extra = set(kwargs) - known
    if extra or kwargs.get("kde_args") or kwargs.get("lprob_args"):
        raise NotImplementedError("sharded_fit_predict (overlapped path): unsupported arguments %s" % sorted(extra))
    dQ = None
    if is_knn:
        # knn.py:830-832 for EVERY object before the objects are dealt out: the random stream -- and so the result -- does not
        # depend on the number of ranks.  (N x B values on the host; a driver that calls step after step passes them in.)
        q = kwargs.get("query_features")
        if q is None:
            hx, hxe = (np.asarray(data), np.asarray(data_err)) if was_numpy else (data.cpu().numpy(), data_err.cpu().numpy())
            q = fitter._query_features(hx, hxe, rstate if rstate is not None else np.random)
        dQ = dv(q)
    dX, dXe, dXm = dv(data), dv(data_err), dv(data_mask)
    eng = fitter._engine()
    lib_stream = on_gpu and hasattr(eng, "set_producer_stream")
    prev_contract = eng.producer_stream() if hasattr(eng, "producer_stream") else (None, 0)
    if lib_stream:
        eng.set_producer_stream(torch.cuda.current_stream().cuda_stream, 1)
    works = []
    try:
        # the reference cleans every object in place (pdf.py:310-311); every rank holds the full object arrays, so every
        # rank's copy ends up cleaned, as after the single-process call (N x B values: negligible)
        if not lib_stream:
            sync()                                                  # (stand-in fitters: the tensors are read on another stream)
        if N:
            eng.clean(dX, dXe, dXm)
        C = max(1, int(chunks))
        cs = max(1, -(-N // (world * C)))
        C = -(-N // (world * cs))                                   # rounds that hold at least one object
        rows = C * world * cs
        pdfs = torch.empty((rows, G), dtype=torch.float64, device=dev)
        lm = torch.empty(rows, dtype=torch.float64, device=dev)
        le = torch.empty(rows, dtype=torch.float64, device=dev)
        # models, dictionary, labels (feature sets): on the device once, not once per round -- and not at all when the caller
        # hands in the handle of an earlier ``prepare_fit_predict`` (a driver that calls this function step after step)
        prep = kwargs.get("prepared")
        if prep is not None and is_knn and search:
            want = (search.get("k", prep.search[0]), search.get("lp_norm", prep.search[1]), search.get("distance_upper_bound", prep.search[2]))
            if tuple(want) != tuple(prep.search):
                raise ValueError("search arguments %r disagree with the prepared handle's %r" % (want, prep.search))
        if prep is None:
            prep = fitter.prepare_fit_predict(model_labels, model_label_errs, label_dict=label_dict, label_grid=label_grid,
                                              kde_kwargs=kwargs.get("kde_kwargs"), lprob_kwargs=kwargs.get("lprob_kwargs"), **search)
        t_comp = 0.0
        sync()
        t_start = time.perf_counter()
        for c in range(C):
            base = c * world * cs
            lo = base + rank * cs
            hi = min(lo + cs, N)
            if hi > lo:
                t0 = time.perf_counter()
                if is_knn:
                    prep.run(dX[lo:hi], dXe[lo:hi], dXm[lo:hi], out=(pdfs[lo:hi], lm[lo:hi], le[lo:hi]), query_features=dQ[lo:hi])
                else:
                    prep.run(dX[lo:hi], dXe[lo:hi], dXm[lo:hi], out=(pdfs[lo:hi], lm[lo:hi], le[lo:hi]))
                t_comp += time.perf_counter() - t0              # the call returns when the rows are in HBM; it waits for nothing else
            for buf in (pdfs, lm, le):
                out_v, in_v = buf[base:base + world * cs], buf[lo:lo + cs]
                if nccl:
                    try:                                             # in place: rank r's rows already sit at offset r cs of the slab
                        works.append(dist.all_gather_into_tensor(out_v, in_v, group=group, async_op=True))
                    except (RuntimeError, ValueError):               # a build that refuses overlapping views: one copy of the rank's rows
                        works.append(dist.all_gather_into_tensor(out_v, in_v.clone(), group=group, async_op=True))
                else:                                                # gloo (tests, several ranks on one GPU): through host memory
                    tmp = torch.empty(out_v.shape, dtype=out_v.dtype)
                    dist.all_gather_into_tensor(tmp, in_v.cpu(), group=group)
                    out_v.copy_(tmp)
        t_fence0 = time.perf_counter()
        for w in works:
            w.wait()
        sync()
        t_end = time.perf_counter()
    finally:
        # collectives already queued are waited for whatever ended the loop (an exception in a later round must not leave them
        # writing into tensors that are about to go away), and the caller's own stream contract is put back
        for w in works:
            try:
                w.wait()
            except Exception:       # noqa: BLE001
                pass
        if lib_stream:
            eng.set_producer_stream(*prev_contract)
    t_total = t_end - t_start
    nbytes = rows * G * 8
    last_stats.clear()
    # ms_compute: host time inside the library calls (kernels + launches; with the stream contract it contains no wait for the
    # collective); ms_fence: what the final fence waited for = the part of the gathers that the compute did not cover
    last_stats.update(ms_compute=t_comp * 1e3, ms_total=t_total * 1e3, ms_gather_exposed=(t_total - t_comp) * 1e3,
                      ms_fence=(t_end - t_fence0) * 1e3, chunks=C, rows_per_chunk=cs, bytes_gathered=nbytes, world=world,
                      busbw_GBs_if_serial=None)
    if was_numpy:
        for src, dst in ((dX, data), (dXe, data_err), (dXm, data_mask)):
            dst[...] = src.cpu().numpy()
        return pdfs[:N].cpu().numpy(), (lm[:N].cpu().numpy(), le[:N].cpu().numpy())
    return pdfs[:N], (lm[:N], le[:N])


def sharded_fit_predict(fitter, data, data_err, data_mask, model_labels, model_label_errs, gather='pdfs',
                        group=None, rstate=None, chunks=4, rounds_on_one_rank=False, **kwargs):
    """``fitter.fit_predict`` (BruteForce or NearestNeighbors) on this rank's block of objects.
    (``prepared=``: a handle from ``fitter.prepare_fit_predict(...)`` with the same labels and options -- the overlapped
    path then skips the uploads / content checks of models, dictionary and labels.  ``rounds_on_one_rank=True``: take the
    overlapped path -- rounds, stream contract, in-place collective -- in a one-rank group too (a test of the whole call form
    through RCCL on a one-GPU box; with one rank the "gather" moves nothing).)

    Returns ``(pdfs, (lmap, levid))``: the FULL arrays when ``gather='pdfs'``; the local
    block when ``gather`` is ``None``; ``(stack, (lmap_local, levid_local))`` with the
    all-reduced  sum_i pdf_i  when ``gather='stack'``.  ``fit_*`` attributes (if kept) cover
    the local block only.  The in-place clean of the reference reaches the caller's
    arrays for the local block."""
    dist = _dist()
    world = dist.get_world_size(group) if dist is not None else 1
    rank = dist.get_rank(group) if dist is not None else 0
    n = len(data)
    # the reference's default is save_fits=True (bruteforce.py:374-378, knn.py:560-563): only an EXPLICIT save_fits=False takes the
    # device-resident path, which has no fit_* arrays to fill
    if (gather == 'pdfs' and dist is not None and (world > 1 or rounds_on_one_rank) and chunks
            and kwargs.get('lprob_func') is None and kwargs.get('save_fits', True) is False
            and (_gpu_ok() or getattr(fitter, 'accepts_tensors', False))):
        # built-in likelihood: device-resident, the gather overlapped with the compute (BruteForce and NearestNeighbors alike)
        return _overlapped(fitter, data, data_err, data_mask, model_labels, model_label_errs, group, chunks, kwargs, rstate)
    if hasattr(data, "data_ptr"):
        raise NotImplementedError("device tensors are taken by the overlapped path only "
                                  "(gather='pdfs', save_fits=False, built-in likelihood)")
    sl = shard_slice(n, world, rank)
    kwargs = dict(kwargs, return_gof=True)
    kwargs.setdefault('verbose', False)
    if hasattr(fitter, 'KDTrees'):
        if kwargs.get('query_features') is not None:
            kwargs['query_features'] = kwargs['query_features'][sl]         # drawn by the caller for every object
        else:
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
