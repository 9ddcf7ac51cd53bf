"""
Thin object wrapper over one ``fz_ctx`` (one per GPU).  Everything numeric happens
inside libfrankenz_hip.so; this file only marshals arrays.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import KdeOpts, LikeOpts, Prior, Timing, check, ptr

# the reference's logprob keywords + one extension: exact_evidence=True sums every weight of the fused path's
# ln-evidence in fp64 (default: the weights below wt_thresh of the best are summed in fp32, ~1e-9 on levid)
_LIKE_KEYS = ("free_scale", "ignore_model_err", "dim_prior", "ltol", "return_scale", "exact_evidence")


def like_opts(lprob_kwargs, max_iter=0):
    """lprob_kwargs of pdf.logprob (pdf.py:326-328) -> fz_like_opts."""
    kw = dict(lprob_kwargs or {})
    extra = set(kw) - set(_LIKE_KEYS)
    if extra:
        raise NotImplementedError(
            "lprob_kwargs %s are not understood by the HIP likelihood "
            "(supported: %s)" % (sorted(extra), ", ".join(_LIKE_KEYS)))
    return LikeOpts(int(bool(kw.get("free_scale", False))),
                    int(bool(kw.get("ignore_model_err", False))),
                    int(bool(kw.get("dim_prior", True))), int(max_iter),
                    float(kw.get("ltol", 1e-4)), int(bool(kw.get("exact_evidence", False))), 0)


def merge_kde_args(kde_args, kde_kwargs, use_dict):
    """Positional ``kde_args`` of predict / fit_predict, as the reference hands them on (bruteforce.py:361-369):
    ``gauss_kde_dict(label_dict, y_idx=.., y_std_idx=.., y_wt=wt, *kde_args, **kde_kwargs)`` -- up to two positionals land
    on ``y`` / ``y_std``, which the function ignores when the indices are given (pdf.py:570-573), a third collides with
    ``y_idx``; ``gauss_kde(labels, label_errs, grid, y_wt=wt, *kde_args, **kde_kwargs)`` -- the first positional is ``dx``,
    a second collides with ``y_wt``.  Returns the keyword dict to use; raises the TypeError Python raises for the rest."""
    kw = dict(kde_kwargs or {})
    args = tuple(kde_args or ())
    if use_dict:
        if len(args) > 2:
            raise TypeError("gauss_kde_dict() got multiple values for argument 'y_idx'")
        return kw
    if len(args) > 1:
        raise TypeError("gauss_kde() got multiple values for argument 'y_wt'")
    if len(args) == 1:
        if 'dx' in kw:
            raise TypeError("gauss_kde() got multiple values for argument 'dx'")
        kw['dx'] = args[0]
    return kw


def kde_opts(kde_kwargs, normalize=True):
    """kde_kwargs of gauss_kde / gauss_kde_dict -> fz_kde_opts.  ``wt_thresh=None``
    with ``cdf_thresh=None`` means no thresholding (pdf.py:495-496, 578-579)."""
    kw = dict(kde_kwargs or {})
    wt = kw.pop("wt_thresh", 1e-3)
    cdf = kw.pop("cdf_thresh", 2e-4)
    ex = int(bool(kw.pop("exact_evidence", False)))         # extension: the whole logsumexp in fp64
    kw.pop("sig_thresh", None)
    kw.pop("dx", None)
    if kw:
        raise NotImplementedError("kde_kwargs %s are not supported" % sorted(kw))
    if wt is None and cdf is None:
        return KdeOpts(-np.inf, 1, int(normalize), 0.0, ex, 0)
    if wt is None:
        return KdeOpts(0.0, 0, int(normalize), float(cdf), ex, 0)
    return KdeOpts(float(wt), 1, int(normalize), 0.0 if cdf is None else float(cdf), ex, 0)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


try:
    import xxhash as _xx

    def _hash_bytes(h, a):
        h.update(memoryview(a).cast('B'))

    def _new_hash():
        return _xx.xxh3_128()
except ImportError:                                   # pragma: no cover  (xxhash ships with the image)
    import hashlib as _hl

    def _hash_bytes(h, a):
        h.update(memoryview(a).cast('B'))

    def _new_hash():
        return _hl.blake2b(digest_size=16)


def _digest(*arrays):
    """(shapes, 128-bit content hash) of C-contiguous arrays: the key under which an upload is remembered"""
    h = _new_hash()
    for a in arrays:
        if a.size:
            _hash_bytes(h, a)
    return (tuple(a.shape for a in arrays), h.hexdigest())


class _PinnedBlock(object):
    """A page-locked host block (fz_host_alloc) behind ``__array_interface__``: the NumPy array made from it keeps it alive
    as its base; when the last view dies the block goes back to a small pool (page-locking 5.6 GB costs about as much
    as copying it) or to the runtime."""
    _pool = []                      # free blocks [(nbytes, ptr)], at most _POOL_MAX bytes in total
    # One result block of the largest call so far stays page-locked between calls (page-locking 5.6 GB costs about as much as
    # copying it); FRANKENZ_PINNED_POOL_GB sizes the pool (0: nothing is kept), ``release_pinned_pool()`` empties it.
    _POOL_MAX = int(float(os.environ.get("FRANKENZ_PINNED_POOL_GB", "6")) * (1 << 30))

    def __init__(self, ptr_, nbytes, shape, dtype):
        self.ptr, self.nbytes = ptr_, nbytes
        self.__array_interface__ = {"shape": tuple(shape), "typestr": np.dtype(dtype).str, "data": (ptr_, False), "version": 3}

    def __del__(self):
        try:
            pool = _PinnedBlock._pool
            if sum(b[0] for b in pool) + self.nbytes <= _PinnedBlock._POOL_MAX:
                pool.append((self.nbytes, self.ptr))
            else:
                _lib.load().fz_host_free(self.ptr)
        except Exception:           # interpreter shutdown
            pass


def release_pinned_pool():
    """give every pooled page-locked block back to the runtime (hosts short of unswappable memory, several ranks per node)"""
    pool = _PinnedBlock._pool
    while pool:
        _lib.load().fz_host_free(pool.pop()[1])


def pinned_empty(shape, dtype=np.float64, min_bytes=1 << 24):
    """Uninitialised result array in page-locked host memory (arrays below ``min_bytes`` and any allocation the runtime
    refuses: ordinary ``np.empty``).  A device-to-host copy into pageable memory is staged by the runtime (~14 GB/s
    measured, and it blocks the host); into page-locked memory it runs at the link rate behind the next chunk's kernel."""
    shape = tuple(int(v) for v in np.atleast_1d(shape)) if not isinstance(shape, tuple) else tuple(int(v) for v in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if nbytes < min_bytes:
        return np.empty(shape, dtype=dtype)
    pool = _PinnedBlock._pool
    best = None
    for k, (nb, _) in enumerate(pool):          # smallest free block that fits without wasting more than half of itself
        if nbytes <= nb <= 2 * nbytes and (best is None or nb < pool[best][0]):
            best = k
    if best is not None:
        nb, p = pool.pop(best)
    else:
        h = C.c_void_p()
        if _lib.load().fz_host_alloc(nbytes, C.byref(h)) != 0:
            while pool:                         # make room once, then give up on page-locking
                _lib.load().fz_host_free(pool.pop()[1])
            if _lib.load().fz_host_alloc(nbytes, C.byref(h)) != 0:
                return np.empty(shape, dtype=dtype)
        nb, p = nbytes, h.value
    return np.asarray(_PinnedBlock(p, nb, shape, dtype))


class Engine(object):
    """One device context.  Not thread-safe."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.fz_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self.M = self.B = 0
        self._models_key = self._dict_key = self._labels_key = self._trees_key = None      # content keys of what the device holds

    def close(self):
        if getattr(self, "h", None):
            self.lib.fz_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- uploads ----------------------------------------------------------
    # What is already on the device is not sent again.  The reference holds REFERENCES to the caller's model
    # arrays (bruteforce.py:54-56), so an in-place edit between two calls must be seen: the key is a hash of the
    # bytes (xxh3: ~1 ms per 12 MB), not the arrays' identity.
    def upload_models(self, models, models_err, models_mask):
        y, ye, ym = _f64(models), _f64(models_err), _f64(models_mask)
        if y.ndim != 2 or ye.shape != y.shape or ym.shape != y.shape:
            raise ValueError("models, models_err, models_mask must share a (Nmodel, Nfilt) shape")
        key = _digest(y, ye, ym)
        if key == self._models_key:
            return
        self._models_key = self._labels_key = None              # labels belong to the model set they were uploaded with
        check(self.lib.fz_models_upload(self.h, ptr(y), ptr(ye), ptr(ym), y.shape[0], y.shape[1]))
        self.M, self.B = y.shape
        self._models_key = key

    def upload_dict(self, pdfdict):
        lens = np.array([len(k) for k in pdfdict.sigma_dict], dtype=np.int64)
        offs = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        kern = _f64(np.concatenate(pdfdict.sigma_dict))
        kcdf = _f64(np.concatenate(pdfdict.sigma_dict_cdf))
        widths = np.ascontiguousarray(pdfdict.sigma_width, dtype=np.int64)
        key = (int(pdfdict.Ngrid),) + _digest(widths, offs, kern, kcdf)
        if key == self._dict_key:
            return
        self._dict_key = self._labels_key = None
        check(self.lib.fz_kdedict_upload(self.h, int(pdfdict.Ngrid), len(lens), ptr(widths),
                                         ptr(offs), ptr(kern), ptr(kcdf)))
        self._dict_key = key

    def upload_labels_dict(self, y_idx, y_std_idx):
        yi = np.ascontiguousarray(y_idx, dtype=np.int64)
        si = np.ascontiguousarray(y_std_idx, dtype=np.int64)
        key = ("dict", self._dict_key, self._models_key) + _digest(yi, si)
        if key == self._labels_key and self._dict_key is not None:
            return
        self._labels_key = None
        check(self.lib.fz_labels_upload_dict(self.h, ptr(yi), ptr(si), len(yi)))
        self._labels_key = key

    def upload_labels_grid(self, y, y_std, grid, dx=None, sig_thresh=5.0):
        y, ys, g = _f64(y), _f64(y_std), _f64(grid)
        if dx is None:
            dx = g[1] - g[0]
        key = ("grid", float(dx), float(sig_thresh), self._models_key) + _digest(y, ys, g)
        if key == self._labels_key:
            return
        self._labels_key = None              # (the library restores the dictionary's grid length itself when dictionary labels return)
        check(self.lib.fz_labels_upload_grid(self.h, ptr(y), ptr(ys), len(y), ptr(g), len(g),
                                             float(dx), float(sig_thresh)))
        self._labels_key = key

    def set_labels(self, labels, label_errs, label_dict=None, label_grid=None, kde_kwargs=None):
        """bruteforce.py:598-599 / 361-369: dictionary path if a PDFDict is given,
        else the direct KDE on ``label_grid``.  Returns Nx."""
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        kw = kde_kwargs or {}
        if label_dict is not None:
            yi, si = label_dict.fit(np.asarray(labels), np.asarray(label_errs))
            self.upload_dict(label_dict)
            self.upload_labels_dict(yi, si)
            return label_dict.Ngrid
        self.upload_labels_grid(labels, label_errs, label_grid, dx=kw.get("dx"),
                                sig_thresh=kw.get("sig_thresh", 5.0))
        return len(label_grid)

    # -- compute ----------------------------------------------------------
    def fit(self, x, xe, xm, opts, lnlike=None, chi2=None, ndim=None, scale=None, scale_err=None,
            n=None):
        n = len(x) if n is None else n
        check(self.lib.fz_fit(self.h, ptr(x), ptr(xe), ptr(xm), n, C.byref(opts), ptr(lnlike),
                              ptr(chi2), ptr(ndim), ptr(scale), ptr(scale_err)))

    def fit_predict(self, x, xe, xm, opts, kopts, pdfs, lmap=None, levid=None, n=None):
        n = len(x) if n is None else n
        check(self.lib.fz_fit_predict(self.h, ptr(x), ptr(xe), ptr(xm), n, C.byref(opts),
                                      C.byref(kopts), ptr(pdfs), ptr(lmap), ptr(levid)))

    @staticmethod
    def _prior_struct(prior):
        """(table, P, rows) -> fz_prior*, or NULL.  The arrays must outlive the call."""
        if prior is None:
            return None
        table, P, rows = prior
        return C.byref(Prior(ptr(table), int(P), ptr(rows)))

    def fit_prior(self, x, xe, xm, opts, prior, lnprior=None, lnlike=None, lnprob=None, chi2=None,
                  ndim=None, scale=None, scale_err=None, n=None):
        """fz_fit_prior: ``prior`` is ``(table (P,M), P, rows (n,) or None)`` or None."""
        n = len(x) if n is None else n
        check(self.lib.fz_fit_prior(self.h, ptr(x), ptr(xe), ptr(xm), n, C.byref(opts),
                                    self._prior_struct(prior), ptr(lnprior), ptr(lnlike), ptr(lnprob),
                                    ptr(chi2), ptr(ndim), ptr(scale), ptr(scale_err)))

    def fit_predict_prior(self, x, xe, xm, opts, kopts, prior, pdfs, lmap=None, levid=None, n=None):
        n = len(x) if n is None else n
        check(self.lib.fz_fit_predict_prior(self.h, ptr(x), ptr(xe), ptr(xm), n, C.byref(opts),
                                            C.byref(kopts), self._prior_struct(prior), ptr(pdfs),
                                            ptr(lmap), ptr(levid)))

    def predict_logwt(self, logwt, kopts, pdfs, lmap=None, levid=None, is_log=True, n=None):
        n = len(logwt) if n is None else n
        check(self.lib.fz_predict_logwt(self.h, ptr(logwt), n, int(bool(is_log)), C.byref(kopts),
                                        ptr(pdfs), ptr(lmap), ptr(levid)))

    # -- k-NN ------------------------------------------------------------
    def knn_upload_trees(self, feats, key=None):
        """``key``: the caller's content key of ``feats`` (``_digest``); an unchanged set is not sent (nor Morton-sorted) again"""
        f = np.ascontiguousarray(feats, dtype=np.float32)
        K, M, F = f.shape
        if key is None:
            key = _digest(f)
        if key == self._trees_key:
            return
        self._trees_key = None
        check(self.lib.fz_knn_upload_trees(self.h, ptr(f), K, M, F))
        self._trees_key = key

    def knn_query(self, q, k, distance_upper_bound, idx, n=None, lp_norm=2):
        n = len(q) if n is None else n
        check(self.lib.fz_knn_query(self.h, ptr(q), n, int(k), float(lp_norm), float(distance_upper_bound), ptr(idx)))

    def knn_fit_predict(self, x, xe, xm, idx, W, opts, kopts, neighbors=None, nnbr=None, lnlike=None,
                        chi2=None, ndim=None, scale=None, scale_err=None, pdfs=None, lmap=None,
                        levid=None, n=None):
        n = len(x) if n is None else n
        check(self.lib.fz_knn_fit_predict(self.h, ptr(x), ptr(xe), ptr(xm), n, ptr(idx), int(W),
                                          C.byref(opts), C.byref(kopts) if kopts is not None else None,
                                          ptr(neighbors), ptr(nnbr), ptr(lnlike), ptr(chi2), ptr(ndim),
                                          ptr(scale), ptr(scale_err), ptr(pdfs), ptr(lmap), ptr(levid)))

    def knn_fit_predict_prior(self, x, xe, xm, idx, W, opts, kopts, prior, neighbors=None, nnbr=None,
                              lnprior=None, lnlike=None, lnprob=None, chi2=None, ndim=None, scale=None,
                              scale_err=None, pdfs=None, lmap=None, levid=None, n=None):
        n = len(x) if n is None else n
        check(self.lib.fz_knn_fit_predict_prior(
            self.h, ptr(x), ptr(xe), ptr(xm), n, ptr(idx), int(W), C.byref(opts),
            C.byref(kopts) if kopts is not None else None, self._prior_struct(prior), ptr(neighbors),
            ptr(nnbr), ptr(lnprior), ptr(lnlike), ptr(lnprob), ptr(chi2), ptr(ndim), ptr(scale),
            ptr(scale_err), ptr(pdfs), ptr(lmap), ptr(levid)))

    def knn_predict_logwt(self, logwt, neighbors, nnbr, W, kopts, pdfs, lmap=None, levid=None, n=None):
        n = len(logwt) if n is None else n
        check(self.lib.fz_knn_predict_logwt(self.h, ptr(logwt), ptr(neighbors), ptr(nnbr), n, int(W),
                                            C.byref(kopts), ptr(pdfs), ptr(lmap), ptr(levid)))

    # -- inference through a trained network (networks.py; fz_net.h) -----------
    def net_select(self, lnprob, use_wt, wt_thresh, cdf_thresh, match, csr_off, nsel, sel, rawlen=None, lmap=None, levid=None):
        n, nn = lnprob.shape
        nnodes = 0 if csr_off is None else len(csr_off) - 1
        check(self.lib.fz_net_select(self.h, ptr(lnprob), n, nn, int(bool(use_wt)), float(wt_thresh), float(cdf_thresh), ptr(match),
                                     ptr(csr_off), nnodes, ptr(nsel), ptr(sel), ptr(rawlen), ptr(lmap), ptr(levid)))

    def net_table(self, nsel, sel, match, csr_off, csr_items, W, idx):
        n, nn = sel.shape
        check(self.lib.fz_net_table(self.h, ptr(nsel), ptr(sel), n, nn, ptr(match), ptr(csr_off), ptr(csr_items), len(csr_off) - 1,
                                    int(W), ptr(idx)))

    def net_gather(self, plane, nsel, sel, W, pad, out):
        n, nn = sel.shape
        bits = int(np.array([pad], dtype=plane.dtype).view(np.uint64)[0])
        check(self.lib.fz_net_gather(self.h, ptr(plane), ptr(nsel), ptr(sel), n, nn, int(W), bits, ptr(out)))

    def net_stack(self, lnprob, nsel, sel, match, node_pdfs, pdfs, lmap=None, levid=None):
        n, nn = sel.shape
        check(self.lib.fz_net_stack(self.h, ptr(lnprob), ptr(nsel), ptr(sel), n, nn, ptr(match), ptr(node_pdfs), node_pdfs.shape[0],
                                    node_pdfs.shape[1], ptr(pdfs), ptr(lmap), ptr(levid)))

    def pdfs_summarize(self, pdfs, pgrid, renormalize, urand, loss, widths, wscale, stats, n=None):
        n = len(pdfs) if n is None else n
        check(self.lib.fz_pdfs_summarize(self.h, ptr(pdfs), n, len(pgrid), ptr(pgrid), int(bool(renormalize)),
                                         ptr(urand), ptr(loss), ptr(widths), float(wscale), ptr(stats)))

    def pdfs_resample(self, pdfs, old_grid, new_grid, left, right, renormalize, out, n=None):
        n = len(pdfs) if n is None else n
        check(self.lib.fz_pdfs_resample(self.h, ptr(pdfs), n, len(old_grid), ptr(old_grid), len(new_grid), ptr(new_grid),
                                        float(left), float(right), int(bool(renormalize)), ptr(out)))

    def nz_assign(self, pdfs, nz, u, bins, counts, n=None):
        n = len(pdfs) if n is None else n
        check(self.lib.fz_nz_assign(self.h, ptr(pdfs), n, len(nz), ptr(nz), ptr(u), ptr(bins), ptr(counts)))

    def overlap_nz(self, pdfs, nz, pair, step, overlap, n=None):
        n = len(pdfs) if n is None else n
        out = np.zeros(1)
        pi, pj = (-1, -1) if pair is None else (int(pair[0]), int(pair[1]))
        check(self.lib.fz_overlap_nz(self.h, ptr(pdfs), n, len(nz), ptr(nz), pi, pj, float(step), ptr(overlap), ptr(out)))
        return float(out[0])

    def knn_search_fit_predict(self, q, x, xe, xm, k, lp_norm, distance_upper_bound, opts, kopts, prior=None,
                               neighbors=None, nnbr=None, lnprior=None, lnlike=None, lnprob=None, chi2=None, ndim=None,
                               scale=None, scale_err=None, pdfs=None, lmap=None, levid=None, n=None):
        """fz_knn_search_fit_predict_prior: the K searches and the subset likelihood / PDFs in one call, the neighbour
        table staying on the device"""
        n = len(x) if n is None else n
        check(self.lib.fz_knn_search_fit_predict_prior(
            self.h, ptr(q), ptr(x), ptr(xe), ptr(xm), n, int(k), float(lp_norm), float(distance_upper_bound),
            C.byref(opts), C.byref(kopts) if kopts is not None else None, self._prior_struct(prior), ptr(neighbors),
            ptr(nnbr), ptr(lnprior), ptr(lnlike), ptr(lnprob), ptr(chi2), ptr(ndim), ptr(scale), ptr(scale_err),
            ptr(pdfs), ptr(lmap), ptr(levid)))

    def set_producer_stream(self, stream=None, mode=1):
        """fz_set_producer_stream.  ``mode`` 0: device-wide wait before device inputs are read (default of a fresh
        engine); 1: wait for ``stream`` only (an int ``hipStream_t``, e.g. ``torch.cuda.current_stream().cuda_stream``;
        None / 0: the legacy default stream); 2: inputs are complete, no wait."""
        check(self.lib.fz_set_producer_stream(self.h, int(stream or 0), int(mode)))
        self._producer = (stream, int(mode))

    def producer_stream(self):
        """(stream, mode) of the last ``set_producer_stream`` (a fresh context: (None, 0)): what a caller that changes the contract
        for the length of one call puts back"""
        return getattr(self, "_producer", (None, 0))

    def modec_info(self):
        """mode C since the last ``timing_reset``: ``(ambiguous objects re-run with IEEE divisions, iterations of the
        slowest object, 1 = one block per object / 2 = state planes, threads per block)``"""
        out = np.zeros(4, dtype=np.int64)
        check(self.lib.fz_modec_info(self.h, ptr(out)))
        return tuple(int(v) for v in out)

    def modec_niter(self, n):
        """passes of the loop at pdf.py:199 each of the first ``n`` objects of the last mode-C chunk took"""
        out = np.zeros(int(n), dtype=np.int32)
        check(self.lib.fz_modec_niter(self.h, int(n), ptr(out)))
        return out

    def clean(self, x, xe, xm):
        check(self.lib.fz_clean(self.h, ptr(x), ptr(xe), ptr(xm), x.shape[0], x.shape[1]))

    def sync(self):
        check(self.lib.fz_sync(self.h))

    def timing_reset(self):
        """zero the per-family timers (and the mode-C bookkeeping of ``modec_info``)"""
        check(self.lib.fz_timing_reset(self.h))

    def timing(self):
        t = Timing()
        check(self.lib.fz_timing_get(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def last_form(self):
        """which kernel form the last fused fit_predict launch took (depends on the data)"""
        return self.lib.fz_last_form(self.h).decode("ascii", "replace")

    def set_workspace_limit(self, nbytes):
        check(self.lib.fz_set_workspace_limit(self.h, int(nbytes)))


_engines = {}


def get_engine(device=None):
    """Process-wide engine for ``device`` (default: LOCAL_RANK or 0)."""
    import os
    if device is None:
        device = int(os.environ.get("FRANKENZ_DEVICE", os.environ.get("LOCAL_RANK", 0)))
        ndev = _lib.load().fz_device_count()
        if ndev > 0:
            device %= ndev
    if device not in _engines:
        _engines[device] = Engine(device)
    return _engines[device]


class HostObjects(object):
    """float64 C-contiguous staging of (data, data_err, data_mask) that writes the
    in-place clean of pdf.py:310-311 back into the caller's arrays."""

    def __init__(self, data, data_err, data_mask):
        self.src = (data, data_err, data_mask)
        self.x, self.xe, self.xm = _f64(data), _f64(data_err), _f64(data_mask)
        if self.x.ndim != 2 or self.xe.shape != self.x.shape or self.xm.shape != self.x.shape:
            raise ValueError("data, data_err, data_mask must share a (Ndata, Nfilt) shape")

    def writeback(self):
        for dst, buf in zip(self.src, (self.x, self.xe, self.xm)):
            if isinstance(dst, np.ndarray) and dst is not buf and not np.shares_memory(dst, buf):
                dst[...] = buf
