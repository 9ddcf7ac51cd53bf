"""
``NearestNeighbors`` -- drop-in for frankenz/knn.py:33-874 (the Monte-Carlo k-nearest-
neighbour variant, "KMCkNN") on the GPU.

Reference recipe: K Monte-Carlo realisations of the models -> K KDTrees in feature
(luptitude) space; per object one Monte-Carlo draw -> k neighbours from each tree ->
first-appearance union (<= K*k models) -> the usual likelihood / weights / KDE on that
subset.  Here the K tree queries are replaced by an EXACT brute-force top-k search
on the GPU (an exact answer satisfies KDTree.query's (1+eps) guarantee; it can differ
from SciPy's pick only among near-ties within eps), and the subset likelihood + PDF
stack is one wave per object.  The random draws stay on the host with the caller's
``numpy.random.RandomState`` so the stream -- and therefore the result -- is the
reference's.
"""
import copy
import sys

import numpy as np

from . import pdf as _pdf
from .bruteforce import _check_lprob, _progress
from .engine import HostObjects, _digest, get_engine, kde_opts, like_opts, merge_kde_args, pinned_empty

__all__ = ["NearestNeighbors"]

_GEN_CHUNK = 4096


class _FeatureSet(object):
    """One entry of the reference's ``KDTrees`` list (knn.py:186): holds the float32 Monte-Carlo feature set (``data``, as the tree
    would).  The library's own searches run over all K sets on the device and never build a tree; a caller that pokes at
    ``KDTrees[i]`` directly -- ``.query(...)``, ``.query_ball_point(...)`` -- gets the SciPy tree the reference would have handed
    out, built on first use."""

    def __init__(self, data, leafsize):
        self.data = data
        self.n, self.m = data.shape
        self.leafsize = leafsize
        self._tree = None

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        if self._tree is None:
            from scipy.spatial import KDTree
            self._tree = KDTree(self.data, leafsize=self.leafsize)      # knn.py:186
        return getattr(self._tree, name)


class _PreparedKnn(object):
    """Model set + feature sets + labels resident on the device and the options of one ``fit_predict(save_fits=False)``
    configuration (``NearestNeighbors.prepare_fit_predict``)."""

    def __init__(self, nn, eng, opts, ko, Nx, prior, labels):
        self.nn, self.eng, self.opts, self.ko, self.Nx, self.prior, self._labels = nn, eng, opts, ko, Nx, prior, labels
        self.search = (nn.k, nn.lp_norm, nn.dbound)
        self._keys = (eng._models_key, eng._dict_key, eng._labels_key, eng._trees_key)

    def _ensure_resident(self):
        eng = self.eng
        if (eng._models_key, eng._dict_key, eng._labels_key, eng._trees_key) == self._keys:
            return
        self.nn._engine()                           # models + feature sets (content keys: free when unchanged)
        eng.set_labels(*self._labels)
        self._keys = (eng._models_key, eng._dict_key, eng._labels_key, eng._trees_key)

    def run(self, data, data_err, data_mask, out=None, query_features=None, rstate=None):
        """-> (pdfs, lmap, levid).  ``data*``: NumPy arrays or contiguous float64 tensors on the engine's GPU (cleaned in place
        by the library, pdf.py:310-311); ``query_features``: (Ndata, Nfilt) float64, NumPy or device tensor -- when None they are
        drawn here from ``rstate`` exactly as knn.py:830-832 does (on the host: the stream is NumPy's), which needs the objects
        on the host (device tensors are copied back for that one step)."""
        nn, eng, Nx = self.nn, self.eng, self.Nx
        self._ensure_resident()
        on_dev = hasattr(data, "data_ptr")
        Ndata = int(data.shape[0])
        if query_features is None:
            hx, hxe = (data.cpu().numpy(), data_err.cpu().numpy()) if on_dev else (np.asarray(data), np.asarray(data_err))
            query_features = nn._query_features(hx, hxe, rstate if rstate is not None else np.random)
        q = query_features
        if not hasattr(q, "data_ptr"):
            q = np.ascontiguousarray(q, dtype=np.float64)
        elif str(q.dtype) != 'torch.float64' or not q.is_contiguous() or (on_dev and q.device != data.device) or \
                (q.is_cuda and q.device.index != eng.device):
            # (a tensor goes to the library as a raw pointer: anything else would be read as float64 (N, F) garbage)
            raise ValueError("a tensor `query_features` must be a contiguous float64 tensor on the engine's device")
        if tuple(q.shape) != (Ndata, nn.NDIM):
            raise ValueError("`query_features` must have shape (Ndata, Nfilt) = (%d, %d)" % (Ndata, nn.NDIM))
        if out is None:
            if on_dev:
                import torch
                out = (torch.empty((Ndata, Nx), dtype=torch.float64, device=data.device),
                       torch.empty(Ndata, dtype=torch.float64, device=data.device),
                       torch.empty(Ndata, dtype=torch.float64, device=data.device))
            else:
                out = (pinned_empty((Ndata, Nx)), np.zeros(Ndata), np.zeros(Ndata))
        pdfs, lmap, levid = out
        for a, shp in ((pdfs, (Ndata, Nx)), (lmap, (Ndata,)), (levid, (Ndata,))):
            if tuple(a.shape) != shp or str(a.dtype).split('.')[-1] != 'float64':
                raise ValueError("`out` must hold float64 arrays of shape (Ndata, Nx), (Ndata,), (Ndata,); got %s %s"
                                 % (tuple(a.shape), a.dtype))
            if (hasattr(a, "is_contiguous") and not a.is_contiguous()) or (isinstance(a, np.ndarray) and not a.flags.c_contiguous):
                raise ValueError("`out` arrays must be C-contiguous")
        if on_dev:
            for a in (data, data_err, data_mask):
                if tuple(a.shape) != tuple(data.shape) or not a.is_contiguous() or str(a.dtype) != 'torch.float64':
                    raise ValueError("device objects must be contiguous float64 tensors of one (Ndata, Nfilt) shape")
            x, xe, xm, obj = data, data_err, data_mask, None
        else:
            obj = HostObjects(data, data_err, data_mask)
            x, xe, xm = obj.x, obj.xe, obj.xm
        k, lp_norm, dbound = self.search
        if Ndata:
            eng.knn_search_fit_predict(q, x, xe, xm, k, lp_norm, dbound, self.opts, self.ko,
                                       self.prior.chunk(0, Ndata, Ndata) if self.prior is not None else None,
                                       pdfs=pdfs, lmap=lmap, levid=levid, n=Ndata)
        if obj is not None:
            obj.writeback()
        return pdfs, lmap, levid


class NearestNeighbors():
    """Fits data and generates predictions using k-nearest neighbours over Monte-Carlo
    realisations of the models (knn.py:33-38)."""

    def __init__(self, models, models_err, models_mask, leafsize=50, K=25, feature_map='luptitude',
                 fmap_args=None, fmap_kwargs=None, rstate=None, verbose=True, device=None):
        self.models = models
        self.models_err = models_err
        self.models_mask = models_mask
        self.NMODEL, self.NDIM = models.shape
        self.NDATA = None
        self.fit_lnprior = None
        self.fit_lnlike = None
        self.fit_lnprob = None
        self.fit_Ndim = None
        self.fit_chi2 = None
        self.fit_scale = None
        self.fit_scale_err = None
        self.leafsize = leafsize
        self.K = K
        self.KDTrees = None
        self.neighbors = None
        self.Nneighbors = None
        self.k = None
        self.eps = None
        self.p = None
        self.lp_norm = None
        self.dbound = None
        self._device = device
        if fmap_args is None:
            fmap_args = []
        if fmap_kwargs is None:
            fmap_kwargs = dict()
        self.fmap_args = fmap_args
        self.fmap_kwargs = fmap_kwargs
        if feature_map == 'identity':
            def feature_map(x, xe, *args, **kwargs):
                return x, xe
        elif feature_map == 'magnitude':
            feature_map = _pdf.magnitude
        elif feature_map == 'luptitude':
            feature_map = _pdf.luptitude
        else:
            # knn.py:131-140 validates a callable against undefined names and therefore
            # always ends here
            raise ValueError("The provided feature map is not valid.")
        self.feature_map = feature_map
        if rstate is None:
            rstate = np.random
        self.KDTrees = []
        for i, tree in enumerate(self._train_kdtrees(rstate=rstate)):
            if verbose:
                sys.stderr.write("\r{0}/{1} KDTrees constructed".format(i + 1, self.K))
                sys.stderr.flush()
            self.KDTrees.append(tree)
        if verbose:
            sys.stderr.write("\n")
            sys.stderr.flush()
        self._uploaded = False

    def _train_kdtrees(self, rstate=None):
        """knn.py:158-188: one float32 Monte-Carlo feature set per "tree" (same RNG calls,
        same float32 roundings as the reference)."""
        if rstate is None:
            rstate = np.random
        for i in range(self.K):
            models_t = np.array(rstate.normal(self.models, self.models_err), dtype='float32')
            Y_t, Ye_t = np.array(self.feature_map(models_t, self.models_err, *self.fmap_args,
                                                  **self.fmap_kwargs), dtype='float32')
            yield _FeatureSet(Y_t, self.leafsize)

    # ------------------------------------------------------------------
    def _engine(self):
        eng = get_engine(self._device)
        eng.upload_models(self.models, self.models_err, self.models_mask)
        if getattr(self, "_feats", None) is None:                 # the K feature sets are fixed at construction (knn.py:158-188)
            self._feats = np.ascontiguousarray(np.stack([t.data for t in self.KDTrees]), dtype=np.float32)
            self._feats_key = _digest(self._feats)
        eng.knn_upload_trees(self._feats, key=self._feats_key)
        return eng

    def _search_setup(self, k, eps, lp_norm, distance_upper_bound):
        if lp_norm not in (1, 2, np.inf):
            raise NotImplementedError("Minkowski norms 1, 2 and inf are implemented on the GPU (got %r)" % (lp_norm,))
        # (k > 64 is served by the Euclidean matrix-pipe search only: the library refuses it for the other norms / > 6 features)
        if k > 256 or self.K * k > 4096:
            raise NotImplementedError("k <= 256 and K*k <= 4096 are required (got k=%d, K=%d)" % (k, self.K))
        self.k = k
        self.eps = eps              # the search is exact; any eps >= 0 is honoured
        self.lp_norm = lp_norm
        self.dbound = distance_upper_bound

    def _query_features(self, data, data_err, rstate):
        """knn.py:830-832 for every object at once: the (N,B) draw consumes the RNG stream
        exactly like the reference's per-object draws."""
        x_t = rstate.normal(data, data_err)
        y_t, _ = self.feature_map(x_t, data_err, *self.fmap_args, **self.fmap_kwargs)
        return np.ascontiguousarray(y_t, dtype=np.float64)

    def _alloc_fits(self, Ndata):
        """knn.py:812-821."""
        W = self.K * self.k
        inf = np.inf
        self.Nneighbors = np.zeros(Ndata, dtype='int')
        self.neighbors = np.zeros((Ndata, W), dtype='int') - 99
        self.fit_lnprior = np.zeros((Ndata, W), dtype='float') - inf
        self.fit_lnlike = np.zeros((Ndata, W), dtype='float') - inf
        self.fit_lnprob = np.zeros((Ndata, W), dtype='float') - inf
        self.fit_Ndim = np.zeros((Ndata, W), dtype='int')
        self.fit_chi2 = np.zeros((Ndata, W), dtype='float') + inf
        self.fit_scale = np.ones((Ndata, W), dtype='float')
        self.fit_scale_err = np.zeros((Ndata, W), dtype='float')

    def _run(self, eng, obj, q, lo, hi, opts, ko, track_scale, save_fits, pdfs=None, lmap=None, levid=None,
             prior=None):
        """search + subset likelihood (+ PDFs) for objects [lo,hi): ONE library call (fz_knn_search_fit_predict_prior), the
        (n, K*k) neighbour table of knn.py:834-837 stays on the device between the K searches and the subset kernel -- it
        only comes back, de-duplicated (knn.py:840), as ``self.neighbors`` when fits are stored."""
        n = hi - lo
        kw = {}
        if save_fits:
            sl = slice(lo, hi)
            free = bool(opts.free_scale)
            kw = dict(neighbors=self.neighbors[sl], nnbr=self.Nneighbors[sl], lnlike=self.fit_lnlike[sl],
                      chi2=self.fit_chi2[sl], ndim=self.fit_Ndim[sl],
                      scale=self.fit_scale[sl] if (track_scale and free) else None,
                      scale_err=self.fit_scale_err[sl] if (track_scale and free) else None)
            if prior is not None:
                # additive ln-prior (pdf.logprob_prior): the three probability planes come from the device
                kw.update(lnprior=self.fit_lnprior[sl], lnprob=self.fit_lnprob[sl])
        eng.knn_search_fit_predict(q[lo:hi], obj.x[lo:hi], obj.xe[lo:hi], obj.xm[lo:hi], self.k, self.lp_norm, self.dbound,
                                   opts, ko, prior.chunk(lo, hi, len(obj.x)) if prior is not None else None,
                                   pdfs=pdfs, lmap=lmap, levid=levid, n=n, **kw)
        if save_fits and prior is None:
            W = self.K * self.k
            self.fit_lnprob[sl] = self.fit_lnlike[sl]
            valid = np.arange(W)[None, :] < self.Nneighbors[sl][:, None]
            self.fit_lnprior[sl] = np.where(valid, 0.0, -np.inf)          # knn.py:852: zeros on the subset

    def _host_run(self, host, data, data_err, data_mask, rstate, track_scale, labels=None):
        """knn.py:355-388 / 826-874 with a user ``lprob_func``: the K searches run on the GPU, the
        first-appearance de-duplication (``pandas.unique``, knn.py:840) on the host,
        the callable is evaluated per object on its neighbour subset like the reference does, and -- with
        ``labels = (model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)`` -- the PDFs come
        from the GPU out of the stored ln-posterior rows.  Fills the padded ``fit_*`` arrays."""
        if rstate is None:
            rstate = np.random
        eng = self._engine()
        q = self._query_features(np.asarray(data), np.asarray(data_err), rstate)
        Ndata = len(data)
        self.NDATA = Ndata
        self._alloc_fits(Ndata)
        # neighbours only: the K searches on the GPU (in chunks), pandas.unique's first-appearance order on the host (knn.py:834-840);
        # no default likelihood is run for rows the callable fills anyway
        W = self.K * self.k
        idx = np.empty((Ndata, W), dtype=np.int64)
        for lo in range(0, Ndata, 1 << 16):
            hi = min(lo + (1 << 16), Ndata)
            eng.knn_query(q[lo:hi], self.k, self.dbound, idx[lo:hi], n=hi - lo, lp_norm=self.lp_norm)
        if Ndata and (idx.min() < 0 or idx.max() >= self.NMODEL):
            raise IndexError("index %d is out of bounds for axis 0 with size %d" % (self.NMODEL, self.NMODEL))   # KDTree's "missing" index (knn.py:847)
        for i in range(Ndata):
            _, first = np.unique(idx[i], return_index=True)
            nb = idx[i][np.sort(first)]
            self.Nneighbors[i] = len(nb)
            self.neighbors[i, :len(nb)] = nb
        inf = np.inf
        self.fit_lnprior[:] = -inf; self.fit_lnlike[:] = -inf; self.fit_lnprob[:] = -inf
        self.fit_Ndim[:] = 0; self.fit_chi2[:] = inf; self.fit_scale[:] = 1.; self.fit_scale_err[:] = 0.
        out = []
        for i, (x, xe, xm) in enumerate(zip(data, data_err, data_mask)):
            n = self.Nneighbors[i]
            idxs = self.neighbors[i, :n]
            results = host.func(x, xe, xm, self.models[idxs], self.models_err[idxs], self.models_mask[idxs],
                                *host.args, **host.kwargs)
            self.fit_lnprior[i, :n] = results[0]
            self.fit_lnlike[i, :n] = results[1]
            self.fit_lnprob[i, :n] = results[2]
            self.fit_Ndim[i, :n] = results[3]
            self.fit_chi2[i, :n] = results[4]
            if track_scale:
                self.fit_scale[i, :n] = results[5]
                self.fit_scale_err[i, :n] = results[6]
            out.append((idxs, n, results))
        if labels is None:
            return out
        ml, mle, ld, lg, kk = labels
        return self.predict(ml, mle, label_dict=ld, label_grid=lg, kde_kwargs=kk, return_gof=True, verbose=False)

    # ------------------------------------------------------------------
    def fit(self, data, data_err, data_mask, lprob_func=None, rstate=None, k=20, eps=1e-3, lp_norm=2,
            distance_upper_bound=np.inf, lprob_args=None, lprob_kwargs=None, track_scale=False, verbose=True):
        """knn.py:190-279."""
        prior, host = _check_lprob(lprob_func, lprob_args, self.NMODEL, lprob_kwargs)
        opts = like_opts(lprob_kwargs) if host is None else None
        if rstate is None:
            rstate = np.random
        self._search_setup(k, eps, lp_norm, distance_upper_bound)
        if host is not None:
            self._host_run(host, data, data_err, data_mask, rstate, track_scale)
            _progress(verbose, 'Fitting object', len(data), len(data))
            if verbose:
                sys.stderr.write('\n')
                sys.stderr.flush()
            return
        eng = self._engine()
        q = self._query_features(np.asarray(data), np.asarray(data_err), rstate)
        obj = HostObjects(data, data_err, data_mask)
        Ndata = len(obj.x)
        self.NDATA = Ndata
        self._alloc_fits(Ndata)
        self._run(eng, obj, q, 0, Ndata, opts, None, track_scale, True, prior=prior)
        obj.writeback()
        _progress(verbose, 'Fitting object', Ndata, Ndata)
        if verbose:
            sys.stderr.write('\n')
            sys.stderr.flush()

    def _fit(self, data, data_err, data_mask, lprob_func=None, rstate=None, lprob_args=None, lprob_kwargs=None,
             track_scale=False, save_fits=True):
        """Generator twin (knn.py:281-388): yields ``(idxs, Nidx, results)`` per object;
        uses the ``k / eps / lp_norm / dbound`` attributes like the reference."""
        prior, host = _check_lprob(lprob_func, lprob_args, self.NMODEL, lprob_kwargs)
        opts = like_opts(lprob_kwargs) if host is None else None
        if rstate is None:
            rstate = np.random
        self._search_setup(self.k, self.eps, self.lp_norm, self.dbound)
        if host is not None:
            keep = self if save_fits else copy.copy(self)
            for r in keep._host_run(host, data, data_err, data_mask, rstate, track_scale):
                yield r
            return
        eng = self._engine()
        q = self._query_features(np.asarray(data), np.asarray(data_err), rstate)
        obj = HostObjects(data, data_err, data_mask)
        Ndata = len(obj.x)
        self.NDATA = Ndata
        keep = self if save_fits else copy.copy(self)      # scratch holder when fits are not kept
        keep._alloc_fits(Ndata)
        keep._run(eng, obj, q, 0, Ndata, opts, None, track_scale, True, prior=prior)
        obj.writeback()
        for i in range(Ndata):
            n = keep.Nneighbors[i]
            res = (keep.fit_lnprior[i, :n], keep.fit_lnlike[i, :n], keep.fit_lnprob[i, :n],
                   keep.fit_Ndim[i, :n], keep.fit_chi2[i, :n])
            if track_scale:
                res = res + (keep.fit_scale[i, :n], keep.fit_scale_err[i, :n])
            yield keep.neighbors[i, :n], n, res

    # ------------------------------------------------------------------
    def predict(self, model_labels, model_label_errs, label_dict=None, label_grid=None, logwt=None,
                kde_args=None, kde_kwargs=None, return_gof=False, verbose=True):
        """knn.py:390-486."""
        kde_kwargs = merge_kde_args(kde_args, kde_kwargs, label_dict is not None)
        if logwt is None:
            logwt = self.fit_lnprob
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        if self.fit_lnprob is None and logwt is None:
            raise ValueError("Fits have not been computed and weights have not been provided.")
        eng = get_engine(self._device)
        Nx = eng.set_labels(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)
        ko = kde_opts(kde_kwargs)
        Ndata = self.NDATA
        W = self.neighbors.shape[1]
        pdfs = pinned_empty((Ndata, Nx))
        lmap, levid = np.zeros(Ndata), np.zeros(Ndata)
        lw = np.ascontiguousarray(logwt, dtype=np.float64)
        nb = np.ascontiguousarray(self.neighbors, dtype=np.int64)
        nn = np.ascontiguousarray(self.Nneighbors, dtype=np.int64)
        if lw.shape != (Ndata, W) or nb.shape != (Ndata, W) or nn.shape != (Ndata,):
            raise ValueError("`logwt` has shape %s; expected (Ndata, K*k) = (%d, %d) like `neighbors`" % (lw.shape, Ndata, W))
        eng.knn_predict_logwt(lw, nb, nn, W, ko, pdfs, lmap, levid, n=Ndata)
        _progress(verbose, 'Generating PDF', Ndata, Ndata)
        if verbose:
            sys.stderr.write('\n')
            sys.stderr.flush()
        if return_gof:
            return pdfs, (lmap, levid)
        return pdfs

    def _predict(self, model_labels, model_label_errs, label_dict=None, label_grid=None, logwt=None,
                 kde_args=None, kde_kwargs=None):
        """Generator twin (knn.py:488-558)."""
        pdfs, (lmap, levid) = self.predict(model_labels, model_label_errs, label_dict=label_dict,
                                           label_grid=label_grid, logwt=logwt, kde_args=kde_args,
                                           kde_kwargs=kde_kwargs, return_gof=True, verbose=False)
        for i in range(len(pdfs)):
            yield pdfs[i], (lmap[i], levid[i])

    # ------------------------------------------------------------------
    def fit_predict(self, data, data_err, data_mask, model_labels, model_label_errs, lprob_func=None,
                    rstate=None, k=20, eps=1e-3, lp_norm=2, distance_upper_bound=np.inf, label_dict=None,
                    label_grid=None, kde_args=None, kde_kwargs=None, lprob_args=None, lprob_kwargs=None,
                    return_gof=False, track_scale=False, verbose=True, save_fits=True, out=None,
                    query_features=None):
        """knn.py:560-720.

        Extensions (no reference counterpart): ``out=(pdfs, lmap, levid)`` -- caller-allocated float64 arrays, NumPy or
        tensors on this engine's GPU -- receives the results in place; ``data`` / ``data_err`` / ``data_mask`` may be
        device tensors; ``query_features`` (Ndata, Nfilt) -- the objects' Monte-Carlo features of knn.py:830-832, NumPy or
        device tensor -- replaces the draw from ``rstate`` (a driver that shards the objects draws them once for all
        ranks).  With device tensors throughout nothing crosses PCIe: query features, neighbour table and PDFs stay in HBM."""
        prior, host = _check_lprob(lprob_func, lprob_args, self.NMODEL, lprob_kwargs)
        kde_kwargs = merge_kde_args(kde_args, kde_kwargs, label_dict is not None)
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        if rstate is None:
            rstate = np.random
        self._search_setup(k, eps, lp_norm, distance_upper_bound)
        if out is not None or query_features is not None or hasattr(data, "data_ptr"):
            if host is not None or (save_fits and (out is not None or hasattr(data, "data_ptr"))):
                raise NotImplementedError("device tensors / `out=` need the built-in likelihood and save_fits=False "
                                          "(the fit_* arrays are host arrays)")
            if not save_fits:
                prep = self.prepare_fit_predict(model_labels, model_label_errs, label_dict=label_dict, label_grid=label_grid,
                                                kde_kwargs=kde_kwargs, lprob_kwargs=lprob_kwargs, prior=prior, k=k, eps=eps,
                                                lp_norm=lp_norm, distance_upper_bound=distance_upper_bound)
                pdfs, lmap, levid = prep.run(data, data_err, data_mask, out=out, query_features=query_features, rstate=rstate)
                return (pdfs, (lmap, levid)) if return_gof else pdfs
        if host is not None:
            keep = self if save_fits else copy.copy(self)
            pdfs, gof = keep._host_run(host, data, data_err, data_mask, rstate, track_scale,
                                       labels=(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs))
            _progress(verbose, 'Generating PDF', len(data), len(data))
            if verbose:
                sys.stderr.write('\n')
                sys.stderr.flush()
            return (pdfs, gof) if return_gof else pdfs
        opts = like_opts(lprob_kwargs)
        ko = kde_opts(kde_kwargs)
        eng = self._engine()
        Nx = eng.set_labels(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)
        q = (np.ascontiguousarray(query_features, dtype=np.float64) if query_features is not None
             else self._query_features(np.asarray(data), np.asarray(data_err), rstate))
        obj = HostObjects(data, data_err, data_mask)
        Ndata = len(obj.x)
        pdfs = pinned_empty((Ndata, Nx))
        lmap, levid = np.zeros(Ndata), np.zeros(Ndata)
        if save_fits:
            self.NDATA = Ndata
            self._alloc_fits(Ndata)
        self._run(eng, obj, q, 0, Ndata, opts, ko, track_scale, save_fits, pdfs, lmap, levid, prior=prior)
        obj.writeback()
        _progress(verbose, 'Generating PDF', Ndata, Ndata)
        if verbose:
            sys.stderr.write('\n')
            sys.stderr.flush()
        if return_gof:
            return pdfs, (lmap, levid)
        return pdfs

    def prepare_fit_predict(self, model_labels, model_label_errs, label_dict=None, label_grid=None, kde_kwargs=None,
                            lprob_kwargs=None, prior=None, k=20, eps=1e-3, lp_norm=2, distance_upper_bound=np.inf):
        """Extension (twin of ``BruteForce.prepare_fit_predict``): model set, feature sets, dictionary and labels on the
        device and the option structs of one ``fit_predict(save_fits=False)`` configuration, done once; the returned
        ``run(data, data_err, data_mask, out=, query_features=)`` only moves objects."""
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        self._search_setup(k, eps, lp_norm, distance_upper_bound)
        opts = like_opts(lprob_kwargs)
        ko = kde_opts(kde_kwargs)
        eng = self._engine()
        Nx = eng.set_labels(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)
        return _PreparedKnn(self, eng, opts, ko, Nx, prior, (model_labels, model_label_errs, label_dict, label_grid, kde_kwargs))

    def _fit_predict(self, data, data_err, data_mask, model_labels, model_label_errs, lprob_func=None,
                     rstate=None, label_dict=None, label_grid=None, kde_args=None, kde_kwargs=None,
                     lprob_args=None, lprob_kwargs=None, track_scale=False, save_fits=True):
        """Generator twin (knn.py:722-874); uses the ``k / eps / lp_norm / dbound`` attributes."""
        pdfs, (lmap, levid) = self.fit_predict(
            data, data_err, data_mask, model_labels, model_label_errs, lprob_func=lprob_func, rstate=rstate,
            k=self.k, eps=self.eps, lp_norm=self.lp_norm, distance_upper_bound=self.dbound,
            label_dict=label_dict, label_grid=label_grid, kde_args=kde_args, kde_kwargs=kde_kwargs,
            lprob_args=lprob_args, lprob_kwargs=lprob_kwargs, return_gof=True, track_scale=track_scale,
            verbose=False, save_fits=save_fits)
        for i in range(len(pdfs)):
            yield pdfs[i], (lmap[i], levid[i])
