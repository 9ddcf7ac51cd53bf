"""
``BruteForce`` -- drop-in for frankenz/bruteforce.py:30-631 with the per-object
Python loops replaced by GPU kernels behind the C ABI.

Same constructor, methods (``fit / predict / fit_predict`` and the generator twins
``_fit / _predict / _fit_predict``), keyword names, defaults, attributes
(``fit_lnprior, fit_lnlike, fit_lnprob, fit_Ndim, fit_chi2, fit_scale,
fit_scale_err, NMODEL, NDIM, NDATA``), return shapes and dtypes.

``lprob_func`` may be ``None`` or this package's ``logprob`` (the reference default,
bruteforce.py:105-106), or a ``pdf.logprob_prior`` instance (default likelihood plus
an additive ln-prior table evaluated on the device): those run on the GPU end to end.
Any other callable -- the reference's plugin hook, bruteforce.py:193-194, e.g. demos/2's
``lprob_bpz`` -- is the USER's code: it is called once per object on the host exactly as
the reference calls it (same arguments, ``lprob_args`` / ``lprob_kwargs`` included), its
``(lnprior, lnlike, lnprob, Ndim, chi2[, scale, scale_err])`` rows fill the ``fit_*``
arrays, and the softmax / KDE half (bruteforce.py:359-370, 619-629) still runs on the GPU
from the ln-posterior rows, a chunk at a time.  The likelihood then runs at the speed of
that callable.
"""
import sys

import numpy as np

from . import pdf as _pdf
from .engine import HostObjects, get_engine, kde_opts, like_opts, merge_kde_args, pinned_empty

__all__ = ["BruteForce"]

_GEN_CHUNK = 1024     # objects per device call inside the generator twins


_HOST_CHUNK = 256    # objects whose ln-posterior rows are handed to the GPU together (host-callable path)


class _HostFunc(object):
    """a user ``lprob_func`` with its ``lprob_args`` / ``lprob_kwargs``: called per object like
    bruteforce.py:193-194"""

    def __init__(self, func, args, kwargs):
        self.func, self.args, self.kwargs = func, list(args or []), dict(kwargs or {})

    def __call__(self, x, xe, xm, bf):
        return self.func(x, xe, xm, bf.models, bf.models_err, bf.models_mask, *self.args, **self.kwargs)


def _check_lprob(lprob_func, lprob_args, Nmodel=None, lprob_kwargs=None):
    """-> ``(prior, host)``: the ``logprob_prior`` to apply on the device (or None), and the
    per-object host callable (or None when the device computes the likelihood)."""
    if isinstance(lprob_func, _pdf.logprob_prior):
        if lprob_args:
            raise NotImplementedError("positional `lprob_args` are not supported with logprob_prior; use `lprob_kwargs`")
        if Nmodel is not None and lprob_func.M != Nmodel:
            raise ValueError("ln-prior rows hold %d models, the model set %d" % (lprob_func.M, Nmodel))
        return lprob_func, None
    if lprob_func is None or lprob_func is _pdf.logprob:
        if not lprob_args:
            return None, None
        lprob_func = _pdf.logprob          # positional arguments: the reference's call, per object
    if not callable(lprob_func):
        raise ValueError("`lprob_func` must be callable")
    return None, _HostFunc(lprob_func, lprob_args, lprob_kwargs)


def _check_logwt(lw, Ndata, Nlabels):
    """the C side reads ``Ndata`` rows of ``len(model_labels)`` weights: refuse anything else here
    (the reference would raise an IndexError / broadcast error from NumPy)"""
    if lw.ndim != 2 or lw.shape[1] != Nlabels or lw.shape[0] < Ndata:
        raise ValueError("`logwt` has shape %s; expected at least (%d, %d) = (Ndata, len(model_labels))"
                         % (lw.shape, Ndata, Nlabels))


def _progress(verbose, what, i, n):
    if verbose:
        sys.stderr.write('\r{0} {1}/{2}'.format(what, i, n))
        sys.stderr.flush()


class _Prepared(object):
    """A model set + labels resident on the device and the option structs of one ``fit_predict`` configuration
    (``BruteForce.prepare_fit_predict``)."""

    def __init__(self, bf, eng, opts, ko, Nx, prior, labels=None):
        self.bf, self.eng, self.opts, self.ko, self.Nx, self.prior = bf, eng, opts, ko, Nx, prior
        # the engine is shared by every fitter of the process: what the device held when this handle was made
        self._labels = labels
        self._keys = (eng._models_key, eng._dict_key, eng._labels_key)

    def _ensure_resident(self):
        """another fitter, ``predict()`` or label upload may have used the engine since: put this handle's model set and
        labels back (content keys make the check free when nothing changed)"""
        eng = self.eng
        if (eng._models_key, eng._dict_key, eng._labels_key) == self._keys:
            return
        if self._labels is None:
            raise RuntimeError("the device no longer holds the model set / labels this handle was prepared with")
        eng.upload_models(self.bf.models, self.bf.models_err, self.bf.models_mask)
        eng.set_labels(*self._labels)
        self._keys = (eng._models_key, eng._dict_key, eng._labels_key)

    def run(self, data, data_err, data_mask, out=None, save_fits=False, track_scale=False):
        """-> (pdfs, lmap, levid): ``out`` if given (float64, C-contiguous, NumPy or tensors on the engine's GPU), else fresh
        arrays of the kind ``data`` is.  Device tensors are cleaned in place by the library (pdf.py:310-311)."""
        bf, eng, Nx = self.bf, self.eng, self.Nx
        self._ensure_resident()
        on_dev = hasattr(data, "data_ptr")
        Ndata = int(data.shape[0])
        bf._ndata_all = Ndata
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
            if save_fits:
                bf.NDATA = Ndata
                bf._alloc_fits(Ndata)
                step = max(1, min(Ndata, (1 << 28) // max(bf.NMODEL, 1)))
                for lo in range(0, Ndata, step):
                    bf._fit_block(eng, obj, lo, min(Ndata, lo + step), self.opts, track_scale, self.prior)
        if Ndata:
            eng.fit_predict_prior(x, xe, xm, self.opts, self.ko,
                                  self.prior.chunk(0, Ndata, Ndata) if self.prior is not None else None,
                                  pdfs, lmap, levid, n=Ndata)
        if obj is not None:
            obj.writeback()
        return pdfs, lmap, levid


class BruteForce():
    """Fits data and generates predictions using a brute-force search over all
    models (bruteforce.py:30-34)."""

    def __init__(self, models, models_err, models_mask, device=None):
        # references, no copy (bruteforce.py:54-56)
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
        self._device = device
        self._ndata_all = None

    # ------------------------------------------------------------------
    def _engine(self):
        eng = get_engine(self._device)
        eng.upload_models(self.models, self.models_err, self.models_mask)
        return eng

    def _alloc_fits(self, Ndata):
        """bruteforce.py:182-189."""
        Nm = self.NMODEL
        self.fit_lnprior = np.zeros((Ndata, Nm), dtype='float')
        self.fit_lnlike = np.zeros((Ndata, Nm), dtype='float')
        self.fit_lnprob = np.zeros((Ndata, Nm), dtype='float')
        self.fit_Ndim = np.zeros((Ndata, Nm), dtype='int')
        self.fit_chi2 = np.zeros((Ndata, Nm), dtype='float')
        self.fit_scale = np.ones((Ndata, Nm), dtype='float')
        self.fit_scale_err = np.zeros((Ndata, Nm), dtype='float')

    def _fit_block(self, eng, obj, lo, hi, opts, track_scale, prior=None, off=0):
        """planes for objects [lo,hi) written straight into the fit_* arrays (bruteforce.py:
        195-203); lnprior = 0 and lnprob = lnlike without a prior (pdf.py:404-405).  ``off``:
        position of row 0 of ``obj`` in the whole data set (prior rows are global)."""
        sl = slice(lo, hi)
        free = bool(opts.free_scale)
        sc = self.fit_scale[sl] if (track_scale and free) else None
        se = self.fit_scale_err[sl] if (track_scale and free) else None
        pr = prior.chunk(off + lo, off + hi, self._ndata_all) if prior is not None else None
        eng.fit_prior(obj.x[sl], obj.xe[sl], obj.xm[sl], opts, pr, self.fit_lnprior[sl],
                      self.fit_lnlike[sl], self.fit_lnprob[sl], self.fit_chi2[sl], self.fit_Ndim[sl],
                      sc, se, n=hi - lo)

    def _store_row(self, i, results, track_scale):
        """bruteforce.py:195-203"""
        self.fit_lnprior[i] = results[0]
        self.fit_lnlike[i] = results[1]
        self.fit_lnprob[i] = results[2]
        self.fit_Ndim[i] = results[3]
        self.fit_chi2[i] = results[4]
        if track_scale:
            self.fit_scale[i] = results[5]
            self.fit_scale_err[i] = results[6]

    def _host_fit(self, host, data, data_err, data_mask, track_scale, save_fits):
        """bruteforce.py:191-205 with a user callable: a host loop, as in the reference."""
        Ndata = len(data)
        self.NDATA = Ndata
        if save_fits:
            self._alloc_fits(Ndata)
        for i, (x, xe, xm) in enumerate(zip(data, data_err, data_mask)):
            results = host(x, xe, xm, self)
            if save_fits:
                self._store_row(i, results, track_scale)
            yield results

    def _host_fit_predict(self, host, data, data_err, data_mask, model_labels, model_label_errs, label_dict,
                          label_grid, kde_kwargs, track_scale, save_fits):
        """bruteforce.py:602-631 with a user callable: its ln-posterior rows (bruteforce.py:616) go to
        the GPU ``_HOST_CHUNK`` objects at a time for max / logsumexp / KDE / normalisation."""
        ko = kde_opts(kde_kwargs)
        eng = get_engine(self._device)
        Ndata = len(data)
        if save_fits:
            self.NDATA = Ndata
            self._alloc_fits(Ndata)
        for lo in range(0, Ndata, _HOST_CHUNK):
            hi = min(Ndata, lo + _HOST_CHUNK)
            plane = np.empty((hi - lo, self.NMODEL))
            for i in range(lo, hi):
                results = host(data[i], data_err[i], data_mask[i], self)
                if save_fits:
                    self._store_row(i, results, track_scale)
                plane[i - lo] = results[2]
            # (after the callbacks: a callable that itself uses this package's likelihood re-uploads model
            # sets, and labels belong to the model set they were uploaded with)
            Nx = eng.set_labels(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)
            pdfs = np.zeros((hi - lo, Nx))
            lmap, levid = np.zeros(hi - lo), np.zeros(hi - lo)
            eng.predict_logwt(plane, ko, pdfs, lmap, levid, n=hi - lo)
            for i in range(hi - lo):
                yield pdfs[i], (lmap[i], levid[i])

    def _row_results(self, i, track_scale):
        r = (self.fit_lnprior[i], self.fit_lnlike[i], self.fit_lnprob[i], self.fit_Ndim[i],
             self.fit_chi2[i])
        if track_scale:
            r = r + (self.fit_scale[i], self.fit_scale_err[i])
        return r

    # ------------------------------------------------------------------
    def fit(self, data, data_err, data_mask, lprob_func=None, lprob_args=None, lprob_kwargs=None,
            track_scale=False, verbose=True):
        """bruteforce.py:66-125.  Fills the (Ndata, Nmodel) ``fit_*`` arrays."""
        prior, host = _check_lprob(lprob_func, lprob_args, self.NMODEL, lprob_kwargs)
        if host is not None:
            for i, _ in enumerate(self._host_fit(host, data, data_err, data_mask, track_scale, True)):
                _progress(verbose, 'Fitting object', i + 1, len(data))
            if verbose:
                sys.stderr.write('\n')
                sys.stderr.flush()
            return
        opts = like_opts(lprob_kwargs)
        eng = self._engine()
        obj = HostObjects(data, data_err, data_mask)
        Ndata = len(obj.x)
        self.NDATA = self._ndata_all = Ndata
        self._alloc_fits(Ndata)
        step = max(1, min(Ndata, (1 << 28) // max(self.NMODEL, 1)))
        for lo in range(0, Ndata, step):
            hi = min(Ndata, lo + step)
            self._fit_block(eng, obj, lo, hi, opts, track_scale, prior)
            _progress(verbose, 'Fitting object', hi, Ndata)
        obj.writeback()
        if verbose:
            sys.stderr.write('\n')
            sys.stderr.flush()

    def _fit(self, data, data_err, data_mask, lprob_func=None, lprob_args=None, lprob_kwargs=None,
             track_scale=False, save_fits=True):
        """Generator twin (bruteforce.py:127-205): yields the per-object result tuple."""
        prior, host = _check_lprob(lprob_func, lprob_args, self.NMODEL, lprob_kwargs)
        if host is not None:
            for results in self._host_fit(host, data, data_err, data_mask, track_scale, save_fits):
                yield results
            return
        opts = like_opts(lprob_kwargs)
        eng = self._engine()
        obj = HostObjects(data, data_err, data_mask)
        Ndata = len(obj.x)
        self.NDATA = Ndata
        keep = self if save_fits else BruteForce(self.models, self.models_err, self.models_mask,
                                                 self._device)
        keep._ndata_all = Ndata
        if save_fits:
            self._alloc_fits(Ndata)
        for lo in range(0, Ndata, _GEN_CHUNK):
            hi = min(Ndata, lo + _GEN_CHUNK)
            if not save_fits:
                keep._alloc_fits(hi - lo)
                sub = HostObjects(obj.x[lo:hi], obj.xe[lo:hi], obj.xm[lo:hi])
                keep._fit_block(eng, sub, 0, hi - lo, opts, track_scale, prior, off=lo)
            else:
                self._fit_block(eng, obj, lo, hi, opts, track_scale, prior)
            obj.writeback()
            for i in range(lo, hi):
                yield keep._row_results(i if save_fits else i - lo, track_scale)

    # ------------------------------------------------------------------
    def predict(self, model_labels, model_label_errs, label_dict=None, label_grid=None, logwt=None,
                kde_args=None, kde_kwargs=None, return_gof=False, verbose=True):
        """bruteforce.py:207-301."""
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
        lw = np.ascontiguousarray(logwt, dtype=np.float64)
        Ndata = self.NDATA if self.NDATA is not None else len(lw)
        _check_logwt(lw, Ndata, len(model_labels))
        pdfs = pinned_empty((Ndata, Nx))
        lmap, levid = np.zeros(Ndata), np.zeros(Ndata)
        eng.predict_logwt(lw, ko, pdfs, lmap, levid, n=Ndata)
        _progress(verbose, 'Generating PDF', Ndata, Ndata)
        if verbose:
            sys.stderr.write('\n')
            sys.stderr.flush()
        if return_gof:
            return pdfs, (lmap, levid)
        return pdfs

    def _predict(self, model_labels, model_label_errs, label_dict=None, label_grid=None, logwt=None,
                 kde_args=None, kde_kwargs=None):
        """Generator twin (bruteforce.py:303-372): yields ``(pdf, (lmap, levid))``."""
        kde_kwargs = merge_kde_args(kde_args, kde_kwargs, label_dict is not None)
        if logwt is None:
            logwt = self.fit_lnprob
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        eng = get_engine(self._device)
        Nx = eng.set_labels(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)
        ko = kde_opts(kde_kwargs)
        n = len(logwt)
        for lo in range(0, n, _GEN_CHUNK):
            hi = min(n, lo + _GEN_CHUNK)
            lw = np.ascontiguousarray(logwt[lo:hi], dtype=np.float64)
            _check_logwt(lw, hi - lo, len(model_labels))
            pdfs = np.zeros((hi - lo, Nx))
            lmap, levid = np.zeros(hi - lo), np.zeros(hi - lo)
            eng.predict_logwt(lw, ko, pdfs, lmap, levid)
            for i in range(hi - lo):
                yield pdfs[i], (lmap[i], levid[i])

    # ------------------------------------------------------------------
    def fit_predict(self, data, data_err, data_mask, model_labels, model_label_errs,
                    lprob_func=None, label_dict=None, label_grid=None, kde_args=None,
                    kde_kwargs=None, lprob_args=None, lprob_kwargs=None, return_gof=False,
                    track_scale=False, verbose=True, save_fits=True, out=None):
        """bruteforce.py:374-503.  ``save_fits=False`` is the streaming path that never
        materialises (Ndata, Nmodel); ``save_fits=True`` additionally fills ``fit_*``.

        Extension (no reference counterpart): ``out=(pdfs, lmap, levid)`` -- caller-allocated float64 arrays of
        shape (Ndata, Nx), (Ndata,), (Ndata,), NumPy or torch tensors on this engine's GPU -- receives the
        results in place and is what the call returns; with device tensors for ``data`` / ``out`` nothing
        crosses PCIe (the sharded driver gathers PDF shards straight out of such a buffer)."""
        prior, host = _check_lprob(lprob_func, lprob_args, self.NMODEL, lprob_kwargs)
        if out is not None or hasattr(data, "data_ptr"):
            if host is not None or save_fits and hasattr(data, "data_ptr"):
                raise NotImplementedError("device tensors / `out=` need the built-in likelihood and save_fits=False "
                                          "(the fit_* planes are host arrays)")
            kde_kwargs = merge_kde_args(kde_args, kde_kwargs, label_dict is not None)
            if label_dict is None and label_grid is None:
                raise ValueError("`label_dict` or `label_grid` must be specified.")
            return self._fit_predict_into(data, data_err, data_mask, model_labels, model_label_errs, label_dict, label_grid,
                                          kde_kwargs, lprob_kwargs, prior, return_gof, track_scale, save_fits, out)
        kde_kwargs = merge_kde_args(kde_args, kde_kwargs, label_dict is not None)
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        if host is not None:
            Ndata = len(data)
            rows = []
            for i, r in enumerate(self._host_fit_predict(host, data, data_err, data_mask, model_labels, model_label_errs,
                                                          label_dict, label_grid, kde_kwargs, track_scale, save_fits)):
                rows.append(r)
                _progress(verbose, 'Generating PDF', i + 1, Ndata)
            if verbose:
                sys.stderr.write('\n')
                sys.stderr.flush()
            pdfs = np.array([r[0] for r in rows]) if rows else np.zeros((0, 0))
            if return_gof:
                return pdfs, (np.array([r[1][0] for r in rows]), np.array([r[1][1] for r in rows]))
            return pdfs
        opts = like_opts(lprob_kwargs)
        ko = kde_opts(kde_kwargs)
        eng = self._engine()
        Nx = eng.set_labels(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)
        obj = HostObjects(data, data_err, data_mask)
        Ndata = len(obj.x)
        self._ndata_all = Ndata
        pdfs = pinned_empty((Ndata, Nx))         # every row is written by the library (the reference fills np.zeros row by row)
        lmap, levid = np.zeros(Ndata), np.zeros(Ndata)
        if save_fits:
            self.NDATA = Ndata
            self._alloc_fits(Ndata)
            step = max(1, min(Ndata, (1 << 28) // max(self.NMODEL, 1)))
            for lo in range(0, Ndata, step):
                self._fit_block(eng, obj, lo, min(Ndata, lo + step), opts, track_scale, prior)
        eng.fit_predict_prior(obj.x, obj.xe, obj.xm, opts, ko,
                              prior.chunk(0, Ndata, Ndata) if prior is not None else None,
                              pdfs, lmap, levid)
        obj.writeback()
        _progress(verbose, 'Generating PDF', Ndata, Ndata)
        if verbose:
            sys.stderr.write('\n')
            sys.stderr.flush()
        if return_gof:
            return pdfs, (lmap, levid)
        return pdfs

    def prepare_fit_predict(self, model_labels, model_label_errs, label_dict=None, label_grid=None, kde_kwargs=None,
                            lprob_kwargs=None, prior=None):
        """Extension: everything of ``fit_predict`` that does not depend on the objects -- model set, dictionary and labels
        on the device, option structs -- done once; the returned ``run(data, data_err, data_mask, out)`` then only moves
        objects.  A driver that feeds many blocks of objects through one model set (``sharded_fit_predict``'s rounds) pays
        the uploads / content checks once instead of per block."""
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        opts = like_opts(lprob_kwargs)
        ko = kde_opts(kde_kwargs)
        eng = self._engine()
        Nx = eng.set_labels(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)
        return _Prepared(self, eng, opts, ko, Nx, prior, labels=(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs))

    def _fit_predict_into(self, data, data_err, data_mask, model_labels, model_label_errs, label_dict, label_grid,
                          kde_kwargs, lprob_kwargs, prior, return_gof, track_scale, save_fits, out):
        """fit_predict with caller-owned outputs and / or device-resident objects (see ``fit_predict``)."""
        prep = self.prepare_fit_predict(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs, lprob_kwargs, prior)
        pdfs, lmap, levid = prep.run(data, data_err, data_mask, out, save_fits=save_fits, track_scale=track_scale)
        return (pdfs, (lmap, levid)) if return_gof else pdfs

    def _fit_predict(self, data, data_err, data_mask, model_labels, model_label_errs,
                     lprob_func=None, label_dict=None, label_grid=None, kde_args=None,
                     kde_kwargs=None, lprob_args=None, lprob_kwargs=None, track_scale=False,
                     save_fits=True):
        """Generator twin (bruteforce.py:505-631): yields ``(pdf, (lmap, levid))``."""
        prior, host = _check_lprob(lprob_func, lprob_args, self.NMODEL, lprob_kwargs)
        kde_kwargs = merge_kde_args(kde_args, kde_kwargs, label_dict is not None)
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        if host is not None:
            for r in self._host_fit_predict(host, data, data_err, data_mask, model_labels, model_label_errs, label_dict,
                                            label_grid, kde_kwargs, track_scale, save_fits):
                yield r
            return
        opts = like_opts(lprob_kwargs)
        ko = kde_opts(kde_kwargs)
        eng = self._engine()
        Nx = eng.set_labels(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)
        obj = HostObjects(data, data_err, data_mask)
        Ndata = len(obj.x)
        self._ndata_all = Ndata
        if save_fits:
            self.NDATA = Ndata
            self._alloc_fits(Ndata)
        for lo in range(0, Ndata, _GEN_CHUNK):
            hi = min(Ndata, lo + _GEN_CHUNK)
            if save_fits:
                self._fit_block(eng, obj, lo, hi, opts, track_scale, prior)
            pdfs = np.zeros((hi - lo, Nx))
            lmap, levid = np.zeros(hi - lo), np.zeros(hi - lo)
            eng.fit_predict_prior(obj.x[lo:hi], obj.xe[lo:hi], obj.xm[lo:hi], opts, ko,
                                  prior.chunk(lo, hi, Ndata) if prior is not None else None,
                                  pdfs, lmap, levid, n=hi - lo)
            obj.writeback()
            for i in range(hi - lo):
                yield pdfs[i], (lmap[i], levid[i])
