"""
``BruteForce`` -- drop-in for frankenz/bruteforce.py:30-631 with the per-object
Python loops replaced by GPU kernels behind the C ABI.

Same constructor, methods (``fit / predict / fit_predict`` and the generator twins
``_fit / _predict / _fit_predict``), keyword names, defaults, attributes
(``fit_lnprior, fit_lnlike, fit_lnprob, fit_Ndim, fit_chi2, fit_scale,
fit_scale_err, NMODEL, NDIM, NDATA``), return shapes and dtypes.

``lprob_func`` may be ``None`` or this package's ``logprob`` (the reference default,
bruteforce.py:105-106), or a ``pdf.logprob_prior`` instance (default likelihood plus
an additive ln-prior table evaluated on the device).  Any other callable would need a
per-object host loop; that is refused loudly rather than silently run on the CPU.
"""
import sys

import numpy as np

from . import pdf as _pdf
from .engine import HostObjects, get_engine, kde_opts, like_opts

__all__ = ["BruteForce"]

_GEN_CHUNK = 1024     # objects per device call inside the generator twins


def _check_lprob(lprob_func, lprob_args, Nmodel=None):
    """-> the ``logprob_prior`` to apply, or None for the plain likelihood."""
    if lprob_args:
        raise NotImplementedError("positional `lprob_args` are not supported; use `lprob_kwargs`")
    if isinstance(lprob_func, _pdf.logprob_prior):
        if Nmodel is not None and lprob_func.M != Nmodel:
            raise ValueError("ln-prior rows hold %d models, the model set %d" % (lprob_func.M, Nmodel))
        return lprob_func
    if lprob_func is not None and lprob_func is not _pdf.logprob:
        raise NotImplementedError(
            "custom `lprob_func` callables are not supported by the HIP path; use the default "
            "logprob with `lprob_kwargs` (free_scale, ignore_model_err, dim_prior, ltol), or "
            "pdf.logprob_prior(lnprior_table, rows) for an additive ln-prior")
    return None


def _progress(verbose, what, i, n):
    if verbose:
        sys.stderr.write('\r{0} {1}/{2}'.format(what, i, n))
        sys.stderr.flush()


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
        prior = _check_lprob(lprob_func, lprob_args, self.NMODEL)
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
        prior = _check_lprob(lprob_func, lprob_args, self.NMODEL)
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
        if kde_args:
            raise NotImplementedError("positional `kde_args` are not supported; use `kde_kwargs`")
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
        pdfs = np.zeros((Ndata, Nx))
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
        if kde_args:
            raise NotImplementedError("positional `kde_args` are not supported; use `kde_kwargs`")
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
            pdfs = np.zeros((hi - lo, Nx))
            lmap, levid = np.zeros(hi - lo), np.zeros(hi - lo)
            eng.predict_logwt(lw, ko, pdfs, lmap, levid)
            for i in range(hi - lo):
                yield pdfs[i], (lmap[i], levid[i])

    # ------------------------------------------------------------------
    def fit_predict(self, data, data_err, data_mask, model_labels, model_label_errs,
                    lprob_func=None, label_dict=None, label_grid=None, kde_args=None,
                    kde_kwargs=None, lprob_args=None, lprob_kwargs=None, return_gof=False,
                    track_scale=False, verbose=True, save_fits=True):
        """bruteforce.py:374-503.  ``save_fits=False`` is the streaming path that never
        materialises (Ndata, Nmodel); ``save_fits=True`` additionally fills ``fit_*``."""
        prior = _check_lprob(lprob_func, lprob_args, self.NMODEL)
        if kde_args:
            raise NotImplementedError("positional `kde_args` are not supported; use `kde_kwargs`")
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        opts = like_opts(lprob_kwargs)
        ko = kde_opts(kde_kwargs)
        eng = self._engine()
        Nx = eng.set_labels(model_labels, model_label_errs, label_dict, label_grid, kde_kwargs)
        obj = HostObjects(data, data_err, data_mask)
        Ndata = len(obj.x)
        self._ndata_all = Ndata
        pdfs = np.zeros((Ndata, Nx))
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

    def _fit_predict(self, data, data_err, data_mask, model_labels, model_label_errs,
                     lprob_func=None, label_dict=None, label_grid=None, kde_args=None,
                     kde_kwargs=None, lprob_args=None, lprob_kwargs=None, track_scale=False,
                     save_fits=True):
        """Generator twin (bruteforce.py:505-631): yields ``(pdf, (lmap, levid))``."""
        prior = _check_lprob(lprob_func, lprob_args, self.NMODEL)
        if kde_args:
            raise NotImplementedError("positional `kde_args` are not supported; use `kde_kwargs`")
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
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
