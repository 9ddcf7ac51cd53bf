"""
Function-level surface of ``frankenz.pdf`` for the hot path (reference
frankenz/pdf.py): ``logprob`` / ``loglike`` (pdf.py:238-411), ``gauss_kde`` /
``gauss_kde_dict`` (pdf.py:444-622), ``PDFDict`` (pdf.py:778-852) and the k-NN
feature maps ``luptitude`` / ``magnitude`` (pdf.py:625-657, 695-734).

The likelihood and KDE arithmetic runs on the GPU through the C ABI.  ``PDFDict``
(a one-off table of a few hundred short kernels) and the feature maps (O(N*B)
elementwise set-up for the neighbour search) are host-side set-up, as in the
reference; their tables are uploaded verbatim so the device reproduces the
reference's kernels, malformed entries included.
"""
import math

import numpy as np

from .engine import HostObjects, get_engine, kde_opts, like_opts

__all__ = ["loglike", "logprob", "logprob_prior", "gaussian", "gauss_kde", "gauss_kde_dict",
           "magnitude", "luptitude", "PDFDict", "pdfs_summarize", "pdfs_resample"]


def _ndim_dtype(data_mask, models_mask):
    """np.sum(data_mask * models_mask, axis=1) keeps the masks' arithmetic type
    (pdf.py:82-83): bool/int masks give int64 counts, float masks float64."""
    t = np.result_type(np.asarray(data_mask).dtype, np.asarray(models_mask).dtype)
    return np.float64 if np.issubdtype(t, np.floating) else np.int64


def loglike(data, data_err, data_mask, models, models_err, models_mask, free_scale=False,
            ignore_model_err=False, dim_prior=True, ltol=1e-4, return_scale=False, device=None,
            *args, **kwargs):
    """One object against all models (pdf.py:238-323).  ``data``, ``data_err`` and
    ``data_mask`` are cleaned IN PLACE exactly like the reference (pdf.py:310-311).
    Returns ``(lnlike, Ndim, chi2[, scale, scale_err])``."""
    eng = get_engine(device)
    eng.upload_models(models, models_err, models_mask)
    obj = HostObjects(np.atleast_2d(data), np.atleast_2d(data_err), np.atleast_2d(data_mask))
    M = eng.M
    lnl, chi2 = np.empty((1, M)), np.empty((1, M))
    ndim = np.empty((1, M), dtype=np.int64)
    sc = se = None
    if free_scale and return_scale:
        sc, se = np.empty((1, M)), np.empty((1, M))
    opts = like_opts(dict(free_scale=free_scale, ignore_model_err=ignore_model_err,
                          dim_prior=dim_prior, ltol=ltol))
    eng.fit(obj.x, obj.xe, obj.xm, opts, lnl, chi2, ndim, sc, se)
    # in-place clean visible to the caller
    for dst, buf in ((data, obj.x), (data_err, obj.xe), (data_mask, obj.xm)):
        if isinstance(dst, np.ndarray) and not np.shares_memory(dst, buf):
            dst[...] = buf.reshape(dst.shape)
    nd = ndim[0].astype(_ndim_dtype(data_mask, models_mask))
    if free_scale and return_scale:
        return lnl[0], nd, chi2[0], sc[0], se[0]
    return lnl[0], nd, chi2[0]


def logprob(data, data_err, data_mask, models, models_err, models_mask, free_scale=False,
            ignore_model_err=False, dim_prior=True, ltol=1e-4, return_scale=False, *args,
            **kwargs):
    """pdf.py:326-411: ``(lnprior=0, lnlike, lnprob, Ndim, chi2[, scale, scale_err])``."""
    res = loglike(data, data_err, data_mask, models, models_err, models_mask,
                  free_scale=free_scale, ignore_model_err=ignore_model_err, dim_prior=dim_prior,
                  ltol=ltol, return_scale=return_scale, *args, **kwargs)
    lnl = res[0]
    return (np.zeros_like(lnl), lnl, lnl.copy()) + tuple(res[1:])       # lnprob = lnlike + lnprior is a fresh array in the reference (pdf.py:405)


class logprob_prior(object):
    """``lprob_func`` with an additive ln-prior, passed as DATA so that it can run on
    the device (extension; the reference's hook is a Python callable returning
    ``(lnprior, lnlike, lnlike + lnprior, ...)`` per object -- bruteforce.py:193-199,
    demos/2 cell 69's ``lprob_bpz``).

    ``lnprior`` is a ``(P, Nmodel)`` float64 table of ln-prior rows (NumPy array or a
    device tensor exposing ``data_ptr()``; ``(Nmodel,)`` means one shared row) and
    ``rows`` the ``(Ndata,)`` row each object reads.  Without ``rows``: ``P == 1``
    broadcasts, ``P == Ndata`` is a dense per-object prior.  ``lprob_kwargs`` keep
    their meaning (free_scale, ignore_model_err, dim_prior, ltol).

    Pass an instance as ``lprob_func=`` to ``BruteForce`` / ``NearestNeighbors``.
    Calling it for one object with the reference's signature also works
    (``row=`` picks the table row)."""

    def __init__(self, lnprior, rows=None):
        if isinstance(lnprior, np.ndarray) or not hasattr(lnprior, "data_ptr"):
            lnprior = np.atleast_2d(np.ascontiguousarray(lnprior, dtype=np.float64))
        elif lnprior.dim() != 2 or not lnprior.is_contiguous() or lnprior.element_size() != 8:
            raise ValueError("a device ln-prior table must be a contiguous 2-D float64 tensor")
        self.table = lnprior
        self.P, self.M = int(lnprior.shape[0]), int(lnprior.shape[1])
        if rows is not None and (isinstance(rows, np.ndarray) or not hasattr(rows, "data_ptr")):
            rows = np.ascontiguousarray(rows, dtype=np.int64)
        self.rows = rows

    def chunk(self, lo, hi, Ndata):
        """the ``(table, P, rows)`` triple of objects [lo, hi) out of Ndata"""
        if self.rows is not None:
            if len(self.rows) != Ndata:
                raise ValueError("ln-prior `rows` has %d entries for %d objects" % (len(self.rows), Ndata))
            return self.table, self.P, self.rows[lo:hi]
        if self.P == 1:
            return self.table, 1, None
        if self.P != Ndata:
            raise ValueError("ln-prior table has %d rows for %d objects and no `rows`" % (self.P, Ndata))
        return self.table[lo:hi], hi - lo, None

    def __call__(self, data, data_err, data_mask, models, models_err, models_mask, row=0,
                 free_scale=False, ignore_model_err=False, dim_prior=True, ltol=1e-4,
                 return_scale=False, device=None):
        eng = get_engine(device)
        eng.upload_models(models, models_err, models_mask)
        if eng.M != self.M:
            raise ValueError("ln-prior rows hold %d models, the model set %d" % (self.M, eng.M))
        obj = HostObjects(np.atleast_2d(data), np.atleast_2d(data_err), np.atleast_2d(data_mask))
        M = eng.M
        lnp, lnl, lpr, chi2 = (np.empty((1, M)) for _ in range(4))
        ndim = np.empty((1, M), dtype=np.int64)
        sc = se = None
        if free_scale and return_scale:
            sc, se = np.empty((1, M)), np.empty((1, M))
        opts = like_opts(dict(free_scale=free_scale, ignore_model_err=ignore_model_err,
                              dim_prior=dim_prior, ltol=ltol))
        rows = np.array([row], dtype=np.int64)
        eng.fit_prior(obj.x, obj.xe, obj.xm, opts, (self.table, self.P, rows), lnp, lnl, lpr, chi2,
                      ndim, sc, se)
        for dst, buf in ((data, obj.x), (data_err, obj.xe), (data_mask, obj.xm)):
            if isinstance(dst, np.ndarray) and not np.shares_memory(dst, buf):
                dst[...] = buf.reshape(dst.shape)
        out = (lnp[0], lnl[0], lpr[0], ndim[0].astype(_ndim_dtype(data_mask, models_mask)), chi2[0])
        if free_scale and return_scale:
            out = out + (sc[0], se[0])
        return out


def gaussian(mu, std, x):
    """N(x | mu, std) on a grid (pdf.py:414-425) -- host helper used to build the
    dictionary tables."""
    d = (np.asarray(x) - mu) / std
    return np.exp(-0.5 * np.square(d)) / (math.sqrt(2.0 * math.pi) * std)


class PDFDict(object):
    """Grid + dictionary of truncated Gaussian kernels (pdf.py:778-852).  Same
    attributes as the reference: ``Ngrid, min, max, delta, grid, Ndict, sigma_grid,
    dsigma, sigma_width, sigma_dict, sigma_dict_cdf`` and ``fit``."""

    def __init__(self, pdf_grid, sigma_grid, sigma_trunc=5.):
        self.Ngrid = len(pdf_grid)
        self.min, self.max = min(pdf_grid), max(pdf_grid)
        self.delta = pdf_grid[1] - pdf_grid[0]
        self.grid = np.array(pdf_grid)
        self.Ndict = len(sigma_grid)
        self.sigma_grid = np.array(sigma_grid)
        self.dsigma = sigma_grid[1] - sigma_grid[0]
        # half-widths in grid cells; the slice below is taken literally so entries
        # wider than half the grid come out malformed just like pdf.py:814-816
        self.sigma_width = np.ceil(self.sigma_grid * sigma_trunc / self.delta).astype('int')
        mid = int(self.Ngrid / 2)
        self.sigma_dict = []
        self.sigma_dict_cdf = []
        for s, w in zip(self.sigma_grid, self.sigma_width):
            k = gaussian(self.grid[mid], s, self.grid[mid - w:mid + w + 1])
            self.sigma_dict.append(k)
            self.sigma_dict_cdf.append(np.cumsum(k))

    def fit(self, X, Xe):
        """(value, error) -> (grid index, dictionary index) (pdf.py:843-852): mean
        index by round-half-even and NOT clamped; error index clamped."""
        X, Xe = np.asarray(X), np.asarray(Xe)
        X_idx = ((X - self.grid[0]) / self.delta).round().astype('int')
        Xe_idx = np.array(np.round((Xe - self.sigma_grid[0]) / self.dsigma), dtype='int')
        np.clip(Xe_idx, 0, self.Ndict - 1, out=Xe_idx)
        return X_idx, Xe_idx


def gauss_kde(y, y_std, x, dx=None, y_wt=None, sig_thresh=5., wt_thresh=1e-3, cdf_thresh=2e-4,
              device=None, *args, **kwargs):
    """Direct weighted Gaussian KDE on grid ``x`` (pdf.py:444-526), un-normalised."""
    eng = get_engine(device)
    y = np.asarray(y, dtype=float)
    wt = np.ones(len(y)) if y_wt is None else np.ascontiguousarray(y_wt, dtype=np.float64)
    if wt.shape != (len(y),) or len(np.atleast_1d(y_std)) != len(y):
        raise ValueError("`y`, `y_std` and `y_wt` must have the same length")
    eng.upload_labels_grid(y, y_std, x, dx=dx, sig_thresh=sig_thresh)
    out = np.empty((1, len(x)))
    ko = kde_opts(dict(wt_thresh=wt_thresh, cdf_thresh=cdf_thresh), normalize=False)
    eng.predict_logwt(wt.reshape(1, -1), ko, out, is_log=False)
    return out[0]


def gauss_kde_dict(pdfdict, y=None, y_std=None, y_idx=None, y_std_idx=None, y_wt=None,
                   wt_thresh=1e-3, cdf_thresh=2e-4, device=None, *args, **kwargs):
    """Dictionary KDE (pdf.py:529-622), un-normalised."""
    if y_idx is not None and y_std_idx is not None:
        pass
    elif y is not None and y_std is not None:
        y_idx, y_std_idx = pdfdict.fit(y, y_std)
    else:
        raise ValueError("At least one pair of (`y`, `y_std`) or (`y_idx`, `y_idx_std`) must "
                         "be specified.")
    eng = get_engine(device)
    wt = np.ones(len(y_idx)) if y_wt is None else np.ascontiguousarray(y_wt, dtype=np.float64)
    if wt.shape != (len(y_idx),) or len(y_std_idx) != len(y_idx):
        raise ValueError("`y_idx`, `y_std_idx` and `y_wt` must have the same length")
    eng.upload_dict(pdfdict)
    eng.upload_labels_dict(y_idx, y_std_idx)
    out = np.empty((1, pdfdict.Ngrid))
    ko = kde_opts(dict(wt_thresh=wt_thresh, cdf_thresh=cdf_thresh), normalize=False)
    eng.predict_logwt(wt.reshape(1, -1), ko, out, is_log=False)
    return out[0]


def magnitude(phot, err, zeropoints=1., *args, **kwargs):
    """AB magnitudes and errors (pdf.py:625-657); k-NN feature map."""
    phot, err = np.asarray(phot), np.asarray(err)
    return -2.5 * np.log10(phot / zeropoints), 2.5 / math.log(10.) * err / phot


def luptitude(phot, err, skynoise=1., zeropoints=1., *args, **kwargs):
    """asinh magnitudes and errors (pdf.py:695-734); the default k-NN feature map."""
    phot, err = np.asarray(phot), np.asarray(err)
    mag = -2.5 / math.log(10.) * (np.arcsinh(phot / (2. * skynoise)) + np.log(skynoise / zeropoints))
    mag_err = np.sqrt(np.square(2.5 * np.log10(np.e) * err)
                      / (np.square(2. * skynoise) + np.square(phot)))
    return mag, mag_err


def _default_wconf(point):
    return (1. + point) * 0.03


def pdfs_summarize(pdfs, pgrid, renormalize=True, rstate=None, pkern='lorentz', pkern_grid=None,
                   wconf_func=None, device=None):
    """PDF summary statistics (pdf.py:899-1074): mean / median / mode / "best" estimators
    with their std, conf and risk, the 68 % / 95 % credible bounds and a Monte-Carlo
    draw.  Same signature and return tuple as the reference; ``pdfs`` is renormalised
    in place.  The (G,G) loss kernel is a host-side table (pdf.py:1003-1023); the
    (N,G)x(G,G) risk product, the CDFs and every per-object reduction run on the GPU.
    A custom ``wconf_func`` is evaluated on the host (once per estimator and object,
    like the reference) between two device passes."""
    if rstate is None:
        rstate = np.random
    if not isinstance(pdfs, np.ndarray) or pdfs.dtype != np.float64 or not pdfs.flags.c_contiguous:
        raise ValueError("`pdfs` must be a C-contiguous float64 array (it is renormalised in place)")
    pgrid_a = np.ascontiguousarray(pgrid, dtype=np.float64)
    Nobj, Ngrid = len(pdfs), len(pgrid_a)
    if pdfs.shape != (Nobj, Ngrid):
        raise ValueError("`pdfs` must have shape (Npdf, len(pgrid))")
    # kernel over (truth, guess) pairs, exactly as pdf.py:1003-1023
    if pkern_grid is None:
        ptrue = pgrid_a.reshape(Ngrid, 1)
        pguess = pgrid_a.reshape(1, Ngrid)
        pkern_grid = (ptrue - pguess) / ((1. + ptrue) * 0.15)
    if isinstance(pkern, str) and pkern == 'tophat':
        kernel = (np.square(pkern_grid) < 1.)
    elif isinstance(pkern, str) and pkern == 'gaussian':
        kernel = np.exp(-0.5 * np.square(pkern_grid))
    elif isinstance(pkern, str) and pkern == 'lorentz':
        kernel = 1. / (1. + np.square(pkern_grid))
    else:
        try:
            kernel = pkern(pkern_grid)
        except Exception:
            raise RuntimeError("The input kernel does not appear to be valid.")
    loss = np.ascontiguousarray(1.0 - kernel, dtype=np.float64)
    urand = np.array([rstate.rand() for _ in range(Nobj)], dtype=np.float64)      # pdf.py:1000, one draw per object
    eng = get_engine(device)
    stats = np.empty((21, Nobj))
    eng.pdfs_summarize(pdfs, pgrid_a, renormalize, urand, loss, None, 0.03, stats)
    if wconf_func is not None:
        widths = np.array([[wconf_func(stats[4 * e, i]) for e in range(4)] for i in range(Nobj)], dtype=np.float64)
        eng.pdfs_summarize(pdfs, pgrid_a, False, urand, loss, np.ascontiguousarray(widths), 0.03, stats)
    s = stats
    return ((s[0], s[1], s[2], s[3]), (s[4], s[5], s[6], s[7]), (s[8], s[9], s[10], s[11]),
            (s[12], s[13], s[14], s[15]), (s[16], s[17], s[18], s[19]), s[20])


def pdfs_resample(pdfs, old_grid, new_grid, renormalize=True, left=0., right=0., device=None):
    """Resample PDFs onto a new grid (pdf.py:855-896: ``numpy.interp`` per row, then an
    optional renormalisation to unit sum)."""
    og = np.ascontiguousarray(old_grid, dtype=np.float64)
    ng = np.ascontiguousarray(new_grid, dtype=np.float64)
    if isinstance(pdfs, np.ndarray) or not hasattr(pdfs, "data_ptr"):
        pdfs = np.ascontiguousarray(pdfs, dtype=np.float64)
    out = np.empty((len(pdfs), len(ng)))
    get_engine(device).pdfs_resample(pdfs, og, ng, left, right, renormalize, out, n=len(pdfs))
    return out
