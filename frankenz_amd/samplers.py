"""
``frankenz.samplers`` pieces that consume the PDF stack on the GPU (reference
frankenz/samplers.py:23-86): the population log-likelihood ``loglike_nz``, and the
redshift-assignment step of the hierarchical / population Gibbs samplers ``nz_assign``
(samplers.py:498-499, 519-520).  The MCMC drivers around them (random-state dependent,
sequential: Dirichlet / reference-sample draws, bookkeeping) stay with the caller.
"""
import numpy as np

from .engine import get_engine

__all__ = ["loglike_nz", "nz_assign"]


def loglike_nz(nz, pdfs, overlap=None, return_overlap=False, pair=None, pair_step=None, device=None):
    """ln-likelihood of a population n(z) given individual PDFs (samplers.py:23-86):
    ``sum(log(pdfs @ nz + pair_step * (pdfs[:, i] - pdfs[:, j])))``; ``-inf`` (and zero
    overlaps) when ``nz`` has a negative or non-finite entry.  ``pdfs`` may be a device
    tensor; a precomputed ``overlap`` is used as is, like the reference."""
    nz = np.ascontiguousarray(nz, dtype=np.float64)
    n = len(pdfs)
    if np.any(~np.isfinite(nz) | (nz < 0.)):
        lnlike, out = -np.inf, np.zeros(n)
        return (lnlike, out) if return_overlap else lnlike
    if overlap is not None:
        perturb = 0.
        if pair is not None and pair_step is not None:
            i, j = pair
            perturb = pair_step * (np.asarray(pdfs)[:, i] - np.asarray(pdfs)[:, j])
        out = overlap + perturb
        lnlike = np.sum(np.log(out))
        return (lnlike, out) if return_overlap else lnlike
    if isinstance(pdfs, np.ndarray):
        pdfs = np.ascontiguousarray(pdfs, dtype=np.float64)
    out = np.empty(n)
    use_pair = pair is not None and pair_step is not None
    lnlike = get_engine(device).overlap_nz(pdfs, nz, pair if use_pair else None, pair_step if use_pair else 0.0, out, n=n)
    return (lnlike, out) if return_overlap else lnlike


def nz_assign(nz, pdfs, u=None, rstate=None, return_bins=False, device=None):
    """One categorical draw per object from ``pdfs[i] * nz / dot(pdfs[i], nz)`` and the number of objects per
    bin -- ``np.sum([rstate.multinomial(1, p * pos / np.dot(p, pos)) for p in pdfs], axis=0)`` of
    samplers.py:498-499 / 519-520, the N x G inner step of every Gibbs sweep.  The draw is the inverse CDF of
    one uniform per object (``u``, or ``rstate.rand(N)``): the same distribution as the reference's
    ``multinomial(1, ...)``, which consumes its random stream differently (one binomial per bin), so the
    realisation for a given seed differs.  ``pdfs`` may be a device tensor.  Returns ``counts`` (G, int64)
    [and ``bins`` (N, int64; -1 for a row without mass)]."""
    nz = np.ascontiguousarray(nz, dtype=np.float64)
    n = len(pdfs)
    if u is None:
        u = (np.random if rstate is None else rstate).rand(n)
    u = np.ascontiguousarray(u, dtype=np.float64)
    if u.shape != (n,) or np.any(~(u >= 0.) | ~(u < 1.)):
        raise ValueError("`u` must hold one uniform in [0, 1) per object")
    if isinstance(pdfs, np.ndarray):
        pdfs = np.ascontiguousarray(pdfs, dtype=np.float64)
        if pdfs.shape != (n, len(nz)):
            raise ValueError("`pdfs` must have shape (Nobj, len(nz))")
    counts = np.zeros(len(nz), dtype=np.int64)
    bins = np.empty(n, dtype=np.int64)
    get_engine(device).nz_assign(pdfs, nz, u, bins, counts, n=n)
    return (counts, bins) if return_bins else counts
