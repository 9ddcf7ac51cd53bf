"""
``frankenz.samplers`` pieces that consume the PDF stack on the GPU (reference
frankenz/samplers.py:23-86): the population log-likelihood ``loglike_nz``.  The MCMC
drivers around it (random-state dependent, sequential) stay with the caller.
"""
import numpy as np

from .engine import get_engine

__all__ = ["loglike_nz"]


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
