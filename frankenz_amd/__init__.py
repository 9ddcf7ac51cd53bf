"""
frankenz_amd -- MI355X (gfx950) engine for frankenz's brute-force photometric
likelihood -> weighted Gaussian-KDE PDF path, behind the reference's own
``BruteForce`` / ``NearestNeighbors`` ``fit() / predict() / fit_predict()`` surface.

All arithmetic of the path runs in hand-written HIP kernels reached through the
C ABI in ``include/frankenz_hip.h`` (``libfrankenz_hip.so``, loaded with ctypes).
There is no CPU fallback: importing the engine without the built library raises.
"""
from .pdf import (PDFDict, gaussian, gauss_kde, gauss_kde_dict, loglike, logprob, logprob_prior,
                  luptitude, magnitude, pdfs_resample, pdfs_summarize)
from .bruteforce import BruteForce
from .knn import NearestNeighbors
from . import fitting, networks, pdf, samplers

__version__ = "0.1.0"
__all__ = ["BruteForce", "NearestNeighbors", "PDFDict", "gaussian", "gauss_kde",
           "gauss_kde_dict", "loglike", "logprob", "logprob_prior", "luptitude", "magnitude",
           "pdfs_resample", "pdfs_summarize", "fitting", "networks", "pdf", "samplers"]
