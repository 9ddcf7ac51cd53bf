import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SDSS_SIGMA = np.array([0.873, 0.348, 0.418, 0.873, 3.476])
# ln-evidence.  Two tolerances, by what the path under test computes:
#   EVID   -- k_fused's weight-space body (many dictionary kernel widths, the direct grid KDE, FZ_HIST=0) and k_plane_fused<X32>
#             (stored rows too long for k_plane_rows): the share of every model within wt_thresh of the best is summed in fp64, the
#             rest (each below wt_thresh of the best) in fp32.  ~1e-9 observed, bounded by ~1e-6 relative on that remainder;
#             north_star's bar is 1e-5.  Tests that cover several routes at once also hold this one.
#             (Since round 4 the default fused kernel, k_hist, is fp64 throughout: test_default_fused_evidence_is_the_fp64_logsumexp
#             holds it to 1e-12.)
#   EVID64 -- every all-fp64 path: exact_evidence=True (lprob_kwargs / kde_kwargs), masked objects or models, padded band
#             counts, wild values (IEEE variant), ln-priors, the Gaussian likelihood, the CDF rule, mode C's materialised rows,
#             the two-pass kernels.  A regression of 1e-8 in any of these fails.
# ln-max and PDFs are fp64 everywhere and are held to their own (tighter) tolerances in every test.
EVID = dict(rtol=1e-7, atol=1e-7)
EVID64 = dict(rtol=1e-9, atol=1e-9)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


class DevArray(object):
    """A float64 / int64 array in device memory, allocated through the SAME HIP runtime the
    library is linked against (ctypes on libamdhip64), exposing the few tensor-like members
    the host layer looks at.  Lets the -m gpu tests pass device pointers across the C ABI
    without bringing up a second GPU runtime stack in the test process."""
    _hip = None

    def __init__(self, a=None, _ptr=None, _shape=None, _base=None):
        import ctypes as C
        if DevArray._hip is None:
            DevArray._hip = C.CDLL('libamdhip64.so.7')
        if a is None:
            self._ptr, self.shape, self._base = _ptr, _shape, _base
            return
        a = np.ascontiguousarray(a)
        assert a.dtype.itemsize == 8
        p = C.c_void_p()
        assert DevArray._hip.hipMalloc(C.byref(p), C.c_size_t(max(a.nbytes, 8))) == 0
        assert DevArray._hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0
        self._ptr, self.shape, self._base = p.value, a.shape, None

    def data_ptr(self):
        return self._ptr

    def dim(self):
        return len(self.shape)

    def is_contiguous(self):
        return True

    def element_size(self):
        return 8

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, sl):                      # contiguous row range -> view
        lo, hi, st = sl.indices(self.shape[0])
        assert st == 1
        row = 8 * int(np.prod(self.shape[1:], dtype=np.int64))
        return DevArray(_ptr=self._ptr + lo * row, _shape=(hi - lo,) + tuple(self.shape[1:]), _base=self)

    def __del__(self):
        if self._base is None and self._ptr and DevArray._hip is not None:
            import ctypes as C
            DevArray._hip.hipFree(C.c_void_p(self._ptr))


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
