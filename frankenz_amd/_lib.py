"""ctypes binding of include/frankenz_hip.h.  Fails loudly if the HIP library is
missing or does not export the whole ABI -- there is no fallback path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FRANKENZ_HIP_LIB",
                          os.path.join(_HERE, "csrc", "libfrankenz_hip.so"))


def source_id():
    """short hash of the kernel sources the shipped library is built from (csrc/*.hip, *.h, *.inc and the ABI header): the stamp
    under which profiles/pmc_latest.json keeps counter values, so that bench.py reports them only for the build they were taken on"""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".inc")))
    for f in files + [os.path.join("..", "..", "include", "frankenz_hip.h")]:
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


class LikeOpts(C.Structure):
    _fields_ = [("free_scale", C.c_int32), ("ignore_model_err", C.c_int32),
                ("dim_prior", C.c_int32), ("max_iter", C.c_int32),
                ("ltol", C.c_double), ("exact_evidence", C.c_int32), ("reserved_", C.c_int32)]


class KdeOpts(C.Structure):
    _fields_ = [("wt_thresh", C.c_double), ("use_wt_thresh", C.c_int32),
                ("normalize", C.c_int32), ("cdf_thresh", C.c_double),
                ("exact_evidence", C.c_int32), ("reserved_", C.c_int32)]


class Prior(C.Structure):
    _fields_ = [("table", C.c_void_p), ("P", C.c_int64), ("rows", C.c_void_p)]


class Timing(C.Structure):
    _fields_ = [("ms_planes", C.c_double), ("n_planes", C.c_int64),
                ("ms_fused", C.c_double), ("n_fused", C.c_int64),
                ("ms_stats", C.c_double), ("n_stats", C.c_int64),
                ("ms_kde", C.c_double), ("n_kde", C.c_int64),
                ("ms_modec", C.c_double), ("n_modec", C.c_int64),
                ("ms_knn", C.c_double), ("n_knn", C.c_int64),
                ("ms_other", C.c_double), ("n_other", C.c_int64)]


_P = C.c_void_p
_I64, _I32, _F64 = C.c_int64, C.c_int32, C.c_double

# name -> (restype, argtypes): exactly the declarations of include/frankenz_hip.h
ABI = {
    "fz_last_error": (C.c_char_p, []),
    "fz_device_count": (C.c_int, []),
    "fz_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "fz_ctx_destroy": (None, [_P]),
    "fz_sync": (C.c_int, [_P]),
    "fz_debug_opts": (C.c_int, [C.c_char_p]),
    "fz_timing_reset": (C.c_int, [_P]),
    "fz_timing_get": (C.c_int, [_P, C.POINTER(Timing)]),
    "fz_last_form": (C.c_char_p, [_P]),
    "fz_set_workspace_limit": (C.c_int, [_P, _I64]),
    "fz_set_producer_stream": (C.c_int, [_P, _P, _I32]),
    "fz_host_alloc": (C.c_int, [_I64, C.POINTER(_P)]),
    "fz_host_free": (C.c_int, [_P]),
    "fz_modec_info": (C.c_int, [_P, _P]),
    "fz_modec_niter": (C.c_int, [_P, _I64, _P]),
    "fz_models_upload": (C.c_int, [_P, _P, _P, _P, _I64, _I32]),
    "fz_kdedict_upload": (C.c_int, [_P, _I64, _I64, _P, _P, _P, _P]),
    "fz_labels_upload_dict": (C.c_int, [_P, _P, _P, _I64]),
    "fz_labels_upload_grid": (C.c_int, [_P, _P, _P, _I64, _P, _I64, _F64, _F64]),
    "fz_clean": (C.c_int, [_P, _P, _P, _P, _I64, _I32]),
    "fz_fit": (C.c_int, [_P, _P, _P, _P, _I64, C.POINTER(LikeOpts), _P, _P, _P, _P, _P]),
    "fz_fit_predict": (C.c_int, [_P, _P, _P, _P, _I64, C.POINTER(LikeOpts),
                                 C.POINTER(KdeOpts), _P, _P, _P]),
    "fz_fit_prior": (C.c_int, [_P, _P, _P, _P, _I64, C.POINTER(LikeOpts), C.POINTER(Prior)] + [_P] * 7),
    "fz_fit_predict_prior": (C.c_int, [_P, _P, _P, _P, _I64, C.POINTER(LikeOpts),
                                       C.POINTER(KdeOpts), C.POINTER(Prior), _P, _P, _P]),
    "fz_predict_logwt": (C.c_int, [_P, _P, _I64, _I32, C.POINTER(KdeOpts), _P, _P, _P]),
    "fz_knn_upload_trees": (C.c_int, [_P, _P, _I32, _I64, _I32]),
    "fz_knn_query": (C.c_int, [_P, _P, _I64, _I32, _F64, _F64, _P]),
    "fz_selftest_math": (C.c_int, [_P, _I32, _P, _I64, _P]),
    "fz_knn_predict_logwt": (C.c_int, [_P, _P, _P, _P, _I64, _I64, C.POINTER(KdeOpts), _P, _P, _P]),
    "fz_knn_fit_predict": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I64, C.POINTER(LikeOpts),
                                     C.POINTER(KdeOpts)] + [_P] * 10),
    "fz_pdfs_summarize": (C.c_int, [_P, _P, _I64, _I64, _P, _I32, _P, _P, _P, _F64, _P]),
    "fz_pdfs_resample": (C.c_int, [_P, _P, _I64, _I64, _P, _I64, _P, _F64, _F64, _I32, _P]),
    "fz_overlap_nz": (C.c_int, [_P, _P, _I64, _I64, _P, _I64, _I64, _F64, _P, _P]),
    "fz_nz_assign": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P]),
    "fz_net_select": (C.c_int, [_P, _P, _I64, _I32, _I32, _F64, _F64, _P, _P, _I64, _P, _P, _P, _P, _P]),
    "fz_net_table": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P, _P, _I64, _I64, _P]),
    "fz_net_gather": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, C.c_uint64, _P]),
    "fz_net_stack": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _I64, _I64, _P, _P, _P]),
    "fz_knn_search_fit_predict_prior": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _F64, _F64, C.POINTER(LikeOpts),
                                                  C.POINTER(KdeOpts), C.POINTER(Prior)] + [_P] * 12),
    "fz_knn_fit_predict_prior": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I64, C.POINTER(LikeOpts),
                                           C.POINTER(KdeOpts), C.POINTER(Prior)] + [_P] * 12),
}

_lib = None


def load():
    """Load libfrankenz_hip.so and bind every ABI symbol (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "frankenz_amd: HIP library not found at %s -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = _Switched(lib)
    return _lib


class _Switched(object):
    """The loaded library.  Its test / tuning switches ("FZ_..." names) are not read from the
    environment by the C side: this layer hands the process's ``FZ_*`` variables over through the ONE entry point
    ``fz_debug_opts`` whenever they have changed since the last call, so that ``FZ_HIST=0 python bench.py`` and a
    test's ``monkeypatch.setenv`` keep working."""
    _SKIP = ("fz_last_error", "fz_debug_opts", "fz_host_alloc", "fz_host_free")

    def __init__(self, lib):
        object.__setattr__(self, "_raw", lib)
        object.__setattr__(self, "_sent", None)
        object.__setattr__(self, "_fn", {})

    def _sync(self):
        spec = ";".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("FZ_"))
        if spec != self._sent:
            self._raw.fz_debug_opts(spec.encode())
            object.__setattr__(self, "_sent", spec)

    def __getattr__(self, name):
        fn = self._fn.get(name)
        if fn is None:
            raw = getattr(self._raw, name)
            if name in self._SKIP:
                fn = raw
            else:
                def fn(*a, _raw=raw, _sync=self._sync):
                    _sync()
                    return _raw(*a)
            self._fn[name] = fn
        return fn


# error code -> Python exception, mirroring what the reference raises
_EXC = {-2: MemoryError, -3: IndexError, -4: ValueError, -5: NotImplementedError,
        -6: NotImplementedError, -7: RuntimeError}


def check(rc):
    if rc != 0:
        msg = load().fz_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, RuntimeError)(msg)


def ptr(a):
    """void* of a NumPy array, a torch tensor (host or device) or a raw int."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return int(a)
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError("cannot take a pointer of %r" % type(a))
