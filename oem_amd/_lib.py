"""ctypes binding of liboemgpu.so (include/oemgpu.h).  No torch types cross this boundary."""
import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("OEMGPU_LIB", _HERE / "liboemgpu.so"))   # OEMGPU_LIB: diagnostic builds only

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

PENALTIES = ["elastic.net", "lasso", "ols", "mcp", "scad", "mcp.net", "scad.net",
             "grp.lasso", "grp.lasso.net", "grp.mcp", "grp.scad", "grp.mcp.net",
             "grp.scad.net", "sparse.grp.lasso"]          # R/oem.R:165-173; index = C penalty code

OEMGPU_SEM_DENSE, OEMGPU_SEM_BIG = 0, 1
ERR_INTERRUPTED = -6
NHOSTSTATS = 9
NTIMERS = 8
T_SHIFT, T_MOMENTS, T_FINAL, T_EIGPATH, T_GRAMK = 0, 1, 2, 3, 4


class OemgpuOpts(C.Structure):
    _fields_ = [
        ("npen", C.c_int32), ("penalty", _ip),
        ("nlambda", C.c_int32), ("lambda_min_ratio", C.c_double),
        ("lambda_user", _dp), ("nlambda_user", C.c_int32),
        ("alpha", C.c_double), ("gamma", C.c_double), ("tau", C.c_double), ("tol", C.c_double),
        ("maxit", C.c_int32), ("accelerate", C.c_int32), ("compute_loss", C.c_int32),
        ("penalty_factor", _dp),
        ("groups", _ip), ("ngroupvars", C.c_int32),
        ("unique_groups", _ip), ("ngroups", C.c_int32),
        ("group_weights", _dp), ("n_group_weights", C.c_int32),
        ("device", C.c_int32),
        ("ngpus", C.c_int32), ("devices", _ip), ("upload_threads", C.c_int32),
        ("interrupt", C.CFUNCTYPE(C.c_int, C.c_void_p)), ("interrupt_arg", C.c_void_p),
    ]


class OemgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"liboemgpu error {code}: {msg}")
        self.code = code


_lib = None

_OUT = [_dp, _dp, _ip, _dp, _dp]           # beta, lambda_out, niter, loss, d
_SIGS = {
    "oemgpu_fit_dense": (C.c_int, [_dp, C.c_int64, C.c_int32, _dp, C.c_int32, C.c_int32, C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_fit_dense_weighted": (C.c_int, [_dp, C.c_int64, C.c_int32, _dp, _dp, C.c_int32, C.c_int32, C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_fit_dense_weighted_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                                C.c_int32, C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_fit_xtx": (C.c_int, [_dp, _dp, C.c_int32, _dp, C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_fit_big": (C.c_int, [C.POINTER(_dp), C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.POINTER(_dp), C.c_int32,
                                 C.c_int32, C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_create": (C.c_void_p, [C.c_int32, C.c_void_p]),
    "oemgpu_destroy": (None, [C.c_void_p]),
    "oemgpu_synchronize": (C.c_int, [C.c_void_p]),
    "oemgpu_shift_sums_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "oemgpu_moments_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "oemgpu_sum_in_order_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "oemgpu_solve_moments_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_fit_dense_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_int32, C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_fit_xtx_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, _dp, C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_last_shift_in_effect": (C.c_int, [C.c_void_p]),
    "oemgpu_last_shift_advised": (C.c_int, [C.c_void_p]),
    "oemgpu_last_eigen_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "oemgpu_last_path_engine": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "oemgpu_last_placement": (C.c_int, [C.c_void_p]),
    "oemgpu_fit_sparse": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_xval_dense": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.POINTER(OemgpuOpts)] + _OUT + [_dp, _dp]),
    "oemgpu_xval_dense_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, C.POINTER(OemgpuOpts)] + _OUT + [_dp, _dp]),
    "oemgpu_xval_moments_len": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "oemgpu_xval_fold_moments_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int32, C.POINTER(OemgpuOpts), C.c_void_p, C.POINTER(C.c_int64)]),
    "oemgpu_xval_solve_folds_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_int32, C.POINTER(OemgpuOpts)] + _OUT),
    "oemgpu_xval_cv_triples_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(OemgpuOpts), _dp]),
    "oemgpu_xval_merge": (C.c_int, [_dp, C.c_int32, C.POINTER(OemgpuOpts), _dp, _dp]),
    "oemgpu_eig_max_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, _dp]),
    "oemgpu_last_timings": (C.c_int, [C.c_void_p, _dp]),
    "oemgpu_set_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "oemgpu_row_split": (None, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "oemgpu_last_host_stats": (C.c_int, [_dp]),
    "oemgpu_release_cache": (None, []),
    "oemgpu_selftest_hold_cus": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "oemgpu_selftest_group_permutation": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "oemgpu_selftest_symcoop_owners": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "oemgpu_selftest_coop_slots": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "oemgpu_selftest_sympk_gemv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "oemgpu_selftest_wcoop_sizing": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "oemgpu_selftest_gram_plan": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "oemgpu_selftest_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                             C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "oemgpu_reload_switches": (None, []),
    "oemgpu_switch_names": (C.c_char_p, []),
    "oemgpu_last_error": (C.c_char_p, []),
    "oemgpu_version": (C.c_char_p, []),
    "oemgpu_device_count": (C.c_int, []),
}
EXPORTS = sorted(_SIGS)


def lib():
    """Loads liboemgpu.so.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64, and device pointers / streams
        # are only interchangeable with torch if liboemgpu binds to that same copy.  Loading torch first makes
        # the dynamic loader resolve our DT_NEEDED libamdhip64.so.7 to the already-loaded one.  Without torch
        # (e.g. under R) the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not LIB_PATH.exists():
            raise OSError(f"{LIB_PATH} is missing: build it with `python -m oem_amd.build` "
                          "(oem_amd has no CPU fallback)")
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_env_seen = None


def sync_switches():
    """Called by the Python mirrors of the R front ends (oem, oem_xtx, big_oem, xval_oem) at every call: if an OEM_* / OEMGPU_*
    environment variable has changed since the last look, the library parses its switch table again.  So a script that flips a
    switch between two calls gets what it asks for, as when the library read the environment at every call (rounds 1-4) -- while
    the C library itself never calls getenv on the path of a call (callers of the C ABI use oemgpu_reload_switches())."""
    global _env_seen
    cur = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith(("OEM_", "OEMGPU_"))))
    if cur != _env_seen:
        _env_seen = cur
        reload_switches()


def reload_switches():
    """The OEM_* / OEMGPU_* environment switches are parsed once, at the first call into the library (include/oemgpu.h); a test that
    changes one calls this (tests/conftest.py does it behind every monkeypatch.setenv / delenv).  A no-op while the library is not loaded."""
    if _lib is not None:
        _lib.oemgpu_reload_switches()


def check(rc):
    if rc != 0:
        raise OemgpuError(rc, lib().oemgpu_last_error().decode())


def host_stats():
    """oemgpu_last_host_stats of this thread, as a dict"""
    out = (C.c_double * NHOSTSTATS)()
    check(lib().oemgpu_last_host_stats(out))
    keys = ["call_ms", "upload_moments_ms", "solve_ms", "bytes_staged", "devices", "row_blocks", "resident", "allocations", "host_staged_handovers"]
    return dict(zip(keys, list(out)))


def sums_len(p):
    """include/oemgpu.h oemgpu_sums_len: sample sums, count, sample sums of squares, one reserved slot"""
    return 2 * (p + 1) + 2


def moments_len(p):
    """include/oemgpu.h oemgpu_moments_len"""
    return (p + 2) * (p + 2)
