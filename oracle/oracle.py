"""ctypes loader for the CPU oracle (oracle/oem_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (oem_amd) must never import it.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

PENALTIES = ["elastic.net", "lasso", "ols", "mcp", "scad", "mcp.net", "scad.net",
             "grp.lasso", "grp.lasso.net", "grp.mcp", "grp.scad", "grp.mcp.net",
             "grp.scad.net", "sparse.grp.lasso"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class OrcOpts(C.Structure):
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
        ("ncores", C.c_int32), ("gigs", C.c_double), ("d_override", C.c_double),
    ]


def build(force=False):
    out = _HERE / "_build" / "liboem_oracle.so"
    src = _HERE / "oem_oracle.c"
    if force or not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE)], check=True, stdout=subprocess.DEVNULL)
    return out


_libs = {}


def _host_cpu():
    """model name + ISA flags of this host: the -march=native build is only valid (and only a fair timing) on the CPU it
    was compiled on, and oracle/_build/ travels to the GPU box prebuilt"""
    import hashlib
    try:
        txt = open("/proc/cpuinfo").read()
    except OSError:
        return "unknown"
    keep = sorted({l.split(":", 1)[1].strip() for l in txt.splitlines() if l.startswith(("model name", "flags"))})
    return hashlib.sha1("|".join(keep).encode()).hexdigest()


def build_native():
    """(re)build liboem_oracle_native.so when it was compiled on another CPU (build it on the box it is timed on)"""
    out = _HERE / "_build" / "liboem_oracle_native.so"
    stamp = _HERE / "_build" / "native.cpu"
    src = _HERE / "oem_oracle.c"
    cpu = _host_cpu()
    if out.exists() and stamp.exists() and stamp.read_text() == cpu and out.stat().st_mtime >= src.stat().st_mtime:
        return out
    (_HERE / "_build").mkdir(exist_ok=True)
    subprocess.run(["gcc", "-O2", "-fPIC", "-std=gnu99", "-fopenmp", "-O3", "-march=native", "-ffp-contract=off", "-shared",
                    "-o", str(out), str(src), "-lm"], check=True)
    stamp.write_text(cpu)
    return out


def lib(native=False):
    key = bool(native)
    if key not in _libs:
        build()
        if native:
            build_native()
        name = "liboem_oracle_native.so" if native else "liboem_oracle.so"
        L = C.CDLL(str(_HERE / "_build" / name))
        L.orc_last_error.restype = C.c_char_p
        L.orc_eig_max.restype = C.c_double
        _libs[key] = L
    return _libs[key]


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, t=_dp):
    return a.ctypes.data_as(t) if a is not None and a.size > 0 else t()


class _Opts:
    """Keeps the numpy buffers alive next to the ctypes struct."""

    def __init__(self, p_pf, penalty, lambda_=None, nlambda=100, lambda_min_ratio=1e-4, alpha=1.0, gamma=3.0,
                 tau=0.5, tol=1e-7, maxit=500, accelerate=False, compute_loss=False, penalty_factor=None,
                 groups=None, unique_groups=None, group_weights=None, ncores=1, gigs=4.0, d_override=0.0):
        if isinstance(penalty, str):
            penalty = [penalty]
        self.names = list(penalty)
        self.pen = np.array([PENALTIES.index(q) for q in penalty], dtype=np.int32)
        self.pf = _d(np.ones(p_pf) if penalty_factor is None else penalty_factor)
        self.lam = None
        nlu = 0
        if lambda_ is not None and len(lambda_) > 0:
            if isinstance(lambda_, (list, tuple)) and np.ndim(lambda_[0]) > 0:
                lam = np.stack([np.sort(_d(l))[::-1] for l in lambda_])
            else:
                lam = np.tile(np.sort(_d(lambda_))[::-1], (len(penalty), 1))
            self.lam = _d(lam)
            nlu = self.lam.shape[1]
        self.groups = None if groups is None or len(groups) == 0 else np.ascontiguousarray(groups, dtype=np.int32)
        self.ug = None if unique_groups is None or len(unique_groups) == 0 else np.ascontiguousarray(unique_groups, dtype=np.int32)
        self.gw = None if group_weights is None or len(group_weights) == 0 else _d(group_weights)
        o = OrcOpts()
        o.npen = len(self.pen); o.penalty = _ptr(self.pen, _ip)
        o.nlambda = int(nlambda); o.lambda_min_ratio = float(lambda_min_ratio)
        o.lambda_user = _ptr(self.lam); o.nlambda_user = nlu
        o.alpha, o.gamma, o.tau, o.tol = float(alpha), float(gamma), float(tau), float(tol)
        o.maxit, o.accelerate, o.compute_loss = int(maxit), int(bool(accelerate)), int(bool(compute_loss))
        o.penalty_factor = _ptr(self.pf)
        o.groups = _ptr(self.groups, _ip); o.ngroupvars = 0 if self.groups is None else len(self.groups)
        o.unique_groups = _ptr(self.ug, _ip); o.ngroups = 0 if self.ug is None else len(self.ug)
        o.group_weights = _ptr(self.gw); o.n_group_weights = 0 if self.gw is None else len(self.gw)
        o.ncores = int(ncores); o.gigs = float(gigs); o.d_override = float(d_override)
        self.c = o
        self.nl = nlu if nlu > 0 else int(nlambda)


def _result(o, beta, lam, niter, loss, d, rows):
    npen, nl = len(o.names), o.nl
    out = {"beta": [], "lambda": [], "niter": [], "loss": [], "d": d.value, "penalty": o.names}
    for k, name in enumerate(o.names):
        b = beta[k].reshape(nl, rows).T.copy()       # rows x nl
        if name == "ols":
            out["beta"].append(b[:, :1]); out["niter"].append(niter[k, :1].copy()); out["loss"].append(loss[k, :1].copy())
        else:
            out["beta"].append(b); out["niter"].append(niter[k].copy()); out["loss"].append(loss[k].copy())
        out["lambda"].append(lam[k].copy())
    return out


def _call(fn, o, rows, *args, native=False):
    npen, nl = len(o.names), o.nl
    beta = np.zeros((npen, nl * rows)); lam = np.zeros((npen, nl))
    niter = np.zeros((npen, nl), dtype=np.int32); loss = np.zeros((npen, nl)); d = C.c_double(0)
    rc = fn(*args, C.byref(o.c), _ptr(beta), _ptr(lam), _ptr(niter, _ip), _ptr(loss), C.byref(d))
    if rc != 0:
        raise RuntimeError(lib(native).orc_last_error().decode())
    return _result(o, beta, lam, niter, loss, d, rows)


def fit_dense(x, y, penalty="elastic.net", standardize=True, intercept=True, native=False, **kw):
    x = np.asfortranarray(x, dtype=np.float64); y = _d(y)
    n, p = x.shape
    o = _Opts(p, penalty, **kw)
    L = lib(native)
    return _call(L.orc_fit_dense, o, p + 1, _ptr(x), C.c_int64(n), C.c_int32(p), _ptr(y),
                 C.c_int32(int(standardize)), C.c_int32(int(intercept)), native=native)


def fit_dense_w(x, y, weights, penalty="elastic.net", standardize=True, intercept=True, native=False, **kw):
    """oemDense with observation weights as the C++ entry computes it (unreachable from R/oem.R:244; parity unpinned)."""
    x = np.asfortranarray(x, dtype=np.float64); y = _d(y); w = _d(weights)
    n, p = x.shape
    o = _Opts(p, penalty, **kw)
    L = lib(native)
    return _call(L.orc_fit_dense_w, o, p + 1, _ptr(x), C.c_int64(n), C.c_int32(p), _ptr(y), _ptr(w),
                 C.c_int32(int(standardize)), C.c_int32(int(intercept)), native=native)


def fit_xtx(xtx, xty, penalty="elastic.net", scale_factor=None, native=False, **kw):
    xtx = np.asfortranarray(xtx, dtype=np.float64); xty = _d(xty)
    p = xtx.shape[0]
    o = _Opts(p, penalty, **kw)
    sf = None if scale_factor is None or len(scale_factor) == 0 else _d(scale_factor)
    L = lib(native)
    return _call(L.orc_fit_xtx, o, p, _ptr(xtx), _ptr(xty), C.c_int32(p), _ptr(sf), native=native)


def fit_big(x, y, penalty="elastic.net", standardize=True, intercept=True, native=False, **kw):
    x = np.asfortranarray(x, dtype=np.float64); y = _d(y)
    n, p = x.shape
    o = _Opts(p, penalty, **kw)
    L = lib(native)
    return _call(L.orc_fit_big, o, p + 1, _ptr(x), C.c_int64(n), C.c_int32(p), _ptr(y),
                 C.c_int32(int(standardize)), C.c_int32(int(intercept)), native=native)


def xval_dense(x, y, foldid, penalty="elastic.net", standardize=True, intercept=True, type_measure="mse", native=False,
               weights=None, **kw):
    """ref src/oem_xval_dense.cpp:31-482.  foldid: values 1..nfolds.  Adds cvm, cvsd (lists per penalty) to the fit.
    weights: observation weights (ref src/oem_xval_dense.h:486-623) or None."""
    x = np.asfortranarray(x, dtype=np.float64); y = _d(y)
    w = None if weights is None or len(weights) == 0 else _d(weights)
    n, p = x.shape
    fid = np.ascontiguousarray(foldid, dtype=np.int32)
    o = _Opts(p, penalty, **kw)
    L = lib(native)
    npen, nl = len(o.names), o.nl
    cvm = np.zeros((npen, nl)); cvsd = np.zeros((npen, nl))
    beta = np.zeros((npen, nl * (p + 1))); lam = np.zeros((npen, nl))
    niter = np.zeros((npen, nl), dtype=np.int32); loss = np.zeros((npen, nl)); d = C.c_double(0)
    rc = L.orc_xval_dense_w(_ptr(x), C.c_int64(n), C.c_int32(p), _ptr(y), _ptr(w), _ptr(fid, _ip), C.c_int32(int(fid.max())),
                          C.c_int32(int(standardize)), C.c_int32(int(intercept)), C.c_int32({"mse": 0, "mae": 1}[type_measure]),
                          C.byref(o.c), _ptr(beta), _ptr(lam), _ptr(niter, _ip), _ptr(loss), C.byref(d), _ptr(cvm), _ptr(cvsd))
    if rc != 0:
        raise RuntimeError(L.orc_last_error().decode())
    out = _result(o, beta, lam, niter, loss, d, p + 1)
    out["cvm"] = [cvm[k, :1].copy() if name == "ols" else cvm[k].copy() for k, name in enumerate(o.names)]
    out["cvsd"] = [cvsd[k, :1].copy() if name == "ols" else cvsd[k].copy() for k, name in enumerate(o.names)]
    return out


def r_sparse_groups(groups, intercept):
    """What R's oem() hands to oem_fit_sparse for a group penalty (ref R/oem.R:285-338, default group weights): with an
    intercept the unpenalised group 0 is added to unique.groups and a 0 is prepended to groups (slot 0 = the intercept)."""
    groups = np.asarray(groups).ravel()
    ug = np.sort(np.unique(groups))
    if intercept:
        if not np.any(ug == 0):
            ug = np.sort(np.concatenate([[0], ug]))
        groups = np.concatenate([[0], groups])
    return groups.astype(np.int32), ug.astype(np.int32)


def fit_sparse(x, y, penalty="elastic.net", standardize=True, intercept=True, native=False, **kw):
    """ref src/oem_sparse.cpp:30-267.  x: a scipy.sparse matrix (converted to CSC with sorted indices).
    groups / unique_groups are the arrays the C++ entry receives (see r_sparse_groups), p + intercept entries."""
    import scipy.sparse as sp
    xc = sp.csc_matrix(x, dtype=np.float64); xc.sort_indices()
    n, p = xc.shape
    colptr = np.ascontiguousarray(xc.indptr, dtype=np.int64); rowidx = np.ascontiguousarray(xc.indices, dtype=np.int32)
    val = np.ascontiguousarray(xc.data, dtype=np.float64); y = _d(y)
    o = _Opts(p, penalty, **kw)
    L = lib(native)
    return _call(L.orc_fit_sparse, o, p + 1, C.c_int64(n), C.c_int32(p), colptr.ctypes.data_as(C.POINTER(C.c_int64)),
                 _ptr(rowidx, _ip), _ptr(val), _ptr(y), C.c_int32(int(standardize)), C.c_int32(int(intercept)), native=native)


def standardize(x, y, standardize=True, intercept=True):
    x = np.array(x, dtype=np.float64, order="F", copy=True); y = np.array(y, dtype=np.float64, copy=True)
    n, p = x.shape
    mx = np.zeros(p); sx = np.zeros(p); my = C.c_double(0); sy = C.c_double(0)
    lib().orc_standardize(_ptr(x), C.c_int64(n), C.c_int32(p), _ptr(y), C.c_int32(int(standardize)),
                          C.c_int32(int(intercept)), _ptr(mx), _ptr(sx), C.byref(my), C.byref(sy))
    return x, y, mx, sx, my.value, sy.value


def gram(x, y, ncores=1, native=False):
    x = np.asfortranarray(x, dtype=np.float64); y = _d(y)
    n, p = x.shape
    xx = np.zeros((p, p), order="F"); xy = np.zeros(p)
    lib(native).orc_gram(_ptr(x), C.c_int64(n), C.c_int32(p), _ptr(y), C.c_int32(ncores), _ptr(xx), _ptr(xy))
    return xx, xy


def eig_max(a):
    a = np.asfortranarray(a, dtype=np.float64)
    return lib().orc_eig_max(_ptr(a), C.c_int32(a.shape[0]))


def path(xx, xy, d, lambda_scaled, penalty="lasso", scale_factor_inv=None, **kw):
    xx = np.asfortranarray(xx, dtype=np.float64); xy = _d(xy)
    p = xx.shape[0]
    o = _Opts(p, penalty, **kw)
    lam = _d(np.atleast_2d(lambda_scaled))
    if lam.shape[0] != len(o.names):
        lam = _d(np.tile(lam, (len(o.names), 1)))
    nl = lam.shape[1]
    beta = np.zeros((len(o.names), nl, p)); niter = np.zeros((len(o.names), nl), dtype=np.int32)
    sfi = None if scale_factor_inv is None else _d(scale_factor_inv)
    rc = lib().orc_path(_ptr(xx), _ptr(xy), C.c_int32(p), C.c_double(d), C.byref(o.c), _ptr(lam), C.c_int32(nl),
                        _ptr(sfi), _ptr(beta), _ptr(niter, _ip))
    if rc != 0:
        raise RuntimeError(lib().orc_last_error().decode())
    return beta, niter


def stop_rule(cur, prev, tol):
    cur = _d(cur); prev = _d(prev)
    return bool(lib().orc_stop_rule(_ptr(cur), _ptr(prev), C.c_int32(len(cur)), C.c_double(tol)))


def threshold(penalty, u, lam, d, alpha=1.0, gamma=3.0, tau=0.5, penalty_factor=None, groups=None,
              unique_groups=None, group_weights=None):
    u = _d(u); p = len(u)
    pf = _d(np.ones(p) if penalty_factor is None else penalty_factor)
    g = None if groups is None else np.ascontiguousarray(groups, dtype=np.int32)
    ug = None if unique_groups is None else np.ascontiguousarray(unique_groups, dtype=np.int32)
    gw = None if group_weights is None else _d(group_weights)
    out = np.zeros(p)
    lib().orc_threshold(C.c_int32(PENALTIES.index(penalty)), _ptr(u), C.c_int32(p), C.c_double(lam), C.c_double(d),
                        C.c_double(alpha), C.c_double(gamma), C.c_double(tau), _ptr(pf), _ptr(g, _ip), _ptr(ug, _ip),
                        C.c_int32(0 if ug is None else len(ug)), _ptr(gw), _ptr(out))
    return out
