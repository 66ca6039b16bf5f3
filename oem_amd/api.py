"""Host-side mirror of the reference's R front ends for the dense Gaussian path.

    oem()      <-> R/oem.R:162-507      (.Call("oem_fit_dense", ...)  R/oem.R:556-575)
    oem_xtx()  <-> R/oem_xtx.R:109-360  (.Call("oem_xtx", ...)        R/oem_xtx.R:389-406)
    big_oem()  <-> R/big_oem.R:121-441  (.Call("oem_fit_big", ...)    R/big_oem.R:447-491)

Same argument names (dots become underscores, `lambda` is `lambda_`), same defaults, same
validation messages, same result structure (`beta` / `lambda` / `niter` / `loss` lists, one entry
per penalty, plus `d`, `nobs`, `nvars`, `penalty`, `family`, `varnames`, `nzero`).  All numerics run
in liboemgpu.so (HIP kernels); there is no CPU fallback.

x may be a numpy array (host: the drop-in entry points upload it) or a CUDA/HIP torch tensor
(device resident: the *_dev entry points are used and X is never copied to the host).
"""
import ctypes as C
import warnings

import numpy as np

from . import _lib as L

PENALTIES = L.PENALTIES


class OemFit(dict):
    """The list returned by oem()/oem.xtx()/big.oem() (classes "oemfit_gaussian", "oem")."""

    r_class = ("oemfit_gaussian", "oem")

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


# ------------------------------------------------------------------------------------------ helpers
def _match_penalty(penalty):
    """match.arg(penalty, several.ok=TRUE) (R/oem.R:202-208): default is the first choice only."""
    if penalty is None:
        return [PENALTIES[0]]
    if isinstance(penalty, str):
        penalty = [penalty]
    out = []
    for q in penalty:
        hits = [c for c in PENALTIES if c == q] or [c for c in PENALTIES if c.startswith(q)]
        if len(hits) != 1:
            raise ValueError("'arg' should be one of " + ", ".join(f"'{c}'" for c in PENALTIES))
        out.append(hits[0])
    return out


def _is_torch_cuda(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def _dptr(a):
    return a.ctypes.data_as(L._dp) if a is not None and a.size > 0 else L._dp()


def _iptr(a):
    return a.ctypes.data_as(L._ip) if a is not None and a.size > 0 else L._ip()


class _Args:
    """Builds oemgpu_opts and keeps the numpy buffers alive."""

    def __init__(self, penalty, lam_list, nlambda, lambda_min_ratio, alpha, gamma, tau, tol, maxit, accelerate,
                 compute_loss, penalty_factor, groups, unique_groups, group_weights, device=-1, ngpus=0, devices=None,
                 upload_threads=0, interrupt=None):
        self.pen = np.array([PENALTIES.index(q) for q in penalty], dtype=np.int32)
        self.pf = np.ascontiguousarray(penalty_factor, dtype=np.float64)
        nlu = len(lam_list[0]) if lam_list else 0
        self.lam = np.ascontiguousarray(np.stack(lam_list), dtype=np.float64) if nlu > 0 else None
        self.groups = np.ascontiguousarray(groups, dtype=np.int32)
        self.ug = np.ascontiguousarray(unique_groups, dtype=np.int32)
        self.gw = np.ascontiguousarray(group_weights, dtype=np.float64)
        o = L.OemgpuOpts()
        o.npen = len(self.pen); o.penalty = _iptr(self.pen)
        o.nlambda = int(nlambda); o.lambda_min_ratio = float(lambda_min_ratio)
        o.lambda_user = _dptr(self.lam); o.nlambda_user = nlu
        o.alpha, o.gamma, o.tau, o.tol = float(alpha), float(gamma), float(tau), float(tol)
        o.maxit, o.accelerate, o.compute_loss = int(maxit), int(bool(accelerate)), int(bool(compute_loss))
        o.penalty_factor = _dptr(self.pf)
        o.groups = _iptr(self.groups); o.ngroupvars = self.groups.size
        o.unique_groups = _iptr(self.ug); o.ngroups = self.ug.size
        o.group_weights = _dptr(self.gw); o.n_group_weights = self.gw.size
        o.device = int(device)
        # host-resident entry points: rows over `ngpus` devices inside the library (include/oemgpu.h)
        self.devices = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        o.ngpus = int(ngpus) if self.devices is None else len(self.devices)
        o.devices = _iptr(self.devices) if self.devices is not None else None
        o.upload_threads = int(upload_threads)
        if interrupt is not None:
            self._cb = L.OemgpuOpts._fields_[-2][1](lambda _arg: int(bool(interrupt())))      # kept alive with the struct
            o.interrupt = self._cb
        self.c = o
        self.nl = nlu if nlu > 0 else int(nlambda)
        self.npen = len(self.pen)

    def subset(self, idx):
        """the penalties idx of this call as a call of their own (penalties are independent cold starts,
        ref src/oem_dense.cpp:206-246): same options, their rows of a user-supplied lambda"""
        o = self.c
        sub = _Args([PENALTIES[self.pen[k]] for k in idx], [] if self.lam is None else [self.lam[k] for k in idx], o.nlambda,
                    o.lambda_min_ratio, o.alpha, o.gamma, o.tau, o.tol, o.maxit, o.accelerate, o.compute_loss, self.pf,
                    self.groups, self.ug, self.gw, device=o.device)
        return sub

    def outputs(self, rows):
        self.beta = np.zeros((self.npen, self.nl, rows))
        self.lam_out = np.zeros((self.npen, self.nl))
        self.niter = np.zeros((self.npen, self.nl), dtype=np.int32)
        self.loss = np.zeros((self.npen, self.nl))
        self.d = C.c_double(0.0)
        return [_dptr(self.beta), _dptr(self.lam_out), _iptr(self.niter), _dptr(self.loss), C.byref(self.d)]


def _lambda_list(lambda_, npen):
    """R/oem.R:366-404"""
    if isinstance(lambda_, (list, tuple)) and len(lambda_) > 0 and np.ndim(lambda_[0]) > 0:
        if len(lambda_) != npen:
            raise ValueError("If list of lambda vectors is provided, it must be \n"
                             "                  the same length as the number of penalties fit")
        n0 = len(lambda_[0])
        out = []
        for l in lambda_:
            if l is None or len(l) < 1:
                raise ValueError("Provided lambda vector must have at least one value")
            if len(l) != n0:
                raise ValueError("All provided lambda vectors must have same length")
            out.append(np.sort(np.asarray(l, dtype=np.float64))[::-1].copy())
        return out
    lam = np.sort(np.asarray(lambda_, dtype=np.float64).ravel())[::-1].copy()
    return [lam.copy() for _ in range(npen)]


def _group_setup(penalty, groups, group_weights, p, intercept_adds_zero_group):
    """R/oem.R:287-338 (dense gaussian: the intercept never adds a group) and R/big_oem.R:226-259."""
    if any("grp" in q for q in penalty):
        groups = np.asarray(groups).ravel()
        if len(groups) != p:
            raise ValueError("If any group penalty is used groups must have same length as number of columns in x")
        unique_groups = np.sort(np.unique(groups))
        has_zero = bool(np.any(unique_groups == 0))
        if group_weights is not None:
            group_weights = np.asarray(group_weights, dtype=np.float64).ravel().copy()
            # `group.weights[zero.idx] <- 0` indexes with the VALUE 0, a no-op in R: kept as is
            if not has_zero and intercept_adds_zero_group:
                unique_groups = np.concatenate([[0], unique_groups])
                group_weights = np.concatenate([[0.0], group_weights])
            if len(group_weights) != len(unique_groups):
                raise ValueError("group.weights must have same length as the number of groups")
        else:
            group_weights = np.zeros(0)
            if not has_zero and intercept_adds_zero_group:
                unique_groups = np.sort(np.concatenate([[0], unique_groups]))
        if intercept_adds_zero_group:
            groups = np.concatenate([[0], groups])
        return groups.astype(np.int32), unique_groups.astype(np.int32), group_weights
    return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)


def _common_checks(nlambda, lambda_min_ratio, maxit, irls_maxit, tol, irls_tol):
    if lambda_min_ratio >= 1 or lambda_min_ratio <= 0:
        raise ValueError("lambda.min.ratio must be between 0 and 1")
    if int(np.ravel(nlambda)[0]) <= 0:
        raise ValueError("nlambda must be a positive integer")
    if maxit <= 0 or irls_maxit <= 0:
        raise ValueError("maxit and irls.maxit should be positive")
    if tol < 0 or irls_tol < 0:
        raise ValueError("tol and irls.tol should be nonnegative")


def _nonzero_lists(beta, zero_first_row=True):
    """predict.oem(type="nonzero") (R/methods.R:93-100): row 1 is always treated as the intercept (quirk Q16)."""
    b = np.array(beta, copy=True)
    if zero_first_row:
        b[0, :] = 0
    return [np.nonzero(np.abs(b[:, j]) > 0)[0] + 1 if np.any(np.abs(b[:, j]) > 0) else None for j in range(b.shape[1])]


def _decorate(a, penalty, varnames, intercept_row, n, p, family="gaussian"):
    """R/oem.R:487-507"""
    res = OemFit()
    res["beta"], res["lambda"], res["niter"], res["loss"] = [], [], [], []
    for k, name in enumerate(penalty):
        b = a.beta[k].T.copy()                       # rows x nl
        if name == "ols":                            # quirk Q9: vector reshaped to a 1-column matrix
            res["beta"].append(b[:, :1]); res["niter"].append(int(a.niter[k, 0])); res["loss"].append(float(a.loss[k, 0]))
        else:
            res["beta"].append(b); res["niter"].append(a.niter[k].copy()); res["loss"].append(a.loss[k].copy())
        res["lambda"].append(a.lam_out[k].copy())
    res["d"] = a.d.value
    res["rownames"] = (["(Intercept)"] if intercept_row else []) + list(varnames)
    # sapply(predict.oem(type = "nonzero"), length): row 1 is always dropped as "the intercept" (quirk Q16)
    res["nzero"] = [(np.abs(np.asarray(b)[1:]) > 0).sum(axis=0) for b in res["beta"]]
    if n is not None:
        res["nobs"] = n
    res["nvars"] = p
    res["penalty"] = list(penalty)
    res["family"] = family
    res["varnames"] = list(varnames)
    return res


def _device_matrix(x):
    """(data_ptr, n, p, ld, keepalive) of a torch device matrix in column-major order."""
    import torch
    if x.dtype != torch.float64:
        x = x.to(torch.float64)
    n, p = x.shape
    if x.stride(0) == 1 and x.stride(1) >= n:           # already column-major
        return x.data_ptr(), n, p, x.stride(1), x
    xt = x.t().contiguous()                              # (p, n) row-major == (n, p) column-major
    return xt.data_ptr(), n, p, n, xt


_ctx_cache = {}


ENGINES = ("none", "rows", "coop", "rowcoop", "symcoop", "launches", "wcoop", "wres", "wstream", "wlaunches")   # OEMGPU_ENGINE_* (include/oemgpu.h)


def last_path_engine(ctx=None):
    """(name of the kernel family that ran the most recent penalty x lambda path on this context, number of persistent-engine calls
    so far that timed out and were made again with launches) -- for device-resident inputs, whose calls run on `context()`."""
    e, f = C.c_int32(0), C.c_int32(0)
    L.check(L.lib().oemgpu_last_path_engine(ctx if ctx is not None else context(), C.byref(e), C.byref(f)))
    return ENGINES[e.value], f.value


def last_placement(ctx=None):
    """"none" / "one-xcd" / "refused": whether the cooperating engine of the most recent path launch on this context ran with all
    workgroups of an instance on one XCD (include/oemgpu.h: oemgpu_last_placement)."""
    return ("none", "one-xcd", "refused", "crowded")[L.lib().oemgpu_last_placement(ctx if ctx is not None else context())]


def context(device=None, stream=None):
    """A cached oemgpu_ctx per (device, stream).  stream: a torch.cuda.Stream or None (own stream)."""
    import torch
    if device is None:
        device = torch.cuda.current_device()
    sptr = None if stream is None else int(stream.cuda_stream)
    key = (int(device), sptr)
    if key not in _ctx_cache:
        h = L.lib().oemgpu_create(int(device), C.c_void_p(sptr) if sptr else None)
        if not h:
            raise L.OemgpuError(-2, L.lib().oemgpu_last_error().decode())
        _ctx_cache[key] = h
    return _ctx_cache[key]


# ------------------------------------------------------------------------------------------ oem()
def oem(x, y, family="gaussian", penalty=None, weights=(), lambda_=(), nlambda=100, lambda_min_ratio=None,
        alpha=1.0, gamma=3.0, tau=0.5, groups=(), penalty_factor=None, group_weights=None, standardize=True,
        intercept=True, maxit=500, tol=1e-7, irls_maxit=100, irls_tol=1e-3, accelerate=False, ncores=-1,
        compute_loss=False, hessian_type="upper.bound", varnames=None, ngpus=0, devices=None, upload_threads=0,
        interrupt=None, _entry_weights=None):
    """oem(): R/oem.R:162-507, dense gaussian branch.  ngpus / devices / upload_threads / interrupt: the host-resident
    options of include/oemgpu.h (`options` gains ngpus / device; absent => one GPU)."""
    L.sync_switches()
    if family not in ("gaussian", "binomial"):
        raise ValueError("'arg' should be one of 'gaussian', 'binomial'")
    penalty = _match_penalty(penalty)
    if getattr(x, "ndim", 0) != 2:
        raise ValueError("x must have at least two columns")
    n, p = x.shape
    if p > n:
        warnings.warn("oem() is optimized for n >> p settings and may be very slow when p > n")
    if p < 2:
        raise ValueError("x must have at least two columns")
    is_sparse = type(x).__module__.startswith("scipy.sparse")         # R/oem.R:236-242: sparseMatrix -> dgCMatrix
    if len(weights) > 0:
        raise ValueError("weights not implemented yet.")
    ylen = y.shape[0] if hasattr(y, "shape") else len(y)
    if ylen != n:
        raise ValueError("x and y lengths do not match")
    if family == "binomial":
        raise NotImplementedError("family='binomial' is outside the dense Gaussian hot path (ref src/oem_logistic_dense.cpp)")
    if penalty_factor is None:
        penalty_factor = np.ones(p)
    penalty_factor = np.asarray(penalty_factor, dtype=np.float64).ravel()
    if varnames is None:
        varnames = [f"V{i + 1}" for i in range(p)]
    if len(penalty_factor) != p:
        raise ValueError("penalty.factor must have same length as number of columns in x")
    groups, unique_groups, group_weights = _group_setup(penalty, groups, group_weights, p, bool(intercept) and is_sparse)   # R/oem.R:296-338
    if lambda_min_ratio is None:
        lambda_min_ratio = 0.01 if n < p else 0.0001
    _common_checks(nlambda, float(lambda_min_ratio), maxit, irls_maxit, tol, irls_tol)
    lam_list = _lambda_list(lambda_, len(penalty))
    a = _Args(penalty, lam_list, int(np.ravel(nlambda)[0]), lambda_min_ratio, alpha, gamma, tau, tol, maxit, accelerate,
              compute_loss, penalty_factor, groups, unique_groups, group_weights, ngpus=ngpus, devices=devices,
              upload_threads=upload_threads, interrupt=interrupt)
    lib = L.lib()
    if is_sparse:                                                      # oem_fit_sparse (ref src/oem_sparse.cpp:30-267)
        import scipy.sparse as sp
        xc = sp.csc_matrix(x, dtype=np.float64); xc.sort_indices()
        colptr = np.ascontiguousarray(xc.indptr, dtype=np.int64); rowidx = np.ascontiguousarray(xc.indices, dtype=np.int32)
        vals = np.ascontiguousarray(xc.data, dtype=np.float64)
        yh = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        a.c.accelerate = 0                                             # oemSparse has no acceleration
        L.check(lib.oemgpu_fit_sparse(n, p, colptr.ctypes.data, _iptr(rowidx), _dptr(vals), _dptr(yh), int(bool(standardize)),
                                      int(bool(intercept)), C.byref(a.c), *a.outputs(p + 1)))
        return _decorate(a, penalty, varnames, True, n, p)
    if _entry_weights is not None:                                     # oem_fit_dense with a weights vector (oem_fit_dense_weighted below)
        if is_sparse:
            raise ValueError("observation weights: dense x only (oem_fit_sparse ignores its weights argument)")
        wh = np.ascontiguousarray(np.asarray(_entry_weights, dtype=np.float64).reshape(-1))
        if wh.shape[0] != n:
            raise ValueError("length of weights not same as number of observations in x")       # R/oem.R:261-266
        if _is_torch_cuda(x):
            import torch
            xp, n_, p_, ld, keep = _device_matrix(x)
            yd = torch.as_tensor(np.asarray(y.cpu() if _is_torch_cuda(y) else y, dtype=np.float64), device=x.device).reshape(-1)
            wd = torch.as_tensor(wh, device=x.device)
            ctx = context(x.device.index)
            torch.cuda.current_stream(x.device).synchronize()
            L.check(lib.oemgpu_fit_dense_weighted_dev(ctx, xp, n, ld, p, yd.data_ptr(), wd.data_ptr(), int(bool(standardize)),
                                                      int(bool(intercept)), C.byref(a.c), *a.outputs(p + 1)))
            del keep
        else:
            xh = np.asfortranarray(x, dtype=np.float64)
            yh = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
            L.check(lib.oemgpu_fit_dense_weighted(_dptr(xh), n, p, _dptr(yh), _dptr(wh), int(bool(standardize)), int(bool(intercept)),
                                                  C.byref(a.c), *a.outputs(p + 1)))
        return _decorate(a, penalty, varnames, True, n, p)
    if _is_torch_cuda(x):
        import torch
        xp, n_, p_, ld, keep = _device_matrix(x)
        yd = y if _is_torch_cuda(y) else torch.as_tensor(np.asarray(y, dtype=np.float64), device=x.device)
        yd = yd.to(torch.float64).contiguous().reshape(-1)
        ctx = context(x.device.index)
        torch.cuda.current_stream(x.device).synchronize()
        L.check(lib.oemgpu_fit_dense_dev(ctx, xp, n, ld, p, yd.data_ptr(), int(bool(standardize)), int(bool(intercept)),
                                         C.byref(a.c), *a.outputs(p + 1)))
        del keep
    else:
        xh = np.asfortranarray(x, dtype=np.float64)
        yh = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        L.check(lib.oemgpu_fit_dense(_dptr(xh), n, p, _dptr(yh), int(bool(standardize)), int(bool(intercept)),
                                     C.byref(a.c), *a.outputs(p + 1)))
    return _decorate(a, penalty, varnames, True, n, p)


def oem_fit_dense_weighted(x, y, weights, **kw):
    """What the compiled entry `oem_fit_dense` computes when it is handed a non-empty `weights_` (ref src/oem_dense.cpp:34,75,152,162;
    src/oem_dense.h:368-414, 699-707, 759-770; src/DataStd.h:94-202).  R's oem() stops with "weights not implemented yet" before it
    gets there (R/oem.R:244) and so does `oem()` here; this is the entry below that check, with oem()'s other arguments."""
    return oem(x, y, _entry_weights=weights, **kw)


# ------------------------------------------------------------------------------------------ oem.xtx()
def oem_xtx(xtx, xty, family="gaussian", penalty=None, lambda_=(), nlambda=100, lambda_min_ratio=None, alpha=1.0,
            gamma=3.0, tau=0.5, groups=(), scale_factor=(), penalty_factor=None, group_weights=None, maxit=500,
            tol=1e-7, irls_maxit=100, irls_tol=1e-3, varnames=None, interrupt=None):
    """oem.xtx(): R/oem_xtx.R:109-360.  interrupt: a callable polled on the calling thread while the library waits for the GPU
    (the R shim's is R_CheckUserInterrupt, ref src/oem_xtx.cpp:160-163); True ends the call with OEMGPU_ERR_INTERRUPTED."""
    L.sync_switches()
    penalty = _match_penalty(penalty)
    if getattr(xtx, "ndim", 0) != 2:
        raise ValueError("xtx must be a matrix")
    if xtx.shape[0] != xtx.shape[1]:
        raise ValueError("xtx must be a square matrix equal to X'X. do NOT provide design matrix")
    p = xtx.shape[1]
    xlen = xty.shape[0] if hasattr(xty, "shape") else len(xty)
    if p != xlen:
        raise ValueError("xty must have length equal to the number of columns and rows of xtx. do NOT provide response vector")
    if p < 2:
        raise ValueError("xtx must have at least two columns")
    if family == "binomial":
        raise ValueError("binomial not implemented yet")
    if penalty_factor is None:
        penalty_factor = np.ones(p)
    penalty_factor = np.asarray(penalty_factor, dtype=np.float64).ravel()
    if varnames is None:
        varnames = [f"V{i + 1}" for i in range(p)]
    if len(penalty_factor) != p:
        raise ValueError("penalty.factor must have same length as number of columns in x")
    if any("grp" in q for q in penalty) and len(np.ravel(groups)) != p:
        raise ValueError("groups must have same length as number of columns in x")
    groups, unique_groups, group_weights = _group_setup(penalty, groups, group_weights, p, False)
    if lambda_min_ratio is None:
        lambda_min_ratio = 0.0001
    _common_checks(nlambda, float(lambda_min_ratio), maxit, irls_maxit, tol, irls_tol)
    lam_list = _lambda_list(lambda_, len(penalty))
    sf = np.asarray(scale_factor, dtype=np.float64).ravel()
    if sf.size > 0 and sf.size != p:
        raise ValueError("scale.factor must be same length as xty (nvars)")
    a = _Args(penalty, lam_list, int(np.ravel(nlambda)[0]), lambda_min_ratio, alpha, gamma, tau, tol, maxit, False,
              False, penalty_factor, groups, unique_groups, group_weights, interrupt=interrupt)
    lib = L.lib()
    if _is_torch_cuda(xtx):
        import torch
        xp, _, _, ld, keep = _device_matrix(xtx)
        if ld != p:
            keep = xtx.t().contiguous(); xp = keep.data_ptr()
        yd = xty if _is_torch_cuda(xty) else torch.as_tensor(np.asarray(xty, dtype=np.float64), device=xtx.device)
        yd = yd.to(torch.float64).contiguous().reshape(-1)
        ctx = context(xtx.device.index)
        torch.cuda.current_stream(xtx.device).synchronize()
        L.check(lib.oemgpu_fit_xtx_dev(ctx, xp, yd.data_ptr(), p, _dptr(sf), C.byref(a.c), *a.outputs(p)))
        del keep
    else:
        xh = np.asfortranarray(xtx, dtype=np.float64)
        yh = np.ascontiguousarray(np.asarray(xty, dtype=np.float64).reshape(-1))
        L.check(lib.oemgpu_fit_xtx(_dptr(xh), _dptr(yh), p, _dptr(sf), C.byref(a.c), *a.outputs(p)))
    return _decorate(a, penalty, varnames, False, None, p)


# ------------------------------------------------------------------------------------------ big.oem()
def big_oem(x, y, family="gaussian", penalty=None, weights=(), lambda_=(), nlambda=100, lambda_min_ratio=None,
            alpha=1.0, gamma=3.0, tau=0.5, groups=(), penalty_factor=None, group_weights=None, standardize=True,
            intercept=True, maxit=500, tol=1e-7, irls_maxit=100, irls_tol=1e-3, compute_loss=False, gigs=4.0,
            hessian_type="full", varnames=None, ngpus=0, devices=None, upload_threads=0, interrupt=None):
    """big.oem(): R/big_oem.R:121-441.  x: a (host) matrix or a list of row shards (the big.matrix stand-in);
    y: a vector or the matching list of shards."""
    L.sync_switches()
    penalty = PENALTIES if penalty is None else _match_penalty(penalty)      # match.arg(several.ok=TRUE), no default narrowing
    shards = list(x) if isinstance(x, (list, tuple)) else [x]
    yshards = list(y) if isinstance(y, (list, tuple)) else [y]
    if family == "binomial":
        raise ValueError("binomial case not implemented yet")
    if any(getattr(s, "ndim", 0) != 2 for s in shards):
        raise ValueError("x must have at least two columns")
    p = shards[0].shape[1]
    n = sum(s.shape[0] for s in shards)
    if p < 2:
        raise ValueError("x must have at least two columns")
    if len(weights) > 0:
        raise ValueError("weights not implemented yet.")
    if len(yshards) != len(shards) or sum(len(v) for v in yshards) != n:
        raise ValueError("x and y lengths do not match")
    if penalty_factor is None:
        penalty_factor = np.ones(p)
    penalty_factor = np.asarray(penalty_factor, dtype=np.float64).ravel()
    if varnames is None:
        varnames = [f"V{i + 1}" for i in range(p)]
    if len(penalty_factor) != p:
        raise ValueError("penalty.factor must have same length as number of columns in x")
    if any("grp" in q for q in penalty) and len(np.ravel(groups)) != p:
        raise ValueError("groups must have same length as number of columns in x")
    groups, unique_groups, group_weights = _group_setup(penalty, groups, group_weights, p, bool(intercept))
    if lambda_min_ratio is None:
        lambda_min_ratio = 0.01 if n < p else 0.0001
    _common_checks(nlambda, float(lambda_min_ratio), maxit, irls_maxit, tol, irls_tol)
    lam_list = _lambda_list(lambda_, len(penalty))
    a = _Args(penalty, lam_list, int(np.ravel(nlambda)[0]), lambda_min_ratio, alpha, gamma, tau, tol, maxit, False,
              compute_loss, penalty_factor, groups, unique_groups, group_weights, ngpus=ngpus, devices=devices,
              upload_threads=upload_threads, interrupt=interrupt)
    xs = [np.asfortranarray(s, dtype=np.float64) for s in shards]
    ys = [np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1)) for v in yshards]
    ns = (C.c_int64 * len(xs))(*[s.shape[0] for s in xs])
    xp = (L._dp * len(xs))(*[_dptr(s) for s in xs])
    yp = (L._dp * len(ys))(*[_dptr(v) for v in ys])
    L.check(L.lib().oemgpu_fit_big(xp, ns, len(xs), p, yp, int(bool(standardize)), int(bool(intercept)),
                                   C.byref(a.c), *a.outputs(p + 1)))
    return _decorate(a, penalty, varnames, True, n, p)



# ------------------------------------------------------------------------------------------ xval.oem()
def _getmin(lam, cvm, cvsd):
    """R/utils.R:3-26 (getmin)"""
    lmin_models, l1se_models, cv_models = [], [], []
    for m in range(len(cvm)):
        cvmin = np.min(cvm[m])
        idmin = cvm[m] <= cvmin
        lmin = np.max(lam[m][idmin])
        cv_models.append(np.min(cvm[m][idmin]))
        i0 = int(np.nonzero(lam[m] == lmin)[0][0])
        semin = (cvm[m] + cvsd[m])[i0]
        l1se_models.append(np.max(lam[m][cvm[m] < semin]))
        lmin_models.append(lmin)
    mmin = int(np.argmin(cv_models))
    return {"lambda.min": lmin_models[mmin], "model.min": mmin + 1, "lambda.1se": l1se_models[mmin],
            "lambda.min.models": np.array(lmin_models), "lambda.1se.models": np.array(l1se_models)}


_TYPE_MEASURES = ("mse", "deviance", "class", "auc", "mae")


def xval_oem(x, y, nfolds=10, foldid=None, type_measure=None, ncores=-1, family="gaussian", penalty=None, weights=(),
             lambda_=(), nlambda=100, lambda_min_ratio=None, alpha=1.0, gamma=3.0, tau=0.5, groups=(), penalty_factor=None,
             group_weights=None, standardize=True, intercept=True, maxit=500, tol=1e-7, irls_maxit=100, irls_tol=1e-3,
             compute_loss=False, varnames=None, rng=None, ngpus=0, devices=None, upload_threads=0, interrupt=None):
    """xval.oem(): R/oem_xval.R:107-460 (gaussian, dense).  foldid: values 1..nfolds; drawn with `rng` (a numpy Generator)
    as sample(rep(seq(nfolds), length = n)) when None.  ngpus / devices (host x only): the rows over several devices inside the
    library, as in oem()."""
    L.sync_switches()
    if family not in ("gaussian", "binomial"):
        raise ValueError("'arg' should be one of 'gaussian', 'binomial'")
    penalty = _match_penalty(penalty)
    if type_measure is None:
        type_measure = "default"
    elif type_measure not in _TYPE_MEASURES:
        raise ValueError("'arg' should be one of " + ", ".join("'%s'" % t for t in _TYPE_MEASURES))
    if family == "binomial":
        raise ValueError("binomial models not yet supported for xval, use cv.oem() instead")
    if getattr(x, "ndim", 0) != 2:
        raise ValueError("x must have at least two columns")
    n, p = x.shape
    if p >= n:
        raise ValueError("number of observations must be greater than the number of variables\n"
                         "             for xval, use cv.oem instead, or, preferably, use another package such as\n"
                         "             glmnet for the lasso, ncvreg for MCP/SCAD, or grpreg or gglasso for group lasso.")
    if p < 2:
        raise ValueError("x must have at least two columns")
    if foldid is None:
        g = np.random.default_rng() if rng is None else rng
        foldid = g.permutation(np.resize(np.arange(1, int(nfolds) + 1), n))
    else:
        foldid = np.asarray(foldid).ravel()
        nfolds = int(foldid.max())
    if nfolds < 3:
        raise ValueError("nfolds must be bigger than 3; nfolds=10 recommended")
    if type(x).__module__.startswith("scipy.sparse"):
        raise ValueError("sparse matrices not supported yet")
    ylen = y.shape[0] if hasattr(y, "shape") else len(y)
    if ylen != n or len(foldid) != n:
        raise ValueError("x and y lengths do not match")
    wh = None
    if len(weights) > 0:                                         # R/oem_xval.R:216-223
        if len(weights) != n:
            raise ValueError("length of weights not same as number of observations in x")
        wh = np.ascontiguousarray(np.asarray(weights, dtype=np.float64).reshape(-1))
    if penalty_factor is None:
        penalty_factor = np.ones(p)
    if varnames is None:
        varnames = [f"V{i + 1}" for i in range(p)]
    penalty_factor = np.asarray(penalty_factor, dtype=np.float64).ravel()
    if len(penalty_factor) != p:
        raise ValueError("penalty.factor must have same length as number of columns in x")
    if any("grp" in q for q in penalty) and len(np.ravel(groups)) != p:
        raise ValueError("groups must have same length as number of columns in x")
    groups, unique_groups, group_weights = _group_setup(penalty, groups, group_weights, p, bool(intercept))
    if lambda_min_ratio is None:
        lambda_min_ratio = 0.01 if n < p else 0.0001
    _common_checks(nlambda, float(lambda_min_ratio), maxit, irls_maxit, tol, irls_tol)
    lam_list = _lambda_list(lambda_, len(penalty))
    if type_measure in ("default", "deviance"):                  # R/oem_xval.R:488-497
        type_measure = "mse"
    if type_measure not in ("mse", "mae"):
        warnings.warn("Only 'mse', 'deviance' or 'mae'  available for Gaussian models; 'mse' used")
        type_measure = "mse"
    a = _Args(penalty, lam_list, int(np.ravel(nlambda)[0]), lambda_min_ratio, alpha, gamma, tau, tol, maxit, False,
              compute_loss, penalty_factor, groups, unique_groups, group_weights, ngpus=ngpus, devices=devices,
              upload_threads=upload_threads, interrupt=interrupt)
    out = a.outputs(p + 1)
    cvm = np.zeros((a.npen, a.nl)); cvsd = np.zeros((a.npen, a.nl))
    fid = np.ascontiguousarray(foldid, dtype=np.int32)
    tm = 1 if type_measure == "mae" else 0
    lib = L.lib()
    if _is_torch_cuda(x):
        import torch
        xp, n_, p_, ld, keep = _device_matrix(x)
        yd = y if _is_torch_cuda(y) else torch.as_tensor(np.asarray(y, dtype=np.float64), device=x.device)
        yd = yd.to(torch.float64).contiguous().reshape(-1)
        fd = torch.as_tensor(fid, device=x.device)
        wd = None if wh is None else torch.as_tensor(wh, device=x.device)
        ctx = context(x.device.index)
        torch.cuda.current_stream(x.device).synchronize()
        L.check(lib.oemgpu_xval_dense_dev(ctx, xp, n, ld, p, yd.data_ptr(), None if wd is None else wd.data_ptr(), fd.data_ptr(),
                                          int(nfolds), int(bool(standardize)),
                                          int(bool(intercept)), tm, C.byref(a.c), *out, _dptr(cvm), _dptr(cvsd)))
        del keep
    else:
        xh = np.asfortranarray(x, dtype=np.float64)
        yh = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        L.check(lib.oemgpu_xval_dense(_dptr(xh), n, p, _dptr(yh), None if wh is None else _dptr(wh), _iptr(fid), int(nfolds),
                                      int(bool(standardize)),
                                      int(bool(intercept)), tm, C.byref(a.c), *out, _dptr(cvm), _dptr(cvsd)))
    res = _decorate(a, penalty, varnames, True, n, p)
    res["cvm"] = [cvm[k, :1].copy() if name == "ols" else cvm[k].copy() for k, name in enumerate(penalty)]
    res["cvsd"] = [cvsd[k, :1].copy() if name == "ols" else cvsd[k].copy() for k, name in enumerate(penalty)]
    res["name"] = {"mse": "Mean-Squared Error", "mae": "Mean Absolute Error"}[type_measure]
    res["foldid"] = fid
    res.update(_getmin([l[:len(c)] for l, c in zip(res["lambda"], res["cvm"])], res["cvm"], res["cvsd"]))
    res["cvup"] = [m + s for m, s in zip(res["cvm"], res["cvsd"])]
    res["cvlo"] = [m - s for m, s in zip(res["cvm"], res["cvsd"])]
    res["best.model"] = penalty[res["model.min"] - 1]
    return res



# ------------------------------------------------------------------------------------------ cv.oem()
def cv_oem(x, y, penalty=None, weights=(), lambda_=None, type_measure=None, nfolds=10, foldid=None, grouped=True, keep=False,
           rng=None, parallel=False, **kw):
    """cv.oem(): R/cv_oem.R:56-221 with cv.oemfit_gaussian (:349-423) and cvcompute (R/utils.R:128-144): K + 1 calls of oem(),
    every fold on its own lambda sequence, errors interpolated onto the full fit's lambdas.
    parallel (R/cv_oem.R:32, 129-150: the folds through foreach): the fold fits from a few host threads at once.  On one GPU that
    pays where a fit leaves most of the chip idle: the path kernels of n >> p fits (one CU each) overlap with other folds' moment
    kernels, and p >= n fits on the cooperating-workgroup engine (a quarter of the CUs each) run side by side -- they queue for CU
    slots by themselves.  Same results as the sequential loop.  (Measured, six folds: 300 x 1500 136 -> 117 ms; 5000 x 40 8 -> 17 ms --
    fits of a millisecond lose more to the thread hand-over than the overlap gains: the default stays sequential, as in R.)"""
    penalty = _match_penalty(penalty)
    if type_measure is None:
        type_measure = "default"
    elif type_measure not in _TYPE_MEASURES:
        raise ValueError("'arg' should be one of " + ", ".join("'%s'" % t for t in _TYPE_MEASURES))
    if lambda_ is not None and len(lambda_) < 2:
        raise ValueError("Need more than one value of lambda for cv.oem")
    if len(weights) > 0:
        raise ValueError("weights not implemented yet.")
    xh = np.asarray(x.cpu().numpy() if _is_torch_cuda(x) else x, dtype=np.float64)
    yh = np.asarray(y.cpu().numpy() if _is_torch_cuda(y) else y, dtype=np.float64).reshape(-1)
    n = xh.shape[0]
    lam_arg = () if lambda_ is None else lambda_
    fit0 = oem(x, y, penalty=penalty, lambda_=lam_arg, **kw)
    nz = [np.array([0 if v is None else len(v) for v in predict(fit0, type="nonzero", which_model=m)]) for m in range(len(penalty))]
    if foldid is None:
        g = np.random.default_rng() if rng is None else rng
        foldid = g.permutation(np.resize(np.arange(1, int(nfolds) + 1), n))
    else:
        foldid = np.asarray(foldid).ravel()
        nfolds = int(foldid.max())
    if nfolds < 3:
        raise ValueError("nfolds must be bigger than 3; nfolds=10 recommended")
    def fold_fit(i):
        keep_rows = foldid != i
        return oem(np.asfortranarray(xh[keep_rows]), yh[keep_rows], penalty=penalty, lambda_=lam_arg, **kw)
    # the p > n warning was given once, by the full fit: only THAT message is silenced for the fold fits, and the filter list is
    # touched by the calling thread alone, around the whole block (catch_warnings is process-global state: not for worker threads)
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", message=".*optimized for n >> p.*")
        if parallel:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(int(nfolds), 3 if parallel is True else int(parallel))) as ex:
                outlist = list(ex.map(fold_fit, range(1, nfolds + 1)))
        else:
            outlist = [fold_fit(i) for i in range(1, nfolds + 1)]
    # cv.oemfit_gaussian
    if type_measure in ("default", "deviance"):
        type_measure = "mse"
    if type_measure not in ("mse", "mae"):
        warnings.warn("Only 'mse', 'deviance' or 'mae'  available for Gaussian models; 'mse' used")
        type_measure = "mse"
    lam = [np.asarray(l, dtype=np.float64) for l in fit0["lambda"]]
    nmodels = len(penalty)
    which_lam = [lam[m] >= max(np.min(o["lambda"][m]) for o in outlist) for m in range(nmodels)]     # no extrapolation to smaller lambdas
    predlist = [np.full((n, len(lam[0])), np.nan) for _ in range(nmodels)]
    nlams = np.zeros(nfolds, dtype=int)
    for i in range(1, nfolds + 1):
        rows = foldid == i
        for m in range(nmodels):
            preds = predict(outlist[i - 1], xh[rows], s=lam[m][which_lam[m]], which_model=m)
            nlami = int(which_lam[m].sum())
            predlist[m][rows, :nlami] = preds
        nlams[i - 1] = nlami
    cvraw = [(yh[:, None] - pm) ** 2 if type_measure == "mse" else np.abs(yh[:, None] - pm) for pm in predlist]
    if n / nfolds < 3 and grouped:
        warnings.warn("Option grouped=FALSE enforced in cv.glmnet, since < 3 observations per fold")
        grouped = False
    w = [np.ones(n) for _ in range(nmodels)]
    N = [n - np.isnan(pm).sum(axis=0) for pm in predlist]
    if grouped:                                                   # cvcompute: fold means, weighted by fold size
        wisum = np.array([np.sum(foldid == i) for i in range(1, nfolds + 1)], dtype=np.float64)
        for m in range(nmodels):
            out = np.full((nfolds, cvraw[m].shape[1]), np.nan)
            good = np.zeros_like(out)
            for i in range(nfolds):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out[i] = np.nanmean(cvraw[m][foldid == i + 1], axis=0)
                good[i, :nlams[i]] = 1
            cvraw[m], w[m], N[m] = out, wisum, good.sum(axis=0)

    def wmean(a, wt):
        ok = ~np.isnan(a)
        return np.array([np.sum(a[ok[:, j], j] * wt[ok[:, j]]) / np.sum(wt[ok[:, j]]) if ok[:, j].any() else np.nan
                         for j in range(a.shape[1])])
    cvm = [wmean(cvraw[m], w[m]) for m in range(nmodels)]
    with np.errstate(invalid="ignore", divide="ignore"):
        cvsd = [np.sqrt(wmean((cvraw[m] - cvm[m]) ** 2, w[m]) / (N[m] - 1)) for m in range(nmodels)]
    nas = np.zeros(len(lam[0]), dtype=bool)
    for m in range(nmodels):
        nas |= np.isnan(cvsd[m])
    if nas.any():
        cvm = [c[~nas] for c in cvm]; cvsd = [c[~nas] for c in cvsd]
        nz = [c[~nas] for c in nz]; lam = [l[~nas] for l in lam]
    res = OemFit()
    res.update({"lambda": lam, "cvm": cvm, "cvsd": cvsd, "cvup": [a + b for a, b in zip(cvm, cvsd)],
                "cvlo": [a - b for a, b in zip(cvm, cvsd)], "nzero": nz,
                "name": {"mse": "Mean-Squared Error", "mae": "Mean Absolute Error"}[type_measure], "oem.fit": fit0})
    if keep:
        res["fit.preval"] = predlist; res["foldid"] = foldid
    res.update(_getmin(lam, cvm, cvsd))
    res["best.model"] = penalty[res["model.min"] - 1]
    res["penalty"] = list(penalty)
    return res


def predict_cv(fit, newx=None, which_model="best.model", s="lambda.min", **kw):
    """predict.cv.oem, R/methods.R (same selection rules as predict.xval.oem, on the full-data fit `oem.fit`)."""
    if isinstance(s, str):
        if s not in ("lambda.min", "lambda.1se"):
            raise ValueError("'arg' should be one of 'lambda.min', 'lambda.1se'")
        lam = fit[s]
    else:
        lam = s
    if isinstance(which_model, str):
        if which_model == "best.model":
            mod = fit["model.min"] - 1
        else:
            if which_model not in fit["penalty"]:
                raise ValueError(f"Model {which_model} specified, but {which_model} not computed.")
            mod = fit["penalty"].index(which_model)
    else:
        mod = int(which_model)
    return predict(fit["oem.fit"], newx, s=lam, which_model=mod, **kw)


# ------------------------------------------------------------------------------------------ consumers
def predict(fit, newx=None, s=None, which_model=0, type="link"):
    """predict.oem, R/methods.R:48-109 (which_model is 0-based or a penalty name)."""
    if isinstance(which_model, str):
        if which_model not in fit["penalty"]:
            raise ValueError(f"Model {which_model} specified, but {which_model} not computed.")
        which_model = fit["penalty"].index(which_model)
    if which_model >= len(fit["beta"]):
        raise ValueError(f"Model {which_model + 1} specified, but only {len(fit['beta'])} were computed.")
    nbeta = np.array(fit["beta"][which_model])
    if s is not None:
        lam = np.asarray(fit["lambda"][which_model], dtype=np.float64)
        left, right, frac = _lambda_interp(lam, np.atleast_1d(np.asarray(s, dtype=np.float64)))
        nbeta = nbeta[:, left] * frac + nbeta[:, right] * (1 - frac)
    if type == "coefficients":
        return nbeta
    if type == "nonzero":
        return _nonzero_lists(nbeta)
    if newx is None:
        raise ValueError("A value for 'newx' must be supplied")
    newx = np.asarray(newx, dtype=np.float64)
    if newx.shape[1] < nbeta.shape[0]:
        newx = np.column_stack([np.ones(newx.shape[0]), newx])
    return newx @ nbeta


def predict_xval(fit, newx=None, which_model="best.model", s="lambda.min", **kw):
    """predict.xval.oem, R/methods.R:765-807 (which_model: "best.model", a penalty name, or a 0-based index)."""
    if isinstance(s, str):
        if s not in ("lambda.min", "lambda.1se"):
            raise ValueError("'arg' should be one of 'lambda.min', 'lambda.1se'")
        lam = fit[s]
    elif np.isscalar(s) or isinstance(s, (list, tuple, np.ndarray)):
        lam = s
    else:
        raise ValueError("Invalid form for s")
    if isinstance(which_model, str):
        if which_model == "best.model":
            mod = fit["model.min"] - 1
        else:
            if which_model not in fit["penalty"]:
                raise ValueError(f"Model {which_model} specified, but {which_model} not computed.")
            mod = fit["penalty"].index(which_model)
    else:
        mod = int(which_model)
        if mod >= len(fit["cvm"]):
            raise ValueError(f"Model {mod + 1} specified, but only {len(fit['cvm'])} were computed.")
    return predict(fit, newx, s=lam, which_model=mod, **kw)


def _lambda_interp(lam, s):
    """lambda.interp, R/utils.R (glmnet's interpolation)."""
    if len(lam) == 1:
        z = np.zeros(len(s), dtype=int)
        return z, z, np.ones(len(s))
    s = np.clip(s, lam.min(), lam.max())
    k = len(lam)
    sfrac = (lam[0] - s) / (lam[0] - lam[k - 1])
    lamn = (lam[0] - lam) / (lam[0] - lam[k - 1])
    coord = np.interp(sfrac, lamn, np.arange(k))
    left, right = np.floor(coord).astype(int), np.ceil(coord).astype(int)
    den = lamn[left] - lamn[right]
    with np.errstate(invalid="ignore", divide="ignore"):
        sf = np.where(left == right, 1.0, (sfrac - lamn[right]) / den)
    return left, right, sf


def logLik(fit, which_model=0):
    """logLik.oem, R/methods.R:431-482 (gaussian)."""
    if isinstance(which_model, str):
        which_model = fit["penalty"].index(which_model)
    loss = np.atleast_1d(np.asarray(fit["loss"][which_model], dtype=np.float64))
    if np.all(loss == 1e99):
        raise ValueError("oem object needed compute.loss set to TRUE. logLik not returned")
    n = float(fit["nobs"])
    return -0.5 * n * (np.log(2 * np.pi) - np.log(n) + np.log(loss)) - 0.5 * n
