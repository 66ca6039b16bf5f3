"""Row-sharded oem(): one process per GPU, X'X assembled with ONE all-reduce over RCCL/xGMI.

The Gram build is additive over row blocks (the reference already splits rows across OpenMP threads,
ref src/oem_dense.h:328-358, and across slices, ref src/oem_big.h:329-358).  Each rank holds a
contiguous block of rows, builds the shifted moment buffer M_r of its block with the MFMA kernel,
and the ranks sum them:

    M      <- M_0 + M_1 + ... + M_{N-1}   ((p+2)^2 doubles about c = 0 per rank, ONE collective -- an all-gather -- and one kernel that adds
                                           them in rank order (sum_over_ranks; c1: 83 KB, c5: 532 KB per rank); reduce="allreduce": dist.all_reduce)
    result <- solve_moments(M)            (replicated: the lambda path is a serial chain of tiny GEMVs)

The shift c that guards the centred moments against cancellation would be a function of ALL the data, which a rank does not
have when it starts its Gram pass; an earlier collective for it would cost more than the Gram pass itself at 8 GPUs (the
whole step is latency-bound).  So every rank builds M_r about c = 0 -- right unless some column has |mean| > 16 sd -- and
the solve reports, from the reduced full-data moments, when that was the wrong guess (oemgpu_last_shift_advised).  Every rank
sees the same reduced M, so all of them take the redo branch or none: sample sums, one all-reduce to agree on c, the pass
about c, its all-reduce, the solve.  No sample pass and no second collective for the usual data.

`backend` supplies the three local stages; the product backend is HipBackend (liboemgpu).  Tests run
the same driver under gloo on CPU with a checker backend, which is how the N > 1 logic is covered
without GPUs.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from . import api as _api


_STAGED_ALWAYS = bool(__import__("os").environ.get("OEM_STAGED_ALWAYS"))      # diagnostic: the N > 1 call sequence on one rank too


def _many(dist, group):
    """Do the collectives of the N > 1 call sequence run?  World size > 1 -- or OEM_FORCE_COLLECTIVES=1 with an initialised process
    group of ANY size: a one-GPU box then executes every RCCL call of this module (all-reduce, all-gather) at world size 1, with
    the same bits as the plain call (tests/test_gpu_distributed.py, bench.py's rccl_selfcheck).  Read at call time."""
    if dist is None:
        return False
    return dist.get_world_size(group) > 1 or __import__("os").environ.get("OEM_FORCE_COLLECTIVES") == "1"


class HipBackend:
    """Local stages on this rank's GPU through the C ABI (oemgpu_*_dev)."""

    def __init__(self, device=None, stream=None):
        import torch
        self.torch = torch
        self.device = torch.cuda.current_device() if device is None else device
        # Kernels (launched by liboemgpu on the context's stream) and RCCL collectives (ordered by torch against its
        # CURRENT stream) must share one stream, or an all-reduce can start before the kernel that feeds it has finished.
        # torch's default stream has handle 0, which the C ABI reads as "create your own": so the backend always owns an
        # explicit side stream, and `section()` makes it torch's current stream around kernels AND collectives.
        self.stream = torch.cuda.Stream(device=self.device) if stream is None else stream
        if int(self.stream.cuda_stream) == 0:
            raise ValueError("HipBackend needs an explicit (non-default) torch.cuda.Stream")
        self.ctx = _api.context(self.device, self.stream)
        self.lib = L.lib()

    def section(self):
        """`with backend.section():` -- the work inside (local stages and torch.distributed collectives) is stream-ordered on
        the backend's stream; inputs produced on the caller's stream are waited for on the device, not on the host."""
        self.stream.wait_stream(self.torch.cuda.current_stream(self.device))
        return self.torch.cuda.stream(self.stream)

    def new_buffer(self, n):
        with self.torch.cuda.stream(self.stream):
            return self.torch.zeros(n, dtype=self.torch.float64, device=f"cuda:{self.device}")

    def to_device(self, a):
        """a host float64 array as a device buffer on the backend's stream"""
        with self.torch.cuda.stream(self.stream):
            return self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=f"cuda:{self.device}")

    def to_host(self, t):
        with self.torch.cuda.stream(self.stream):
            return t.detach().cpu().numpy()

    def shift_sums(self, x, n, ld, p, y, out):
        L.check(self.lib.oemgpu_shift_sums_dev(self.ctx, x.data_ptr(), n, ld, p, y.data_ptr(), out.data_ptr()))

    def moments(self, x, n, ld, p, y, sums, out):
        """sums None: accumulate about c = 0"""
        L.check(self.lib.oemgpu_moments_dev(self.ctx, x.data_ptr(), n, ld, p, y.data_ptr(),
                                            None if sums is None else sums.data_ptr(), out.data_ptr()))

    def solve(self, moments, sums, p, semantics, standardize, intercept, args, outs=None):
        """outs: the host result buffers of an earlier args.outputs(p + 1), to be written again (repeated solves)"""
        L.check(self.lib.oemgpu_solve_moments_dev(self.ctx, moments.data_ptr(), None if sums is None else sums.data_ptr(), p, semantics,
                                                  int(bool(standardize)), int(bool(intercept)), C.byref(args.c),
                                                  *(args.outputs(p + 1) if outs is None else outs)))

    def fit_dense(self, x, n, ld, p, y, standardize, intercept, args, outs=None):
        """oemgpu_fit_dense_dev: the whole single-GPU solve (moments, verdict on the shift, redo if advised, paths) in one call"""
        L.check(self.lib.oemgpu_fit_dense_dev(self.ctx, x.data_ptr(), n, ld, p, y.data_ptr(), int(bool(standardize)),
                                              int(bool(intercept)), C.byref(args.c), *(args.outputs(p + 1) if outs is None else outs)))

    # ---- xval.oem over row shards: the three phases of oemgpu_xval_dense_dev (include/oemgpu.h)
    def xval_moments_len(self, p, nfolds, weighted):
        return int(self.lib.oemgpu_xval_moments_len(p, nfolds, int(bool(weighted))))

    def xval_fold_moments(self, x, n, ld, p, y, w, foldid, nfolds, args, out):
        """out <- per-fold moments of the local rows (device buffer); returns the local fold sizes"""
        fold_n = (C.c_int64 * nfolds)()
        L.check(self.lib.oemgpu_xval_fold_moments_dev(self.ctx, x.data_ptr(), n, ld, p, y.data_ptr(), None if w is None else w.data_ptr(),
                                                      foldid.data_ptr(), nfolds, C.byref(args.c), out.data_ptr(), fold_n))
        return np.array(list(fold_n), dtype=np.int64)

    def xval_solve_folds(self, moments, fold_n_total, n_local, p, nfolds, weighted, standardize, intercept, args):
        fn = (C.c_int64 * nfolds)(*[int(v) for v in fold_n_total])
        L.check(self.lib.oemgpu_xval_solve_folds_dev(self.ctx, moments.data_ptr(), fn, n_local, p, nfolds, int(bool(weighted)),
                                                     int(bool(standardize)), int(bool(intercept)), C.byref(args.c), *args.outputs(p + 1)))

    def xval_cv_triples(self, n_local, p, nfolds, weighted, type_measure, args):
        """(count, mean, M2) of the local rows' cross-validation errors per (penalty, lambda): array [npen, nl, 3]"""
        t = np.zeros((args.npen, args.nl, 3))
        L.check(self.lib.oemgpu_xval_cv_triples_dev(self.ctx, n_local, p, nfolds, int(bool(weighted)), int(type_measure), C.byref(args.c),
                                                    t.ctypes.data_as(C.POINTER(C.c_double))))
        return t

    def xval_merge(self, triples, args):
        """triples [nsets, npen, nl, 3] in rank order -> (cvm, cvsd), each [npen, nl]"""
        t = np.ascontiguousarray(triples, dtype=np.float64)
        cvm = np.zeros((args.npen, args.nl)); cvsd = np.zeros((args.npen, args.nl))
        dp = C.POINTER(C.c_double)
        L.check(self.lib.oemgpu_xval_merge(t.ctypes.data_as(dp), int(t.shape[0]), C.byref(args.c), cvm.ctypes.data_as(dp), cvsd.ctypes.data_as(dp)))
        return cvm, cvsd

    def sum_in_order(self, parts, nparts, n, out):
        """out <- parts[0] + parts[1] + ... (nparts buffers of n doubles, contiguous), added in that order by one kernel"""
        L.check(self.lib.oemgpu_sum_in_order_dev(self.ctx, parts.data_ptr(), nparts, n, out.data_ptr()))

    def shift_in_effect(self):
        """did the last solve() find that its sums call for a shift (and read the moments as shifted)?"""
        return self.lib.oemgpu_last_shift_in_effect(self.ctx) == 1

    def shift_advised(self):
        """was the last solve() given moments about 0 of columns with |mean| >> sd (redo about a shift)?"""
        return self.lib.oemgpu_last_shift_advised(self.ctx) == 1


REDUCE = "ordered"         # how buffers are summed over the ranks: "ordered" (default) or "allreduce" (sum_over_ranks)


def sum_over_ranks(backend, dist, group, buf, reduce=None):
    """buf <- the sum of every rank's buf.  "ordered" (the default): ONE all-gather of the buffers and ONE kernel that adds them in
    rank order, ((M_0 + M_1) + M_2) + ... -- the order the in-library multi-GPU path (opts.ngpus, hoststream.hip) adds its devices'
    buffers in.  The result is then a function of the row partition alone: the same bits from every rank, from run to run, on any
    topology, and the same as the in-library form on the same shards (SURVEY section 8e asks for fixed-rank-order summation; an
    all-reduce leaves the order to the collective library's choice of algorithm).  The payload is N (p+2)^2 doubles (c5 at 8 GPUs:
    4.3 MB) -- latency-bound like the all-reduce it replaces.  "allreduce": dist.all_reduce, kept selectable for the day both can be
    timed on real links."""
    mode = reduce or REDUCE
    if mode == "allreduce":
        dist.all_reduce(buf, group=group)
        return
    if mode != "ordered":
        raise ValueError("reduce must be 'ordered' or 'allreduce'")
    world, n = dist.get_world_size(group), buf.numel()
    cache = backend.__dict__.setdefault("_gathered", {})
    gathered = cache.get((world, n))
    if gathered is None:
        gathered = cache[(world, n)] = backend.new_buffer(world * n)
    dist.all_gather_into_tensor(gathered, buf, group=group)
    backend.sum_in_order(gathered, world, n, buf)


def row_partition(n, world):
    """floor(n / world) rows per rank, remainder on the last (mirrors ref src/oem_dense.h:328,343)."""
    base = n // world
    return [(r * base, (r + 1) * base if r + 1 < world else n) for r in range(world)]


def sharded_buffers(backend, p):
    """(sums, moments) device buffers of one solve"""
    return backend.new_buffer(L.sums_len(p)), backend.new_buffer(L.moments_len(p))


SMALL_P_MAX = 288          # csrc/common.hpp: up to here ONE launch walks all penalties side by side on one GPU


def _split_solve(backend, dist, group, mom, sums, p, semantics, standardize, intercept, args, outs):
    """Penalties dealt round-robin to the ranks ("lambda loop"): rank r solves penalties r, r + N, ... from the
    reduced moments every rank holds, then ONE all-gather assembles the result on every rank.  Penalties are independent cold
    starts (ref src/oem_dense.cpp:206-246), so nothing else crosses ranks.  A rank beyond the number of penalties repeats one
    (its copy is dropped): every rank runs a solve and so sees the shift verdict of the reduced moments."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    npen, nl, rows = args.npen, args.nl, p + 1
    mine = list(range(rank, npen, world)) if rank < npen else [rank % npen]
    sub = args.subset(mine)
    backend.solve(mom, sums, p, semantics, standardize, intercept, sub)
    per = nl * rows + 3 * nl                                  # beta | lambda | niter | loss of one penalty
    slots = (npen + world - 1) // world
    buf = backend.new_buffer(slots * per + 1)
    flat = np.zeros(slots * per + 1)
    for s_, k in enumerate(mine if rank < npen else []):
        o = s_ * per
        flat[o:o + nl * rows] = sub.beta[s_].ravel()
        flat[o + nl * rows:o + nl * rows + nl] = sub.lam_out[s_]
        flat[o + nl * rows + nl:o + nl * rows + 2 * nl] = sub.niter[s_]
        flat[o + nl * rows + 2 * nl:o + per] = sub.loss[s_]
    flat[-1] = sub.d.value
    import torch
    buf.copy_(torch.from_numpy(flat))
    parts = [backend.new_buffer(slots * per + 1) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    if outs is None:
        args.outputs(rows)
    elif C.cast(outs[0], C.c_void_p).value != args.beta.ctypes.data:
        # the assembled fit is written in place into args.beta / lam_out / niter / loss / d -- which is what `outs` points at
        # when it came from args.outputs(); foreign result buffers are not supported on this branch
        raise ValueError("split_penalties: `outs` must be the buffers of an earlier args.outputs(p + 1) of the same args")
    for r in range(min(world, npen)):
        got = parts[r].cpu().numpy()
        for s_, k in enumerate(range(r, npen, world)):
            o = s_ * per
            args.beta[k] = got[o:o + nl * rows].reshape(nl, rows)
            args.lam_out[k] = got[o + nl * rows:o + nl * rows + nl]
            args.niter[k] = np.rint(got[o + nl * rows + nl:o + nl * rows + 2 * nl]).astype(np.int32)
            args.loss[k] = got[o + nl * rows + 2 * nl:o + per]
    args.d.value = float(parts[0].cpu().numpy()[-1])


def solve_row_shards(backend, dist, group, x, n_local, ld, p, y, bufs, semantics, standardize, intercept, args, outs=None,
                     split_penalties=None, reduce=None):
    """The local stages and the collective between them (module docstring); call inside backend.section().
    Every rank ends with the full result in `args`.
    split_penalties: None = when it pays (several penalties on the engines that walk them one after the other: p + intercept
    column > SMALL_P_MAX); True / False force it.  reduce: sum_over_ranks' mode (None: the module default, "ordered")."""
    sums, mom = bufs
    many = _many(dist, group)
    q = p + (1 if (semantics != L.OEMGPU_SEM_DENSE and intercept) else 0)
    if split_penalties is None:
        split_penalties = q > SMALL_P_MAX
    split = many and args.npen > 1 and split_penalties

    def solve(sm):
        if split:
            _split_solve(backend, dist, group, mom, sm, p, semantics, standardize, intercept, args, outs)
        else:
            backend.solve(mom, sm, p, semantics, standardize, intercept, args, outs)
    if not many and semantics == L.OEMGPU_SEM_DENSE and hasattr(backend, "fit_dense") and not _STAGED_ALWAYS:
        backend.fit_dense(x, n_local, ld, p, y, standardize, intercept, args, outs)      # one rank: the drop-in entry point does it all
        return
    backend.moments(x, n_local, ld, p, y, None, mom)              # about c = 0: the usual verdict
    if many:
        sum_over_ranks(backend, dist, group, mom, reduce)         # the single Gram exchange of the north star, summed in rank order
    solve(None)
    if backend.shift_advised():                                   # same reduced moments on every rank: all redo or none
        backend.shift_sums(x, n_local, ld, p, y, sums)
        if many:
            sum_over_ranks(backend, dist, group, sums, reduce)
        backend.moments(x, n_local, ld, p, y, sums, mom)
        if many:
            sum_over_ranks(backend, dist, group, mom, reduce)
        solve(sums)


def oem_sharded(x_local, y_local, backend=None, dist=None, group=None, big=False, penalty=None, lambda_=(),
                nlambda=100, lambda_min_ratio=None, alpha=1.0, gamma=3.0, tau=0.5, groups=(), penalty_factor=None,
                group_weights=None, standardize=True, intercept=True, maxit=500, tol=1e-7, accelerate=False,
                compute_loss=False, varnames=None, split_penalties=None, n_total=None, reduce=None):
    """oem() (big=False: DataStd + oemDense semantics) or big.oem() (big=True) on row shards.
    n_total: the global number of rows if the caller knows it (the default lambda.min.ratio depends on n < p, R/oem.R:348-354);
    otherwise one tiny all-reduce of the local row counts finds it.

    x_local: this rank's rows as a column-major device matrix (torch tensor of shape (n_local, p) with
    stride (1, ld)); y_local: its responses.  Every rank returns the full OemFit.
    dist: torch.distributed (already initialised) or None for a single process.
    """
    penalty = _api._match_penalty(penalty)
    n_local, p = x_local.shape
    ld = x_local.stride(1) if hasattr(x_local, "stride") and callable(x_local.stride) else n_local
    if backend is None:
        backend = HipBackend()
    if penalty_factor is None:
        penalty_factor = np.ones(p)
    g, ug, gw = _api._group_setup(penalty, groups, group_weights, p, bool(big and intercept))
    many = _many(dist, group)
    if n_total is None:
        n_total = int(n_local)
        if many and (lambda_min_ratio is None or big):           # the default grid and big.oem's n > p + intercept test need the GLOBAL n
            with backend.section():                              # the fill, the add and the collective on ONE stream
                t = backend.to_device(np.array([float(n_local)]))
                dist.all_reduce(t, group=group)
                n_total = int(round(float(backend.to_host(t)[0])))
    if lambda_min_ratio is None:
        lambda_min_ratio = 0.01 if n_total < p else 0.0001        # R/oem.R:348-354, R/big_oem.R:206-212
    if big and n_total <= p + (1 if intercept else 0):
        raise L.OemgpuError(-4, "p >= n: the XXt branch (ref src/oem_big.h:547-551) is not part of this path")
    _api._common_checks(nlambda, float(lambda_min_ratio), maxit, 1, tol, 0.0)
    args = _api._Args(penalty, _api._lambda_list(lambda_, len(penalty)), int(nlambda), lambda_min_ratio, alpha, gamma,
                      tau, tol, maxit, accelerate and not big, compute_loss, np.asarray(penalty_factor, dtype=np.float64),
                      g, ug, gw)
    with backend.section():                        # kernels and collectives on one stream (see HipBackend)
        bufs = sharded_buffers(backend, p)
        solve_row_shards(backend, dist, group, x_local, n_local, ld, p, y_local, bufs,
                         L.OEMGPU_SEM_BIG if big else L.OEMGPU_SEM_DENSE, standardize, intercept, args,
                         split_penalties=split_penalties, reduce=reduce)
        if many:
            n_total = int(round(float(bufs[1].reshape(p + 2, p + 2)[p + 1, p + 1])))      # the reduced row count
    if varnames is None:
        varnames = [f"V{i + 1}" for i in range(p)]
    return _api._decorate(args, penalty, varnames, True, n_total, p)


def xval_oem_sharded(x_local, y_local, foldid_local, nfolds, backend=None, dist=None, group=None, type_measure="mse", penalty=None,
                     weights_local=None, lambda_=(), nlambda=100, lambda_min_ratio=None, alpha=1.0, gamma=3.0, tau=0.5, groups=(),
                     penalty_factor=None, group_weights=None, standardize=True, intercept=True, maxit=500, tol=1e-7,
                     compute_loss=False, varnames=None):
    """xval.oem() (R/oem_xval.R:107-460, ref src/oem_xval_dense.cpp:31-482) on row shards, one process per GPU.

    The cross-validation is additive the way the fit is: each rank builds the per-fold moment matrices of ITS rows (MFMA Gram
    kernel per fold segment), ONE all-reduce sums the K buffers (K (p+2)^2 doubles; 10 folds at p = 100: 0.8 MB), every rank
    runs the K + 1 fits of the reduced moments (replicated: one launch, a workgroup set per fit), computes the errors of its own
    rows under the fit that left their fold out, and one all-gather of (count, mean, M2) per (penalty, lambda) merges them.
    foldid_local: this rank's fold labels 1..nfolds (every rank passes the same nfolds).  Every rank returns the full result."""
    penalty = _api._match_penalty(penalty)
    n_local, p = x_local.shape
    ld = x_local.stride(1) if hasattr(x_local, "stride") and callable(x_local.stride) else n_local
    if backend is None:
        backend = HipBackend()
    if type_measure in (None, "default", "deviance"):
        type_measure = "mse"
    if type_measure not in ("mse", "mae"):
        raise ValueError("Only 'mse', 'deviance' or 'mae'  available for Gaussian models")
    if int(nfolds) < 3:
        raise ValueError("nfolds must be bigger than 3; nfolds=10 recommended")
    if penalty_factor is None:
        penalty_factor = np.ones(p)
    g, ug, gw = _api._group_setup(penalty, groups, group_weights, p, bool(intercept))
    many = _many(dist, group)
    weighted = weights_local is not None
    K = int(nfolds)
    with backend.section():
        # n decides the default grid (R/oem_xval.R: lambda.min.ratio) and must be known before the options are frozen
        cnt = backend.new_buffer(1)
        cnt += float(n_local)
        if many:
            dist.all_reduce(cnt, group=group)
        n_total = int(round(float(cnt.cpu()[0])))
    if p >= n_total:
        raise ValueError("number of observations must be greater than the number of variables")
    if lambda_min_ratio is None:
        lambda_min_ratio = 0.01 if n_total < p else 0.0001
    _api._common_checks(nlambda, float(lambda_min_ratio), maxit, 1, tol, 0.0)
    args = _api._Args(penalty, _api._lambda_list(lambda_, len(penalty)), int(nlambda), lambda_min_ratio, alpha, gamma, tau, tol, maxit,
                      False, compute_loss, np.asarray(penalty_factor, dtype=np.float64), g, ug, gw)
    tm = 1 if type_measure == "mae" else 0
    with backend.section():
        mom = backend.new_buffer(backend.xval_moments_len(p, K, weighted) + K)      # the fold sizes ride behind the moments
        fold_n = backend.xval_fold_moments(x_local, n_local, ld, p, y_local, weights_local, foldid_local, K, args, mom)
        mom[-K:] = backend.to_device(fold_n.astype(np.float64))
        if many:
            sum_over_ranks(backend, dist, group, mom)             # the one exchange of the Gram stage, the K buffers summed in rank order
        fold_tot = np.rint(backend.to_host(mom[-K:])).astype(np.int64)
        backend.xval_solve_folds(mom, fold_tot, n_local, p, K, weighted, standardize, intercept, args)
        tri = backend.xval_cv_triples(n_local, p, K, weighted, tm, args)
        if many:
            mine = backend.to_device(tri.ravel())
            parts = [backend.new_buffer(tri.size) for _ in range(dist.get_world_size(group))]
            dist.all_gather(parts, mine, group=group)
            tri_all = np.stack([backend.to_host(t).reshape(tri.shape) for t in parts])
        else:
            tri_all = tri[None]
        cvm, cvsd = backend.xval_merge(tri_all, args)
    if varnames is None:
        varnames = [f"V{i + 1}" for i in range(p)]
    res = _api._decorate(args, penalty, varnames, True, n_total, p)
    res["cvm"] = [cvm[k, :1].copy() if name == "ols" else cvm[k].copy() for k, name in enumerate(penalty)]
    res["cvsd"] = [cvsd[k, :1].copy() if name == "ols" else cvsd[k].copy() for k, name in enumerate(penalty)]
    res["name"] = {"mse": "Mean-Squared Error", "mae": "Mean Absolute Error"}[type_measure]
    res.update(_api._getmin([l[:len(c)] for l, c in zip(res["lambda"], res["cvm"])], res["cvm"], res["cvsd"]))
    res["cvup"] = [m + s_ for m, s_ in zip(res["cvm"], res["cvsd"])]
    res["cvlo"] = [m - s_ for m, s_ in zip(res["cvm"], res["cvsd"])]
    res["best.model"] = penalty[res["model.min"] - 1]
    return res
