"""Randomised parity: seeded random shapes, option sets and entry points, GPU against the oracle.  Each case is small enough for
the oracle to finish in well under a second; the point is the combinations (sizes that straddle the kernel dispatch limits,
penalty mixes, flags) that the hand-written cases do not enumerate."""
import warnings

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ELEMENTWISE = ["elastic.net", "lasso", "ols", "mcp", "scad", "mcp.net", "scad.net"]
GROUPED = ["grp.lasso", "grp.lasso.net", "grp.mcp", "grp.scad", "grp.mcp.net", "grp.scad.net", "sparse.grp.lasso"]
import os

SCALE = int(os.environ.get("OEM_FUZZ_SCALE", "1"))        # OEM_FUZZ_SCALE=20: a bug hunt, not part of the regular run
P_CHOICES = [2, 3, 7, 16, 17, 31, 33, 64, 65, 80, 81, 104, 105, 128, 129, 160, 176, 177, 192, 208, 209, 256, 257, 287, 288, 289, 300]


@pytest.fixture(scope="module")
def oa():
    import oem_amd
    return oem_amd


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    p = int(rng.choice(P_CHOICES))
    n = int(p + 1 + rng.integers(1, 4 * p + 200))
    x = np.asfortranarray(rng.normal(size=(n, p)) * rng.uniform(0.5, 3.0) + rng.uniform(-1, 1))
    nnz = int(min(p, rng.integers(1, 8)))
    b = np.zeros(p); b[rng.choice(p, nnz, replace=False)] = rng.uniform(-1.5, 1.5, nnz)
    y = x @ b + rng.normal(size=n) * rng.uniform(0.3, 2.0) + rng.uniform(-1, 1)
    npen = int(rng.integers(1, 4))
    pool = ELEMENTWISE + (GROUPED if rng.random() < 0.6 else [])
    pens = list(rng.choice(pool, npen, replace=False))
    gsz = int(rng.integers(1, 6))
    groups = np.arange(p) // gsz + (0 if rng.random() < 0.3 else 1)          # sometimes a group 0 (unpenalised)
    pf = np.where(rng.random(p) < 0.1, 0.0, rng.uniform(0.5, 2.0, p))
    kw = dict(penalty=pens, nlambda=int(rng.integers(1, 9)), alpha=float(rng.uniform(0.2, 1.0)), gamma=float(rng.uniform(2.1, 5.0)),
              tau=float(rng.uniform(0.1, 0.9)), tol=float(10.0 ** rng.uniform(-10, -6)), maxit=int(rng.choice([30, 200, 500])),
              penalty_factor=pf)
    if any("grp" in q for q in pens):
        kw["groups"] = groups
    return rng, x, y, kw, pens, groups


def _check(f, r, pens, tol=2e-7):
    assert abs(f["d"] - r["d"]) <= 1e-10 * abs(r["d"])
    for k, name in enumerate(pens):
        fb, rb = np.asarray(f["beta"][k]), np.asarray(r["beta"][k])
        assert fb.shape == rb.shape, (name, fb.shape, rb.shape)
        assert np.allclose(f["lambda"][k], r["lambda"][k], rtol=1e-10)
        assert np.abs(fb - rb).max() <= tol * max(1.0, float(np.abs(rb).max())), (name, float(np.abs(fb - rb).max()))
        dn = np.abs(np.ravel(f["niter"][k]).astype(int) - np.ravel(r["niter"][k]).astype(int))
        assert np.mean(dn > 1) <= 0.25, (name, dn)


@pytest.mark.parametrize("seed", list(range(36)) + list(range(10000, 10000 + 36 * (SCALE - 1))))
def test_random_dense(oa, seed):
    rng, x, y, kw, pens, groups = _case(seed)
    std, icpt = bool(rng.integers(2)), bool(rng.integers(2))
    extra = dict(standardize=std, intercept=icpt, accelerate=bool(rng.random() < 0.3), compute_loss=bool(rng.random() < 0.5))
    okw = dict(kw); okw.pop("groups", None)
    if "groups" in kw:
        okw.update(groups=groups, unique_groups=np.unique(groups))
    f = oa.oem(x, y, **kw, **extra)
    r = orc.fit_dense(x, y, lambda_min_ratio=1e-4, **okw, **extra)
    _check(f, r, pens)
    if extra["compute_loss"]:
        for k in range(len(pens)):
            assert np.allclose(np.ravel(f["loss"][k]), np.ravel(r["loss"][k]), rtol=1e-7)


@pytest.mark.parametrize("seed", list(range(36, 54)) + list(range(20000, 20000 + 18 * (SCALE - 1))))
def test_random_big_and_xtx(oa, seed):
    rng, x, y, kw, pens, groups = _case(seed)
    n, p = x.shape
    std, icpt = bool(rng.integers(2)), bool(rng.integers(2))
    okw = dict(kw); okw.pop("groups", None)
    if seed % 2 == 0:                                             # big.oem
        if "groups" in kw:
            g = np.concatenate([[0], groups]) if icpt else groups
            ug = np.unique(np.concatenate([[0], groups])) if icpt else np.unique(groups)
            okw.update(groups=g, unique_groups=ug)
        f = oa.big_oem(x, y, standardize=std, intercept=icpt, **kw)
        r = orc.fit_big(x, y, standardize=std, intercept=icpt, lambda_min_ratio=1e-4, **okw)
    else:                                                         # oem.xtx, with a scale factor half of the time
        sf = rng.uniform(0.5, 2.0, p) if rng.random() < 0.5 else None
        if "groups" in kw:
            okw.update(groups=groups, unique_groups=np.unique(groups))
        xtx, xty = x.T @ x / n, x.T @ y / n
        f = oa.oem_xtx(xtx, xty, scale_factor=() if sf is None else sf, **kw)
        r = orc.fit_xtx(xtx, xty, scale_factor=sf, lambda_min_ratio=1e-4, **okw)
    _check(f, r, pens)


@pytest.mark.parametrize("seed", list(range(54, 66)) + list(range(30000, 30000 + 12 * (SCALE - 1))))
def test_random_xval_and_sparse(oa, seed):
    import scipy.sparse as sp
    rng, x, y, kw, pens, groups = _case(seed)
    n, p = x.shape
    std, icpt = bool(rng.integers(2)), bool(rng.integers(2))
    okw = dict(kw); okw.pop("groups", None)
    if seed % 2 == 0:                                             # xval.oem
        nf = int(rng.integers(3, 8))
        foldid = rng.permutation(np.resize(np.arange(1, nf + 1), n))
        left = n - (n + nf - 1) // nf                              # rows of the fit that leaves the largest fold out
        if left <= p + 1:
            # a fold fit with n <= p (+ intercept): the reference stops ("dimension of x larger than number of observations",
            # ref src/oem_xval_dense.h:690-731, :849-852) and so must the library -- an error code, not a number
            if left <= p:
                with pytest.raises(oa.OemgpuError) as e:
                    oa.xval_oem(x, y, foldid=foldid, standardize=std, intercept=icpt, **kw)
                assert e.value.code == -4
            else:                                                 # exactly p + 1 rows left: served; the fit must be finite
                f = oa.xval_oem(x, y, foldid=foldid, standardize=std, intercept=icpt, **kw)
                assert all(np.all(np.isfinite(b)) for b in f["beta"]) and all(np.all(np.isfinite(c)) for c in f["cvm"])
            return
        if "groups" in kw:
            g = np.concatenate([[0], groups]) if icpt else groups
            ug = np.unique(np.concatenate([[0], groups])) if icpt else np.unique(groups)
            okw.update(groups=g, unique_groups=ug)
        measure = "mae" if rng.random() < 0.5 else "mse"
        f = oa.xval_oem(x, y, foldid=foldid, standardize=std, intercept=icpt, type_measure=measure, **kw)
        r = orc.xval_dense(x, y, foldid, standardize=std, intercept=icpt, type_measure=measure, lambda_min_ratio=1e-4, **okw)
        _check(f, r, pens)
        for k in range(len(pens)):
            assert np.allclose(f["cvm"][k], r["cvm"][k], rtol=1e-6), pens[k]
            assert np.allclose(f["cvsd"][k], r["cvsd"][k], rtol=1e-5), pens[k]
    else:                                                         # sparse x
        xs = sp.csc_matrix(np.where(rng.random(x.shape) < 0.15, x, 0.0))
        if "groups" in kw:
            rg, rug = orc.r_sparse_groups(groups, icpt)
            okw.update(groups=rg, unique_groups=rug)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            f = oa.oem(xs, y, standardize=std, intercept=icpt, **kw)
        r = orc.fit_sparse(xs, y, standardize=std, intercept=icpt, lambda_min_ratio=1e-4, **okw)
        _check(f, r, pens)


@pytest.mark.parametrize("seed", list(range(66, 82)) + list(range(40000, 40000 + 16 * (SCALE - 1))))
def test_random_large_p_wide_and_shifted(oa, seed):
    """the launch-per-iteration engines (p > 288), the p >= n branch, and columns with |mean| >> sd (the shifted redo)"""
    rng = np.random.default_rng(5000 + seed)
    kind = seed % 4
    if kind == 0:                                                 # large p, n > p
        p = int(rng.choice([300, 320, 511, 512, 513, 700])); n = p + int(rng.integers(20, 400))
    elif kind == 1:                                               # wide
        p = int(rng.choice([20, 64, 100, 130, 300])); n = int(rng.integers(3, p + 1))
    elif kind == 2:                                               # far-from-zero columns
        p = int(rng.choice([5, 40, 100, 200])); n = 4 * p + int(rng.integers(50, 500))
    else:                                                         # device-resident, row-major or padded leading dimension
        p = int(rng.choice([10, 100, 150])); n = 3 * p + int(rng.integers(50, 2000))
    x = rng.normal(size=(n, p)) * rng.uniform(0.5, 2.0)
    if kind == 2:
        x += rng.choice([0.0, 50.0, 1e3, -3e4], size=p)
    x = np.asfortranarray(x)
    b = np.zeros(p); b[:min(p, 5)] = rng.uniform(-1, 1, min(p, 5))
    y = x @ b + rng.normal(size=n) + (rng.uniform(-100, 100) if kind == 2 else 0.5)
    pens = list(rng.choice(["lasso", "mcp", "scad", "elastic.net", "grp.lasso"], int(rng.integers(1, 3)), replace=False))
    groups = np.arange(p) // 3 + 1
    std, icpt = bool(rng.integers(2)), bool(rng.integers(2))
    kw = dict(penalty=pens, nlambda=int(rng.integers(2, 7)), alpha=0.8, gamma=3.0, tol=1e-8, maxit=int(rng.choice([50, 300])),
              standardize=std, intercept=icpt)
    okw = dict(kw)
    if "grp.lasso" in pens:
        kw["groups"] = groups; okw.update(groups=groups, unique_groups=np.unique(groups))
    xin = x
    if kind == 3:
        import torch
        if seed % 8 == 3:
            xin = torch.as_tensor(np.ascontiguousarray(x), device="cuda")                       # row-major device matrix
        else:
            buf = torch.zeros((p, n + 6), device="cuda", dtype=torch.float64)                   # column-major, ld = n + 6
            buf[:, :n] = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda")
            xin = buf[:, :n].t()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f = oa.oem(xin, y, **kw)
    r = orc.fit_dense(x, y, lambda_min_ratio=0.01 if n < p else 1e-4, **okw)
    _check(f, r, pens, tol=5e-7)


@pytest.mark.parametrize("seed", list(range(82, 94)) + list(range(50000, 50000 + 12 * (SCALE - 1))))
def test_random_degenerate_inputs(oa, seed):
    """constant and all-zero columns, exact duplicates, a constant response, n = p + 1: whatever the reference's arithmetic makes
    of them (a zero scale becomes 1, src/DataStd.h:237-240), the GPU and the oracle must make the same"""
    rng = np.random.default_rng(9000 + seed)
    p = int(rng.choice([4, 20, 70, 150])); n = int(rng.choice([p + 1, 2 * p + 3, 500 + p]))
    x = rng.normal(size=(n, p))
    x[:, int(rng.integers(p))] = 0.0                               # an all-zero column
    # a constant column.  The value is dyadic: a constant whose column mean is not exact leaves ~1e-17 of "centred" noise in the
    # reference's two-pass arithmetic, which it then divides by its own ~1e-17 "standard deviation" (only an EXACT zero falls back
    # to 1, src/DataStd.h:237-240) -- an accident of rounding that no other summation order reproduces.  (The library treats such
    # a column as the constant it is: gram.hip, Mom::flat.)
    x[:, int(rng.integers(p))] = float(rng.choice([-3.0, -1.5, 0.25, 0.5, 2.0]))
    j, k = rng.choice(p, 2, replace=False); x[:, j] = x[:, k]      # duplicates
    x = np.asfortranarray(x)
    y = np.full(n, 2.5) if seed % 4 == 0 else x[:, :2] @ np.array([1.0, -1.0]) + rng.normal(size=n)
    pens = list(rng.choice(["lasso", "mcp", "elastic.net", "ols"], 2, replace=False))
    std, icpt = bool(rng.integers(2)), bool(rng.integers(2))
    kw = dict(penalty=pens, nlambda=5, alpha=0.7, tol=1e-8, maxit=100, standardize=std, intercept=icpt)
    f = oa.oem(x, y, **kw)
    r = orc.fit_dense(x, y, lambda_min_ratio=1e-4, **kw)
    ok = np.isfinite(r["d"]) and all(np.all(np.isfinite(bk)) for bk in r["beta"]) and all(np.all(np.isfinite(lk)) for lk in r["lambda"])
    if not ok:
        # The reference arithmetic itself blows up (a constant response under standardisation: scale(y) = 0, y / 0, every lambda NaN).
        # What the library returns is pinned all the same: the same d, NaN wherever the reference's lambdas are NaN, iteration
        # counts within one, and coefficients that are the reference's or NaN -- the reference's branchy operators turn a NaN
        # argument into 0 (both comparisons false), the library's branch-free ones (penalty_ops.hpp: shrink) propagate it.
        # (The reference can also return a finite lambda here: when EVERY entry of X'(y/0) is NaN its max scan keeps the initial
        # 0, log(0) = -inf ends the grid and exp(-inf) * 0 = 0 is "lambda_min"; the library scales X'y by 1/0 after the product,
        # sees +-inf, and inf * scale(y) = NaN.  So: NaN where the reference is NaN, the reference's value or NaN elsewhere.)
        if np.isfinite(r["d"]):
            assert abs(f["d"] - r["d"]) <= 1e-9 * abs(r["d"])
        for k in range(len(pens)):
            fl, rl = np.ravel(f["lambda"][k]), np.ravel(r["lambda"][k])
            assert np.all(np.isnan(fl[np.isnan(rl)])) and not np.any(np.isinf(fl)), pens[k]
            both = ~np.isnan(fl) & ~np.isnan(rl)
            assert np.allclose(fl[both], rl[both], rtol=1e-11)
            # (a NaN iterate "converges" at once or one round later, depending on which comparison of the stop rule sees it first)
            assert np.abs(np.ravel(f["niter"][k]).astype(int) - np.ravel(r["niter"][k]).astype(int)).max() <= 1, pens[k]
            fb, rb = np.asarray(f["beta"][k], dtype=float), np.asarray(r["beta"][k], dtype=float)
            both = np.isfinite(fb) & np.isfinite(rb)
            assert np.abs(fb[both] - rb[both]).max(initial=0.0) <= 1e-6
            assert not np.any(np.isinf(fb)) and np.all(np.isnan(fb) | np.isfinite(rb) | np.isnan(rb))
            assert np.all(np.isfinite(fb) | np.isnan(fb))
            # a finite library coefficient where the reference has NaN never happens; NaN where the reference has a number only
            # under a NaN lambda (or the one "ols" solve, which divides the NaN u by d in both)
            lam_nan = np.isnan(rl).any()
            assert lam_nan or not np.any(np.isnan(fb) & np.isfinite(rb)), pens[k]
        return
    _check(f, r, pens, tol=1e-6)


@pytest.mark.parametrize("seed", list(range(94, 106)) + list(range(60000, 60000 + 12 * (SCALE - 1))))
def test_random_shards_and_device_entries(oa, seed):
    """big.oem on ragged row-shard lists, the row-sharded driver with one rank (dense and big semantics) on device data, and a
    device-resident oem.xtx"""
    import torch
    from oem_amd.distributed import oem_sharded
    rng, x, y, kw, pens, groups = _case(seed + 500)
    n, p = x.shape
    std, icpt = bool(rng.integers(2)), bool(rng.integers(2))
    okw = dict(kw); okw.pop("groups", None)
    kind = seed % 3
    if kind == 0:                                                 # shard lists
        cuts = np.sort(rng.choice(np.arange(1, n), size=int(rng.integers(1, 5)), replace=False))
        cuts = np.concatenate([[0], cuts, [n]])
        xs = [np.asfortranarray(x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)]
        ys = [y[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
        if "groups" in kw:
            g = np.concatenate([[0], groups]) if icpt else groups
            okw.update(groups=g, unique_groups=np.unique(np.concatenate([[0], groups])) if icpt else np.unique(groups))
        f = oa.big_oem(xs, ys, standardize=std, intercept=icpt, **kw)
        r = orc.fit_big(x, y, standardize=std, intercept=icpt, lambda_min_ratio=1e-4, **okw)
    elif kind == 1:                                               # one-rank sharded driver on device data
        big = bool(rng.integers(2))
        xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
        yd = torch.as_tensor(y, device="cuda")
        if "groups" in kw:
            if big:
                g = np.concatenate([[0], groups]) if icpt else groups
                okw.update(groups=g, unique_groups=np.unique(np.concatenate([[0], groups])) if icpt else np.unique(groups))
            else:
                okw.update(groups=groups, unique_groups=np.unique(groups))
        f = oem_sharded(xd, yd, big=big, standardize=std, intercept=icpt, **kw)
        r = (orc.fit_big if big else orc.fit_dense)(x, y, standardize=std, intercept=icpt, lambda_min_ratio=1e-4, **okw)
    else:                                                         # device-resident Gram for oem.xtx
        xtx, xty = x.T @ x / n, x.T @ y / n
        if "groups" in kw:
            okw.update(groups=groups, unique_groups=np.unique(groups))
        f = oa.oem_xtx(torch.as_tensor(xtx, device="cuda"), torch.as_tensor(xty, device="cuda"), **kw)
        r = orc.fit_xtx(xtx, xty, lambda_min_ratio=1e-4, **okw)
    _check(f, r, pens)


@pytest.mark.parametrize("seed", list(range(106, 136)) + list(range(70000, 70000 + 30 * (SCALE - 1))))
def test_random_moment_kernels(oa, seed):
    """the moment buffer itself against numpy: row counts around the 64-row slab and chunk boundaries, every tile-count / strip
    variant of the ring kernel (p + 2 around multiples of 16), the shared-slab and block kernels, padded leading dimensions,
    8-byte-aligned (not 16) inputs, shifted and unshifted accumulation, garbage behind the last row"""
    import torch
    from oem_amd import _lib as L
    from oem_amd.distributed import HipBackend
    rng = np.random.default_rng(7000 + seed)
    # (129 ... 420: the deals of the shared-slab kernel's tile columns into rows of eight, a six and a four; 161 ... 192, 225 ... 256: the
    # one-read eight-wave workgroup of gram_wd.hip; 481 ... 512, 750, 1000: its units of sixteen tile columns)
    p = int(rng.choice([2, 13, 14, 15, 29, 30, 31, 46, 62, 63, 78, 94, 100, 105, 106, 109, 110, 111, 112, 126, 129, 145, 161, 177, 200, 209, 225, 226, 240, 241, 254, 255,
                        256, 257, 273, 300, 321, 340, 401, 420, 481, 497, 511, 512, 750, 1000]))
    n = int(rng.choice([1, 63, 64, 65, 127, 128, 129, 191, 1000, 4095, 4096, 4097, 16383, 20001, 65537]))
    pad = int(rng.choice([0, 0, 2, 6, 10]))
    odd = bool(rng.random() < 0.25)                               # leading dimension odd and base 8-byte aligned only
    ld = n + pad + (1 if odd and (n + pad) % 2 == 0 else 0) if odd else (n + 1) // 2 * 2 + pad
    shift = bool(rng.random() < 0.4)
    x = rng.normal(size=(n, p)) * rng.uniform(0.5, 2.0) + (rng.uniform(-50, 50, p) if shift else 0.0)
    y = rng.normal(size=n) + (30.0 if shift else 0.0)
    be = HipBackend(0)
    off = 1 if odd else 0
    flat = torch.full((p * ld + off + 16,), 7.5, device="cuda", dtype=torch.float64)
    view = flat[off:off + p * ld].view(p, ld)
    view[:, :n] = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda")
    yflat = torch.full((n + off + 16,), -3.25, device="cuda", dtype=torch.float64)
    yd = yflat[off:off + n]; yd[:] = torch.as_tensor(y, device="cuda")
    with be.section():
        mom = be.new_buffer(L.moments_len(p)); sums = be.new_buffer(L.sums_len(p))
        if shift:
            be.shift_sums(view[:, :n].t(), n, ld, p, yd, sums)
        be.moments(view[:, :n].t(), n, ld, p, yd, sums if shift else None, mom)
    torch.cuda.synchronize()
    M = mom.cpu().numpy().reshape(p + 2, p + 2)
    c = np.zeros(p + 1)
    if shift:
        sh = sums.cpu().numpy()
        m = sh[:p + 1] / sh[p + 1]
        var = np.maximum(sh[p + 2:2 * p + 3] / sh[p + 1] - m * m, 0.0)
        c = m if np.any(m * m > 256.0 * var) else c
    z = np.column_stack([x - c[:p], y - c[p], np.ones(n)])
    ref = z.T @ z
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref))) + 1e-300
    assert np.abs(M - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), (p, n, ld, odd, shift, float(np.abs((M - ref) / scale).max()))


@pytest.mark.parametrize("seed", list(range(136, 148)) + list(range(80000, 80000 + 12 * (SCALE - 1))))
def test_random_xval_fold_layouts(oa, seed):
    """xval.oem with many folds of very uneven size, folds of a single row, ids that never occur, n at the 1024-row layout-block
    boundaries"""
    rng = np.random.default_rng(8000 + seed)
    p = int(rng.choice([3, 10, 40, 100])); n = int(rng.choice([1023, 1024, 1025, 2047, 2049, 3000, 5000])) + 3 * p
    nf = int(rng.choice([3, 7, 16, 37, 64, 130]))
    w = rng.dirichlet(np.full(nf, 0.3))                           # very uneven fold sizes, some (nearly) empty
    foldid = rng.choice(np.arange(1, nf + 1), size=n, p=w)
    foldid[:1] = nf                                               # nfolds = max(foldid) as R/oem_xval.R:189 has it
    x = np.asfortranarray(rng.normal(size=(n, p)) + rng.uniform(-0.5, 0.5))
    y = x[:, :2] @ np.array([1.0, -0.5]) + rng.normal(size=n) + 0.3
    while n - np.bincount(foldid, minlength=nf + 1).max() <= p + 2:   # a fold fit would have n <= p: thin the dominant fold
        big = np.flatnonzero(foldid == np.bincount(foldid, minlength=nf + 1).argmax())
        foldid[big[::2]] = rng.integers(1, nf + 1, size=big[::2].size)
        foldid[:1] = nf
    pens = list(rng.choice(["lasso", "mcp", "elastic.net", "ols"], 2, replace=False))
    std, icpt = bool(rng.integers(2)), bool(rng.integers(2))
    measure = "mae" if rng.random() < 0.5 else "mse"
    kw = dict(penalty=pens, nlambda=int(rng.integers(2, 8)), alpha=0.6, tol=1e-8, maxit=400, standardize=std, intercept=icpt)
    f = oa.xval_oem(x, y, foldid=foldid, type_measure=measure, **kw)
    r = orc.xval_dense(x, y, foldid, type_measure=measure, lambda_min_ratio=1e-4, **kw)
    _check(f, r, pens)
    for k in range(len(pens)):
        assert np.allclose(f["cvm"][k], r["cvm"][k], rtol=1e-7), pens[k]
        assert np.allclose(f["cvsd"][k], r["cvsd"][k], rtol=1e-6), pens[k]


@pytest.mark.parametrize("seed", list(range(148, 160)) + list(range(90000, 90000 + 12 * (SCALE - 1))))
def test_random_large_q_engines(oa, seed):
    """the launch-per-iteration engines through oem.xtx: q on both sides of 512 / 1024 (the fused kernels' exact sizes and their
    prefetch limit), group / element-wise / mixed penalties, scale.factor, and the dense entry's accelerate + compute.loss"""
    rng = np.random.default_rng(9500 + seed)
    q = int(rng.choice([289, 300, 511, 512, 513, 700, 1023, 1024, 1025, 1100, 1500]))
    n = q + int(rng.integers(50, 300))
    x = rng.normal(size=(n, q)) * rng.uniform(0.5, 2.0)
    b = np.zeros(q); b[:6] = rng.uniform(-1, 1, 6)
    y = x @ b + rng.normal(size=n)
    pens = list(rng.choice(["lasso", "mcp", "scad.net", "grp.lasso", "grp.mcp", "sparse.grp.lasso", "ols"], int(rng.integers(1, 3)), replace=False))
    groups = np.arange(q) // int(rng.integers(2, 9)) + 1
    kw = dict(penalty=pens, nlambda=int(rng.integers(2, 6)), alpha=0.7, gamma=3.0, tau=0.4, tol=1e-7, maxit=int(rng.choice([40, 150])))
    okw = dict(kw)
    if any("grp" in t for t in pens):
        kw["groups"] = groups; okw.update(groups=groups, unique_groups=np.unique(groups))
    if seed % 2 == 0:
        xtx, xty = x.T @ x / n, x.T @ y / n
        sf = rng.uniform(0.5, 2.0, q) if rng.random() < 0.5 else None
        f = oa.oem_xtx(xtx, xty, scale_factor=() if sf is None else sf, **kw)
        r = orc.fit_xtx(xtx, xty, scale_factor=sf, lambda_min_ratio=1e-4, d_override=f["d"] if q > 1100 else 0.0, **okw)
    else:
        extra = dict(standardize=bool(rng.integers(2)), intercept=bool(rng.integers(2)), accelerate=bool(rng.random() < 0.5),
                     compute_loss=bool(rng.random() < 0.5))
        xf = np.asfortranarray(x)
        f = oa.oem(xf, y, **kw, **extra)
        r = orc.fit_dense(xf, y, lambda_min_ratio=1e-4, d_override=f["d"] if q > 1100 else 0.0, **okw, **extra)
        if extra["compute_loss"]:
            for k in range(len(pens)):
                assert np.allclose(np.ravel(f["loss"][k]), np.ravel(r["loss"][k]), rtol=1e-7)
    _check(f, r, pens, tol=5e-7)


@pytest.mark.parametrize("seed", list(range(130, 148)) + list(range(80000, 80000 + 18 * (SCALE - 1))))
def test_random_wide(oa, seed, monkeypatch):
    """p >= n through the wide engine (the reference's own two-product iteration, OEM_WIDE=1 at every size it can run): random
    shapes with n <= p (column heights across the kernel's register sizes), DataStd flags, penalty mixes with and without group
    operators, accelerate, compute.loss, penalty factors -- against the oracle's restatement of that branch."""
    monkeypatch.setenv("OEM_WIDE", "1")
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([2, 5, 40, 64, 65, 130, 200, 257, 400, 700]))
    p = int(n + rng.integers(0, 3 * n + 60))
    if p < 2:
        p = 2
    x = np.asfortranarray(rng.normal(size=(n, p)) * rng.uniform(0.5, 3.0) + rng.uniform(-1, 1))
    nnz = int(min(p, rng.integers(1, 6)))
    b = np.zeros(p); b[rng.choice(p, nnz, replace=False)] = rng.uniform(-1.5, 1.5, nnz)
    y = x @ b + rng.normal(size=n) * rng.uniform(0.3, 2.0) + rng.uniform(-1, 1)
    npen = int(rng.integers(1, 4))
    pool = ELEMENTWISE + (GROUPED if rng.random() < 0.4 else [])
    pens = list(rng.choice(pool, npen, replace=False))
    groups = np.arange(p) // int(rng.integers(1, 6)) + (0 if rng.random() < 0.3 else 1)
    pf = np.where(rng.random(p) < 0.1, 0.0, rng.uniform(0.5, 2.0, p))
    kw = dict(penalty=pens, nlambda=int(rng.integers(1, 7)), alpha=float(rng.uniform(0.2, 1.0)), gamma=float(rng.uniform(2.1, 5.0)),
              tau=float(rng.uniform(0.1, 0.9)), tol=float(10.0 ** rng.uniform(-9, -6)), maxit=int(rng.choice([30, 200, 400])),
              penalty_factor=pf, standardize=bool(rng.integers(2)), intercept=bool(rng.integers(2)),
              accelerate=bool(rng.random() < 0.25), compute_loss=bool(rng.random() < 0.4))
    okw = dict(kw)
    if any("grp" in q for q in pens):
        kw["groups"] = groups
        okw.update(groups=groups, unique_groups=np.unique(groups))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f = oa.oem(x, y, **kw)
    r = orc.fit_dense(x, y, lambda_min_ratio=0.01 if n < p else 1e-4, **okw)
    _check(f, r, pens, tol=5e-7)
    if kw["compute_loss"]:
        for k in range(len(pens)):
            assert np.allclose(np.ravel(f["loss"][k]), np.ravel(r["loss"][k]), rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("seed", list(range(160, 176)) + list(range(95000, 95000 + 16 * (SCALE - 1))))
def test_random_wide_cooperating(oa, seed, monkeypatch):
    """p >= n on the persistent cooperating-workgroup form of the wide engine (path_wcoop.hip): random shapes over every column height
    and workgroup count, ragged all-reduce slices, DataStd flags, penalty factors, one to four penalties (side by side in workgroup
    sets), group operators with contiguous and scattered groups, Nesterov's step, compute.loss -- against the oracle's restatement of
    the branch and against the launch-per-iteration engine."""
    monkeypatch.setenv("OEM_WIDE", "1")
    rng = np.random.default_rng(7700 + seed)
    n = int(rng.choice([1, 2, 7, 33, 64, 65, 100, 129, 192, 193, 250, 300, 385, 450, 520, 769, 900, 1024]))
    cap = {1: 16, 2: 16, 3: 16, 4: 16, 6: 8, 8: 8, 12: 4, 16: 4}[[v for v in (1, 2, 3, 4, 6, 8, 12, 16) if 64 * v >= n][0]] * 4 * 192
    p = int(min(cap, n + rng.integers(0, max(2, min(4 * n + 200, 400_000 // max(n, 1))))))
    p = max(p, 2)
    x = np.asfortranarray(rng.normal(size=(n, p)) * rng.uniform(0.5, 3.0) + rng.uniform(-1, 1))
    nnz = int(min(p, rng.integers(1, 6)))
    b = np.zeros(p); b[rng.choice(p, nnz, replace=False)] = rng.uniform(-1.5, 1.5, nnz)
    y = x @ b + rng.normal(size=n) * rng.uniform(0.3, 2.0) + rng.uniform(-1, 1)
    grouped = rng.random() < 0.45 and p <= 8000                  # group operators: one more exchange, of the members of a workgroup's own groups
    pens = list(rng.choice(ELEMENTWISE + (GROUPED if grouped else []), int(rng.integers(1, 5)), replace=False))
    pf = np.where(rng.random(p) < 0.1, 0.0, rng.uniform(0.5, 2.0, p))
    kw = dict(penalty=pens, nlambda=int(rng.integers(1, 7)), alpha=float(rng.uniform(0.2, 1.0)), gamma=float(rng.uniform(2.1, 5.0)),
              tau=float(rng.uniform(0.1, 0.9)), tol=float(10.0 ** rng.uniform(-9, -6)), maxit=int(rng.choice([30, 200, 400])), penalty_factor=pf,
              standardize=bool(rng.integers(2)), intercept=bool(rng.integers(2)), compute_loss=bool(rng.random() < 0.4),
              accelerate=bool(rng.random() < 0.3))
    okw = dict(kw)
    if any("grp" in q for q in pens):
        gs = int(rng.choice([1, 2, 3, 5, 8, 13, 40]))
        groups = np.arange(p) // gs + (0 if rng.random() < 0.3 else 1)            # (ids from 0: group 0 is unpenalised)
        if rng.random() < 0.4:
            groups = groups[rng.permutation(p)]                     # members scattered over the workgroups
        kw["groups"] = groups
        okw.update(groups=groups, unique_groups=np.unique(groups))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f = oa.oem(x, y, **kw)
        monkeypatch.setenv("OEM_NO_WCOOP", "1")
        g = oa.oem(x, y, **kw)
    r = orc.fit_dense(x, y, lambda_min_ratio=0.01 if n < p else 1e-4, **okw)
    ok = np.isfinite(r["d"]) and r["d"] > 0 and all(np.all(np.isfinite(bk)) for bk in r["beta"]) and all(np.all(np.isfinite(lk)) for lk in r["lambda"])
    if not ok:                                                    # (n = 1 under standardisation: the reference divides by a zero scale; n = 1
                                                                  # with an intercept: every centred column is zero, d = 0 and u / d is 0 / 0 --
                                                                  # the reference's branchy operators turn that into 0, the library's propagate it)
        assert not np.any(np.isinf(np.concatenate([np.ravel(bk) for bk in f["beta"]])))
        return
    _check(f, r, pens, tol=5e-7)
    assert abs(f["d"] - g["d"]) <= 1e-10 * abs(g["d"])           # (two Lanczos recurrences with their own start vectors and looks)
    for k in range(len(pens)):
        scale = max(1.0, float(np.abs(np.asarray(g["beta"][k])).max()))
        assert np.abs(np.asarray(f["beta"][k]) - np.asarray(g["beta"][k])).max() <= 1e-8 * scale, pens[k]
        assert np.abs(np.ravel(f["niter"][k]).astype(int) - np.ravel(g["niter"][k]).astype(int)).max() <= 1, pens[k]
        if kw["compute_loss"]:
            assert np.allclose(np.ravel(f["loss"][k]), np.ravel(r["loss"][k]), rtol=1e-7, atol=1e-9), pens[k]


@pytest.mark.parametrize("seed", list(range(180, 192)) + list(range(97000, 97000 + 12 * (SCALE - 1))))
def test_random_wide_streamed(oa, seed, monkeypatch):
    """p >= n on the STREAMED persistent form (path_wcoop.hip: path_wstream_kernel), forced onto random shapes (OEM_WSTREAM=1 with
    OEM_NO_WCOOP=1): every column height it is built for, one chunk and several per wave, ragged last chunks, DataStd flags, penalty
    factors, one to four element-wise penalties, compute.loss -- against the oracle's restatement of the branch and against the
    launch-per-iteration engine."""
    monkeypatch.setenv("OEM_WIDE", "1"); monkeypatch.setenv("OEM_WSTREAM", "1"); monkeypatch.setenv("OEM_NO_WCOOP", "1")
    rng = np.random.default_rng(7900 + seed)
    n = int(rng.choice([1, 2, 9, 40, 64, 65, 120, 129, 200, 257, 320, 385, 500, 512]))
    p = int(max(2, n + rng.integers(0, max(2, min(6 * n + 300, 600_000 // max(n, 1))))))
    if rng.random() < 0.2:
        p = int(rng.choice([12288, 12289, 13000]))                # more than one chunk per wave at the widest tile
        n = int(rng.choice([20, 64, 100]))
    x = np.asfortranarray(rng.normal(size=(n, p)) * rng.uniform(0.5, 3.0) + rng.uniform(-1, 1))
    nnz = int(min(p, rng.integers(1, 6)))
    b = np.zeros(p); b[rng.choice(p, nnz, replace=False)] = rng.uniform(-1.5, 1.5, nnz)
    y = x @ b + rng.normal(size=n) * rng.uniform(0.3, 2.0) + rng.uniform(-1, 1)
    pens = list(rng.choice(ELEMENTWISE, int(rng.integers(1, 5)), replace=False))
    pf = np.where(rng.random(p) < 0.1, 0.0, rng.uniform(0.5, 2.0, p))
    kw = dict(penalty=pens, nlambda=int(rng.integers(1, 6)), alpha=float(rng.uniform(0.2, 1.0)), gamma=float(rng.uniform(2.1, 5.0)),
              tol=float(10.0 ** rng.uniform(-9, -6)), maxit=int(rng.choice([30, 200, 400])), penalty_factor=pf,
              standardize=bool(rng.integers(2)), intercept=bool(rng.integers(2)), compute_loss=bool(rng.random() < 0.4))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f = oa.oem(x, y, **kw)
        monkeypatch.setenv("OEM_NO_WSTREAM", "1")
        g = oa.oem(x, y, **kw)
    r = orc.fit_dense(x, y, lambda_min_ratio=0.01 if n < p else 1e-4, **kw)
    ok = np.isfinite(r["d"]) and r["d"] > 0 and all(np.all(np.isfinite(bk)) for bk in r["beta"]) and all(np.all(np.isfinite(lk)) for lk in r["lambda"])
    if not ok:                                                    # (the blow-up regimes of test_random_wide_cooperating)
        assert not np.any(np.isinf(np.concatenate([np.ravel(bk) for bk in f["beta"]])))
        return
    _check(f, r, pens, tol=5e-7)
    assert abs(f["d"] - g["d"]) <= 1e-10 * abs(g["d"])
    for k in range(len(pens)):
        scale = max(1.0, float(np.abs(np.asarray(g["beta"][k])).max()))
        assert np.abs(np.asarray(f["beta"][k]) - np.asarray(g["beta"][k])).max() <= 1e-8 * scale, pens[k]
        assert np.abs(np.ravel(f["niter"][k]).astype(int) - np.ravel(g["niter"][k]).astype(int)).max() <= 1, pens[k]
        if kw["compute_loss"]:
            assert np.allclose(np.ravel(f["loss"][k]), np.ravel(r["loss"][k]), rtol=1e-7, atol=1e-9), pens[k]


@pytest.mark.parametrize("seed", list(range(200, 212)) + list(range(98000, 98000 + 12 * (SCALE - 1))))
def test_random_wide_resident_in_the_accumulator_file(oa, seed, monkeypatch):
    """p >= n with column sets of every wave in the accumulator file too (path_wcoop.hip: path_wres_kernel, round 4), forced onto random
    shapes (OEM_WRES=1: also where the vector registers alone would do): every column height it is built for (3 .. 9 column sets per
    wave), ragged last sets / waves / workgroups, DataStd flags, penalty factors, one to four element-wise penalties, compute.loss --
    against the oracle's restatement of the branch and against the launch-per-iteration engine."""
    monkeypatch.setenv("OEM_WIDE", "1"); monkeypatch.setenv("OEM_WRES", "1")
    rng = np.random.default_rng(8100 + seed)
    n = int(rng.choice([1, 2, 9, 40, 64, 65, 120, 128, 129, 192, 200, 256, 257, 320, 384, 385, 500, 512, 513, 700, 768, 769, 900, 1024]))
    p = int(max(2, n + rng.integers(0, max(2, min(8 * n + 300, 700_000 // max(n, 1))))))
    if rng.random() < 0.25:
        n = int(rng.choice([20, 64, 100, 130])); p = int(rng.choice([9000, 12289, 20000]))       # many workgroups, short columns
    x = np.asfortranarray(rng.normal(size=(n, p)) * rng.uniform(0.5, 3.0) + rng.uniform(-1, 1))
    nnz = int(min(p, rng.integers(1, 6)))
    b = np.zeros(p); b[rng.choice(p, nnz, replace=False)] = rng.uniform(-1.5, 1.5, nnz)
    y = x @ b + rng.normal(size=n) * rng.uniform(0.3, 2.0) + rng.uniform(-1, 1)
    pens = list(rng.choice(ELEMENTWISE, int(rng.integers(1, 5)), replace=False))
    pf = np.where(rng.random(p) < 0.1, 0.0, rng.uniform(0.5, 2.0, p))
    kw = dict(penalty=pens, nlambda=int(rng.integers(1, 6)), alpha=float(rng.uniform(0.2, 1.0)), gamma=float(rng.uniform(2.1, 5.0)),
              tol=float(10.0 ** rng.uniform(-9, -6)), maxit=int(rng.choice([30, 200, 400])), penalty_factor=pf,
              standardize=bool(rng.integers(2)), intercept=bool(rng.integers(2)), compute_loss=bool(rng.random() < 0.4))
    import torch
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f = oa.oem(xd, y, **kw)
        assert oa.last_path_engine()[0] == "wres"
        monkeypatch.setenv("OEM_NO_WCOOP", "1"); monkeypatch.setenv("OEM_NO_WSTREAM", "1")       # (the first also switches path_wres_kernel off)
        g = oa.oem(xd, y, **kw)
        assert oa.last_path_engine()[0] == "wlaunches"
    r = orc.fit_dense(x, y, lambda_min_ratio=0.01 if n < p else 1e-4, **kw)
    ok = np.isfinite(r["d"]) and r["d"] > 0 and all(np.all(np.isfinite(bk)) for bk in r["beta"]) and all(np.all(np.isfinite(lk)) for lk in r["lambda"])
    if not ok:                                                    # (the blow-up regimes of test_random_wide_cooperating)
        assert not np.any(np.isinf(np.concatenate([np.ravel(bk) for bk in f["beta"]])))
        return
    _check(f, r, pens, tol=5e-7)
    assert abs(f["d"] - g["d"]) <= 1e-10 * abs(g["d"])
    for k in range(len(pens)):
        scale = max(1.0, float(np.abs(np.asarray(g["beta"][k])).max()))
        assert np.abs(np.asarray(f["beta"][k]) - np.asarray(g["beta"][k])).max() <= 1e-8 * scale, pens[k]
        assert np.abs(np.ravel(f["niter"][k]).astype(int) - np.ravel(g["niter"][k]).astype(int)).max() <= 1, pens[k]
        if kw["compute_loss"]:
            assert np.allclose(np.ravel(f["loss"][k]), np.ravel(r["loss"][k]), rtol=1e-7, atol=1e-9), pens[k]


@pytest.mark.parametrize("seed", list(range(150, 152)) + list(range(90000, 90000 + 2 * (SCALE - 1))))
def test_random_packed_triangle_engine(oa, seed):
    """oem.xtx at a random 4096 < p <= 4500 (the first sizes beyond the register engines: a ragged last block, odd sizes with rows that are
    only 8-byte aligned) on the packed lower triangle (path_large.hip: sympk_*): random element-wise penalty mixes -- the (head, product)
    pairs -- or group penalties in the call -- their operators in the same head, or product, slot sum, update kernel for large groups --, penalty factors and maxit, against the oracle
    (d handed over, as in the config-4 tests: the comparison is the path's)"""
    from scipy.sparse.linalg import eigsh
    rng = np.random.default_rng(9900 + seed)
    p = int(rng.integers(4097, 4500)); n = p + int(rng.integers(100, 3000))
    x = rng.normal(size=(n, p)) * (1.0 + rng.uniform(size=p))
    b = np.zeros(p); b[rng.choice(p, 12, replace=False)] = rng.uniform(-1, 1, 12)
    y = x @ b + rng.normal(size=n)
    xtx, xty = x.T @ x / n, x.T @ y / n
    pens = list(rng.choice(ELEMENTWISE, int(rng.integers(1, 4)), replace=False))
    pf = np.where(rng.random(p) < 0.05, 0.0, rng.uniform(0.5, 2.0, p))
    kw = dict(penalty=pens, nlambda=int(rng.integers(2, 5)), alpha=float(rng.uniform(0.3, 1.0)), gamma=float(rng.uniform(2.5, 5.0)),
              tol=float(10.0 ** rng.uniform(-9, -7)), maxit=int(rng.choice([60, 300])), penalty_factor=pf)
    if seed % 2:
        # a group penalty beside them: groups of <= hi members dealt at random over the coordinates.  hi <= 32: reordered into runs, the group
        # operators in the head of the pairs (sympk_head_kernel<1>); 50, 96: the wider windows (<2>, <3>); 130: as they are, on the update-kernel form
        hi = int(rng.choice([6, 32, 50, 96, 130]))
        sizes = []
        while sum(sizes) < p:
            sizes.append(int(rng.integers(1, hi + 1)))
        sizes[-1] -= sum(sizes) - p
        kw["penalty"] = pens = pens[:1] + list(rng.choice(GROUPED, int(rng.integers(1, 3)), replace=False))
        kw["groups"] = rng.permutation(np.repeat(np.arange(len(sizes)), sizes))      # label 0: unpenalised
        kw["group_weights"] = rng.uniform(0.5, 2.0, len(sizes)); kw["tau"] = float(rng.uniform(0.2, 0.8))
    import torch
    f = oa.oem_xtx(torch.as_tensor(xtx, device="cuda"), xty, **kw)
    assert oa.last_path_engine()[0] == "launches"
    r = orc.fit_xtx(xtx, xty, d_override=f["d"], lambda_min_ratio=1e-4, unique_groups=np.unique(kw["groups"]) if "groups" in kw else None, **kw)
    lam_max = float(eigsh(xtx, k=1, which="LA", tol=0, ncv=24, return_eigenvectors=False)[0])
    assert abs(f["d"] - 1.005 * lam_max) <= 1e-10 * lam_max
    _check(f, r, pens, tol=5e-7)


@pytest.mark.parametrize("seed", list(range(160, 166)) + list(range(95000, 95000 + 6 * (SCALE - 1))))
def test_random_register_resident_symmetric_engine(oa, seed):
    """oem.xtx at a random 1024 < p <= 4096 on the register-resident symmetric engine (path_symcoop.hip; every tile count per wave,
    ragged last tiles): random element-wise penalty mixes, penalty factors and maxit, against the oracle (d handed over)"""
    rng = np.random.default_rng(9950 + seed)
    p = int(rng.choice([int(rng.integers(1025, 1400)), int(rng.integers(1400, 2433)), int(rng.integers(2433, 3457)), int(rng.integers(3457, 4097))]))
    n = p + int(rng.integers(100, 2000))
    x = rng.normal(size=(n, p)) * (1.0 + rng.uniform(size=p))
    b = np.zeros(p); b[rng.choice(p, 12, replace=False)] = rng.uniform(-1, 1, 12)
    y = x @ b + rng.normal(size=n)
    xtx, xty = x.T @ x / n, x.T @ y / n
    pens = list(rng.choice(ELEMENTWISE, int(rng.integers(1, 4)), replace=False))
    pf = np.where(rng.random(p) < 0.05, 0.0, rng.uniform(0.5, 2.0, p))
    kw = dict(penalty=pens, nlambda=int(rng.integers(2, 5)), alpha=float(rng.uniform(0.3, 1.0)), gamma=float(rng.uniform(2.5, 5.0)),
              tol=float(10.0 ** rng.uniform(-9, -7)), maxit=int(rng.choice([60, 300])), penalty_factor=pf)
    f = oa.oem_xtx(xtx, xty, **kw)
    r = orc.fit_xtx(xtx, xty, d_override=f["d"], lambda_min_ratio=1e-4, **kw)
    lam_max = np.linalg.eigvalsh(xtx)[-1]
    assert abs(f["d"] - 1.005 * lam_max) <= 1e-10 * lam_max
    _check(f, r, pens, tol=5e-7)



@pytest.mark.parametrize("seed", list(range(170, 173)) + list(range(97000, 97000 + 3 * (SCALE - 1))))
def test_random_scattered_groups_on_the_register_resident_engine(oa, seed):
    """oem.xtx at a random 1024 < p <= 4096 with group penalties whose groups are NOT runs of neighbouring coordinates (sizes 1-400 in a
    random layout, group 0 unpenalised, weights, penalty factors, now and then `scale.factor`): reordered into runs (api.hip:
    group_run_permutation), solved on path_symcoop_kernel<.., GEN>, put back -- against the oracle (d handed over), also where there are
    fewer runs than workgroups (some owners own nothing)."""
    import torch
    rng = np.random.default_rng(9970 + seed)
    p = int(rng.choice([int(rng.integers(1025, 1600)), int(rng.integers(1600, 2600)), int(rng.integers(2600, 4097))]))
    n = p + int(rng.integers(100, 1500))
    x = rng.normal(size=(n, p)) * (1.0 + rng.uniform(size=p))
    b = np.zeros(p); b[rng.choice(p, 12, replace=False)] = rng.uniform(-1, 1, 12)
    y = x @ b + rng.normal(size=n)
    xtx, xty = x.T @ x / n, x.T @ y / n
    hi = int(rng.choice([4, 12, 32, 90, 400]))                           # (32: few large groups -- fewer runs than workgroups; 90, 400: groups in several owners' slices)
    sizes = []
    while sum(sizes) < p:
        sizes.append(int(rng.integers(max(1, hi // 2), hi + 1)))
    sizes[-1] -= sum(sizes) - p
    groups = rng.permutation(np.repeat(np.arange(len(sizes)), sizes))   # group 0 exists: unpenalised (ref src/oem_dense.h:207)
    gw = rng.uniform(0.5, 2.0, len(sizes))
    pens = list(rng.choice(GROUPED, int(rng.integers(1, 3)), replace=False)) + list(rng.choice(ELEMENTWISE, int(rng.integers(0, 2)), replace=False))
    pf = np.where(rng.random(p) < 0.05, 0.0, rng.uniform(0.5, 2.0, p))
    kw = dict(penalty=pens, groups=groups, group_weights=gw, nlambda=int(rng.integers(2, 5)), alpha=float(rng.uniform(0.3, 1.0)), gamma=float(rng.uniform(2.5, 5.0)),
              tau=float(rng.uniform(0.2, 0.8)), tol=float(10.0 ** rng.uniform(-9, -7)), maxit=int(rng.choice([60, 300])), penalty_factor=pf)
    if seed % 3 == 0:
        kw["scale_factor"] = rng.uniform(0.5, 2.0, p)
    f = oa.oem_xtx(torch.as_tensor(xtx, device="cuda"), xty, **kw)
    # (the register engine wherever the runs can be dealt to its owners -- <= 32 coordinates each, whole runs: groups of 17-32 members need an
    #  owner each, and more of them than the engine has workgroups go to the launch engines: the same answer either way)
    assert oa.last_path_engine()[0] in ("symcoop", "launches")
    if hi <= 12 or hi > 32:
        assert oa.last_path_engine()[0] == "symcoop"
    r = orc.fit_xtx(xtx, xty, d_override=f["d"], lambda_min_ratio=1e-4, unique_groups=np.unique(groups), **kw)
    _check(f, r, pens, tol=5e-7)
