"""Size bands no test had run (VERDICT r4 "next round" item 1): every engine at the sizes it exists for, against the oracle.

* the Gram form beyond q = 4096 -- the launch-per-iteration engines of path_large.hip at 4,097 / 5,000 / 8,192 (`oem.xtx`) and
  `oem()` with n > p at p = 5,000 (ref src/oem_xtx.h:347-381, src/oem_dense.h:508-524);
* `path_wres_kernel` where it was built for: 500 x 20,000 and 200 x 30,000 with the DEFAULT selection (no OEM_WRES);
* config 4's only remaining production case of the launch engines at that size: `scale.factor` at p = 4,096;
* `xval.oem` between p = 300 and the old limit of the CV-error kernel, and beyond it (ref src/oem_xval_dense.cpp:343-461).

d = 1.005 lambda_max is held against ARPACK / LAPACK to 1e-10 and then handed to the oracle (as the config-4 tests do): what is
compared is the path -- coefficients to 1e-9, iteration counts within one."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc

DTOL = 1e-10


@pytest.fixture(scope="module")
def oa():
    import torch
    assert torch.cuda.is_available()
    import oem_amd
    oem_amd.lib()
    return oem_amd


def _lam_max(a):
    """largest eigenvalue of a symmetric matrix to machine precision (ARPACK, tol = 0) -- LAPACK's full solve takes minutes at 8,192"""
    from scipy.sparse.linalg import eigsh
    return float(eigsh(a, k=1, which="LA", tol=0, ncv=24, return_eigenvectors=False)[0])


def _gram_problem(p, n, seed, nnz=25):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, p))
    b = np.zeros(p); b[rng.choice(p, nnz, replace=False)] = rng.uniform(-1, 1, nnz)
    y = x @ b + rng.normal(size=n)
    return x.T @ x / n, x.T @ y / n


def _same_path(f, r, k, label, tol=1e-9):
    fb, rb = np.asarray(f["beta"][k]), np.asarray(r["beta"][k])
    assert fb.shape == rb.shape, label
    assert np.allclose(f["lambda"][k], r["lambda"][k], rtol=1e-12, atol=0), label
    scale = max(1.0, float(np.abs(rb).max()))
    assert np.abs(fb - rb).max() <= tol * scale, (label, np.abs(fb - rb).max())
    dn = np.abs(np.ravel(f["niter"][k]).astype(int) - np.ravel(r["niter"][k]).astype(int))
    assert dn.max() <= 1, (label, dn)


@pytest.mark.parametrize("p", [4097, 5000, 8192])
def test_xtx_beyond_4096_runs_the_launch_engines(oa, p):
    """oem.xtx at q > 4096: no register file holds the matrix any more (p = 8,192: 512 MB, beyond the Infinity Cache), so the
    launch-per-iteration engines serve it -- lasso + a group penalty in one call; at 5,000 also `scale.factor`."""
    import torch
    xtx, xty = _gram_problem(p, p + p // 2, 4000 + p)
    xd = torch.as_tensor(xtx, device="cuda")
    groups = np.arange(p) // 5 + 1
    kw = dict(penalty=["lasso", "grp.lasso"], nlambda=4, lambda_min_ratio=0.05, tol=1e-9, maxit=400)
    f = oa.oem_xtx(xd, xty, groups=groups, **kw)
    assert oa.last_path_engine()[0] == "launches"
    lmax = _lam_max(xtx)
    assert abs(f["d"] - 1.005 * lmax) <= DTOL * lmax, (f["d"], 1.005 * lmax)
    r = orc.fit_xtx(xtx, xty, native=True, groups=groups, unique_groups=np.unique(groups), d_override=f["d"], **kw)
    for k in range(2):
        _same_path(f, r, k, (p, kw["penalty"][k]))
        assert (np.asarray(f["beta"][k])[:, -1] != 0).sum() >= 10
    if p == 5000:
        sf = np.linspace(0.5, 2.0, p)
        kw = dict(penalty=["lasso", "mcp"], nlambda=3, lambda_min_ratio=0.1, tol=1e-9, maxit=400)
        f = oa.oem_xtx(xd, xty, scale_factor=sf, **kw)
        assert oa.last_path_engine()[0] == "launches"
        s = xtx / sf[:, None] / sf[None, :]                      # ref src/oem_xtx.h:349-356: S^-1 XX S^-1
        lmax = _lam_max(s)
        assert abs(f["d"] - 1.005 * lmax) <= DTOL * lmax
        r = orc.fit_xtx(xtx, xty, native=True, scale_factor=sf, d_override=f["d"], **kw)
        for k in range(2):
            _same_path(f, r, k, (p, "scale.factor", kw["penalty"][k]))


@pytest.mark.parametrize("p,layout", [(4500, "runs"), (6145, "scattered"), (2500, "launches below 4096")])
def test_group_operators_in_the_head_of_the_packed_triangle_pairs(oa, p, layout, monkeypatch):
    """Launch engines on the packed triangle, group penalties whose groups have <= 32 members (PathArgs::grp_head, path_large.hip:
    sympk_head_kernel<true>): the group operators run in the head of the (head, product) pairs -- every workgroup forms u of the 32
    coordinates before its own, its own and the 32 behind them, and sums a group's squares in member order (ref src/oem_dense.h:193-315) --
    instead of the product + slot sum + single-workgroup update kernel.  'runs': groups of 1 .. 32 neighbouring coordinates in a random
    order of sizes (they straddle the workgroups' 32-coordinate blocks), group 0 unpenalised, weights, penalty factors, every group kind
    with a lasso beside them; 'scattered': the same sizes dealt at random over the coordinates of an odd ragged q -- reordered into runs
    first (api.hip: group_run_permutation, beyond 4096 where every group has <= 32 members); 'launches below 4096': what the register
    engine's second attempt runs (OEM_NO_SYMCOOP=1).  Against the oracle; the same bits run to run; groups of 40 and (scattered) 90 members on
    the wider windows, groups of 120 on the update-kernel form."""
    import torch
    rng = np.random.default_rng(900 + p)
    xtx, xty = _gram_problem(p, p + p // 2, 5100 + p)
    xd = torch.as_tensor(xtx, device="cuda")
    sizes = []
    while sum(sizes) < p:
        sizes.append(int(rng.integers(1, 33)))
    sizes[-1] -= sum(sizes) - p
    groups = np.repeat(np.arange(len(sizes)), sizes)                  # label 0: unpenalised (ref src/oem_dense.h:207)
    if layout == "scattered":
        groups = rng.permutation(groups)
    ug = np.unique(groups)
    gw = rng.uniform(0.5, 2.0, len(ug))
    pf = np.ones(p); pf[:3] = 0.0; pf[3:9] = 2.0
    kw = dict(penalty=["grp.lasso", "lasso", "sparse.grp.lasso", "grp.mcp", "grp.scad.net"], groups=groups, group_weights=gw, penalty_factor=pf,
              tau=0.4, gamma=3.5, alpha=0.8, nlambda=4, lambda_min_ratio=0.05, tol=1e-9, maxit=400)
    if layout.startswith("launches"):
        monkeypatch.setenv("OEM_NO_SYMCOOP", "1")
    f = oa.oem_xtx(xd, xty, **kw)
    assert oa.last_path_engine()[0] == "launches"
    r = orc.fit_xtx(xtx, xty, native=True, unique_groups=ug, d_override=f["d"], **kw)
    for k in range(len(kw["penalty"])):
        _same_path(f, r, k, (p, layout, kw["penalty"][k]))
        assert (np.asarray(f["beta"][k])[:, -1] != 0).sum() >= 5
    f2 = oa.oem_xtx(xd, xty, **kw)
    assert all(np.array_equal(np.asarray(f["beta"][k]), np.asarray(f2["beta"][k])) for k in range(len(kw["penalty"])))
    if layout == "runs":
        # larger groups: <= 64 / <= 96 members widen the window to two / three blocks on either side (sympk_head_kernel<2>, <3>); 120: the update-kernel form
        for size in (40, 90, 120):
            big = rng.permutation(np.arange(p) // size + 1) if size == 90 else np.arange(p) // size + 1
            kwb = dict(penalty=["grp.lasso", "grp.scad"], groups=big, nlambda=3, lambda_min_ratio=0.1, tol=1e-9, maxit=400)
            fb = oa.oem_xtx(xd, xty, **kwb)
            assert oa.last_path_engine()[0] == "launches"
            rb = orc.fit_xtx(xtx, xty, native=True, unique_groups=np.unique(big), d_override=fb["d"], **kwb)
            for k in range(2):
                _same_path(fb, rb, k, (p, "groups of %d" % size, kwb["penalty"][k]))


def test_xtx_beyond_4096_streams_the_packed_lower_triangle(oa, monkeypatch):
    """q > 4096 (round 6): every product is ONE sweep over a packed copy of the lower triangle of XX (path_large.hip: sympk_*, 4 q^2 bytes
    where the row-streaming kernel reads 8 q^2; ref src/oem_xtx.h:378-381, src/oem_dense.h:508-512).  q = 6,145: a ragged last block
    and rows that are only 8-byte aligned (q odd).  Element-wise penalties take the (head, product) pairs, and so do lasso + grp.lasso in the
    parametrised test above (group operators in the head; groups of more than 96 members: the product + slot sum + update kernel form, see
    test_group_operators_in_the_head_of_the_packed_triangle_pairs).  Against the oracle, run to run (fixed summation order:
    the same bits), and against the row-streaming kernels (OEM_NO_SYM=1) on the same problem."""
    import torch
    p = 6145
    xtx, xty = _gram_problem(p, p + p // 2, 6145)
    xd = torch.as_tensor(xtx, device="cuda")
    kw = dict(penalty=["lasso", "mcp", "scad.net", "ols"], alpha=0.7, nlambda=4, lambda_min_ratio=0.05, tol=1e-9, maxit=400)
    f = oa.oem_xtx(xd, xty, **kw)
    assert oa.last_path_engine()[0] == "launches"
    lmax = _lam_max(xtx)
    assert abs(f["d"] - 1.005 * lmax) <= DTOL * lmax, (f["d"], 1.005 * lmax)
    r = orc.fit_xtx(xtx, xty, native=True, d_override=f["d"], **kw)
    for k in range(4):
        _same_path(f, r, k, (p, kw["penalty"][k]))
    g = oa.oem_xtx(xd, xty, **kw)
    assert g["d"] == f["d"] and all(np.array_equal(np.asarray(g["beta"][k]), np.asarray(f["beta"][k])) for k in range(4))
    monkeypatch.setenv("OEM_NO_SYM", "1")
    h = oa.oem_xtx(xd, xty, **kw)
    monkeypatch.delenv("OEM_NO_SYM")
    assert abs(h["d"] - f["d"]) <= 1e-13 * f["d"]
    for k in range(4):
        _same_path(f, h, k, ("row-streaming", kw["penalty"][k]), tol=1e-12)
    # `scale.factor` (the rescale round in the head of the pairs) at the same ragged size
    sf = np.linspace(0.5, 2.0, p)
    kw2 = dict(penalty=["lasso"], nlambda=3, lambda_min_ratio=0.1, tol=1e-9, maxit=400, scale_factor=sf)
    f2 = oa.oem_xtx(xd, xty, **kw2)
    r2 = orc.fit_xtx(xtx, xty, native=True, d_override=f2["d"], **kw2)
    _same_path(f2, r2, 0, (p, "scale.factor"))


@pytest.mark.parametrize("q", [4097, 4224, 6145])
def test_packed_triangle_product_against_numpy(oa, q):
    """The product kernel on its own (oemgpu_selftest_sympk_gemv): out = XX v through the packed lower triangle -- the first size past the
    register engines (one ragged block of one row), a multiple of 128, and an odd ragged size -- against numpy; and the same bits twice."""
    import ctypes as C
    import torch
    from oem_amd import _lib as L
    rng = np.random.default_rng(q)
    a = rng.normal(size=(q, q)); a = (a + a.T) / 2
    v = rng.normal(size=q)
    ad, vd = torch.as_tensor(a, device="cuda"), torch.as_tensor(v, device="cuda")
    o1, o2 = torch.empty_like(vd), torch.empty_like(vd)
    us = C.c_double(0.0)
    ctx = oa.context()
    L.check(L.lib().oemgpu_selftest_sympk_gemv(ctx, ad.data_ptr(), q, vd.data_ptr(), o1.data_ptr(), 2, C.byref(us)))
    L.check(L.lib().oemgpu_selftest_sympk_gemv(ctx, ad.data_ptr(), q, vd.data_ptr(), o2.data_ptr(), 0, C.byref(us)))
    ref = a @ v
    assert np.abs(o1.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
    assert torch.equal(o1, o2)


def test_dense_n_gt_p_at_p_5000(oa):
    """oem() with n > p at p = 5,000 (standardised, with an intercept): the MFMA moment pass at 5,002 columns, the eigen step and
    the launch-per-iteration path engines behind it."""
    rng = np.random.default_rng(5005)
    n, p = 6000, 5000
    x = np.asfortranarray(rng.normal(size=(n, p)) * 2.0 + 0.3)
    b = np.zeros(p); b[:20] = rng.uniform(-1, 1, 20)
    y = x @ b + rng.normal(size=n) + 0.5
    kw = dict(penalty=["lasso", "scad"], nlambda=4, lambda_min_ratio=0.05, tol=1e-9, maxit=400)
    f = oa.oem(x, y, **kw)
    mu = x.mean(axis=0)
    xs = x - mu
    xs /= np.sqrt((xs ** 2).sum(axis=0) / n)
    lmax = _lam_max(xs.T @ xs / n)
    assert abs(f["d"] - 1.005 * lmax) <= DTOL * lmax
    r = orc.fit_dense(x, y, native=True, d_override=f["d"], **kw)
    for k in range(2):
        _same_path(f, r, k, kw["penalty"][k])
    import torch
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    g = oa.oem(xd, y, **kw)
    assert oa.last_path_engine()[0] == "launches"
    for k in range(2):
        _same_path(g, r, k, ("device-resident", kw["penalty"][k]))
    # Nesterov's step and compute.loss on the (head, product) pairs (sympk_head_kernel<.., true>: the restart test's sum and the loss's are
    # taken one launch later, like the stop rule), with group operators in the same head: against the oracle (ref src/oem_dense.h:633-651, 759-770)
    groups = np.arange(p) // 7 + 1
    kw2 = dict(penalty=["lasso", "grp.lasso"], groups=groups, nlambda=4, lambda_min_ratio=0.05, tol=1e-9, maxit=400, accelerate=True, compute_loss=True)
    f2 = oa.oem(xd, y, **kw2)
    assert oa.last_path_engine()[0] == "launches"
    r2 = orc.fit_dense(x, y, native=True, unique_groups=np.unique(groups), d_override=f2["d"], **kw2)
    for k in range(2):
        _same_path(f2, r2, k, ("accelerate + loss", kw2["penalty"][k]), tol=1e-7)
        assert np.allclose(f2["loss"][k], r2["loss"][k], rtol=1e-8), kw2["penalty"][k]


def test_config4_scale_factor_at_p4096(oa, monkeypatch):
    """config 4 with `scale.factor` (ref src/oem_xtx.h:347-381, 576-581: the iterate is rescaled IN PLACE when a lambda ends, quirk Q5) --
    until round 5 the one production case at p = 4,096 that the register-resident engine did not take; now the owners rescale their
    coordinates and publish them in a round of their own (path_symcoop.hip, the general form).  Against the oracle through the default
    selection, and the launch-per-iteration engine (the fallback) against the same."""
    import torch
    p = 4096
    xtx, xty = _gram_problem(p, 65536 // 4, 9)
    sf = np.random.default_rng(3).uniform(0.5, 2.0, p)
    kw = dict(penalty="lasso", nlambda=3, lambda_min_ratio=0.1, tol=1e-10, maxit=500)
    xd = torch.as_tensor(xtx, device="cuda")
    f = oa.oem_xtx(xd, xty, scale_factor=sf, **kw)
    assert oa.last_path_engine()[0] == "symcoop"
    lmax = _lam_max(xtx / sf[:, None] / sf[None, :])
    assert abs(f["d"] - 1.005 * lmax) <= DTOL * lmax
    r = orc.fit_xtx(xtx, xty, native=True, scale_factor=sf, d_override=f["d"], **kw)
    _same_path(f, r, 0, "scale.factor at 4096")
    monkeypatch.setenv("OEM_NO_SYMCOOP", "1")
    g = oa.oem_xtx(xd, xty, scale_factor=sf, **kw)
    assert oa.last_path_engine()[0] == "launches"
    _same_path(g, r, 0, "scale.factor at 4096, launches")
    monkeypatch.delenv("OEM_NO_SYMCOOP")
    # several penalties, more lambdas, a smaller size of the same engine (two tiles per wave), user lambdas
    xtx2, xty2 = _gram_problem(3000, 4500, 11)
    sf2 = np.random.default_rng(4).uniform(0.25, 4.0, 3000)
    kw2 = dict(penalty=["lasso", "mcp", "scad.net", "ols"], alpha=0.6, nlambda=9, lambda_min_ratio=0.02, tol=1e-9, maxit=300)
    f2 = oa.oem_xtx(torch.as_tensor(xtx2, device="cuda"), xty2, scale_factor=sf2, **kw2)
    assert oa.last_path_engine()[0] == "symcoop"
    r2 = orc.fit_xtx(xtx2, xty2, native=True, scale_factor=sf2, d_override=f2["d"], **kw2)
    for k in range(4):
        _same_path(f2, r2, k, ("scale.factor at 3000", kw2["penalty"][k]))


@pytest.mark.parametrize("n,p", [(500, 20_000), (200, 30_000)])
def test_wres_at_its_operating_sizes_against_the_oracle(oa, n, p):
    """p >= n with Xs in the vector AND accumulator registers of ~200 CUs (path_wcoop.hip: path_wres_kernel) at the sizes it was
    built for, through the DEFAULT engine selection, against the oracle's restatement of the reference's two-product iteration
    (ref src/oem_dense.h:513-521) -- these shapes used to be held through KKT only."""
    import torch
    rng = np.random.default_rng(n + p)
    x = np.asfortranarray(rng.normal(size=(n, p)) * 1.5 + 0.2)
    b = np.zeros(p); b[:10] = rng.uniform(1.0, 2.0, 10)
    y = x @ b + rng.normal(size=n) + 0.3
    pf = rng.uniform(0.5, 2.0, p)
    kw = dict(penalty=["lasso", "mcp"], nlambda=4, lambda_min_ratio=0.05, tol=1e-8, maxit=600, penalty_factor=pf, compute_loss=True)
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f = oa.oem(xd, y, **kw)
    assert oa.last_path_engine()[0] == "wres"
    xs = x - x.mean(axis=0)
    xs /= np.sqrt((xs ** 2).sum(axis=0) / n)
    lmax = float(np.linalg.eigvalsh(xs @ xs.T / n)[-1])
    assert abs(f["d"] - 1.005 * lmax) <= DTOL * lmax
    r = orc.fit_dense(x, y, native=True, d_override=f["d"], **kw)
    for k in range(2):
        _same_path(f, r, k, (n, p, kw["penalty"][k]))
        assert np.allclose(f["loss"][k], r["loss"][k], rtol=1e-8)
    # the lasso KKT conditions on the standardised data beside it (independent of the oracle)
    ys = y - y.mean()
    sy = np.sqrt((ys ** 2).sum() / n)
    beta = np.asarray(f["beta"][0])[1:]
    sx = np.sqrt(((x - x.mean(axis=0)) ** 2).sum(axis=0) / n)
    bs = beta * sx[:, None] / sy                                 # back to the standardised scale (ref src/DataStd.h:269-293)
    grad = xs.T @ (ys[:, None] / sy - xs @ bs) / n
    lam = f["lambda"][0] / sy
    for i in range(1, 4):
        if f["niter"][0][i] > kw["maxit"]:
            continue
        nz = bs[:, i] != 0
        assert np.all(np.abs(grad[~nz, i]) <= pf[~nz] * lam[i] * (1 + 1e-5))
        assert np.abs(grad[nz, i] - pf[nz] * lam[i] * np.sign(bs[nz, i])).max() <= 1e-5 * lam[0]


@pytest.mark.parametrize("p", [600, 1300])
def test_xval_at_larger_p(oa, p):
    """xval.oem (ref src/oem_xval_dense.cpp:343-461) between p = 300 -- the largest any test had run -- and 1,183, where the
    CV-error kernel's coefficient tile used to end, and beyond it at p = 1,300 (the coefficient tile through LDS in chunks: the
    reference has no limit there; round 5 held p = 2,000 the same way, dropped for the suite's time)."""
    rng = np.random.default_rng(p)
    n, nf = (4000, 5)
    x = np.asfortranarray(rng.normal(size=(n, p)) + 0.2)
    y = x[:, :8] @ rng.uniform(0.5, 1.5, 8) + rng.normal(size=n) + 1.0
    foldid = rng.permutation(np.resize(np.arange(1, nf + 1), n))
    kw = dict(penalty=["lasso", "mcp"], nlambda=8, lambda_min_ratio=0.02, tol=1e-9, maxit=2000)
    f = oa.xval_oem(x, y, foldid=foldid, **kw)
    r = orc.xval_dense(x, y, foldid, native=True, **kw)
    assert abs(f["d"] - r["d"]) <= DTOL * r["d"]
    for k in range(2):
        _same_path(f, r, k, (p, kw["penalty"][k]))
        assert np.allclose(f["cvm"][k], r["cvm"][k], rtol=1e-8), (k, f["cvm"][k], r["cvm"][k])
        assert np.allclose(f["cvsd"][k], r["cvsd"][k], rtol=1e-8)
