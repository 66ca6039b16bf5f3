"""Worker of tests/test_gpu_distributed.py: `python -m torch.distributed.run --nproc-per-node 2 tests/dist_gpu_worker.py`.
Both ranks sit on cuda:0 (a one-GPU box) and talk through gloo -- RCCL refuses two ranks on one device -- which still
exercises what matters: kernels launched by liboemgpu and torch.distributed collectives must be ordered on one stream."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oem_amd  # noqa: E402
from oem_amd.distributed import HipBackend, oem_sharded, row_partition, xval_oem_sharded  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
n, p = 399_996, 100             # (shards of 2 and 3 ranks with an even number of rows: both multi-GPU forms then run the same moment kernel on a shard)
g = torch.Generator(device="cuda"); g.manual_seed(7)
xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64) * 3.0      # every rank generates the whole problem
b = torch.cat([torch.rand(25, generator=g, device="cuda", dtype=torch.float64), torch.zeros(75, device="cuda", dtype=torch.float64)])
y = xt.t() @ b + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
lo, hi = row_partition(n, world)[rank]
xl = xt[:, lo:hi].contiguous().t()
yl = y[lo:hi].contiguous()
kw = dict(penalty=["elastic.net", "mcp"], intercept=True, standardize=True, tol=1e-10, nlambda=30)
be = HipBackend(0)
ok = True
for rep in range(6):                         # back to back, no host sync in between: ordering has to come from the stream
    fit = oem_sharded(xl, yl, backend=be, dist=dist, **kw)
    if rep == 0:
        ref = oem_amd.oem(xt.t(), y, **kw)
    for k in range(2):
        err = float(np.abs(fit["beta"][k] - ref["beta"][k]).max())
        ok &= err < 1e-9 and np.array_equal(fit["niter"][k], ref["niter"][k])
ok &= not be.shift_advised() and not be.shift_in_effect()      # centred data: one collective, no redo
# the ranks' moment buffers are all-gathered and added in RANK order -- the order the in-library multi-GPU form
# (opts.ngpus; here `world` contexts of the one device) adds its devices' buffers in: the two forms agree BIT FOR BIT on the same shards
if rank == 0:
    inlib = oem_amd.oem(np.asfortranarray(xt.t().cpu().numpy()), y.cpu().numpy(), devices=[0] * world, **kw)
    same_bits = inlib["d"] == fit["d"] and all(np.array_equal(np.asarray(inlib["beta"][k]), np.asarray(fit["beta"][k])) and
                                               np.array_equal(inlib["niter"][k], fit["niter"][k]) for k in range(2))
    if not same_bits:
        print("in-library sum differs from the rank-order sum: d", inlib["d"], fit["d"],
              [float(np.abs(np.asarray(inlib["beta"][k]) - np.asarray(fit["beta"][k])).max()) for k in range(2)], flush=True)
    ok &= bool(same_bits)
# columns with |mean| = 67 sd: the reduced sums call for the shift, every rank redoes its pass about the agreed c
xs = (xt[:, lo:hi] + 200.0).contiguous().t()
whole = (xt + 200.0).t()
ref = oem_amd.oem(whole, y, **kw)
for rep in range(3):
    fit = oem_sharded(xs, yl, backend=be, dist=dist, **kw)
    for k in range(2):
        err = float(np.abs(fit["beta"][k] - ref["beta"][k]).max())
        ok &= err < 1e-9 and np.abs(fit["niter"][k] - ref["niter"][k]).max() <= 1
ok &= be.shift_in_effect()
# seeded random shapes and option sets through the N > 1 call sequence (dense and big.oem semantics, ragged row split,
# columns far from zero now and then): every rank must return what the one-process fit returns
rng = np.random.default_rng(2024)
for case in range(14):
    p2 = int(rng.choice([3, 17, 64, 100, 130, 257, 300])); n2 = int(p2 + 5 + rng.integers(50, 3000))
    xh = rng.normal(size=(n2, p2)) * rng.uniform(0.5, 2.0)
    if case % 5 == 4:
        xh += rng.choice([0.0, 40.0, -700.0], size=p2)
    yh = xh[:, :2] @ np.array([1.0, -1.0]) + rng.normal(size=n2) + 0.3
    big = bool(case % 2)
    pens = [str(t) for t in rng.choice(["lasso", "mcp", "elastic.net", "grp.lasso"], int(rng.integers(1, 3)), replace=False)]
    kw2 = dict(penalty=pens, nlambda=int(rng.integers(2, 7)), alpha=0.7, tol=1e-9, maxit=300, standardize=bool(rng.integers(2)),
               intercept=bool(rng.integers(2)))
    if "grp.lasso" in pens:
        kw2["groups"] = np.arange(p2) // 3 + 1
    xfull = torch.as_tensor(np.ascontiguousarray(xh.T), device="cuda")
    yfull = torch.as_tensor(yh, device="cuda")
    lo2, hi2 = row_partition(n2, world)[rank]
    fit2 = oem_sharded(xfull[:, lo2:hi2].contiguous().t(), yfull[lo2:hi2].contiguous(), backend=be, dist=dist, big=big, **kw2)
    ref2 = (oem_amd.big_oem if big else oem_amd.oem)(np.asfortranarray(xh), yh, **kw2)
    for k in range(len(pens)):
        err = float(np.abs(fit2["beta"][k] - ref2["beta"][k]).max())
        ok &= err < 1e-8 * max(1.0, float(np.abs(ref2["beta"][k]).max()))
        ok &= bool(np.abs(np.ravel(fit2["niter"][k]).astype(int) - np.ravel(ref2["niter"][k]).astype(int)).max() <= 1)
    ok &= fit2["nobs"] == n2
# p > 288 with several penalties: the penalties are dealt round-robin to the ranks (one all-gather assembles the fit) -- the
# same numbers as every rank solving everything, and as the one-process fit
p3, n3 = 320, 4000
xh = rng.normal(size=(n3, p3)); yh = xh[:, :5] @ np.array([1.0, -1.0, 0.5, 2.0, -0.7]) + rng.normal(size=n3)
kw3 = dict(penalty=["lasso", "grp.lasso", "mcp", "scad"], groups=np.arange(p3) // 4 + 1, nlambda=6, tol=1e-9, maxit=1000)
xfull = torch.as_tensor(np.ascontiguousarray(xh.T), device="cuda"); yfull = torch.as_tensor(yh, device="cuda")
lo3, hi3 = row_partition(n3, world)[rank]
xl3, yl3 = xfull[:, lo3:hi3].contiguous().t(), yfull[lo3:hi3].contiguous()
dealt = oem_sharded(xl3, yl3, backend=be, dist=dist, **kw3)
whole3 = oem_sharded(xl3, yl3, backend=be, dist=dist, split_penalties=False, **kw3)
ref3 = oem_amd.oem(np.asfortranarray(xh), yh, **kw3)
for k in range(4):
    ok &= bool(np.array_equal(dealt["beta"][k], whole3["beta"][k]) and np.array_equal(dealt["niter"][k], whole3["niter"][k]))
    ok &= float(np.abs(dealt["beta"][k] - ref3["beta"][k]).max()) < 1e-8
# xval.oem over the row shards (three phases with the collectives between them) == the one-process xval.oem: element-wise and
# group penalties, observation weights, both error measures, a size beyond one workgroup (p + 1 > 288: the cooperating engine)
for case, (n4, p4, kw4) in enumerate([
        (30_000, 40, dict(penalty=["lasso", "mcp"], nlambda=12, tol=1e-9, maxit=2000)),
        (8_000, 25, dict(penalty=["grp.lasso", "elastic.net"], groups=np.arange(25) // 5 + 1, alpha=0.5, nlambda=8, tol=1e-9, maxit=2000,
                         type_measure="mae", standardize=False)),
        (6_000, 30, dict(penalty=["lasso"], nlambda=7, tol=1e-9, maxit=2000, intercept=False)),
        (9_000, 20, dict(penalty=["lasso", "scad"], nlambda=6, tol=1e-9, maxit=2000, weighted=True)),
        (5_000, 300, dict(penalty=["lasso"], nlambda=5, tol=1e-8, maxit=2000))]):
    xh = rng.normal(size=(n4, p4)) * 1.5 + 0.2
    yh = xh[:, :3] @ np.array([1.0, -2.0, 0.5]) + rng.normal(size=n4) + 0.7
    K4 = 5 + case
    fid = rng.permutation(np.resize(np.arange(1, K4 + 1), n4)).astype(np.int32)
    wts = rng.uniform(0.5, 2.0, size=n4) if kw4.pop("weighted", False) else None
    xfull = torch.as_tensor(np.ascontiguousarray(xh.T), device="cuda"); yfull = torch.as_tensor(yh, device="cuda")
    lo4, hi4 = row_partition(n4, world)[rank]
    fit4 = xval_oem_sharded(xfull[:, lo4:hi4].contiguous().t(), yfull[lo4:hi4].contiguous(),
                            torch.as_tensor(fid[lo4:hi4].copy(), device="cuda"), K4, backend=be, dist=dist,
                            weights_local=None if wts is None else torch.as_tensor(wts[lo4:hi4].copy(), device="cuda"), **kw4)
    ref4 = oem_amd.xval_oem(np.asfortranarray(xh), yh, foldid=fid, weights=() if wts is None else wts, **kw4)
    ok4 = fit4["nobs"] == n4 and abs(fit4["d"] - ref4["d"]) < 1e-10 * ref4["d"] and abs(fit4["lambda.min"] - ref4["lambda.min"]) < 1e-12 * ref4["lambda.min"]
    for k in range(len(kw4["penalty"])):
        ok4 &= float(np.abs(fit4["beta"][k] - ref4["beta"][k]).max()) < 1e-8 * max(1.0, float(np.abs(ref4["beta"][k]).max()))
        ok4 &= bool(np.allclose(fit4["cvm"][k], ref4["cvm"][k], rtol=1e-9) and np.allclose(fit4["cvsd"][k], ref4["cvsd"][k], rtol=1e-7))
    if not ok4 and rank == 0:
        print("xval case", case, "mismatch: nobs", fit4["nobs"], "d", fit4["d"], ref4["d"], "lambda.min", fit4["lambda.min"], ref4["lambda.min"],
              [(float(np.abs(fit4["beta"][k] - ref4["beta"][k]).max()), float(np.abs(fit4["cvm"][k] / ref4["cvm"][k] - 1).max()),
                float(np.abs(fit4["cvsd"][k] / ref4["cvsd"][k] - 1).max())) for k in range(len(kw4["penalty"]))], flush=True)
    ok &= bool(ok4)
if rank == 0:
    print("DIST_GPU_OK" if ok else "DIST_GPU_MISMATCH", flush=True)
dist.destroy_process_group()
