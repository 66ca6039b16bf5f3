"""Worker of tests/test_gpu_distributed.py::test_rccl_collectives_at_world_size_one (run as a child process: a process group is
process-wide state).  One rank, backend "nccl" (= RCCL on ROCm), OEM_FORCE_COLLECTIVES=1: every collective of
oem_amd/distributed.py -- the Gram exchange (all-gather + in-order sum; dist.all_reduce on request), the shift redo's two extra ones, the
penalty split's all-gather, xval.oem's exchange and all-gather, the global-n all-reduce -- EXECUTES on a one-GPU box and must leave the bits of the plain call."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oem_amd  # noqa: E402
from oem_amd.distributed import HipBackend, oem_sharded, xval_oem_sharded  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
be = HipBackend(0)
rng = np.random.default_rng(31)
ok = True
calls = {"all_reduce": 0, "all_gather": 0, "gather_sum": 0}           # gather_sum: all_gather_into_tensor, behind which the buffers are added in rank order
_ar, _ag, _agt = dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor


def ar(*a, **k):
    calls["all_reduce"] += 1
    return _ar(*a, **k)


def ag(*a, **k):
    calls["all_gather"] += 1
    return _ag(*a, **k)


def agt(*a, **k):
    calls["gather_sum"] += 1
    return _agt(*a, **k)


dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor = ar, ag, agt          # count what really reaches RCCL


def both(fn, *a, **k):
    os.environ.pop("OEM_FORCE_COLLECTIVES", None)
    plain = fn(*a, dist=None, **k)
    os.environ["OEM_FORCE_COLLECTIVES"] = "1"
    forced = fn(*a, dist=dist, **k)
    os.environ.pop("OEM_FORCE_COLLECTIVES", None)
    return plain, forced


def same(a, b, keys=("beta", "lambda", "niter")):
    r = True
    for key in keys:
        for u, v in zip(a[key], b[key]):
            r &= bool(np.array_equal(np.asarray(u), np.asarray(v)))
    return r and a["d"] == b["d"]


def check(name, cond):
    global ok
    if not cond:
        print("FAILED:", name, calls, flush=True)
    ok &= bool(cond)


# 1. the c1-shaped solve: the global-n all-reduce (default lambda.min.ratio), moments -> all-reduce -> solve
n, p = 200_000, 100
xt = torch.randn((p, n), device=dev, dtype=torch.float64) * 3.0
y = xt.t()[:, :5] @ torch.tensor([1.0, -1.0, 0.5, 2.0, -0.7], device=dev, dtype=torch.float64) + torch.randn(n, device=dev, dtype=torch.float64)
c0 = dict(calls)
a, b = both(oem_sharded, xt.t(), y, backend=be, penalty=["elastic.net", "mcp"], nlambda=30, tol=1e-10)
check("dense", same(a, b) and calls["all_reduce"] == c0["all_reduce"] + 1 and calls["gather_sum"] == c0["gather_sum"] + 1)      # the global n; the moments
c0 = dict(calls)
a, b = both(oem_sharded, xt.t(), y, backend=be, big=True, penalty=["lasso"], nlambda=20, tol=1e-9)
check("big", same(a, b) and calls["all_reduce"] == c0["all_reduce"] + 1 and calls["gather_sum"] == c0["gather_sum"] + 1)
c0 = dict(calls)
a, b = both(oem_sharded, xt.t(), y, backend=be, penalty=["lasso"], nlambda=10, tol=1e-9, lambda_min_ratio=1e-3, n_total=n)
check("dense, n known", same(a, b) and calls["all_reduce"] == c0["all_reduce"] and calls["gather_sum"] == c0["gather_sum"] + 1)       # the ONE exchange of the north star
c0 = dict(calls)
a, b = both(oem_sharded, xt.t(), y, backend=be, penalty=["lasso"], nlambda=10, tol=1e-9, lambda_min_ratio=1e-3, n_total=n, reduce="allreduce")
check("dense, n known, all-reduce", same(a, b) and calls["all_reduce"] == c0["all_reduce"] + 1 and calls["gather_sum"] == c0["gather_sum"])   # ... as dist.all_reduce, still selectable
# 2. columns far from zero: the reduced moments advise a shift -> sample sums all-reduce + second moment all-reduce
xs = (xt + 200.0).t()
c0 = dict(calls)
a, b = both(oem_sharded, xs, y, backend=be, penalty=["lasso"], nlambda=15, tol=1e-10)
check("shift redo", same(a, b) and be.shift_in_effect() and calls["all_reduce"] == c0["all_reduce"] + 1 and calls["gather_sum"] == c0["gather_sum"] + 3)      # n; moments, sample sums, moments about c
# 3. p > 288 with several penalties: penalties dealt to the ranks, one all-gather
p3, n3 = 320, 4000
xh = rng.normal(size=(n3, p3)); yh = xh[:, :5] @ np.array([1.0, -1.0, 0.5, 2.0, -0.7]) + rng.normal(size=n3)
x3 = torch.as_tensor(np.ascontiguousarray(xh.T), device=dev).t(); y3 = torch.as_tensor(yh, device=dev)
c0 = dict(calls)
a, b = both(oem_sharded, x3, y3, backend=be, penalty=["lasso", "grp.lasso", "mcp"], groups=np.arange(p3) // 4 + 1, nlambda=6, tol=1e-9, maxit=1000)
check("penalty split", same(a, b) and calls["all_gather"] == c0["all_gather"] + 1)
# 4. xval.oem over "row shards": fold-moment all-reduce (+ the n all-reduce), triple all-gather
n4, p4, K4 = 20_000, 40, 6
xh = rng.normal(size=(n4, p4)) * 1.5 + 0.2; yh = xh[:, :3] @ np.array([1.0, -2.0, 0.5]) + rng.normal(size=n4) + 0.7
fid = rng.permutation(np.resize(np.arange(1, K4 + 1), n4)).astype(np.int32)
x4 = torch.as_tensor(np.ascontiguousarray(xh.T), device=dev).t(); y4 = torch.as_tensor(yh, device=dev); f4 = torch.as_tensor(fid, device=dev)
c0 = dict(calls)
a, b = both(xval_oem_sharded, x4, y4, f4, K4, backend=be, penalty=["lasso", "mcp"], nlambda=10, tol=1e-9, maxit=2000)
check("xval", same(a, b, keys=("beta", "lambda", "niter", "cvm", "cvsd")) and calls["all_reduce"] == c0["all_reduce"] + 1 and calls["gather_sum"] == c0["gather_sum"] + 1 and
      calls["all_gather"] == c0["all_gather"] + 1)
print("RCCL_W1_OK" if ok else "RCCL_W1_MISMATCH", calls, flush=True)
dist.destroy_process_group()
