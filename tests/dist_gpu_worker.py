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
from oem_amd.distributed import HipBackend, oem_sharded, row_partition  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
n, p = 400_000, 100
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
if rank == 0:
    print("DIST_GPU_OK" if ok else "DIST_GPU_MISMATCH", flush=True)
dist.destroy_process_group()
