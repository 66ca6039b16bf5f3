#!/usr/bin/env python3
"""config 3 (n = 1e6, p = 512, 64 groups of 8, grp.lasso, 100 lambdas, tol 1e-10) or, with `c5`, one rank's share of config 5
(1.25e7 x 256, big.oem lasso) a few times on device-resident data: the command the rocprofv3 passes of tools/round_artifacts.sh wrap."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import oem_amd

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device="cuda"); g.manual_seed(3)
if which == "c3":
    n, p = 1_000_000, 512
    xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
    bb = torch.zeros(p, dtype=torch.float64, device="cuda"); bb[:24] = torch.rand(24, generator=g, device="cuda", dtype=torch.float64) - 0.5
    yd = (xt.t() @ bb + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)).contiguous()
    kw = dict(penalty="grp.lasso", groups=np.repeat(np.arange(1, 65), 8), nlambda=100, tol=1e-10, standardize=False, intercept=False)
    for _ in range(reps):
        fit = oem_amd.oem(xt.t(), yd, **kw)
else:
    from oem_amd.distributed import HipBackend, oem_sharded
    n, p = 12_500_000, 256
    xt = torch.empty((p, n), device="cuda", dtype=torch.float64)
    for j0 in range(0, p, 32):
        xt[j0:j0 + 32].normal_(generator=g)
    bb = torch.zeros(p, dtype=torch.float64, device="cuda"); bb[:20] = torch.rand(20, generator=g, device="cuda", dtype=torch.float64)
    yd = torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
    yd += torch.mv(xt.t(), bb)
    be = HipBackend()
    for _ in range(reps):
        fit = oem_sharded(xt.t(), yd, backend=be, big=True, penalty="lasso", nlambda=100, tol=1e-7)
torch.cuda.synchronize()
print(which, "iterations", int(np.sum(fit["niter"][0])))
