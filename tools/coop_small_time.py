#!/usr/bin/env python3
"""Eigen + path milliseconds of 100-lambda paths at 129 <= q <= 288 on the engine the library picks and, with OEM_COOP_MIN_Q=129, on the
cooperating-workgroup engine: the measurements behind COOP_MIN_Q / COOP_MIN_Q_GROUPS (csrc/common.hpp)."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oem_amd
from oem_amd import _lib as L
from oem_amd.distributed import HipBackend, oem_sharded
lib = L.lib()
g = torch.Generator(device="cuda"); g.manual_seed(3)
for (n, p, big, kw) in [(300_000, 130, False, dict(penalty="grp.lasso", groups=np.arange(130) // 5 + 1, nlambda=100, tol=1e-10)),
                        (300_000, 160, False, dict(penalty="grp.lasso", groups=np.arange(160) // 5 + 1, nlambda=100, tol=1e-10)),
                        (300_000, 192, False, dict(penalty="grp.lasso", groups=np.arange(192) // 6 + 1, nlambda=100, tol=1e-10)),
                        (300_000, 192, False, dict(penalty=["lasso", "grp.lasso"], groups=np.arange(192) // 6 + 1, nlambda=100, tol=1e-10)),
                        (300_000, 209, False, dict(penalty="lasso", nlambda=100, tol=1e-10)),
                        (300_000, 200, False, dict(penalty="lasso", nlambda=100, tol=1e-10)),
                        (300_000, 288, False, dict(penalty="mcp", nlambda=100, tol=1e-10, intercept=False))]:
    xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
    bb = torch.zeros(p, dtype=torch.float64, device="cuda"); bb[:20] = torch.rand(20, generator=g, device="cuda", dtype=torch.float64)
    yd = (xt.t() @ bb + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)).contiguous()
    be = HipBackend()
    L.check(lib.oemgpu_set_timing(be.ctx, 1))
    ms = (C.c_double * 8)()
    t = []
    for _ in range(3):
        fit = oem_sharded(xt.t(), yd, backend=be, big=big, **kw)
        L.check(lib.oemgpu_last_timings(be.ctx, ms)); t.append(ms[3])
    print(f"OEM_COOP_MIN_Q={os.environ.get('OEM_COOP_MIN_Q','-')} n={n} p={p} big={big} {kw['penalty']}: eigen+path {min(t):.3f} ms, iterations {int(sum(np.sum(v) for v in fit['niter']))}")
