#!/usr/bin/env python3
"""config 3 (p = 512, grp.lasso) and a p = 1024 lasso on the persistent cooperating-workgroup engine vs the launch-per-iteration
engines: eigen + path milliseconds (HIP events) and agreement."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oem_amd
from oem_amd import _lib as L
lib = L.lib(); ctx = oem_amd.context()
def run(f):
    L.check(lib.oemgpu_set_timing(ctx, 1)); fit = f(); fit = f()
    ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms)); L.check(lib.oemgpu_set_timing(ctx, 0))
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    return fit, ms[3], 1e3 * wall
g = torch.Generator(device="cuda"); g.manual_seed(3)
for (n, p, kw) in [(1_000_000, 256, dict(penalty="lasso", nlambda=100, tol=1e-7)),
                   (1_000_000, 512, dict(penalty="grp.lasso", groups=np.repeat(np.arange(1, 65), 8), nlambda=100, tol=1e-10, standardize=False, intercept=False)),
                   (200_000, 1024, dict(penalty="lasso", nlambda=100, tol=1e-10)),
                   (200_000, 300, dict(penalty=["lasso", "mcp", "grp.lasso"], groups=np.arange(300) // 6 + 1, nlambda=100, tol=1e-10)),
                   (100_000, 700, dict(penalty=["lasso", "scad"], nlambda=50, tol=1e-9, compute_loss=True))]:
    xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
    bb = torch.zeros(p, dtype=torch.float64, device="cuda"); bb[:24] = torch.rand(24, generator=g, device="cuda", dtype=torch.float64) - 0.5
    yd = (xt.t() @ bb + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)).contiguous()
    f = lambda: oem_amd.oem(xt.t(), yd, **kw)
    res = {}
    for name, env in (("coop", {}), ("coop, device-scope exchange", {"OEM_NO_ONE_XCD": "1"}), ("launch-per-iteration", {"OEM_NO_COOP": "1"})):
        for k, v in env.items(): os.environ[k] = v
        try:
            res[name] = run(f)
            if name == "coop": print(f"  placement of the cooperating engine: {oem_amd.api.last_placement()}")
        finally:
            for k in env: del os.environ[k]
    base = res["launch-per-iteration"][0]
    for name, (fit, eig, wall) in res.items():
        err = max(np.abs(fit["beta"][k] - base["beta"][k]).max() for k in range(len(fit["beta"])))
        dn = max(np.abs(fit["niter"][k].astype(int) - base["niter"][k].astype(int)).max() for k in range(len(fit["beta"])))
        print(f"n={n} p={p} {kw['penalty']}: {name:28s} eigen+path {eig:8.3f} ms, whole call {wall:8.3f} ms, iterations {int(sum(np.sum(v) for v in fit['niter']))}, max|dbeta| {err:.1e}, max dniter {dn}")
    del xt, yd
