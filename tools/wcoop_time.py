#!/usr/bin/env python3
"""p >= n, small n p: the persistent cooperating-workgroup form of the wide engine (path_wcoop.hip) against the launch-per-iteration
form (path_large.hip: run_path_wide, OEM_NO_WCOOP=1) -- same call, same box: eigen + path milliseconds, iterations, microseconds per
iteration, and the largest differences between the two results."""
import ctypes as C, os, sys, time, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
from oem_amd import _lib as L
warnings.simplefilter("ignore")
rng = np.random.default_rng(5)
lib = L.lib()
shapes = ((500, 2000, 50, "lasso"), (500, 2000, 50, ["lasso", "mcp", "scad"]), (500, 2000, 50, "mcp"), (200, 5000, 30, "lasso"), (100, 1500, 30, "scad"), (64, 8000, 20, "lasso"),
          (1000, 2048, 20, "elastic.net"), (2000, 4096, 10, "lasso"), (300, 1200, 30, "lasso"))
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for n, p, nlam, pen in shapes:
    x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    kw = dict(penalty=pen, nlambda=nlam, tol=1e-7)
    os.environ["OEM_WIDE"] = "1"
    res = {}
    for mode in (("coop", "coop, one workgroup set", "launches") if isinstance(pen, list) else ("coop", "launches")):
        os.environ.pop("OEM_NO_WCOOP", None); os.environ.pop("OEM_WCOOP_ONE_SET", None)
        if mode == "launches": os.environ["OEM_NO_WCOOP"] = "1"
        if mode.endswith("set"): os.environ["OEM_WCOOP_ONE_SET"] = "1"
        ctx = oem_amd.context()
        L.check(lib.oemgpu_set_timing(ctx, 1))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); fit = oem_amd.oem(xd, y, **kw); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms))
        it = int(sum(np.sum(nk) for nk in fit["niter"]))
        res[mode] = fit
        print(f"n={n} p={p} {pen} {nlam} lambdas [{mode}]: resident {1e3 * best:.2f} ms (eigen + path {ms[3]:.2f} ms); {it} iterations, "
              f"{1e3 * ms[3] / it:.2f} us per iteration, d = {fit['d']:.12g}", flush=True)
    a, b = res["coop"], res["launches"]
    print(f"    coop vs launches: |d| rel {abs(a['d'] - b['d']) / abs(b['d']):.1e}, beta {np.abs(a['beta'][0] - b['beta'][0]).max():.1e}, "
          f"niter differ at {int((a['niter'][0] != b['niter'][0]).sum())} of {len(a['niter'][0])} lambdas", flush=True)
