#!/usr/bin/env python3
"""p >= n with Xs in the register files INCLUDING the accumulator file (path_wcoop.hip: path_wres_kernel, up to ~11 M entries) against
what served those sizes before (OEM_NO_WRES=1: the streamed persistent form or the launch-per-iteration wide engine): eigen + path
milliseconds, microseconds per iteration, agreement.    python tools/wres_time.py [number of shapes]"""
import ctypes as C, os, sys, time, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
from oem_amd import _lib as L
warnings.simplefilter("ignore")
rng = np.random.default_rng(5); lib = L.lib()
shapes = ((500, 20000, 30, "lasso"), (500, 8000, 30, "lasso"), (250, 16000, 10, "lasso"), (190, 20000, 10, "lasso"), (200, 30000, 20, "lasso"),
          (64, 100000, 10, "lasso"), (128, 40000, 10, "lasso"), (100, 30000, 10, "lasso"), (64, 50000, 20, "scad"), (1000, 8000, 10, "lasso"), (700, 12000, 10, "mcp"), (30, 100000, 10, "lasso"))
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for n, p, nlam, pen in shapes:
    x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    res = {}
    for mode in ("resident", "before"):
        os.environ.pop("OEM_NO_WRES", None)
        if mode == "before": os.environ["OEM_NO_WRES"] = "1"
        ctx = oem_amd.context(); L.check(lib.oemgpu_set_timing(ctx, 1))
        for _ in range(2):
            fit = oem_amd.oem(xd, y, penalty=pen, nlambda=nlam, tol=1e-7, compute_loss=True); torch.cuda.synchronize()
            ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms))
        it = int(fit["niter"][0].sum()); res[mode] = fit
        print(f"n={n} p={p} {pen} {nlam} lambdas [{mode}]: eigen + path {ms[3]:.1f} ms, {it} iterations, {1e3 * ms[3] / it:.2f} us per iteration", flush=True)
    a, b = res["resident"], res["before"]
    print(f"    resident vs before: |d| rel {abs(a['d'] - b['d']) / abs(b['d']):.1e}, beta {np.abs(a['beta'][0] - b['beta'][0]).max():.1e}, "
          f"loss rel {np.abs(np.ravel(a['loss'][0]) / np.ravel(b['loss'][0]) - 1).max():.1e}, "
          f"niter differ at {int((a['niter'][0] != b['niter'][0]).sum())} of {len(a['niter'][0])} lambdas", flush=True)
