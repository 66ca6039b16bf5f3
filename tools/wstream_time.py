#!/usr/bin/env python3
"""p >= n where Xs does not fit the registers: the STREAMED persistent form (path_wcoop.hip: path_wstream_kernel -- one launch, the
column tiles re-read every iteration, the all-reduce in-kernel) against the launch-per-iteration wide engine (OEM_NO_WSTREAM=1):
eigen + path milliseconds, microseconds per iteration, TB/s over one read of Xs per iteration, agreement."""
import ctypes as C, os, sys, time, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
from oem_amd import _lib as L
warnings.simplefilter("ignore")
rng = np.random.default_rng(5); lib = L.lib()
shapes = ((30, 100000, 10, "lasso"), (64, 100000, 10, "lasso"), (100, 30000, 10, "lasso"), (128, 40000, 10, "lasso"), (128, 200000, 10, "lasso"), (64, 50000, 20, "scad"),
          (500, 8000, 30, "lasso"), (250, 16000, 10, "lasso"), (190, 20000, 10, "lasso"), (500, 20000, 30, "lasso"), (200, 30000, 20, "lasso"), (200, 100000, 20, "lasso"))
os.environ["OEM_WSTREAM"] = "1"                                   # (the comparison also where the library would not take it)
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for n, p, nlam, pen in shapes:
    x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    res = {}
    for mode in ("streamed", "launches"):
        os.environ.pop("OEM_NO_WSTREAM", None)
        if mode == "launches": os.environ["OEM_NO_WSTREAM"] = "1"
        ctx = oem_amd.context(); L.check(lib.oemgpu_set_timing(ctx, 1))
        for _ in range(2):
            fit = oem_amd.oem(xd, y, penalty=pen, nlambda=nlam, tol=1e-7); torch.cuda.synchronize()
            ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms))
        it = int(fit["niter"][0].sum()); res[mode] = fit
        byt = 8.0 * 64 * ((n + 63) // 64) * p
        print(f"n={n} p={p} {pen} {nlam} lambdas [{mode}]: eigen + path {ms[3]:.1f} ms, {it} iterations, {1e3 * ms[3] / it:.2f} us per iteration, "
              f"{byt * it / (ms[3] * 1e-3) / 1e12:.2f} TB/s over one read of Xs per iteration", flush=True)
    a, b = res["streamed"], res["launches"]
    print(f"    streamed vs launches: |d| rel {abs(a['d'] - b['d']) / abs(b['d']):.1e}, beta {np.abs(a['beta'][0] - b['beta'][0]).max():.1e}, "
          f"niter differ at {int((a['niter'][0] != b['niter'][0]).sum())} of {len(a['niter'][0])} lambdas", flush=True)
