#!/usr/bin/env python3
"""A call that mixes group and element-wise penalties, made in two parts (api.hip: run_paths_parts) against one call (OEM_NO_PENALTY_SPLIT=1):
p >= n at path_wres_kernel's sizes, and n > p with 1024 < p <= 2048 (the row-split engine for the element-wise part)."""
import os, sys, time, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
warnings.simplefilter("ignore")
rng = np.random.default_rng(5)
for (n, p, nlam, tol) in ((500, 20000, 20, 1e-7), (20000, 2000, 50, 1e-9)):
    x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    kw = dict(penalty=["lasso", "mcp", "grp.lasso"], groups=np.arange(p) // 10 + 1, nlambda=nlam, tol=tol)
    for mode in ("two parts", "one call"):
        os.environ.pop("OEM_NO_PENALTY_SPLIT", None)
        __import__('oem_amd')._lib.reload_switches()      # (the library parses its switches once)
        if mode == "one call": os.environ["OEM_NO_PENALTY_SPLIT"] = "1"
        __import__('oem_amd')._lib.reload_switches()      # (the library parses its switches once)
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); f = oem_amd.oem(xd, y, **kw); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"{n} x {p}, lasso + mcp + grp.lasso, {nlam} lambdas [{mode}]: {1e3 * best:.1f} ms, iterations {[int(np.sum(v)) for v in f['niter']]}", flush=True)
