#!/usr/bin/env python3
"""Microseconds per OEM iteration of 100-lambda paths on the row-split kernel (p <= 208) over its sizes, operators and options:
a table to read for cliffs (an operator that costs a multiple of its neighbours has started to spill: DESIGN.md section 3.2)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oem_amd

rng = np.random.default_rng(7)
pens = [("lasso", {}), ("mcp", {}), ("scad", {}), ("elastic.net", dict(alpha=0.5)), ("mcp.net", dict(alpha=0.5)), ("grp.lasso", {}), ("sparse.grp.lasso", {}), ("grp.mcp", {})]
print("p      " + "".join(f"{name:>18s}" for name, _ in pens) + "   (+ the same with accelerate = TRUE)")
for p in (30, 64, 80, 100, 128, 150, 176, 200, 208):
    n = 20 * p
    x = rng.normal(size=(n, p)) * 2.0
    y = x[:, :5] @ np.array([1.0, -1.0, 0.5, 2.0, -0.7]) + rng.normal(size=n)
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    row = []
    for acc in (False, True):
        for name, extra in pens:
            kw = dict(penalty=name, nlambda=100, tol=1e-9, maxit=2000, accelerate=acc, **extra)
            if "grp" in name:
                kw["groups"] = np.arange(p) // 5 + 1
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); fit = oem_amd.oem(xd, y, **kw); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            row.append(1e6 * best / max(1, int(np.sum(fit["niter"][0]))))
    k = len(pens)
    print(f"{p:<6d} " + "".join(f"{v:18.3f}" for v in row[:k]))
    print("  acc  " + "".join(f"{v:18.3f}" for v in row[k:]))
