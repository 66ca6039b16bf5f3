#!/usr/bin/env python3
"""q > 4096: the packed-triangle products (path_large.hip: sympk_*) against the row-streaming kernels (OEM_NO_SYM=1) on the same problem, both
forms (head + product pairs for element-wise penalties; product + slot sum + update kernel for everything else), and run to run:
python tools/spk_check.py [p ...]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import oem_amd
from oem_amd import _lib as L

def run(xtx, xty, env=None, **kw):
    for k, v in (env or {}).items(): os.environ[k] = v
    L.reload_switches()
    try:
        f = oem_amd.oem_xtx(xtx, xty, **kw)
        return f, oem_amd.last_path_engine()[0]
    finally:
        for k in (env or {}): del os.environ[k]
        L.reload_switches()

for p in [int(a) for a in sys.argv[1:]] or [4097, 6145, 8192]:
    g = torch.Generator(device="cuda"); g.manual_seed(p)
    n = p + p // 2
    x = torch.randn((n, p), generator=g, device="cuda", dtype=torch.float64)
    b = torch.zeros(p, dtype=torch.float64, device="cuda"); b[:25] = 1.0
    y = x @ b + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
    xtx = (x.t() @ x) / n; xty = ((x.t() @ y) / n).cpu().numpy()
    del x
    grp = np.arange(p) // 7 + 1
    for label, kw in (("element-wise", dict(penalty=["lasso", "mcp", "ols", "scad.net"], alpha=0.7)),
                      ("general", dict(penalty=["lasso", "grp.lasso"], groups=grp)),
                      ("accelerate+loss... xtx has neither: scale.factor", dict(penalty=["lasso"], scale_factor=np.linspace(0.5, 2.0, p)))):
        kw = dict(kw, nlambda=6, tol=1e-9, lambda_min_ratio=0.02, maxit=300)
        f1, e1 = run(xtx, xty, **kw)
        f2, e2 = run(xtx, xty, **kw)
        f0, e0 = run(xtx, xty, env={"OEM_NO_SYM": "1"}, **kw)
        same = all(np.array_equal(np.asarray(f1["beta"][k]), np.asarray(f2["beta"][k])) for k in range(len(kw["penalty"]))) and f1["d"] == f2["d"]
        err = max(float(np.abs(np.asarray(f1["beta"][k]) - np.asarray(f0["beta"][k])).max()) for k in range(len(kw["penalty"])))
        dn = max(int(np.abs(np.asarray(f1["niter"][k]).astype(int) - np.asarray(f0["niter"][k]).astype(int)).max()) for k in range(len(kw["penalty"])))
        print(f"p={p} {label}: engines {e1}/{e0}  same bits run to run: {same}  max|beta - beta(row-streaming)| = {err:.2e}  niter diff {dn}  "
              f"d rel diff {abs(f1['d'] - f0['d']) / f0['d']:.1e}  iterations {int(sum(np.sum(f1['niter'][k]) for k in range(len(kw['penalty']))))}", flush=True)
    del xtx; torch.cuda.empty_cache()
