#!/usr/bin/env python3
"""q > 4096 through oem() (n > p) with the options that need a sum over all coordinates -- accelerate, compute.loss -- and a group penalty:
the (head, product) pairs (sympk_head_kernel<HB, true>: the sums taken one launch later) against the product + slot sum + single-workgroup update
kernel form (OEM_NO_FUSED=1): us per iteration.  python tools/large_q_general_time.py [p]"""
import os, sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch, oem_amd
from oem_amd import _lib as L
p = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = p + p // 2
g = torch.Generator(device="cuda"); g.manual_seed(p)
xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
b = torch.zeros(p, dtype=torch.float64, device="cuda"); b[:25] = 1.0
x = xt.t()
y = (x @ b + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)).cpu().numpy()
lib = L.lib(); ctx = oem_amd.context()
L.check(lib.oemgpu_set_timing(ctx, 1))
groups = np.arange(p) // 8 + 1
for label, kw in (("lasso, accelerate + compute.loss", dict(penalty=["lasso"], accelerate=True, compute_loss=True)),
                  ("lasso, compute.loss", dict(penalty=["lasso"], compute_loss=True)),
                  ("grp.lasso (groups of 8), accelerate + compute.loss", dict(penalty=["grp.lasso"], groups=groups, accelerate=True, compute_loss=True))):
    for env in ({}, {"OEM_NO_FUSED": "1"}):
        for k, v in env.items(): os.environ[k] = v
        best = 1e9
        for _ in range(2):
            fit = oem_amd.oem(x, y, nlambda=20, tol=1e-8, lambda_min_ratio=0.01, standardize=False, intercept=False, **kw); torch.cuda.synchronize()
            ms = (C.c_double * L.NTIMERS)(); L.check(lib.oemgpu_last_timings(ctx, ms)); best = min(best, ms[L.T_EIGPATH])
        for k in env: del os.environ[k]
        st, cp = C.c_int32(-1), C.c_int32(-1); lib.oemgpu_last_eigen_info(ctx, C.byref(st), C.byref(cp))
        it = int(np.sum(fit["niter"][0])) + int(st.value)
        print(f"p={p} {label}{' [update-kernel form, OEM_NO_FUSED=1]' if env else ''}: {oem_amd.last_path_engine()[0]} eigen+path {best:.1f} ms, {it} products: {1e3 * best / it:.1f} us each", flush=True)
