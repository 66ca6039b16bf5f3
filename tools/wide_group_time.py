#!/usr/bin/env python3
"""p >= n beyond the persistent engines, group penalties against the element-wise fused form: eigen + path ms and us per iteration.
   python tools/wide_group_time.py [n] [p] [group size]"""
import ctypes as C, os, sys, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
from oem_amd import _lib as L
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
p = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
gs = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(5); lib = L.lib()
x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
ctx = oem_amd.context(); L.check(lib.oemgpu_set_timing(ctx, 1))
for kw in (dict(penalty="lasso"), dict(penalty="grp.lasso", groups=np.arange(p) // gs + 1), dict(penalty="sparse.grp.lasso", groups=np.arange(p) // gs + 1),
           dict(penalty="lasso", accelerate=True)):
    for _ in range(2):
        fit = oem_amd.oem(xd, y, nlambda=10, tol=1e-7, maxit=200, **kw); torch.cuda.synchronize()
        ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms))
    it = int(sum(np.sum(k) for k in fit["niter"]))
    print(n, p, {k: (v if isinstance(v, (str, bool)) else "...") for k, v in kw.items()}, f"eigen+path {ms[3]:.1f} ms, {it} iterations, {1e3 * ms[3] / it:.2f} us/iter", flush=True)
