#!/usr/bin/env python3
"""The moment kernel of config 3 (n = 1e6, p = 512) or any (n, p): milliseconds back to back (oemgpu_moments_dev, wall clock over 10 calls) and inside
whole oem() calls (the library's own HIP events), with the fraction of the FP64-MFMA peak.  python tools/gram_c3_time.py [n] [p]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
from oem_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = torch.Generator(device="cuda"); g.manual_seed(3)
xt = torch.empty((p, n), device="cuda", dtype=torch.float64)
for j0 in range(0, p, 64):
    xt[j0:j0 + 64].normal_(generator=g)
yd = torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
lib = L.lib(); ctx = oem_amd.context()
M = torch.zeros((p + 2) * (p + 2), device="cuda", dtype=torch.float64)
fl = float(n) * p * (p + 1.0) + 2.0 * n * p
for _ in range(3):
    L.check(lib.oemgpu_moments_dev(ctx, xt.data_ptr(), n, n, p, yd.data_ptr(), None, M.data_ptr()))
L.check(lib.oemgpu_synchronize(ctx))
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        L.check(lib.oemgpu_moments_dev(ctx, xt.data_ptr(), n, n, p, yd.data_ptr(), None, M.data_ptr()))
    L.check(lib.oemgpu_synchronize(ctx))
    ts.append((time.perf_counter() - t0) / 10)
bb = min(ts)
L.check(lib.oemgpu_set_timing(ctx, 1))
ins = []
grp = np.repeat(np.arange(1, p // 8 + 1), 8) if p % 8 == 0 else ()
for _ in range(5):
    oem_amd.oem(xt.t(), yd, penalty="grp.lasso" if len(grp) else "lasso", groups=grp, nlambda=100, tol=1e-10, standardize=False, intercept=False)
    ms = (C.c_double * L.NTIMERS)(); L.check(lib.oemgpu_last_timings(ctx, ms)); ins.append(ms[L.T_GRAMK])
print(f"n={n} p={p}: moments back to back {1e3 * bb:.3f} ms (reduce included) = {fl / bb / 1e12:.1f} TF = {fl / bb / 1e12 / 78.6:.3f} of peak; "
      f"Gram kernel inside oem() calls {np.median(ins):.3f} ms = {fl / (np.median(ins) * 1e-3) / 1e12 / 78.6:.3f} of peak  {np.round(ins, 3)}")
