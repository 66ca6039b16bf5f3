#!/usr/bin/env python3
"""Gram kernel time alone (HIP events inside liboemgpu: T_GRAMK), median / min over many launches.

    [OEMGPU_LIB=...] python tools/gram_time.py [n] [p] [reps]
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from oem_amd import _lib as L  # noqa: E402
from oem_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
g = torch.Generator(device="cuda"); g.manual_seed(1)
xt = torch.empty((p, n), device="cuda", dtype=torch.float64)
for j0 in range(0, p, 16):
    xt[j0:j0 + 16].normal_(generator=g)
xt *= 3.0
y = torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
sums = torch.zeros(L.sums_len(p), dtype=torch.float64, device="cuda")
mom = torch.zeros(L.moments_len(p), dtype=torch.float64, device="cuda")
lib = L.lib()
ctx = api.context(0, torch.cuda.current_stream())
L.check(lib.oemgpu_set_timing(ctx, 1))
ms = (C.c_double * L.NTIMERS)()
L.check(lib.oemgpu_shift_sums_dev(ctx, xt.data_ptr(), n, n, p, y.data_ptr(), sums.data_ptr()))
t = []
for it in range(reps + 10):
    L.check(lib.oemgpu_moments_dev(ctx, xt.data_ptr(), n, n, p, y.data_ptr(), sums.data_ptr(), mom.data_ptr()))
    L.check(lib.oemgpu_synchronize(ctx))
    L.check(lib.oemgpu_last_timings(ctx, ms))
    if it >= 10:
        t.append(ms[L.T_GRAMK] * 1e3)
t.sort()
print(f"n={n} p={p}: gram kernel us  min {t[0]:.1f}  p25 {t[len(t)//4]:.1f}  median {t[len(t)//2]:.1f}  p75 {t[3*len(t)//4]:.1f}  max {t[-1]:.1f}")
