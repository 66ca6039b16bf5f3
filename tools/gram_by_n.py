#!/usr/bin/env python3
"""The moment kernel over row counts (the plan of row chunks matters most at moderate n).  Back-to-back launches (no
synchronisation in between, so the clocks stay up); the time is the library's own event pair around the LAST launch's moment
kernel (T_GRAMK), best of four rounds; flops n p (p + 1).

    python tools/gram_by_n.py p n1 n2 ...
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from oem_amd import _lib as L  # noqa: E402
from oem_amd import api  # noqa: E402

p = int(sys.argv[1])
lib = L.lib()
ctx = api.context(0, torch.cuda.current_stream())
for n in [int(a) for a in sys.argv[2:]]:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    xt = torch.empty((p, n), device="cuda", dtype=torch.float64)
    for j0 in range(0, p, 16):
        xt[j0:j0 + 16].normal_(generator=g)
    y = torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
    sums = torch.zeros(L.sums_len(p), dtype=torch.float64, device="cuda")
    mom = torch.zeros(L.moments_len(p), dtype=torch.float64, device="cuda")
    L.check(lib.oemgpu_shift_sums_dev(ctx, xt.data_ptr(), n, n, p, y.data_ptr(), sums.data_ptr()))
    reps = max(3, int(2e11 / (n * p * (p + 1.0))))
    best = None
    L.check(lib.oemgpu_set_timing(ctx, 1))
    tm = (C.c_double * L.NTIMERS)()
    for rnd in range(4):
        for it in range(reps):
            L.check(lib.oemgpu_moments_dev(ctx, xt.data_ptr(), n, n, p, y.data_ptr(), sums.data_ptr(), mom.data_ptr()))
        L.check(lib.oemgpu_synchronize(ctx))
        L.check(lib.oemgpu_last_timings(ctx, tm))
        ms = tm[L.T_GRAMK]
        best = ms if best is None else min(best, ms)
    fl = n * p * (p + 1.0)
    print(f"p={p} n={n} ({8.0 * n * p / 1e6:.0f} MB of X, {reps} launches back to back): {best * 1e3:.1f} us per moment kernel = {fl / best / 1e9:.1f} TF = {fl / best / 1e9 / 78.6:.3f} of peak")
    del xt, y
