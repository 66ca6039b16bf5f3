#!/usr/bin/env python3
"""Stamp breakdown of the ring Gram kernel (liboemgpu_diag.so, built by `python -m oem_amd.build --diag`).

    OEMGPU_LIB=oem_amd/liboemgpu_diag.so python tools/gram_diag.py [n] [p] [mean]

Prints s_memtime ticks (100 MHz) of workgroup 7 / wave 0: prologue, steady (sum), drain, epilogue, slab count.
"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("OEMGPU_LIB", str(ROOT / "oem_amd" / "liboemgpu_diag.so"))

import torch  # noqa: E402
from oem_amd import _lib as L  # noqa: E402
from oem_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mean = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
g = torch.Generator(device="cuda"); g.manual_seed(1)
xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64) * 3.0 + mean
y = torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
sums = torch.zeros(L.sums_len(p), dtype=torch.float64, device="cuda")
mom = torch.zeros(L.moments_len(p), dtype=torch.float64, device="cuda")
lib = L.lib()
ctx = api.context(0, torch.cuda.current_stream())
reader = "oemgpu_gram_sb_diag_read" if p + 2 > 112 else "oemgpu_gram_diag_read"      # (the shared-slab kernel is a translation unit of its own: gram_sb.hip)
have_diag = hasattr(lib, reader)                              # only the -DOEM_GRAM_DIAG build exports it
if have_diag:
    getattr(lib, reader).argtypes = [C.POINTER(C.c_ulonglong)]
L.check(lib.oemgpu_set_timing(ctx, 1))
ms = (C.c_double * L.NTIMERS)()
for it in range(5):
    L.check(lib.oemgpu_shift_sums_dev(ctx, xt.data_ptr(), n, n, p, y.data_ptr(), sums.data_ptr()))
    L.check(lib.oemgpu_moments_dev(ctx, xt.data_ptr(), n, n, p, y.data_ptr(), sums.data_ptr(), mom.data_ptr()))
    L.check(lib.oemgpu_synchronize(ctx))
    L.check(lib.oemgpu_last_timings(ctx, ms))
out = (C.c_ulonglong * 8)()
if have_diag:
    assert getattr(lib, reader)(out) == 0
d = list(out)
if p + 2 > 112:
    # shared-slab kernel: one diagonal (sbk 0) and one off-diagonal (sbk 1) workgroup of the first row chunk
    print(f"n={n} p={p}: gram kernel {ms[L.T_GRAMK]*1e3:.1f} us")
    for k, name in ((0, "diagonal super-block    (18 MFMA per wave and slab: floor 1152 cycles)"), (1, "off-diagonal super-block (32 MFMA per wave and slab: floor 2048 cycles)")):
        cyc, ticks, slabs, isd = d[4 * k:4 * k + 4]
        if ticks and slabs:
            print(f"  {name}: {cyc} cycles in {ticks / 100.0:.1f} us = {cyc / ticks * 0.1:.3f} GHz held, {slabs} slabs, {cyc / slabs:.0f} cycles per slab (incl. prologue / epilogue)")
else:
    ns = d[7]
    print(f"n={n} p={p} mean={mean}: gram kernel {ms[L.T_GRAMK]*1e3:.1f} us; cycles: prologue {d[0]} steady {d[4]} drain {d[5]} epilogue {d[6]}; slabs/wave {ns}")
    if ns:
        print(f"steady cycles per slab {d[4] / max(1, ns - 5):.2f}  (issue floor at p = 100: 2,877)")
