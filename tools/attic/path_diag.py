#!/usr/bin/env python3
"""Round-segment stamps of the small-p path kernel (liboemgpu_diag.so; `python -m oem_amd.build --diag`).

    python tools/path_diag.py [p] [n]

Segments (shader cycles, wave 0, summed over rounds): 0 = threshold / stop rule / everything between rounds,
1 = strip write + broadcast reads + FMAs, 2 = partial write + barrier, 3 = partial reads + adds (+ exchange).
Slots 4..7 hold the Lanczos share of the same four.
"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("OEMGPU_LIB", str(ROOT / "oem_amd" / "liboemgpu_diag.so"))

import oem_amd as oa  # noqa: E402
from oem_amd import _lib as L  # noqa: E402

p = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(123)
b = np.concatenate([rng.uniform(size=p // 4), np.zeros(p - p // 4)])
x = np.asfortranarray(rng.normal(size=(n, p)) * 3.0)
y = x @ b + rng.normal(size=n)
fit = oa.oem(x, y, penalty="elastic.net", intercept=True, standardize=False, tol=1e-10)
fit = oa.oem(x, y, penalty="elastic.net", intercept=True, standardize=False, tol=1e-10, lambda_=fit["lambda"][0])
lib = L.lib()
lib.oemgpu_diag_read.argtypes = [C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 24)()
assert lib.oemgpu_diag_read(out) == 0
d = np.array(list(out), dtype=np.float64)
nit = int(np.sum(fit["niter"][0]))
lz = d[4:8]; tot = d[0:4]; oem = tot - lz
print('per-lambda overhead (slot 8):', int(d[8]), 'cycles total,', round(d[8] / max(1, len(fit['lambda'][0])), 1), 'per lambda')
print(f"p={p}: OEM iterations {nit}; Lanczos steps {int(d[11])}; top_ritz calls total {int(d[9])} cycles; Lanczos vector work {int(d[10])} cycles")
if p > 64 and p <= 128:
    r = d[0:5]
    print("row-split kernel, cycles per OEM round by segment [threshold+stop+loop | stores+barrier | reads | FMAs | adds+reduce]:")
    print("   ", np.round(r / max(nit, 1), 1), "sum", round(r.sum() / max(nit, 1), 1))
    print("    per-lambda (slot 8):", round(d[8] / len(fit["lambda"][0]), 1), " prologue:", int(d[5]), " top_ritz calls total:", int(d[9]),
          " Lanczos steps:", int(d[6]), "x", round(d[10] / max(d[6], 1), 1), "cycles (stamped)")
    ns = max(d[6], 1)
    print("    Lanczos step by segment [product + reduce-scatter + alpha share | stores + barrier | reads + alpha sum | orthogonalise copy + in-wave norm + rsqrt | T stores, tests, next vector]:")
    seg = np.array([d[15], d[13], d[14], d[17], d[19]])
    print("   ", np.round(seg / ns, 1), "sum", round(seg.sum() / ns, 1))
    sys.exit(0)
print("Lanczos cycles by segment:", lz.astype(int), "sum", int(lz.sum()))
print("OEM cycles by segment    :", oem.astype(int), "sum", int(oem.sum()))
print("per OEM round            :", np.round(oem / max(nit, 1), 1), "sum", round(oem.sum() / max(nit, 1), 1))
