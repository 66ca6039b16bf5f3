#!/usr/bin/env python3
"""Stamped segments of the wide cooperating-workgroup path kernel (liboemgpu_diag.so): python tools/wcoop_diag.py [n] [p] [nlambda]"""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("OEMGPU_LIB", str(ROOT / "oem_amd" / "liboemgpu_diag.so"))
os.environ["OEM_WIDE"] = "1"
import oem_amd as oa
from oem_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
p = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rng = np.random.default_rng(5)
x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
gen = len(sys.argv) > 4
fit = oa.oem(x, y, penalty="grp.lasso", groups=np.arange(p) // 5 + 1, nlambda=nl, tol=1e-7) if gen else oa.oem(x, y, penalty="lasso", nlambda=nl, tol=1e-7)
lib = L.lib()
# (path_wres.hip is a translation unit of its own since round 6: its kernels' stamps have a reader of their own)
out = (C.c_ulonglong * 16)()
for rd in (lib.oemgpu_diag_read_wcoop, lib.oemgpu_diag_read_wres):          # (whichever engine ran left its stamps)
    rd.argtypes = [C.POINTER(C.c_ulonglong)]
    assert rd(out) == 0
    if any(out):
        break
d = np.array(list(out), dtype=np.float64)
names = "product + between | barrier | publish 1 | gather 1 | barrier | slice sums + publish 2 | gather 2 | barrier"
print(f"n={n} p={p}: OEM iterations {int(np.sum(fit['niter'][0]))}, all-reduces {int(d[8])}: cycles per iteration [{names}]")
print("   ", np.round(d[0:8] / max(d[8], 1), 0), "sum", round(d[0:8].sum() / max(d[8], 1)))
print("    product, further: [between + vector reads | dot products | column sums | operator + stop rule | update] (the partial-vector stores are in the first segment above)")
print("   ", np.round(d[9:14] / max(d[8], 1), 0))
if gen:
    print("    group operators: exchange of u", round(d[14] / max(d[8], 1)), "cycles, group norms + factors", round(d[15] / max(d[8], 1)), "cycles")
