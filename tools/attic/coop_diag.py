#!/usr/bin/env python3
"""Stamped segments of the cooperating-workgroup path kernel (liboemgpu_diag.so): python tools/coop_diag.py [p] [n]"""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("OEMGPU_LIB", str(ROOT / "oem_amd" / "liboemgpu_diag.so"))
import oem_amd as oa
from oem_amd import _lib as L
p = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(123)
b = np.concatenate([rng.uniform(size=24) - 0.5, np.zeros(p - 24)])
x = np.asfortranarray(rng.normal(size=(n, p))); y = x @ b + rng.normal(size=n)
kw = dict(penalty="grp.lasso", groups=np.arange(p) // 8 + 1, nlambda=100, tol=1e-10, standardize=False, intercept=False) if len(sys.argv) <= 3 else dict(penalty="lasso", nlambda=100, tol=1e-10)
fit = oa.oem(x, y, **kw)
lib = L.lib(); lib.oemgpu_diag_read_coop.argtypes = [C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)(); assert lib.oemgpu_diag_read_coop(out) == 0
d = np.array(list(out), dtype=np.float64)
names = "between rounds | product | column parts | publish | polling | stores+barrier"
print(f"p={p}: Lanczos steps {int(d[7])}, rounds {int(d[6])}: cycles per round [{names}]")
print("   ", np.round(d[0:6] / max(d[6], 1), 0), "sum", round(d[0:6].sum() / max(d[6], 1)))
print(f"OEM iterations {int(np.sum(fit['niter'][0]))}, rounds {int(d[14])}: cycles per round [{names}]")
print("   ", np.round(d[8:14] / max(d[14], 1), 0), "sum", round(d[8:14].sum() / max(d[14], 1)))
print("    between rounds, further: [u read + group factors | threshold + stop rule | OR barrier | rest (state machine, stores) -> in 'between rounds']")
print("   ", np.round(d[0:3] / max(d[14], 1), 0))
