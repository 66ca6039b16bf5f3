#!/usr/bin/env python3
"""Stamped phases of path_update_kernel<R> behind the packed-triangle product (liboemgpu_diag.so): cycles of thread 0 per call.
python tools/attic/update_diag.py [q] [penalty]"""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("OEMGPU_LIB", str(ROOT / "oem_amd" / "liboemgpu_diag.so"))
import torch, oem_amd
from oem_amd import _lib as L
p = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
pen = sys.argv[2] if len(sys.argv) > 2 else "grp.lasso"
g = torch.Generator(device="cuda"); g.manual_seed(p)
n = p + p // 2
x = torch.randn((n, p), generator=g, device="cuda", dtype=torch.float64)
b = torch.zeros(p, dtype=torch.float64, device="cuda"); b[:25] = 1.0
y = x @ b + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
xtx = (x.t() @ x) / n; xty = ((x.t() @ y) / n).cpu().numpy(); del x
grp = np.arange(p) // 8 + 1 if pen.startswith("grp") else ()
kw = dict(penalty=pen, groups=grp, nlambda=20, tol=1e-8, lambda_min_ratio=0.01)
if not pen.startswith("grp"): kw["scale_factor"] = np.ones(p)          # (element-wise: the general three-launch form all the same)
lib = L.lib(); lib.oemgpu_diag_read_update.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 8)()
oem_amd.oem_xtx(xtx, xty, **kw); lib.oemgpu_diag_read_update(out, 1)
fit = oem_amd.oem_xtx(xtx, xty, **kw); lib.oemgpu_diag_read_update(out, 1)
d = np.array(list(out), dtype=np.float64); n_ = max(d[7], 1)
print(f"q={p} {pen}: {int(d[7])} update launches; cycles of thread 0 per launch: state loaded {d[0]/n_:.0f} | constants {d[1]/n_:.0f} | u in LDS {d[2]/n_:.0f} | "
      f"group factors {d[3]/n_:.0f} | coefficients {d[4]/n_:.0f} | stop vote {d[5]/n_:.0f} | bookkeeping + outputs {d[6]/n_:.0f} | sum {d[:7].sum()/n_:.0f}")
