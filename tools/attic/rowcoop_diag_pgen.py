#!/usr/bin/env python3
"""Stamped segments of the row-split engine (path_rowcoop_kernel; liboemgpu_diag.so) on oem() with p >= n through its Gram -- the bench's
500 x 2,000 lasso, whose iterates are SPARSE (the probes of tools/symcoop_diag.py end dense):  python tools/rowcoop_diag_pgen.py [n] [p] [nlambda]"""
import ctypes as C, os, sys, warnings
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("OEMGPU_LIB", str(ROOT / "oem_amd" / "liboemgpu_diag.so"))
import torch
import oem_amd as oa
from oem_amd import _lib as L
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
p = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 50
g = torch.Generator(device="cuda"); g.manual_seed(7)
xw = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
bw = torch.zeros(p, dtype=torch.float64, device="cuda"); bw[:10] = 1.0
yw = (xw.t() @ bw + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)).cpu().numpy()
fit = oa.oem(xw.t(), yw, penalty="lasso", nlambda=nl, tol=1e-7)
print("engine", oa.last_path_engine(), "non-zeros at the last lambda:", int((np.asarray(fit["beta"][0])[:, -1] != 0).sum()))
lib = L.lib(); lib.oemgpu_diag_read_rowcoop.argtypes = [C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)(); assert lib.oemgpu_diag_read_rowcoop(out) == 0
d = np.array(list(out), dtype=np.float64); it = max(d[8], 1)
print(f"n={n} p={p}: OEM iterations {int(np.sum(fit['niter'][0]))}, all-gathers of the path phase {int(d[8])}; cycles of wave 0 of workgroup 0 per iteration\n"
      "  [operator + between | products + wave sums | LDS partials barrier + publish | gather | LDS stores + vote + barrier]")
print("  path:   ", np.round(d[0:5] / it, 0), "sum", round(d[0:5].sum() / it))
