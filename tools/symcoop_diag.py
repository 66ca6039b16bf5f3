#!/usr/bin/env python3
"""Stamped segments of the register-resident symmetric engine (path_symcoop.hip; liboemgpu_diag.so): python tools/symcoop_diag.py [p] [nlambda] [n] [lambda.min.ratio]"""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("OEMGPU_LIB", str(ROOT / "oem_amd" / "liboemgpu_diag.so"))
import torch
import oem_amd as oa
from oem_amd import _lib as L
p = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(123)
n = int(sys.argv[3]) if len(sys.argv) > 3 else max(2 * p, 16384)
x = rng.normal(size=(n, p)); b = np.zeros(p); b[:25] = rng.uniform(-1, 1, 25); y = x @ b + rng.normal(size=n)
xtx, xty = x.T @ x / n, x.T @ y / n
lmr = float(sys.argv[4]) if len(sys.argv) > 4 else None
fit = oa.oem_xtx(torch.as_tensor(xtx, device="cuda"), xty, penalty="lasso", nlambda=nl, tol=1e-10, lambda_min_ratio=lmr)
print("non-zeros at the last lambda:", int((np.asarray(fit["beta"][0])[:, -1] != 0).sum()), " kernel ms:", None)
lib = L.lib()
if p <= 2048 and not os.environ.get("OEM_NO_ROWCOOP"):
    # the row-split one-exchange engine (path_rowcoop_kernel) served this size
    lib.oemgpu_diag_read_rowcoop.argtypes = [C.POINTER(C.c_ulonglong)]
    out = (C.c_ulonglong * 16)(); assert lib.oemgpu_diag_read_rowcoop(out) == 0
    d = np.array(list(out), dtype=np.float64); it = max(d[8], 1)
    print(f"p={p} (row-split engine): OEM iterations {int(np.sum(fit['niter'][0]))}, all-gathers of the path phase {int(d[8])}; cycles of wave 0 of workgroup 0 per iteration\n"
          "  [operator + between | products + wave sums | LDS partials barrier + publish | gather | LDS stores + vote + barrier]")
    print("  path:   ", np.round(d[0:5] / it, 0), "sum", round(d[0:5].sum() / it))
    sys.exit(0)
lib.oemgpu_diag_read_symcoop.argtypes = [C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)(); assert lib.oemgpu_diag_read_symcoop(out) == 0
d = np.array(list(out), dtype=np.float64)
names = "between + owners' arithmetic of the previous | products | block sums + publish 1 | gather 1 | (alpha) + vote barrier | operator + publish 2 | gather 2 | barrier behind the LDS stores"
# (round 5: the wait behind the slowest wave's products and the LDS stores + ballots have slots of their own, 9 and 10: the per-wave lines below)
it = max(d[7], 1)
print(f"p={p}: OEM iterations {int(np.sum(fit['niter'][0]))}, all-reduces of the path phase {int(d[7])}; cycles of wave 0 of workgroup 0 per iteration\n  [{names}]")
print("  path:   ", np.round(d[8:16] / it, 0), "sum", round(d[8:16].sum() / it))
print("  Lanczos (totals over its steps, slot 7 overwritten):", np.round(d[0:7], 0))
lib.oemgpu_diag_read_symcoop_waves.argtypes = [C.POINTER(C.c_ulonglong)]
ow = (C.c_ulonglong * 64)(); assert lib.oemgpu_diag_read_symcoop_waves(ow) == 0
dw = np.array(list(ow), dtype=np.float64).reshape(4, 16) / it
print("  one level deeper, every wave of workgroup 0 (cycles per iteration): products | wait behind the slowest wave's products | block sums + publish 1 |"
      " gather 1 | vote barrier | operator + publish 2 | gather 2 | LDS stores + ballots | wait behind the slowest wave's gather")
for wv in range(4):
    r = dw[wv]
    print(f"    wave {wv}:", np.round([r[1], r[9], r[2], r[3], r[4], r[5], r[6], r[10], r[7]], 0), "sum", round(float(r[0] + r[1] + r[9] + r[2] + r[3] + r[4] + r[5] + r[6] + r[10] + r[7])))
print("  arrival spread inside the two gathers (cycles per iteration): gather 1 [first sweep -> the wave's FIRST pair | first pair -> its LAST pair], gather 2 [the same]")
for wv in range(4):
    r = dw[wv]
    print(f"    wave {wv}:", np.round([r[11], r[12]], 0), np.round([r[13], r[14]], 0))
