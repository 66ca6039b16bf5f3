#!/usr/bin/env python3
"""The Gram form beyond the register-resident engines (q > 4096): us per iteration of the launch-per-iteration engines and the bandwidth
that implies at 8 q^2 bytes per iteration (what a row-streaming product reads): python tools/large_q_time.py [p ...]"""
import sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import oem_amd
from oem_amd import _lib as L
for p in [int(a) for a in sys.argv[1:]] or [5000, 6144, 8192, 8193, 12288]:
    g = torch.Generator(device="cuda"); g.manual_seed(p)
    n = p + p // 2
    x = torch.randn((n, p), generator=g, device="cuda", dtype=torch.float64)
    b = torch.zeros(p, dtype=torch.float64, device="cuda"); b[:25] = 1.0
    y = x @ b + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
    xtx = (x.t() @ x) / n; xty = ((x.t() @ y) / n).cpu().numpy()
    del x
    lib = L.lib(); ctx = oem_amd.context()
    L.check(lib.oemgpu_set_timing(ctx, 1))
    # groups of 8: the operators in the head of the (head, product) pairs; of 50 / 90: the wider windows of that head; of 120: the update-kernel form
    cases = [(["lasso"], (), "")] + [(["grp.lasso"], np.arange(p) // gsz + 1, f" (groups of {gsz})") for gsz in ((8, 50, 90, 120) if p == 8192 else (8,))]
    for pens, grp, note in cases:
        best = 1e9
        for _ in range(2):
            fit = oem_amd.oem_xtx(xtx, xty, penalty=pens, groups=grp, nlambda=20, tol=1e-8, lambda_min_ratio=0.01); torch.cuda.synchronize()
            ms = (C.c_double * L.NTIMERS)(); L.check(lib.oemgpu_last_timings(ctx, ms)); best = min(best, ms[L.T_EIGPATH])
        st, cp = C.c_int32(-1), C.c_int32(-1); lib.oemgpu_last_eigen_info(ctx, C.byref(st), C.byref(cp))
        it = int(np.sum(fit["niter"][0])) + int(st.value)
        print(f"p={p} {pens[0]}{note}: {oem_amd.last_path_engine()[0]} eigen+path {best:.1f} ms, {it} products (OEM iterations + {int(st.value)} Lanczos steps): {1e3 * best / it:.1f} us each = "
              f"{8.0 * p * p * it / (best * 1e-3) / 1e12:.2f} TB/s at 8 q^2 bytes per product")
    del xtx
    torch.cuda.empty_cache()
