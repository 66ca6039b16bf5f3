#!/usr/bin/env python3
"""`accelerate = TRUE` at 1024 < p <= 2048, element-wise penalties: us per iteration on the one-exchange engine (path_rowcoop_kernel<ACC>) and on the
symmetric engine's general form it used to take (OEM_NO_ROWCOOP=1).  python tools/rowcoop_accel_time.py [p ...]"""
import os, sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch, oem_amd
from oem_amd import _lib as L
for p in [int(a) for a in sys.argv[1:]] or [1100, 1536, 2048]:
    g = torch.Generator(device="cuda"); g.manual_seed(p)
    n = 2 * p
    xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
    b = torch.zeros(p, dtype=torch.float64, device="cuda"); b[:25] = 1.0
    y = (xt.t() @ b + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)).contiguous()
    lib = L.lib(); ctx = oem_amd.context(); L.check(lib.oemgpu_set_timing(ctx, 1))
    for label, env, acc in (("accelerate, one-exchange engine", {}, True), ("accelerate, symmetric engine (OEM_NO_ROWCOOP=1)", {"OEM_NO_ROWCOOP": "1"}, True), ("plain, one-exchange engine", {}, False)):
        for k, v in env.items(): os.environ[k] = v
        best = 1e9
        for _ in range(2):
            fit = oem_amd.oem(xt.t(), y, penalty=["lasso", "mcp"], nlambda=20, tol=1e-8, lambda_min_ratio=0.01, accelerate=acc); torch.cuda.synchronize()
            ms = (C.c_double * L.NTIMERS)(); L.check(lib.oemgpu_last_timings(ctx, ms)); best = min(best, ms[L.T_EIGPATH])
        for k in env: del os.environ[k]
        st, cp = C.c_int32(-1), C.c_int32(-1); lib.oemgpu_last_eigen_info(ctx, C.byref(st), C.byref(cp))
        it = int(sum(np.sum(v) for v in fit["niter"])) + int(st.value)
        print(f"p={p} {label}: engine {oem_amd.last_path_engine()[0]}, eigen+path {best:.2f} ms, {it} iterations + Lanczos steps: {1e3 * best / it:.2f} us each", flush=True)
    del xt
