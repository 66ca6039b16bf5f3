#!/usr/bin/env python3
"""oem.xtx with scale.factor (the iterate rescaled in place at every lambda, ref src/oem_xtx.h:576-581) on the register-resident symmetric
engine (round 5) against the launch-per-iteration engines that served it before: python tools/scale_factor_time.py [p ...]"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import oem_amd
from oem_amd import _lib as L
import ctypes as C
for p in [int(a) for a in sys.argv[1:]] or [4096, 3000, 2048]:
    rng = np.random.default_rng(123)
    n = max(2 * p, 8192)
    x = rng.normal(size=(n, p)); b = np.zeros(p); b[:25] = rng.uniform(-1, 1, 25); y = x @ b + rng.normal(size=n)
    xtx = torch.as_tensor(x.T @ x / n, device="cuda"); xty = x.T @ y / n
    sf = rng.uniform(0.5, 2.0, p)
    lib = L.lib(); ctx = oem_amd.context()
    L.check(lib.oemgpu_set_timing(ctx, 1))
    res = {}
    for mode in ("symcoop", "launches"):
        os.environ.pop("OEM_NO_SYMCOOP", None)
        if mode == "launches": os.environ["OEM_NO_SYMCOOP"] = "1"
        best = 1e9
        for _ in range(3):
            fit = oem_amd.oem_xtx(xtx, xty, penalty="lasso", nlambda=100, tol=1e-10, scale_factor=sf); torch.cuda.synchronize()
            ms = (C.c_double * L.NTIMERS)(); L.check(lib.oemgpu_last_timings(ctx, ms)); best = min(best, ms[L.T_EIGPATH])
        res[mode] = (best, fit, oem_amd.last_path_engine()[0])
    os.environ.pop("OEM_NO_SYMCOOP", None)
    a, bfit = res["symcoop"], res["launches"]
    it = int(np.sum(a[1]["niter"][0]))
    print(f"p={p} oem.xtx lasso 100 lambdas tol 1e-10 WITH scale.factor: {a[2]} {a[0]:.2f} ms ({1e3 * a[0] / it:.2f} us per iteration, {it} iterations) | "
          f"{bfit[2]} {bfit[0]:.2f} ms ({1e3 * bfit[0] / int(np.sum(bfit[1]['niter'][0])):.2f} us) | max|dbeta| {np.abs(np.asarray(a[1]['beta'][0]) - np.asarray(bfit[1]['beta'][0])).max():.2e}, "
          f"max|dniter| {np.abs(a[1]['niter'][0].astype(int) - bfit[1]['niter'][0].astype(int)).max()}")
