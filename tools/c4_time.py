#!/usr/bin/env python3
"""config 4 (oem.xtx, p = 4096, 100-lambda lasso, tol 1e-10): eigen + path milliseconds and microseconds per iteration."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
from oem_amd import _lib as L
rng = np.random.default_rng(123)
p, n = 4096, 65536
x = rng.normal(size=(n, p)); b = np.zeros(p); b[:25] = rng.uniform(-1, 1, 25); y = x @ b + rng.normal(size=n)
xtx, xty = x.T @ x / n, x.T @ y / n
xtxd = torch.as_tensor(xtx, device="cuda")
lib = L.lib(); ctx = oem_amd.context()
L.check(lib.oemgpu_set_timing(ctx, 1))
for knob in sys.argv[1:] or [""]:
    os.environ.pop("OEM_NO_SYM", None); os.environ.pop("OEM_NO_SYMCOOP", None)
    if knob == "nosym": os.environ["OEM_NO_SYM"] = "1"; os.environ["OEM_NO_SYMCOOP"] = "1"          # the row-streaming engine that reads all of XX (round 2)
    elif knob == "nosymcoop": os.environ["OEM_NO_SYMCOOP"] = "1"          # the symmetric-tile launches of round 3 (the lower triangle streamed per iteration)
    __import__('oem_amd')._lib.reload_switches()      # (the library parses its switches once)
    t = []
    for _ in range(3):
        fit = oem_amd.oem_xtx(xtxd, xty, penalty="lasso", nlambda=100, tol=1e-10)
        ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms)); t.append(ms[3])
    it = int(fit["niter"][0].sum())
    print(f"{knob or 'default'}: eigen+path {min(t):.2f} ms, {it} iterations, {1e3 * min(t) / it:.2f} us per iteration (incl. Lanczos), {8 * p * p * it / (min(t) * 1e-3) / 1e12:.2f} TB/s over the path counted at 8 p^2 bytes per iteration (the symmetric-tile engine reads 4 p^2 + 4 * 128 p: {(4 * p * p + 512 * p) * it / (min(t) * 1e-3) / 1e12:.2f} TB/s)")
