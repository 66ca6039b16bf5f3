#!/usr/bin/env python3
"""path_symcoop.hip (1024 < p <= 4096, the lower triangle of XX in registers) against the launch-per-iteration engines
(OEM_NO_SYMCOOP=1): coefficients, iteration counts, d, and eigen + path milliseconds.   python tools/symcoop_check.py [p ...]"""
import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
from oem_amd import _lib as L
lib = L.lib(); ctx = oem_amd.context()
L.check(lib.oemgpu_set_timing(ctx, 1))
ps = [int(a) for a in sys.argv[1:]] or [1088, 2048, 3000, 4096]
for p in ps:
    rng = np.random.default_rng(p)
    n = max(2 * p, 8192)
    x = rng.normal(size=(n, p)); b = np.zeros(p); b[rng.choice(p, 25, replace=False)] = rng.uniform(-1, 1, 25); y = x @ b + rng.normal(size=n)
    xtx, xty = x.T @ x / n, x.T @ y / n
    xd = torch.as_tensor(xtx, device="cuda")
    res = {}
    grp = os.environ.get("SYMCOOP_CHECK_GROUPS")                  # SYMCOOP_CHECK_GROUPS=8: group penalties (groups of that many neighbours) instead
    pkw = dict(penalty=["grp.lasso", "grp.mcp"], groups=np.arange(p) // int(grp) + 1) if grp else dict(penalty=["lasso", "mcp"])
    for name in ("symcoop", "launches", "default"):               # "default": the row-split one-exchange engine where it applies
        os.environ.pop("OEM_NO_SYMCOOP", None); os.environ.pop("OEM_NO_ROWCOOP", None)
        if name == "launches": os.environ["OEM_NO_SYMCOOP"] = "1"
        if name == "symcoop": os.environ["OEM_NO_ROWCOOP"] = "1"
        t = []
        for _ in range(2):
            t0 = time.perf_counter()
            fit = oem_amd.oem_xtx(xd, xty, nlambda=20, tol=1e-10, **pkw)
            wall = (time.perf_counter() - t0) * 1e3
            ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms)); t.append((ms[3], wall))
        res[name] = (fit, min(t))
    a, b_ = res["symcoop"][0], res["launches"][0]
    its = sum(int(np.sum(v)) for v in a["niter"])
    err = max(float(np.abs(np.asarray(a["beta"][k]) - np.asarray(b_["beta"][k])).max()) for k in range(2))
    dn = max(int(np.abs(np.ravel(a["niter"][k]).astype(int) - np.ravel(b_["niter"][k]).astype(int)).max()) for k in range(2))
    print(f"p={p}: max|dbeta|={err:.2e} max|dniter|={dn} d rel diff={abs(a['d'] - b_['d']) / b_['d']:.2e} iterations={its} "
          f"symcoop {res['symcoop'][1][0]:.2f} ms (wall {res['symcoop'][1][1]:.2f}) = {1e3 * res['symcoop'][1][0] / its:.2f} us/it | launches {res['launches'][1][0]:.2f} ms = {1e3 * res['launches'][1][0] / its:.2f} us/it", flush=True)
    if p <= 2048 and not grp:
        c = res["default"][0]
        e2 = max(float(np.abs(np.asarray(c["beta"][k]) - np.asarray(b_["beta"][k])).max()) for k in range(2))
        i2 = sum(int(np.sum(v)) for v in c["niter"])
        print(f"        row-split one-exchange engine (the default here): {res['default'][1][0]:.2f} ms = {1e3 * res['default'][1][0] / i2:.2f} us/it, "
              f"max|dbeta| vs launches {e2:.2e}, iterations {i2}", flush=True)
