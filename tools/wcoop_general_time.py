#!/usr/bin/env python3
"""p >= n with group operators / Nesterov's step: the general form of the persistent cooperating engine (path_wcoop.hip) against the
launch-per-iteration general form (OEM_WCOOP_NO_GENERAL=1) -- eigen + path milliseconds, microseconds per iteration, agreement."""
import ctypes as C, os, sys, time, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
from oem_amd import _lib as L
warnings.simplefilter("ignore")
rng = np.random.default_rng(5); lib = L.lib()
for n, p, nlam, kw in ((500, 2000, 30, dict(penalty="grp.lasso", groups=np.arange(2000)//5+1)), (500, 2000, 30, dict(penalty="lasso", accelerate=True)),
                       (200, 5000, 20, dict(penalty="grp.lasso", groups=np.arange(5000)//10+1)), (500, 2000, 30, dict(penalty=["grp.lasso", "grp.mcp", "sparse.grp.lasso"], groups=np.arange(2000)//5+1))):
    x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    res = {}
    for mode in ("coop", "launches"):
        os.environ.pop("OEM_WCOOP_NO_GENERAL", None)
        if mode == "launches": os.environ["OEM_WCOOP_NO_GENERAL"] = "1"
        ctx = oem_amd.context(); L.check(lib.oemgpu_set_timing(ctx, 1))
        for _ in range(2):
            fit = oem_amd.oem(xd, y, nlambda=nlam, tol=1e-7, **kw); torch.cuda.synchronize()
            ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms))
        it = int(sum(np.sum(k) for k in fit["niter"])); res[mode] = fit
        print(n, p, {k: (v if isinstance(v, (str, bool, list)) else "...") for k, v in kw.items()}, mode, f"eigen+path {ms[3]:.1f} ms, {it} iterations, {1e3*ms[3]/it:.2f} us/iter", flush=True)
    a, b = res["coop"], res["launches"]
    print("    beta diff", max(np.abs(a["beta"][k] - b["beta"][k]).max() for k in range(len(a["beta"]))), "niter diff", max(np.abs(a["niter"][k].astype(int) - b["niter"][k].astype(int)).max() for k in range(len(a["niter"]))))
