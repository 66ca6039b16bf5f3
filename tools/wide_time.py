#!/usr/bin/env python3
"""oem() with p >= n: the wide engine (the reference's two-product iteration through the standardised X, one
read of X per iteration) against the Gram form of the same iteration.  Prints milliseconds per call (host x), eigen + path
milliseconds, iterations, microseconds per iteration and the rate over 8 n p bytes per iteration."""
import ctypes as C, os, sys, time, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
from oem_amd import _lib as L
warnings.simplefilter("ignore")
rng = np.random.default_rng(5)
lib = L.lib()
for n, p, nlam, gram in ((500, 2000, 50, True), (500, 20000, 50, False), (2000, 20000, 20, False), (200, 100000, 20, False), (4000, 50000, 10, False)):
    x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    kw = dict(penalty="lasso", nlambda=nlam, tol=1e-7)
    for mode in (["wide", "gram"] if gram else ["wide"]):
        os.environ.pop("OEM_NO_WIDE", None)
        if mode == "gram": os.environ["OEM_NO_WIDE"] = "1"
        ctx = oem_amd.context()
        L.check(lib.oemgpu_set_timing(ctx, 1))
        best, th = 1e9, 1e9
        for _ in range(3):
            t0 = time.perf_counter(); fit = oem_amd.oem(xd, y, **kw); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms))
        eng = oem_amd.last_path_engine()[0]                       # (registers: wcoop / wres / rowcoop ...; wlaunches: X streamed every iteration)
        for _ in range(2):                                        # (the first host call of a shape allocates its staging and device buffers)
            t0 = time.perf_counter(); oem_amd.oem(x, y, **kw); th = min(th, time.perf_counter() - t0)
        it = int(fit["niter"][0].sum())
        byt = 8.0 * (64 * ((n + 63) // 64)) * p * (2 if n > 2048 else 1)      # beyond 2048 rows: two passes over Xs per iteration
        print(f"n={n} p={p} {nlam} lambdas [{mode}, engine {eng}]: resident {1e3 * best:.1f} ms (stage reading X {ms[1]:.2f} ms, eigen + path {ms[3]:.1f} ms), "
              f"host x {1e3 * th:.1f} ms; {it} iterations, {1e3 * ms[3] / it:.2f} us per iteration"
              + (f", {byt * it / (ms[3] * 1e-3) / 1e12:.2f} TB/s over {'two passes over Xs per iteration (the X beta pass skips columns whose coefficient is zero, so the rate can exceed the HBM peak)' if n > 2048 else 'one read of Xs per iteration'}" if mode == "wide" else ""), flush=True)
