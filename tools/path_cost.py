#!/usr/bin/env python3
"""Cost model of the small-p path kernel from whole-kernel cycle counts (no stamps, the product library):
cycles(nlambda, maxit) = lanczos + nlambda * per_lambda + rounds * per_round, fitted from a few runs.

    python tools/path_cost.py [p] [penalty]
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oem_amd as oa  # noqa: E402
from oem_amd import _lib as L  # noqa: E402
from oem_amd import api  # noqa: E402

p = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pen = sys.argv[2] if len(sys.argv) > 2 else "elastic.net"
n = 20000
rng = np.random.default_rng(123)
b = np.concatenate([rng.uniform(size=p // 4), np.zeros(p - p // 4)])
x = np.asfortranarray(rng.normal(size=(n, p)) * 3.0)
y = x @ b + rng.normal(size=n)
lib = L.lib()
import torch  # noqa: E402
from oem_amd.distributed import HipBackend, oem_sharded  # noqa: E402
backend = HipBackend(0)
xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
yd = torch.as_tensor(y, device="cuda")


def cycles(**kw):
    ctx = backend.ctx
    L.check(lib.oemgpu_set_timing(ctx, 1))
    best = None
    for _ in range(3):
        fit = oem_sharded(xd, yd, backend=backend, penalty=pen, intercept=True, standardize=False, **kw)
        ms = (C.c_double * L.NTIMERS)()
        L.check(lib.oemgpu_last_timings(ctx, ms))
        c = ms[6]
        best = c if best is None else min(best, c)
    return best, int(np.sum(np.minimum(fit["niter"][0], kw.get("maxit", 500))))


lam = oa.oem(x, y, penalty=pen, intercept=True, standardize=False)["lambda"][0]
rows = []
for nl, maxit in ((1, 1), (100, 1), (100, 2), (100, 4), (100, 8), (50, 8)):
    c, r = cycles(lambda_=lam[:nl], maxit=maxit, tol=1e-300)
    rows.append((nl, maxit, r, c))
    print(f"nlambda={nl:4d} maxit={maxit:2d} rounds={r:5d} cycles={c:10.0f}")
A = np.array([[1.0, nl, r] for nl, _, r, _ in rows]); bvec = np.array([c for *_, c in rows])
sol, *_ = np.linalg.lstsq(A, bvec, rcond=None)
print(f"p={p} {pen}: lanczos+fixed {sol[0]:.0f} cycles, per lambda {sol[1]:.0f}, per round {sol[2]:.0f}")
