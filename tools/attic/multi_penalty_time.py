#!/usr/bin/env python3
"""Config-2 shape (n = 5000, p = 200, 200 lambdas, tol 1e-10) with MCP and SCAD in ONE call: the penalties are independent cold
starts, so each gets a workgroup of its own in the same launch."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import oem_amd  # noqa: E402

rng = np.random.default_rng(123)
n, p, m = 5000, 200, 25
b = np.concatenate([rng.uniform(-0.5, 0.5, m), np.zeros(p - m)])
x = np.asfortranarray(rng.normal(size=(n, p)) * 3.0); y = x @ b + rng.normal(size=n)
xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


out = {}
for pens in (["mcp"], ["scad"], ["mcp", "scad"], ["lasso", "mcp", "scad", "elastic.net", "mcp.net", "scad.net"]):
    out["+".join(pens)] = timeit(lambda: oem_amd.oem(xd, y, penalty=pens, gamma=3.0, alpha=0.5, nlambda=200, tol=1e-10))
print(json.dumps(out))
