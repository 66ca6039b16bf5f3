#!/usr/bin/env python3
"""Time xval.oem (oemgpu_xval_dense_dev) on a device-resident X of the README shape: n x 100, 10 folds, 100 lambdas.

    python tools/xval_time.py [n] [reps] [--cpu N_CPU]     (--cpu: also time the oracle on the first N_CPU rows)
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import oem_amd as oa  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(float(args[0])) if args else 1_000_000
reps = int(args[1]) if len(args) > 1 else 5
ncpu = int(float(sys.argv[sys.argv.index("--cpu") + 1])) if "--cpu" in sys.argv else 0
p, K = 100, 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(123)
xt = torch.randn((p, n), generator=g, device=dev, dtype=torch.float64) * 3.0
b = torch.cat([torch.rand(25, generator=g, device=dev, dtype=torch.float64), torch.zeros(75, device=dev, dtype=torch.float64)])
y = xt.t() @ b + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
foldid = np.random.default_rng(5).permutation(np.resize(np.arange(1, K + 1), n))
x = xt.t()
kw = dict(foldid=foldid, penalty="elastic.net", alpha=1.0, intercept=True, standardize=False, tol=1e-10)
fit = oa.xval_oem(x, y, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    fit = oa.xval_oem(x, y, **kw)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(reps):
    one = oa.oem(x, y, penalty="elastic.net", alpha=1.0, intercept=True, standardize=False, tol=1e-10)
torch.cuda.synchronize()
ms_fit = 1e3 * (time.perf_counter() - t0) / reps
out = {"n": n, "p": p, "nfolds": K, "nlambda": 100, "xval_ms": ms, "single_fit_ms": ms_fit, "iters_full_fit": int(np.sum(fit["niter"][0])),
       "cvm_min": float(np.min(fit["cvm"][0])), "lambda_min": float(fit["lambda.min"])}
if ncpu:
    from oracle import oracle as orc
    xh = np.asfortranarray(x[:ncpu].cpu().numpy()); yh = y[:ncpu].cpu().numpy()
    t0 = time.perf_counter()
    r = orc.xval_dense(xh, yh, foldid[:ncpu], penalty=["elastic.net"], alpha=1.0, intercept=True, standardize=False, tol=1e-10,
                       nlambda=100, lambda_min_ratio=1e-4, native=True)
    out["cpu_oracle_s"] = time.perf_counter() - t0
    out["cpu_rows"] = ncpu
print(json.dumps(out))
