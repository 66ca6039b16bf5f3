#!/usr/bin/env python3
"""The xval.oem documentation example (man/xval.oem.Rd, docs/reference/xval.oem.html): n = 1e4, p = 100, 10 folds, 100 lambdas,
2 and then 11 penalties in one call (the rendered docs report 14.08 s and 67.78 s of user time on their machine)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oem_amd  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle.r_rng import RRng  # noqa: E402

r = RRng(123)
n, p = 10000, 100
tb = np.concatenate([r.runif(15, -0.25, 0.25), np.zeros(p - 15)])
x = np.asfortranarray(r.rnorm(n * p).reshape(p, n).T)
y = r.rnorm(n, sd=3) + x @ tb
foldid = r.sample(np.resize(np.arange(1, 11), n))
groups = np.repeat(np.arange(1, 21), 5)
two = ["lasso", "grp.lasso"]
eleven = ["lasso", "grp.lasso", "mcp", "scad", "mcp.net", "scad.net", "grp.lasso", "grp.lasso.net", "grp.mcp", "grp.scad", "sparse.grp.lasso"]
out = {}
for name, pens in (("2 penalties", two), ("11 penalties", eleven)):
    f = oem_amd.xval_oem(x, y, foldid=foldid, penalty=pens, groups=groups)
    t0 = time.perf_counter()
    f = oem_amd.xval_oem(x, y, foldid=foldid, penalty=pens, groups=groups)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    g = orc.xval_dense(x, y, foldid, penalty=pens, groups=np.concatenate([[0], groups]), unique_groups=np.arange(0, 21), nlambda=100,
                       lambda_min_ratio=1e-4, tol=1e-7, maxit=500, native=True)
    t_cpu = time.perf_counter() - t0
    err = max(float(np.abs(f["cvm"][k] - g["cvm"][k]).max() / np.abs(g["cvm"][k]).max()) for k in range(len(pens)))
    out[name] = {"gpu_ms_host_x": 1e3 * t_gpu, "cpu_oracle_1thread_ms": 1e3 * t_cpu, "max_rel_cvm_diff": err, "best_model": f["best.model"]}
print(json.dumps(out))
