#!/usr/bin/env python3
"""The sparse-design example of man/oem.Rd (R/oem.R:103-123): x 2.5e5 x 200 at density 0.01, lasso + grp.lasso, no intercept, no
standardisation; the dense and the sparse copy must agree to rounding (the rendered docs: 1.58e-15 / 1.61e-15; 25.91 s dense,
0.56 s sparse on their machine)."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oem_amd  # noqa: E402
from oracle import oracle as orc  # noqa: E402

rng = np.random.default_rng(123)
n, p = 250_000, 200
xs = sp.random(n, p, density=0.01, random_state=7, format="csc", data_rvs=lambda k: rng.normal(size=k))
tb = np.concatenate([rng.uniform(-0.25, 0.25, 15), np.zeros(p - 15)])
ys = rng.normal(size=n) * 3.0 + xs @ tb
xd = np.asfortranarray(xs.toarray())
groups = np.repeat(np.arange(1, 41), 5)
kw = dict(penalty=["lasso", "grp.lasso"], groups=groups, intercept=False, standardize=False)


def timeit(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return 1e3 * (time.perf_counter() - t0) / reps, r


import os
t_dense, fit = timeit(lambda: oem_amd.oem(xd, ys, **kw))
t_sparse, fits = timeit(lambda: oem_amd.oem(xs, ys, lambda_=fit["lambda"], **kw))           # density 1 %: the compressed-column Gram
os.environ["OEM_SPARSE_GRAM"] = "dense"
__import__('oem_amd')._lib.reload_switches()      # (the library parses its switches once)
t_sparse_tiles, fitt = timeit(lambda: oem_amd.oem(xs, ys, lambda_=fit["lambda"], **kw))     # zero-filled tiles + MFMA pass
del os.environ["OEM_SPARSE_GRAM"]
__import__('oem_amd')._lib.reload_switches()      # (the library parses its switches once)
other = {}
for dens in (0.001, 0.05):                                                                  # both ways at other densities
    xo = sp.random(n, p, density=dens, random_state=8, format="csc", data_rvs=lambda k: rng.normal(size=k))
    yo = rng.normal(size=n) + xo @ tb
    r = {}
    for mode in ("csc", "dense"):
        os.environ["OEM_SPARSE_GRAM"] = mode
        __import__('oem_amd')._lib.reload_switches()      # (the library parses its switches once)
        r[mode + "_ms"], f_ = timeit(lambda: oem_amd.oem(xo, yo, **kw))
        r[mode] = f_
    del os.environ["OEM_SPARSE_GRAM"]
    __import__('oem_amd')._lib.reload_switches()      # (the library parses its switches once)
    other[str(dens)] = {"csc_ms": r["csc_ms"], "dense_tiles_ms": r["dense_ms"],
                        "max_abs_diff": float(max(np.abs(r["csc"]["beta"][k] - r["dense"]["beta"][k]).max() for k in range(2)))}
t0 = time.perf_counter()
ref = orc.fit_sparse(xs, ys, unique_groups=np.unique(groups), lambda_=fit["lambda"], native=True, **kw)
t_cpu = 1e3 * (time.perf_counter() - t0)
print(json.dumps({"n": n, "p": p, "nnz": int(xs.nnz), "gpu_dense_host_x_ms": t_dense, "gpu_sparse_host_x_ms": t_sparse,
                  "gpu_sparse_host_x_dense_tiles_ms": t_sparse_tiles, "other_densities_host_x": other,
                  "max_abs_csc_vs_tiles": [float(np.abs(fitt["beta"][k] - fits["beta"][k]).max()) for k in range(2)],
                  "cpu_oracle_sparse_1thread_ms": t_cpu,
                  "max_abs_dense_vs_sparse": [float(np.abs(fit["beta"][k] - fits["beta"][k]).max()) for k in range(2)],
                  "max_abs_sparse_vs_oracle": [float(np.abs(ref["beta"][k] - fits["beta"][k]).max()) for k in range(2)]}))
