#!/usr/bin/env python3
"""Times BASELINE.json configs 2-5 (and config 1 host-resident) on one MI355X next to the CPU oracle.
Writes one JSON document (profiles/rNN_config_times.json is a copy of its output)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oem_amd
from oracle import oracle as orc


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()


out = {}
orc.lib(True)      # builds the -march=native oracle on THIS host before anything is timed
rng = np.random.default_rng(123)

# config 2: n=5000, p=200, MCP gamma=2 / SCAD gamma=4, 200 lambdas, tol 1e-10, standardize + intercept
n, p, m = 5000, 200, 25
b = np.concatenate([rng.uniform(-0.5, 0.5, m), np.zeros(p - m)])
x = np.asfortranarray(rng.normal(size=(n, p)) * 3.0); y = x @ b + rng.normal(size=n)
xd = dev(x)
for pen, gam in (("mcp", 2.0), ("scad", 4.0)):
    kw = dict(penalty=pen, gamma=gam, nlambda=200, tol=1e-10)
    t_gpu = timeit(lambda: oem_amd.oem(xd, y, **kw), 3)
    t0 = time.perf_counter(); ref = orc.fit_dense(x, y, native=True, **kw); t_cpu = time.perf_counter() - t0
    fit = oem_amd.oem(xd, y, **kw)
    out[f"config2_{pen}"] = {"gpu_ms": 1e3 * t_gpu, "cpu_port_1thread_ms": 1e3 * t_cpu, "iterations": int(fit["niter"][0].sum()),
                             "max_abs_err": float(np.abs(fit["beta"][0] - ref["beta"][0]).max()),
                             "reference_readme_ms": 105.9 if pen == "mcp" else 80.2}

# config 3: n=1e6, p=512, 64 groups of 8, grp.lasso, 100 lambdas, tol 1e-10, no intercept / standardize
n, p = 1_000_000, 512
g = torch.Generator(device="cuda"); g.manual_seed(3)
xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
bb = torch.zeros(p, dtype=torch.float64, device="cuda"); bb[:24] = torch.rand(24, generator=g, device="cuda", dtype=torch.float64) - 0.5
yd = (xt.t() @ bb + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)).contiguous()
groups = np.repeat(np.arange(1, 65), 8)
kw = dict(penalty="grp.lasso", groups=groups, nlambda=100, tol=1e-10, standardize=False, intercept=False)
t_gpu = timeit(lambda: oem_amd.oem(xt.t(), yd, **kw), 2)
fit = oem_amd.oem(xt.t(), yd, **kw)
from oem_amd import _lib as L
import ctypes as C
ctx = oem_amd.context()
L.check(L.lib().oemgpu_set_timing(ctx, 1)); oem_amd.oem(xt.t(), yd, **kw)
ms = (C.c_double * L.NTIMERS)(); L.check(L.lib().oemgpu_last_timings(ctx, ms)); L.check(L.lib().oemgpu_set_timing(ctx, 0))
flops = float(n) * p * (p + 1) + 2.0 * n * p
out["config3_grp_lasso_p512"] = {"gpu_ms": 1e3 * t_gpu, "gram_kernel_ms": ms[L.T_GRAMK], "gram_TFLOPs": flops / (ms[L.T_GRAMK] * 1e-3) / 1e12,
                                 "eigen_plus_path_ms": ms[L.T_EIGPATH], "iterations": int(fit["niter"][0].sum())}
del xt, yd

# config 4: oem.xtx p=4096, 100-lambda lasso, tol 1e-10
p, n = 4096, 65536
x = rng.normal(size=(n, p)); b = np.zeros(p); b[:25] = rng.uniform(-1, 1, 25); y = x @ b + rng.normal(size=n)
xtx, xty = x.T @ x / n, x.T @ y / n
xtxd = torch.as_tensor(xtx, device="cuda")
kw = dict(penalty="lasso", nlambda=100, tol=1e-10)
t_gpu = timeit(lambda: oem_amd.oem_xtx(xtxd, xty, **kw), 1)
fit = oem_amd.oem_xtx(xtxd, xty, **kw)
its = int(fit["niter"][0].sum())
its_c4, fit_c4_lambda = its, fit["lambda"][0]
out["config4_xtx_p4096"] = {"gpu_ms": 1e3 * t_gpu, "iterations": its, "bytes_per_iteration_symmetric_tile_engine": 4.0 * p * p + 4.0 * 128 * p + 8.0 * p * (p // 128),
                            "bytes_per_iteration_row_streaming": 8.0 * p * p + 24 * p,
                            "note": "includes ~140 Lanczos products; symmetric-tile engine: the lower triangle of XX once per iteration (DESIGN 3.3b)",
                            "approx_GBps_over_all_products": (its + 140) * (4.0 * p * p + 4.0 * 128 * p) / t_gpu / 1e9}
del xtxd, x

# config 5 semantics on one GPU: big.oem n=4e6 (of 1e8), p=256, lasso, device resident via the sharded driver
from oem_amd.distributed import HipBackend, oem_sharded
n, p = 4_000_000, 256
g.manual_seed(5)
xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
bb = torch.zeros(p, dtype=torch.float64, device="cuda"); bb[:20] = torch.rand(20, generator=g, device="cuda", dtype=torch.float64)
yd = (xt.t() @ bb + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)).contiguous()
be = HipBackend()
t_gpu = timeit(lambda: oem_sharded(xt.t(), yd, backend=be, big=True, penalty="lasso", nlambda=100, tol=1e-10), 2)
out["config5_big_p256_n4e6_one_gpu"] = {"gpu_ms": 1e3 * t_gpu, "rows": n, "note": "1/25 of the 1e8-row config; one rank's share at 8 GPUs is 1.25e7 rows"}
del xt, yd
# ... and the full share of one rank at 8 GPUs: 1.25e7 rows = 25.6 GB of X
n = 12_500_000
xt = torch.empty((p, n), device="cuda", dtype=torch.float64)
for j0 in range(0, p, 64):
    xt[j0:j0 + 64] = torch.randn((64, n), generator=g, device="cuda", dtype=torch.float64)
yd = torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
yd += torch.mv(xt.t(), bb)
lib_ = oem_amd.lib(); lib_.oemgpu_set_timing(be.ctx, 1)
t_gpu = timeit(lambda: oem_sharded(xt.t(), yd, backend=be, big=True, penalty="lasso", nlambda=100, tol=1e-10), 2)
import ctypes as C_
ms_ = (C_.c_double * 8)(); lib_.oemgpu_last_timings(be.ctx, ms_); lib_.oemgpu_set_timing(be.ctx, 0)
fl = float(n) * p * (p + 1) + 2.0 * n * p
out["config5_big_p256_n1.25e7_one_rank_share"] = {"gpu_ms": 1e3 * t_gpu, "rows": n, "gram_kernel_ms": ms_[4],
                                                  "gram_TFLOPs": fl / (ms_[4] * 1e-3) / 1e12 if ms_[4] > 0 else None,
                                                  "note": "25.6 GB of X resident in HBM; at 8 GPUs add one 532 KB all-reduce"}
del xt, yd

# config 1, host-resident X through oemgpu_fit_dense (PCIe inclusive)
n, p = 1_000_000, 100
x = np.asfortranarray(rng.normal(size=(n, p)) * 3.0); b = np.concatenate([rng.uniform(size=25), np.zeros(75)]); y = x @ b + rng.normal(size=n)
kw = dict(penalty="elastic.net", standardize=False, tol=1e-10)
t = timeit(lambda: oem_amd.oem(x, y, **kw), 2)
from oem_amd import _lib as L_
out["config1_host_resident"] = {"ms": 1e3 * t, "stats_last_call": L_.host_stats(),
                                "note": "oemgpu_fit_dense on pageable host x: 8 staging lanes -> pinned slots -> HBM, block moments, solve; cached contexts"}
for thr in (1, 2, 4, 8, 16):
    tt = timeit(lambda: oem_amd.oem(x, y, upload_threads=thr, **kw), 2)
    out["config1_host_resident"][f"ms_with_{thr}_lanes"] = 1e3 * tt
tt = timeit(lambda: oem_amd.oem(x, y, devices=[0, 0], **kw), 2)
out["config1_host_resident"]["ms_rows_over_two_contexts_of_one_device"] = 1e3 * tt

# next rows: xval.oem (f-1) on the same device-resident data, and oem() with p >= n (f-3)
xd = dev(x)
foldid = rng.permutation(np.resize(np.arange(1, 11), n))
t = timeit(lambda: oem_amd.xval_oem(xd, y, foldid=foldid, penalty="elastic.net", standardize=False, tol=1e-10), 3)
nc = 100_000
t0 = time.perf_counter()
orc.xval_dense(x[:nc], y[:nc], foldid[:nc], penalty=["elastic.net"], standardize=False, tol=1e-10, nlambda=100, lambda_min_ratio=1e-4,
               native=True)
t_cpu = time.perf_counter() - t0
out["xval_n1e6_p100_10folds"] = {"gpu_ms": 1e3 * t, "cpu_port_1thread_ms_on_1e5_rows": 1e3 * t_cpu,
                                 "note": "fold gather + 10 fold moments + 11 paths in one launch + MFMA CV error; X resident in HBM"}
del xd
n2, p2 = 500, 2000
xw = np.asfortranarray(rng.normal(size=(n2, p2))); yw = xw[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n2)
import warnings
warnings.simplefilter("ignore")
kw = dict(penalty="lasso", nlambda=50, tol=1e-7)
t = timeit(lambda: oem_amd.oem(xw, yw, **kw), 2)
t0 = time.perf_counter(); orc.fit_dense(xw, yw, lambda_min_ratio=0.01, native=True, **kw); t_cpu = time.perf_counter() - t0
out["wide_n500_p2000_lasso"] = {"gpu_ms": 1e3 * t, "cpu_port_1thread_ms": 1e3 * t_cpu,
                                "note": "p >= n: the reference's own iteration through the standardised X (wide engine, one read of X per iteration); OEM_NO_WIDE=1 runs the Gram form"}

# ---- CPU baselines for configs 3, 4, 5 (the oracle = the C restatement of the reference path, 1 thread = the reference's effective
# default; -O3 -march=native built on this host).  The Gram pass is linear in n and the path does not depend on n, so configs 3 and 5
# are timed at two sub-sampled n and extrapolated linearly, t(n) = a + b n; both points are reported.
orc.lib(True)
def cpu_two_points(make, fit, n_small, n_big, n_full):
    ts = []
    for nn in (n_small, n_big):
        xs, ys = make(nn)
        t0 = time.perf_counter(); fit(xs, ys); ts.append(time.perf_counter() - t0)
    b_ = (ts[1] - ts[0]) / (n_big - n_small); a_ = ts[0] - b_ * n_small
    return {"rows": [n_small, n_big], "seconds": ts, "extrapolated_full_seconds": a_ + b_ * n_full, "n_full": n_full,
            "model": "t(n) = a + b n through the two points", "threads": 1}
r3 = np.random.default_rng(33)
def make3(nn):
    xs = np.asfortranarray(r3.normal(size=(nn, 512))); bb_ = np.zeros(512); bb_[:24] = r3.uniform(-0.5, 0.5, 24)
    return xs, xs @ bb_ + r3.normal(size=nn)
g3 = np.repeat(np.arange(1, 65), 8)
out["config3_cpu_port"] = cpu_two_points(make3, lambda xs, ys: orc.fit_dense(xs, ys, native=True, penalty=["grp.lasso"], groups=g3, unique_groups=np.unique(g3),
                                                                            nlambda=100, tol=1e-10, standardize=False, intercept=False), 20000, 40000, 1_000_000)
def make5(nn):
    xs = np.asfortranarray(r3.normal(size=(nn, 256))); bb_ = np.zeros(256); bb_[:20] = r3.uniform(0, 1, 20)
    return xs, xs @ bb_ + r3.normal(size=nn)
out["config5_cpu_port"] = cpu_two_points(make5, lambda xs, ys: orc.fit_big(xs, ys, native=True, penalty=["lasso"], nlambda=100, tol=1e-7), 40000, 80000, 100_000_000)
# config 4: the path is the whole cost and each iteration is one dense 4096 x 4096 GEMV: time the first 12 lambdas of the 100-lambda grid
# and scale by the iteration counts the GPU fit reports for the full path
lam12 = np.asarray(fit_c4_lambda)[:12]
t0 = time.perf_counter(); r4 = orc.fit_xtx(xtx, xty, native=True, penalty=["lasso"], lambda_=lam12, tol=1e-10); t4 = time.perf_counter() - t0
it12 = int(np.sum(r4["niter"][0]))
out["config4_cpu_port"] = {"seconds_first_12_lambdas": t4, "iterations_first_12_lambdas": it12, "ms_per_iteration": 1e3 * t4 / max(it12, 1),
                           "iterations_full_path": its_c4, "extrapolated_full_seconds": t4 / max(it12, 1) * its_c4, "threads": 1,
                           "model": "ms per iteration (one 134 MB GEMV + threshold) x the full path's iteration count"}

# ---- config 5's per-GPU share from HOST memory (oemgpu_fit_big, one device): a 2e6 x 256 sample (4.1 GB) of the 1.25e7 x 256 share (25.6 GB)
n5 = 2_000_000
x5, y5 = make5(n5)
kw5 = dict(penalty="lasso", nlambda=100, tol=1e-7)
t5 = timeit(lambda: oem_amd.big_oem(x5, y5, **kw5), 2)
st5 = L_.host_stats()
out["config5_share_host_resident"] = {"rows": n5, "ms": 1e3 * t5, "GBps": 8.0 * n5 * 257 / t5 / 1e9, "stats_last_call": st5,
                                      "extrapolated_ms_for_1.25e7_rows": 1e3 * t5 * 12_500_000 / n5,
                                      "note": "upload-bound: linear in rows; at 8 GPUs each device streams its own share over its own PCIe link"}
print(json.dumps(out, indent=1))
