#!/usr/bin/env python3
"""bench.py -- full-lambda-path solves/sec on BASELINE.json config 1 (the reference's README benchmark).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (README.md:44-66 of the reference): n = 1e6, p = 100, X ~ N(0, 3^2) iid, b = [U(0,1) x 25, 0 x 75],
y = X b + N(0,1); oem(penalty="elastic.net", alpha=1) (= lasso), intercept, no standardisation, the 100 lambdas of
a first default fit supplied back, tol = 1e-10.  Synthetic data, generated on the device, resident in HBM when
the timed region starts.

Before the W warmup steps the bench runs a fixed 400 untimed solves (0.2 s: the GPU's clocks settle over milliseconds; `prewarm_solves` in the line).
A "step" is one complete solve: one-pass MFMA moment build (about 0; a shifted redo only if finalize asks) -> (N > 1: all-reduce) -> finalize ->
eigenvalue -> 100-lambda path -> results on the host.  With N > 1 the n rows are split across the ranks (strong
scaling: the total problem stays n = 1e6) and the (p+2)^2 moment buffer is summed with one RCCL all-reduce
(oem_amd/distributed.py: solve_row_shards).

`python bench.py --gpus N` WITHOUT a torchrun environment launches its own N workers: the parent makes no GPU call, starts
`python -m torch.distributed.run ... bench.py --gpus N ...` as a fresh child process and exits with its code (fewer than N devices:
one line on stderr, exit code 2).  The N > 1 line carries `rccl_ranks` (the world size RCCL saw), the all-reduce's own time, and
`host_resident_ms` at opts.ngpus = N -- the in-library multi-GPU path an R caller gets (rank 0, after the process group is gone).

Rank 0 prints ONE JSON line (see the repo README / DESIGN.md for the roofline and cpu_baseline fields).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix peak (AMD spec; 32 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz)
# config 4's iteration on the register-resident symmetric engine, as far as THIS design goes: two memory-side
# exchanges (1.05 us per all-gather at the hop floor, profiles/r4_exchange_probes.txt) + the products at the issue floor of an
# accumulator-fed tile (tools/areg_fma_probe.hip, profiles/r5_c4_areg_fma_probe.txt: 3 tiles per wave) + the owners' arithmetic
C4_FLOOR_US = 4.6
README_SECONDS = 1.600241        # reference README.md:73, oem[lasso] mean over 5 runs, hardware unstated


def c5_weak(torch, dist, world, rank, dev, backend, rows, steps, warmup):
    """BASELINE.json config 5 as its per-GPU share: big.oem() lasso, 100 lambdas, p = 256, `rows` rows on EVERY rank (weak
    scaling: n = world * rows), X ~ N(0,1) generated on the device, one all-reduce of the (p+2)^2 moment buffer."""
    import numpy as np
    from oem_amd import _lib as L
    from oem_amd import api
    from oem_amd.distributed import sharded_buffers, solve_row_shards
    p, m = 256, 25
    g = torch.Generator(device=dev); g.manual_seed(4242)
    b = torch.cat([torch.rand(m, generator=g, device=dev, dtype=torch.float64), torch.zeros(p - m, device=dev, dtype=torch.float64)])
    g.manual_seed(5000 + rank)
    xt = torch.empty((p, rows), device=dev, dtype=torch.float64)
    for j0 in range(0, p, 32):                                         # 32 columns at a time: no second 25.6 GB temporary
        xt[j0:j0 + 32].normal_(generator=g)
    x = xt.t()
    y = torch.randn(rows, generator=g, device=dev, dtype=torch.float64)
    for j0 in range(0, m, 5):
        y += x[:, j0:j0 + 5] @ b[j0:j0 + 5]
    torch.cuda.synchronize()
    q = p + 1
    args = api._Args(["lasso"], [], 100, 1e-4, 1.0, 3.0, 0.5, 1e-7, 500, False, False, np.ones(p),
                     np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    bufs = sharded_buffers(backend, p)
    outs = args.outputs(q)
    dd = dist if world > 1 else None

    def solve():
        solve_row_shards(backend, dd, None, x, rows, rows, p, y, bufs, L.OEMGPU_SEM_BIG, True, True, args, outs)
    with backend.section():
        for _ in range(warmup):
            solve()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with backend.section():
        for _ in range(steps):
            solve()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    lib = L.lib()
    L.check(lib.oemgpu_set_timing(backend.ctx, 1))
    acc = np.zeros(L.NTIMERS)
    ms = (C.c_double * L.NTIMERS)()
    with backend.section():
        for _ in range(5):
            solve()
            L.check(lib.oemgpu_last_timings(backend.ctx, ms))
            acc += np.array(list(ms))
    L.check(lib.oemgpu_set_timing(backend.ctx, 0))
    acc /= 5
    # the collective on its own: the (p+2)^2 moment buffer of this workload (HIP events on the stream the collectives are ordered on)
    allreduce_ms = allgather_ms = None
    if world > 1:
        ex = time_exchanges(torch, dist, backend, bufs[1])
        allreduce_ms, allgather_ms = ex["allreduce_ms"], ex["allgather_ms"]
    flops = float(rows) * p * (p + 1) + 2.0 * rows * p
    nit = int(np.sum(args.niter))
    del x, xt, y
    torch.cuda.empty_cache()
    return {"workload": "config 5 share: big.oem() lasso, p=256, 100 lambdas, tol 1e-7 (default), %d rows per GPU, n = %d" % (rows, rows * world),
            "value": steps / dt, "unit": "solves/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "scaling": "weak",
            "rows_per_gpu": rows, "n_total": rows * world, "rows_per_second": rows * world * steps / dt,
            "stage_ms": {"moments_total": acc[L.T_MOMENTS], "gram_kernel": acc[L.T_GRAMK], "finalize": acc[L.T_FINAL],
                         "eigen_plus_path": acc[L.T_EIGPATH]},
            "allreduce_ms": allreduce_ms, "allgather_ms": allgather_ms, "allreduce_doubles": (p + 2) * (p + 2),
            "gram_TFLOPs": flops / (acc[L.T_GRAMK] * 1e-3) / 1e12 if acc[L.T_GRAMK] > 0 else None,
            "gram_frac_of_fp64_mfma_peak": flops / (acc[L.T_GRAMK] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if acc[L.T_GRAMK] > 0 else None,
            "oem_iterations_per_solve": nit}


def time_exchanges(torch, dist, backend, buf, reps=50):
    """The two ways of summing a moment buffer over the ranks, each on its own (HIP events on the stream the collectives are ordered on):
    dist.all_reduce, and the default of oem_amd/distributed.py -- one all-gather + one kernel that adds the buffers in rank order."""
    from oem_amd.distributed import sum_over_ranks
    res = {}
    for name, mode in (("allreduce_ms", "allreduce"), ("allgather_ms", "ordered")):
        scratch = buf.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with backend.section():
            for _ in range(5):
                sum_over_ranks(backend, dist, None, scratch, mode)
            e0.record()
            for _ in range(reps):
                sum_over_ranks(backend, dist, None, scratch, mode)
            e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps
    return res


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(a, argv):
    """--gpus N without WORLD_SIZE in the environment: N fresh worker processes under torch.distributed.run.  This (parent)
    process never touches the GPU -- torch.cuda.device_count() does not initialise it on this image -- and never execs."""
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if ndev < a.gpus and os.environ.get("OEM_BENCH_ONE_DEVICE") != "1":
        print(f"bench.py: --gpus {a.gpus} asked for, but this node shows {ndev} GPU(s): nothing run", file=sys.stderr)
        sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(Path(__file__).resolve())]
    # the options travel in the environment: torch.distributed.run's own parser trips over script options it can prefix-match
    # (`--n` is "ambiguous" between --nnodes, --nproc-per-node, ...) even behind the script's path
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", OEM_BENCH_ARGV=json.dumps(list(argv)))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def live_pmc(kernel_substring, counter_sets, timeout=180, program=None):
    """Hardware counters of one kernel, measured in THIS run: each counter set is one child process
    `rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 3 ...` (counters in passes of their own, the
    program itself behind `--`, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; a child process, never an exec).  Returns
    {counter: mean value per dispatch of the kernels whose name contains `kernel_substring`}; raises on any failure (the caller
    falls back to the figures under profiles/)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    out = {}
    for ctrs in counter_sets:
        tmp = tempfile.mkdtemp(prefix="oem_pmc_", dir="/tmp")
        try:
            prog = program or [str(Path(__file__).resolve()), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-c5", "--no-host",
                               "--no-two-callers", "--no-rccl-check", "--no-live-pmc"]
            cmd = ["rocprofv3", "--pmc"] + list(ctrs) + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--", sys.executable] + prog
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OEM_BENCH_ARGV")}
            env["TMPDIR"] = "/tmp"
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            if r.returncode != 0:
                raise RuntimeError("rocprofv3 pass %s exited with %d" % (ctrs, r.returncode))
            acc = {}
            for path in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
                with open(path, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if kernel_substring in row["Kernel_Name"]:
                            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for c in ctrs:
                if c not in acc:
                    raise RuntimeError("counter %s not reported for %s" % (c, kernel_substring))
                out[c] = sum(acc[c]) / len(acc[c])
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def gen_c1(torch, dev, n, p, m, lo, hi, rank, world):
    """Rows [lo, hi) of the README-shaped problem on `dev`: (x as an (hi - lo, p) column-major view, y).  The SAME n x p problem for
    every N in {1, 2, 4, 8}: the rows come in 8 blocks, block k from its own seed, and a rank generates exactly the blocks of its
    row range -- so iteration counts (and the path's share of the time) do not move with N."""
    n_loc = hi - lo
    g = torch.Generator(device=dev); g.manual_seed(123)
    b = torch.cat([torch.rand(m, generator=g, device=dev, dtype=torch.float64), torch.zeros(p - m, device=dev, dtype=torch.float64)])
    xt = torch.empty((p, n_loc), device=dev, dtype=torch.float64)
    y = torch.empty(n_loc, device=dev, dtype=torch.float64)
    NBLK = 8
    if n % NBLK == 0 and NBLK % world == 0:
        blk = n // NBLK
        for k in range(lo // blk, hi // blk):
            g.manual_seed(1000 + k)
            c0 = k * blk - lo
            xt[:, c0:c0 + blk] = torch.randn((p, blk), generator=g, device=dev, dtype=torch.float64) * 3.0
            y[c0:c0 + blk] = torch.randn(blk, generator=g, device=dev, dtype=torch.float64)
    else:
        g.manual_seed(1000 + rank)
        xt.copy_(torch.randn((p, n_loc), generator=g, device=dev, dtype=torch.float64) * 3.0)
        y.copy_(torch.randn(n_loc, generator=g, device=dev, dtype=torch.float64))
    x = xt.t()                                   # (n_loc, p), stride (1, n_loc)
    y += x @ b
    torch.cuda.synchronize()
    return x, y


def host_resident(xh, yh, lambdas, p, ngpus=1, calls=5, devices=None):
    """The drop-in level: oemgpu_fit_dense on PAGEABLE host x / y, exactly what `.Call("oem_fit_dense")` hands over
    (staged upload through pinned lanes + moments + solve; contexts and buffers cached by the library).  ngpus > 1: the rows
    split over that many devices INSIDE the library (opts.ngpus: one host thread, staging pipeline and PCIe link per device, the
    moment buffers summed on the first).  Never `value`."""
    import numpy as np
    from oem_amd import _lib as L
    from oem_amd import api
    lib = L.lib()
    n = xh.shape[0]
    args = api._Args(["elastic.net"], [np.asarray(lambdas)], 100, 1e-4, 1.0, 3.0, 0.5, 1e-10, 500, False, False, np.ones(p),
                     np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), ngpus=ngpus if ngpus > 1 else 0, devices=devices)
    outs = args.outputs(p + 1)
    ts, st = [], None
    for k in range(calls + 1):
        t0 = time.perf_counter()
        L.check(lib.oemgpu_fit_dense(api._dptr(xh), n, p, api._dptr(yh), 0, 1, C.byref(args.c), *outs))
        ts.append(1e3 * (time.perf_counter() - t0))
        st = L.host_stats()
    first, ts = ts[0], ts[1:]
    gb = 8.0 * (n * p + n) / 1e9
    return {"what": "oemgpu_fit_dense(host x, host y): pageable rows -> pinned bounce slots -> HBM, MFMA moments per row block, solve",
            "ngpus": int(st["devices"]), "first_call_ms": first, "min_ms": float(np.min(ts)), "median_ms": float(np.median(ts)), "calls": calls,
            "GBps_at_median": gb / (float(np.median(ts)) * 1e-3), "upload_threads_per_device": 8,
            "steady_state_allocations": st["allocations"], "row_blocks": st["row_blocks"],
            "upload_plus_moments_ms": st["upload_moments_ms"], "solve_ms": st["solve_ms"],
            "handovers_staged_through_host": st["host_staged_handovers"]}, args


def host_resident_c5(torch, dev, ngpus, rows_per_device=2_000_000, calls=3, devices=None):
    """A sample of config 5's per-GPU share from PAGEABLE host memory through oemgpu_fit_big: `rows_per_device` x 256 rows per
    device (4.1 GB each; the full share is 1.25e7), opts.ngpus devices inside the library.  Every device streams the same host
    block (it is handed over as `ngpus` row shards): the measurement is the staging pipelines, one PCIe link each."""
    import numpy as np
    from oem_amd import _lib as L
    from oem_amd import api
    p, m = 256, 25
    g = torch.Generator(device=dev); g.manual_seed(777)
    xt = torch.randn((p, rows_per_device), generator=g, device=dev, dtype=torch.float64)
    b = torch.cat([torch.rand(m, generator=g, device=dev, dtype=torch.float64), torch.zeros(p - m, device=dev, dtype=torch.float64)])
    y = xt.t() @ b + torch.randn(rows_per_device, generator=g, device=dev, dtype=torch.float64)
    xh = xt.t().cpu().numpy(); yh = y.cpu().numpy()
    del xt, y
    torch.cuda.empty_cache()
    assert xh.flags.f_contiguous
    G = max(1, ngpus)
    args = api._Args(["lasso"], [], 100, 1e-4, 1.0, 3.0, 0.5, 1e-7, 500, False, False, np.ones(p),
                     np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), ngpus=G if G > 1 else 0, devices=devices)
    outs = args.outputs(p + 1)
    ns = (C.c_int64 * G)(*[rows_per_device] * G)
    xp = (L._dp * G)(*[api._dptr(xh)] * G)
    yp = (L._dp * G)(*[api._dptr(yh)] * G)
    lib = L.lib()
    ts, st = [], None
    for k in range(calls + 1):
        t0 = time.perf_counter()
        L.check(lib.oemgpu_fit_big(xp, ns, G, p, yp, 1, 1, C.byref(args.c), *outs))
        ts.append(1e3 * (time.perf_counter() - t0))
        st = L.host_stats()
    gb = 8.0 * G * (rows_per_device * p + rows_per_device) / 1e9
    med = float(np.median(ts[1:]))
    lib.oemgpu_release_cache()
    return {"what": "oemgpu_fit_big(host shards): big.oem() lasso, p=256, 100 lambdas, %d rows per device (a sample of config 5's 1.25e7-row share)" % rows_per_device,
            "ngpus": int(st["devices"]), "rows_per_device": rows_per_device, "first_call_ms": ts[0], "median_ms": med, "calls": calls,
            "GBps_at_median_all_devices": gb / (med * 1e-3), "upload_plus_moments_ms": st["upload_moments_ms"], "solve_ms": st["solve_ms"],
            "row_blocks": st["row_blocks"], "handovers_staged_through_host": st["host_staged_handovers"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--p", type=int, default=100)
    ap.add_argument("--workload", choices=["c1", "c5"], default="c1",
                    help="c1: BASELINE.json config 1, n fixed (strong scaling; the headline).  c5: config 5's per-GPU share, "
                         "1.25e7 x 256 rows PER RANK, big.oem semantics (weak scaling)")
    ap.add_argument("--no-c5", action="store_true", help="c1 run: skip the appended c5 weak-scaling measurement")
    ap.add_argument("--c5-rows", type=int, default=12_500_000, help="rows per rank of the c5 workload")
    ap.add_argument("--no-host", action="store_true", help="skip the host-resident (drop-in .Call level) measurements")
    ap.add_argument("--no-two-callers", action="store_true", help="skip the two-concurrent-callers extra (profiling runs: kernels of two "
                                                                   "callers overlap and their durations no longer describe one solve)")
    ap.add_argument("--no-rccl-check", action="store_true", help="N = 1: skip the one-rank RCCL self-check (process group of world size 1)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure the Gram kernel's HBM traffic / MFMA utilisation in child rocprofv3 "
                                                                 "passes (the figures then come from profiles/, flagged as constants)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU baseline sample (0 = the full workload)")
    argv = sys.argv[1:]
    if not argv and "OEM_BENCH_ARGV" in os.environ and "WORLD_SIZE" in os.environ:      # a worker of self_launch()
        argv = json.loads(os.environ["OEM_BENCH_ARGV"])
    a = ap.parse_args(argv)
    if a.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a, argv)                     # never returns; this process has made no GPU call

    # ONE line on stdout, whatever the libraries print: RCCL writes a version banner to the C-level stdout when a communicator is
    # created (seen after the JSON line at process exit).  File descriptor 1 is pointed at stderr for the whole run and the JSON
    # line goes out through the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {a.gpus}, or without torchrun)", file=sys.stderr)
        sys.exit(2)
    # OEM_BENCH_ONE_DEVICE=1: a functional check of the N > 1 code path on a one-GPU box (all ranks on device 0, gloo
    # instead of RCCL, which refuses two ranks on one device); never a measurement
    one_dev = os.environ.get("OEM_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
    if local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} wants GPU {local}, the node shows {torch.cuda.device_count()}", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    in_group = "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ          # under torchrun, N = 1 included
    coll_backend = None
    if in_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev and world > 1:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        coll_backend = dist.get_backend()

    import oem_amd
    from oem_amd import _lib as L
    from oem_amd import api
    from oem_amd.distributed import HipBackend, oem_sharded, row_partition, sharded_buffers, solve_row_shards

    if a.workload == "c5":
        backend = HipBackend(local)
        steps = a.steps if a.steps != 200 else 10
        warm = a.warmup if a.warmup != 10 else 2
        r = c5_weak(torch, dist, world, rank, dev, backend, a.c5_rows, steps, warm)
        if rank == 0:
            emit(({"metric": "full-lambda-path solves/sec (config 5 share: big.oem lasso, 1.25e7 x 256 rows per GPU)",
                              "value": r["value"], "unit": "solves/s", "n_gpus": world, "steps": steps, "warmup": warm,
                              "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "f64", "data": "synthetic", "config": {"workload": r["workload"], "rows_per_gpu": r["rows_per_gpu"]},
                              "roofline": {"bound": "mfma", "kernel": "gram_wd_kernel (v_mfma_f64_16x16x4_f64; one eight-wave workgroup per row chunk, one read of X)", "achieved": r["gram_TFLOPs"],
                                           "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": r["gram_frac_of_fp64_mfma_peak"],
                                           "traffic": None},
                              "rccl_ranks": world if coll_backend == "nccl" else None, "collective_backend": coll_backend,
                              "detail": r}))
        if in_group:
            dist.destroy_process_group()
        return

    n, p, m = a.n, a.p, 25
    lo, hi = row_partition(n, world)[rank]
    n_loc = hi - lo
    # Synthetic data of the README's shape, generated on the device (column-major: a (p, n_loc) row-major tensor)
    x, y = gen_c1(torch, dev, n, p, m, lo, hi, rank, world)

    kw = dict(penalty="elastic.net", alpha=1.0, intercept=True, standardize=False)
    backend = HipBackend(local)
    dd = dist if world > 1 else None

    def solve_py(lam, tol):
        return oem_sharded(x, y, backend=backend, dist=dd, lambda_=lam, tol=tol, **kw)

    # the README first runs a default fit to obtain the lambda sequence, then times tol = 1e-10 fits on it
    lambdas = solve_py((), 1e-7)["lambda"][0]

    # The timed step is the C-ABI call sequence itself (what `.Call("oem_fit_dense")` is to the reference):
    # arguments marshalled once, then per step: moments -> [all-reduce] -> solve (results on host) [-> shifted redo if advised].
    def new_args():
        return api._Args(["elastic.net"], [np.asarray(lambdas)], 100, 1e-4, 1.0, 3.0, 0.5, 1e-10, 500, False, False,
                         np.ones(p), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    args = new_args()
    bufs = sharded_buffers(backend, p)
    outs = args.outputs(p + 1)                  # caller-allocated result buffers, written by every step

    def solve(lam=None, tol=None):              # call inside `with backend.section():`
        # N = 1: oemgpu_fit_dense_dev (the drop-in entry point); N > 1: moments -> all-reduce -> solve
        solve_row_shards(backend, dd, None, x, n_loc, n_loc, p, y, bufs, L.OEMGPU_SEM_DENSE, False, True, args, outs)
        return args
    # kernels and RCCL collectives are stream-ordered on the backend's stream: one section around each loop
    # (a fixed 0.2 s of untimed solves first, whatever W is: the GPU's clocks settle over milliseconds, W = 5 steps are 2.5 ms -- the timed K steps
    #  then measure the steady state the metric means, not the ramp the driver's choice of W happens to leave: 20 steps behind 5 gave 0.538 ms where
    #  200 behind 10 gave 0.517-0.524 on the same box.  The same count on every rank: the solves contain a collective.  Reported as `prewarm_solves`.)
    PREWARM = 0 if one_dev else 400                     # (OEM_BENCH_ONE_DEVICE: a functional check of the N > 1 path on one GPU, never a measurement)
    with backend.section():
        for _ in range(PREWARM):
            solve()
        for _ in range(a.warmup):
            solve()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with backend.section():
        for _ in range(a.steps):
            solve()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    beta_timed = args.beta.copy()
    # ---- the collective on its own: the (p+2)^2 moment buffer, HIP events on the stream the collectives are ordered on
    allreduce_ms = allgather_ms = None
    if world > 1:
        ex = time_exchanges(torch, dist, backend, bufs[1])
        allreduce_ms, allgather_ms = ex["allreduce_ms"], ex["allgather_ms"]
    # Two callers at once (two host threads, two contexts / streams, the same resident X): the path kernel of one solve occupies ONE
    # CU for 0.30 ms, so the moment pass of the other caller's solve runs beside it.  Reported as an extra, never as `value`: a
    # solve is still 0.5 ms long, this is what the chip delivers when the solves are independent.  Each thread runs for a fixed
    # wall time (>= 0.25 s), not for steps / 2 solves: ten solves per thread were noise.
    two_callers = None
    if world == 1 and not a.no_two_callers:
        import threading
        backs = [HipBackend(local), HipBackend(local)]
        sets = []
        for bk in backs:
            ar = new_args()
            sets.append((bk, ar, sharded_buffers(bk, p), ar.outputs(p + 1)))
        counts = [0, 0]
        WALL = 0.25

        def caller(i, bk, ar, bf, ou, wall):
            tend = time.perf_counter() + wall
            with bk.section():
                while True:
                    solve_row_shards(bk, None, None, x, n_loc, n_loc, p, y, bf, L.OEMGPU_SEM_DENSE, False, True, ar, ou)
                    counts[i] += 1
                    if time.perf_counter() >= tend:
                        break
        for i, st_ in enumerate(sets):
            caller(i, *st_, 0.02)                         # warm both contexts
        torch.cuda.synchronize()
        counts = [0, 0]
        th = [threading.Thread(target=caller, args=(i,) + st_ + (WALL,)) for i, st_ in enumerate(sets)]
        t2 = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        same = bool(np.array_equal(sets[0][1].beta, args.beta) and np.array_equal(sets[1][1].beta, args.beta))
        two_callers = {"solves_per_s": sum(counts) / dt2, "callers": 2, "solves": sum(counts), "wall_s": dt2,
                       "same_bits_as_the_single_caller": same,
                       "note": "independent solves from two host threads on two contexts for a fixed wall time; NOT the headline value"}
    # the same through the Python mirror of the R front end (argument checks, result decoration): reported, not `value`
    t1 = time.perf_counter()
    for _ in range(20):
        fit = solve_py(lambdas, 1e-10)
    dt_py = (time.perf_counter() - t1) / 20
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel (the MFMA moment build), HIP events on the kernel's own stream
    lib = L.lib()
    L.check(lib.oemgpu_set_timing(backend.ctx, 1))
    reps = max(10, min(50, a.steps))
    acc = np.zeros(L.NTIMERS)
    ms = (C.c_double * L.NTIMERS)()
    with backend.section():
        for _ in range(reps):
            solve()
            L.check(lib.oemgpu_last_timings(backend.ctx, ms))
            acc += np.array(list(ms))
    L.check(lib.oemgpu_set_timing(backend.ctx, 0))
    acc /= reps
    _st, _cp = C.c_int32(-1), C.c_int32(-1)
    L.lib().oemgpu_last_eigen_info(backend.ctx, C.byref(_st), C.byref(_cp))
    eig_info = {"lanczos_steps": int(_st.value), "step_cap_reached": bool(_cp.value)}
    gram_ms = acc[L.T_GRAMK]
    flops = float(n_loc) * p * (p + 1) + 2.0 * n_loc * p            # lower-triangular syrk + X'y
    bytes_alg = 8.0 * n_loc * p + 8.0 * n_loc
    achieved_tf = flops / (gram_ms * 1e-3) / 1e12 if gram_ms > 0 else 0.0
    # HBM traffic of that kernel: PMC counters cannot be read from inside this process, so the figure is the one the
    # separate rocprofv3 --pmc passes of this same command measured (tools/round_artifacts.sh -> profiles/),
    # FETCH_SIZE x2 + WRITE_SIZE as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950; null if it does not apply.
    traffic, traffic_src = None, None
    pmcs = sorted((ROOT / "profiles").glob("r*_pmc_gram.json"))
    if pmcs and n == 1_000_000 and p == 100 and world == 1:
        traffic = float(json.loads(pmcs[-1].read_text())["hbm_bytes_per_dispatch"])
        traffic_src = ("CONSTANT, not measured in this run: profiles/%s (the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                       "command, tools/round_artifacts.sh); counters cannot be read from inside the process" % pmcs[-1].name)

    # ---- N = 1 without torchrun: RCCL executed once all the same -- a process group of world size 1 and the solve forced through
    # its collective branch (OEM_FORCE_COLLECTIVES: moments -> all-reduce -> solve).  Same bits as the timed solves, or it says so.
    rccl_check = None
    if world == 1 and not a.no_rccl_check and not (in_group and coll_backend != "nccl"):
        try:
            if not in_group:
                os.environ["MASTER_ADDR"] = "127.0.0.1"
                os.environ["MASTER_PORT"] = str(_free_port())
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            os.environ["OEM_FORCE_COLLECTIVES"] = "1"
            ar = new_args()
            bf, ou = sharded_buffers(backend, p), ar.outputs(p + 1)
            with backend.section():
                for _ in range(3):
                    solve_row_shards(backend, dist, None, x, n_loc, n_loc, p, y, bf, L.OEMGPU_SEM_DENSE, False, True, ar, ou)
            torch.cuda.synchronize()
            ex = time_exchanges(torch, dist, backend, bf[1])
            rccl_check = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                          "same_bits_as_the_plain_solve": bool(np.array_equal(ar.beta, beta_timed)),
                          "allreduce_ms": ex["allreduce_ms"], "allgather_ms": ex["allgather_ms"], "allreduce_doubles": int(bf[1].numel()),
                          "exchange": "the solve sums its moment buffers by ONE all-gather + one kernel in rank order (allgather_ms); allreduce_ms is dist.all_reduce of the same buffer"}
            if not in_group:
                dist.destroy_process_group()
        except Exception as e:                   # the headline line must print whatever RCCL does on this box
            rccl_check = {"error": repr(e)}
        finally:
            os.environ.pop("OEM_FORCE_COLLECTIVES", None)

    mfma_util, mfma_src = None, None
    mf = sorted((ROOT / "profiles").glob("r*_pmc_gram_mfma.json"))
    if mf and n == 1_000_000 and p == 100 and world == 1:
        mfma_util = float(json.loads(mf[-1].read_text())["mfma_util"])
        mfma_src = ("CONSTANT, not measured in this run: profiles/%s (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on this command): "
                    "matrix-pipe busy cycles / (GPU cycles x 1024 SIMDs), i.e. at the clock the chip held" % mf[-1].name)
    out = None
    if rank == 0:
        niter_total = int(np.sum(fit["niter"][0]))
        c1_exact = n == 1_000_000 and p == 100
        out = {
            "metric": "full-lambda-path solves/sec (n=1e6 p=100 lasso, 100 lambdas, tol 1e-10)",
            "value": a.steps / dt, "unit": "solves/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "prewarm_solves": PREWARM,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": (a.steps / dt) * README_SECONDS if c1_exact else None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config 1: oem() lasso (penalty='elastic.net', alpha=1), dense Gaussian X~N(0,9), "
                                   "intercept, no standardize, 100-lambda path supplied, tol 1e-10 (README benchmark)",
                       "n": n, "p": p, "nlambda": int(len(lambdas)), "rows_per_gpu": n_loc,
                       "sharding": "rows/N + one all-reduce of the (p+2)^2 moment buffer" if world > 1 else "none",
                       "oem_iterations_per_solve": niter_total},
            "roofline": {"bound": "mfma", "kernel": "gram_ring_kernel<7> (v_mfma_f64_16x16x4_f64; diagonal tiles and the ragged strip as v_mfma_f64_4x4x4_4b_f64 sub-blocks)",
                         "achieved": achieved_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src, "mfma_util_counters": mfma_util, "mfma_util_source": mfma_src,
                         "kernel_ms": gram_ms, "algorithmic_flops": flops, "algorithmic_bytes": bytes_alg,
                         "hbm_GBps_algorithmic": bytes_alg / (gram_ms * 1e-3) / 1e9 if gram_ms > 0 else 0.0},
            "stage_ms": {"shift_sample": acc[L.T_SHIFT], "moments_total": acc[L.T_MOMENTS], "finalize": acc[L.T_FINAL],
                         "eigen_plus_path": acc[L.T_EIGPATH], "gram_kernel": gram_ms,
                         "python_front_end_total": 1e3 * dt_py},
            "path_kernel_clock_GHz": (acc[6] / acc[7] * 0.1) if acc[7] > 0 else None,
            "path_kernel_cycles": acc[6],
            "path_kernel_cycles_per_oem_iteration": acc[6] / niter_total if niter_total > 0 else None,
            "path_kernel_note": "eigenvalue (Lanczos) + 100-lambda path in ONE launch; cycles include the eigen step's fixed cost",
            "eigen_step": eig_info,
            "vs_baseline_note": "reference README: 1.600 s per solve on unstated CPU hardware, x and y in host memory.  `vs_baseline` divides "
                                "that by a DEVICE-RESIDENT step (X already in HBM); the like-for-like drop-in figure is "
                                "`vs_baseline_host_resident` = README seconds / host_resident_ms.c1.median_ms (the same .Call from pageable host memory)",
            "rccl_ranks": world if coll_backend == "nccl" else (rccl_check or {}).get("rccl_ranks"),
            "collective_backend": coll_backend if coll_backend else (rccl_check or {}).get("backend"),
            "allreduce_ms": allreduce_ms if allreduce_ms is not None else (rccl_check or {}).get("allreduce_ms"),
            "allgather_ms": allgather_ms if allgather_ms is not None else (rccl_check or {}).get("allgather_ms"),
            "allreduce_doubles": (p + 2) * (p + 2),
            "rccl_selfcheck": rccl_check,
            "throughput_two_callers": two_callers,
        }
    # every rank takes part in the appended weak-scaling measurement (it has a collective)
    xh_full = yh_full = None
    if rank == 0 and world == 1 and not (a.no_host and a.no_cpu_baseline):
        xh_full = x.cpu().numpy()                  # (n, p) with strides (8, 8n): column-major, as R holds it
        yh_full = y.cpu().numpy()
        assert xh_full.flags.f_contiguous
    if rank == 0 and world == 1 and not a.no_host:
        # what the drop-in .Call delivers: the same solve from pageable host memory (never `value`)
        hr, hargs = host_resident(xh_full, yh_full, lambdas, p)
        hr["max_abs_beta_diff_vs_the_resident_solve"] = float(np.abs(hargs.beta - beta_timed).max())      # (row blocks are summed in another order)
        out["host_resident_ms"] = {"c1": hr}
        if c1_exact:
            out["vs_baseline_host_resident"] = README_SECONDS / (hr["median_ms"] * 1e-3)
    if not a.no_c5 and n == 1_000_000 and p == 100:
        try:
            del x
            torch.cuda.empty_cache()
            r5 = c5_weak(torch, dist, world, rank, dev, backend, a.c5_rows, 10, 2)
            if rank == 0:
                out["c5_weak"] = r5
                if world == 1 and not a.no_cpu_baseline:
                    # CPU figure, EXTRAPOLATED: the oracle's big.oem on 40,000 and 80,000 rows of the same shape, t(n) = a + b n to this share's rows
                    from oracle import oracle as orc
                    orc.lib(True)
                    r5g = np.random.default_rng(55)
                    pts5 = []
                    for nn in (40_000, 80_000):
                        xs = np.asfortranarray(r5g.normal(size=(nn, 256))); bb5 = np.zeros(256); bb5[:25] = r5g.uniform(0, 1, 25)
                        ys = xs @ bb5 + r5g.normal(size=nn)
                        t0 = time.perf_counter(); orc.fit_big(xs, ys, native=True, penalty=["lasso"], nlambda=100, tol=1e-7); pts5.append(time.perf_counter() - t0)
                    full5 = pts5[0] + (pts5[1] - pts5[0]) / 40_000.0 * (a.c5_rows - 40_000)
                    r5["cpu_baseline"] = {"value": 1.0 / full5, "unit": "solves/s", "cores": 1, "kind": "port", "seconds": full5,
                                          "sample": "EXTRAPOLATED: the oracle on 40,000 and 80,000 rows (%.2f s, %.2f s), t(n) = a + b n to the %d rows of one GPU's share" % (pts5[0], pts5[1], a.c5_rows)}
        except Exception as e:          # e.g. not enough free HBM on a shared device: the headline line must still print
            if rank == 0:
                out["c5_weak"] = {"error": repr(e)}
    if rank == 0:
        # ONE object per run that explains this point of the 1 -> N curve for BOTH workloads: per-GPU rows and
        # where a step's time goes -- the moment pass on the local rows, the one all-reduce, the replicated solve
        def _point(value, ms_step, rows, st, ar, ag):
            return {"value_solves_per_s": value, "ms_per_step": ms_step, "rows_per_gpu": rows, "moments_ms": st.get("moments_total"),
                    "allreduce_ms": ar, "allgather_ms": ag, "solve_ms": (st.get("finalize") or 0.0) + (st.get("eigen_plus_path") or 0.0)}
        curve = {"n_gpus": world, "c1_strong": _point(out["value"], out["ms_per_step"], n_loc, out["stage_ms"], out.get("allreduce_ms") if world > 1 else None,
                                                      out.get("allgather_ms") if world > 1 else None)}
        r5o = out.get("c5_weak")
        if isinstance(r5o, dict) and "value" in r5o:
            curve["c5_weak"] = _point(r5o["value"], r5o["ms_per_step"], r5o["rows_per_gpu"], r5o["stage_ms"], r5o.get("allreduce_ms"), r5o.get("allgather_ms"))
        curve["note"] = ("c1: n = 1e6 rows split over the GPUs (strong: the moment pass shrinks, the exchange and the solve do not); c5: 1.25e7 rows on "
                         "EVERY GPU (weak: all three stay).  The timed steps sum the moment buffers by one all-gather + one kernel in rank order "
                         "(allgather_ms; fixed summation order); allreduce_ms is dist.all_reduce of the same buffer, for comparison.  No 1 -> 8 curve has been measured on hardware "
                         "by the builder: these objects from the driver's N = 1, 2, 4, 8 runs are the curve and its explanation")
        out["scaling_point"] = curve
    if rank == 0 and world == 1 and not a.no_host:
        try:
            out["host_resident_ms"]["c5_sample"] = host_resident_c5(torch, dev, 1)
        except Exception as e:
            out["host_resident_ms"]["c5_sample"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_host:
        # SURVEY section 8 row f-3 on the record: oem() with p >= n, X resident -- 500 x 2,000: since round 4 the Gram form of the iteration
        # on the row-split engine (the whole 2000 x 2000 Gram in the accumulator files of 125 CUs, ONE exchange per iteration; round 3:
        # the reference's own two-product form as one persistent launch, two exchanges); 500 x 20,000: that two-product form with the
        # 82 MB of standardised X in the vector AND accumulator registers of 209 CUs (path_wres_kernel, round 4; until then streamed from
        # HBM every iteration at 20 us).  Never `value`.
        try:
            import warnings
            wide = {}
            for (wn, wp, wl) in ((500, 2000, 50), (500, 20000, 20)):
                gw = torch.Generator(device=dev); gw.manual_seed(7)
                xw = torch.randn((wp, wn), generator=gw, device=dev, dtype=torch.float64)
                bw = torch.zeros(wp, dtype=torch.float64, device=dev); bw[:10] = 1.0
                yw = (xw.t() @ bw + torch.randn(wn, generator=gw, device=dev, dtype=torch.float64)).cpu().numpy()
                best, wfit = 1e9, None
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    for _ in range(2):
                        t0 = time.perf_counter(); wfit = oem_amd.oem(xw.t(), yw, penalty="lasso", nlambda=wl, tol=1e-7); torch.cuda.synchronize()
                        best = min(best, time.perf_counter() - t0)
                it = int(np.sum(wfit["niter"][0]))
                wide[f"{wn}x{wp}_lasso_{wl}_lambdas"] = {"ms": 1e3 * best, "iterations": it, "us_per_iteration": 1e6 * best / max(it, 1),
                                                         "engine": oem_amd.last_path_engine()[0]}
                del xw
            out["p_ge_n_ms"] = wide
        except Exception as e:
            out["p_ge_n_ms"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_host:
        # BASELINE config 4 on the record: oem.xtx, p = 4096, 100-lambda lasso, tol 1e-10 -- eigen + path in ONE persistent launch with the
        # lower triangle of XX in the register files of 174 CUs (path_symcoop.hip): nothing is streamed per iteration.  Never `value`.
        try:
            g4 = torch.Generator(device=dev); g4.manual_seed(9)
            p4, n4 = 4096, 65536
            x4 = torch.randn((n4, p4), generator=g4, device=dev, dtype=torch.float64)
            b4 = torch.zeros(p4, dtype=torch.float64, device=dev); b4[:25] = 2.0 * torch.rand(25, generator=g4, device=dev, dtype=torch.float64) - 1.0
            y4 = x4 @ b4 + torch.randn(n4, generator=g4, device=dev, dtype=torch.float64)
            xtx4 = (x4.t() @ x4) / n4
            xty4 = ((x4.t() @ y4) / n4).cpu().numpy()
            del x4
            ctx4 = oem_amd.context()
            L.check(L.lib().oemgpu_set_timing(ctx4, 1))
            best, f4 = 1e9, None
            for _ in range(3):
                f4 = oem_amd.oem_xtx(xtx4, xty4, penalty="lasso", nlambda=100, tol=1e-10); torch.cuda.synchronize()
                ms4 = (C.c_double * L.NTIMERS)(); L.check(L.lib().oemgpu_last_timings(ctx4, ms4))
                best = min(best, ms4[L.T_EIGPATH])
            L.check(L.lib().oemgpu_set_timing(ctx4, 0))
            st4, cp4 = C.c_int32(-1), C.c_int32(-1)
            L.lib().oemgpu_last_eigen_info(ctx4, C.byref(st4), C.byref(cp4))
            it4 = int(np.sum(f4["niter"][0]))
            alg = (8.0 * p4 * p4 + 24.0 * p4) * it4
            out["c4_ms"] = {"workload": "config 4: oem.xtx, p = 4096, 100-lambda lasso, tol 1e-10 (X'X/n of n = 65536 Gaussian rows, 25 non-zeros)",
                            "eigen_plus_path_ms": best, "oem_iterations": it4, "us_per_iteration_incl_lanczos": 1e3 * best / max(it4, 1),
                            "lanczos_steps": int(st4.value), "persistent_kernel_cycles": ms4[6],
                            "algorithmic_GBps_at_8p2_plus_24p_bytes_per_iteration": alg / (best * 1e-3) / 1e9,
                            # the bound that applies: FP64 VALU (2 p^2 flops per product, OEM iterations + Lanczos steps), and the floor of an
                            # iteration as stamped and probed (profiles/r5_c4_*, profiles/r6_symcoop_stamped.txt)
                            "fp64_valu_TFLOPs": 2.0 * p4 * p4 * (it4 + int(st4.value)) / (best * 1e-3) / 1e12,
                            "fp64_valu_frac_of_peak": 2.0 * p4 * p4 * (it4 + int(st4.value)) / (best * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                            "us_per_iteration_floor_of_this_design": C4_FLOOR_US,
                            "frac_of_that_floor": C4_FLOOR_US / (1e3 * best / max(it4 + int(st4.value), 1)),
                            "note": "the matrix is register-resident for the whole call: the 'bandwidth' above is SURVEY 8(d)'s byte count over the "
                                    "measured time (multiples of the 8 TB/s HBM peak because the bytes are never read); the loop is bound by two "
                                    "exchanges per iteration through the memory side and by FP64 VALU, not by HBM"}
            if not a.no_cpu_baseline:
                # CPU figure, EXTRAPOLATED: the oracle on the first four lambdas of the same grid (one 134 MB GEMV + threshold per iteration),
                # its ms per iteration x the full path's iteration count
                from oracle import oracle as orc
                orc.lib(True)
                xtx4h = xtx4.cpu().numpy()
                t0 = time.perf_counter()
                r4 = orc.fit_xtx(xtx4h, xty4, native=True, penalty=["lasso"], lambda_=np.asarray(f4["lambda"][0])[:4], tol=1e-10, d_override=float(f4["d"]))
                t4c = time.perf_counter() - t0
                it4c = int(np.sum(r4["niter"][0]))
                out["c4_ms"]["cpu_baseline"] = {"value": t4c / max(it4c, 1) * it4 * 1e3, "unit": "ms per call", "cores": 1, "kind": "port",
                                                "sample": "EXTRAPOLATED: the oracle on the first 4 of the 100 lambdas (%d iterations in %.2f s, d handed over: no eigen-solve), "
                                                          "ms per iteration x the %d iterations of the full path" % (it4c, t4c, it4)}
                del xtx4h
            del xtx4
        except Exception as e:
            out["c4_ms"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_host:
        # The one GEMV loop that IS HBM-bound (north_star: "OEM GEMV loop at >= 70 % HBM peak, evidenced by rocprof"): oem.xtx beyond the
        # register-resident engines, q = 8,192 -- config 4's recipe at twice the size (X'X/n of n = 16 q Gaussian rows, 25 non-zeros, 100-lambda
        # lasso, tol 1e-10).  The matrix (537 MB) fits neither registers nor the Infinity Cache: every product streams the packed lower
        # triangle once (path_large.hip: sympk_*).  `roofline` prices the product kernel's own time (HIP events on its stream,
        # oemgpu_selftest_sympk_gemv) on the bytes its algorithm moves; `traffic` is measured now by child rocprofv3 passes.  Never `value`.
        try:
            q8, n8 = 8192, 16 * 8192
            g8 = torch.Generator(device=dev); g8.manual_seed(8192)
            xtx8 = torch.zeros((q8, q8), device=dev, dtype=torch.float64)
            b8 = torch.zeros(q8, dtype=torch.float64, device=dev); b8[:25] = 2.0 * torch.rand(25, generator=g8, device=dev, dtype=torch.float64) - 1.0
            xty8 = torch.zeros(q8, device=dev, dtype=torch.float64)
            for _ in range(8):                                   # rows in eight blocks: no 8.6 GB temporary
                xb = torch.randn((n8 // 8, q8), generator=g8, device=dev, dtype=torch.float64)
                yb = xb @ b8 + torch.randn(n8 // 8, generator=g8, device=dev, dtype=torch.float64)
                xtx8 += xb.t() @ xb; xty8 += xb.t() @ yb
                del xb, yb
            xtx8 /= n8; xty8h = (xty8 / n8).cpu().numpy()
            ctx8 = oem_amd.context()
            L.check(L.lib().oemgpu_set_timing(ctx8, 1))
            ts8, f8 = [], None
            for _ in range(3):
                f8 = oem_amd.oem_xtx(xtx8, xty8h, penalty="lasso", nlambda=100, tol=1e-10); torch.cuda.synchronize()
                ms8 = (C.c_double * L.NTIMERS)(); L.check(L.lib().oemgpu_last_timings(ctx8, ms8))
                ts8.append(ms8[L.T_EIGPATH])
            L.check(L.lib().oemgpu_set_timing(ctx8, 0))
            st8, cp8 = C.c_int32(-1), C.c_int32(-1)
            L.lib().oemgpu_last_eigen_info(ctx8, C.byref(st8), C.byref(cp8))
            it8 = int(np.sum(f8["niter"][0])); prod8 = it8 + int(st8.value)
            v8 = torch.randn(q8, generator=g8, device=dev, dtype=torch.float64); o8 = torch.empty_like(v8)
            us8 = C.c_double(0.0)
            L.check(L.lib().oemgpu_selftest_sympk_gemv(ctx8, xtx8.data_ptr(), q8, v8.data_ptr(), o8.data_ptr(), 50, C.byref(us8)))
            gemv_err = float((o8 - xtx8 @ v8).abs().max() / (xtx8 @ v8).abs().max())
            nblk8 = (q8 + 127) // 128
            bytes8 = 4.0 * (128 * nblk8) ** 2 + 512.0 * 128 * nblk8 + 8.0 * 128 * nblk8 * nblk8      # blocks of the lower triangle (diagonal ones whole) + the partial vectors written
            med8 = float(np.median(ts8))
            rf8 = {"bound": "hbm", "kernel": "sympk_gemv_kernel (one 128 x 128 block of the packed lower triangle per workgroup, both products from one read)",
                   "achieved": bytes8 / (us8.value * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": bytes8 / (us8.value * 1e-6) / 1e9 / 8000.0,
                   "traffic": None, "traffic_unit": "bytes per launch", "kernel_us": us8.value, "algorithmic_bytes": bytes8,
                   "note": "algorithmic bytes = 4 q^2 + 512 q (blocks) + 8 q ceil(q / 128) (partial vectors written); SURVEY 8(d)'s 8 q^2 + 24 q "
                           "is what a row-streaming product reads -- half of it is never read here"}
            if not a.no_live_pmc:
                try:
                    t0 = time.perf_counter()
                    c8 = live_pmc("sympk_gemv_kernel", [["FETCH_SIZE"], ["WRITE_SIZE"]], program=[str(ROOT / "tools" / "run_q8192.py"), "8192", "full"])
                    rf8["traffic"] = (2.0 * c8["FETCH_SIZE"] + c8["WRITE_SIZE"]) * 1024.0
                    rf8["traffic_source"] = ("measured in this run: child passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE -- python3 tools/run_q8192.py 8192 full` (dense-vector "
                                             "products: no block skipped), mean per launch of sympk_gemv_kernel; FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes")
                    rf8["achieved_on_traffic_GBps"] = rf8["traffic"] / (us8.value * 1e-6) / 1e9
                    rf8["live_pmc_seconds"] = time.perf_counter() - t0
                except Exception as e:
                    rf8["live_pmc_error"] = repr(e)
            out["q8192_ms"] = {"workload": "oem.xtx, q = 8192 (X'X/n of n = 131072 Gaussian rows, 25 non-zeros), 100-lambda lasso, tol 1e-10: the Gram form beyond the register-resident engines",
                               "eigen_plus_path_ms": med8, "eigen_plus_path_ms_runs": ts8, "oem_iterations": it8, "lanczos_steps": int(st8.value),
                               "us_per_product_all_in": 1e3 * med8 / max(prod8, 1),
                               "us_per_product_note": "over the whole path; a product skips the 128 x 128 blocks whose row AND column block of the iterate are zero "
                                                      "(the first half of this path has <= 33 non-zeros) -- the roofline below is the FULL product (a dense vector, no block skipped)",
                               "GBps_at_8q2_plus_24q_bytes_per_product": (8.0 * q8 * q8 + 24.0 * q8) * prod8 / (med8 * 1e-3) / 1e9,
                               "engine": oem_amd.last_path_engine()[0], "product_rel_err_vs_torch": gemv_err, "roofline": rf8}
            del xtx8, v8, o8
            torch.cuda.empty_cache()
        except Exception as e:
            out["q8192_ms"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_host:
        # BASELINE configs 2 and 3 on the record, as SURVEY section 8(d) and the reference's README state them (device-resident X, whole oem()
        # calls, median of five, tol 1e-10): c2 = MCP (gamma 2) and SCAD (gamma 4) at n = 5,000, p = 200, b ~ U(-0.5, 0.5) x 25, X ~ N(0, 9), 200
        # lambdas, intercept + standardize (README.md:100-141; one CU: the serial chain of the row-split kernel); c3 = grp.lasso at n = 1e6,
        # p = 512, 64 groups of 8, 100 lambdas with intercept = FALSE, standardize = FALSE as README.md:199-213 -- and the defaults beside it.
        # Each with its CPU figure: the oracle (C port, one thread, -O3 -march=native) on the same data for c2, for c3 extrapolated from
        # two sub-sampled row counts (the moment pass is linear in n, the path does not depend on n).  Parity of both at full size:
        # tests/test_gpu_configs.py.  Never `value`.
        cpu_ok = not a.no_cpu_baseline
        if cpu_ok:
            from oracle import oracle as orc
            orc.lib(True)
        ctx3 = oem_amd.context()
        try:
            rec = {}
            g2 = torch.Generator(device=dev); g2.manual_seed(21)
            x2 = torch.randn((200, 5000), generator=g2, device=dev, dtype=torch.float64) * 3.0
            b2 = torch.zeros(200, dtype=torch.float64, device=dev); b2[:25] = torch.rand(25, generator=g2, device=dev, dtype=torch.float64) - 0.5
            y2 = (x2.t() @ b2 + torch.randn(5000, generator=g2, device=dev, dtype=torch.float64)).contiguous()
            x2h = np.asfortranarray(x2.t().cpu().numpy()); y2h = y2.cpu().numpy()
            for pen, gam in (("mcp", 2.0), ("scad", 4.0)):
                ts2, f2 = [], None
                for k2 in range(6):
                    t0 = time.perf_counter(); f2 = oem_amd.oem(x2.t(), y2, penalty=pen, gamma=gam, nlambda=200, tol=1e-10); torch.cuda.synchronize()
                    if k2:
                        ts2.append(1e3 * (time.perf_counter() - t0))
                rec[pen] = {"ms": float(np.median(ts2)), "ms_runs": ts2, "iterations": int(np.sum(f2["niter"][0])), "engine": oem_amd.last_path_engine()[0],
                            "reference_readme_ms": 105.9 if pen == "mcp" else 80.2}
                if cpu_ok:
                    t0 = time.perf_counter(); r2 = orc.fit_dense(x2h, y2h, native=True, penalty=pen, gamma=gam, nlambda=200, tol=1e-10); tc2_ = time.perf_counter() - t0
                    rec[pen]["cpu_baseline"] = {"value": 1e3 * tc2_, "unit": "ms per call", "cores": 1, "kind": "port", "sample": "the full workload, 1 call"}
                    rec[pen]["max_abs_beta_err_vs_cpu"] = float(np.abs(np.asarray(f2["beta"][0]) - np.asarray(r2["beta"][0])).max())
            out["c2_ms"] = {"workload": "config 2: oem() MCP (gamma 2) / SCAD (gamma 4), n = 5000, p = 200, 200 lambdas, tol 1e-10, intercept + standardize (README.md:100-141)", **rec}
            del x2, y2
        except Exception as e:
            out["c2_ms"] = {"error": repr(e)}
        try:
            g3 = torch.Generator(device=dev); g3.manual_seed(22)
            n3, p3 = 1_000_000, 512
            x3 = torch.empty((p3, n3), device=dev, dtype=torch.float64)
            for j0 in range(0, p3, 64):
                x3[j0:j0 + 64].normal_(generator=g3)
            b3 = torch.zeros(p3, dtype=torch.float64, device=dev); b3[:24] = torch.rand(24, generator=g3, device=dev, dtype=torch.float64) - 0.5
            y3 = (x3.t() @ b3 + torch.randn(n3, generator=g3, device=dev, dtype=torch.float64)).contiguous()
            grp3 = np.repeat(np.arange(1, 65), 8)
            fl3 = float(n3) * p3 * (p3 + 1.0) + 2.0 * n3 * p3
            L.check(L.lib().oemgpu_set_timing(ctx3, 1))
            c3 = {}
            for label, kw3 in (("readme", dict(intercept=False, standardize=False)), ("defaults", dict())):
                ts3, tm3s, f3 = [], [], None
                for k3 in range(6):
                    t0 = time.perf_counter()
                    f3 = oem_amd.oem(x3.t(), y3, penalty="grp.lasso", groups=grp3, nlambda=100, tol=1e-10, **kw3)
                    torch.cuda.synchronize()
                    if k3:
                        ts3.append(1e3 * (time.perf_counter() - t0))
                        tm3 = (C.c_double * L.NTIMERS)(); L.check(L.lib().oemgpu_last_timings(ctx3, tm3))
                        tm3s.append((tm3[L.T_GRAMK], tm3[L.T_EIGPATH]))
                gk = float(np.median([t[0] for t in tm3s]))
                c3[label] = {"ms": float(np.median(ts3)), "ms_runs": ts3, "gram_kernel_ms": gk, "gram_TFLOPs": fl3 / (gk * 1e-3) / 1e12,
                             "gram_frac_of_fp64_mfma_peak": fl3 / (gk * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             "eigen_plus_path_ms": float(np.median([t[1] for t in tm3s])), "iterations": int(np.sum(f3["niter"][0])),
                             "engine": oem_amd.last_path_engine()[0], "placement": oem_amd.api.last_placement()}
            L.check(L.lib().oemgpu_set_timing(ctx3, 0))
            out["c3_ms"] = {"workload": "config 3: oem() grp.lasso, n = 1e6, p = 512, 64 groups of 8, 100 lambdas, tol 1e-10; `readme`: intercept = FALSE, "
                                        "standardize = FALSE (README.md:199-213), `defaults`: both TRUE", **c3["readme"], "defaults": c3["defaults"]}
            if cpu_ok:
                pts = []
                for nn in (20_000, 40_000):
                    xs = np.asfortranarray(x3[:, :nn].t().cpu().numpy()); ys = y3[:nn].cpu().numpy()
                    t0 = time.perf_counter()
                    orc.fit_dense(xs, ys, native=True, penalty=["grp.lasso"], groups=grp3, unique_groups=np.unique(grp3), nlambda=100, tol=1e-10,
                                  standardize=False, intercept=False)
                    pts.append(time.perf_counter() - t0)
                slope = (pts[1] - pts[0]) / 20_000.0
                full = pts[0] + slope * (n3 - 20_000)
                out["c3_ms"]["cpu_baseline"] = {"value": 1e3 * full, "unit": "ms per call", "cores": 1, "kind": "port",
                                                "sample": "EXTRAPOLATED: the oracle on the first 20,000 and 40,000 of the 1e6 rows (%.2f s, %.2f s), t(n) = a + b n "
                                                          "through the two points (the moment pass is linear in n, the path does not depend on n)" % (pts[0], pts[1])}
            del x3, y3
        except Exception as e:
            out["c3_ms"] = {"error": repr(e)}
    if in_group:
        dist.destroy_process_group()
    # ---- N > 1: the in-library multi-GPU path (opts.ngpus = N: what an R caller gets), on rank 0 once the other ranks are gone
    if rank == 0 and world > 1 and not a.no_host:
        # (OEM_BENCH_ONE_DEVICE: the functional check on a one-GPU box runs the leg too -- N contexts of device 0, a small c5 sample)
        devs = [0] * world if one_dev else None
        try:
            del backend, bufs
            torch.cuda.empty_cache()
            time.sleep(3.0)                        # the other ranks leave their devices
            if n % 8 == 0 and 8 % world == 0:
                xf, yf = gen_c1(torch, dev, n, p, m, 0, n, 0, 1)
                xh = xf.cpu().numpy(); yh = yf.cpu().numpy()
                del xf, yf
            else:
                # (n not a multiple of 8: every rank drew its rows from its own seed -- the same rows again, rank by rank)
                xh = np.empty((n, p), order="F"); yh = np.empty(n)
                for r_, (lo_, hi_) in enumerate(row_partition(n, world)):
                    xr, yr = gen_c1(torch, dev, n, p, m, lo_, hi_, r_, world)
                    xh[lo_:hi_] = xr.cpu().numpy(); yh[lo_:hi_] = yr.cpu().numpy()
                    del xr, yr
            torch.cuda.empty_cache()
            hr, hargs = host_resident(xh, yh, lambdas, p, ngpus=world, devices=devs)
            hr["max_abs_beta_diff_vs_the_rank_sharded_solve"] = float(np.abs(hargs.beta - beta_timed).max())
            out["host_resident_ms"] = {"c1": hr}
            if n == 1_000_000 and p == 100:
                out["vs_baseline_host_resident"] = README_SECONDS / (hr["median_ms"] * 1e-3)
            del xh, yh
            out["host_resident_ms"]["c5_sample"] = host_resident_c5(torch, dev, world, rows_per_device=100_000 if one_dev else 2_000_000, devices=devs)
        except Exception as e:
            out.setdefault("host_resident_ms", {})["error"] = repr(e)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # CPU baseline: the C restatement of the reference path (oracle/, -O3 -march=native build), 1 thread = the
        # reference's effective default (R/oem.R:270-273), on a bounded sample of the same workload.
        from oracle import oracle as orc
        rows = n if a.cpu_rows <= 0 else min(n, a.cpu_rows)
        xh = np.asfortranarray(xh_full[:rows])
        yh = yh_full[:rows]
        orc.lib(True)                              # builds the -march=native oracle ON THIS HOST if it was compiled elsewhere
        t0 = time.perf_counter()
        ref = orc.fit_dense(xh, yh, native=True, lambda_=lambdas, tol=1e-10, **kw)
        tc = time.perf_counter() - t0
        sample = f"{rows} of {n} rows, 1 solve (value scaled by rows/n)" if rows != n else "the full workload, 1 solve"
        out["cpu_baseline"] = {"value": 1.0 / tc * (rows / n), "unit": "solves/s", "cores": 1, "kind": "port",
                               "seconds": tc, "sample": sample, "flags": "-O3 -march=native (compiled on this host)",
                               "host_cpus": os.cpu_count()}
        # R's default flags (-O2, no -march; ref src/Makevars:9-12): what an installed reference package is built with
        t0 = time.perf_counter()
        orc.fit_dense(xh, yh, native=False, lambda_=lambdas, tol=1e-10, **kw)
        tc2 = time.perf_counter() - t0
        out["cpu_baseline_O2"] = {"value": 1.0 / tc2 * (rows / n), "unit": "solves/s", "cores": 1, "kind": "port", "seconds": tc2,
                                  "sample": sample, "flags": "-O2 (R's default build flags)"}
        if rows == n:
            out["max_abs_beta_err_vs_cpu"] = float(np.abs(fit["beta"][0] - ref["beta"][0]).max())
            out["niter_equal_cpu"] = bool(np.array_equal(fit["niter"][0], ref["niter"][0]))
        # the same port with the reference's row-block OpenMP Gram (ref src/oem_dense.h:328-358) on every host core;
        # the standardisation passes and the path stay single-threaded, as in the reference
        nc = os.cpu_count() or 1
        t0 = time.perf_counter()
        orc.fit_dense(xh, yh, native=True, lambda_=lambdas, tol=1e-10, ncores=nc, **kw)
        tca = time.perf_counter() - t0
        out["cpu_baseline_all_cores"] = {"value": 1.0 / tca * (rows / n), "unit": "solves/s", "cores": nc, "kind": "port",
                                         "seconds": tca, "sample": out["cpu_baseline"]["sample"]}
    if rank == 0 and world == 1 and not a.no_live_pmc and n == 1_000_000 and p == 100:
        # HBM traffic and matrix-pipe utilisation of the dominant kernel, measured NOW (three short child passes under rocprofv3) instead
        # of quoted from profiles/: FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes; the guide's gfx950 correction for 16-byte-per-lane
        # streaming reads), and rocprofv3's own MfmaUtil formula (GRBM_GUI_ACTIVE arrives summed over the 8 XCDs)
        try:
            t0 = time.perf_counter()
            c = live_pmc("gram_ring_kernel", [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]])
            rf = out["roofline"]
            rf["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            rf["traffic_source"] = ("measured in this run: child passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of this command "
                                    "(--steps 3), mean per launch of gram_ring_kernel; FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes")
            rf["mfma_util_counters"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            rf["mfma_util_source"] = ("measured in this run: child pass `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`: busy cycles / "
                                      "(GPU cycles x 1024 SIMDs), i.e. at the clock the chip held")
            rf["live_pmc_seconds"] = time.perf_counter() - t0
        except Exception as e:
            out["roofline"]["live_pmc_error"] = repr(e)
    if rank == 0:
        if "vs_baseline_host_resident" in out:            # the like-for-like drop-in figure leads, the device-resident one follows
            ordered = {}
            for k, v in out.items():
                if k == "vs_baseline":
                    ordered["vs_baseline_host_resident"] = out["vs_baseline_host_resident"]
                if k != "vs_baseline_host_resident":
                    ordered[k] = v
            out = ordered
        emit(out)


if __name__ == "__main__":
    main()
