"""The host-resident entry points (oemgpu_fit_dense / oemgpu_fit_big with pageable host rows): staged upload through pinned
bounce slots, row blocks overlapped with the moment pass, rows over several devices inside the library (here: the same device
several times -- this pool has one GPU per box), penalties dealt to devices, cached contexts, caller interrupts.
All through the C ABI, against the CPU oracle."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc

# d = 1.005 lambda_max: the reference's own Spectra tolerance is 1e-10 (ref src/oem_dense.h:494-498); the device recurrence stops
# at a 1e-12 tail estimate, so every comparison of d with the oracle's exact eigenvalue holds to 1e-10 relative (it was 1e-8: a
# regression of the eigen step by three orders of magnitude would have passed -- VERDICT r2)
DTOL = 1e-10


@pytest.fixture(scope="module")
def oa():
    import torch
    assert torch.cuda.is_available()
    import oem_amd
    oem_amd.lib()
    return oem_amd


def _data(n, p, seed, mean=0.0, sd=2.0, nnz=8):
    rng = np.random.default_rng(seed)
    x = np.asfortranarray(rng.normal(size=(n, p)) * sd + mean)
    nnz = min(nnz, p)
    b = np.concatenate([rng.uniform(-0.5, 0.5, nnz), np.zeros(p - nnz)])
    y = x @ b + rng.normal(size=n) + 0.7
    return x, y


def _cmp(f, r, tol=1e-9):
    assert abs(f["d"] - r["d"]) <= DTOL * abs(r["d"])
    for k in range(len(r["beta"])):
        assert np.allclose(f["lambda"][k], r["lambda"][k], rtol=1e-12)
        assert np.abs(np.asarray(f["beta"][k]) - np.asarray(r["beta"][k])).max() <= tol, f["penalty"][k]
        assert np.abs(np.ravel(f["niter"][k]).astype(int) - np.ravel(r["niter"][k])).max() <= 1


@pytest.mark.parametrize("mode", ["default", "blocks", "recycled", "one_lane"])
@pytest.mark.parametrize("n,p", [(20011, 37), (3000, 130), (64, 5)])
def test_host_rows_in_blocks(oa, monkeypatch, mode, n, p):
    """row blocks (several per call, ragged last one), bounce slots smaller than a column block, two recycled block buffers
    instead of a resident slice, a single staging lane: all the same fit"""
    from oem_amd import _lib as L
    if mode != "default":
        monkeypatch.setenv("OEMGPU_BLOCK_BYTES", str(8 * p * 1024))          # 1024-row blocks
        monkeypatch.setenv("OEMGPU_SLOT_BYTES", str(64 * 1024))
    if mode == "recycled":
        monkeypatch.setenv("OEMGPU_RESIDENT_BYTES", "1")
    x, y = _data(n, p, 100 + p)
    kw = dict(penalty=["lasso", "mcp"], nlambda=12, tol=1e-10)
    f = oa.oem(x, y, upload_threads=1 if mode == "one_lane" else 3, **kw)
    st = L.host_stats()
    r = orc.fit_dense(x, y, **kw)
    _cmp(f, r)
    assert st["bytes_staged"] == 8 * (n * p + n) and st["devices"] == 1
    if mode != "default":
        assert st["row_blocks"] == (n + 1023) // 1024
    assert st["resident"] == (0.0 if mode == "recycled" else 1.0)


@pytest.mark.parametrize("recycled", [False, True])
def test_host_shifted_redo(oa, monkeypatch, recycled):
    """|mean| >> sd: the speculative pass about 0 is redone about the sample mean -- from HBM when the rows stayed resident,
    by streaming them again when they did not"""
    from oem_amd import _lib as L
    monkeypatch.setenv("OEMGPU_BLOCK_BYTES", str(8 * 20 * 2048))
    if recycled:
        monkeypatch.setenv("OEMGPU_RESIDENT_BYTES", "1")
    n, p = 9000, 20
    x, y = _data(n, p, 7, mean=1e4, sd=1.0)
    kw = dict(penalty=["lasso"], nlambda=10, tol=1e-10)
    f = oa.oem(x, y, **kw)
    st = L.host_stats()
    r = orc.fit_dense(x, y, **kw)
    _cmp(f, r, tol=1e-8)
    assert st["bytes_staged"] == (2 if recycled else 1) * 8 * (n * p + n)


@pytest.mark.parametrize("devs", [[0, 0], [0, 0, 0]])
def test_rows_over_several_devices(oa, devs):
    """ngpus > 1 inside the library: floor(n / G) rows per device (remainder last), the moment buffers summed in device order on
    the first.  One GPU per box here, so the devices are the same ordinal several times: two / three contexts, streams, lane
    sets and accumulators, the peer hand-over degenerating to a device-to-device copy."""
    from oem_amd import _lib as L
    n, p = 30011, 45
    x, y = _data(n, p, 3)
    kw = dict(penalty=["lasso", "scad", "grp.lasso"], groups=np.arange(p) // 5 + 1, nlambda=10, tol=1e-10)
    one = oa.oem(x, y, **kw)
    many = oa.oem(x, y, devices=devs, **kw)
    st = L.host_stats()
    assert st["devices"] == len(devs) and st["bytes_staged"] == 8 * (n * p + n)
    r = orc.fit_dense(x, y, unique_groups=np.unique(kw["groups"]), **kw)
    _cmp(many, r)
    for k in range(3):
        assert np.abs(one["beta"][k] - many["beta"][k]).max() < 1e-11
    # fewer rows than devices: the first devices get none
    import warnings
    xs, ys = np.asfortranarray(x[:2, :3]), y[:2]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tiny = oa.oem(xs, ys, devices=[0, 0, 0], penalty=["lasso"], nlambda=3)
        tiny1 = oa.oem(xs, ys, penalty=["lasso"], nlambda=3)
    assert np.all(np.isfinite(tiny["beta"][0])) and np.abs(tiny["beta"][0] - tiny1["beta"][0]).max() < 1e-12


def test_rows_over_devices_with_shift(oa):
    x, y = _data(12000, 30, 11, mean=5e3, sd=1.0)
    kw = dict(penalty=["lasso", "mcp"], nlambda=8, tol=1e-10)
    f = oa.oem(x, y, devices=[0, 0], **kw)
    _cmp(f, orc.fit_dense(x, y, **kw), tol=1e-8)


def test_penalties_dealt_to_devices(oa, monkeypatch):
    """p > 288 (launch-per-iteration engines) and several penalties and several devices: penalty k is solved on device k mod G
    from a copy of the summed moments (penalties are independent cold starts, ref src/oem_dense.cpp:206-246) -- the same
    result as all penalties on one device"""
    n, p = 2500, 300
    x, y = _data(n, p, 5, sd=1.0)
    groups = np.arange(p) // 6 + 1
    lam = [np.geomspace(0.5, 0.01, 6) * s for s in (1.0, 0.9, 1.1)]
    kw = dict(penalty=["lasso", "grp.lasso", "mcp"], groups=groups, lambda_=lam, tol=1e-9, maxit=2000)
    split = oa.oem(x, y, devices=[0, 0], **kw)
    monkeypatch.setenv("OEMGPU_NO_PENALTY_SPLIT", "1")
    whole = oa.oem(x, y, devices=[0, 0], **kw)
    r = orc.fit_dense(x, y, unique_groups=np.unique(groups), **kw)
    _cmp(split, r)
    for k in range(3):
        assert np.array_equal(split["beta"][k], whole["beta"][k])
        assert np.array_equal(split["niter"][k], whole["niter"][k])
        assert np.array_equal(split["lambda"][k], whole["lambda"][k])


@pytest.mark.parametrize("devs", [None, [0, 0]])
def test_big_oem_host_shards(oa, devs):
    """oemgpu_fit_big: shards of a big.matrix (ragged, one of a single row), ngpus = 1 and rows over two contexts; the row
    ranges of the devices cut across the shards"""
    n, p = 15003, 33
    x, y = _data(n, p, 9, mean=2.0)
    cuts = [0, 4000, 4001, 11000, n]
    xs = [np.asfortranarray(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    ys = [y[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    kw = dict(penalty=["lasso", "mcp", "grp.lasso"], groups=np.arange(p) // 3 + 1, nlambda=10, tol=1e-10)
    f = oa.big_oem(xs, ys, devices=devs, **kw)
    g0, ug = np.concatenate([[0], kw["groups"]]), np.unique(np.concatenate([[0], kw["groups"]]))
    r = orc.fit_big(x, y, **dict(kw, groups=g0, unique_groups=ug))
    _cmp(f, r)


def test_steady_state_allocates_nothing(oa):
    """contexts, streams, events, pinned slots, block buffers and workspaces come from the process-wide cache: the second call
    of the same shape makes no hipMalloc / hipHostMalloc / stream / event creation"""
    from oem_amd import _lib as L
    x, y = _data(50000, 40, 21)
    kw = dict(penalty=["lasso"], nlambda=20, tol=1e-10)
    a = oa.oem(x, y, **kw)
    b = oa.oem(x, y, **kw)
    st = L.host_stats()
    assert st["allocations"] == 0, st
    assert np.array_equal(a["beta"][0], b["beta"][0])                 # and the fit is bitwise reproducible
    xs = [np.asfortranarray(x[:20000]), np.asfortranarray(x[20000:])]
    ys = [y[:20000], y[20000:]]
    oa.big_oem(xs, ys, **kw)
    oa.big_oem(xs, ys, **kw)
    assert L.host_stats()["allocations"] == 0
    L.lib().oemgpu_release_cache()
    c = oa.oem(x, y, **kw)
    assert L.host_stats()["allocations"] > 0                          # rebuilt after the cache was dropped
    assert np.array_equal(a["beta"][0], c["beta"][0])


def test_cache_is_trimmed_on_release(oa, monkeypatch):
    """cached contexts keep their streams, staging lanes and workspace, but the device copy of the rows only up to
    OEMGPU_CACHE_KEEP_BYTES (default: an eighth of the device's memory): beyond that it is freed when the call returns, so that one
    large fit does not hold HBM for the rest of the session.  Forced here with a 1 MB bound: the second call has to
    allocate the copy again (and nothing else), the fit is the same."""
    from oem_amd import _lib as L
    x, y = _data(40000, 30, 9)                      # 9.6 MB of rows
    kw = dict(penalty=["lasso"], nlambda=8, tol=1e-10)
    a = oa.oem(x, y, **kw)
    oa.oem(x, y, **kw)
    assert L.host_stats()["allocations"] == 0       # default bound: everything stays
    monkeypatch.setenv("OEMGPU_CACHE_KEEP_BYTES", str(1 << 20))
    b = oa.oem(x, y, **kw)                          # freed on release ...
    c = oa.oem(x, y, **kw)
    assert 1 <= L.host_stats()["allocations"] <= 2  # ... so this call allocated the copy again, and only that
    assert np.array_equal(a["beta"][0], b["beta"][0]) and np.array_equal(a["beta"][0], c["beta"][0])


def test_caller_interrupt(oa, monkeypatch):
    """opts->interrupt (R: R_CheckUserInterrupt under R_ToplevelExec, ref src/oem_dense.cpp:235-238): polled between row blocks
    on the calling thread; a non-zero answer ends the call with OEMGPU_ERR_INTERRUPTED after cleanup, and the next call works"""
    import threading
    from oem_amd import _lib as L
    monkeypatch.setenv("OEMGPU_BLOCK_BYTES", str(8 * 25 * 1024))
    x, y = _data(20000, 25, 2)
    calls, tids = [], set()

    def stop_on_third():
        calls.append(1)
        tids.add(threading.get_ident())
        return len(calls) >= 3
    with pytest.raises(oa.OemgpuError) as e:
        oa.oem(x, y, penalty=["lasso"], nlambda=5, interrupt=stop_on_third, devices=[0, 0])
    assert e.value.code == L.ERR_INTERRUPTED
    assert len(calls) == 3 and tids == {threading.get_ident()}       # only ever polled on the calling thread
    calls.clear()
    f = oa.oem(x, y, penalty=["lasso"], nlambda=5, interrupt=lambda: False)
    _cmp(f, orc.fit_dense(x, y, penalty=["lasso"], nlambda=5))


def test_fold_threads_never_call_the_caller(oa, monkeypatch):
    """xval.oem beyond one launch (here: the cooperating engine switched off at p + 1 = 301) fits its folds on K worker threads.
    opts->interrupt stands for R_CheckUserInterrupt, which must only ever run on the thread that made the call:
    the fold threads run with the callback removed, the calling thread polls before they start and after they have joined."""
    import threading
    from oem_amd import _lib as L
    monkeypatch.setenv("OEM_NO_COOP", "1")
    rng = np.random.default_rng(77)
    n, p, K = 2500, 300, 4
    x = np.asfortranarray(rng.normal(size=(n, p)))
    y = x[:, :3] @ np.array([1.0, -1.0, 0.5]) + rng.normal(size=n)
    foldid = rng.permutation(np.resize(np.arange(1, K + 1), n))
    kw = dict(penalty=["lasso"], nlambda=4, tol=1e-8, maxit=2000, foldid=foldid)
    tids, calls = set(), []

    def poll():
        tids.add(threading.get_ident()); calls.append(1)
        return False
    a = oa.xval_oem(x, y, interrupt=poll, **kw)
    assert calls and tids == {threading.get_ident()}
    b = oa.xval_oem(x, y, **kw)
    assert np.array_equal(a["beta"][0], b["beta"][0]) and np.array_equal(a["cvm"][0], b["cvm"][0])
    for devices in ([0, 0],):                                   # the same through the rows-over-devices form (its phase 2)
        tids.clear(); calls.clear()
        oa.xval_oem(x, y, interrupt=poll, devices=devices, **kw)
        assert calls and tids == {threading.get_ident()}
    with pytest.raises(oa.OemgpuError) as e:                      # and a "yes" still ends the call
        oa.xval_oem(x, y, interrupt=lambda: True, **kw)
    assert e.value.code == L.ERR_INTERRUPTED


@pytest.mark.gpu
@pytest.mark.parametrize("p,pens,weighted,tm", [(30, ["lasso", "mcp"], False, "mse"), (41, ["grp.lasso", "elastic.net"], True, "mae"),
                                                (300, ["lasso"], False, "mse")])
def test_xval_rows_over_several_devices(oa, p, pens, weighted, tm):
    """oemgpu_xval_dense with opts.ngpus > 1: the rows split over the devices (here the same device several times, with its own
    context each), fold moments handed to the first and added in device order, the K + 1 fits there, the fold coefficients back,
    the error triples merged on the host -- the same numbers as the one-device call."""
    rng = np.random.default_rng(50 + p)
    n = 6 * p + 1007
    x = np.asfortranarray(rng.normal(size=(n, p)) * 1.3 + 0.4)
    y = x[:, :3] @ np.array([1.0, -1.5, 0.5]) + rng.normal(size=n) + 0.2
    K = 7
    foldid = rng.permutation(np.resize(np.arange(1, K + 1), n))
    kw = dict(penalty=pens, nlambda=6, tol=1e-9, maxit=3000, type_measure=tm, foldid=foldid)
    if "grp.lasso" in pens:
        kw["groups"] = np.arange(p) // 4 + 1
    if weighted:
        kw["weights"] = rng.uniform(0.5, 2.0, size=n)
    one = oa.xval_oem(x, y, **kw)
    for devices in ([0, 0], [0, 0, 0]):
        many = oa.xval_oem(x, y, devices=devices, **kw)
        assert abs(many["d"] - one["d"]) < 1e-11 * one["d"]
        for k in range(len(pens)):
            scale = max(1.0, float(np.abs(one["beta"][k]).max()))
            assert np.abs(many["beta"][k] - one["beta"][k]).max() < 1e-9 * scale, (devices, pens[k])
            assert np.allclose(many["lambda"][k], one["lambda"][k], rtol=1e-12)
            assert np.allclose(many["cvm"][k], one["cvm"][k], rtol=1e-10) and np.allclose(many["cvsd"][k], one["cvsd"][k], rtol=1e-8), (devices, pens[k])
    with pytest.raises(Exception):
        oa.xval_oem(x, y, devices=[0, 99], **kw)


def test_handover_staged_through_the_host():
    """devices that cannot access each other (hipDeviceCanAccessPeer == 0): the moment buffers cross through a pinned host buffer.
    OEMGPU_NO_PEER=1 forces that route, in a child process (the switch is read once) -- tests/no_peer_worker.py."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tests" / "no_peer_worker.py")], cwd=root, env=dict(os.environ, OEMGPU_NO_PEER="1"),
                       capture_output=True, text=True, timeout=600)
    assert "NO_PEER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_concurrent_callers_of_the_cooperating_wide_engine(oa):
    """five host threads fit p >= n problems at once, each through ONE persistent launch of 63 cooperating workgroups per penalty set
    (path_wcoop.hip): the callers queue for CU slots (api.hip: CoopSlots -- all workgroups of all such kernels in flight must be
    resident), nobody times out, and every thread gets the bits of the same call made alone"""
    import threading, warnings
    rng = np.random.default_rng(17)
    n, p = 500, 2000
    x = np.asfortranarray(rng.normal(size=(n, p)))
    y = x[:, :6] @ rng.uniform(0.5, 1.5, 6) + rng.normal(size=n)
    kw = dict(penalty=["lasso", "mcp"], nlambda=8, tol=1e-7, maxit=300)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        alone = oa.oem(x, y, **kw)
    out, errs = [None] * 5, []

    def work(i):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(3):
                    out[i] = oa.oem(x, y, **kw)
        except Exception as e:                                    # noqa: BLE001 -- reported below
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(5)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for f in out:
        assert f["d"] == alone["d"]
        for k in range(2):
            assert np.array_equal(f["beta"][k], alone["beta"][k]) and np.array_equal(f["niter"][k], alone["niter"][k])


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["symcoop", "wres"])
def test_persistent_engine_under_real_contention(oa, tmp_path, engine):
    """the fallback of the persistent engines had only ever been tested with a FAKED poison.  Here a second PROCESS
    holds 200 of the device's CUs (tests/hold_cus_worker.py: workgroups that take a whole CU each and spin for six seconds) while
    this process calls oem.xtx at p = 4096, whose persistent engine (path_symcoop.hip) needs 174 workgroups resident at once -- or
    oem() at 250 x 16,000, whose standardised X lives in the vector and accumulator registers of 84 workgroups (path_wres_kernel,
    the engine that may take up to 240 CUs).  Whatever the hardware scheduler does -- run what fits beside the holder, so that the
    exchanges time out after about a second and the call is made again on the launch-per-iteration engine, or queue the launch
    behind the holder -- the call must come back, within a bounded time, with a right answer: the bits of one of the two engines."""
    import os, subprocess, sys, time, warnings
    from oem_amd import _lib as L
    import torch
    if engine == "symcoop":
        xtx, xty = _xtx_problem_host(4096, 8192, 5)
        xd = torch.as_tensor(xtx, device="cuda")
        kw = dict(penalty="lasso", nlambda=5, tol=1e-8)
        call = lambda: oa.oem_xtx(xd, xty, **kw)
        off = ("OEM_NO_SYMCOOP",)
    else:
        rng = np.random.default_rng(77)
        xh = rng.normal(size=(250, 16000)); yh = xh[:, :8] @ rng.uniform(0.5, 1.5, 8) + rng.normal(size=250)
        xd = torch.as_tensor(np.ascontiguousarray(xh.T), device="cuda").t()
        kw = dict(penalty="lasso", nlambda=5, tol=1e-8)

        def call():
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return oa.oem(xd, yh, **kw)
        off = ("OEM_NO_WCOOP", "OEM_NO_WSTREAM")
    alone = call()
    assert oa.last_path_engine()[0] == engine
    for k in off:
        os.environ[k] = "1"
    L.reload_switches()                                            # (the library parses its switches once: oem_amd/csrc/switches.hpp)
    try:
        launches = call()
    finally:
        for k in off:
            del os.environ[k]
        L.reload_switches()
    flag = tmp_path / "hold.flag"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    holder = subprocess.Popen([sys.executable, os.path.join(root, "tests", "hold_cus_worker.py"), "200", "6000", str(flag)], cwd=root,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        t0 = time.time()
        while not flag.exists() and time.time() - t0 < 120:
            time.sleep(0.05)
        assert flag.exists(), "the holder never started"
        time.sleep(0.3)
        t1 = time.time()
        during = call()
        wall = time.time() - t1
        held = flag.read_text() == "holding"                       # (still holding when the call came back?)
    finally:
        out, err = holder.communicate(timeout=120)
    assert "HOLD_DONE" in out, (out[-500:], err[-1500:])
    assert wall < 30.0, wall
    same_as = [name for name, ref in (("persistent", alone), ("launches", launches))
               if np.array_equal(np.asarray(during["beta"][0]), np.asarray(ref["beta"][0])) and np.array_equal(during["niter"][0], ref["niter"][0])]
    assert same_as, "the answer under contention matches neither engine"
    print(f"{engine} under contention: {wall:.2f} s, answer of the {same_as[0]} engine, holder still holding at return: {held}, "
          f"engine of the last attempt {oa.last_path_engine()}")


def _xtx_problem_host(p, n, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, p))
    b = np.zeros(p); b[:25] = rng.uniform(-1, 1, 25)
    y = x @ b + rng.normal(size=n)
    return x.T @ x / n, x.T @ y / n


@pytest.mark.parametrize("engine", ["wres", "wcoop", "wstream", "symcoop", "rowcoop", "coop"])
def test_interrupt_reaches_a_persistent_launch(oa, engine):
    """The persistent engines run the whole penalty x lambda path in ONE kernel (config 4: 23 ms; p >= n at maxit: seconds), and the
    reference polls for user interrupts every third lambda (ref src/oem_dense.cpp:235-238).  With an interrupt callback the kernel gets
    an abort word in host-coherent memory: the host waits for the launch by polling the stream and the callback (calling thread only),
    sets the word when the callback fires, every workgroup sees it within 128 iterations or inside the exchange it is waiting in, and
    the call returns OEMGPU_ERR_INTERRUPTED -- here within 50 ms of the callback's first True, out of a call that would run for
    seconds.  The context is as good as new: the next call returns the bits of the call before."""
    import threading
    import time
    import torch
    from oem_amd import _lib as L
    rng = np.random.default_rng(17)
    if engine in ("wres", "wcoop", "wstream"):
        n, p = {"wres": (500, 20000), "wcoop": (500, 2500), "wstream": (128, 200_000)}[engine]
        x = np.asfortranarray(rng.normal(size=(n, p)))
        y = x[:, :10] @ rng.uniform(1.0, 2.0, 10) + rng.normal(size=n)
        xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
        pens = ["lasso"] if engine == "wres" else ["lasso", "mcp"]
        call = lambda **kw: oa.oem(xd, y, penalty=pens, standardize=False, intercept=False, **kw)
    else:
        p = {"symcoop": 4096, "rowcoop": 2048, "coop": 1000}[engine]
        x = rng.normal(size=(p + p // 2, p))
        y = x[:, :25] @ rng.uniform(-1, 1, 25) + rng.normal(size=x.shape[0])
        xtx = torch.as_tensor(x.T @ x / x.shape[0], device="cuda")
        xty = x.T @ y / x.shape[0]
        call = lambda **kw: oa.oem_xtx(xtx, xty, penalty="lasso", **kw)
    small = dict(nlambda=4, tol=1e-8, maxit=300)
    long_ = dict(nlambda=100, tol=0.0, maxit=5000, lambda_min_ratio=1e-4)      # tol 0: every lambda runs into maxit -- seconds of iterations
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        before = call(**small)
        assert oa.last_path_engine()[0] == engine
        state = {"t0": None, "fired": None, "calls": 0, "tids": set()}

        def poll():
            now = time.perf_counter()
            state["calls"] += 1
            state["tids"].add(threading.get_ident())
            if state["t0"] is None:
                state["t0"] = now
            if now - state["t0"] >= 0.1 and state["fired"] is None:
                state["fired"] = now
            return state["fired"] is not None
        with pytest.raises(oa.OemgpuError) as e:
            call(interrupt=poll, **long_)
        t_back = time.perf_counter()
        assert e.value.code == L.ERR_INTERRUPTED
        assert oa.last_path_engine()[0] == engine                  # (it was the persistent launch that was interrupted, not a fallback)
        assert state["fired"] is not None and t_back - state["fired"] <= 0.05, t_back - state["fired"]
        # polled on the calling thread only, about once per millisecond (the 100 ms before it fires: ~100 calls on a quiet host; boxes of
        # this pool with a busy host have been seen at 25-47 -- the bound that matters is the 50 ms above)
        assert state["tids"] == {threading.get_ident()} and state["calls"] >= 10
        after = call(**small)
        again = call(interrupt=lambda: False, **small)             # (a callback that never fires changes nothing)
    for k in range(len(before["beta"])):
        assert np.array_equal(np.asarray(before["beta"][k]), np.asarray(after["beta"][k])) and np.array_equal(before["niter"][k], after["niter"][k])
        assert np.array_equal(np.asarray(before["beta"][k]), np.asarray(again["beta"][k]))


def test_a_timed_out_persistent_engine_is_not_tried_again_at_once(oa, monkeypatch):
    """a caller on a shared GPU paid the second of an exchange timeout on EVERY call, with no memory of the last failure.
    Now the context remembers: after a timeout the next 4 (.. 64, doubling) calls that would take a persistent engine go straight to
    the launch-per-iteration engines, for at most 30 s; a persistent launch that comes back, or a re-read of the switches, starts
    over.  (The timeout is the faked one: OEM_WCOOP_FAKE_TIMEOUT poisons the launch's result, as in the fallback tests.)"""
    import torch
    xtx, xty = _xtx_problem_host(2048, 3000, 8)
    xd = torch.as_tensor(xtx, device="cuda")
    kw = dict(penalty="lasso", nlambda=4, tol=1e-8)
    good = oa.oem_xtx(xd, xty, **kw)
    assert oa.last_path_engine()[0] == "rowcoop"
    f0 = oa.last_path_engine()[1]
    monkeypatch.setenv("OEM_WCOOP_FAKE_TIMEOUT", "1")
    first = oa.oem_xtx(xd, xty, **kw)                               # the persistent launch "times out": made again on the launches
    assert oa.last_path_engine() == ("launches", f0 + 1)
    # the fake stays on: what follows is the context's memory -- four calls that never try the persistent engine (no new timeout) ...
    skipped = [oa.oem_xtx(xd, xty, **kw) for _ in range(4)]
    assert oa.last_path_engine() == ("launches", f0 + 1)
    for f in [first] + skipped:
        assert np.abs(np.asarray(f["beta"][0]) - np.asarray(good["beta"][0])).max() < 1e-9
        assert np.array_equal(np.asarray(f["beta"][0]), np.asarray(first["beta"][0]))
    # ... and then it is tried again: it "times out" as well, and the back-off doubles
    again = oa.oem_xtx(xd, xty, **kw)
    assert oa.last_path_engine() == ("launches", f0 + 2)
    monkeypatch.delenv("OEM_WCOOP_FAKE_TIMEOUT")                     # the switches are read again without the fake: the memory starts over
    back = oa.oem_xtx(xd, xty, **kw)
    assert oa.last_path_engine() == ("rowcoop", f0 + 2)
    assert np.array_equal(np.asarray(back["beta"][0]), np.asarray(good["beta"][0])) and np.array_equal(np.asarray(again["beta"][0]), np.asarray(first["beta"][0]))


def _same_fit(a, b):
    return all(np.array_equal(np.asarray(a["beta"][k]), np.asarray(b["beta"][k])) and np.array_equal(a["niter"][k], b["niter"][k])
               for k in range(len(a["beta"])))


@pytest.mark.parametrize("case", ["lasso-256", "grp.lasso-512", "three-penalties-300", "scad-accelerate-loss-430"])
def test_cooperating_engine_on_one_xcd(oa, case, monkeypatch):
    """209 <= q <= 512: the <= 16 cooperating workgroups of an instance are placed on ONE XCD (eight times the workgroups are launched, the
    ones of the other seven XCDs leave at once), where the exchange is plain stores and sc1 loads through that XCD's L2 -- 0.65 us per
    all-gather instead of 1.05.  The arithmetic is the same: bit-identical to the device-scope exchange (OEM_NO_ONE_XCD), which is in turn
    held to the launch-per-iteration engines and the oracle elsewhere; the launch proves its placement before it relies on it, and a
    refused proof (faked: one workgroup reports another XCD) makes the call again at device scope -- same bits -- and this context does
    not ask again until the switches are re-read."""
    import torch
    rng = np.random.default_rng(77)
    if case == "lasso-256":
        q, kw = 256, dict(penalty="lasso", nlambda=30, tol=1e-9)
    elif case == "grp.lasso-512":
        q, kw = 512, dict(penalty="grp.lasso", groups=np.repeat(np.arange(1, 65), 8), nlambda=25, tol=1e-9)
    elif case == "three-penalties-300":
        q, kw = 300, dict(penalty=["lasso", "mcp", "grp.lasso"], groups=np.arange(300) // 6 + 1, nlambda=20, tol=1e-9)
    else:
        q, kw = 430, dict(penalty=["scad", "elastic.net"], alpha=0.6, nlambda=15, tol=1e-9, accelerate=True, compute_loss=True)
    n = 3000
    x = rng.normal(size=(n, q)) * rng.uniform(0.5, 2.0, q)
    b = np.zeros(q); b[:20] = rng.uniform(-1, 1, 20)
    y = x @ b + rng.normal(size=n)
    xd = torch.as_tensor(np.asfortranarray(x), device="cuda"); yd = torch.as_tensor(y, device="cuda")
    fit = oa.oem(xd, yd, **kw)
    assert oa.last_path_engine()[0] == "coop" and oa.api.last_placement() == "one-xcd"
    monkeypatch.setenv("OEM_NO_ONE_XCD", "1")
    ref = oa.oem(xd, yd, **kw)
    assert oa.last_path_engine()[0] == "coop" and oa.api.last_placement() == "none"
    assert _same_fit(fit, ref)
    monkeypatch.delenv("OEM_NO_ONE_XCD")
    f0 = oa.last_path_engine()[1]
    monkeypatch.setenv("OEM_FAKE_XCD_MISMATCH", "1")
    refused = oa.oem(xd, yd, **kw)
    assert oa.api.last_placement() == "refused" and oa.last_path_engine() == ("coop", f0)      # made again on the SAME engine: not a fallback
    assert _same_fit(refused, ref)
    after = oa.oem(xd, yd, **kw)                                     # the context remembers: not asked for again
    assert oa.api.last_placement() == "none" and _same_fit(after, ref)
    monkeypatch.delenv("OEM_FAKE_XCD_MISMATCH")                      # switches re-read: asked for again
    back = oa.oem(xd, yd, **kw)
    assert oa.api.last_placement() == "one-xcd" and _same_fit(back, ref)


def test_cooperating_engine_on_one_xcd_many_instances(oa, monkeypatch):
    """xval.oem's K + 1 fits as instances of ONE cooperating launch: instance y on XCD (first + y) mod 8, two instances on an XCD where
    there are more than eight -- identical to the device-scope exchange; and concurrent callers (threads) start at different XCDs."""
    import threading
    import torch
    rng = np.random.default_rng(78)
    n, q = 4000, 300
    x = rng.normal(size=(n, q)); b = np.zeros(q); b[:10] = rng.uniform(-1, 1, 10)
    y = x @ b + rng.normal(size=n)
    foldid = np.arange(n) % 10 + 1
    kw = dict(penalty="lasso", nlambda=12, tol=1e-8, foldid=foldid)
    xd = torch.as_tensor(np.asfortranarray(x), device="cuda"); yd = torch.as_tensor(y, device="cuda")
    fit = oa.xval_oem(xd, yd, **kw)
    placed = oa.api.last_placement()
    assert placed == "one-xcd"                                       # 11 instances x 10 workgroups: two instances on three of the XCDs
    monkeypatch.setenv("OEM_NO_ONE_XCD", "1")
    ref = oa.xval_oem(xd, yd, **kw)
    assert oa.api.last_placement() == "none"
    monkeypatch.delenv("OEM_NO_ONE_XCD")
    assert np.array_equal(np.asarray(fit["beta"][0]), np.asarray(ref["beta"][0])) and np.array_equal(np.asarray(fit["cvm"][0]), np.asarray(ref["cvm"][0]))
    # several host threads, each with its own context: their launches overlap on the device
    xs = [rng.normal(size=(2000, 260 + 20 * t)) for t in range(4)]
    ys = [xx[:, :5] @ np.ones(5) + rng.normal(size=2000) for xx in xs]
    single = [oa.oem(xx, yy, penalty="lasso", nlambda=10, tol=1e-8) for xx, yy in zip(xs, ys)]
    out, errs = [None] * 4, []

    def work(t):
        try:
            for _ in range(5):
                out[t] = oa.oem(xs[t], ys[t], penalty="lasso", nlambda=10, tol=1e-8)
        except Exception as e:                                        # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    for t in range(4):
        assert np.array_equal(np.asarray(out[t]["beta"][0]), np.asarray(single[t]["beta"][0]))
