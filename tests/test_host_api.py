"""Host-side logic of the Python mirror of the R front ends (no GPU): argument validation with the reference's
messages, lambda / group preprocessing, result consumers."""
import numpy as np
import pytest

from oem_amd import api


def _xy(n=40, p=6, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, p))
    return x, x[:, 0] + rng.normal(size=n)


@pytest.mark.parametrize("kw,msg", [
    (dict(lambda_min_ratio=0.0), "lambda.min.ratio must be between 0 and 1"),
    (dict(lambda_min_ratio=1.0), "lambda.min.ratio must be between 0 and 1"),
    (dict(nlambda=0), "nlambda must be a positive integer"),
    (dict(maxit=0), "maxit and irls.maxit should be positive"),
    (dict(tol=-1.0), "tol and irls.tol should be nonnegative"),
    (dict(weights=[1.0]), "weights not implemented yet."),
    (dict(penalty_factor=[1.0, 2.0]), "penalty.factor must have same length as number of columns in x"),
    (dict(penalty="grp.lasso", groups=[1, 2]), "groups must have same length as number of columns in x"),
    (dict(penalty="grp.lasso", groups=[1, 1, 2, 2, 3, 3], group_weights=[1.0]), "group.weights must have same length as the number of groups"),
    (dict(penalty=["lasso", "mcp"], lambda_=[[1.0, 0.5]]), "If list of lambda vectors is provided"),
    (dict(penalty=["lasso", "mcp"], lambda_=[[1.0, 0.5], [1.0]]), "All provided lambda vectors must have same length"),
    (dict(penalty="nope"), "'arg' should be one of"),
])
def test_oem_argument_errors(kw, msg):
    x, y = _xy()
    with pytest.raises(ValueError, match=msg.replace(".", r"\.").replace("(", r"\(")):
        api.oem(x, y, **kw)


def test_shape_errors():
    x, y = _xy()
    with pytest.raises(ValueError, match="x and y lengths do not match"):
        api.oem(x, y[:-1])
    with pytest.raises(ValueError, match="x must have at least two columns"):
        api.oem(x[:, :1], y)
    with pytest.raises(ValueError, match="x must have at least two columns"):
        api.oem(y, y)
    with pytest.raises(NotImplementedError):
        api.oem(x, (y > 0).astype(float), family="binomial")
    with pytest.raises(ValueError, match="xtx must be a square matrix"):
        api.oem_xtx(x, y)
    with pytest.raises(ValueError, match="xty must have length equal"):
        api.oem_xtx(x.T @ x, y)
    with pytest.raises(ValueError, match="scale.factor must be same length"):
        api.oem_xtx(x.T @ x, x.T @ y, scale_factor=[1.0, 2.0])
    with pytest.raises(ValueError, match="binomial case not implemented yet"):
        api.big_oem(x, y, family="binomial")


def test_penalty_matching_and_defaults():
    assert api._match_penalty(None) == ["elastic.net"]                 # R/oem.R:202-208: default = first choice only
    assert api._match_penalty("sparse") == ["sparse.grp.lasso"]         # partial matching like match.arg
    assert api._match_penalty(["lasso", "grp.mcp.net"]) == ["lasso", "grp.mcp.net"]
    with pytest.raises(ValueError):
        api._match_penalty("grp")                                       # ambiguous


def test_lambda_lists_are_sorted_decreasing():
    lam = api._lambda_list([0.1, 1.0, 0.5], 2)
    assert len(lam) == 2 and np.array_equal(lam[0], [1.0, 0.5, 0.1]) and np.array_equal(lam[1], lam[0])
    lam = api._lambda_list([[0.1, 1.0], [3.0, 2.0]], 2)
    assert np.array_equal(lam[0], [1.0, 0.1]) and np.array_equal(lam[1], [3.0, 2.0])
    assert all(len(l) == 0 for l in api._lambda_list((), 3))


def test_group_setup_dense_and_big():
    g, ug, gw = api._group_setup(["grp.lasso"], [3, 3, 1, 1, 2, 2], None, 6, False)
    assert np.array_equal(g, [3, 3, 1, 1, 2, 2]) and np.array_equal(ug, [1, 2, 3]) and gw.size == 0
    # big.oem with intercept: groups gets a leading 0 and unique.groups gains 0 (R/big_oem.R:226-259)
    g, ug, gw = api._group_setup(["grp.lasso"], [3, 3, 1, 1, 2, 2], None, 6, True)
    assert np.array_equal(g, [0, 3, 3, 1, 1, 2, 2]) and np.array_equal(ug, [0, 1, 2, 3])
    g, ug, gw = api._group_setup(["grp.lasso"], [1, 1, 2, 2, 3, 3], [1.0, 2.0, 3.0], 6, True)
    assert np.array_equal(ug, [0, 1, 2, 3]) and np.array_equal(gw, [0.0, 1.0, 2.0, 3.0])
    g, ug, gw = api._group_setup(["lasso"], [], None, 6, True)
    assert g.size == 0 and ug.size == 0 and gw.size == 0


def _fake_fit():
    beta = np.array([[1.0, 1.0, 1.0], [0.0, 0.5, 0.7], [0.0, 0.0, -0.2]])
    f = api.OemFit(beta=[beta], **{"lambda": [np.array([1.0, 0.5, 0.1])]}, loss=[np.array([10.0, 8.0, 7.0])],
                   niter=[np.array([1, 3, 4])], penalty=["lasso"], nobs=20, nvars=2, family="gaussian")
    return f


def test_predict_and_loglik():
    f = _fake_fit()
    assert np.array_equal(api.predict(f, type="coefficients"), f["beta"][0])
    nz = api.predict(f, type="nonzero")
    assert nz[0] is None and np.array_equal(nz[1], [2]) and np.array_equal(nz[2], [2, 3])     # row 1 always dropped (Q16)
    x = np.array([[1.0, 2.0], [0.0, 1.0]])
    assert np.allclose(api.predict(f, x, type="response"), np.column_stack([np.ones(2), x]) @ f["beta"][0])
    mid = api.predict(f, x, s=[0.75], type="response")
    assert np.allclose(mid[:, 0], 0.5 * (api.predict(f, x)[:, 0] + api.predict(f, x)[:, 1]))
    ll = api.logLik(f)
    assert np.allclose(ll, -0.5 * 20 * (np.log(2 * np.pi) - np.log(20) + np.log(f["loss"][0])) - 10)
    f["loss"] = [np.full(3, 1e99)]
    with pytest.raises(ValueError, match="compute.loss"):
        api.logLik(f)


def _oracle_fit():
    """a fit dictionary as oem_amd.api._decorate builds it, from the CPU oracle (no GPU needed for the consumers)"""
    from oracle import oracle as orc
    rng = np.random.default_rng(3)
    n, p = 400, 12
    x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :3] @ np.array([1.0, -2.0, 0.5]) + rng.normal(size=n)
    r = orc.fit_dense(x, y, penalty=["lasso", "mcp"], nlambda=15, compute_loss=True)
    fit = {"beta": r["beta"], "lambda": r["lambda"], "loss": r["loss"], "penalty": ["lasso", "mcp"], "family": "gaussian",
           "nobs": n, "nvars": p, "varnames": [f"V{j + 1}" for j in range(p)],
           "nzero": [(np.abs(b[1:]) > 0).sum(axis=0) for b in r["beta"]]}
    return fit


def test_plot_and_summary_methods():
    """plot.oem / plot.cv.oem / plot.xval.oem / summary.* (ref R/methods.R:143-330, 841-1056) over the fit dictionaries"""
    import oem_amd as oa
    fit = _oracle_fit()
    for xvar in ("norm", "lambda", "loglambda", "dev"):
        d = oa.plot_oem(fit, "mcp", xvar=xvar, show=False)
        assert d["curves"].shape[1] == len(d["index"]) == 15 and len(d["labels"]) == d["curves"].shape[0] <= 12
        assert d["reversed_x"] == (xvar in ("lambda", "loglambda")) and len(d["top_axis_at"]) == len(d["top_axis_df"])
    assert np.allclose(oa.plot_oem(fit, 0, show=False)["index"], np.abs(fit["beta"][0][1:]).sum(0))
    with pytest.raises(ValueError):
        oa.plot_oem(fit, "scad", show=False)
    with pytest.raises(ValueError):
        oa.plot_oem(fit, 2, show=False)
    drawn = oa.plot_oem(fit, 0, xvar="loglambda")                       # matplotlib (Agg) when it is there
    # a cross-validation object on top of it
    cvm = [np.linspace(3.0, 1.0, 15) ** 2, np.linspace(3.2, 1.1, 15) ** 2]
    cv = dict(fit, cvm=cvm, cvsd=[0.1 * c for c in cvm], name="Mean-Squared Error")
    cv["cvup"] = [a + b for a, b in zip(cv["cvm"], cv["cvsd"])]; cv["cvlo"] = [a - b for a, b in zip(cv["cvm"], cv["cvsd"])]
    cv.update(api._getmin(fit["lambda"], cv["cvm"], cv["cvsd"]))
    d = oa.plot_xval(cv, "lasso", show=False)
    assert len(d["x"]) == 15 and np.allclose(d["x"], np.log(fit["lambda"][0])) and len(d["vlines"]) == 2
    assert oa.plot_xval(cv, 1, type="coefficients", show=False)["main"] == "mcp"
    s = oa.summary_xval(cv)
    assert s["model"] == "linear" and s["n"] == 400 and s["p"] == 12 and np.allclose(s["sigma"][0], np.sqrt(cvm[0]))
    txt = oa.format_summary(s)
    assert txt.startswith("lasso-penalized linear regression with n=400, p=12") and "Scale estimate (sigma): 1.000" in txt
    assert "<===============================================>" in txt and "mcp-penalized" in txt
    cvo = {"oem.fit": fit, "cvm": cv["cvm"], "cvup": cv["cvup"], "cvlo": cv["cvlo"], "lambda": fit["lambda"], "nzero": fit["nzero"],
           "name": cv["name"], "lambda.min.models": cv["lambda.min.models"], "lambda.1se.models": cv["lambda.1se.models"]}
    assert np.allclose(oa.plot_cv(cvo, "mcp", sign_lambda=-1, show=False)["x"], -np.log(fit["lambda"][1]))
    assert oa.format_summary(oa.summary_cv(cvo)) == txt


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` on a node with fewer devices: one clear line on stderr and a non-zero exit code, from the parent
    process, before anything touches a GPU (it used to die on an AssertionError)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OEM_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "64"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and r.stdout == ""
    err = [l for l in r.stderr.splitlines() if l.strip()]
    assert len(err) == 1 and "--gpus 64" in err[0] and "Traceback" not in r.stderr


def test_persistent_wide_engine_scratch_holds_every_partition():
    """the exchange scratch of path_wcoop was sized at the LARGEST workgroup count although a set's size is not monotone
    in it (n = 100, p = 6,300: 104,724 doubles needed, 93,072 sized -- the launch refused itself).  Host arithmetic only: every
    (n, workgroup count) the engine can be asked for gets >= 1 set and never more sets than fit, on 256- and 304-CU devices."""
    import oem_amd
    L = oem_amd.lib()
    for n, p in ((100, 6300), (960, 2912), (192, 12200), (36, 2000), (500, 2000), (1000, 3000)):
        for npen in (1, 3, 8):
            assert L.oemgpu_selftest_wcoop_sizing(n, p, npen, 256) == 0, (n, p, npen)
    bad = []
    for n in range(1, 1025):
        cpg = 64 if n <= 256 else (32 if n <= 512 else 16)       # columns per workgroup by column height (path_wcoop.hip: wc_cw)
        for G in list(range(1, 193, 7)) + [99, 141, 150, 182, 191, 192]:
            p = max(n, G * cpg - 3)
            for npen, cu in ((1, 256), (8, 256), (8, 304), (2, 128)):
                if L.oemgpu_selftest_wcoop_sizing(n, p, npen, cu) != 0:
                    bad.append((n, p, npen, cu))
    assert not bad, bad[:10]


def test_moment_plan_deals_the_tile_columns_into_eights_sixes_and_a_four():
    """gram.hip: gram_plan / gram_sb_deal (host arithmetic): the shared-slab kernel's super-block rows cover every tile column, never
    cost more multiply time than eights alone (the deal before round 5: p = 160 multiplied 136 tiles' worth for 55 real ones), and from
    p = 177 on at most a fifth of it is padding (a quarter below: 129 <= p <= 176 is a six and a four, or an eight and a four); the row
    chunks cover n."""
    import ctypes as C
    import oem_amd
    L = oem_amd.lib()
    out = (C.c_int64 * 8)()
    worst = {}
    for p in list(range(1, 1300)) + [2048, 3000, 4096, 5000, 8192, 12000]:
        for n in (64, 5000, 100_000, 1_000_000):
            assert L.oemgpu_selftest_gram_plan(n, p, 256, out) == 0, L.oemgpu_last_error().decode()
            ntc, n8, n6, nchunk, steps, cost, real, n4 = list(out)
            assert nchunk >= 1 and steps >= 1 and nchunk * steps * 64 >= n, (n, p)
            if p + 2 <= 112:
                assert (n8, n6, n4) == (0, 0, 0) and ntc == (p + 2 + 15) // 16
                continue
            if (n8, n6, n4) == (-1, -1, -1):                                     # gram_wd.hip: the whole triangle of 15-16 tile columns in one workgroup
                nu = (ntc + 15) // 16
                assert (ntc % 16 in (15, 0) and cost == 136 * nu + 128 * nu * (nu - 1)) or (ntc in (11, 12) and cost == 80)     # (groups of three: eight waves x ten tile slots)
                worst[p] = real / cost
                assert nchunk % 8 == 0
                continue
            assert ntc == (p + 15) // 16 and 8 * n8 + 6 * n6 + 4 * n4 >= ntc and n4 <= 1 and n6 <= 1, (p, n8, n6, n4)
            assert 8 * n8 + 6 * n6 + 4 * n4 - ntc < 4, (p, n8, n6, n4)            # no super-block row is all padding
            e = (ntc + 7) // 8
            assert cost <= 64 * (e * (e - 1) // 2) + 36 * e, (p, n8, n6)
            assert nchunk % 8 == 0                                               # the kernel deals row chunks to XCDs in eights
            worst[p] = real / cost
    assert min(v for p, v in worst.items() if p >= 177) >= 0.8, min((v, p) for p, v in worst.items() if p >= 177)
    assert min(worst.values()) >= 0.75, min((v, p) for p, v in worst.items())
    assert worst[256] == 1.0 and worst[512] == 1.0 and worst[4096] == 1.0         # the configurations' sizes stay all eights
    # rounds of workgroups per CU: few where a workgroup would otherwise have a handful of row steps (the ring fill and the partials it
    # writes cost ~25 k cycles), several at the configurations' row counts
    for n, p, lo, hi in ((100_000, 128, 1, 1), (100_000, 256, 1, 2), (100_000, 384, 2, 4), (200_000, 300, 2, 4), (12_500_000, 256, 1, 12), (1_000_000, 512, 1, 12), (1_000_000, 640, 3, 12)):
        assert L.oemgpu_selftest_gram_plan(n, p, 256, out) == 0
        nsb = out[1] + out[2] + out[7]
        nu = (out[0] + 15) // 16
        rounds = out[3] * (nu * nu if out[1] < 0 else nsb * (nsb + 1) // 2) / 256.0
        assert lo - 0.1 <= rounds <= hi + 0.1, (n, p, rounds)


# ---------------------------------------------------------------------------------------------- the engine plan (api.hip: plan_paths)
_ENGINES = api.ENGINES
_SEM_DENSE, _SEM_BIG, _SEM_XTX, _SEM_XVAL = 0, 1, 2, 3


def _plan(p, pens, sem=_SEM_DENSE, intercept=0, groups=None, has_scale=False, nbatch=1, wide_n=0, num_cu=256, accelerate=False,
          compute_loss=False, nlambda=100, user_lambda=False):
    """oemgpu_selftest_plan: (engine name, frame bytes, reserved bytes, scratch need, scratch have) -- pure host arithmetic"""
    import ctypes as C
    from oem_amd import _lib as L
    q = p + (1 if (sem in (_SEM_BIG, _SEM_XVAL) and intercept) else 0)
    ngv = q if sem in (_SEM_BIG, _SEM_XVAL) else p
    if groups is None:
        g, ug, gw = [], [], []
    else:
        g = np.asarray(groups(ngv), dtype=np.int32)
        ug = np.unique(g); gw = []
    lam = [np.linspace(1.0, 0.1, 7)] * len(pens) if user_lambda else []
    a = api._Args(pens, lam, nlambda, 1e-4, 0.7, 3.0, 0.4, 1e-7, 500, accelerate, compute_loss, np.ones(p), g, ug, gw)
    eng, frame, res, need, have = C.c_int32(-1), C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
    rc = L.lib().oemgpu_selftest_plan(p, q, sem, intercept, C.byref(a.c), int(has_scale), nbatch, wide_n, num_cu, C.byref(eng), C.byref(frame),
                                      C.byref(res), C.byref(need), C.byref(have))
    assert rc == 0, (p, pens, sem, L.lib().oemgpu_last_error().decode())
    _plan.one_xcd = bool(eng.value & 256)                         # (the cooperating engine planned with every instance on one XCD)
    _plan.grp_head = (eng.value >> 9) & 3                         # (the launch engines' head form for group operators: window blocks on either side, 0 = update kernel)
    return _ENGINES[eng.value & 255], frame.value, res.value, need.value, have.value


_RUNS4 = lambda n: np.arange(n) // 4 + 1                    # groups of four neighbouring columns
_SCATTER = lambda n: np.arange(n) % 7 + 1                   # seven groups dealt round robin: no group is a run
_RUNS50 = lambda n: np.arange(n) // 50 + 1                  # runs of fifty: more than an owner's slice of the register engine holds (1024 < q <= 4096)
_RUNS7 = lambda n: np.arange(n) * 7 // n + 1                # seven runs of q / 7
_PEN_SETS = [(["lasso"], None), (["lasso", "mcp", "scad"], None), (["grp.lasso"], _RUNS4), (["grp.lasso", "lasso"], _RUNS4),
             (["grp.mcp", "sparse.grp.lasso"], _SCATTER), (["grp.lasso", "grp.scad.net"], _RUNS50), (["grp.lasso"], _RUNS7),
             (["elastic.net", "lasso", "ols", "mcp", "scad", "mcp.net", "scad.net", "grp.lasso"], _RUNS4)]


def test_the_plan_puts_the_cooperating_engine_on_one_xcd_only_where_its_instances_fit():
    """path_coop.hip's one-XCD form (q <= 512): instance y goes to XCD (first + y) mod 8, so ceil(instances / 8) workgroup sets share an
    XCD and must fit its num_cu / 8 CUs together; never beyond q = 512, never on another engine."""
    for num_cu in (64, 104, 128, 256, 304):
        for q in (150, 208, 209, 256, 300, 400, 512, 513, 700, 1024, 1025):
            for pens in (["lasso"], ["lasso", "mcp", "scad"], ["elastic.net", "lasso", "ols", "mcp", "scad", "mcp.net", "scad.net", "grp.lasso"]):
                for sem, nbatch in ((_SEM_DENSE, 1), (_SEM_XTX, 1), (_SEM_XVAL, 6), (_SEM_XVAL, 11)):
                    groups = _RUNS4 if "grp.lasso" in pens else None
                    try:
                        eng = _plan(q, pens, sem=sem, nbatch=nbatch, groups=groups, num_cu=num_cu)[0]
                    except AssertionError:
                        assert nbatch > 1                        # (batched fits the single launch does not take: xval.oem runs its folds on threads)
                        continue
                    if not _plan.one_xcd:
                        continue
                    assert eng == "coop" and 209 <= q <= 512 and num_cu % 8 == 0, (num_cu, q, pens, sem, nbatch, eng)
                    W = (q + 31) // 32
                    # (penalties side by side only when all their sets fit half the device; otherwise one set runs them in turn)
                    for ninst in {nbatch, nbatch * len(pens)}:
                        if W * ((ninst + 7) // 8) <= num_cu // 8:
                            break
                    else:
                        raise AssertionError((num_cu, q, pens, sem, nbatch))
    assert _plan(300, ["lasso"], num_cu=256)[0] == "coop" and _plan.one_xcd
    assert _plan(512, ["lasso"], sem=_SEM_XVAL, nbatch=11, num_cu=256)[0] == "coop" and _plan.one_xcd      # 11 x 16 workgroups: two sets on three XCDs
    assert _plan(400, ["lasso"], sem=_SEM_XVAL, nbatch=6, num_cu=128)[0] == "coop" and _plan.one_xcd        # 13 workgroups of the 16 CUs of an XCD
    assert _plan(300, ["lasso", "mcp", "scad"], num_cu=64)[0] == "coop" and not _plan.one_xcd              # 10 workgroups do not fit 8 CUs


def test_scattered_groups_are_reordered_into_runs():
    """api.hip: group_run_permutation (host arithmetic): groups of <= 32 members that are not runs of neighbouring coordinates are made runs --
    groups in the order of their first member, members in their own order, ungrouped coordinates where they stood between them -- so that
    the register-resident engine at 1024 < q <= 4096 takes them (ref src/oem_dense.h:421-456: the reference knows no group layout)."""
    import ctypes as C
    from oem_amd import _lib as L
    rng = np.random.default_rng(5)

    def perm_of(groups, q):
        g = np.asarray(groups, dtype=np.int32)
        a = api._Args(["grp.lasso"], [], 10, 1e-4, 1.0, 3.0, 0.5, 1e-7, 500, False, False, np.ones(q), g, np.unique(g), [])
        out = (C.c_int32 * q)()
        n = L.lib().oemgpu_selftest_group_permutation(C.byref(a.c), q, out)
        return np.array(out[:n])
    for q, groups in ((1500, np.arange(1500) % 75 + 1), (3000, rng.permutation(np.arange(3000) // 25 + 1)), (2000, np.arange(2000) % 100), (3000, rng.permutation(np.arange(3000) // 70))):
        pm = perm_of(groups, q)
        assert sorted(pm.tolist()) == list(range(q))                                   # a permutation
        gp = np.asarray(groups)[pm]
        starts = np.flatnonzero(np.r_[True, gp[1:] != gp[:-1]])
        assert len(starts) == len(np.unique(groups))                                   # every group one run
        first = [np.flatnonzero(np.asarray(groups) == gp[s])[0] for s in starts]
        assert first == sorted(first)                                                  # groups in the order of their first member
        for s_, e_ in zip(starts, np.r_[starts[1:], q]):
            assert np.all(np.diff(pm[s_:e_]) > 0)                                      # members in their own order
    assert len(perm_of(np.arange(1500) // 5 + 1, 1500)) == 0                           # runs already: nothing to do
    pm = perm_of(np.arange(3000) % 60 + 1, 3000)                                       # 50 members, more than an owner's slice holds: made runs all the same
    assert np.array_equal(pm, np.arange(3000).reshape(50, 60).T.ravel())


def test_groups_larger_than_an_owners_slice_are_dealt_in_fragments():
    """path_symcoop.hip: symcoop_plan (host arithmetic) for group runs at 1024 < q <= 4096: every owner holds <= 32 coordinates; a run of <= 32 is
    never cut; a longer run (a group of more than 32 members) is cut only after every fourth of its coordinates; and the fragment table the kernel
    sums a split group's norm by names, for every coordinate, the first owner of its run, how many owners the run lies in, and whether the run
    starts that owner's slice -- re-derived here from the slices alone (ref src/oem_dense.h:193-315: the reference sums a group's squares whatever
    its size)."""
    import ctypes as C
    from oem_amd import _lib as L
    rng = np.random.default_rng(11)
    lib = L.lib()

    def owners(q, sizes, num_cu=256):
        runs = np.r_[0, np.cumsum(sizes)].astype(np.int32)
        assert runs[-1] == q
        c0, cn, frag = (C.c_int32 * 192)(), (C.c_int32 * 192)(), (C.c_int32 * (2 * q))()
        G, sp = C.c_int32(-1), C.c_int32(-1)
        L.check(lib.oemgpu_selftest_symcoop_owners(q, num_cu, runs.ctypes.data_as(C.POINTER(C.c_int32)), len(sizes), c0, cn, frag, C.byref(G), C.byref(sp)))
        return runs, np.array(c0[:G.value]), np.array(cn[:G.value]), np.array(frag[:]).reshape(q, 2), sp.value

    def sizes_for(q, pick):
        out = []
        while sum(out) < q:
            out.append(int(pick()))
        out[-1] -= sum(out) - q
        return [s for s in out if s > 0]

    cases = [(3000, [50] * 60), (4096, [68] * 60 + [16]), (1536, [25, 26] * 30 + [6]), (2600, [520] * 5), (1100, [1100]),
             (2048, sizes_for(2048, lambda: rng.choice([1, 3, 20, 33, 47, 64, 120]))), (3777, sizes_for(3777, lambda: rng.integers(1, 400))),
             (4000, sizes_for(4000, lambda: rng.choice([31, 32, 33]))), (1300, [1] * 1300)]
    for q, sizes in cases:
        runs, c0, cn, frag, split = owners(q, sizes)
        assert len(c0) > 0, (q, sizes[:5])
        live = cn > 0
        assert cn.max() <= 32 and cn.sum() == q
        if not live.all():
            assert not live[np.argmin(live):].any()                                             # owners of nothing sit behind the others
        starts = c0[live]
        assert starts[0] == 0 and np.array_equal(starts[1:], (starts + cn[live])[:-1])          # the slices follow one another
        ends = np.r_[starts[1:], q]
        most = 0
        for r in range(len(sizes)):
            gs, ge = int(runs[r]), int(runs[r + 1])
            inside = starts[(starts > gs) & (starts < ge)]
            if ge - gs <= 32:
                assert len(inside) == 0, (q, r, gs, ge)                                         # a group that fits an owner is never cut
            else:
                assert np.all((inside - gs) % 4 == 0)                                           # cuts after every fourth coordinate at most
            a = int(np.searchsorted(starts, gs, side="right")) - 1
            b = int(np.searchsorted(starts, ge - 1, side="right")) - 1
            assert ends[b] >= ge and starts[a] <= gs
            k = b - a + 1
            most = max(most, k if k > 1 else 0)
            if split > 0:
                assert np.all(frag[gs:ge, 1] == k) and np.all(frag[gs:ge, 0] == 2 * a + (0 if starts[a] == gs else 1)), (q, r)
        assert split == most, (q, split, most)
    # the verdict's layout: sixty groups of fifty -- every group in a few owners' slices (the slices hold ~18 coordinates each)
    assert 2 <= owners(3000, [50] * 60)[4] <= 4
    assert owners(1536, [25] * 61 + [11])[4] == 0                                               # nothing to split: no table, no hop


def test_concurrent_one_xcd_launches_are_booked_per_xcd():
    """the CU slots of the one-XCD cooperating launches are booked per XCD, and a call's first XCD is the one where the load
    is lowest -- three calls of one 16-workgroup instance (q = 512) in flight at once must not meet on one XCD (the old turn counter put
    calls 0, 8 and 16 all on XCD 0: 48 workgroups on 32 CUs, an exchange timeout and a second of waiting)."""
    import ctypes as C
    from oem_amd import _lib as L
    lib = L.lib()
    def book(num_cu, W, ninst, calls):
        bases, peak = (C.c_int32 * calls)(), C.c_int32(-1)
        L.check(lib.oemgpu_selftest_coop_slots(num_cu, W, ninst, calls, bases, C.byref(peak)))
        return list(bases), peak.value
    for calls in (2, 3, 8):
        bases, peak = book(256, 16, 1, calls)
        assert len(set(bases)) == calls and peak == 16, (calls, bases, peak)          # every call an XCD of its own
    bases, peak = book(256, 16, 1, 12)
    assert peak == 32                                                                 # twelve on eight XCDs: two on four of them, never three
    bases, peak = book(256, 10, 11, 1)                                                # xval.oem's eleven instances: two on three XCDs
    assert peak == 20
    bases, peak = book(256, 10, 3, 4)                                                 # four callers x three penalties side by side
    assert peak <= 20
    assert lib.oemgpu_selftest_coop_slots(256, 16, 8, 2, (C.c_int32 * 2)(), C.byref(C.c_int32())) != 0      # 256 workgroups: the second call would wait


def test_the_plan_names_one_engine_and_a_workspace_that_fits_at_every_size():
    """VERDICT r4 item 6 / ADVICE r3: the launch of an engine must never reject the workspace its own caller sized.  plan_paths is the
    host function that decides engine and sizes for every call; this sweeps it without a GPU over q in [2, 20,000] (every q up to
    300, then the engines' boundaries and a coarse grid), penalty families (element-wise, groups that are runs, scattered groups,
    eight penalties), options (accelerate, compute.loss, scale.factor, user lambdas), the entry points' semantics, K + 1 batched
    fits, and devices of 64 .. 304 CUs."""
    qs = sorted(set(list(range(2, 301)) + [q + d for q in (512, 1024, 2048, 3457, 4096, 8192) for d in (-1, 0, 1)] +
                    list(range(320, 4200, 97)) + [5000, 12000, 20000]))
    seen = set()
    for p in qs:
        for pens, grp in _PEN_SETS:
            variants = [dict(), dict(accelerate=True, compute_loss=True), dict(user_lambda=True, nlambda=7)]
            if p % 5 == 0:
                variants += [dict(sem=_SEM_XTX, has_scale=True), dict(sem=_SEM_BIG, intercept=1), dict(sem=_SEM_XVAL, intercept=1, compute_loss=True),
                             dict(num_cu=304), dict(num_cu=64)]
            if p < 512 and p % 7 == 0:                                 # (xval.oem batches its K + 1 fits only where all their workgroup sets fit the chip)
                variants += [dict(sem=_SEM_XVAL, intercept=1, nbatch=6)]
            for kw in variants:
                eng, frame, res, need, have = _plan(p, pens, groups=grp, **kw)
                assert eng in _ENGINES[1:6], (p, pens, kw, eng)           # a Gram engine
                assert frame <= res, (p, pens, kw, eng, frame, res)
                seen.add(eng)
    assert seen == {"rows", "coop", "rowcoop", "symcoop", "launches"}
    # the sizes BASELINE.json names
    assert _plan(100, ["elastic.net"])[0] == "rows"                       # config 1
    assert _plan(200, ["mcp"])[0] == "rows" and _plan(512, ["grp.lasso"], groups=lambda n: np.arange(n) // 8 + 1)[0] == "coop"      # configs 2, 3
    assert _plan(4096, ["lasso"], sem=_SEM_XTX)[0] == "symcoop" and _plan(4096, ["lasso"], sem=_SEM_XTX, has_scale=True)[0] == "symcoop"      # config 4 (scale.factor: on the register-resident engine since round 5)
    assert _plan(257 - 1, ["lasso"], sem=_SEM_BIG, intercept=1)[0] == "coop"                                              # config 5: q = 257 (from 209 on)
    assert _plan(2048, ["lasso"])[0] == "rowcoop" and _plan(2048, ["grp.lasso"], groups=_RUNS4)[0] == "symcoop" and _plan(4097, ["lasso"])[0] == "launches"
    # the launch engines beyond 4096 (and behind the register engines below): group operators in the head of the (head, product) pairs for groups
    # of <= 96 members that are runs -- one, two or three 32-coordinate blocks on either side of a workgroup's own --, the update kernel beyond
    for q, grp, level in ((5000, _RUNS4, 1), (8192, lambda n: np.arange(n) // 32 + 1, 1), (8192, _RUNS50, 2), (6145, lambda n: np.arange(n) // 96 + 1, 3),
                          (8192, lambda n: np.arange(n) // 120 + 1, 0), (5000, _SCATTER, 0), (3000, _RUNS50, 2)):
        assert _plan(q, ["grp.lasso", "lasso"], groups=grp, accelerate=True, compute_loss=True)[0] in ("launches", "symcoop") and _plan.grp_head == level, (q, level, _plan.grp_head)
    assert _plan(5000, ["lasso"])[0] == "launches" and _plan.grp_head == 0
    # groups of more than 32 members stay on the register engine (their norms summed over several owners' slices)
    for q in (1025, 1536, 3000, 4096):
        assert _plan(q, ["grp.lasso"], groups=_RUNS50)[0] == "symcoop" and _plan(q, ["grp.mcp"], groups=_RUNS7, accelerate=True)[0] == "symcoop", q


def test_the_plan_for_p_ge_n_fits_the_scratch_the_callers_allocate():
    """the same for the two-product form (no Gram matrix): rows 3 .. 5,000 x columns up to 100,000 -- the persistent engines' exchange
    buffers against wide_scratch_doubles, the frame against the reservation"""
    seen = set()
    for n in (3, 40, 64, 100, 128, 192, 200, 256, 500, 700, 960, 1000, 1024, 1500, 2048, 2100, 5000):
        for p in sorted({n, n + 1, 2 * n + 3, 1100, 1500, 2500, 2912, 6300, 9000, 12200, 20000, 30000, 100000}):
            if p < n:
                continue
            for pens, grp in _PEN_SETS:
                for kw in (dict(), dict(accelerate=True, compute_loss=True), dict(num_cu=304), dict(num_cu=104), dict(sem=_SEM_BIG)):
                    eng, frame, res, need, have = _plan(p, pens, groups=grp, wide_n=n, **kw)
                    assert eng in _ENGINES[6:], (n, p, pens, kw, eng)
                    assert frame <= res, (n, p, pens, kw, eng, frame, res)
                    assert need <= have, (n, p, pens, kw, eng, need, have)
                    seen.add(eng)
    assert seen == {"wcoop", "wres", "wstream", "wlaunches"}
    assert _plan(20000, ["lasso"], wide_n=500)[0] == "wres" and _plan(2500, ["lasso", "mcp"], wide_n=500)[0] == "wcoop"
    assert _plan(200000, ["lasso"], wide_n=128)[0] == "wstream" and _plan(20000, ["lasso"], wide_n=2000)[0] == "wlaunches"


def test_every_switch_is_documented():
    """the library's environment switches live in ONE table (oem_amd/csrc/switches.hpp) that is parsed once; DESIGN.md section 7b must
    name every one of them, and no source may call getenv on its own"""
    import re
    from pathlib import Path
    from oem_amd import _lib as L
    root = Path(__file__).resolve().parent.parent
    names = L.lib().oemgpu_switch_names().decode().split()
    assert 28 <= len(names) <= 34 and len(set(names)) == len(names)     # (round 4: 45 names read by getenv at call time; round 5: 39; round 6: 33)
    design = (root / "DESIGN.md").read_text()
    sec = design[design.index("## 7b."):]
    sec = sec[:sec.index("\n## ", 5)]
    missing = [n for n in names if n not in sec]
    assert not missing, missing
    for src in sorted((root / "oem_amd" / "csrc").glob("*.h*")):
        if src.name == "switches.hpp":
            continue
        calls = [ln for ln in src.read_text().splitlines() if re.search(r"\bgetenv\(", ln) and "sw_parse" not in ln and "const char *e = getenv(name)" not in ln]
        assert not calls, (src.name, calls)
