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
