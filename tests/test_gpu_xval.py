"""xval.oem on the GPU (oemgpu_xval_dense / oemgpu_xval_dense_dev) against the oracle's restatement of
ref src/oem_xval_dense.{h,cpp} and the reference's documented known answer."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import kat_inputs as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oa():
    import oem_amd
    return oem_amd


def _oracle(x, y, foldid, penalty, groups=None, intercept=True, **kw):
    if groups is not None:
        g = np.concatenate([[0], groups]) if intercept else np.asarray(groups)           # R/oem_xval.R:279-283
        ug = np.unique(np.concatenate([[0], groups])) if intercept else np.unique(groups)
        kw.update(groups=g, unique_groups=ug)
    return orc.xval_dense(x, y, foldid, penalty=penalty, intercept=intercept, **kw)


def _compare(f, r, npen, tol_b=1e-8, tol_cv=1e-9):
    assert abs(f["d"] - r["d"]) < 1e-11 * r["d"]
    for k in range(npen):
        assert np.allclose(f["lambda"][k], r["lambda"][k], rtol=1e-11)
        scale = max(1.0, float(np.abs(r["beta"][k]).max()))
        assert np.abs(f["beta"][k] - r["beta"][k]).max() < tol_b * scale
        assert np.allclose(f["cvm"][k], r["cvm"][k], rtol=tol_cv), (k, f["cvm"][k], r["cvm"][k])
        assert np.allclose(f["cvsd"][k], r["cvsd"][k], rtol=10 * tol_cv)


def test_kat4_doc_example(oa):
    """docs/reference/predict.xval.oem.html: test-set MSE at lambda.min 9.099371 (lasso) / 9.091854 (grp.lasso, best.model)."""
    x, y, xt, yt, foldid = K.kat_xval()
    f = oa.xval_oem(x, y, foldid=foldid, penalty=["lasso", "grp.lasso"], groups=np.repeat(np.arange(1, 11), 10), nlambda=10)
    assert f["best.model"] == "grp.lasso"
    # the calls of the documentation example: which.model = "best.model", "grp.lasso", 1
    mse = ["%.6f" % float(np.mean((yt - oa.predict_xval(f, xt, which_model=m, type="response")[:, 0]) ** 2))
           for m in ("best.model", "grp.lasso", 0)]
    assert mse == ["9.091854", "9.091854", "9.099371"]
    r = _oracle(x, y, foldid, ["lasso", "grp.lasso"], groups=np.repeat(np.arange(1, 11), 10), nlambda=10, lambda_min_ratio=1e-4)
    _compare(f, r, 2)


@pytest.mark.parametrize("std,icpt", [(True, True), (False, True), (True, False), (False, False)])
@pytest.mark.parametrize("measure", ["mse", "mae"])
def test_parity_small(oa, std, icpt, measure):
    rng = np.random.default_rng(11)
    n, p, nf = 3001, 23, 5                                       # ragged: folds of 601 / 600 rows, p + 1 not a multiple of 4
    x = np.asfortranarray(rng.normal(size=(n, p)) * 2 + 0.3)
    y = x[:, :4] @ np.array([1.0, -1.5, 0.5, 2.0]) + rng.normal(size=n) + 0.4
    foldid = rng.permutation(np.resize(np.arange(1, nf + 1), n))
    groups = np.arange(p) // 4 + 1
    pens = ["lasso", "mcp", "grp.lasso", "ols"]
    kw = dict(nlambda=21, tol=1e-9, maxit=1000, standardize=std)
    f = oa.xval_oem(x, y, foldid=foldid, penalty=pens, groups=groups, intercept=icpt, type_measure=measure, compute_loss=True, **kw)
    r = _oracle(x, y, foldid, pens, groups=groups, intercept=icpt, type_measure=measure, compute_loss=True,
                lambda_min_ratio=1e-4, **kw)
    _compare(f, r, 4)
    for k in range(4):
        assert np.allclose(np.ravel(f["loss"][k]), np.ravel(r["loss"][k]), rtol=1e-9)
        assert np.array_equal(np.ravel(f["niter"][k]), np.ravel(r["niter"][k]))


def test_uneven_folds_and_user_lambda(oa):
    rng = np.random.default_rng(12)
    n, p = 5000, 40
    x = np.asfortranarray(rng.normal(size=(n, p)))
    y = x[:, :5] @ rng.uniform(0.5, 1.5, 5) + 2 * rng.normal(size=n)
    foldid = np.concatenate([np.full(2500, 1), np.full(1700, 2), np.full(17, 3), np.full(783, 4)])    # contiguous, very uneven
    lam = [np.geomspace(1.0, 1e-3, 37), np.geomspace(2.0, 1e-2, 37)]
    f = oa.xval_oem(x, y, foldid=foldid, penalty=["elastic.net", "scad"], alpha=0.7, lambda_=lam, tol=1e-9)
    r = _oracle(x, y, foldid, ["elastic.net", "scad"], alpha=0.7, lambda_=lam, tol=1e-9)
    _compare(f, r, 2)
    with pytest.raises(oa.OemgpuError, match="foldid must hold values"):
        bad = foldid.copy(); bad[7] = 0
        oa.xval_oem(x, y, foldid=bad, penalty="lasso")


def test_device_resident_readme_shape(oa):
    """n = 2e5, p = 100, 10 folds, 100 lambdas on a device-resident X: the shape of the benchmark (config 1) at a size the
    oracle walks in seconds."""
    import torch
    rng = np.random.default_rng(13)
    n, p = 200_000, 100
    x = np.asfortranarray(rng.normal(size=(n, p)) * 3.0)
    b = np.concatenate([rng.uniform(size=25), np.zeros(75)])
    y = x @ b + rng.normal(size=n)
    foldid = rng.permutation(np.resize(np.arange(1, 11), n))
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    f = oa.xval_oem(xd, y, foldid=foldid, penalty="lasso", tol=1e-9)
    r = _oracle(x, y, foldid, ["lasso"], nlambda=100, lambda_min_ratio=1e-4, tol=1e-9)
    _compare(f, r, 1)
    imin = int(np.argmin(r["cvm"][0]))
    assert int(np.nonzero(f["lambda"][0] == f["lambda.min"])[0][0]) == imin


@pytest.mark.parametrize("route", ["one launch of cooperating workgroup sets", "a host thread and child context per fold"])
def test_large_p_takes_the_threaded_fold_fits(oa, monkeypatch, route):
    """p + 1 > 288: the K + 1 fits as workgroup sets of ONE cooperating-engine launch when they all fit the chip; otherwise (here:
    forced with OEM_NO_COOP) one host thread and child context per fold on the launch-per-iteration engines."""
    if route.startswith("a host thread"):
        monkeypatch.setenv("OEM_NO_COOP", "1")
    rng = np.random.default_rng(14)
    n, p = 3000, 300
    x = np.asfortranarray(rng.normal(size=(n, p)))
    y = x[:, :6] @ rng.uniform(0.5, 1.5, 6) + rng.normal(size=n) + 1.0
    foldid = rng.permutation(np.resize(np.arange(1, 4), n))
    kw = dict(nlambda=8, tol=1e-8, maxit=400, lambda_min_ratio=1e-2)
    f = oa.xval_oem(x, y, foldid=foldid, penalty=["lasso", "mcp"], **kw)
    r = _oracle(x, y, foldid, ["lasso", "mcp"], **kw)
    _compare(f, r, 2, tol_b=1e-7, tol_cv=1e-8)


def test_kat5_cv_oem_doc_example(oa):
    """docs/reference/predict.cv.oem.html: cv.oem(lasso, grp.lasso; groups rep(1:10, each = 10); nlambda = 10) with the folds R
    draws itself (the same RNG position as in the xval example), then the test-set MSE of the full-data oem() fit at lambda.min:
    9.091859 for "best.model" and "grp.lasso", 9.099376 for model 1."""
    x, y, xt, yt, foldid = K.kat_xval()
    cv = oa.cv_oem(x, y, penalty=["lasso", "grp.lasso"], groups=np.repeat(np.arange(1, 11), 10), nlambda=10, foldid=foldid)
    assert cv["best.model"] == "grp.lasso"
    mse = ["%.6f" % float(np.mean((yt - oa.predict_cv(cv, xt, which_model=m, type="response")[:, 0]) ** 2))
           for m in ("best.model", "grp.lasso", 0)]
    assert mse == ["9.091859", "9.091859", "9.099376"]


@pytest.mark.parametrize("n,p", [(5000, 40), (300, 1500)])
def test_cv_oem_folds_from_several_threads(oa, n, p):
    """cv.oem(parallel = TRUE) (R/cv_oem.R:129-150): the fold fits from three host threads at once -- n >> p fits overlap their
    one-CU path kernels with other folds' moment kernels, p >= n fits run side by side on the cooperating-workgroup engine --
    with the results of the sequential loop, bit for bit."""
    import time, warnings
    rng = np.random.default_rng(77 + p)
    x = np.asfortranarray(rng.normal(size=(n, p)))
    y = x[:, :5] @ rng.uniform(0.5, 1.5, 5) + rng.normal(size=n)
    foldid = rng.permutation(np.resize(np.arange(1, 7), n))
    kw = dict(penalty=["lasso", "mcp"], nlambda=12, tol=1e-7, foldid=foldid)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        oa.cv_oem(x, y, **kw)                                     # (contexts and staging buffers exist from here on)
        t0 = time.perf_counter(); a = oa.cv_oem(x, y, **kw); ta = time.perf_counter() - t0
        t0 = time.perf_counter(); b = oa.cv_oem(x, y, parallel=True, **kw); tb = time.perf_counter() - t0
    print(f"cv.oem n={n} p={p}, 6 folds: sequential {1e3 * ta:.1f} ms, three threads {1e3 * tb:.1f} ms")
    for m in range(2):
        assert np.array_equal(a["cvm"][m], b["cvm"][m]) and np.array_equal(a["cvsd"][m], b["cvsd"][m])
    assert a["lambda.min"] == b["lambda.min"] and a["best.model"] == b["best.model"]


def test_many_small_folds_and_an_empty_one(oa):
    """37 folds of ~54 rows (one of them empty: its id never occurs), n not a multiple of anything."""
    rng = np.random.default_rng(15)
    n, p, nf = 1999, 10, 37
    x = np.asfortranarray(rng.normal(size=(n, p)) + 0.2)
    y = x[:, :3] @ np.array([1.0, -1.0, 0.5]) + rng.normal(size=n)
    foldid = rng.permutation(np.resize(np.arange(1, nf), n))          # ids 1..36 only: fold 37 is empty
    foldid[0] = nf - 1
    f = oa.xval_oem(x, y, foldid=np.where(foldid == 5, nf, foldid), penalty=["lasso", "scad"], nlambda=16, tol=1e-9)     # fold 5 empty instead
    r = _oracle(x, y, np.where(foldid == 5, nf, foldid), ["lasso", "scad"], nlambda=16, lambda_min_ratio=1e-4, tol=1e-9)
    _compare(f, r, 2)


@pytest.mark.parametrize("std,icpt", [(True, True), (False, True), (True, False), (False, False)])
@pytest.mark.parametrize("measure", ["mse", "mae"])
def test_observation_weights(oa, std, icpt, measure):
    """xval.oem(weights = w) (ref src/oem_xval_dense.h:486-623, src/oem_xval_dense.cpp:389-437): weighted fold Grams with the
    intercept border carrying the weights, unweighted column scales and divisor, the error of row i times w_i -- device-resident
    and host-resident x, against the oracle (which tests/test_oracle_independent.py holds against scikit-learn)."""
    import torch
    rng = np.random.default_rng(21)
    n, p, nf = 4003, 19, 4
    x = np.asfortranarray(rng.normal(size=(n, p)) * 1.5 + 0.2)
    y = x[:, :4] @ np.array([1.0, -1.5, 0.5, 2.0]) + rng.normal(size=n) + 0.4
    w = rng.uniform(0.1, 4.0, n); w[:7] = 0.0                   # a few rows that only count in nobs and the column scales
    foldid = rng.permutation(np.resize(np.arange(1, nf + 1), n))
    groups = np.arange(p) // 4 + 1
    pens = ["lasso", "scad", "grp.lasso", "ols"]
    kw = dict(nlambda=15, tol=1e-9, maxit=2000, standardize=std)
    r = _oracle(x, y, foldid, pens, groups=groups, intercept=icpt, type_measure=measure, weights=w, lambda_min_ratio=1e-4, **kw)
    f = oa.xval_oem(x, y, foldid=foldid, penalty=pens, groups=groups, intercept=icpt, type_measure=measure, weights=w, **kw)
    _compare(f, r, len(pens))
    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
    g = oa.xval_oem(xd, y, foldid=foldid, penalty=pens, groups=groups, intercept=icpt, type_measure=measure, weights=w, **kw)
    _compare(g, r, len(pens))
    # unit weights are no weights
    a = oa.xval_oem(x, y, foldid=foldid, penalty=["lasso"], intercept=icpt, type_measure=measure, weights=np.ones(n), **kw)
    b = oa.xval_oem(x, y, foldid=foldid, penalty=["lasso"], intercept=icpt, type_measure=measure, **kw)
    assert np.abs(a["beta"][0] - b["beta"][0]).max() < 1e-10 and np.allclose(a["cvm"][0], b["cvm"][0], rtol=1e-10)
    with pytest.raises(oa.OemgpuError):                          # the reference's loss is unweighted: refused, not faked
        oa.xval_oem(x, y, foldid=foldid, penalty=["lasso"], weights=w, compute_loss=True, **kw)


def test_observation_weights_large_p(oa):
    """p + 1 > 288: the K + 1 weighted fits run one after the other on the cooperating-workgroup engine"""
    rng = np.random.default_rng(22)
    n, p, nf = 2400, 300, 3
    x = np.asfortranarray(rng.normal(size=(n, p)))
    y = x[:, :5] @ np.array([1.0, -1.5, 0.5, 2.0, -0.8]) + rng.normal(size=n) + 0.4
    w = rng.uniform(0.5, 2.0, n)
    foldid = rng.permutation(np.resize(np.arange(1, nf + 1), n))
    kw = dict(nlambda=6, tol=1e-9, maxit=3000)
    r = orc.xval_dense(x, y, foldid, penalty=["lasso", "mcp"], weights=w, lambda_min_ratio=1e-4, **kw)
    f = oa.xval_oem(x, y, foldid=foldid, penalty=["lasso", "mcp"], weights=w, **kw)
    _compare(f, r, 2, tol_b=1e-7, tol_cv=1e-7)


def test_cv_error_variance_with_a_tiny_spread(oa):
    """when the per-row errors barely vary next to their mean (mae on responses far from the fit), a one-pass
    sum v, sum v^2 loses the variance to cancellation; the kernel accumulates about a per-wave centre and merges by Chan's
    formula, like the reference's Welford update (ref src/oem_xval_dense.cpp:420-461)"""
    rng = np.random.default_rng(31)
    n, p, nf = 6000, 8, 3
    x = np.asfortranarray(rng.normal(size=(n, p)) * 1e-3)
    y = 1e4 + x @ rng.normal(size=p) + rng.normal(size=n) * 1e-4        # |y - yhat| ~ 1e4 +- 1e-4 without an intercept
    foldid = rng.permutation(np.resize(np.arange(1, nf + 1), n))
    kw = dict(penalty=["lasso"], nlambda=5, intercept=False, standardize=False, type_measure="mae")
    f = oa.xval_oem(x, y, foldid=foldid, **kw)
    r = orc.xval_dense(x, y, foldid, lambda_min_ratio=1e-4, **kw)
    assert np.allclose(f["cvm"][0], r["cvm"][0], rtol=1e-12)
    assert np.allclose(f["cvsd"][0], r["cvsd"][0], rtol=1e-6), (f["cvsd"][0], r["cvsd"][0])
