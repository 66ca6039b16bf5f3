"""Pins the CPU oracle against the reference's own reproducible known answers."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import kat_inputs as K


def _printed_equal(v, expected, decimals):
    """R prints a vector with a common number of decimals (7 significant digits for the
    element that needs the most): equal iff every value rounds to the printed one."""
    return np.all(np.abs(np.asarray(v) - np.asarray(expected)) <= 0.5 * 10.0 ** (-decimals) * (1 + 1e-9))


def test_kat1_loglik_lasso_mcp(doc_kats):
    x, y = K.kat1()
    fit = orc.fit_dense(x, y, penalty=["lasso", "mcp"], compute_loss=True)
    k = doc_kats["kat1"]
    assert _printed_equal(K.loglik(fit["loss"][0], 2000), k["loglik_lasso"], 3)
    assert _printed_equal(K.loglik(fit["loss"][1], 2000), k["loglik_mcp"], 3)
    assert fit["niter"][0][0] == 1 and np.all(fit["beta"][0][1:, 0] == 0)       # Q13


def test_kat1b_cv_full_fit_25_lambda(doc_kats):
    x, y = K.kat1()
    fit = orc.fit_dense(x, y, penalty=["lasso", "mcp"], compute_loss=True, nlambda=25)
    k = doc_kats["kat1b"]
    assert _printed_equal(K.loglik(fit["loss"][0], 2000), k["loglik_lasso"], 3)
    assert _printed_equal(K.loglik(fit["loss"][1], 2000), k["loglik_mcp"], 3)


def test_kat2_predict_mse_lasso_grp_lasso(doc_kats):
    x, y, xt, yt = K.kat2()
    groups = np.repeat(np.arange(1, 11), 10)
    fit = orc.fit_dense(x, y, penalty=["lasso", "grp.lasso"], groups=groups, unique_groups=np.arange(1, 11),
                        nlambda=10)
    k = doc_kats["kat2"]
    for m, key in enumerate(["mse_lasso", "mse_grp_lasso"]):
        pred = fit["beta"][m][0][None, :] + xt @ fit["beta"][m][1:]
        mse = ((yt[:, None] - pred) ** 2).mean(0)
        assert _printed_equal(mse, k[key], 6), key


def test_kat3_big_vs_dense(doc_kats):
    x, y = K.kat3()
    groups = np.repeat(np.arange(1, 21), 5)
    dense = orc.fit_dense(x, y, penalty=["lasso", "grp.lasso"], groups=groups, unique_groups=np.arange(1, 21))
    # R/big_oem.R:226-259: intercept => groups gets a leading 0, unique.groups gains 0
    big = orc.fit_big(x, y, penalty=["lasso", "grp.lasso"], groups=np.concatenate([[0], groups]),
                      unique_groups=np.arange(0, 21))
    diff = np.abs(big["beta"][0] - dense["beta"][0]).max()
    assert float(f"{diff:.7g}") == doc_kats["kat3"]["max_abs_big_minus_dense_lasso"]


def test_prop_dense_equals_xtx(doc_kats):
    """R/oem_xtx.R:78-103: oem(standardize=F, intercept=F) == oem.xtx(crossprod(x)/n, crossprod(x,y)/n)."""
    x, y = K.kat1()
    n = x.shape[0]
    a = orc.fit_dense(x, y, penalty=["lasso", "mcp"], standardize=False, intercept=False)
    b = orc.fit_xtx(x.T @ x / n, x.T @ y / n, penalty=["lasso", "mcp"])
    for m in range(2):
        assert np.abs(a["beta"][m][1:] - b["beta"][m]).max() < 1e-13
        assert np.all(a["beta"][m][0] == 0)


@pytest.mark.parametrize("n,p", [(40, 100), (64, 64), (30, 31)])
def test_prop_wide_branch_equals_gram_form(n, p):
    """The p >= n branch (ref src/oem_dense.h:476-482,513-521: d from XXt/n, u = X'(Y - X b)/n + d b) is the same
    iteration as the n > p one written on the Gram (u = (dI - X'X/n) b + X'Y/n): the non-zero spectra of XXt and XtX
    coincide.  The reference holds no known answer for this branch; this ties it to the pinned n > p code."""
    rng = np.random.default_rng(17)
    x = np.asfortranarray(rng.normal(size=(n, p)) * 1.5)
    b = np.zeros(p); b[:6] = rng.uniform(1, 2, 6)
    y = x @ b + 0.2 * rng.normal(size=n)
    groups = np.arange(p) // 4 + 1
    kw = dict(penalty=["lasso", "scad", "grp.lasso"], groups=groups, unique_groups=np.unique(groups), nlambda=15, lambda_min_ratio=0.01, tol=1e-9,
              maxit=3000)
    a = orc.fit_dense(x, y, standardize=False, intercept=False, **kw)
    g = orc.fit_xtx(x.T @ x / n, x.T @ y / n, **kw)
    assert abs(a["d"] - g["d"]) < 1e-12 * g["d"]
    for m in range(3):
        assert np.allclose(a["lambda"][m], g["lambda"][m], rtol=1e-13)
        assert np.abs(a["beta"][m][1:] - g["beta"][m]).max() < 1e-9
        assert np.abs(a["niter"][m].astype(int) - g["niter"][m]).max() <= 1
        assert np.abs(g["beta"][m]).max() > 0.1


def test_kat4_xval_best_model_test_mse():
    """docs/reference/predict.xval.oem.html: xval.oem(lasso, grp.lasso; groups rep(1:10, each = 10); nlambda = 10), then the
    test-set MSE of the full-data fit at each model's lambda.min: 9.099371 (lasso), 9.091854 (grp.lasso = best.model)."""
    x, y, xt, yt, foldid = K.kat_xval()
    g = np.concatenate([[0], np.repeat(np.arange(1, 11), 10)])                   # R/oem_xval.R:279-283
    f = orc.xval_dense(x, y, foldid, penalty=["lasso", "grp.lasso"], groups=g, unique_groups=np.arange(0, 11), nlambda=10,
                       lambda_min_ratio=1e-4, tol=1e-7, maxit=500)
    mse = []
    for k in range(2):
        b = f["beta"][k][:, int(np.argmin(f["cvm"][k]))]
        mse.append(float(np.mean((yt - (xt @ b[1:] + b[0])) ** 2)))
    assert "%.6f" % mse[0] == "9.099371" and "%.6f" % mse[1] == "9.091854"
    assert np.min(f["cvm"][1]) < np.min(f["cvm"][0])                             # best.model is grp.lasso


def test_xval_against_fold_by_fold_numpy():
    """The restated xval against a direct numpy computation: every fold fit as its own problem (big.oem-style scaling of the
    other folds' rows, ref src/oem_xval_dense.h:791-853), then mean / sd of the per-observation errors."""
    rng = np.random.default_rng(4)
    n, p, K_ = 600, 12, 4
    x = np.asfortranarray(rng.normal(size=(n, p)) * 2 + 0.5)
    y = x[:, :3] @ np.array([1.0, -2.0, 0.5]) + rng.normal(size=n) + 1.0
    foldid = rng.permutation(np.resize(np.arange(1, K_ + 1), n)).astype(np.int32)
    f = orc.xval_dense(x, y, foldid, penalty=["lasso", "mcp"], nlambda=8, lambda_min_ratio=1e-3, tol=1e-10, maxit=2000,
                       type_measure="mae")
    err = np.zeros((2, n, 8))
    for k in range(1, K_ + 1):
        tr = foldid != k
        xs, ys = x[tr], y[tr]; m = xs.shape[0]
        sc = 1.0 / np.sqrt((xs ** 2).sum(0) / (m - 1))
        z = np.column_stack([np.ones(m), xs * sc])
        xx, xy = z.T @ z / m, z.T @ ys / m
        d = orc.eig_max(xx) * 1.005
        lam = np.stack([f["lambda"][0], f["lambda"][1]])
        pf = np.concatenate([[0.0], np.ones(p)])
        b, _ = orc.path(xx, xy, d, lam, penalty=["lasso", "mcp"], tol=1e-10, maxit=2000, penalty_factor=pf)
        te = foldid == k
        for m_ in range(2):
            coef = b[m_].T if b[m_].shape[0] == 8 else b[m_]                     # (q, nl)
            pred = x[te] @ (coef[1:] * sc[:, None]) + coef[0]
            err[m_, te] = np.abs(y[te][:, None] - pred)
    for m_ in range(2):
        assert np.allclose(f["cvm"][m_], err[m_].mean(0), rtol=1e-9)
        assert np.allclose(f["cvsd"][m_], err[m_].std(0, ddof=1) / np.sqrt(n), rtol=1e-8)


def test_prop_sparse_equals_dense_without_intercept_and_scaling():
    """docs/reference/oem.html: max(abs(fit$beta[[1]] - fits$beta[[1]])) = 1.58e-15 for a dense and a sparse copy of x with
    standardize = FALSE (intercept = FALSE there too: R/oem.R example); and with an intercept the converged sparse fit is the
    optimum of big.oem's formulation (same column scaling, unpenalised intercept), reached along a different iteration."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2)
    n, p = 4000, 30
    x = sp.random(n, p, density=0.05, random_state=3, format="csc", data_rvs=lambda k: rng.normal(size=k))
    b = np.zeros(p); b[:5] = [1, -1, 0.5, 2, -0.7]
    y = x @ b + rng.normal(size=n) * 0.5 + 0.3
    xd = np.asfortranarray(x.toarray())
    kw = dict(penalty=["lasso", "mcp"], nlambda=20, tol=1e-9)
    a = orc.fit_sparse(x, y, standardize=False, intercept=False, **kw)
    d = orc.fit_dense(xd, y, standardize=False, intercept=False, **kw)
    for k in range(2):
        assert np.abs(a["beta"][k] - d["beta"][k]).max() < 1e-13
    lam = [a["lambda"][0] * 0.5]
    g = orc.fit_big(xd, y, penalty=["lasso"], lambda_=lam, tol=1e-12, maxit=20000)
    h = orc.fit_sparse(x, y, penalty=["lasso"], lambda_=lam, tol=1e-12, maxit=20000)
    assert np.abs(g["beta"][0] - h["beta"][0]).max() < 1e-9


def test_sparse_groups_have_an_intercept_slot():
    """ref R/oem.R:296-338 + src/oem_sparse.h:465: with an intercept a sparse x gets p + 1 group entries (slot 0 = the
    intercept's unpenalised group 0).  The converged grp.lasso fit then is the optimum the dense path reaches (standardize =
    FALSE); one slot off, the last variable would be in no group and stay 0."""
    import scipy.sparse as sp
    rng = np.random.default_rng(47)
    n, p = 3000, 16
    x = sp.random(n, p, density=0.25, random_state=3, format="csc", data_rvs=lambda k: rng.normal(size=k))
    b = np.zeros(p); b[-4:] = [1.5, -1.0, 0.8, 2.0]; b[:2] = [0.5, -0.6]
    y = x @ b + rng.normal(size=n) * 0.3 + 0.4
    groups = np.arange(p) // 4 + 1
    dense = orc.fit_dense(np.asfortranarray(x.toarray()), y, penalty=["grp.lasso"], groups=groups, unique_groups=np.unique(groups),
                          nlambda=6, standardize=False, tol=1e-13, maxit=100000)
    rg, rug = orc.r_sparse_groups(groups, True)
    assert len(rg) == p + 1 and rg[0] == 0 and list(rug[:2]) == [0, 1]
    s = orc.fit_sparse(x, y, penalty=["grp.lasso"], groups=rg, unique_groups=rug, lambda_=dense["lambda"][0], standardize=False,
                       tol=1e-13, maxit=100000)
    assert np.abs(s["beta"][0][-1, 1:]).min() > 0.5
    assert np.abs(s["beta"][0] - dense["beta"][0]).max() < 1e-7
