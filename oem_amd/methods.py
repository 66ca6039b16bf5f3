"""plot and summary methods of the reference's S3 classes (R/methods.R:143-330, 841-1060), over the fit dictionaries of oem_amd.api.

These are pure consumers of the beta / lambda / cvm lists.  Every plot function first computes what R
draws -- the x index, one curve per non-zero coefficient, the degrees-of-freedom labels of the top axis, the vertical lines at
lambda.min / lambda.1se -- and returns it as a dict; it draws with matplotlib when that is importable (`show=False` skips drawing).
"""
import numpy as np


def _which(fit_pen, num_models, which_model):
    """R/methods.R:150-171: an index (0-based here) or a penalty name"""
    if isinstance(which_model, str):
        if which_model not in fit_pen:
            raise ValueError(f"Model {which_model} specified, but {which_model} not computed.")
        return fit_pen.index(which_model)
    if which_model >= num_models:
        raise ValueError(f"Model {which_model + 1} specified, but only {num_models} were computed.")
    return int(which_model)


def _approx_const(x, y, xout, f):
    """approx(x, y, xout, rule = 2, method = "constant", f = f) of R"""
    x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    o = np.argsort(x, kind="stable"); x, y = x[o], y[o]
    out = np.empty(len(xout))
    for i, v in enumerate(xout):
        if v <= x[0]:
            out[i] = y[0]
        elif v >= x[-1]:
            out[i] = y[-1]
        else:
            k = np.searchsorted(x, v, side="right") - 1
            out[i] = y[k] if x[k] == v else (1 - f) * y[k] + f * y[k + 1]
    return out


def _pretty(lo, hi, n=10):
    """a close relative of R's pretty(): about n round tick positions covering [lo, hi]"""
    if hi == lo:
        return np.array([lo])
    raw = (hi - lo) / n
    mag = 10.0 ** np.floor(np.log10(raw))
    step = min((1, 2, 5, 10), key=lambda m: abs(m * mag - raw)) * mag
    return np.arange(np.floor(lo / step) * step, np.ceil(hi / step) * step + step / 2, step)


def plot_oem(fit, which_model=0, xvar="norm", labsize=0.6, main=None, ax=None, show=True, _beta=None, _lam=None, _nzero=None):
    """plot.oem (R/methods.R:143-262): coefficient paths of one model against the L1 norm, lambda, log(lambda) or the loss."""
    if xvar not in ("norm", "lambda", "loglambda", "dev"):
        raise ValueError("'arg' should be one of 'norm', 'lambda', 'loglambda', 'dev'")
    betas = fit["beta"] if _beta is None else _beta
    m = _which(list(fit["penalty"]), len(betas), which_model)
    nbeta = np.asarray(betas[m])[1:, :]                                   # without the intercept row
    lam = np.asarray((fit["lambda"] if _lam is None else _lam)[m])[:nbeta.shape[1]]
    keep = ~np.all(nbeta == 0, axis=1)
    if not keep.any():
        raise ValueError("All beta estimates are zero for all values of lambda. No plot returned.")
    if xvar == "norm":
        index, xlab, rev, f = np.abs(nbeta).sum(axis=0), "L1 Norm", False, 1
    elif xvar == "lambda":
        index, xlab, rev, f = lam, "lambda", True, 0
    elif xvar == "loglambda":
        index, xlab, rev, f = np.log(lam), "log(lambda)", True, 1
    else:
        index, xlab, rev, f = np.asarray(fit["loss"][m], dtype=np.float64), "Sum of Squares", False, 1
    nz = (fit["nzero"] if _nzero is None else _nzero)[m]
    at = _pretty(float(np.min(index)), float(np.max(index)))
    names = list(fit.get("varnames", [f"V{j + 1}" for j in range(nbeta.shape[0])]))
    out = {"index": index, "curves": nbeta[keep], "labels": [names[j] for j in np.nonzero(keep)[0]], "xlab": xlab,
           "reversed_x": rev, "top_axis_at": at, "top_axis_df": _approx_const(index, nz, at, f),
           "main": fit["penalty"][m] if main is None else main}
    if show:
        try:
            import matplotlib
            matplotlib.use("Agg", force=False)
            import matplotlib.pyplot as plt
        except Exception:
            return out
        ax = ax or plt.gca()
        for c, lab in zip(out["curves"], out["labels"]):
            ax.plot(index, c, lw=1)
            if labsize > 0:
                ax.annotate(lab, (index[-1], c[-1]), fontsize=8 * labsize / 0.6)
        if rev:
            ax.invert_xaxis()
        ax.set_xlabel(xlab); ax.set_ylabel("beta hat"); ax.set_title(out["main"])
        top = ax.twiny(); top.set_xlim(ax.get_xlim()); top.set_xticks(at); top.set_xticklabels([str(int(v)) for v in out["top_axis_df"]])
        out["ax"] = ax
    return out


def _plot_cv_like(lam, cvm, cvup, cvlo, nzero, name, lmin, l1se, main, sign_lambda, ax, show):
    x = sign_lambda * np.log(np.asarray(lam))
    out = {"x": x, "cvm": np.asarray(cvm), "cvup": np.asarray(cvup), "cvlo": np.asarray(cvlo), "top_axis_df": np.asarray(nzero),
           "vlines": [sign_lambda * np.log(lmin), sign_lambda * np.log(l1se)], "ylab": name, "main": main,
           "xlab": ("-" if sign_lambda < 0 else "") + "log(lambda)"}
    if show:
        try:
            import matplotlib
            matplotlib.use("Agg", force=False)
            import matplotlib.pyplot as plt
        except Exception:
            return out
        ax = ax or plt.gca()
        ax.errorbar(x, out["cvm"], yerr=[out["cvm"] - out["cvlo"], out["cvup"] - out["cvm"]], fmt=".", color="dodgerblue", ecolor="darkgrey")
        for v in out["vlines"]:
            ax.axvline(v, ls="--", lw=2, color="firebrick")
        ax.set_xlabel(out["xlab"]); ax.set_ylabel(name); ax.set_title(main)
        out["ax"] = ax
    return out


def plot_cv(fit, which_model=0, sign_lambda=1, ax=None, show=True):
    """plot.cv.oem (R/methods.R:283-330): the cross-validation curve with its error bars, lambda.min and lambda.1se."""
    pens = list(fit["oem.fit"]["penalty"])
    m = _which(pens, len(fit["cvm"]), which_model)
    return _plot_cv_like(fit["lambda"][m], fit["cvm"][m], fit["cvup"][m], fit["cvlo"][m], fit["nzero"][m], fit["name"],
                         fit["lambda.min.models"][m], fit["lambda.1se.models"][m], pens[m], sign_lambda, ax, show)


def plot_xval(fit, which_model=0, type="cv", xvar="norm", sign_lambda=1, ax=None, show=True, **kw):
    """plot.xval.oem (R/methods.R:841-985): type = "cv": as plot.cv.oem; type = "coefficients": as plot.oem."""
    if type not in ("cv", "coefficients"):
        raise ValueError("'arg' should be one of 'cv', 'coefficients'")
    m = _which(list(fit["penalty"]), len(fit["beta"]), which_model)
    if type == "coefficients":
        return plot_oem(fit, m, xvar=xvar, ax=ax, show=show, **kw)
    k = len(fit["cvm"][m])
    return _plot_cv_like(np.asarray(fit["lambda"][m])[:k], fit["cvm"][m], fit["cvup"][m], fit["cvlo"][m], np.asarray(fit["nzero"][m])[:k],
                         fit["name"], fit["lambda.min.models"][m], fit["lambda.1se.models"][m], fit["penalty"][m], sign_lambda, ax, show)


def _summary(inner, cv):
    nvars = [(np.asarray(b) != 0).sum(axis=0) for b in inner["beta"]]           # R counts every row, the intercept included
    model = {"gaussian": "linear", "binomial": "logistic"}[inner["family"]]
    val = {"penalty": list(inner["penalty"]), "model": model, "n": inner.get("nobs"), "p": inner["nvars"],
           "lambda.min.models": np.asarray(cv["lambda.min.models"]), "lambda": cv["lambda"], "cve": cv["cvm"], "nvars": nvars,
           "type.measure": cv["name"]}
    if inner["family"] == "gaussian":
        val["sigma"] = [np.sqrt(c) for c in cv["cvm"]]
    return val


def summary_cv(fit):
    """summary.cv.oem (R/methods.R:992-1004)"""
    return _summary(fit["oem.fit"], fit)


def summary_xval(fit):
    """summary.xval.oem (R/methods.R:1012-1024)"""
    return _summary(fit, fit)


def format_summary(s, digits=None):
    """print.summary.cv.oem (R/methods.R:1035-1056) as a string"""
    digits = (2, 4, 2, 2, 3) if digits is None else tuple(np.resize(np.atleast_1d(digits), 5))
    lines = []
    for m, pen in enumerate(s["penalty"]):
        cve = np.asarray(s["cve"][m])
        i = int(np.argmin(cve))
        lines.append(f"{pen}-penalized {s['model']} regression with n={s['n']}, p={s['p']}")
        lines.append(f"At minimum cross-validation error (lambda={s['lambda.min.models'][m]:.{digits[1]}f}):")
        lines.append("-------------------------------------------------")
        lines.append(f"  Nonzero coefficients: {int(s['nvars'][m][i])}")
        lines.append(f"  Cross-validation error ({s['type.measure']}): {cve.min():.{digits[0]}f}")
        if s["model"] == "linear":
            lines.append(f"  Scale estimate (sigma): {np.sqrt(cve[i]):.{digits[4]}f}")
            lines.append("")
        if m + 1 < len(s["penalty"]):
            lines.append("<===============================================>")
            lines.append("")
    return "\n".join(lines)
