"""Inputs of the reference's documentation known-answer examples, regenerated from R's
set.seed(123) stream by oracle/r_rng.py (draw order: SURVEY.md Appendix B.1)."""
import functools

import numpy as np

from oracle.r_rng import RRng


@functools.lru_cache(maxsize=None)
def kat1():
    """docs/reference/logLik.html:166-175 (source R/methods.R:416-429)."""
    r = RRng(123)
    n, p = 2000, 50
    tb = np.concatenate([r.runif(15, -0.25, 0.25), np.zeros(p - 15)])
    x = np.asfortranarray(r.rnorm(n * p).reshape(p, n).T)
    y = r.rnorm(n, sd=3) + x @ tb
    return x, y


@functools.lru_cache(maxsize=None)
def kat2():
    """docs/reference/predict.oem.html:180-195 (source R/methods.R:25-46)."""
    r = RRng(123)
    n, p, nt = 10000, 100, 1000
    tb = np.concatenate([r.runif(15, -0.5, 0.5), np.zeros(p - 15)])
    x = np.asfortranarray(r.rnorm(n * p).reshape(p, n).T)
    y = r.rnorm(n, sd=3) + x @ tb
    xt = np.asfortranarray(r.rnorm(nt * p).reshape(p, nt).T)
    yt = r.rnorm(nt, sd=3) + xt @ tb
    return x, y, xt, yt


@functools.lru_cache(maxsize=None)
def kat3():
    """vignettes/oem_vignette.Rmd:398-425."""
    r = RRng(123)
    n, p = 50000, 100
    x = np.empty((n, p), order="F")
    for i in range(p):
        x[:, i] = r.rnorm(n) * (i + 1)
    y = r.rnorm(n) + x[:, 0] - x[:, 1]
    return x, y


def loglik(loss, n):
    """R/methods.R:465-466"""
    return -0.5 * n * (np.log(2 * np.pi) - np.log(n) + np.log(loss)) - 0.5 * n


@functools.lru_cache(maxsize=None)
def kat_xval():
    """docs/reference/predict.xval.oem.html (source R/methods.R, predict.xval.oem example): the data of kat2, then the
    fold assignment xval.oem draws itself, foldid = sample(rep(seq(nfolds), length = n)) (R/oem_xval.R:187-188)."""
    r = RRng(123)
    n, p, nt = 10000, 100, 1000
    tb = np.concatenate([r.runif(15, -0.5, 0.5), np.zeros(p - 15)])
    x = np.asfortranarray(r.rnorm(n * p).reshape(p, n).T)
    y = r.rnorm(n, sd=3) + x @ tb
    xt = np.asfortranarray(r.rnorm(nt * p).reshape(p, nt).T)
    yt = r.rnorm(nt, sd=3) + xt @ tb
    foldid = r.sample(np.resize(np.arange(1, 11), n)).astype(np.int32)
    return x, y, xt, yt, foldid
