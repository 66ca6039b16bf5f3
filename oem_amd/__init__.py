"""oem_amd -- MI355X-native Orthogonalizing-EM solver for penalized least squares.

Drop-in for the dense Gaussian hot path of jaredhuling/oem (oem(), oem.xtx(), big.oem()).
The numerics live in liboemgpu.so (hand-written HIP for gfx950, C ABI in include/oemgpu.h);
this package is the host-side mirror of the R front ends.  There is no CPU fallback.
"""
from .api import OemFit, PENALTIES, big_oem, context, last_path_engine, cv_oem, predict_cv, logLik, oem, oem_fit_dense_weighted, oem_xtx, predict, predict_xval, xval_oem  # noqa: F401
from .methods import format_summary, plot_cv, plot_oem, plot_xval, summary_cv, summary_xval  # noqa: F401
from ._lib import EXPORTS, LIB_PATH, OemgpuError, lib  # noqa: F401

__all__ = ["oem", "oem_xtx", "big_oem", "xval_oem", "cv_oem", "predict_cv", "predict", "predict_xval", "logLik", "plot_oem", "plot_cv", "plot_xval", "summary_cv", "summary_xval", "format_summary", "OemFit", "PENALTIES", "lib", "OemgpuError"]
