// penalty_ops.hpp -- device-side penalty operators shared by the small-p and large-p path engines.
// Restates (ref paths under the reference tree) src/oem_dense.h:76-315 (operators) and :527-628 (dispatch).
#pragma once

#include "common.hpp"

namespace oemgpu {

// u - clamp(u, -t, t): the soft-threshold numerator (u - t, u + t or 0) without divergent branches
__device__ __forceinline__ double shrink(double u, double t)
{
    return u - fmin(fmax(u, -t), t);
}

// num / den with den fixed per lambda and rden = 1/den precomputed: q = num*rden refined with one residual step.
// (a full fp64 divide is ~25 dependent instructions and there are several per coefficient per iteration on the
// serial critical path.)  The result is the correctly rounded quotient except for rare last-bit ties.
__device__ __forceinline__ double cdiv(double num, double den, double rden)
{
    const double q = num * rden;
    return fma(fma(-den, q, num), rden, q);
}

// ---- element-wise operators, ref src/oem_dense.h:76-149 -------------------------------------------------
__device__ __forceinline__ double soft1(double u, double tp, double d)
{
    if (u > tp) return (u - tp) / d;
    if (u < -tp) return (u + tp) / d;
    return 0.0;
}
__device__ __forceinline__ double mcp1(double u, double tp, double d, double gamma)
{
    const double gammad = gamma * d, dmg = d - 1.0 / gamma;
    if (fabs(u) > gammad * tp) return u / d;
    if (u > tp) return (u - tp) / dmg;
    if (u < -tp) return (u + tp) / dmg;
    return 0.0;
}
__device__ __forceinline__ double scad1(double u, double tp, double d, double gamma)
{
    const double gammad = gamma * d, gm1d = (gamma - 1.0) * d;
    if (fabs(u) > gammad * tp) return u / d;
    if (fabs(u) > (d + 1.0) * tp) {
        const double gp = (gamma - 1.0) * u, gq = gamma * tp;
        if (gp > gq) return (gp - gq) / (gm1d - 1.0);
        if (gp < -gq) return (gp + gq) / (gm1d - 1.0);
        return 0.0;
    }
    if (u > tp) return (u - tp) / d;
    if (u < -tp) return (u + tp) / d;
    return 0.0;
}
// ---- group factors, ref src/oem_dense.h:151-191, 277-315 --------------------------------------------------
__device__ __forceinline__ double scad_norm(double b, double pen, double d, double gamma)
{
    const double gammad = gamma * d, gm1d = (gamma - 1.0) * d;
    if (fabs(b) > gammad * pen) return 1.0;
    if (fabs(b) > (d + 1.0) * pen) {
        const double gp = gamma - 1.0, gq = gamma * pen / b;
        if (gp > gq) return d * (gp - gq) / (gm1d - 1.0);
        if (gp < -gq) return d * (gp + gq) / (gm1d - 1.0);
        return 0.0;
    }
    if (b > pen) return 1.0 - pen / b;
    if (b < -pen) return 1.0 + pen / b;
    return 0.0;
}
__device__ __forceinline__ double mcp_norm(double b, double pen, double d, double gamma)
{
    const double gammad = gamma * d, dmg = d - 1.0 / gamma;
    if (fabs(b) > gammad * pen) return 1.0;
    if (b > pen) return d * (1.0 - pen / b) / dmg;
    if (b < -pen) return d * (1.0 + pen / b) / dmg;
    return 0.0;
}

enum { K_SOFT = 0, K_MCP = 1, K_SCAD = 2, K_OLS = 3, K_GRP = 4, K_GRP_MCP = 5, K_GRP_SCAD = 6, K_SGL = 7 };

// per-lambda constants of next_beta's dispatch, ref src/oem_dense.h:527-628
struct PenK {
    int kind;
    double L;      // lambda' multiplying penalty_factor / group weight
    double D;      // denominator
    double L1;     // sparse.grp.lasso: tau * lambda (soft threshold, denominator 1)
    double gamma;
};
__device__ __forceinline__ PenK pen_consts(int pen, double lam, double d, double alpha, double gamma, double tau)
{
    PenK k; k.gamma = gamma; k.L1 = 0.0; k.L = lam; k.D = d; k.kind = K_SOFT;
    const double Ln = lam * alpha, Dn = d + (1.0 - alpha) * lam;
    switch (pen) {
    case OEMGPU_LASSO: k.kind = K_SOFT; break;
    case OEMGPU_OLS: k.kind = K_OLS; break;
    case OEMGPU_ELASTIC_NET: k.kind = K_SOFT; k.L = Ln; k.D = Dn; break;
    case OEMGPU_SCAD: k.kind = K_SCAD; break;
    case OEMGPU_SCAD_NET:
        k.kind = K_SCAD; k.L = Ln; k.D = Dn;
        if (alpha == 0.0) { k.L = 0.0; k.D = d + lam; }
        break;
    case OEMGPU_MCP: k.kind = K_MCP; break;
    case OEMGPU_MCP_NET: k.kind = K_MCP; k.L = Ln; k.D = Dn; break;
    case OEMGPU_GRP_LASSO: k.kind = K_GRP; break;
    case OEMGPU_GRP_LASSO_NET: k.kind = K_GRP; k.L = Ln; k.D = Dn; break;
    case OEMGPU_GRP_MCP: k.kind = K_GRP_MCP; break;
    case OEMGPU_GRP_SCAD: k.kind = K_GRP_SCAD; break;
    case OEMGPU_GRP_MCP_NET: k.kind = K_GRP_MCP; k.L = Ln; k.D = Dn; break;
    case OEMGPU_GRP_SCAD_NET: k.kind = K_GRP_SCAD; k.L = Ln; k.D = Dn; break;
    case OEMGPU_SPARSE_GRP_LASSO: k.kind = K_SGL; k.L = (1.0 - tau) * lam; k.L1 = tau * lam; break;
    default: break;
    }
    return k;
}

// pen_consts is a 15-way switch: on the serial path kernels a taken scalar branch costs ~80 cycles, so the dispatch
// is done once per penalty and the per-lambda constants follow from three coefficients with the SAME arithmetic:
// L = cL * lam, D = d + cD * lam (d itself when cD == 0), L1 = cL1 * lam.
struct PenLin {
    int kind;
    double cL, cD, cL1;
};
__device__ __forceinline__ PenLin pen_linear(int pen, double alpha, double tau)
{
    PenLin k; k.kind = K_SOFT; k.cL = 1.0; k.cD = 0.0; k.cL1 = 0.0;
    switch (pen) {
    case OEMGPU_LASSO: k.kind = K_SOFT; break;
    case OEMGPU_OLS: k.kind = K_OLS; break;
    case OEMGPU_ELASTIC_NET: k.kind = K_SOFT; k.cL = alpha; k.cD = 1.0 - alpha; break;
    case OEMGPU_SCAD: k.kind = K_SCAD; break;
    case OEMGPU_SCAD_NET:
        k.kind = K_SCAD; k.cL = alpha; k.cD = 1.0 - alpha;
        if (alpha == 0.0) { k.cL = 0.0; k.cD = 1.0; }
        break;
    case OEMGPU_MCP: k.kind = K_MCP; break;
    case OEMGPU_MCP_NET: k.kind = K_MCP; k.cL = alpha; k.cD = 1.0 - alpha; break;
    case OEMGPU_GRP_LASSO: k.kind = K_GRP; break;
    case OEMGPU_GRP_LASSO_NET: k.kind = K_GRP; k.cL = alpha; k.cD = 1.0 - alpha; break;
    case OEMGPU_GRP_MCP: k.kind = K_GRP_MCP; break;
    case OEMGPU_GRP_SCAD: k.kind = K_GRP_SCAD; break;
    case OEMGPU_GRP_MCP_NET: k.kind = K_GRP_MCP; k.cL = alpha; k.cD = 1.0 - alpha; break;
    case OEMGPU_GRP_SCAD_NET: k.kind = K_GRP_SCAD; k.cL = alpha; k.cD = 1.0 - alpha; break;
    case OEMGPU_SPARSE_GRP_LASSO: k.kind = K_SGL; k.cL = 1.0 - tau; k.cL1 = tau; break;
    default: break;
    }
    return k;
}
__device__ __forceinline__ PenK pen_from_linear(const PenLin &c, double lam, double d, double gamma)
{
    PenK k;
    k.kind = c.kind; k.gamma = gamma;
    k.L = (c.cL == 1.0) ? lam : lam * c.cL;          // pen_consts: lam, lam * alpha, (1 - tau) * lam
    k.D = (c.cD == 0.0) ? d : d + c.cD * lam;        // pen_consts: d, d + (1 - alpha) * lam
    k.L1 = c.cL1 * lam;
    return k;
}

}  // namespace oemgpu
