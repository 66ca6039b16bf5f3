// path_small.hip -- fused eigenvalue + penalty x lambda path for p <= 192: ONE launch, ONE workgroup.
//
// Replaces (ref paths under the reference tree):
//   Spectra::SymEigsSolver(nev=1, ncv=4).compute(10000, 1e-10), d = 1.005 lambda_max   src/oem_dense.h:485-498
//   A = d I - XX                                                                         src/oem_dense.h:501-505
//   oemBase::solve: beta_prev = beta; u = A beta_prev + XY; beta = T(u); stopRule        src/oem_base.h:90-110,
//                                                                                        src/oem_dense.h:508-653, src/utils.cpp:537-549
//   the penalty x lambda driver loops with warm starts                                  src/oem_dense.cpp:206-297
//
// Why one workgroup: the lambda path is a strictly serial chain of small dependent GEMVs (c1: ~730 rounds of a
// 100x100 GEMV).  Launch boundaries (~1.5-2 us each) or grid barriers (~4-7 us) would cost 10-30x the arithmetic,
// so the matrix lives in the VGPRs of one CU and a round costs one LDS exchange and ONE workgroup barrier:
//   * wave w holds columns [w CW, (w+1) CW) of all rows (lane = row mod 64, R rows per lane): a[R][CW] registers;
//   * every wave keeps the FULL current vector (R entries per lane) and does all O(p) work (threshold, stop
//     rule, dot products) redundantly and bit-identically, so no flag or scalar ever crosses waves;
//   * a round = write own vector to a wave-private LDS strip, broadcast-read the wave's CW entries, CW*R FMAs,
//     write R partial sums per lane, barrier, add the NW partials in fixed order.
//   * u = A beta does not depend on lambda, so the GEMV that follows convergence at lambda_i is exactly the first
//     GEMV of lambda_{i+1}: warm starts cost nothing extra.
// The eigenvalue step is an m-step Lanczos recurrence on the same register-resident matrix (no
// re-orthogonalisation: the top Ritz value still converges to lambda_max, Paige), followed by a 64-way
// multisection of the tridiagonal Sturm count, all redundantly per wave.
#include "common.hpp"

namespace oemgpu {

namespace {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);   // x+y == y+x: every lane ends with the same bits
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
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

// largest eigenvalue of the symmetric tridiagonal (al[0..m), be[0..m-1)) by 64-way multisection of the Sturm
// count; every lane of the wave returns the same value (an upper bracket end, so d never undershoots).
__device__ double tridiag_max(const double *al, const double *be, int m, int lane)
{
    if (m == 1) return al[0];
    double lo = -1e300, hi = -1e300;
    for (int j = lane; j < m; j += 64) {
        const double bl = j > 0 ? fabs(be[j - 1]) : 0.0, br = j < m - 1 ? fabs(be[j]) : 0.0;
        lo = fmax(lo, al[j]);
        hi = fmax(hi, al[j] + bl + br);
    }
    lo = wave_max(lo); hi = wave_max(hi);
    const double tiny = 1e-300;
    for (int round = 0; round < 12; ++round) {
        const double w = hi - lo;
        if (!(w > 4.0e-16 * fabs(hi))) break;
        const double th = lo + w * ((double)(lane + 1) / 65.0);
        double qv = al[0] - th;
        int neg = qv < 0.0;
        for (int k = 1; k < m; ++k) {
            if (qv == 0.0) qv = tiny;
            const double b = be[k - 1];
            qv = (al[k] - th) - b * b / qv;
            neg += qv < 0.0;
        }
        const int above = neg < m;                         // an eigenvalue >= th exists
        const int kk = __popcll(__ballot(above));          // monotone in the lane index
        const double nlo = kk == 0 ? lo : lo + w * ((double)kk / 65.0);
        const double nhi = kk == 64 ? hi : lo + w * ((double)(kk + 1) / 65.0);
        lo = nlo; hi = nhi;
    }
    return hi;
}

template <int R, int NW, int CW> struct Cfg {
    static constexpr int PR = 64 * R;       // padded rows
    static constexpr int PC = NW * CW;      // padded columns
    static constexpr int ML = 128;          // max Lanczos steps kept
    // LDS carve (doubles)
    static constexpr int OFF_P = 0;                         // partials [2][NW][PR]
    static constexpr int OFF_U = OFF_P + 2 * NW * PR;       // wave-private vector strip [NW][PR]
    static constexpr int OFF_F = OFF_U + NW * PR;           // wave-private group factors [NW][PR]
    static constexpr int OFF_T = OFF_F + NW * PR;           // wave-private Lanczos alpha/beta [NW][2][ML]
    static constexpr int OFF_I = OFF_T + NW * 2 * ML;       // ints: gstart[PR+1], gidx[PR], gzero[PR] (as int)
    static constexpr int N_DBL = OFF_I + (3 * PR + 4) / 2 + 2;
};

// one GEMV round: out = M vec, M = the register-resident matrix.  One workgroup barrier.
template <int R, int NW, int CW>
__device__ __forceinline__ void gemv_round(const double (&a)[R][CW], const double (&vec)[R], double (&out)[R],
                                           double *P, double *Uw, int w, int lane, int &buf)
{
    constexpr int PR = 64 * R;
#pragma unroll
    for (int r = 0; r < R; ++r) Uw[lane + 64 * r] = vec[r];
    double acc0[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }
    const v2d *bc = reinterpret_cast<const v2d *>(Uw + w * CW);     // same wave: DS ops execute in order
#pragma unroll
    for (int k = 0; k < CW; k += 2) {
        const v2d b = bc[k >> 1];                                    // uniform address: LDS broadcast
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc0[r] = fma(a[r][k], b.x, acc0[r]);
            acc1[r] = fma(a[r][k + 1], b.y, acc1[r]);
        }
    }
    double *Pb = P + buf * NW * PR;
#pragma unroll
    for (int r = 0; r < R; ++r) Pb[w * PR + lane + 64 * r] = acc0[r] + acc1[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double s = Pb[lane + 64 * r];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) s += Pb[ww * PR + lane + 64 * r];
        out[r] = s;
    }
    buf ^= 1;
}

template <int R, int NW, int CW>
__global__ __launch_bounds__(NW * 64) void path_small_kernel(PathArgs A)
{
    typedef Cfg<R, NW, CW> C;
    constexpr int PR = C::PR;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int p = A.p;
    double *P = lds + C::OFF_P;
    double *Uw = lds + C::OFF_U + w * PR;
    double *Fw = lds + C::OFF_F + w * PR;
    double *Tal = lds + C::OFF_T + w * 2 * C::ML, *Tbe = Tal + C::ML;
    int *gstart = reinterpret_cast<int *>(lds + C::OFF_I), *gidx = gstart + PR + 1, *gzero = gidx + PR;

    // ---- matrix slice and per-row vectors
    double a[R][CW];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = lane + 64 * r;
#pragma unroll
        for (int k = 0; k < CW; ++k) {
            const int col = w * CW + k;
            a[r][k] = (row < p && col < p) ? A.xx[(size_t)col * p + row] : 0.0;
        }
    }
    double xy[R], pf[R], sinv[R];
    int gid[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = lane + 64 * r;
        const bool ok = row < p;
        xy[r] = ok ? A.xy[row] : 0.0;
        pf[r] = ok ? A.pf[row] : 0.0;
        sinv[r] = (ok && A.sinv) ? A.sinv[row] : 1.0;
        gid[r] = (ok && A.ngroups > 0) ? A.gid[row] : -1;
    }
    const int ng = A.ngroups;
    if (ng > 0) {
        for (int g = tid; g <= ng; g += NW * 64) gstart[g] = A.gstart[g];
        for (int g = tid; g < ng; g += NW * 64) gzero[g] = A.gzero[g];
        const int nm = A.gstart[ng];
        for (int m = tid; m < nm; m += NW * 64) gidx[m] = A.gidx[m];
    }
    __syncthreads();
    int buf = 0;

    // ---- eigenvalue step: m-step Lanczos on XX
    double v[R], vp[R], wv[R];
    {
        double nn = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned row = lane + 64 * r;
            const unsigned h = row * 2654435761u + 12345u;               // deterministic non-structured start
            v[r] = (row < (unsigned)p) ? ((double)(h >> 8) * (1.0 / 16777216.0) - 0.5) : 0.0;
            vp[r] = 0.0;
            nn = fma(v[r], v[r], nn);
        }
        nn = 1.0 / sqrt(wave_sum(nn));
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] *= nn;
    }
    int msteps = A.lanczos_steps < C::ML ? A.lanczos_steps : C::ML;
    if (msteps > p) msteps = p;
    int nst = 0;
    double bprev = 0.0;
    for (int j = 0; j < msteps; ++j) {
        gemv_round<R, NW, CW>(a, v, wv, P, Uw, w, lane, buf);
        double al = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) al = fma(v[r], wv[r], al);
        al = wave_sum(al);
        double bb = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            wv[r] = (wv[r] - al * v[r]) - bprev * vp[r];
            bb = fma(wv[r], wv[r], bb);
        }
        bb = sqrt(wave_sum(bb));
        if (lane == 0) { Tal[j] = al; Tbe[j] = bb; }
        nst = j + 1;
        if (!(bb > 1e-13 * fabs(al))) break;            // invariant subspace reached: T is exact
        const double ib = 1.0 / bb;
#pragma unroll
        for (int r = 0; r < R; ++r) { vp[r] = v[r]; v[r] = wv[r] * ib; }
        bprev = bb;
    }
    const double theta = tridiag_max(Tal, Tbe, nst, lane);
    const double d = theta * 1.005;                       // ref src/oem_dense.h:498
    if (tid == 0) { A.d_out[0] = d; A.d_out[1] = theta; }

    // ---- A = d I - XX   (ref src/oem_dense.h:501-505)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = lane + 64 * r;
#pragma unroll
        for (int k = 0; k < CW; ++k) {
            const int col = w * CW + k;
            a[r][k] = ((row == col && row < p) ? d : 0.0) - a[r][k];
        }
    }

    // ---- lambda grid constants (ref src/oem_dense.cpp:175-192)
    const double scaley = A.yscale ? A.stats[1] : 1.0;
    const double yy = A.stats[2], nobs = A.stats[3];
    double lmax = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) lmax = fmax(lmax, fabs(xy[r]));
    lmax = wave_max(lmax) * scaley;
    const int nl = A.nl;
    const double llo = log(lmax), lhi = log(A.lambda_min_ratio * lmax);
    const double lstep = nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0;
    const bool lflip = fabs(lhi) < fabs(llo);

    double beta[R], bold[R], ab[R], u[R];
    for (int pp = 0; pp < A.npen; ++pp) {
        const int pen = A.penalty[pp];
        const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
        const bool isnet = pen_is_net(pen);
#pragma unroll
        for (int r = 0; r < R; ++r) { beta[r] = 0.0; ab[r] = 0.0; }     // cold start: A 0 = 0
        double ak = 1.0;
        for (int i = 0; i < nl; ++i) {
            // lambda_i (Eigen's setLinSpaced incl. its "flip" form, then exp; *.net: / alpha)
            double lam;
            if (A.user_lambda) lam = A.lambda_user[(size_t)pp * nl + i];
            else {
                double lv;
                if (nl == 1) lv = lhi;
                else if (lflip) lv = (i == 0) ? llo : lhi - (double)(nl - 1 - i) * lstep;
                else lv = (i == nl - 1) ? lhi : llo + (double)i * lstep;
                lam = exp(lv);
                if (isnet) lam = lam / A.alpha;
            }
            if (tid == 0) A.lambda_out[(size_t)pp * nl + i] = lam;
            if (i >= nlam) continue;
            const double il = lam / scaley;                               // ref src/oem_dense.cpp:241
            const PenK K = pen_consts(pen, il, d, A.alpha, A.gamma, A.tau);
            int it = 0, conv = 0;
            for (;;) {
#pragma unroll
                for (int r = 0; r < R; ++r) { bold[r] = beta[r]; u[r] = ab[r] + xy[r]; }
                // ---- beta = T(u)
                if (K.kind <= K_OLS) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double tp = pf[r] * K.L;
                        beta[r] = K.kind == K_SOFT ? soft1(u[r], tp, K.D)
                                : K.kind == K_MCP ? mcp1(u[r], tp, K.D, K.gamma)
                                : K.kind == K_SCAD ? scad1(u[r], tp, K.D, K.gamma)
                                                   : u[r] / d;
                    }
                } else {
                    double vv[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        vv[r] = (K.kind == K_SGL) ? soft1(u[r], pf[r] * K.L1, 1.0) : u[r];
                        Uw[lane + 64 * r] = vv[r];
                    }
                    for (int g = lane; g < ng; g += 64) {
                        double f = 1.0;
                        if (!gzero[g]) {
                            double s = 0.0;
                            for (int m = gstart[g]; m < gstart[g + 1]; ++m) { const double x = Uw[gidx[m]]; s += x * x; }
                            s = sqrt(s);
                            const double pen_g = K.L * A.gw[g];
                            if (K.kind == K_GRP || K.kind == K_SGL) { const double t = 1.0 - pen_g / s; f = (0.0 < t) ? t : 0.0; }
                            else if (K.kind == K_GRP_MCP) f = mcp_norm(s, pen_g, K.D, K.gamma);
                            else f = scad_norm(s, pen_g, K.D, K.gamma);
                        }
                        Fw[g] = f;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double f = gid[r] >= 0 ? Fw[gid[r]] : 0.0;
                        beta[r] = (f != 0.0) ? vv[r] * f / K.D : 0.0;
                    }
                }
                if (A.accelerate) {                                        // ref src/oem_dense.h:633-651
                    const double akp = ak;
                    ak = 0.5 * (1.0 + sqrt(1.0 + 4.0 * ak * ak));
                    const double ratio = (akp - 1.0) / ak;
                    double adp = 0.0;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double upd = beta[r], diff = upd - bold[r];
                        beta[r] = upd + ratio * diff;
                        adp += (beta[r] - upd) * diff;
                    }
                    if (wave_sum(adp) > 0.0) ak = 1.0;
                }
                ++it;
                // ---- stopRule, ref src/utils.cpp:537-549
                bool bad = false;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double c = fabs(beta[r]), q = fabs(bold[r]);
                    const bool cn = c > 1e-13, qn = q > 1e-13;
                    bad |= (cn != qn);
                    bad |= (cn && qn && fabs((beta[r] - bold[r]) / bold[r]) > A.tol);
                }
                conv = (__ballot(bad) == 0ull);
                if (conv || it >= A.maxit) break;
                gemv_round<R, NW, CW>(a, beta, ab, P, Uw, w, lane, buf);
            }
            // oemXTX::get_beta rescales the member in place (ref src/oem_xtx.h:576-581, quirk Q5)
            if (A.sinv) {
#pragma unroll
                for (int r = 0; r < R; ++r) beta[r] *= sinv[r];
            }
            if (w == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int row = lane + 64 * r;
                    if (row < p) A.beta[((size_t)pp * nl + i) * p + row] = beta[r];
                }
                if (lane == 0) A.niter[(size_t)pp * nl + i] = conv ? it : A.maxit + 1;   // ref src/oem_base.h:94-109
            }
            // warm start of the next lambda (and the loss) need A beta
            const bool last = (pp == A.npen - 1) && (i == nlam - 1);
            if (!last || A.compute_loss) gemv_round<R, NW, CW>(a, beta, ab, P, Uw, w, lane, buf);
            if (A.compute_loss) {
                // sum (Y - X beta)^2 on the standardised data (ref src/oem_dense.h:759-770) through the Gram identity
                // yy - 2 n beta'XY + n beta' XX beta, with XX beta = d beta - A beta
                double t = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) t += beta[r] * ((d * beta[r] - ab[r]) - 2.0 * xy[r]);
                t = wave_sum(t);
                if (tid == 0) A.loss[(size_t)pp * nl + i] = yy + nobs * t;
            } else if (tid == 0) A.loss[(size_t)pp * nl + i] = 1e99;
        }
    }
}

template <int R, int NW, int CW> int launch_cfg(hipStream_t s, const PathArgs &a)
{
    typedef Cfg<R, NW, CW> C;
    const size_t sh = (size_t)C::N_DBL * sizeof(double);
    if (sh > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&path_small_kernel<R, NW, CW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS %zu): %s", sh, hipGetErrorString(e)); return OEMGPU_ERR_HIP; }
    }
    hipLaunchKernelGGL((path_small_kernel<R, NW, CW>), dim3(1), dim3(NW * 64), sh, s, a);
    OEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int launch_path_small(hipStream_t s, const PathArgs &a)
{
    if (a.p <= 64) return launch_cfg<1, 4, 16>(s, a);
    if (a.p <= 128) return launch_cfg<2, 4, 32>(s, a);
    if (a.p <= 192) return launch_cfg<3, 8, 24>(s, a);
    set_error("path_small: p = %d exceeds %d", a.p, SMALL_P_MAX);
    return OEMGPU_ERR_INTERNAL;
}

}  // namespace oemgpu
