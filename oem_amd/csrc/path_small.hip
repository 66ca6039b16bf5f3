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
#include <cstdlib>
#include "common.hpp"
#include "penalty_ops.hpp"
#include "path_dev.hpp"

namespace oemgpu {

// granule exchange buffer of the cooperating-workgroup form: [2 parities][4 workgroups][<= 320 rows][2 granules]
size_t path_small_xchg_bytes() { return (size_t)2 * 4 * 320 * 2 * sizeof(unsigned long long); }

namespace {

// -DOEM_PATH_DIAG: a diagnostic build that splits the round into stamped segments (cycles summed per wave 0).
// Its fences forbid overlaps the real kernel has: read the SHARES, never the total.
#ifdef OEM_PATH_DIAG
__device__ unsigned long long g_diag[24];
#define OEM_STAMP(slot)                                                                    \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        unsigned long long t__;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        diag_acc[slot] += t__ - diag_last;                                                 \
        diag_last = t__;                                                                   \
    } while (0)
#define OEM_DIAG_DECL unsigned long long diag_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, diag_last = __builtin_amdgcn_s_memtime();
#define OEM_DIAG_ARGS , unsigned long long (&diag_acc)[24], unsigned long long &diag_last
#define OEM_DIAG_PASS , diag_acc, diag_last
#else
#define OEM_STAMP(slot) do { } while (0)
#define OEM_DIAG_DECL
#define OEM_DIAG_ARGS
#define OEM_DIAG_PASS
#endif

template <int R, int NW, int CW> struct Cfg {
    static constexpr int PR = 64 * R;       // padded rows
    static constexpr int PC = NW * CW;      // padded columns
    static constexpr int ML = 288;          // max Lanczos steps kept: as many as the largest p served, so the Krylov space can be exhausted
    // LDS carve (doubles)
    static constexpr int OFF_P = 0;                         // partials [2][NW][PR]
    static constexpr int OFF_U = OFF_P + 2 * NW * PR;       // wave-private vector strip [NW][PR]
    static constexpr int OFF_F = OFF_U + NW * PR;           // wave-private group factors [NW][PR]
    static constexpr int OFF_T = OFF_F + NW * PR;           // wave-private Lanczos alpha/beta [NW][2][ML]
    static constexpr int OFF_I = OFF_T + NW * 2 * ML;       // ints: gstart[PR+1], gidx[PR], gzero[PR] (as int)
    static constexpr int OFF_X = (OFF_I + (3 * PR + 4) / 2 + 2 + 1) / 2 * 2;   // sliced form: flags / scalars, 16-byte aligned
    static constexpr int N_DBL = OFF_X + 4 * NW + 64 * NW;  // (unused [NW]) | double aux [2][NW] | double sum [NW] | int flags [2][NW][64]
};

// ---- cooperating workgroups (G > 1: 192 < p <= 256).  Each workgroup holds a column slice of the matrix and ends a
// round with its partial sums of every row; the G partial vectors are exchanged through L2 as data-tagged 8-byte
// granules {tag = round epoch, 32 value bits} (guide 6 G16, recipe R2: one aligned atomic store per granule, the data IS
// the flag, no fence; polls are relaxed agent-scope loads that bypass L1).  Two buffers by round parity: a workgroup
// can publish round t+1 while a slower one still reads round t, and cannot reach t+2 without that one's t+1.
// Every spin is bounded; a timeout poisons the result (theta = -1) and lets every workgroup run out quickly.
struct Xchg {
    unsigned long long *buf;     // [2][G][PR][2] granules
    unsigned epoch;              // round counter, never 0; identical in every workgroup (all decisions are replicated)
    int gidx;                    // this workgroup's index
    bool failed;
};
typedef __attribute__((address_space(1))) unsigned long long gu64;

template <int R, int G>
__device__ __forceinline__ void exchange_partials(Xchg &X, double (&out)[R], double *scratch, int w, int lane)
{
    // wave 0 publishes this workgroup's partial sums; wave g (g < G, g != own) alone polls producer g -- one poller
    // per producer keeps the L2 polling traffic an eighth of "every wave polls everyone" -- and drops its vector into
    // LDS; after one workgroup barrier every wave adds the G vectors in producer order (identical bits everywhere).
    constexpr int PR = 64 * R;
    ++X.epoch;
    gu64 *base = (gu64 *)X.buf + (size_t)(X.epoch & 1) * G * PR * 2;
    if (w == 0) {
        gu64 *mine = base + (size_t)X.gidx * PR * 2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned lo = (unsigned)__double2loint(out[r]), hi = (unsigned)__double2hiint(out[r]);
            const size_t o = (size_t)(lane + 64 * r) * 2;
            __hip_atomic_store(mine + o, ((unsigned long long)X.epoch << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + o + 1, ((unsigned long long)X.epoch << 32) | hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (w < G) {
        if (w == X.gidx) {
#pragma unroll
            for (int r = 0; r < R; ++r) scratch[w * PR + lane + 64 * r] = out[r];
        } else {
            const gu64 *src = base + (size_t)w * PR * 2;
            unsigned long long v[2 * R];
            bool ok = false;
            for (unsigned spins = 0; !ok; ++spins) {
                bool all = true;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const size_t o = (size_t)(lane + 64 * r) * 2;
                    v[2 * r] = __hip_atomic_load(src + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[2 * r + 1] = __hip_atomic_load(src + o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    all &= ((unsigned)(v[2 * r] >> 32) == X.epoch) & ((unsigned)(v[2 * r + 1] >> 32) == X.epoch);
                }
                ok = __all(all);
                if (!ok && spins > 1000000u) break;                 // ~1 s: the partner is gone
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                scratch[w * PR + lane + 64 * r] = ok ? __hiloint2double((int)(unsigned)v[2 * r + 1], (int)(unsigned)v[2 * r])
                                                     : __builtin_nan("");   // poison: every wave sees the failure below
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < G; ++g) t += scratch[g * PR + lane + 64 * r];
        out[r] = t;
    }
    // a timed-out poller wrote NaNs: detect them uniformly (rows beyond p are exact zeros, real data is finite)
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; ++r) bad |= (out[r] != out[r]);
    if (__any(bad)) X.failed = true;
}

// one GEMV round: out = M vec, M = the register-resident matrix.  One workgroup barrier (plus, for G > 1, one
// granule exchange between the cooperating workgroups).
template <int R, int NW, int CW, int G>
__device__ __forceinline__ void gemv_round(const double (&a)[R][CW], const double (&vec)[R], double (&out)[R],
                                           double *P, double *Uw, int w, int lane, int &buf, Xchg &X OEM_DIAG_ARGS)
{
    constexpr int PR = 64 * R;
    OEM_STAMP(0);                       // everything since the previous round's partial sums (threshold, stop rule)
#pragma unroll
    for (int r = 0; r < R; ++r) Uw[lane + 64 * r] = vec[r];
    // all CW/2 broadcast reads are issued before the first FMA: one LDS latency per round instead of CW/2
    const v2d *bc = reinterpret_cast<const v2d *>(Uw + (X.gidx * NW + w) * CW);     // same wave: DS ops execute in order
    v2d b[CW / 2];
#pragma unroll
    for (int k = 0; k < CW / 2; ++k) b[k] = bc[k];                   // uniform address: LDS broadcast
    double acc0[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }
#pragma unroll
    for (int k = 0; k < CW; k += 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc0[r] = fma(a[r][k], b[k >> 1].x, acc0[r]);
            acc1[r] = fma(a[r][k + 1], b[k >> 1].y, acc1[r]);
        }
    }
    OEM_STAMP(1);                       // strip write, broadcast reads, FMAs
    double *Pb = P + buf * NW * PR;
#pragma unroll
    for (int r = 0; r < R; ++r) Pb[w * PR + lane + 64 * r] = acc0[r] + acc1[r];
    __syncthreads();
    OEM_STAMP(2);                       // partial write + workgroup barrier
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double t[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) t[ww] = Pb[ww * PR + lane + 64 * r];
#pragma unroll
        for (int h = 1; h < NW; h <<= 1)                       // fixed pairwise tree: short dependent chain
#pragma unroll
            for (int ww = 0; ww + h < NW; ww += 2 * h) t[ww] += t[ww + h];
        out[r] = t[0];
    }
    if (G > 1) exchange_partials<R, G>(X, out, P + (buf ^ 1) * NW * PR, w, lane);   // the idle half of P as scratch
    OEM_STAMP(3);                       // partial reads + adds (+ exchange)
    buf ^= 1;
}

// LDS words of a workgroup's cross-wave sums outside the round (cross_wave_sum)
struct SliceLds {
    double *P;          // partials [2][NW][PR]
    int *flag;          // [2][NW][64]: every lane of wave w stores the wave's flag in its own word
    double *aux;        // [2][NW]   (accelerate: partial inner products)
    double *sum;        // [NW]      (cross-wave scalar sums outside the round)
};

// Safe to reuse S.sum on every call: consecutive calls are separated by at least one round barrier.
template <int NW>
__device__ __forceinline__ double cross_wave_sum(double v, const SliceLds &S, int w, int lane)
{
    if (lane == 0) S.sum[w] = v;
    __syncthreads();
    double t[NW];
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) t[ww] = S.sum[ww];
#pragma unroll
    for (int h = 1; h < NW; h <<= 1)
#pragma unroll
        for (int ww = 0; ww + h < NW; ww += 2 * h) t[ww] += t[ww + h];
    return t[0];
}

// element-wise operators on N values per lane (ref src/oem_dense.h:76-149), branch-free
struct ThrK {
    double D, rD, gammad, dmg, rdmg, gm1, gamma, dsc, rdsc, d, rd;
};
template <int KIND>
__device__ __forceinline__ ThrK thr_consts(const PenK &K, double d)
{
    // only the reciprocals the operator uses: each is a ~300-cycle dependent chain paid once per lambda
    ThrK c = {};
    c.D = K.D; c.gammad = K.gamma * K.D; c.gamma = K.gamma; c.gm1 = K.gamma - 1.0; c.d = d;
    if (KIND != K_OLS) c.rD = 1.0 / K.D;
    if (KIND == K_MCP) { c.dmg = K.D - 1.0 / K.gamma; c.rdmg = 1.0 / c.dmg; }
    if (KIND == K_SCAD) { c.dsc = c.gm1 * K.D - 1.0; c.rdsc = 1.0 / c.dsc; }
    if (KIND == K_OLS) c.rd = 1.0 / d;
    return c;
}
template <int KIND>
__device__ __forceinline__ double threshold1(double u, double tp, const ThrK &c)
{
    if (KIND == K_SOFT) return cdiv(shrink(u, tp), c.D, c.rD);                // ref src/oem_dense.h:76-92
    if (KIND == K_MCP) {                                                      // ref src/oem_dense.h:94-117
        const bool big = fabs(u) > c.gammad * tp;
        const double num = big ? u : shrink(u, tp);
        return cdiv(num, big ? c.D : c.dmg, big ? c.rD : c.rdmg);
    }
    if (KIND == K_SCAD) {                                                     // ref src/oem_dense.h:119-149
        const double au = fabs(u);
        const bool big = au > c.gammad * tp, mid = !big && au > (c.D + 1.0) * tp;
        const double nmid = shrink(c.gm1 * u, c.gamma * tp);
        const double nsoft = shrink(u, tp);
        const double num = big ? u : (mid ? nmid : nsoft);
        return cdiv(num, mid ? c.dsc : c.D, mid ? c.rdsc : c.rD);
    }
    return cdiv(u, c.d, c.rd);                                                // K_OLS
}

// The operator's constants are the same in every lane: through v_readfirstlane they live in SGPRs, not in eleven VGPR pairs of
// a kernel whose eight-wave forms have none to spare (config 2's SCAD loop spilled into every round -- 21.7 instead of 6.1 ms --
// the moment an unrelated store elsewhere in the kernel moved the register allocation).
__device__ __forceinline__ double uniform_d(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ ThrK uniform_thr(const ThrK &c)
{
    ThrK u;
    u.D = uniform_d(c.D); u.rD = uniform_d(c.rD); u.gammad = uniform_d(c.gammad); u.dmg = uniform_d(c.dmg); u.rdmg = uniform_d(c.rdmg);
    u.gm1 = uniform_d(c.gm1); u.gamma = uniform_d(c.gamma); u.dsc = uniform_d(c.dsc); u.rdsc = uniform_d(c.rdsc); u.d = uniform_d(c.d);
    u.rd = uniform_d(c.rd);
    return u;
}

// The OEM iteration for one lambda (ref src/oem_base.h:90-110), specialised per operator so that the serial loop
// carries only the arithmetic of the penalty in use.  KIND == K_GRP covers every group penalty (K.kind selects).
template <int R, int NW, int CW, int G, int KIND>
__device__ __forceinline__ void iterate(const PathArgs &A, const PenK &K, double d, const double (&a)[R][CW],
                                        const double (&xy)[R], const double (&pf)[R], const int (&gid)[R],
                                        double (&beta)[R], double (&bold)[R], double (&ab)[R], double &ak, int &it,
                                        int &conv, double *P, double *Uw, double *Fw, const int *gstart,
                                        const int *gidx, const int *gzero, int ng, int w, int lane, int &buf, Xchg &X OEM_DIAG_ARGS)
{
    double tp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tp[r] = pf[r] * K.L;
    const double D = K.D, rD = 1.0 / K.D;
    const double gammad = K.gamma * D;
    const double dmg = D - 1.0 / K.gamma, rdmg = 1.0 / dmg;                  // mcp
    const double gm1 = K.gamma - 1.0, dsc = gm1 * D - 1.0, rdsc = 1.0 / dsc;  // scad
    const double tol = A.tol;
    double u[R];
    for (;;) {
#pragma unroll
        for (int r = 0; r < R; ++r) { bold[r] = beta[r]; u[r] = ab[r] + xy[r]; }
        if (KIND == K_SOFT) {                                   // ref src/oem_dense.h:76-92
#pragma unroll
            for (int r = 0; r < R; ++r) {
                beta[r] = cdiv(shrink(u[r], tp[r]), D, rD);
            }
        } else if (KIND == K_MCP) {                             // ref src/oem_dense.h:94-117
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool big = fabs(u[r]) > gammad * tp[r];
                const double num = big ? u[r] : shrink(u[r], tp[r]);
                beta[r] = cdiv(num, big ? D : dmg, big ? rD : rdmg);
            }
        } else if (KIND == K_SCAD) {                            // ref src/oem_dense.h:119-149
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double au = fabs(u[r]);
                const bool big = au > gammad * tp[r], mid = !big && au > (D + 1.0) * tp[r];
                const double gp = gm1 * u[r], gq = K.gamma * tp[r];
                const double nmid = shrink(gp, gq);
                const double nsoft = shrink(u[r], tp[r]);
                const double num = big ? u[r] : (mid ? nmid : nsoft);
                beta[r] = cdiv(num, mid ? dsc : D, mid ? rdsc : rD);
            }
        } else if (KIND == K_OLS) {
#pragma unroll
            for (int r = 0; r < R; ++r) beta[r] = cdiv(u[r], d, 1.0 / d);
        } else {                                                // group operators, ref src/oem_dense.h:193-315
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
        // A beta is needed whatever the stop rule says (next iteration, or the warm start of the next lambda), so the
        // GEMV round is issued first and the stop rule (ref src/utils.cpp:537-549; |(cur - prev) / prev| > tol written as
        // |cur - prev| > tol |prev|) resolves in the shadow of its LDS exchange instead of ahead of it.
        bool bad = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double c = fabs(beta[r]), q = fabs(bold[r]);
            const bool cn = c > 1e-13, qn = q > 1e-13;
            bad |= (cn != qn);
            bad |= (cn && qn && fabs(beta[r] - bold[r]) > tol * q);
        }
        gemv_round<R, NW, CW, G>(a, beta, ab, P, Uw, w, lane, buf, X OEM_DIAG_PASS);
        conv = (__ballot(bad) == 0ull);
        if (conv || it >= A.maxit) break;
    }
}

template <int R, int NW, int CW, int G>
__global__ __launch_bounds__(NW * 64) void path_small_kernel(PathArgs A_)
{
    const PathArgs A = path_instance(A_);
    static_assert(CW % 2 == 0, "the broadcast strip is read in pairs");
    typedef Cfg<R, NW, CW> C;
    constexpr int PR = C::PR;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int p = A.p;
    // whole-kernel clock reading (shader cycles and 100 MHz ticks), stored beside d: a diagnostic of the clock the
    // chip holds while a single CU runs the serial chain.  Not used by any result.
    const unsigned long long t_cyc0 = __builtin_amdgcn_s_memtime(), t_rt0 = __builtin_amdgcn_s_memrealtime();
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
            const int col = (blockIdx.x * NW + w) * CW + k;
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
    Xchg X;
    X.buf = reinterpret_cast<unsigned long long *>(A.work); X.epoch = 0; X.gidx = blockIdx.x; X.failed = false;
    const bool writer = (blockIdx.x == 0);          // workgroup 0 writes the results (all hold identical copies)
    OEM_DIAG_DECL

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
    int nst = 0;
    double bprev = 0.0;
    // Wave 0 evaluates the top Ritz value (the other waves wait at the barrier and read it from LDS).  From step 16
    // on it is checked every 8 steps and the recurrence stops once it has moved by less than 1e-14 (relative) since
    // the previous check (steps 16, 32, 48, then every 8): the top Ritz value of a Gram matrix typically settles in 30-50 steps, and OEM needs d only as an
    // upper bound of lambda_max that both sides compute alike (the fixed point does not depend on d).
    double *theta_slot = lds + C::OFF_X + 3 * NW;                   // S.sum[0..1]: free until the path starts
    auto top_ritz = [&](int m, double hint) {
        if (w == 0) {
            // scratch: the vector strips of all waves (only wave 0 is running; every round rewrites its strip)
            const double th = tridiag_max(Tal, Tbe, m, lane, lds + C::OFF_U, hint);
            if (lane == 0) theta_slot[0] = th;
        }
        __syncthreads();
        const double th = theta_slot[0];
        __syncthreads();                                            // the slot and the strips are reused
        return th;
    };
    double theta = 0.0, theta_prev = -__builtin_inf(), mv_prev = __builtin_inf();
    bool have_theta = false;
    for (int j = 0; j < msteps; ++j) {
        gemv_round<R, NW, CW, G>(a, v, wv, P, Uw, w, lane, buf, X OEM_DIAG_PASS);
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
        if (lanczos_check_due(nst) && nst < msteps) {
            OEM_STAMP(10);
            const double th = top_ritz(nst, theta_prev);
            OEM_STAMP(9);
            if (lanczos_converged(th, theta_prev, mv_prev)) { theta = th; have_theta = true; break; }
        }
        const double ib = 1.0 / bb;
#pragma unroll
        for (int r = 0; r < R; ++r) { vp[r] = v[r]; v[r] = wv[r] * ib; }
        bprev = bb;
    }
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 4; ++k) diag_acc[4 + k] = diag_acc[k];      // Lanczos share of the four segments
#endif
    OEM_STAMP(10);
    if (!have_theta) theta = top_ritz(nst, theta_prev);
    OEM_STAMP(9);
#ifdef OEM_PATH_DIAG
    diag_acc[11] = (unsigned long long)nst;
#endif
    const double d = theta * 1.005;                       // ref src/oem_dense.h:498
    if (tid == 0 && writer) {
        A.d_out[0] = d; A.d_out[1] = theta; A.d_out[4] = (double)nst; A.d_out[5] = (!have_theta && nst >= msteps && nst < p) ? 1.0 : 0.0;
        if (G == 1) A.d_out[6] = 0.0;                     // (G > 1: the launcher cleared the poison slot; a timed-out workgroup sets it)
    }

    // ---- A = d I - XX   (ref src/oem_dense.h:501-505)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = lane + 64 * r;
#pragma unroll
        for (int k = 0; k < CW; ++k) {
            const int col = (blockIdx.x * NW + w) * CW + k;
            a[r][k] = ((row == col && row < p) ? d : 0.0) - a[r][k];
        }
    }

    // ---- lambda grid constants (ref src/oem_dense.cpp:175-192)
    const double scaley = A.yscale ? A.stats[1] : 1.0;
    const double yy = A.stats[2], nobs = A.stats[3];
    double lmax = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = lane + 64 * r;
        const double xl = (A.lmax_xy && row < p) ? A.lmax_xy[row] : xy[r];
        lmax = fmax(lmax, (row >= A.lmax_from) ? fabs(xl) : 0.0);
    }
    lmax = wave_max(lmax) * scaley;
    const int nl = A.nl;
    const double llo = log(lmax), lhi = log(A.lambda_min_ratio * lmax);
    const double lstep = nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0;
    const bool lflip = fabs(lhi) < fabs(llo);

    SliceLds S;
    S.P = P;
    S.aux = lds + C::OFF_X + NW;
    S.sum = lds + C::OFF_X + 3 * NW;
    S.flag = reinterpret_cast<int *>(lds + C::OFF_X + 4 * NW);

    double beta[R], bold[R], ab[R];
    for (int pp = A.pen_lo; pp < A.pen_hi; ++pp) {
        const int pen = A.penalty[pp];
        const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
        const bool isnet = pen_is_net(pen);
#pragma unroll
        for (int r = 0; r < R; ++r) { beta[r] = 0.0; ab[r] = 0.0; }     // cold start: A 0 = 0
        double ak = 1.0;
        // user-supplied lambdas are fetched one lambda ahead: a dependent global load costs 1-2 us on this serial chain
        double lam_next = A.user_lambda ? A.lambda_user[(size_t)pp * nl] : 0.0;
        for (int i = 0; i < nl; ++i) {
            // lambda_i (Eigen's setLinSpaced incl. its "flip" form, then exp; *.net: / alpha)
            double lam;
            if (A.user_lambda) {
                lam = lam_next;
                if (i + 1 < nl) lam_next = A.lambda_user[(size_t)pp * nl + i + 1];
            } else {
                double lv;
                if (nl == 1) lv = lhi;
                else if (lflip) lv = (i == 0) ? llo : lhi - (double)(nl - 1 - i) * lstep;
                else lv = (i == nl - 1) ? lhi : llo + (double)i * lstep;
                lam = exp(lv);
                if (isnet) lam = lam / A.alpha;
            }
            if (tid == 0 && writer) A.lambda_out[(size_t)pp * nl + i] = lam;
            if (i >= nlam) continue;
            const double il = lam / scaley;                               // ref src/oem_dense.cpp:241
            const PenK K = pen_consts(pen, il, d, A.alpha, A.gamma, A.tau);
            int it = 0, conv = 0;
            const size_t orow = ((size_t)pp * nl + i);
            switch (K.kind) {
            case K_SOFT: iterate<R, NW, CW, G, K_SOFT>(A, K, d, a, xy, pf, gid, beta, bold, ab, ak, it, conv, P, Uw, Fw, gstart, gidx, gzero, ng, w, lane, buf, X OEM_DIAG_PASS); break;
            case K_MCP: iterate<R, NW, CW, G, K_MCP>(A, K, d, a, xy, pf, gid, beta, bold, ab, ak, it, conv, P, Uw, Fw, gstart, gidx, gzero, ng, w, lane, buf, X OEM_DIAG_PASS); break;
            case K_SCAD: iterate<R, NW, CW, G, K_SCAD>(A, K, d, a, xy, pf, gid, beta, bold, ab, ak, it, conv, P, Uw, Fw, gstart, gidx, gzero, ng, w, lane, buf, X OEM_DIAG_PASS); break;
            case K_OLS: iterate<R, NW, CW, G, K_OLS>(A, K, d, a, xy, pf, gid, beta, bold, ab, ak, it, conv, P, Uw, Fw, gstart, gidx, gzero, ng, w, lane, buf, X OEM_DIAG_PASS); break;
            default: iterate<R, NW, CW, G, K_GRP>(A, K, d, a, xy, pf, gid, beta, bold, ab, ak, it, conv, P, Uw, Fw, gstart, gidx, gzero, ng, w, lane, buf, X OEM_DIAG_PASS); break;
            }
            // the loss belongs to the coordinates the iteration ran in: before any in-place rescale (oemSparse takes it after
            // get_beta, where the rescaled intercept slot gives the same fitted values, ref src/oem_sparse.h:897-944)
            if (A.compute_loss) {
                // sum (Y - X beta)^2 on the standardised data (ref src/oem_dense.h:759-770) through the Gram identity
                // yy - 2 n beta'XY + n beta' XX beta, with XX beta = d beta - A beta
                double t = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) t += beta[r] * ((d * beta[r] - ab[r]) - 2.0 * xy[r]);
                t = wave_sum(t);
                if (tid == 0 && writer) A.loss[orow] = yy + nobs * t;
            } else if (tid == 0 && writer) A.loss[orow] = 1e99;
            // oemXTX::get_beta rescales the member in place (ref src/oem_xtx.h:576-581, quirk Q5)
            if (A.sinv) {
#pragma unroll
                for (int r = 0; r < R; ++r) beta[r] *= sinv[r];
            }
            if (w == 0 && writer) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int row = lane + 64 * r;
                    if (row < p) A.beta[orow * p + row] = beta[r];
                }
                if (lane == 0) A.niter[orow] = conv ? it : A.maxit + 1;   // ref src/oem_base.h:94-109
            }
            // ab = A beta (warm start of the next lambda, and the loss) is already there from the last round, unless
            // beta has just been rescaled in place
            if (A.sinv) gemv_round<R, NW, CW, G>(a, beta, ab, P, Uw, w, lane, buf, X OEM_DIAG_PASS);
        }
    }
#ifdef OEM_PATH_DIAG
    if (tid == 0) for (int k = 0; k < 24; ++k) g_diag[k] = diag_acc[k];
#endif
    if (tid == 0 && writer) {
        A.d_out[2] = (double)(__builtin_amdgcn_s_memtime() - t_cyc0);
        A.d_out[3] = (double)(__builtin_amdgcn_s_memrealtime() - t_rt0);
    }
    if (X.failed && lane == 0) A.d_out[6] = 1.0;      // exchange timeout: poison slot, cleared by the host before the launch (the host turns it into an error)
}

// ================================================================================================
// Row-split form (64 < p <= 128, element-wise penalties): the fastest round for these sizes.
//
// Measured on MI355X (tools/valu_probe.hip, tools/path_cost.py): a round costs (FP64 VALU instructions of the
// busiest SIMD) x ~5 cycles PLUS the exposed LDS exchange (write -> barrier -> read: ~190-260 cycles, ~25 more per
// extra 8-byte read per lane), with no overlap between the two -- the chain is serial.  So the round that wins issues
// the fewest instructions and moves the fewest words:
//   * four waves, one per SIMD; wave w owns rows [w RWp, (w+1) RWp), RWp = ceil(p/4) <= 32;
//   * the 16-lane row group g of every wave multiplies the column slice [g CGp, (g+1) CGp), CGp = ceil(p/4): lane
//     (g, l) holds a[r][k] = M[row(l, r)][g CGp + k] for its two row slots r = 0, 1 -- 2 CG FMAs, each taking its
//     beta entry from a neighbour lane (v_fmac_f64_dpp row_newbcast);
//   * the four slice sums of a row meet WITHOUT LDS, and as a reduce-scatter: one v_permlane16_swap pair + add leaves
//     slot 0's pair sums in even row groups and slot 1's in odd ones, one v_permlane32_swap pair + add finishes, so
//     lane (g, l) ends with the full (M beta) of ONE row, 16 (g & 1) + l, (g0 + g1) + (g2 + g3) in every lane;
//   * threshold, stop rule and warm start then touch ONE register per lane (rows replicated in groups 2, 3);
//   * what crosses waves is the NEW beta, not partial sums: one 8-byte store per row, one barrier, two 8-byte reads
//     per lane (the entries of its row group's slice).  Stop flags ride along as one 4-byte word per lane.
// The eigenvalue step runs the same exchange; its two inner products per Lanczos step cross waves through per-lane
// words and DPP butterflies (fixed order, identical in every lane).
// ================================================================================================
template <int CTRL> __device__ __forceinline__ double dpp_xchg(double v) { return dpp_mov<CTRL, 0xf>(v, 0.0); }
// lane l holds the word of wave l mod NW (NW = 4 or 8): sum over aligned groups of NW lanes, every lane gets the total
template <int NW> __device__ __forceinline__ double lanes_sum(double v)
{
    v += dpp_xchg<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_xchg<0x4E>(v);      // quad_perm [2,3,0,1]
    if (NW == 8) v += dpp_xchg<0x141>(v);     // row_half_mirror
    return v;
}
// sum of a per-row value over the wave's rows (row groups 0, 1; groups 2, 3 hold replicas; padding lanes hold 0):
// butterfly inside the 16-lane row, then one v_permlane16_swap pair joins the two groups.  Every lane gets the total.
__device__ __forceinline__ double rows_sum(double v)
{
    v += dpp_xchg<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_xchg<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_xchg<0x141>(v);     // row_half_mirror
    v += dpp_xchg<0x140>(v);     // row_mirror
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto l1 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)h1[0], (int)l1[0]) + __hiloint2double((int)h1[1], (int)l1[1]);
}
// p0 / p1: this row group's slice sums of row slots 0 / 1.  Returns the sum over the four row groups of slot (g & 1).
__device__ __forceinline__ double rowgroup_reduce_scatter(double p0, double p1)
{
    // v_permlane16_swap X, Y: rows 1, 3 of X <-> rows 0, 2 of Y.  X' + Y' = {p0(g0)+p0(g1), p1(g0)+p1(g1), p0(g2)+p0(g3), p1(g2)+p1(g3)}
    auto l1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(p0), (unsigned)__double2loint(p1), false, false);
    auto h1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(p0), (unsigned)__double2hiint(p1), false, false);
    const double s = __hiloint2double((int)h1[0], (int)l1[0]) + __hiloint2double((int)h1[1], (int)l1[1]);
    // v_permlane32_swap Z, W: rows 2, 3 of Z <-> rows 0, 1 of W
    const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
    auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}
template <int CG, int C> struct GroupFma {
    template <int NBC>
    static __device__ __forceinline__ void run(double (&acc)[2][2], const double (&B)[NBC], const double (&a)[2][CG])
    {
        if constexpr (C < CG) {
            BcFma<(C & 15)>::fmac(acc[0][C & 1], B[C >> 4], a[0][C]);
            BcFma<(C & 15)>::fmac(acc[1][C & 1], B[C >> 4], a[1][C]);
            GroupFma<CG, C + 1>::run(acc, B, a);
        }
    }
};

// CG columns of the lane's slice sit in registers, CGL more in LDS (one word per lane and coefficient, read back in the
// shadow of the exchange): at two waves per SIMD a wave has 256 VGPRs, a[2][44] is what fits next to the rest.
// The LDS-resident columns are streamed two at a time (four coefficients per lane), the next pair in flight while the
// current one is multiplied: all of them in registers at once is exactly the pressure that put them in LDS.
template <int CG, int CGL, int NWT, int J> struct LdsFma {
    template <int NBC>
    static __device__ __forceinline__ void run(double (&acc)[2][2], const double (&B)[NBC], const double *aL, double (&t)[2][2])
    {
        if constexpr (2 * J < CGL) {
            double u[2][2];
            if constexpr (2 * (J + 1) < CGL) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 2; ++r) u[c][r] = aL[(2 * (2 * (J + 1) + c) + r) * NWT];
            }
            constexpr int C0 = CG + 2 * J, C1 = C0 + 1;
            BcFma<(C0 & 15)>::fmac(acc[0][C0 & 1], B[C0 >> 4], t[0][0]);
            BcFma<(C0 & 15)>::fmac(acc[1][C0 & 1], B[C0 >> 4], t[0][1]);
            BcFma<(C1 & 15)>::fmac(acc[0][C1 & 1], B[C1 >> 4], t[1][0]);
            BcFma<(C1 & 15)>::fmac(acc[1][C1 & 1], B[C1 >> 4], t[1][1]);
            if constexpr (2 * (J + 1) < CGL) LdsFma<CG, CGL, NWT, J + 1>::run(acc, B, aL, u);
        }
    }
};

// sum over all 64 lanes (rows_sum joins row groups 0 + 1 and 2 + 3; one v_permlane32_swap pair joins the halves); every lane gets it
__device__ __forceinline__ double wave_allsum(double v)
{
    const double s = rows_sum(v);
    const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
    auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}
// the product alone: B holds the gathered vector (this row group's column slice), returns (M vec)[own row]
template <int NW, int CG, int CGL, int NBC>
__device__ __forceinline__ double rows_product(const double (&a)[2][CG], const double *aL, double (&B)[NBC])
{
    double tl[2][2];
    if (CGL > 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) tl[c][r] = aL[(2 * c + r) * NW * 64];
    }
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    dpp_hazard_fence(B);
    GroupFma<CG, 0>::run(acc, B, a);
    LdsFma<CG, CGL, NW * 64, 0>::run(acc, B, aL, tl);
    return rowgroup_reduce_scatter(acc[0][0] + acc[0][1], acc[1][0] + acc[1][1]);
}

template <int NW_, int CG, int CGL> struct RowsCfg {
    static constexpr int NW = NW_;
    static constexpr int NBC = (CG + CGL + 15) / 16;
    static constexpr int VS = 64 * NW;                    // vector words per buffer: 32 NW rows + one dummy word per lane
    static constexpr int ML = 208;                        // as many Lanczos steps as the largest p this kernel serves
    // LDS carve (doubles)
    static constexpr int OFF_V = 0;                       // exchanged vector [2][VS]
    static constexpr int OFF_XA = OFF_V + 2 * VS;         // per-lane scalar words [2][NW][64] (aux / alpha)
    static constexpr int OFF_XN = OFF_XA + 2 * NW * 64;   // per-lane scalar words [2][NW][64] (norms, sums)
    static constexpr int OFF_F = OFF_XN + 2 * NW * 64;    // int flags [2][NW][64]
    static constexpr int OFF_T = OFF_F + NW * 64;         // Lanczos alpha[ML], beta[ML]
    static constexpr int OFF_S = OFF_T + 2 * ML;          // Sturm scratch 2 (ML + 8)
    static constexpr int OFF_TH = OFF_S + 2 * (ML + 8);   // theta slot
    static constexpr int LCH = 1024;                      // lambdas staged in LDS at a time
    static constexpr int OFF_L = OFF_TH + 2 + 64 * NW;    // (+ one scratch word per lane before it)
    static constexpr int OFF_A = OFF_L + LCH;             // LDS-resident matrix columns [CGL][2][NW * 64]
    static constexpr int OFF_GX = OFF_A + 2 * CGL * NW * 64;   // group member lists (ints), group penalties only
    static constexpr int N_DBL = OFF_GX + (32 * NW + 8) / 2 + 4;
};

// this lane's group (group penalties in the row-split kernel): the members of the group of the lane's row are read from the
// exchanged u in LDS -- the first eight slots from registers (eight independent reads: one latency), longer groups walk the list
struct RowGrp {
    int gi, cnt, start, gmax;       // group index (-1: none), its size, its first entry in the member list, the wave's largest group
    bool gz;                        // unpenalised group (factor 1)
    double gw;                      // group weight
    int gm[8];                      // LDS slots of the first eight members (zero word beyond the group's end)
    const int *GX;                  // member lists in LDS
};

struct RowsLds {
    double *V, *XA, *XN;
    int *F;
};

// exchange + GEMV: every owner lane publishes `mine` (its row's entry), every lane picks up the entries of its
// row group's column slice and returns (M vec)[own row].  flag / aux as in gemv_sliced.
// NORM (Lanczos): aux carries this wave's share of || vec ||^2; the gathered entries are divided by the norm before
// the product, so the result is M (vec / || vec ||); aux returns the norm and scale its reciprocal.
template <int NW, int CG, int CGL, bool FLAGS, bool USE_AUX, bool NORM = false>
__device__ __forceinline__ double gemv_rows(const double (&a)[2][CG], const double *aL, double mine, int wslot,
                                            const int (&ecol)[(CG + CGL + 15) / 16],
                                            bool moving, bool &any, double &aux, const RowsLds &S, int w, int lane, int &buf,
                                            double *scale OEM_DIAG_ARGS)
{
    constexpr int NBC = (CG + CGL + 15) / 16, VS = RowsCfg<NW, CG, CGL>::VS;
    const int b = __builtin_amdgcn_readfirstlane(buf);              // provably uniform: addresses stay scalar + immediate
    const int any_mine = FLAGS ? ((__ballot(moving) != 0ull) ? 1 : 0) : 0;
    OEM_STAMP(0);                       // threshold, stop rule, loop control since the previous round
    static_assert(CGL % 2 == 0, "LDS-resident columns are streamed in pairs");
    double tl[2][2];                    // first pair of LDS-resident coefficients [column][row slot]: independent of the
    if (CGL > 0) {                      // vector, so it is fetched before the barrier and lands in its shadow
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) tl[c][r] = aL[(2 * c + r) * NW * 64];
    }
    S.V[b * VS + wslot] = mine;
    if (FLAGS) S.F[(b * NW + w) * 64 + lane] = any_mine;
    if (USE_AUX) S.XA[(b * NW + w) * 64 + lane] = aux;
    __syncthreads();
    OEM_STAMP(1);                       // stores + barrier
    int f = 0;
    if (FLAGS) f = S.F[(b * NW + (lane & (NW - 1))) * 64 + lane];
    double xa = 0.0;
    if (USE_AUX) xa = S.XA[(b * NW + (lane & (NW - 1))) * 64 + lane];
    double B[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) B[j] = S.V[b * VS + ecol[j]];
    __builtin_amdgcn_sched_barrier(0);
    OEM_STAMP(2);                       // reads (the stamp waits for them)
    if (NORM) {
        double nb, ib;
        sqrt_rsqrt(lanes_sum<NW>(xa), nb, ib);
#pragma unroll
        for (int j = 0; j < NBC; ++j) B[j] *= ib;
        aux = nb; *scale = ib;
    }
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    dpp_hazard_fence(B);                                        // VALU write of B -> DPP read: 2 wait states
    GroupFma<CG, 0>::run(acc, B, a);
    LdsFma<CG, CGL, NW * 64, 0>::run(acc, B, aL, tl);
    OEM_STAMP(3);                       // FMAs issued
    const double out = rowgroup_reduce_scatter(acc[0][0] + acc[0][1], acc[1][0] + acc[1][1]);
    if (USE_AUX && !NORM) aux = lanes_sum<NW>(xa);
    any = FLAGS ? __any(f != 0) : false;
    buf = b ^ 1;
#ifdef OEM_PATH_DIAG
    { unsigned sink; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sink) : "v"(__double2loint(out))); diag_acc[11] += sink & 1; }   // results are in before the stamp
#endif
    OEM_STAMP(4);                       // chain adds + reduce-scatter
    return out;
}

// sum over the waves of a per-wave value that every lane of the wave holds; own barrier, double-buffered by `par`
template <int NW>
__device__ __forceinline__ double waves_sum(double v, double *X, int &par, int w, int lane)
{
    const int b = __builtin_amdgcn_readfirstlane(par);
    X[(b * NW + w) * 64 + lane] = v;
    __syncthreads();
    const double x = X[(b * NW + (lane & (NW - 1))) * 64 + lane];
    par = b ^ 1;
    return lanes_sum<NW>(x);
}

// ACC (the accelerate option) is a template parameter: as a run-time test it costs two taken scalar branches per round
// on a chain where a fetch redirect is ~50 cycles.
template <int NW, int CG, int CGL, int KIND, bool ACC>
__device__ __forceinline__ void iterate_rows_t(const PathArgs &A, const PenK &K, const ThrK &c, const double (&a)[2][CG],
                                               const double *aL, double xy, double pf, int wslot,
                                               const int (&ecol)[(CG + CGL + 15) / 16],
                                               double &beta, double &ab, double &ak, int &it, int &conv, const RowsLds &S,
                                               int w, int lane, int &buf, const RowGrp &G OEM_DIAG_ARGS)
{
    const double tp = pf * K.L, tol = A.tol;
    const int maxit = A.maxit;
    constexpr int VS = RowsCfg<NW, CG, CGL>::VS;
    OEM_STAMP(8);                       // per-lambda work since the last round
    auto round = [&]() -> bool {
        asm volatile("; oem-round-begin %0" ::"n"(KIND * 2 + (ACC ? 1 : 0)));      // markers for oem_amd/build.py: audit_round_spills
        const double bold = beta;
        if constexpr (KIND == K_GRP) {
            // Group operators (ref src/oem_dense.h:193-315).  A group's members are rows of other lanes and waves, so u crosses
            // the waves once before the threshold (one more exchange than the element-wise operators: the same vector buffers,
            // which alternate u, beta, u, ...): every lane then sums its own group's squares in member order like the reference
            // and forms the factor itself; lanes of one group do so redundantly, which costs nothing.
            double u = ab + xy;
            if (K.kind == K_SGL) u = soft1(u, pf * K.L1, 1.0);     // sparse group lasso: the soft-thresholded u feeds the norms
            const int b = __builtin_amdgcn_readfirstlane(buf);
            S.V[b * VS + wslot] = u;
            __syncthreads();
            buf = b ^ 1;
            double x[8], s2 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = S.V[b * VS + G.gm[k]];
#pragma unroll
            for (int k = 0; k < 8; ++k) s2 += x[k] * x[k];
            for (int m = 8; m < G.gmax; ++m) {
                const double xm = S.V[b * VS + (m < G.cnt ? G.GX[G.start + m] : VS - 1)];
                s2 += xm * xm;
            }
            double f = 1.0;
            const double pen_g = K.L * G.gw;
            if (K.kind == K_GRP || K.kind == K_SGL) {
                // 1 - pen / ||u_g|| with the root and its reciprocal from v_rsq_f64 + Goldschmidt, the quotient refined like cdiv;
                // ||u_g|| = 0 => f = 0 (quirk Q6)
                double nrm, rn;
                sqrt_rsqrt_lane(s2, nrm, rn);
                const double t = 1.0 - cdiv(pen_g, nrm, rn);
                f = (s2 > 0.0 && 0.0 < t) ? t : 0.0;
            } else {
                const double nr = sqrt(s2);
                f = (K.kind == K_GRP_MCP) ? mcp_norm(nr, pen_g, K.D, K.gamma) : scad_norm(nr, pen_g, K.D, K.gamma);
            }
            f = G.gz ? 1.0 : f;
            f = G.gi >= 0 ? f : 0.0;
            beta = (f != 0.0) ? cdiv(u * f, c.D, c.rD) : 0.0;
        } else
        beta = threshold1<KIND>(ab + xy, tp, c);                    // padding lanes: zero matrix rows, xy = 0 => stays 0
        double aux = 0.0;
        if (ACC) {                                                 // ref src/oem_dense.h:633-651
            const double akp = ak;
            ak = 0.5 * (1.0 + sqrt(1.0 + 4.0 * ak * ak));
            const double ratio = (akp - 1.0) / ak;
            const double upd = beta, diff = upd - bold;
            beta = upd + ratio * diff;
            aux = rows_sum((beta - upd) * diff);                   // this wave's rows (padding lanes hold 0)
        }
        ++it;
        // stop rule (ref src/utils.cpp:537-549)
        const double cu = fabs(beta), q = fabs(bold);
        const bool cn = cu > 1e-13, qn = q > 1e-13;
        const bool moving = (cn != qn) || (cn && qn && fabs(beta - bold) > tol * q);
        bool any;
        ab = gemv_rows<NW, CG, CGL, true, ACC>(a, aL, beta, wslot, ecol, moving, any, aux, S, w, lane, buf, nullptr OEM_DIAG_PASS);
        if (ACC && aux > 0.0) ak = 1.0;
        conv = !any;
        asm volatile("; oem-round-end");
        return conv || it >= maxit;
    };
    // two rounds per trip: a taken branch (fetch redirect) costs ~80 cycles on this chain, a fall-through one nothing
    for (;;) {
        if (round()) break;
        if (round()) break;
    }
}
template <int NW, int CG, int CGL>
__global__ __launch_bounds__(NW * 64) void path_rows_kernel(PathArgs A_)
{
    const PathArgs A = path_instance(A_);
    typedef RowsCfg<NW, CG, CGL> C;
    constexpr int NBC = C::NBC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int p = A.p;
    const unsigned long long t_cyc0 = __builtin_amdgcn_s_memtime(), t_rt0 = __builtin_amdgcn_s_memrealtime();
    RowsLds S;
    S.V = lds + C::OFF_V; S.XA = lds + C::OFF_XA; S.XN = lds + C::OFF_XN;
    S.F = reinterpret_cast<int *>(lds + C::OFF_F);
    double *Tal = lds + C::OFF_T, *Tbe = Tal + C::ML, *LAM = lds + C::OFF_L;

    const int RWp = (p + NW - 1) / NW, CGp = (p + 3) / 4;           // rows per wave, columns per row group
    // this lane's row after the reduce-scatter: slot g & 1 of the wave's rows; groups 2, 3 replicate groups 0, 1
    const int rloc = 16 * (g & 1) + l16;
    const bool rowok = rloc < RWp && w * RWp + rloc < p;
    const int row = rowok ? w * RWp + rloc : 0;
    const bool owner = rowok && g < 2;
    const int wslot = owner ? row : 32 * NW + lane;                 // replicas and padding lanes store a word of their own
    int ecol[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) {
        const int loc = 16 * j + l16, col = g * CGp + loc;
        ecol[j] = (loc < CGp && col < p) ? col : 32 * NW + lane;    // own dummy word: finite, meets zero columns only
    }
    double a[2][CG];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lr = 16 * r + l16, grow = w * RWp + lr;
        const bool rok = lr < RWp && grow < p;
#pragma unroll
        for (int k = 0; k < CG; ++k) {
            const int col = g * CGp + k;
            a[r][k] = (rok && k < CGp && col < p) ? A.xx[(size_t)col * p + grow] : 0.0;
        }
    }
    double *aL = lds + C::OFF_A + tid;                               // [k][r] at (2 k + r) * NW * 64: conflict-free, lane-private
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lr = 16 * r + l16, grow = w * RWp + lr;
        const bool rok = lr < RWp && grow < p;
#pragma unroll
        for (int k = 0; k < CGL; ++k) {
            const int col = g * CGp + CG + k;
            aL[(2 * k + r) * NW * 64] = (rok && CG + k < CGp && col < p) ? A.xx[(size_t)col * p + grow] : 0.0;
        }
    }
    const double xy = rowok ? A.xy[row] : 0.0, pf = rowok ? A.pf[row] : 0.0;
    const double sinv = (rowok && A.sinv) ? A.sinv[row] : 1.0;
    // dummy words must hold finite numbers before anyone reads them
    for (int k = tid; k < 2 * C::VS; k += NW * 64) S.V[k] = 0.0;
    __syncthreads();
    int buf = 0, par = 0;
    bool any_unused;
    double aux_unused = 0.0;
    OEM_DIAG_DECL
    if (A.ngroups > 0) {                                             // group member lists into LDS (group operators only)
        int *gx = reinterpret_cast<int *>(lds + C::OFF_GX);
        const int nm = A.gstart[A.ngroups];
        for (int m = tid; m < nm && m < 32 * NW + 8; m += NW * 64) gx[m] = A.gidx[m];
        __syncthreads();
    }

    // ---- eigenvalue step: Lanczos with the vector spread over the waves (one entry per owner lane)
    int msteps = A.lanczos_steps < C::ML ? A.lanczos_steps : C::ML;
    double *theta_slot = lds + C::OFF_TH;
    double *tsink = lds + C::OFF_TH + 2 + tid;                       // a scratch word per lane: stores without exec branches
    auto top_ritz = [&](int m, double hint) {
        if (w == 0) {
            const double th = tridiag_max(Tal, Tbe, m, lane, lds + C::OFF_S, hint);
            if (lane == 0) theta_slot[0] = th;
        }
        __syncthreads();
        const double th = theta_slot[0];
        __syncthreads();
        return th;
    };
    // ONE exchange per step.  Every wave keeps the whole Lanczos vector (and its predecessor) a second time in the gathered
    // layout of the product (Bv / Bvp: this row group's column slice, a copy of every entry per wave).  What crosses waves is
    // the raw product w = A v with the waves' shares of alpha = v'w; each wave then orthogonalises ITS gathered copy with the
    // same two FMAs per entry as the owners use on their rows, and takes || w' || from that copy with an in-wave reduction --
    // identical operands in an identical order in every wave, so the replicas stay bit-identical and the norm needs no second
    // exchange.  (The two-exchange form spent ~2,100 cycles a step, an OEM round of the same product ~750.)
    double v, vp = 0.0, bprev = 0.0;
    double Bv[NBC], Bvp[NBC], Bw[NBC];
    int nst = 0;
    double theta = 0.0, theta_prev = -__builtin_inf(), mv_prev = __builtin_inf();
    bool have_theta = false;
    auto colmask = [&](int j) { return ecol[j] < 32 * NW ? 1.0 : 0.0; };   // dummy words carry replicas' values: not part of the vector
    {
        const unsigned h = (unsigned)row * 2654435761u + 12345u;     // deterministic non-structured start
        const double st = rowok ? ((double)(h >> 8) * (1.0 / 16777216.0) - 0.5) : 0.0;
        const int b = __builtin_amdgcn_readfirstlane(buf);
        S.V[b * C::VS + wslot] = st;
        __syncthreads();
        double n2 = 0.0;
#pragma unroll
        for (int j = 0; j < NBC; ++j) { Bw[j] = S.V[b * C::VS + ecol[j]] * colmask(j); n2 = fma(Bw[j], Bw[j], n2); }
        buf = b ^ 1;
        double nb, ib;
        sqrt_rsqrt(wave_allsum(n2), nb, ib);
        v = st * ib;
#pragma unroll
        for (int j = 0; j < NBC; ++j) { Bv[j] = Bw[j] * ib; Bvp[j] = 0.0; }
    }
    OEM_STAMP(5);                                                    // prologue: matrix and vectors into registers
    for (int j = 0; j < msteps; ++j) {
        const double wv = rows_product<NW, CG, CGL>(a, aL, Bv);
        double share = rows_sum(v * wv);                             // this wave's rows of alpha = v'Av (padding lanes hold 0)
        OEM_STAMP(3);
        const int b = __builtin_amdgcn_readfirstlane(buf);
        S.V[b * C::VS + wslot] = wv;
        S.XA[(b * NW + w) * 64 + lane] = share;
        __syncthreads();
        OEM_STAMP(1);
        const double xa = S.XA[(b * NW + (lane & (NW - 1))) * 64 + lane];
#pragma unroll
        for (int jj = 0; jj < NBC; ++jj) Bw[jj] = S.V[b * C::VS + ecol[jj]];
        buf = b ^ 1;
        const double al = lanes_sum<NW>(xa);
        OEM_STAMP(2);
        *(tid == 0 ? &Tal[j] : tsink) = al;                          // read by wave 0 only (top_ritz)
        const double wn = fma(-bprev, vp, fma(-al, v, wv));          // the owners' rows ...
        double n2 = 0.0;
#pragma unroll
        for (int jj = 0; jj < NBC; ++jj) {                           // ... and the same for the gathered copy
            Bw[jj] = fma(-bprev, Bvp[jj], fma(-al, Bv[jj], Bw[jj])) * colmask(jj);
            n2 = fma(Bw[jj], Bw[jj], n2);
        }
        double bb, ib;
        sqrt_rsqrt(wave_allsum(n2), bb, ib);
        OEM_STAMP(17);
        *(tid == 0 ? &Tbe[j] : tsink) = bb;
        nst = j + 1;
        if (__builtin_expect(__any(!(bb > 1e-13 * fabs(al))), 0)) break;   // invariant subspace reached: T is exact
        if (__builtin_expect(lanczos_check_due(nst) && nst < msteps, 0)) {
            OEM_STAMP(6);
            const double th = top_ritz(nst, theta_prev);
            OEM_STAMP(7);
            if (__any(lanczos_converged(th, theta_prev, mv_prev))) { theta = th; have_theta = true; break; }
        }
        vp = v; v = wn * ib; bprev = bb;
#pragma unroll
        for (int jj = 0; jj < NBC; ++jj) { Bvp[jj] = Bv[jj]; Bv[jj] = Bw[jj] * ib; }
        OEM_STAMP(19);
    }
    if (!have_theta) theta = top_ritz(nst, theta_prev);
    const double d = theta * 1.005;                                  // ref src/oem_dense.h:498
    if (tid == 0) { A.d_out[0] = d; A.d_out[1] = theta; A.d_out[4] = (double)nst; A.d_out[5] = (!have_theta && nst >= msteps && nst < p) ? 1.0 : 0.0; A.d_out[6] = 0.0; }
#ifdef OEM_PATH_DIAG
    OEM_STAMP(7);                                                    // the final top_ritz (slot 7: all top_ritz calls)
    unsigned long long lz[5];
    for (int k = 0; k < 5; ++k) { lz[k] = diag_acc[k]; diag_acc[12 + k] = lz[k]; diag_acc[k] = 0; }
    diag_acc[10] = lz[0] + lz[1] + lz[2] + lz[3] + lz[4] + diag_acc[6];   // Lanczos steps: gemv_rows + vector work
    diag_acc[9] = diag_acc[7]; diag_acc[6] = (unsigned long long)nst;
#endif

    // ---- A = d I - XX   (ref src/oem_dense.h:501-505)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lr = 16 * r + l16, grow = w * RWp + lr;
        const bool rok = lr < RWp && grow < p;
#pragma unroll
        for (int k = 0; k < CG; ++k) {
            const int col = g * CGp + k;
            a[r][k] = ((rok && k < CGp && col == grow) ? d : 0.0) - a[r][k];
        }
#pragma unroll
        for (int k = 0; k < CGL; ++k) {
            const int col = g * CGp + CG + k;
            aL[(2 * k + r) * NW * 64] = ((rok && CG + k < CGp && col == grow) ? d : 0.0) - aL[(2 * k + r) * NW * 64];
        }
    }

    // ---- lambda grid constants (ref src/oem_dense.cpp:175-192)
    const double scaley = A.yscale ? A.stats[1] : 1.0;
    const double yy = A.stats[2], nobs = A.stats[3];
    double lmax;
    {
        // max |xy| over all rows: wave maxima through per-lane words, then a max over the four words
        const double xl = (A.lmax_xy && rowok) ? A.lmax_xy[row] : xy;
        const double wm = wave_max(row >= A.lmax_from ? fabs(xl) : 0.0);      // padding lanes: xy = 0
        S.XN[(par * NW + w) * 64 + lane] = wm;
        __syncthreads();
        const double x = S.XN[(par * NW + (lane & (NW - 1))) * 64 + lane];
        par ^= 1;
        lmax = wave_max(x) * scaley;
    }
    const int nl = A.nl;
    const double llo = log(lmax), lhi = log(A.lambda_min_ratio * lmax);
    const double lstep = nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0;
    const bool lflip = fabs(lhi) < fabs(llo);

    // Stores without exec-masked branches (a taken branch is ~80 cycles on this chain and three of the four waves would
    // take every one of them): lanes that own nothing aim at a scratch word of their own in A.work.
    double *sinkd = A.work + tid;                                    // [0, 256)
    int *sinki = reinterpret_cast<int *>(A.work + 512) + tid;
    const bool t0 = tid == 0;
    // one lambda loop per (operator, accelerate) pair: the dispatch happens once per penalty, not once per lambda
    auto lambda_loop = [&](auto KIND_, auto ACC_, int pp, int pen) {
        constexpr int KIND = decltype(KIND_)::value;
        constexpr bool ACC = decltype(ACC_)::value;
        const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
        const bool isnet = pen_is_net(pen);
        double beta = 0.0, ab = 0.0, ak = 1.0;                       // cold start: A 0 = 0
        // this lane's group, looked up here and not in the kernel's prologue: the eight-wave forms have no registers to keep it
        // alive through the loops of the element-wise operators
        RowGrp G;
        G.gi = -1; G.cnt = 0; G.start = 0; G.gmax = 0; G.gz = false; G.gw = 0.0;
        G.GX = reinterpret_cast<const int *>(lds + C::OFF_GX);
#pragma unroll
        for (int k = 0; k < 8; ++k) G.gm[k] = C::VS - 1;             // a word that stays zero
        if constexpr (KIND == K_GRP) {
            if (A.ngroups > 0) {
                if (rowok) {
                    G.gi = A.gid[row];
                    if (G.gi >= 0) {
                        G.start = A.gstart[G.gi]; G.cnt = A.gstart[G.gi + 1] - G.start;
                        G.gz = A.gzero[G.gi] != 0; G.gw = A.gw[G.gi];
#pragma unroll
                        for (int k = 0; k < 8; ++k) G.gm[k] = k < G.cnt ? G.GX[G.start + k] : C::VS - 1;
                    }
                }
                G.gmax = (int)wave_max((double)G.cnt);
            }
        }
        // Per-lambda constants.  An FP64 division is a ~300-cycle dependent chain and a taken scalar branch ~80 cycles,
        // so: lambda / scale(y) goes through the corrected reciprocal; the reciprocals of the operator's denominators
        // are recomputed per lambda only if the penalty has a ridge part (otherwise the denominator is d); and the
        // lambda values themselves (user-supplied, or exp of Eigen's setLinSpaced grid incl. its "flip" form, *.net:
        // / alpha) are produced by a lane-parallel pre-pass into LDS, LCH at a time, so the serial loop only reads them.
        const double rscaley = 1.0 / scaley;
        const PenLin PL = pen_linear(pen, A.alpha, A.tau);
        const bool ridge = PL.cD != 0.0;
        ThrK c = uniform_thr(thr_consts<KIND>(pen_from_linear(PL, 0.0, d, A.gamma), d));
        for (int base = 0; base < nl; base += C::LCH) {
            const int cnt = nl - base < C::LCH ? nl - base : C::LCH;
            for (int k = tid; k < cnt; k += NW * 64) {
                const int i = base + k;
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
                LAM[k] = cdiv(lam, scaley, rscaley);                      // lambda / scale(y), ref src/oem_dense.cpp:241
                A.lambda_out[(size_t)pp * nl + i] = lam;
                A.loss[(size_t)pp * nl + i] = 1e99;                       // "not computed" (overwritten below if compute.loss)
            }
            __syncthreads();
            const int kend = nlam - base < cnt ? nlam - base : cnt;
            double lam_next = LAM[0];
            // output cursors: one pointer bump per lambda instead of a 64-bit multiply-add per store (lanes that own nothing
            // keep aiming at their sink word: stride 0)
            double *bptr = owner ? &A.beta[((size_t)pp * nl + base) * p + row] : sinkd;
            const size_t bstride = owner ? (size_t)p : 0;
            int *nptr = t0 ? &A.niter[(size_t)pp * nl + base] : sinki;
            const size_t nstride = t0 ? 1 : 0;
            for (int k = 0; k < kend; ++k) {
                const double il = lam_next;
                lam_next = LAM[k + 1 < cnt ? k + 1 : k];
                const PenK K = pen_from_linear(PL, il, d, A.gamma);
                if (__builtin_expect(ridge, 0)) c = uniform_thr(thr_consts<KIND>(K, d));
                int it = 0, conv = 0;
                iterate_rows_t<NW, CG, CGL, KIND, ACC>(A, K, c, a, aL, xy, pf, wslot, ecol, beta, ab, ak, it, conv, S, w, lane, buf, G OEM_DIAG_PASS);
                // (the loss is taken in the coordinates of the iteration, before any in-place rescale)
                if (__builtin_expect(A.compute_loss != 0, 0)) {
                    // sum (Y - X beta)^2 through the Gram identity (ref src/oem_dense.h:759-770):
                    // yy - 2 n beta'XY + n beta' XX beta, with XX beta = d beta - A beta
                    const double lossv = yy + nobs * waves_sum<NW>(rows_sum(beta * ((d * beta - ab) - 2.0 * xy)), S.XN, par, w, lane);
                    *(t0 ? &A.loss[(size_t)pp * nl + base + k] : sinkd + 256) = lossv;
                }
                // oemXTX::get_beta rescales the member in place (ref src/oem_xtx.h:576-581, quirk Q5)
                if (__builtin_expect(A.sinv != nullptr, 0)) beta *= sinv;
                *bptr = beta; bptr += bstride;
                *nptr = conv ? it : A.maxit + 1; nptr += nstride;         // ref src/oem_base.h:94-109
                if (__builtin_expect(A.sinv != nullptr, 0))
                    ab = gemv_rows<NW, CG, CGL, false, false>(a, aL, beta, wslot, ecol, false, any_unused, aux_unused, S, w, lane, buf, nullptr OEM_DIAG_PASS);
            }
            __syncthreads();                                             // LAM is rewritten by the next chunk
        }
    };
    for (int pp = A.pen_lo; pp < A.pen_hi; ++pp) {
        const int pen = A.penalty[pp];
        const int kind = pen_consts(pen, 1.0, d, A.alpha, A.gamma, A.tau).kind;
        using T = std::true_type; using F = std::false_type;
        if (A.accelerate) {
            switch (kind) {
            case K_SOFT: lambda_loop(std::integral_constant<int, K_SOFT>{}, T{}, pp, pen); break;
            case K_MCP: lambda_loop(std::integral_constant<int, K_MCP>{}, T{}, pp, pen); break;
            case K_SCAD: lambda_loop(std::integral_constant<int, K_SCAD>{}, T{}, pp, pen); break;
            case K_OLS: lambda_loop(std::integral_constant<int, K_OLS>{}, T{}, pp, pen); break;
            default: lambda_loop(std::integral_constant<int, K_GRP>{}, T{}, pp, pen); break;       // every group operator (K.kind selects inside)
            }
        } else {
            switch (kind) {
            case K_SOFT: lambda_loop(std::integral_constant<int, K_SOFT>{}, F{}, pp, pen); break;
            case K_MCP: lambda_loop(std::integral_constant<int, K_MCP>{}, F{}, pp, pen); break;
            case K_SCAD: lambda_loop(std::integral_constant<int, K_SCAD>{}, F{}, pp, pen); break;
            case K_OLS: lambda_loop(std::integral_constant<int, K_OLS>{}, F{}, pp, pen); break;
            default: lambda_loop(std::integral_constant<int, K_GRP>{}, F{}, pp, pen); break;
            }
        }
    }
#ifdef OEM_PATH_DIAG
    if (tid == 0) for (int k = 0; k < 24; ++k) g_diag[k] = diag_acc[k];
#endif
    if (tid == 0) {
        A.d_out[2] = (double)(__builtin_amdgcn_s_memtime() - t_cyc0);
        A.d_out[3] = (double)(__builtin_amdgcn_s_memrealtime() - t_rt0);
    }
    if (A.stats_out) for (int k = tid; k < A.stats_n; k += NW * 64) A.stats_out[k] = A.stats[k];      // results in host memory: stats beside them
}

// workgroup sets of one launch: instances x (penalties, when split)
static inline int path_grid_y(const PathArgs &a) { return (a.nbatch > 1 ? a.nbatch : 1) * (a.pen_split ? a.npen : 1); }

template <int NW, int CG, int CGL = 0> int launch_rows(hipStream_t s, const PathArgs &a)
{
    const size_t sh = (size_t)RowsCfg<NW, CG, CGL>::N_DBL * sizeof(double);
    if (sh > 64 * 1024 && lds_limit_once(reinterpret_cast<const void *>(&path_rows_kernel<NW, CG, CGL>), sh)) return OEMGPU_ERR_HIP;
    hipLaunchKernelGGL((path_rows_kernel<NW, CG, CGL>), dim3(1, path_grid_y(a)), dim3(NW * 64), sh, s, a);
    OEM_HIP(hipGetLastError());
    return 0;
}

template <int R, int NW, int CW, int G = 1> int launch_cfg(hipStream_t s, const PathArgs &a)
{
    typedef Cfg<R, NW, CW> C;
    const size_t sh = (size_t)C::N_DBL * sizeof(double);
    const int nbatch = path_grid_y(a);
    if (G > 1) {                                                                    // granule tags must start at 0
        if (nbatch > 1 && (size_t)a.bs_work * 8 != path_small_xchg_bytes()) { set_error("internal: batch work stride"); return OEMGPU_ERR_INTERNAL; }
        OEM_HIP(hipMemsetAsync(a.work, 0, path_small_xchg_bytes() * nbatch, s));
        // the poison slot of every instance (common.hpp: d_out[6]) starts at 0; only a timed-out workgroup writes it
        OEM_HIP(hipMemset2DAsync(a.d_out + 6, nbatch > 1 ? (size_t)a.bs_out : sizeof(double), 0, sizeof(double), (size_t)nbatch, s));
    }
    if (sh > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&path_small_kernel<R, NW, CW, G>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS %zu): %s", sh, hipGetErrorString(e)); return OEMGPU_ERR_HIP; }
    }
    hipLaunchKernelGGL((path_small_kernel<R, NW, CW, G>), dim3(G, nbatch), dim3(NW * 64), sh, s, a);
    OEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

#ifdef OEM_PATH_DIAG
extern "C" __attribute__((visibility("default"))) int oemgpu_diag_read(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag), sizeof(unsigned long long) * 24) == hipSuccess ? 0 : -1;
}
#endif

bool path_small_takes_rows(const PathArgs &a)
{
    return a.p <= 208;
}

int launch_path_small(hipStream_t s, const PathArgs &a)
{
    // p <= 208: the row-split form (beta all-gather, permlane reduce-scatter; group operators exchange u as well); four waves
    // up to p = 128, eight (two per SIMD, 256 VGPRs each: a[2][CG] must leave room) beyond.
    if (path_small_takes_rows(a)) {
        if (a.p <= 32) return launch_rows<4, 8>(s, a);
        if (a.p <= 64) return launch_rows<4, 16>(s, a);
        if (a.p <= 80) return launch_rows<4, 20>(s, a);
        if (a.p <= 104) return launch_rows<4, 26>(s, a);
        if (a.p <= 128) return launch_rows<4, 32>(s, a);
        if (a.p <= 160) return launch_rows<8, 40>(s, a);
        // a[2][44] = 176 of the 256 VGPRs a wave has at two waves per SIMD (46+ spills into the loop); the slice's
        // remaining columns live in LDS (config 2: p = 200 -> 6 of 50 columns)
        if (a.p <= 176) return launch_rows<8, 44>(s, a);
        return launch_rows<8, 44, 8>(s, a);
    }
    // 208 < p <= 288 that the cooperating engine (path_coop.hip) does not take -- scale.factor together with compute.loss, or the
    // engine switched off: four cooperating workgroups of eight waves, every wave with the whole vector.  (Until round 5 this kernel
    // also had single-workgroup forms for p <= 192 -- replicated for group penalties, "sliced" for element-wise ones -- that nothing
    // but two A/B switches could reach since the row-split kernel took every penalty in round 2: removed)
    if (a.p <= 256) return launch_cfg<4, 8, 8, 4>(s, a);          // four cooperating workgroups, 64 columns each
    if (a.p <= 288) return launch_cfg<5, 8, 10, 4>(s, a);         // ... 80 column slots each (CW must be even): big.oem's p = 256 + intercept lands here
    set_error("path_small: p = %d exceeds %d", a.p, SMALL_P_MAX);
    return OEMGPU_ERR_INTERNAL;
}

}  // namespace oemgpu
