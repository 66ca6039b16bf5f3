// path_large.hip -- eigenvalue + penalty x lambda path for p > 192 (XX no longer fits one CU's registers).
//
// Same arithmetic as path_small.hip (ref src/oem_base.h:90-110, src/oem_dense.h:485-653, src/utils.cpp:537-549),
// organised as a device-resident state machine driven by two kernels per OEM iteration:
//   gemv_sym_kernel     g = XX beta        every CU streams rows of XX (HBM/L2-bound at p = 4096: 134 MB per pass)
//   path_update_kernel  u = d beta - g + XY; beta = T(u); stop rule; lambda / penalty bookkeeping; outputs
// The host only enqueues batches of these pairs and polls one "done" word per batch: convergence decisions,
// warm starts and result packing never leave the device.  After "done" the remaining launches of a batch
// return immediately.
// The eigenvalue step is the same Lanczos recurrence as in the small engine with the GEMV as its own launch;
// the (tiny) tridiagonal eigenproblem is bisected on the host every 32 steps, which is also the convergence test.
#include "common.hpp"
#include "penalty_ops.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace oemgpu {

namespace {

struct LState {
    int pp, i, it, done;
    int pending_loss;        // index k*nl+i whose loss is due from the next g = XX beta, or -1
    int reset_next;          // the next update starts a penalty from beta = 0
    int finish_after_loss;   // all lambdas done; stop once the pending loss is written
    int pen;                 // penalty[pp] and lambda_out[pp * nl + i] of the position in use, written by whoever advances it: the update kernel
    double ak, d, theta, lmax;     // runs behind a product that has swept the caches, where every DEPENDENT load is an HBM round trip
    double lam;
};
static const int STATE_DBL = 16;
static const int MAXL = 512;       // Lanczos steps kept

// wave-uniform sum of v over the 64 lanes (same DPP scan as the small engine)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dppm(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int rlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int rhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ double wsum(double v)
{
    v += dppm<0x111, 0xf>(v); v += dppm<0x112, 0xf>(v); v += dppm<0x114, 0xf>(v); v += dppm<0x118, 0xf>(v);
    v += dppm<0x142, 0xa>(v); v += dppm<0x143, 0xc>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// deterministic block sum (1024 threads): per-wave DPP sums, then a fixed-order add of the 16 wave sums
__device__ __forceinline__ double block_sum(double v, double *sh)
{
    const int w = threadIdx.x >> 6;
    const double s = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = s;
    __syncthreads();
    double t = 0.0;
    const int nw = blockDim.x >> 6;
    for (int k = 0; k < nw; ++k) t += sh[k];
    return t;
}

// ------------------------------------------------------------------------------------------------
// g = M vec for symmetric column-major M (q x q): row r is the contiguous column r.  One wave per row, the vector
// held in registers (VPL doubles per lane), each row read with 1 KiB-coalesced dwordx4 loads.
// ------------------------------------------------------------------------------------------------
// FULL: q == 64 VPL exactly and 16-byte aligned rows, so there is no bounds logic at all (q = 512, 1024, 2048, 4096).
// Otherwise out-of-range columns read a clamped address and are multiplied by 0 (branch-free; hipcc turns
// per-element guards into exec-mask branches around every load, which serialises the stream).
template <int VPL, bool ALIGNED, bool FULL>
__global__ __launch_bounds__(256) void gemv_sym_kernel(const double *__restrict__ M, int q, const double *__restrict__ vec,
                                                        double *__restrict__ out, const int *__restrict__ done)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
    const int dn = done ? *done : 0;                      // consumed only after the loads below are in flight
    v2d v[VPL / 2];
    int off[VPL / 2];
    double msk0[FULL ? 1 : VPL / 2], msk1[FULL ? 1 : VPL / 2];
#pragma unroll
    for (int j = 0; j < VPL / 2; ++j) {
        const int c = 2 * lane + 128 * j;
        if (FULL) { off[j] = c; v[j] = *reinterpret_cast<const v2d *>(vec + c); }
        else {
            const int c0 = c < q ? c : q - 1, c1 = c + 1 < q ? c + 1 : q - 1;
            msk0[j] = c < q ? 1.0 : 0.0; msk1[j] = c + 1 < q ? 1.0 : 0.0;
            off[j] = (ALIGNED && c + 1 < q) ? c : c0;
            v[j].x = vec[c0] * msk0[j]; v[j].y = vec[c1] * msk1[j];
            if (!(ALIGNED && c + 1 < q)) { /* unaligned or edge pair: scalar loads below */ }
        }
    }
    if (dn) return;
    for (int r = wave; r < q; r += nwave) {
        const double *row = M + (size_t)r * q;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < VPL / 2; ++j) {
            if (FULL) {
                const v2d t = *reinterpret_cast<const v2d *>(row + off[j]);
                a0 = fma(t.x, v[j].x, a0);
                a1 = fma(t.y, v[j].y, a1);
            } else {
                const int c = 2 * lane + 128 * j;
                const int c0 = c < q ? c : q - 1, c1 = c + 1 < q ? c + 1 : q - 1;
                a0 = fma(row[c0], v[j].x, a0);           // v is already 0 where the column does not exist
                a1 = fma(row[c1], v[j].y, a1);
            }
        }
        const double s = wsum(a0 + a1);
        if (lane == 0) out[r] = s;
    }
}

// q > 4096, aligned rows: the matrix no longer sits in the Infinity Cache (q = 8,192: 537 MB), so this is a plain HBM stream and has to
// be issued like one.  The generic kernel of rounds 1-4 read 8 bytes per lane with ONE dependent accumulator and fetched the vector again
// for every row: 2.1 TB/s (profiles/r5_large_q_times.txt).  Here a workgroup takes GT = 16 consecutive rows (four per wave), stages the
// vector through LDS in panels of GP columns (once per row tile: 1 / 16 of the matrix bytes, from L2), and every lane keeps eight
// 16-byte loads in flight -- two column steps of four rows -- with an accumulator pair per row across the panels.  Columns beyond q read a
// clamped address against a zero of the staged vector (no exec-masked loads).
constexpr int GL_GP = 8192, GL_RW = 4;               // columns per LDS panel (64 KB: two workgroups per CU); rows per wave
__global__ __launch_bounds__(256) void gemv_large_kernel(const double *__restrict__ M, int q, const double *__restrict__ vec,
                                                          double *__restrict__ out, const int *__restrict__ done)
{
    extern __shared__ __attribute__((aligned(16))) double vsh[];
    if (done && *done) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ntile = (q + 4 * GL_RW - 1) / (4 * GL_RW);
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int r0 = tile * 4 * GL_RW + w * GL_RW;
        const double *row[GL_RW];
#pragma unroll
        for (int i = 0; i < GL_RW; ++i) row[i] = M + (size_t)(r0 + i < q ? r0 + i : q - 1) * q;      // (rows beyond q: the last row again, never stored)
        double a0[GL_RW], a1[GL_RW];
#pragma unroll
        for (int i = 0; i < GL_RW; ++i) { a0[i] = 0.0; a1[i] = 0.0; }
        for (int c0 = 0; c0 < q; c0 += GL_GP) {
            const int pc = q - c0 < GL_GP ? q - c0 : GL_GP, pcp = (pc + 255) & ~255;
            __syncthreads();                                     // (the previous panel has been read)
            for (int j = 2 * tid; j < pcp; j += 512) {
                v2d t = v2d{0.0, 0.0};
                if (j + 1 < pc) t = *reinterpret_cast<const v2d *>(vec + c0 + j);
                else if (j < pc) t.x = vec[c0 + j];
                *reinterpret_cast<v2d *>(vsh + j) = t;
            }
            __syncthreads();
            const int cmax = q - 2;                              // the last pair of a row (q is even: aligned rows)
            for (int j = 2 * lane; j < pcp; j += 256) {
                const int ja = c0 + j < cmax ? c0 + j : cmax, jb = c0 + j + 128 < cmax ? c0 + j + 128 : cmax;
                v2d ta[GL_RW], tb[GL_RW];
#pragma unroll
                for (int i = 0; i < GL_RW; ++i) { ta[i] = *reinterpret_cast<const v2d *>(row[i] + ja); tb[i] = *reinterpret_cast<const v2d *>(row[i] + jb); }
                const v2d va = *reinterpret_cast<const v2d *>(vsh + j), vb = *reinterpret_cast<const v2d *>(vsh + j + 128);
#pragma unroll
                for (int i = 0; i < GL_RW; ++i) {
                    a0[i] = fma(ta[i].x, va.x, a0[i]); a1[i] = fma(ta[i].y, va.y, a1[i]);
                    a0[i] = fma(tb[i].x, vb.x, a0[i]); a1[i] = fma(tb[i].y, vb.y, a1[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < GL_RW; ++i) {
            const double sres = wsum(a0[i] + a1[i]);
            if (lane == 0 && r0 + i < q) out[r0 + i] = sres;
        }
    }
}

// ... and the same stream where the rows are only 8-byte aligned (q odd: big.oem / xval.oem / a sparse x with an intercept have q = p + 1):
// 8-byte loads, four column steps of four rows in flight per lane instead of two
__global__ __launch_bounds__(256) void gemv_large_u_kernel(const double *__restrict__ M, int q, const double *__restrict__ vec,
                                                            double *__restrict__ out, const int *__restrict__ done)
{
    extern __shared__ __attribute__((aligned(16))) double vsh[];
    if (done && *done) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ntile = (q + 4 * GL_RW - 1) / (4 * GL_RW);
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int r0 = tile * 4 * GL_RW + w * GL_RW;
        const double *row[GL_RW];
#pragma unroll
        for (int i = 0; i < GL_RW; ++i) row[i] = M + (size_t)(r0 + i < q ? r0 + i : q - 1) * q;
        double a0[GL_RW], a1[GL_RW];
#pragma unroll
        for (int i = 0; i < GL_RW; ++i) { a0[i] = 0.0; a1[i] = 0.0; }
        for (int c0 = 0; c0 < q; c0 += GL_GP) {
            const int pc = q - c0 < GL_GP ? q - c0 : GL_GP, pcp = (pc + 255) & ~255;
            __syncthreads();
            for (int j = tid; j < pcp; j += 256) vsh[j] = j < pc ? vec[c0 + j] : 0.0;
            __syncthreads();
            const int cmax = q - 1;
            for (int j = lane; j < pcp; j += 256) {
                double t[4][GL_RW], vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jc = c0 + j + 64 * u < cmax ? c0 + j + 64 * u : cmax;
#pragma unroll
                    for (int i = 0; i < GL_RW; ++i) t[u][i] = row[i][jc];
                    vv[u] = vsh[j + 64 * u];
                }
#pragma unroll
                for (int i = 0; i < GL_RW; ++i) {
                    a0[i] = fma(t[0][i], vv[0], a0[i]); a1[i] = fma(t[1][i], vv[1], a1[i]);
                    a0[i] = fma(t[2][i], vv[2], a0[i]); a1[i] = fma(t[3][i], vv[3], a1[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < GL_RW; ++i) {
            const double sres = wsum(a0[i] + a1[i]);
            if (lane == 0 && r0 + i < q) out[r0 + i] = sres;
        }
    }
}

int launch_gemv(hipStream_t s, const double *M, int q, const double *vec, double *out, const int *done, int num_cu)
{
    int blocks = (q + 3) / 4;
    if (blocks > num_cu * 2) blocks = num_cu * 2;       // two 4-wave workgroups per CU, >= 2 rows per wave at q = 4096
    const bool al = (q % 2 == 0) && (((uintptr_t)M & 15) == 0);
    const bool vec_al = (((uintptr_t)vec & 15) == 0);
#define OEM_GEMV(V)                                                                                              \
    do {                                                                                                         \
        if (al && vec_al && q == 64 * V)                                                                         \
            hipLaunchKernelGGL((gemv_sym_kernel<V, true, true>), dim3(blocks), dim3(256), 0, s, M, q, vec, out, done); \
        else hipLaunchKernelGGL((gemv_sym_kernel<V, false, false>), dim3(blocks), dim3(256), 0, s, M, q, vec, out, done); \
    } while (0)
    if (q <= 512) OEM_GEMV(8);
    else if (q <= 1024) OEM_GEMV(16);
    else if (q <= 2048) OEM_GEMV(32);
    else if (q <= 4096) OEM_GEMV(64);
    else if (al && vec_al) {
        const size_t lds = sizeof(double) * GL_GP;
        if (lds_limit_once(reinterpret_cast<const void *>(&gemv_large_kernel), lds)) return OEMGPU_ERR_HIP;
        const int ntile = (q + 4 * GL_RW - 1) / (4 * GL_RW);
        hipLaunchKernelGGL(gemv_large_kernel, dim3(ntile < num_cu * 2 ? ntile : num_cu * 2), dim3(256), lds, s, M, q, vec, out, done);
    } else {
        const size_t lds = sizeof(double) * GL_GP;
        if (lds_limit_once(reinterpret_cast<const void *>(&gemv_large_u_kernel), lds)) return OEMGPU_ERR_HIP;
        const int ntile = (q + 4 * GL_RW - 1) / (4 * GL_RW);
        hipLaunchKernelGGL(gemv_large_u_kernel, dim3(ntile < num_cu * 2 ? ntile : num_cu * 2), dim3(256), lds, s, M, q, vec, out, done);
    }
#undef OEM_GEMV
    OEM_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ Lanczos
__global__ __launch_bounds__(1024) void lanczos_init_kernel(int q, double *__restrict__ v, double *__restrict__ vp)
{
    __shared__ double sh[16];
    double nn = 0.0;
    for (int j = threadIdx.x; j < q; j += blockDim.x) {
        const unsigned h = (unsigned)j * 2654435761u + 12345u;
        const double x = (double)(h >> 8) * (1.0 / 16777216.0) - 0.5;
        v[j] = x; vp[j] = 0.0;
        nn = fma(x, x, nn);
    }
    const double inv = 1.0 / sqrt(block_sum(nn, sh));
    for (int j = threadIdx.x; j < q; j += blockDim.x) v[j] *= inv;
}

// w = M v is in `w`.  alpha_j = v.w ; w -= alpha v + beta_{j-1} v_prev ; beta_j = |w| ; v_prev = v ; v = w / beta_j
__device__ __forceinline__ void lanczos_update(int q, int j, double *__restrict__ v, double *__restrict__ vp,
                                               double *__restrict__ w, double *__restrict__ T, double *sh)
{
    double *al = T, *be = T + MAXL;
    if (j > 0 && !(be[j - 1] > 1e-13 * fabs(al[j - 1]))) {          // invariant subspace already reached
        if (threadIdx.x == 0) { al[j] = al[j - 1]; be[j] = 0.0; }
        return;
    }
    const double bprev = j > 0 ? be[j - 1] : 0.0;
    double a = 0.0;
    for (int k = threadIdx.x; k < q; k += blockDim.x) a = fma(v[k], w[k], a);
    a = block_sum(a, sh);
    double bb = 0.0;
    for (int k = threadIdx.x; k < q; k += blockDim.x) {
        const double t = (w[k] - a * v[k]) - bprev * vp[k];
        w[k] = t;
        bb = fma(t, t, bb);
    }
    bb = sqrt(block_sum(bb, sh));
    if (threadIdx.x == 0) { al[j] = a; be[j] = bb; }
    if (bb > 1e-13 * fabs(a)) {
        const double ib = 1.0 / bb;
        for (int k = threadIdx.x; k < q; k += blockDim.x) { vp[k] = v[k]; v[k] = w[k] * ib; }
    }
}

// Fused Lanczos step (q <= 4096): one launch per step.  Launch j receives w = M v_j from its predecessor; EVERY workgroup
// forms alpha_j, w - alpha v - beta v_prev, beta_j and v_{j+1} itself (identical everywhere: two block reductions over q
// numbers), then streams its rows of M against v_{j+1}; workgroup 0 leaves v_{j+1}, v_j and T[j] behind.  Same arithmetic
// and the same breakdown rule as lanczos_update.
template <int VPL, bool FULL>
__global__ __launch_bounds__(256) void lanczos_fused_kernel(const double *__restrict__ M, int q, int j, double *__restrict__ Vc,
                                                             double *__restrict__ Vp, double *__restrict__ Wb,
                                                             double *__restrict__ T, int par)
{
    extern __shared__ __attribute__((aligned(16))) double dyn[];     // v_{j+1} [q]
    __shared__ double sh[16];
    const int tid = threadIdx.x, nt = 256, lane = tid & 63;
    const int wave = blockIdx.x * 4 + (tid >> 6), nwave = gridDim.x * 4;
    const bool b0 = blockIdx.x == 0;
    double *al = T, *be = T + MAXL;
    const double *__restrict__ v = Vc + (size_t)par * (q + 8), *__restrict__ vp = Vp + (size_t)par * (q + 8);
    const double *__restrict__ w = Wb + (size_t)par * (q + 8);
    double *__restrict__ vn = Vc + (size_t)(par ^ 1) * (q + 8), *__restrict__ vpn = Vp + (size_t)(par ^ 1) * (q + 8);
    double *__restrict__ wn = Wb + (size_t)(par ^ 1) * (q + 8);
    const bool broken = j > 0 && !(be[j - 1] > 1e-13 * fabs(al[j - 1]));     // invariant subspace already reached
    if (broken) {
        if (b0) {
            if (tid == 0) { al[j] = al[j - 1]; be[j] = 0.0; }
            for (int k = tid; k < q; k += nt) { vn[k] = v[k]; vpn[k] = vp[k]; wn[k] = w[k]; }
        }
        return;
    }
    const double bprev = j > 0 ? be[j - 1] : 0.0;
    double a = 0.0;
    for (int k = tid; k < q; k += nt) a = fma(v[k], w[k], a);
    a = block_sum(a, sh);
    double bb = 0.0;
    for (int k = tid; k < q; k += nt) {
        const double t = (w[k] - a * v[k]) - bprev * vp[k];
        dyn[k] = t;
        bb = fma(t, t, bb);
    }
    bb = sqrt(block_sum(bb, sh));
    if (b0 && tid == 0) { al[j] = a; be[j] = bb; }
    if (!(bb > 1e-13 * fabs(a))) {                                  // breakdown now: T is exact, the vectors stay
        if (b0) for (int k = tid; k < q; k += nt) { vn[k] = v[k]; vpn[k] = vp[k]; wn[k] = w[k]; }
        return;
    }
    const double ib = 1.0 / bb;
    for (int k = tid; k < q; k += nt) {
        const double x = dyn[k] * ib;
        dyn[k] = x;
        if (b0) { vn[k] = x; vpn[k] = v[k]; }
    }
    __syncthreads();
    v2d vv[VPL / 2];
    int c0[FULL ? 1 : VPL / 2], c1[FULL ? 1 : VPL / 2];
#pragma unroll
    for (int jj = 0; jj < VPL / 2; ++jj) {
        const int c = 2 * lane + 128 * jj;
        if (FULL) vv[jj] = *reinterpret_cast<const v2d *>(dyn + c);
        else {
            c0[jj] = c < q ? c : q - 1; c1[jj] = c + 1 < q ? c + 1 : q - 1;
            vv[jj].x = c < q ? dyn[c0[jj]] : 0.0; vv[jj].y = c + 1 < q ? dyn[c1[jj]] : 0.0;
        }
    }
    for (int r = wave; r < q; r += nwave) {
        const double *row = M + (size_t)r * q;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int jj = 0; jj < VPL / 2; ++jj) {
            if (FULL) {
                const v2d t = *reinterpret_cast<const v2d *>(row + 2 * lane + 128 * jj);
                a0 = fma(t.x, vv[jj].x, a0);
                a1 = fma(t.y, vv[jj].y, a1);
            } else {
                a0 = fma(row[c0[jj]], vv[jj].x, a0);
                a1 = fma(row[c1[jj]], vv[jj].y, a1);
            }
        }
        const double g = wsum(a0 + a1);
        if (lane == 0) wn[r] = g;
    }
}

// largest eigenvalue of the tridiagonal (host): bisection on the Sturm count; returns an upper bracket end
double tridiag_max_host(const double *al, const double *be, int m)
{
    if (m == 1) return al[0];
    double lo = -1e300, hi = -1e300;
    for (int j = 0; j < m; ++j) {
        const double bl = j > 0 ? std::fabs(be[j - 1]) : 0.0, br = j < m - 1 ? std::fabs(be[j]) : 0.0;
        if (al[j] > lo) lo = al[j];
        if (al[j] + bl + br > hi) hi = al[j] + bl + br;
    }
    for (int it = 0; it < 200 && hi - lo > 4e-16 * std::fabs(hi); ++it) {
        const double th = 0.5 * (lo + hi);
        if (th <= lo || th >= hi) break;
        double qv = al[0] - th;
        int neg = qv < 0.0;
        for (int k = 1; k < m; ++k) {
            if (qv == 0.0) qv = 1e-300;
            qv = (al[k] - th) - be[k - 1] * be[k - 1] / qv;
            neg += qv < 0.0;
        }
        if (neg < m) lo = th; else hi = th;
    }
    return hi;
}

// ------------------------------------------------------------------------------------------------ path
__global__ __launch_bounds__(1024) void path_init_kernel(PathArgs A, LState *st, double *__restrict__ beta, double d, double theta,
                                                          int lz_steps, int lz_capped)
{
    __shared__ double sh[16];
    double m = 0.0;
    const double *__restrict__ lx = A.lmax_xy ? A.lmax_xy : A.xy;       // (big.oem with p >= n: lambda_zero from the SCALED X'y, the iteration from the raw one)
    for (int j = threadIdx.x; j < A.p; j += blockDim.x) { beta[j] = 0.0; if (j >= A.lmax_from) m = fmax(m, fabs(lx[j])); }
    // block max through the sum helper's layout: per-wave max, then 16-way
    for (int s = 1; s < 64; s <<= 1) m = fmax(m, __shfl_xor(m, s, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double mm = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) mm = fmax(mm, sh[k]);
        st->pp = 0; st->i = 0; st->it = 0; st->done = (A.npen == 0);
        st->pending_loss = -1; st->reset_next = 1; st->finish_after_loss = 0; st->pen = 0; st->lam = 0.0;
        st->ak = 1.0; st->d = d; st->theta = theta;
        st->lmax = mm * (A.yscale ? A.stats[1] : 1.0);
        A.d_out[0] = d; A.d_out[1] = theta; A.d_out[2] = 0.0; A.d_out[3] = 0.0; A.d_out[4] = (double)lz_steps; A.d_out[5] = (double)lz_capped; A.d_out[6] = 0.0;
        sh[0] = st->lmax;
    }
    __syncthreads();
    // the whole lambda table once (ref src/oem_dense.cpp:175-227); the update kernel then only indexes it
    const double lmax = sh[0];
    const int nl = A.nl;
    const double llo = log(lmax), lhi = log(A.lambda_min_ratio * lmax);
    const double lstep = nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0;
    const bool lflip = fabs(lhi) < fabs(llo);
    for (int idx = threadIdx.x; idx < A.npen * nl; idx += blockDim.x) {
        const int pp = idx / nl, k = idx % nl;
        double l;
        if (A.user_lambda) l = A.lambda_user[idx];
        else {
            double lv;
            if (nl == 1) lv = lhi;
            else if (lflip) lv = (k == 0) ? llo : lhi - (double)(nl - 1 - k) * lstep;
            else lv = (k == nl - 1) ? lhi : llo + (double)k * lstep;
            l = exp(lv);
            if (pen_is_net(A.penalty[pp])) l = l / A.alpha;
        }
        A.lambda_out[idx] = l;
    }
    __syncthreads();
    if (threadIdx.x == 0 && A.npen > 0) { st->pen = A.penalty[0]; st->lam = A.lambda_out[0]; }
}

// R > 0: the thread's R coordinates of beta, g, XY, the penalty factors and the group ids are loaded into registers FIRST, before the
// state's dependent loads -- every load of the kernel that does not depend on another is then in flight together (q = 8,192 with 1,024
// groups: 14.4 us of dependent round trips, one per loop trip, for 256 KB of operands).  R = 0: the loops read memory as they go (any q).
// Same operations in the same order either way.
#ifdef OEM_PATH_DIAG
__device__ unsigned long long g_diag_update[8];       // cycles of thread 0 per phase of path_update<R > 0> (tools/attic/update_diag.py): [7] = calls
#define UPD_STAMP(k) do { if (R && tid == 0) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); g_diag_update[k] += t__ - upd_t; upd_t = t__; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define UPD_STAMP(k) do { } while (0)
#endif
// a workgroup barrier behind LDS traffic only (s_waitcnt lgkmcnt(0)), or __syncthreads() where global memory crosses threads too
__device__ __forceinline__ void upd_barrier(bool lds_only)
{
    if (lds_only) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
    else __syncthreads();
}
template <int R>
__device__ __forceinline__ void path_update(const PathArgs &A, LState *st, double *__restrict__ beta,
                                            const double *__restrict__ g, double *dyn, double *sh, bool lds_only, int *__restrict__ nz32)
{
    constexpr int RR = R ? R : 1;
    const int q = A.p, nl = A.nl, tid = threadIdx.x, nt = blockDim.x;
#ifdef OEM_PATH_DIAG
    unsigned long long upd_t = __builtin_amdgcn_s_memtime();
#endif
    double rb[RR], rg[RR], rxy[RR], rpf[RR];
    int rgid[RR];
    if (R) {
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const int j = tid + nt * r, jc = j < q ? j : q - 1;
            rb[r] = beta[jc]; rg[r] = g[jc]; rxy[r] = A.xy[jc]; rpf[r] = A.pf[jc];
            rgid[r] = A.ngroups > 0 ? A.gid[jc] : -1;
        }
    }
    int pgz = 1, pgs0 = 0, pgs1 = 0, pix[8];
    double pgw = 0.0;
    if (R && tid < A.ngroups) { pgz = A.gzero[tid]; pgs0 = A.gstart[tid]; pgs1 = A.gstart[tid + 1]; pgw = A.gw[tid]; }
    if (R) {                                                   // ... and the first eight member indices of that group: a second chain next to the state's
#pragma unroll
        for (int u = 0; u < 8; ++u) pix[u] = (tid < A.ngroups && pgs0 < pgs1) ? A.gidx[pgs0 + u < pgs1 ? pgs0 + u : pgs1 - 1] : 0;
    }
    // Everything the kernel needs from memory is asked for HERE, and consumed before the first branch: hipcc sinks a load below a branch it
    // does not have to precede, and behind a product that has swept the caches every such load is an HBM round trip of its own (stamped,
    // tools/attic/update_diag.py, q = 8,192 with groups: 37.3 k -> 34.5 k cycles per launch, element-wise 21.5 k -> 18.2 k).  What remains is
    // ONE CU's memory path: 330 KB of operands in and 64 KB out at the ~55 GB/s a single CU pulls are 6-7 us whatever the order -- the
    // stamps show the waves arriving at the first barrier 5 us apart.  Faster means more workgroups.
    const int done0 = st->done, pp = st->pp, i = st->i, pen = st->pen, pl = st->pending_loss, fin0 = st->finish_after_loss;
    const bool reset = st->reset_next != 0;
    int it = st->it;
    const double d = st->d, ak0 = st->ak, lam = st->lam;
    const double scaley = A.yscale ? A.stats[1] : 1.0;
    const double ilam = lam / scaley;
    double ru[RR];                                             // u = d beta - g + XY of the thread's coordinates (R > 0)
    if (R) {
#pragma unroll
        for (int r = 0; r < RR; ++r) ru[r] = (d * (reset ? 0.0 : rb[r]) - (reset ? 0.0 : rg[r])) + rxy[r];
    }
    if (done0) return;
    UPD_STAMP(0);                                              // the state and the operands are there (the one memory round trip)
    double *U = dyn, *F = dyn + q;
#define OEM_UPD_LOOP _Pragma("unroll RR") for (int r = 0, j = tid; R ? r < RR : j < q; ++r, j += nt) if (!R || j < q)
#define OEM_UPD(reg, mem) (R ? (reg)[R ? r : 0] : (mem))

    // ---- loss of the lambda that converged in the previous update: g is XX beta_final (Gram identity, see path_small)
    if (pl >= 0) {
        const double yy = A.stats[2], nobs = A.stats[3];
        double t = 0.0;
        OEM_UPD_LOOP t += OEM_UPD(rb, beta[j]) * (OEM_UPD(rg, g[j]) - 2.0 * OEM_UPD(rxy, A.xy[j]));
        t = block_sum(t, sh);
        if (tid == 0) A.loss[pl] = yy + nobs * t;
    }
    if (fin0) {
        __syncthreads();
        if (tid == 0) { st->pending_loss = -1; st->done = 1; }
        return;
    }
    const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
    const PenK K = pen_consts(pen, ilam, d, A.alpha, A.gamma, A.tau);
    const double rD = 1.0 / K.D, gammad = K.gamma * K.D, dmg = K.D - 1.0 / K.gamma, rdmg = 1.0 / dmg;
    const double gm1 = K.gamma - 1.0, dsc = gm1 * K.D - 1.0, rdsc = 1.0 / dsc;
    const bool grp = K.kind >= K_GRP;
    double ak = reset ? 1.0 : ak0;
    UPD_STAMP(1);                                              // the operator's constants

    // ---- u and (for group operators) the group factors
    if (grp) {
        OEM_UPD_LOOP {
            const double u = R ? ru[R ? r : 0] : (d * (reset ? 0.0 : beta[j]) - (reset ? 0.0 : g[j])) + A.xy[j];
            U[j] = (K.kind == K_SGL) ? soft1(u, OEM_UPD(rpf, A.pf[j]) * K.L1, 1.0) : u;
        }
        upd_barrier(lds_only);
        UPD_STAMP(2);                                          // u of every coordinate in LDS (the operands are there)
        for (int gi = tid; gi < A.ngroups; gi += nt) {
            double f = 1.0;
            const bool first = R && gi == tid;                   // (the first trip's group was fetched with the operands)
            if (!(first ? pgz : A.gzero[gi])) {
                double s = 0.0;
                // members in member order, their indices fetched eight at a time (one dependent load per member was a memory round trip each)
                const int m1 = first ? pgs1 : A.gstart[gi + 1];
                const int mbeg = first ? pgs0 : A.gstart[gi];
                for (int m = mbeg; m < m1; m += 8) {
                    int ix[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) ix[u] = (R && first && m == mbeg) ? pix[u] : A.gidx[m + u < m1 ? m + u : m1 - 1];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (m + u < m1) { const double x = U[ix[u]]; s += x * x; }
                }
                s = sqrt(s);
                const double pen_g = K.L * (first ? pgw : A.gw[gi]);
                if (K.kind == K_GRP || K.kind == K_SGL) { const double t = 1.0 - pen_g / s; f = (0.0 < t) ? t : 0.0; }
                else if (K.kind == K_GRP_MCP) f = mcp_norm(s, pen_g, K.D, K.gamma);
                else f = scad_norm(s, pen_g, K.D, K.gamma);
            }
            F[gi] = f;
        }
        upd_barrier(lds_only);
        UPD_STAMP(3);                                          // the group factors
    }
    // ---- beta = T(u), acceleration, stop rule.  Each thread owns its coordinates: it reads the old value, then writes.
    bool bad = false;
    double adp = 0.0;
    const double akn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * ak * ak)), ratio = (ak - 1.0) / akn;
    OEM_UPD_LOOP {
        const double bo = reset ? 0.0 : OEM_UPD(rb, beta[j]);
        double bn;
        if (grp) {
            const int gi = OEM_UPD(rgid, A.gid[j]);
            const double f = gi >= 0 ? F[gi] : 0.0;
            bn = (f != 0.0) ? cdiv(U[j] * f, K.D, rD) : 0.0;       // (a true FP64 division per coordinate is ~40 instructions on the one CU this kernel has)
        } else {
            const double u = R ? ru[R ? r : 0] : (d * bo - (reset ? 0.0 : g[j])) + A.xy[j];
            const double tp = OEM_UPD(rpf, A.pf[j]) * K.L;
            if (K.kind == K_SOFT) bn = cdiv(shrink(u, tp), K.D, rD);
            else if (K.kind == K_MCP) {
                const bool big = fabs(u) > gammad * tp;
                bn = cdiv(big ? u : shrink(u, tp), big ? K.D : dmg, big ? rD : rdmg);
            } else if (K.kind == K_SCAD) {
                const double au = fabs(u);
                const bool big = au > gammad * tp, mid = !big && au > (K.D + 1.0) * tp;
                const double num = big ? u : (mid ? shrink(gm1 * u, K.gamma * tp) : shrink(u, tp));
                bn = cdiv(num, mid ? dsc : K.D, mid ? rdsc : rD);
            } else bn = cdiv(u, d, 1.0 / d);
        }
        if (A.accelerate) {                                   // ref src/oem_dense.h:633-651
            const double upd = bn, diff = upd - bo;
            bn = upd + ratio * diff;
            adp += (bn - upd) * diff;
        }
        const double c = fabs(bn), qo = fabs(bo);
        const bool cn = c > 1e-13, qn = qo > 1e-13;
        bad |= (cn != qn);
        bad |= (cn && qn && fabs(bn - bo) > A.tol * qo);
        beta[j] = bn;
        if (R) rb[R ? r : 0] = bn;
        if (nz32) {                                            // which 32-coordinate pieces of the new iterate hold a non-zero (sympk_gemv_kernel skips blocks by them)
            const unsigned long long nzb = __ballot(bn != 0.0);
            if ((tid & 31) == 0) nz32[j >> 5] = ((nzb >> (tid & 32)) & 0xffffffffull) != 0ull ? 1 : 0;
        }
    }
    if (A.accelerate) {
        adp = block_sum(adp, sh);
        ak = (adp > 0.0) ? 1.0 : akn;
    }
    UPD_STAMP(4);                                              // the new coefficients, stored
    // (the vote crosses threads through LDS alone: no wait for the coefficient stores' acknowledgements in front of the barrier)
    int anybad;
    {
        int *vote = reinterpret_cast<int *>(sh + 16);
        const int wb = __ballot(bad) != 0ull ? 1 : 0;
        if ((tid & 63) == 0) vote[tid >> 6] = wb;
        upd_barrier(true);
        int o = 0;
        for (int k = 0; k < (nt >> 6); ++k) o |= vote[k];
        anybad = o;
    }
    ++it;
    UPD_STAMP(5);
    const bool conv = !anybad;
    if (conv || it >= A.maxit) {
        const size_t ki = (size_t)pp * nl + i;
        OEM_UPD_LOOP {
            double b = OEM_UPD(rb, beta[j]);
            if (A.sinv) { b *= A.sinv[j]; beta[j] = b; }            // quirk Q5: the member itself is rescaled
            A.beta[ki * q + j] = b;
        }
        if (tid == 0) {
            A.niter[ki] = conv ? it : A.maxit + 1;                  // ref src/oem_base.h:94-109
            if (!A.compute_loss) A.loss[ki] = 1e99;
            st->pending_loss = A.compute_loss ? (int)ki : -1;
            st->it = 0; st->ak = ak;
            if (i + 1 < nlam) { st->i = i + 1; st->reset_next = 0; st->lam = A.lambda_out[(size_t)pp * nl + i + 1]; }
            else if (pp + 1 < A.npen) { st->pp = pp + 1; st->i = 0; st->reset_next = 1; st->pen = A.penalty[pp + 1]; st->lam = A.lambda_out[(size_t)(pp + 1) * nl]; }
            else {
                st->reset_next = 0;
                if (A.compute_loss) st->finish_after_loss = 1; else st->done = 1;
            }
        }
    } else if (tid == 0) { st->it = it; st->ak = ak; st->reset_next = 0; st->pending_loss = -1; }
    UPD_STAMP(6);
#ifdef OEM_PATH_DIAG
    if (R && tid == 0) g_diag_update[7] += 1;
#endif
#undef OEM_UPD_LOOP
#undef OEM_UPD
}

// uf: where the group operand U[q] and the factors F[ngroups] live when q + ngroups doubles do not fit the LDS of one workgroup
// (q + ngroups beyond about 19,900: p >= n with a group penalty at p = 20,000 used to be refused) -- global memory, written and
// read by this one workgroup between its own barriers; null: LDS.
template <int R>
__global__ __launch_bounds__(1024) void path_update_kernel(PathArgs A, LState *st, double *__restrict__ beta,
                                                            const double *__restrict__ g, double *uf, int *nz32)
{
    extern __shared__ __attribute__((aligned(16))) double dyn[];     // U[q] (group operand), F[ngroups]
    __shared__ double sh[16 + 8];                                    // block sums | the stop rule's 16 wave votes (ints)
    path_update<R>(A, st, beta, g, uf ? uf : dyn, sh, uf == nullptr, nz32);
}

__global__ __launch_bounds__(1024) void lanczos_update_kernel(int q, int j, double *__restrict__ v, double *__restrict__ vp,
                                                               double *__restrict__ w, double *__restrict__ T)
{
    __shared__ double sh[16];
    lanczos_update(q, j, v, vp, w, T, sh);
}

// The same update with a thread's R entries of v, v_prev and w in registers: every load of the kernel is issued before the first
// reduction (q = 8,192: 12.9 -> one memory round trip and two block sums).  Same operations in the same order as lanczos_update.
template <int R>
__global__ __launch_bounds__(1024) void lanczos_update_reg_kernel(int q, int j, double *__restrict__ v, double *__restrict__ vp,
                                                                   const double *__restrict__ w, double *__restrict__ T)
{
    __shared__ double sh[16];
    double *al = T, *be = T + MAXL;
    const int tid = threadIdx.x;
    double xv[R], xp[R], xw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = tid + 1024 * r, kc = k < q ? k : q - 1;
        const double m = k < q ? 1.0 : 0.0;
        xv[r] = v[kc] * m; xp[r] = vp[kc] * m; xw[r] = w[kc] * m;
    }
    if (j > 0 && !(be[j - 1] > 1e-13 * fabs(al[j - 1]))) {          // invariant subspace already reached
        if (tid == 0) { al[j] = al[j - 1]; be[j] = 0.0; }
        return;
    }
    const double bprev = j > 0 ? be[j - 1] : 0.0;
    double a = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) a = fma(xv[r], xw[r], a);
    a = block_sum(a, sh);
    double bb = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double t = (xw[r] - a * xv[r]) - bprev * xp[r];
        xw[r] = t;
        bb = fma(t, t, bb);
    }
    bb = sqrt(block_sum(bb, sh));
    if (tid == 0) { al[j] = a; be[j] = bb; }
    if (bb > 1e-13 * fabs(a)) {
        const double ib = 1.0 / bb;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = tid + 1024 * r;
            if (k < q) { vp[k] = xv[r]; v[k] = xw[r] * ib; }
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Fused iteration (element-wise penalties, q = 512 / 1024 / 2048 / 4096, no accelerate / loss / scale.factor):
// ONE kernel per OEM iteration -- g = XX beta streamed exactly like gemv_sym_kernel, and the wave that finishes row r
// thresholds coordinate r on the spot (the operator is row-local).  What is not row-local is the stop rule and the
// lambda / penalty bookkeeping.  Instead of a second kernel (7.4 us + a launch boundary per iteration at p = 4096) or a
// last-arriver fence (measured: no faster), the bookkeeping is REPLICATED one launch later: every workgroup of launch
// k+1 ORs the per-workgroup "still moving" words that launch k left behind, and takes the same state transition
// (converged -> store this lambda's beta, move to the next lambda or penalty).  State, beta and flags are double-buffered
// by launch parity, which is a kernel argument of the graph node.  u = d beta - g + XY does not depend on lambda, so the
// launch that detects convergence of lambda_i already performs the first iteration of lambda_{i+1}.
// ------------------------------------------------------------------------------------------------
struct FState {
    int pp, i, it, done, fresh, pad0, pad1, pad2;
};
static const int FMAXB = 1024;          // flag words per parity

template <int VPL>
__global__ __launch_bounds__(256) void oem_fused_kernel(PathArgs A, FState *__restrict__ S, double *__restrict__ B,
                                                         int *__restrict__ flags, int *__restrict__ fdone, int par, double d)
{
    const int q = A.p, nl = A.nl, tid = threadIdx.x, lane = tid & 63;
    const int wave = blockIdx.x * 4 + (tid >> 6), nwave = gridDim.x * 4;
    const FState st = S[par];
    if (st.done) {                                                  // the launch after the last one: make both copies agree
        if (blockIdx.x == 0 && tid == 0) { S[par ^ 1].done = 1; *fdone = 1; }
        return;
    }
    const double *__restrict__ bin = B + (size_t)par * (q + 8);
    double *__restrict__ bout = B + (size_t)(par ^ 1) * (q + 8);
    v2d v[VPL / 2];
#pragma unroll
    for (int j = 0; j < VPL / 2; ++j) v[j] = *reinterpret_cast<const v2d *>(bin + 2 * lane + 128 * j);
    int f = 0;
    for (int t = tid; t < (int)gridDim.x; t += 256) f |= flags[par * FMAXB + t];
    const int any = __syncthreads_or(f);
    // ---- the replicated state transition
    int pp = st.pp, i = st.i, it = st.it;
    bool fresh = st.fresh != 0, finalize = false, done_now = false;
    size_t kfin = 0;
    int niter_fin = 0;
    if (!fresh) {
        const bool conv = !any;
        if (conv || it >= A.maxit) {
            finalize = true; kfin = (size_t)pp * nl + i; niter_fin = conv ? it : A.maxit + 1;     // ref src/oem_base.h:94-109
            const int nlam = (A.penalty[pp] == OEMGPU_OLS) ? 1 : nl;
            if (i + 1 < nlam) i = i + 1;
            else if (pp + 1 < A.npen) { pp = pp + 1; i = 0; fresh = true; }
            else done_now = true;
            it = 0;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        FState nx; nx.pp = pp; nx.i = i; nx.it = it + 1; nx.done = done_now ? 1 : 0; nx.fresh = 0; nx.pad0 = nx.pad1 = nx.pad2 = 0;
        S[par ^ 1] = nx;
        if (finalize) { A.niter[kfin] = niter_fin; A.loss[kfin] = 1e99; }
    }
    if (done_now) {                                                 // only the last lambda's coefficients are left to store
        for (int r = wave * 64 + lane; r < q; r += nwave * 64) A.beta[kfin * q + r] = bin[r];
        return;
    }
    // ---- constants of the (possibly new) lambda
    const int pen = A.penalty[pp];
    const double scaley = A.yscale ? A.stats[1] : 1.0;
    const PenK K = pen_consts(pen, A.lambda_out[(size_t)pp * nl + i] / scaley, d, A.alpha, A.gamma, A.tau);
    const double rD = 1.0 / K.D, gammad = K.gamma * K.D, dmg = K.D - 1.0 / K.gamma, rdmg = 1.0 / dmg;
    const double gm1 = K.gamma - 1.0, dsc = gm1 * K.D - 1.0, rdsc = 1.0 / dsc, rd = 1.0 / d, tol = A.tol;
    bool moving = false;
    for (int r = wave; r < q; r += nwave) {
        const double bo = bin[r], xyr = A.xy[r], tp = A.pf[r] * K.L;        // in flight while the row streams
        const double *row = A.xx + (size_t)r * q;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < VPL / 2; ++j) {
            const v2d t = *reinterpret_cast<const v2d *>(row + 2 * lane + 128 * j);
            a0 = fma(t.x, v[j].x, a0);
            a1 = fma(t.y, v[j].y, a1);
        }
        const double g = wsum(a0 + a1);
        if (finalize && lane == 0) A.beta[kfin * q + r] = bo;
        const double b0 = fresh ? 0.0 : bo;
        const double u = (d * b0 - (fresh ? 0.0 : g)) + xyr;
        double bn;
        if (K.kind == K_SOFT) bn = cdiv(shrink(u, tp), K.D, rD);
        else if (K.kind == K_MCP) {
            const bool big = fabs(u) > gammad * tp;
            bn = cdiv(big ? u : shrink(u, tp), big ? K.D : dmg, big ? rD : rdmg);
        } else if (K.kind == K_SCAD) {
            const double au = fabs(u);
            const bool big = au > gammad * tp, mid = !big && au > (K.D + 1.0) * tp;
            const double num = big ? u : (mid ? shrink(gm1 * u, K.gamma * tp) : shrink(u, tp));
            bn = cdiv(num, mid ? dsc : K.D, mid ? rdsc : rD);
        } else bn = cdiv(u, d, rd);
        const double c = fabs(bn), qo = fabs(b0);
        const bool cn = c > 1e-13, qn = qo > 1e-13;
        moving |= (cn != qn) || (cn && qn && fabs(bn - b0) > tol * qo);
        if (lane == 0) bout[r] = bn;
    }
    const int mv = __syncthreads_or(moving ? 1 : 0);
    if (tid == 0) flags[(par ^ 1) * FMAXB + blockIdx.x] = mv;
}

// ------------------------------------------------------------------------------------------------
// Symmetric-tile engine (q = 128 NBLK, element-wise penalties): XX is symmetric and this engine reads only its lower triangle
// -- 4 q^2 bytes per iteration instead of 8 q^2 (the row-streaming kernels above read all of it).
//
//   tiles      XX in 128 x 128 blocks; workgroup (I, J), I > J, owns block (I, J) (128 KB) and feeds BOTH products it holds:
//              g_I += T beta_J ("direct", rows of block I) and g_J += T' beta_I ("transposed", columns of block J).  The NBLK
//              diagonal blocks are read whole by workgroups of their own (direct product only: +3 % bytes at q = 4096) and are
//              launched last, so that the half-length workgroups fill the dispatch tail.
//   lanes      lane (a = lane / 8, b = lane % 8) of wave w loads rows 32 w + 8 g + a (g = 0..3), columns 16 h + 2 b, + 1
//              (h = 0..7): a wave load is 8 rows x 128 contiguous bytes (whole cache lines), the whole block is in flight in
//              registers (32 KB per wave) before the first FMA.  Direct product: a lane accumulates its row over the 8 column
//              groups (4 accumulators), the 8 lanes of a row meet in LDS.  Transposed product: a lane accumulates its two
//              columns over the 4 row groups (16 accumulators), the 8 row lanes x 4 waves meet in LDS.  Both sums run in a fixed
//              order -- no atomics, bitwise reproducible.
//   partials   block B of g receives exactly NBLK partial vectors: slot k < B from workgroup (B, k) (direct), slot B from the
//              diagonal workgroup, slot k > B from workgroup (k, B) (transposed): P[slot][q], 1 MB at q = 4096, double-buffered
//              by launch parity.
//   head       The reduction over the slots is the HEAD OF THE NEXT LAUNCH (no second kernel, no fence): workgroup (I, J) sums
//              the NBLK slots of its own 256 coordinates (blocks I and J) in slot order, thresholds them (the operator is
//              coordinate-local), and has beta_I, beta_J for its products; the stop rule and the lambda / penalty bookkeeping are
//              replicated one launch later exactly as in oem_fused_kernel (per-workgroup "still moving" words, state by launch
//              parity).  The head's loads are issued before the block's (vmcnt retires in order), so the threshold runs while
//              the block streams in.  The diagonal workgroups leave beta behind for the next launch (beta_prev of the stop rule).
//   bytes      per iteration: 4 q^2 + 4 q * 128 (blocks) + 8 q NBLK written, + 2 * 128 * NBLK * 8 B per workgroup re-read of the
//              partials by the heads (L2 / Infinity Cache hits: the buffer is 1 MB).
// ------------------------------------------------------------------------------------------------
static const int SYM_TB = 128;
#ifndef OEM_SYM_MINWG
#define OEM_SYM_MINWG 2          // workgroups per CU the symmetric-tile kernels are compiled for (3 spills: 168 VGPRs do not hold both halves of a block)
#endif
__host__ __device__ static inline int sym_nwg(int nblk) { return nblk * (nblk - 1) / 2 + nblk; }

// block (I, J) of workgroup b: the nblk (nblk - 1) / 2 off-diagonal ones first (b = I (I - 1) / 2 + J), then the diagonal ones
__device__ __forceinline__ void sym_block(int b, int nblk, int &I, int &J, bool &diag)
{
    const int noff = nblk * (nblk - 1) / 2;
    diag = b >= noff;
    if (diag) { I = J = b - noff; return; }
    int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)b)) * 0.5f);
    while (i * (i - 1) / 2 > b) --i;
    while ((i + 1) * i / 2 <= b) ++i;
    I = i; J = b - i * (i - 1) / 2;
}

// half a block per lane: the 4 row groups x 4 of the 8 column groups (16 loads of 16 B); a block is loaded as two halves so that
// three workgroups fit a CU (<= 168 VGPRs): all nblk (nblk + 1) / 2 workgroups of a launch are then resident at once (528 at
// q = 4096 against 768 slots -- with two per CU the last 16 waited for a second round)
struct SymHalf { v2d t[4][4]; };

__device__ __forceinline__ void sym_load(SymHalf &T, const double *__restrict__ xx, int q, int I, int J, int w, int a, int bb, int half)
{
    const double *tp = xx + (size_t)(SYM_TB * I + 32 * w + a) * q + SYM_TB * J + 64 * half + 2 * bb;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h) T.t[g][h] = *reinterpret_cast<const v2d *>(tp + (size_t)(8 * g) * q + 16 * h);
}

// the two products of a block with bsh[0..127] = vec_I, bsh[128..255] = vec_J, combined over lanes and waves in LDS, stored as
// partial vectors: direct -> Pout[J][128 I + r], transposed (off-diagonal blocks) -> Pout[I][128 J + c]
struct SymLds {
    double bsh[2 * SYM_TB];
    double dsh[SYM_TB][9];
    double tsh[4][8][SYM_TB + 16];
};
__device__ __forceinline__ void sym_half_products(const SymHalf &T, SymLds &L, double (&ds)[4], const double (&bi)[4], bool diag,
                                                  int w, int a, int bb, int half)
{
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const v2d bj = *reinterpret_cast<const v2d *>(&L.bsh[SYM_TB + 64 * half + 16 * h + 2 * bb]);
#pragma unroll
        for (int g = 0; g < 4; ++g) { ds[g] = fma(T.t[g][h].x, bj.x, ds[g]); ds[g] = fma(T.t[g][h].y, bj.y, ds[g]); }
        if (!diag) {
            v2d ts; ts.x = 0.0; ts.y = 0.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) { ts.x = fma(T.t[g][h].x, bi[g], ts.x); ts.y = fma(T.t[g][h].y, bi[g], ts.y); }
            *reinterpret_cast<v2d *>(&L.tsh[w][a][64 * half + 16 * h + 2 * bb]) = ts;
        }
    }
}
__device__ __forceinline__ void sym_combine(SymLds &L, const double (&ds)[4], double *__restrict__ Pout, int q, int I, int J, bool diag,
                                            int tid, int w, int a, int bb)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) L.dsh[32 * w + 8 * g + a][bb] = ds[g];
    __syncthreads();
    if (tid < SYM_TB) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += L.dsh[tid][k];
        Pout[(size_t)J * q + SYM_TB * I + tid] = s;
    } else if (!diag) {
        const int c = tid - SYM_TB;
        double s = 0.0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww)
#pragma unroll
            for (int aa = 0; aa < 8; ++aa) s += L.tsh[ww][aa][c];
        Pout[(size_t)I * q + SYM_TB * J + c] = s;
    }
}

// g = XX vec as NBLK partial vectors (Lanczos): P[slot][q]; symgemv_sum_kernel adds the slots in slot order
template <int NBLK>
__global__ __launch_bounds__(256, OEM_SYM_MINWG) void symgemv_kernel(const double *__restrict__ xx, const double *__restrict__ vec, double *__restrict__ P)
{
    constexpr int q = SYM_TB * NBLK;
    __shared__ __attribute__((aligned(16))) SymLds L;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, a = lane >> 3, bb = lane & 7;
    int I, J; bool diag;
    sym_block(blockIdx.x, NBLK, I, J, diag);
    const double mine = vec[tid < SYM_TB ? SYM_TB * I + tid : SYM_TB * J + (tid - SYM_TB)];
    SymHalf T0, T1;
    sym_load(T0, xx, q, I, J, w, a, bb, 0);
    sym_load(T1, xx, q, I, J, w, a, bb, 1);
    L.bsh[tid] = mine;
    __syncthreads();
    double bi[4], ds[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; ++g) bi[g] = L.bsh[32 * w + 8 * g + a];
    sym_half_products(T0, L, ds, bi, diag, w, a, bb, 0);
    sym_half_products(T1, L, ds, bi, diag, w, a, bb, 1);
    sym_combine(L, ds, P, q, I, J, diag, tid, w, a, bb);
}

template <int NBLK>
__global__ __launch_bounds__(128) void symgemv_sum_kernel(const double *__restrict__ P, double *__restrict__ out)
{
    constexpr int q = SYM_TB * NBLK;
    const int c = blockIdx.x * 128 + threadIdx.x;
    double ps[NBLK];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) ps[k] = P[(size_t)k * q + c];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NBLK; ++k) s += ps[k];
    out[c] = s;
}

// State of the symmetric-tile engine.  The head of a launch sits in front of its products, so what it needs must be ONE load away:
// the penalty code and lambda of the position (pp, i) AND of its successor travel inside the state (written by whoever advances
// the position, off everybody else's path), and so does scale(y) -- no load that depends on another load (the row-streaming
// kernel thresholds at the end of a row and can afford penalty[pp] -> lambda[pp][i] chains).
struct SState {
    int pp, i, it, done, fresh, pen, pen_next, pad;
    double lam, lam_next, scaley, pad2;
};
__device__ __forceinline__ void sym_successor(const PathArgs &A, int pp, int i, int pen, int &pen_next, double &lam_next)
{
    const int nlam = (pen == OEMGPU_OLS) ? 1 : A.nl;
    int sp = -1, si = 0;
    if (i + 1 < nlam) { sp = pp; si = i + 1; }
    else if (pp + 1 < A.npen) { sp = pp + 1; si = 0; }
    pen_next = sp >= 0 ? A.penalty[sp] : 0;
    lam_next = sp >= 0 ? A.lambda_out[(size_t)sp * A.nl + si] : 0.0;
}
__global__ void sym_init_kernel(SState *S, PathArgs A)
{
    SState z;
    z.pp = 0; z.i = 0; z.it = 0; z.done = (A.npen == 0) ? 1 : 0; z.fresh = 1; z.pad = 0; z.pad2 = 0.0;
    z.pen = A.npen > 0 ? A.penalty[0] : 0; z.lam = A.npen > 0 ? A.lambda_out[0] : 0.0;
    z.scaley = A.yscale ? A.stats[1] : 1.0;
    z.pen_next = 0; z.lam_next = 0.0;
    if (A.npen > 0) sym_successor(A, 0, 0, z.pen, z.pen_next, z.lam_next);
    S[0] = z; S[1] = z; S[1].done = 0;
}

template <int NBLK>
__global__ __launch_bounds__(256, OEM_SYM_MINWG) void oem_symfused_kernel(PathArgs A, SState *__restrict__ S, double *__restrict__ B,
                                                               double *__restrict__ P, int *__restrict__ flags, int *__restrict__ fdone,
                                                               int par, double d)
{
    constexpr int q = SYM_TB * NBLK;
    __shared__ __attribute__((aligned(16))) SymLds L;
    const int nl = A.nl, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, a = lane >> 3, bb = lane & 7;
    int I, J; bool diag;
    sym_block(blockIdx.x, NBLK, I, J, diag);
    // ---- the head's loads first (they retire first): state, flags, this thread's coordinate of beta_t, XY, the penalty factor
    // and the NBLK partial sums of g = XX beta_t
    const SState st = S[par];
    const int cm = tid < SYM_TB ? SYM_TB * I + tid : SYM_TB * J + (tid - SYM_TB);
    const double *__restrict__ bin = B + (size_t)par * (q + 8);
    double *__restrict__ bout = B + (size_t)(par ^ 1) * (q + 8);
    const double *__restrict__ Pin = P + (size_t)par * NBLK * q;
    int fl[FMAXB / 256];                                             // branch-free: a loop here would put a memory round trip in front of everything
#pragma unroll
    for (int k = 0; k < FMAXB / 256; ++k) { const int t = tid + 256 * k; fl[k] = flags[par * FMAXB + (t < (int)gridDim.x ? t : 0)]; }
    const double bo = bin[cm], xyc = A.xy[cm], pfc = A.pf[cm];
    double ps[NBLK];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) ps[k] = Pin[(size_t)k * q + cm];
    // ---- then the first half of the block: it streams in while the head computes
    SymHalf T0, T1;
    sym_load(T0, A.xx, q, I, J, w, a, bb, 0);
    if (st.done) {                                                  // the launch after the last one: make both copies agree
        if (blockIdx.x == 0 && tid == 0) { S[par ^ 1].done = 1; *fdone = 1; }
        return;
    }
    int f = 0;
#pragma unroll
    for (int k = 0; k < FMAXB / 256; ++k) f |= (tid + 256 * k < (int)gridDim.x) ? fl[k] : 0;
    const int any = __syncthreads_or(f);
    // ---- the replicated state transition (as oem_fused_kernel)
    int pp = st.pp, i = st.i, it = st.it, pen = st.pen;
    double lam = st.lam;
    bool fresh = st.fresh != 0, finalize = false, done_now = false, advanced = false;
    size_t kfin = 0;
    int niter_fin = 0;
    if (!fresh) {
        const bool conv = !any;
        if (conv || it >= A.maxit) {
            finalize = true; kfin = (size_t)pp * nl + i; niter_fin = conv ? it : A.maxit + 1;     // ref src/oem_base.h:94-109
            const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
            if (i + 1 < nlam) { i = i + 1; advanced = true; }
            else if (pp + 1 < A.npen) { pp = pp + 1; i = 0; fresh = true; advanced = true; }
            else done_now = true;
            if (advanced) { pen = st.pen_next; lam = st.lam_next; }
            it = 0;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        SState nx = st;
        nx.pp = pp; nx.i = i; nx.it = it + 1; nx.done = done_now ? 1 : 0; nx.fresh = 0; nx.pen = pen; nx.lam = lam;
        if (advanced) sym_successor(A, pp, i, pen, nx.pen_next, nx.lam_next);      // two dependent loads, on nobody else's path
        S[par ^ 1] = nx;
        if (finalize) { A.niter[kfin] = niter_fin; A.loss[kfin] = 1e99; }
    }
    if (finalize && diag && tid < SYM_TB) A.beta[kfin * q + cm] = bo;       // the diagonal workgroups own their block of beta
    if (done_now) return;
    // ---- beta_{t+1} of this thread's coordinate: g summed in slot order, u = d beta - g + XY, the operator, the stop rule
    const PenK K = pen_consts(pen, lam / st.scaley, d, A.alpha, A.gamma, A.tau);
    const double rD = 1.0 / K.D, gammad = K.gamma * K.D, dmg = K.D - 1.0 / K.gamma, rdmg = 1.0 / dmg;
    const double gm1 = K.gamma - 1.0, dsc = gm1 * K.D - 1.0, rdsc = 1.0 / dsc, rd = 1.0 / d;
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < NBLK; ++k) g += ps[k];
    const double b0 = fresh ? 0.0 : bo;
    const double u = (d * b0 - (fresh ? 0.0 : g)) + xyc;
    const double tp = pfc * K.L;
    double bn;
    if (K.kind == K_SOFT) bn = cdiv(shrink(u, tp), K.D, rD);
    else if (K.kind == K_MCP) {
        const bool big = fabs(u) > gammad * tp;
        bn = cdiv(big ? u : shrink(u, tp), big ? K.D : dmg, big ? rD : rdmg);
    } else if (K.kind == K_SCAD) {
        const double au = fabs(u);
        const bool big = au > gammad * tp, mid = !big && au > (K.D + 1.0) * tp;
        const double num = big ? u : (mid ? shrink(gm1 * u, K.gamma * tp) : shrink(u, tp));
        bn = cdiv(num, mid ? dsc : K.D, mid ? rdsc : rD);
    } else bn = cdiv(u, d, rd);
    const double c = fabs(bn), qo = fabs(b0);
    const bool cn = c > 1e-13, qn = qo > 1e-13;
    const bool moving = (cn != qn) || (cn && qn && fabs(bn - b0) > A.tol * qo);
    sym_load(T1, A.xx, q, I, J, w, a, bb, 1);                        // the partial sums' registers are free again: the second half
    if (diag && tid < SYM_TB) bout[cm] = bn;
    L.bsh[tid] = bn;
    const int mv = __syncthreads_or(moving ? 1 : 0);                 // (also the barrier behind bsh)
    if (tid == 0) flags[(par ^ 1) * FMAXB + blockIdx.x] = mv;
    double bi[4], ds[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; ++g) bi[g] = L.bsh[32 * w + 8 * g + a];
    sym_half_products(T0, L, ds, bi, diag, w, a, bb, 0);
    sym_half_products(T1, L, ds, bi, diag, w, a, bb, 1);
    sym_combine(L, ds, P + (size_t)(par ^ 1) * NBLK * q, q, I, J, diag, tid, w, a, bb);
}

// ------------------------------------------------------------------------------------------------
// q > 4096: the lower triangle of XX PACKED once per call, then streamed once per product (sympk_*).
// Beyond q = 4096 the matrix is neither register- nor Infinity-Cache-resident (q = 8,192: 537 MB): every product is an HBM stream, and
// the row-streaming kernels above read all 8 q^2 bytes of a symmetric matrix.  Here the 128 x 128 blocks of the lower triangle
// (diagonal blocks whole) are copied ONCE per call into a buffer of their own, block b = I (I + 1) / 2 + J behind block b - 1 and
// inside a block in the order the product kernel's lanes read it -- wave w, load k, lane l at ((32 w + k) 64 + l) 16 bytes -- so a
// product is ONE contiguous sweep of 4 q^2 + 512 q bytes: every wave instruction 1 KiB, every workgroup 128 KiB, no bounds logic
// (the ragged last block row / column is zero-filled by the pack, which also takes rows that are only 8-byte aligned: q odd).
// The arithmetic of a block is oem_symfused_kernel's (both products from one read, fixed summation order, partial vectors by slot);
// the slots are summed by a kernel of its own -- a workgroup's head would re-read as many bytes of partials as its block has at
// NBLK = 64 -- which is also the operator, the stop rule's "still moving" words and the replicated lambda / penalty bookkeeping
// (sympk_head_kernel: oem_symfused_kernel's head, 32 coordinates per workgroup -- element-wise operators, group operators whose groups
// have <= 96 members, Nesterov's step, compute.loss, scale.factor); groups of more than 96 members and the Lanczos steps get
// g = XX beta from sympk_sum_kernel and run path_update_kernel / lanczos_update_kernel unchanged.
// bytes per product: 4 q^2 + 512 q read + 8 q NBLK written and read again (q = 8,192: 272.6 + 4.2 + 4.2 MB).
// ------------------------------------------------------------------------------------------------
static const int SPK_TILE = SYM_TB * SYM_TB;             // doubles of a packed block (128 KiB)
__host__ __device__ static inline int spk_nblk(int q) { return (q + SYM_TB - 1) / SYM_TB; }
__host__ __device__ static inline size_t spk_ntile(int q) { const size_t nb = (size_t)spk_nblk(q); return nb * (nb + 1) / 2; }

// block (I, J), J <= I, of workgroup b: row-major over the lower triangle
__device__ __forceinline__ void spk_block(int b, int &I, int &J)
{
    int i = (int)((sqrtf(8.0f * (float)b + 1.0f) - 1.0f) * 0.5f);
    while (i * (i + 1) / 2 > b) --i;
    while ((i + 1) * (i + 2) / 2 <= b) ++i;
    I = i; J = b - i * (i + 1) / 2;
}

__global__ __launch_bounds__(256) void sympk_pack_kernel(const double *__restrict__ xx, int q, double *__restrict__ pk)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, a = lane >> 3, bb = lane & 7;
    int I, J;
    spk_block(blockIdx.x, I, J);
    double *tp = pk + (size_t)blockIdx.x * SPK_TILE + ((size_t)w * 32 * 64 + lane) * 2;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {                          // k = 16 half + 4 g + h: SymHalf::t[g][h] of that half
        const int half = k >> 4, g = (k >> 2) & 3, h = k & 3;
        const int r = SYM_TB * I + 32 * w + 8 * g + a, c = SYM_TB * J + 64 * half + 16 * h + 2 * bb;
        v2d t = v2d{0.0, 0.0};
        if (r < q) {
            const double *row = xx + (size_t)r * q;          // (symmetric: row r is the contiguous column r)
            if (c < q) t.x = row[c];
            if (c + 1 < q) t.y = row[c + 1];
        }
        *reinterpret_cast<v2d *>(tp + (size_t)k * 128) = t;
    }
}

__device__ __forceinline__ void spk_load(SymHalf &T, const double *__restrict__ tp, int half)
{
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h) T.t[g][h] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(tp + (size_t)(16 * half + 4 * g + h) * 128));
}

// one product with the packed triangle: P[slot][qpad] partial vectors (qpad = 128 NBLK), vec read as 0 beyond q
// nz32 (or null): one word per 32 coordinates of vec, 0 = all of them are zero (written by the head that produced vec).  A lasso iterate is
// sparse: a block whose row block AND column block of the vector are zero contributes nothing -- it is not read at all, its two partial
// vectors are written as zeros (the slot sum reads every slot).  Early in a path that is most blocks: the product then streams the block rows
// and columns of the active set only.  The flags are looked at BEFORE the block is asked for (one small dependent load per workgroup; the
// other workgroup of the CU keeps streaming meanwhile); a dense vector costs nothing measurable (oemgpu_selftest_sympk_gemv passes null).
__global__ __launch_bounds__(256, OEM_SYM_MINWG) void sympk_gemv_kernel(const double *__restrict__ pk, int q, int qpad, const double *__restrict__ vec,
                                                                         double *__restrict__ P, const int *__restrict__ done, const int *__restrict__ nz32)
{
    __shared__ __attribute__((aligned(16))) SymLds L;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, a = lane >> 3, bb = lane & 7;
    int I, J;
    spk_block(blockIdx.x, I, J);
    const bool diag = I == J;
    const int cm = tid < SYM_TB ? SYM_TB * I + tid : SYM_TB * J + (tid - SYM_TB);
    const int dn = done ? *done : 0;
    if (nz32) {
        const int fi = nz32[4 * I] | nz32[4 * I + 1] | nz32[4 * I + 2] | nz32[4 * I + 3], fj = nz32[4 * J] | nz32[4 * J + 1] | nz32[4 * J + 2] | nz32[4 * J + 3];
        if (!(fi | fj)) {                                     // (wave-uniform: every lane read the same eight words)
            if (dn) return;
            if (tid < SYM_TB) P[(size_t)J * qpad + SYM_TB * I + tid] = 0.0;
            else if (!diag) P[(size_t)I * qpad + SYM_TB * J + (tid - SYM_TB)] = 0.0;
            return;
        }
    }
    const double mine = cm < q ? vec[cm] : 0.0;
    const double *tp = pk + (size_t)blockIdx.x * SPK_TILE + ((size_t)w * 32 * 64 + lane) * 2;
    SymHalf T0, T1;
    spk_load(T0, tp, 0);                                  // (in flight before the done word is looked at: no round trip in front of the stream)
    if (dn) return;
    spk_load(T1, tp, 1);
    L.bsh[tid] = mine;
    __syncthreads();
    double bi[4], ds[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; ++g) bi[g] = L.bsh[32 * w + 8 * g + a];
    sym_half_products(T0, L, ds, bi, diag, w, a, bb, 0);
    sym_half_products(T1, L, ds, bi, diag, w, a, bb, 1);
    sym_combine(L, ds, P, qpad, I, J, diag, tid, w, a, bb);
}

// the NBLK slots of SPK_HC = 32 coordinates per workgroup: thread (ch = tid / 32, l = tid % 32) adds slots [ch NBLK / 8, (ch + 1) NBLK / 8) of
// coordinate l in slot order -- at q = 8,192 eight loads, all in flight at once: the kernel is one memory round trip -- and the eight
// chunk sums meet as ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)): one fixed order, the same in the sum kernel and in the head
constexpr int SPK_HC = 32;
// (the chunk's part alone: slots [ch NBLK / 8, (ch + 1) NBLK / 8) of coordinate c, in slot order)
__device__ __forceinline__ double spk_chunk_sum(const double *__restrict__ P, int nblk, int qpad, int c, int ch)
{
    const int k0 = (ch * nblk) >> 3, k1 = ((ch + 1) * nblk) >> 3;
    const double *pc = P + (size_t)k0 * qpad + c;
    double s = 0.0;
    int k = k0;
    for (; k + 8 <= k1; k += 8, pc += (size_t)8 * qpad) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = pc[(size_t)u * qpad];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = pc[(ptrdiff_t)(k + u < k1 ? u : k1 - 1 - k) * qpad];      // (never past the chunk's last slot)
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (k + u < k1) ? t[u] : 0.0;
    }
    return s;
}
__device__ __forceinline__ double spk_chunk_tree(const double (*sh)[SPK_HC], int l)
{
    return ((sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l])) + ((sh[4][l] + sh[5][l]) + (sh[6][l] + sh[7][l]));
}
__device__ __forceinline__ double spk_slot_sum(const double *__restrict__ P, int nblk, int qpad, int c, int ch, double (*sh)[SPK_HC], int l)
{
    const int k0 = (ch * nblk) >> 3, k1 = ((ch + 1) * nblk) >> 3;
    const double *pc = P + (size_t)k0 * qpad + c;
    double s = 0.0;
    int k = k0;
    for (; k + 8 <= k1; k += 8, pc += (size_t)8 * qpad) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = pc[(size_t)u * qpad];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    {                                                      // the rest of the chunk (< 8 slots), branch-free loads of a clamped slot
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = pc[(ptrdiff_t)(k + u < k1 ? u : k1 - 1 - k) * qpad];      // (never past the chunk's last slot)
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (k + u < k1) ? t[u] : 0.0;
    }
    sh[ch][l] = s;
    __syncthreads();
    return ((sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l])) + ((sh[4][l] + sh[5][l]) + (sh[6][l] + sh[7][l]));
}

__global__ __launch_bounds__(256) void sympk_sum_kernel(const double *__restrict__ P, int nblk, int q, int qpad, double *__restrict__ out,
                                                         const int *__restrict__ done)
{
    __shared__ double sh[8][SPK_HC];
    const int dn = done ? *done : 0;                      // (consumed behind the slot loads)
    const int l = threadIdx.x & (SPK_HC - 1), ch = threadIdx.x / SPK_HC, c = blockIdx.x * SPK_HC + l;
    const double s = spk_slot_sum(P, nblk, qpad, c, ch, sh, l);
    if (ch == 0 && c < q && !dn) out[c] = s;
}

// Element-wise penalties: the slot sum IS the head of oem_symfused_kernel -- state, "still moving" words of the previous launch, the
// replicated transition, the operator on the workgroup's 32 coordinates, beta_{t+1} into B[par ^ 1] (which the product kernel of
// this iteration then reads), the workgroup's "still moving" word.  One (head, product) pair of launches per iteration.
// GRP (PathArgs::grp_head = HB: every group a run of <= 32 HB neighbouring coordinates, no Nesterov step, no loss, no scale.factor): the group
// operators as well (ref src/oem_dense.h:193-315).  A group of one of the workgroup's coordinates lies inside the window of the 32 HB coordinates
// before them, themselves and the 32 HB behind them: the workgroup forms u of the whole window (2 HB + 1 slot sums instead of one -- the partial
// vectors are 4 MB against the product's 270 -- thread row ch takes coordinate l of one block of the window),
// puts it through LDS and every own coordinate sums the squares of ITS group in member order.  A group that straddles two workgroups is summed by
// both, from the same numbers in the same order.  Everything else -- state, flags, the blocks of zeros -- is the element-wise head's.
// HB = 1, 2, 3 blocks of 32 coordinates on either side of the own one: groups of <= 32 HB members (HB = 0: the element-wise head).
// GEN (gen != null): the two options that need a sum over ALL coordinates, with the sum taken one launch later like the stop rule --
//   Nesterov's step (ref src/oem_dense.h:529, 633-651): every workgroup leaves its part of the restart test sum_j (beta_j - T(u)_j)(T(u)_j - beta_prev_j)
//       in gen[parity][workgroup]; the next head adds the parts in workgroup order (every workgroup the same sum) and so knows the sequence's
//       member of ITS iteration -- exactly when it needs it; the candidate successor travels in gen's state words;
//   compute.loss (ref :759-770, Gram identity): when a lambda ends, g = XX beta of the finished iterate is in hand -- the workgroups' parts of
//       beta'(g - 2 XY) go to gen, the NEXT head's workgroup 0 adds them and writes the loss (the launch behind the last one as well: the done word
//       is then raised by that launch).
// gen: [2][FMAXB] restart parts | [2][FMAXB] loss parts | [2][4] state words {akn, 1 + index of the pending loss or 0}
template <int HB, bool GEN>
__global__ __launch_bounds__(256) void sympk_head_kernel(PathArgs A, SState *__restrict__ S, double *__restrict__ B, const double *__restrict__ P,
                                                          int *__restrict__ flags, int *__restrict__ fdone, int par, double d, int nblk, int qpad, int *__restrict__ nz32,
                                                          double *__restrict__ gen)
{
    constexpr bool GRP = HB > 0;
    constexpr int NWB = 2 * HB + 1;                                 // blocks of the window
    static_assert(NWB <= 8, "one thread row (ch) per block of the window");
    __shared__ double sh[NWB][8][SPK_HC];
    __shared__ double Ush[NWB * SPK_HC];
    const int q = A.p, nl = A.nl, tid = threadIdx.x, l = tid & (SPK_HC - 1), ch = tid / SPK_HC, c0 = blockIdx.x * SPK_HC, cm = c0 + l;
    const bool own = ch == 0 && cm < q;
    // the window coordinate of this thread (GRP; ch >= NWB: none) -- or its own one.  Thread row 0 takes the own block (HB of the window),
    // rows 1 .. HB the blocks below it, rows HB + 1 .. 2 HB those above
    const int wslot = ch == 0 ? HB : (ch <= HB ? ch - 1 : ch);
    const int cw = GRP ? c0 + SPK_HC * (wslot - HB) + l : cm;
    const bool wok = GRP ? (ch < NWB && cw >= 0 && cw < q) : own;
    const SState st = S[par];
    const double *__restrict__ bin = B + (size_t)par * qpad;
    double *__restrict__ bout = B + (size_t)(par ^ 1) * qpad;
    int fl[FMAXB / 256];
#pragma unroll
    for (int k = 0; k < FMAXB / 256; ++k) { const int t = tid + 256 * k; fl[k] = flags[par * FMAXB + (t < (int)gridDim.x ? t : 0)]; }
    const double bo = wok ? bin[cw] : 0.0, xyc = wok ? A.xy[cw] : 0.0, pfc = wok ? A.pf[cw] : 0.0;
    const double sinvc = (A.sinv && own) ? A.sinv[cm] : 1.0;      // oem.xtx's scale.factor: the iterate is rescaled IN PLACE when a lambda ends (ref src/oem_xtx.h:576-581, quirk Q5)
    // GEN: the parts of the previous launch (restart test, loss) and its state words, asked for with everything else
    __shared__ double gsh[8];
    double ap[FMAXB / 256], lp[FMAXB / 256], akn_prev = 1.0;
    int loss_k = -1;
    if constexpr (GEN) {
#pragma unroll
        for (int k = 0; k < FMAXB / 256; ++k) {
            const int t = tid + 256 * k, tc = t < (int)gridDim.x ? t : 0;
            ap[k] = gen[(size_t)par * FMAXB + tc]; lp[k] = gen[(size_t)(2 + par) * FMAXB + tc];
        }
        akn_prev = gen[4 * FMAXB + 4 * par]; loss_k = (int)gen[4 * FMAXB + 4 * par + 1] - 1;      // (the word holds index + 1: the area starts zeroed = nothing pending)
    }
    int gs = 1, ge = 0;
    bool gz = false;
    double gwc = 0.0;
    if (GRP && own) {
        const int e = A.grun[2 * cm + 1];
        gs = A.grun[2 * cm]; ge = e & 0x3fffffff; gz = ((e >> 30) & 1) != 0; gwc = A.gwc[cm];
    }
    double g;                                                       // (fresh: P is not there yet and g is not used; the loads are issued before the state is looked at)
    if constexpr (!GRP) g = spk_slot_sum(P, nblk, qpad, cm, ch, sh[0], l);
    else {
#pragma unroll
        for (int b = 0; b < NWB; ++b) {
            int cb = c0 + SPK_HC * (b - HB) + l;
            cb = cb < 0 ? 0 : (cb < qpad ? cb : qpad - 1);          // (beyond the vector: read somewhere, never used)
            sh[b][ch][l] = spk_chunk_sum(P, nblk, qpad, cb, ch);
        }
        __syncthreads();
        g = spk_chunk_tree(sh[wslot < NWB ? wslot : 0], l);
    }
    // GEN: the loss of the lambda that ended in the previous launch (workgroup 0: the parts in workgroup order)
    double atot = 0.0;
    if constexpr (GEN) {
        if (A.compute_loss && loss_k >= 0 && blockIdx.x == 0) {     // (uniform in the workgroup)
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < FMAXB / 256; ++k) t += (tid + 256 * k < (int)gridDim.x) ? lp[k] : 0.0;
            t = block_sum(t, gsh);
            if (tid == 0) A.loss[loss_k] = A.stats[2] + A.stats[3] * t;
        }
    }
    if (st.done) {                                                  // the launch after the last one: make both copies agree
        if (blockIdx.x == 0 && tid == 0) { S[par ^ 1].done = 1; *fdone = 1; if (GEN) gen[4 * FMAXB + 4 * (par ^ 1) + 1] = 0.0; }
        return;
    }
    if constexpr (GEN) {
        if (A.accelerate) {                                         // the restart test of the previous iteration: every workgroup the same sum
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < FMAXB / 256; ++k) t += (tid + 256 * k < (int)gridDim.x) ? ap[k] : 0.0;
            atot = block_sum(t, gsh);
        }
    }
    int f = 0;
#pragma unroll
    for (int k = 0; k < FMAXB / 256; ++k) f |= (tid + 256 * k < (int)gridDim.x) ? fl[k] : 0;
    const int any = __syncthreads_or(f);
    int pp = st.pp, i = st.i, it = st.it, pen = st.pen;
    double lam = st.lam;
    bool fresh = st.fresh != 0, finalize = false, done_now = false, advanced = false;
    size_t kfin = 0;
    int niter_fin = 0;
    if (!fresh) {
        const bool conv = !any;
        if (conv || it >= A.maxit) {
            finalize = true; kfin = (size_t)pp * nl + i; niter_fin = conv ? it : A.maxit + 1;     // ref src/oem_base.h:94-109
            const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
            if (i + 1 < nlam) { i = i + 1; advanced = true; }
            else if (pp + 1 < A.npen) { pp = pp + 1; i = 0; fresh = true; advanced = true; }
            else done_now = true;
            if (advanced) { pen = st.pen_next; lam = st.lam_next; }
            it = 0;
        }
    }
    // scale.factor: the next lambda starts from beta / s, whose product is not the one in hand -- this launch publishes the rescaled iterate
    // (no operator, not an iteration, marked "moving" so that nobody takes the round for a converged one) and the path goes on from there
    const bool resc = A.sinv != nullptr && finalize && !done_now && !fresh;
    if (blockIdx.x == 0 && tid == 0) {
        SState nx = st;
        nx.pp = pp; nx.i = i; nx.it = resc ? 0 : it + 1; nx.done = done_now ? 1 : 0; nx.fresh = 0; nx.pen = pen; nx.lam = lam;
        if (advanced) sym_successor(A, pp, i, pen, nx.pen_next, nx.lam_next);
        S[par ^ 1] = nx;
        if (finalize) { A.niter[kfin] = niter_fin; if (!(GEN && A.compute_loss)) A.loss[kfin] = 1e99; }
    }
    if (finalize && own) A.beta[kfin * q + cm] = bo * sinvc;
    const bool want_loss = GEN && A.compute_loss != 0;
    if constexpr (GEN) {
        if (want_loss && finalize) {                                // (uniform) this workgroup's part of beta'(XX beta - 2 XY) of the finished iterate: g IS XX beta
            const double lt = block_sum(own ? bo * (g - 2.0 * xyc) : 0.0, gsh);
            if (tid == 0) gen[(size_t)(2 + (par ^ 1)) * FMAXB + blockIdx.x] = lt;
        }
        if (blockIdx.x == 0 && tid == 0) gen[4 * FMAXB + 4 * (par ^ 1) + 1] = (want_loss && finalize) ? (double)(kfin + 1) : 0.0;
    }
    if (done_now) {
        // (the products behind this launch return at once; a pending loss is written by the NEXT head, which then raises the word)
        if (blockIdx.x == 0 && tid == 0 && !want_loss) *fdone = 1;
        return;
    }
    const PenK K = pen_consts(pen, lam / st.scaley, d, A.alpha, A.gamma, A.tau);
    const double rD = 1.0 / K.D, gammad = K.gamma * K.D, dmg = K.D - 1.0 / K.gamma, rdmg = 1.0 / dmg;
    const double gm1 = K.gamma - 1.0, dsc = gm1 * K.D - 1.0, rdsc = 1.0 / dsc, rd = 1.0 / d;
    const double b0 = fresh ? 0.0 : bo;
    const double u = (d * b0 - (fresh ? 0.0 : g)) + xyc;
    const double tp = pfc * K.L;
    double bn;
    if (GRP && K.kind >= K_GRP) {                                   // (wave-uniform: the penalty is the state's)
        const double uu = (K.kind == K_SGL) ? soft1(u, pfc * K.L1, 1.0) : u;
        if (ch < NWB) Ush[SPK_HC * wslot + l] = wok ? uu : 0.0;
        __syncthreads();
        double s2 = 0.0;
        for (int m = gs; m < ge; ++m) { const double x = Ush[m - (c0 - SPK_HC * HB)]; s2 += x * x; }      // member order (a run: ascending coordinates)
        double fg = 1.0;
        if (ge < gs) fg = 0.0;                                      // (in no group: its coefficient stays 0, as path_update has it)
        else if (!gz) {
            const double sn = sqrt(s2), pen_g = K.L * gwc;
            if (K.kind == K_GRP || K.kind == K_SGL) { const double t = 1.0 - pen_g / sn; fg = (0.0 < t) ? t : 0.0; }      // (quirk Q6: 0 / 0 -> NaN -> 0)
            else if (K.kind == K_GRP_MCP) fg = mcp_norm(sn, pen_g, K.D, K.gamma);
            else fg = scad_norm(sn, pen_g, K.D, K.gamma);
        }
        bn = (fg != 0.0) ? cdiv(uu * fg, K.D, rD) : 0.0;
    } else if (K.kind == K_SOFT) bn = cdiv(shrink(u, tp), K.D, rD);
    else if (K.kind == K_MCP) {
        const bool big = fabs(u) > gammad * tp;
        bn = cdiv(big ? u : shrink(u, tp), big ? K.D : dmg, big ? rD : rdmg);
    } else if (K.kind == K_SCAD) {
        const double au = fabs(u);
        const bool big = au > gammad * tp, mid = !big && au > (K.D + 1.0) * tp;
        const double num = big ? u : (mid ? shrink(gm1 * u, K.gamma * tp) : shrink(u, tp));
        bn = cdiv(num, mid ? dsc : K.D, mid ? rdsc : rD);
    } else bn = cdiv(u, d, rd);
    if constexpr (GEN) {
        if (A.accelerate) {                                         // ref src/oem_dense.h:633-651 (the sequence restarts at a penalty's cold start, not between lambdas)
            const double ak = fresh ? 1.0 : (atot > 0.0 ? 1.0 : akn_prev);
            const double akn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * ak * ak)), ratio = (ak - 1.0) / akn;
            const double upd = bn, diff = upd - b0;
            bn = upd + ratio * diff;
            const double at = block_sum(own ? (bn - upd) * diff : 0.0, gsh);
            if (tid == 0) { gen[(size_t)(par ^ 1) * FMAXB + blockIdx.x] = at; if (blockIdx.x == 0) gen[4 * FMAXB + 4 * (par ^ 1)] = akn; }
        }
    }
    if (resc) bn = bo * sinvc;
    const double c = fabs(bn), qo = fabs(b0);
    const bool cn = c > 1e-13, qn = qo > 1e-13;
    const bool moving = own && (resc || (cn != qn) || (cn && qn && fabs(bn - b0) > A.tol * qo));
    if (own) bout[cm] = bn;
    const unsigned long long nzb = __ballot(own && bn != 0.0);       // (wave 0 holds this workgroup's 32 coordinates)
    const int mv = __syncthreads_or(moving ? 1 : 0);
    if (tid == 0) { flags[(par ^ 1) * FMAXB + blockIdx.x] = mv; nz32[(size_t)(par ^ 1) * (qpad / SPK_HC) + blockIdx.x] = nzb != 0ull ? 1 : 0; }
}

// out = XX vec through the packed triangle (pack, then `reps` products back to back between two HIP events on the stream):
// *us_per_product is the product kernel's own duration -- what bench.py prices against the HBM peak (oemgpu_selftest_sympk_gemv)
int sympk_gemv_probe(hipStream_t s, const double *xx, int q, double *pk, const double *vec, double *out, int reps, double *us_per_product)
{
    const int nb = spk_nblk(q), qpad = nb * SYM_TB, nt = (int)spk_ntile(q);
    double *P = pk + (size_t)nt * SPK_TILE;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    OEM_HIP(hipEventCreate(&e0));
    OEM_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(sympk_pack_kernel, dim3(nt), dim3(256), 0, s, xx, q, pk);
    hipLaunchKernelGGL(sympk_gemv_kernel, dim3(nt), dim3(256), 0, s, pk, q, qpad, vec, P, (const int *)nullptr, (const int *)nullptr);     // (warm)
    OEM_HIP(hipEventRecord(e0, s));
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(sympk_gemv_kernel, dim3(nt), dim3(256), 0, s, pk, q, qpad, vec, P, (const int *)nullptr, (const int *)nullptr);
    OEM_HIP(hipEventRecord(e1, s));
    hipLaunchKernelGGL(sympk_sum_kernel, dim3(qpad / SPK_HC), dim3(256), 0, s, P, nb, q, qpad, out, (const int *)nullptr);
    OEM_HIP(hipGetLastError());
    OEM_HIP(hipStreamSynchronize(s));
    float ms = 0.0f;
    OEM_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (us_per_product) *us_per_product = reps > 0 ? 1e3 * (double)ms / reps : 0.0;
    return 0;
}

size_t sympk_doubles(int q)
{
    if (q <= 1024) return 0;
    const size_t nblk = (size_t)spk_nblk(q), qpad = nblk * SYM_TB;
    // blocks | partial vectors P[NBLK][qpad] | B[2][qpad] | flags[2][FMAXB] ints | SState[2] | done word | nz32[2][qpad / 32] ints
    // ... | gen: the head's parts of the sums over all coordinates [4][FMAXB] + its state words [2][4] (sympk_head_kernel<.., true>)
    return spk_ntile(q) * SPK_TILE + nblk * qpad + 2 * qpad + FMAXB + 16 + 8 + qpad / 32 + 8 + 4 * FMAXB + 16;
}

// ------------------------------------------------------------------------------------------------
// Fused iteration, replicated-update form (group penalties, accelerate, compute.loss, scale.factor; same q): what crosses
// launches is u = d beta - XX beta + XY.  EVERY workgroup thresholds the whole u itself (q <= 4096 coordinates: a few
// hundred nanoseconds, identical in every workgroup), applies the stop rule, takes the state transition at once, puts
// beta into registers and streams its rows of XX; workgroup 0 also leaves beta behind for the next stop rule.  One launch
// per iteration instead of two; at p = 512 the iteration is launch-latency bound, so that halves it.
// ------------------------------------------------------------------------------------------------
struct GState {
    int pp, i, it, done, fresh, pending_loss, finish_after_loss, pen;     // pen = penalty[pp], lam = lambda_out[pp * nl + i]:
    double ak, lam;                                                        // written by whoever advances (pp, i), so that a launch needs ONE dependent load
};

// FULL: q == 64 VPL and 16-byte aligned rows (no bounds logic); otherwise any q <= 64 VPL: columns past q read a clamped
// address and meet a zero vector entry (branch-free, as in gemv_sym_kernel).
template <int VPL, bool FULL>
__global__ __launch_bounds__(256) void oem_fused_rep_kernel(PathArgs A, GState *__restrict__ S, double *__restrict__ Ubuf,
                                                             double *__restrict__ Bbuf, int *__restrict__ fdone, int par, double d)
{
    extern __shared__ __attribute__((aligned(16))) double dyn[];     // Ush[q] | Bsh[q] | F[ngroups]
    __shared__ double sh[16];
    const int q = A.p, nl = A.nl, tid = threadIdx.x, nt = 256, lane = tid & 63;
    const int wave = blockIdx.x * 4 + (tid >> 6), nwave = gridDim.x * 4;
    // One dependent global load per launch: the state.  Everything else a launch reads is requested BEFORE the state is looked at
    // (this thread's entries of u, beta_t, XY, the penalty factors, and the first matrix row of its wave), and the penalty code and
    // lambda of (pp, i) travel inside the state: a launch is nothing but latency (6.4 us at q = 512 before this, 5.6 us after).
    const GState st = S[par];
    const double *__restrict__ uin = Ubuf + (size_t)par * (q + 8), *__restrict__ bprev = Bbuf + (size_t)par * (q + 8);
    constexpr int NPRE = 4;                                    // entries per thread covered by the prefetch: q <= 1024
    double pre_u[NPRE], pre_xy[NPRE], pre_b[NPRE], pre_pf[NPRE];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const int j = tid + k * nt;
        const bool ok = j < q;
        pre_u[k] = ok ? uin[j] : 0.0; pre_xy[k] = ok ? A.xy[j] : 0.0; pre_b[k] = ok ? bprev[j] : 0.0; pre_pf[k] = ok ? A.pf[j] : 0.0;
    }
    constexpr bool PREROW = FULL && VPL <= 16;                 // the first row of this wave (rows wave, wave + nwave, ...)
    v2d prow[PREROW ? VPL / 2 : 1];
    if (PREROW && wave < q) {
        const double *row = A.xx + (size_t)wave * q;
#pragma unroll
        for (int j = 0; j < VPL / 2; ++j) prow[j] = *reinterpret_cast<const v2d *>(row + 2 * lane + 128 * j);
    }
    if (st.done) {
        if (blockIdx.x == 0 && tid == 0) { S[par ^ 1].done = 1; *fdone = 1; }
        return;
    }
    const bool b0 = blockIdx.x == 0;
    double *Ush = dyn, *Bsh = dyn + q, *F = dyn + 2 * q;
    double *__restrict__ uout = Ubuf + (size_t)(par ^ 1) * (q + 8), *__restrict__ bkeep = Bbuf + (size_t)(par ^ 1) * (q + 8);
    const double scaley = A.yscale ? A.stats[1] : 1.0;
    const double yy = A.stats[2], nobs = A.stats[3];
    const bool fresh = st.fresh != 0;
    // ---- loss of the lambda that converged in the previous launch: XX beta_final = d beta_final - u + XY
    if (st.pending_loss >= 0) {
        double t = 0.0;
        for (int j = tid; j < q; j += nt) { const double b = bprev[j], g = (d * b - uin[j]) + A.xy[j]; t += b * (g - 2.0 * A.xy[j]); }
        t = block_sum(t, sh);
        if (b0 && tid == 0) A.loss[st.pending_loss] = yy + nobs * t;
    }
    if (st.finish_after_loss) {
        if (b0 && tid == 0) { GState nx = st; nx.done = 1; nx.pending_loss = -1; S[par ^ 1] = nx; }
        return;
    }
    const int pp = st.pp, i = st.i;
    const int pen = st.pen;
    const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
    const PenK K = pen_consts(pen, st.lam / scaley, d, A.alpha, A.gamma, A.tau);
    const double rD = 1.0 / K.D, gammad = K.gamma * K.D, dmg = K.D - 1.0 / K.gamma, rdmg = 1.0 / dmg;
    const double gm1 = K.gamma - 1.0, dsc = gm1 * K.D - 1.0, rdsc = 1.0 / dsc;
    const bool grp = K.kind >= K_GRP;
    double ak = fresh ? 1.0 : st.ak;
    // ---- u (a fresh penalty starts from beta = 0: u = XY) and, for the group operators, the group factors
    auto put_u = [&](int j, double u, double pfj) { Ush[j] = (grp && K.kind == K_SGL) ? soft1(u, pfj * K.L1, 1.0) : u; };
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {                               // the prefetched entries: constant register indices
        const int j = tid + k * nt;
        if (j < q) put_u(j, fresh ? pre_xy[k] : pre_u[k], pre_pf[k]);
    }
    for (int j = tid + NPRE * nt; j < q; j += nt) put_u(j, fresh ? A.xy[j] : uin[j], A.pf[j]);
    __syncthreads();
    if (grp) {
        for (int gi = tid; gi < A.ngroups; gi += nt) {
            double f = 1.0;
            if (!A.gzero[gi]) {
                double s2 = 0.0;
                for (int m = A.gstart[gi]; m < A.gstart[gi + 1]; ++m) { const double x = Ush[A.gidx[m]]; s2 += x * x; }
                s2 = sqrt(s2);
                const double pen_g = K.L * A.gw[gi];
                if (K.kind == K_GRP || K.kind == K_SGL) { const double t = 1.0 - pen_g / s2; f = (0.0 < t) ? t : 0.0; }
                else if (K.kind == K_GRP_MCP) f = mcp_norm(s2, pen_g, K.D, K.gamma);
                else f = scad_norm(s2, pen_g, K.D, K.gamma);
            }
            F[gi] = f;
        }
        __syncthreads();
    }
    // ---- beta = T(u), acceleration, stop rule (every workgroup, identically)
    bool bad = false;
    double adp = 0.0;
    const double akn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * ak * ak)), ratio = (ak - 1.0) / akn;
    auto update = [&](int j, double bprev_j, double pf_j) {
        const double bo = fresh ? 0.0 : bprev_j;
        const double u = Ush[j];
        double bn;
        if (grp) {
            const int gi = A.gid[j];
            const double f = gi >= 0 ? F[gi] : 0.0;
            bn = (f != 0.0) ? u * f / K.D : 0.0;
        } else {
            const double tp = pf_j * K.L;
            if (K.kind == K_SOFT) bn = cdiv(shrink(u, tp), K.D, rD);
            else if (K.kind == K_MCP) {
                const bool big = fabs(u) > gammad * tp;
                bn = cdiv(big ? u : shrink(u, tp), big ? K.D : dmg, big ? rD : rdmg);
            } else if (K.kind == K_SCAD) {
                const double au = fabs(u);
                const bool big = au > gammad * tp, mid = !big && au > (K.D + 1.0) * tp;
                const double num = big ? u : (mid ? shrink(gm1 * u, K.gamma * tp) : shrink(u, tp));
                bn = cdiv(num, mid ? dsc : K.D, mid ? rdsc : rD);
            } else bn = cdiv(u, d, 1.0 / d);
        }
        if (A.accelerate) {                                   // ref src/oem_dense.h:633-651
            const double upd = bn, diff = upd - bo;
            bn = upd + ratio * diff;
            adp += (bn - upd) * diff;
        }
        const double c = fabs(bn), qo = fabs(bo);
        const bool cn = c > 1e-13, qn = qo > 1e-13;
        bad |= (cn != qn);
        bad |= (cn && qn && fabs(bn - bo) > A.tol * qo);
        Bsh[j] = bn;
    };
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const int j = tid + k * nt;
        if (j < q) update(j, pre_b[k], pre_pf[k]);
    }
    for (int j = tid + NPRE * nt; j < q; j += nt) update(j, bprev[j], A.pf[j]);
    if (A.accelerate) {
        adp = block_sum(adp, sh);
        ak = (adp > 0.0) ? 1.0 : akn;
    }
    const int anybad = __syncthreads_or(bad ? 1 : 0);
    const int it = st.it + 1;
    const bool conv = !anybad;
    GState nx = st;
    nx.fresh = 0; nx.pending_loss = -1; nx.ak = ak; nx.it = it;
    bool done_now = false;
    if (conv || it >= A.maxit) {
        const size_t ki = (size_t)pp * nl + i;
        if (A.sinv) {                                           // quirk Q5: the member itself is rescaled
            for (int j = tid; j < q; j += nt) Bsh[j] *= A.sinv[j];
            __syncthreads();
        }
        for (int j = wave * 64 + lane; j < q; j += nwave * 64) A.beta[ki * q + j] = Bsh[j];
        if (b0 && tid == 0) {
            A.niter[ki] = conv ? it : A.maxit + 1;                  // ref src/oem_base.h:94-109
            if (!A.compute_loss) A.loss[ki] = 1e99;
        }
        nx.pending_loss = A.compute_loss ? (int)ki : -1;
        nx.it = 0;
        if (i + 1 < nlam) nx.i = i + 1;
        else if (pp + 1 < A.npen) { nx.pp = pp + 1; nx.i = 0; nx.fresh = 1; }
        else if (A.compute_loss) nx.finish_after_loss = 1;
        else { nx.done = 1; done_now = true; }
    }
    if (b0) {
        if (tid == 0) {
            if (nx.pp != pp || nx.i != i) { nx.pen = A.penalty[nx.pp]; nx.lam = A.lambda_out[(size_t)nx.pp * nl + nx.i]; }   // off the other workgroups' path
            S[par ^ 1] = nx;
        }
        for (int j = tid; j < q; j += nt) bkeep[j] = Bsh[j];       // beta_t for the next launch's stop rule / loss
    }
    if (done_now) return;
    // ---- g = XX beta for this workgroup's rows, u' = d beta - g + XY
    v2d v[VPL / 2];
    int c0[FULL ? 1 : VPL / 2], c1[FULL ? 1 : VPL / 2];
#pragma unroll
    for (int j = 0; j < VPL / 2; ++j) {
        const int c = 2 * lane + 128 * j;
        if (FULL) v[j] = *reinterpret_cast<const v2d *>(Bsh + c);
        else {
            c0[j] = c < q ? c : q - 1; c1[j] = c + 1 < q ? c + 1 : q - 1;
            v[j].x = c < q ? Bsh[c0[j]] : 0.0; v[j].y = c + 1 < q ? Bsh[c1[j]] : 0.0;
        }
    }
    for (int r = wave; r < q; r += nwave) {
        const double br = Bsh[r], xyr = A.xy[r];
        const double *row = A.xx + (size_t)r * q;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < VPL / 2; ++j) {
            if (FULL) {
                const v2d t = (PREROW && r == wave) ? prow[PREROW ? j : 0] : *reinterpret_cast<const v2d *>(row + 2 * lane + 128 * j);
                a0 = fma(t.x, v[j].x, a0);
                a1 = fma(t.y, v[j].y, a1);
            } else {
                a0 = fma(row[c0[j]], v[j].x, a0);
                a1 = fma(row[c1[j]], v[j].y, a1);
            }
        }
        const double g = wsum(a0 + a1);
        if (lane == 0) uout[r] = (d * br - g) + xyr;
    }
}

__global__ void fused_rep_init_kernel(GState *S, int npen, const int *penalty, const double *lambda_out)
{
    GState z; z.pp = 0; z.i = 0; z.it = 0; z.done = (npen == 0) ? 1 : 0; z.fresh = 1; z.pending_loss = -1; z.finish_after_loss = 0; z.ak = 1.0;
    z.pen = npen > 0 ? penalty[0] : 0; z.lam = npen > 0 ? lambda_out[0] : 0.0;
    S[0] = z; S[1] = z; S[1].done = 0;
}

__global__ void fused_init_kernel(FState *S, int npen)
{
    FState z; z.pp = 0; z.i = 0; z.it = 0; z.done = (npen == 0) ? 1 : 0; z.fresh = 1; z.pad0 = z.pad1 = z.pad2 = 0;
    S[0] = z; S[1] = z; S[1].done = 0;
}

static size_t update_uf_doubles(int p) { return 2 * (size_t)(p + 8); }       // U[q] | F[ngroups <= q]
static size_t sym_part_doubles(int p) { return p == 4096 ? 2 * (size_t)(p / SYM_TB) * p + 16 : 0; }

size_t path_large_work_doubles(int p, int nsteps)
{
    (void)nsteps;
    // + fused engine: FState[2] (8 doubles), done word, beta[2][p+8], flags[2][FMAXB] ints
    // + replicated-update engine: GState[2] (16 doubles), u[2][p+8] (beta[2] shared with the fused engine)
    // + fused Lanczos: two copies each of v, v_prev, w
    // + symmetric-tile engine (p = 4096): the partial vectors P[2][p / 128][p]
    return (size_t)STATE_DBL + 5 * (size_t)(p + 8) + 2 * MAXL + 64 + 16 + 2 * (size_t)(p + 8) + FMAXB + 16 + 2 * (size_t)(p + 8) + 6 * (size_t)(p + 8) +
           sym_part_doubles(p) + update_uf_doubles(p);
}
// where path_update_kernel keeps U | F when they do not fit its LDS: the tail of the workspace
static double *update_uf(const PathArgs &a) { return a.work + (path_large_work_doubles(a.p, 0) - update_uf_doubles(a.p)); }
// LDS bytes of the update kernel (U[q] | F[ngroups] for group operators), or 0 with *uf set when they go to global memory
static size_t update_lds(const PathArgs &a, double **uf)
{
    *uf = nullptr;
    if (a.ngroups <= 0) return 64;
    const size_t sh = sizeof(double) * (size_t)(a.p + a.ngroups + 8);
    if (sh > 160 * 1024 - 4096 && a.ngroups <= a.p) { *uf = update_uf(a); return 64; }
    return sh;
}

// `enq(FB)` enqueues FB iterations (FB even: every batch starts at launch parity 0); they are captured ONCE into a hipGraph and replayed
// until the device says done (eager launches are host-bound at ~3.5 us each); the host looks at one word per batch.
template <typename F>
static int replay_batches(hipStream_t s, F &&enq, const int *done_dev, int *hdone, long long max_it, const char *what)
{
    const int FB = 128;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        enq(FB);
        if (hipStreamEndCapture(s, &graph) != hipSuccess || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            graph = nullptr; exec = nullptr;
            (void)hipGetLastError();
        }
    } else (void)hipGetLastError();
    long long launched = 0;
    int rc = 0;
    for (;;) {
        if (exec) { if (hipGraphLaunch(exec, s) != hipSuccess) { set_error("hipGraphLaunch failed"); rc = OEMGPU_ERR_HIP; break; } }
        else enq(FB);
        if (hipGetLastError() != hipSuccess) { set_error("%s: launch failed", what); rc = OEMGPU_ERR_HIP; break; }
        launched += FB;
        if (hipMemcpyAsync(hdone, done_dev, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { set_error("%s: device error", what); rc = OEMGPU_ERR_HIP; break; }
        if (*hdone) break;
        if (caller_interrupted()) { set_error("interrupted by the caller"); rc = OEMGPU_ERR_INTERRUPTED; break; }
        if (launched > max_it) { set_error("%s did not finish within %lld iterations", what, max_it); rc = OEMGPU_ERR_INTERNAL; break; }
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
}

#ifdef OEM_PATH_DIAG
extern "C" __attribute__((visibility("default"))) int oemgpu_diag_read_update(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_update), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_diag_update), z, sizeof z); }
    return 0;
}
#endif

// host_scratch: pinned host memory (>= 8 KB)
int run_path_large(hipStream_t s, const PathArgs &a, double *host_scratch)
{
    const int q = a.p;
    int dev = 0, num_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev);
    LState *st = reinterpret_cast<LState *>(a.work);
    static_assert(sizeof(LState) + 16 <= STATE_DBL * sizeof(double), "state block too small");
    double *beta = a.work + STATE_DBL, *g = beta + (q + 8), *v = g + (q + 8), *vp = v + (q + 8), *w = vp + (q + 8);
    double *T = w + (q + 8);
    OEM_HIP(hipMemsetAsync(a.work, 0, sizeof(double) * path_large_work_doubles(q, 0), s));

    // ---- eigenvalue step: Lanczos (GEMV + single-workgroup vector update), checked every
    //      16 steps on the host, which is also the convergence test
    const int mmax = q < MAXL ? q : MAXL;
    double theta = 0.0, theta_prev = -1.0;
    bool lz_capped = true;                              // cleared by whichever rule ends the recurrence
    double *hT = host_scratch;
    int m = 0;
    // fused steps (one launch each) ping-pong between two copies of (v, v_prev, w) kept in their own area
    // (measured at q = 4096: 35.7 us fused vs 23.8 + 7.9 + a boundary -- the replicated reductions over q numbers sit in
    // front of the stream -- so the fused step is used where the launch count dominates)
    const bool lz_fused = q <= 1024 && !sw().OEM_NO_FUSED.set;
    const bool lz_full = (q == 512 || q == 1024 || q == 2048 || q == 4096) && (((uintptr_t)a.xx) & 15) == 0;
    double *LZ = T + 2 * MAXL + 64 + 16 + 2 * (size_t)(q + 8) + FMAXB + 16 + 2 * (size_t)(q + 8);
    double *Vc = LZ, *Vp = LZ + 2 * (size_t)(q + 8), *Wb = LZ + 4 * (size_t)(q + 8);
    double *SP = LZ + 6 * (size_t)(q + 8);             // symmetric-tile engine: partial vectors P[2][q / 128][q]
    // XX is symmetric: at q = 4096 the products read its lower triangle only (symgemv_kernel, oem_symfused_kernel)
    const bool sym_ok = q == 4096 && (((uintptr_t)a.xx) & 15) == 0 && !sw().OEM_NO_SYM.set && !sw().OEM_NO_FUSED.set;
    // ... and beyond 4096 a packed copy of it, made here (sympk_*: one contiguous sweep of 4 q^2 bytes per product)
    // (1024 < q <= 4096 too: what the register-resident engines do not take -- groups that are not runs of neighbours, a timed-out persistent
    // launch's second attempt -- used to run the row-streaming kernels with bounds logic at every q that is not 2048 or 4096: 50-60 us per
    // iteration at q = 3,000; the packed blocks have no ragged edge)
    const bool spk = q > 1024 && a.sympk != nullptr && !sw().OEM_NO_SYM.set;
    const int spk_nb = spk_nblk(q), spk_qpad = spk_nb * SYM_TB, spk_nt = (int)spk_ntile(q);
    double *spk_P = spk ? a.sympk + (size_t)spk_nt * SPK_TILE : nullptr, *spk_B = spk ? spk_P + (size_t)spk_nb * spk_qpad : nullptr;
    bool spk_packed = false;
    auto spk_pack = [&]() {                               // before the first product (never inside a graph capture)
        if (spk_packed) return;
        (void)hipMemsetAsync(spk_B, 0, sizeof(double) * (2 * (size_t)spk_qpad + FMAXB + 16 + 8 + spk_qpad / 32 + 8 + 4 * FMAXB + 16), s);
        hipLaunchKernelGGL(sympk_pack_kernel, dim3(spk_nt), dim3(256), 0, s, a.xx, q, a.sympk);
        spk_packed = true;
    };
    int *spk_nz = spk ? reinterpret_cast<int *>(spk_B + 2 * (size_t)spk_qpad + FMAXB + 16 + 8) : nullptr;      // nz32[2][qpad / 32] (the general form uses [0])
    auto spk_gemv = [&](const double *vec, double *out, const int *done, const int *nz = nullptr) {
        hipLaunchKernelGGL(sympk_gemv_kernel, dim3(spk_nt), dim3(256), 0, s, a.sympk, q, spk_qpad, vec, spk_P, done, nz);
        if (out) hipLaunchKernelGGL(sympk_sum_kernel, dim3(spk_qpad / SPK_HC), dim3(256), 0, s, spk_P, spk_nb, q, spk_qpad, out, done);
    };
    auto sym_gemv = [&](const double *vec, double *out) {
        if (!sym_ok) { spk_pack(); spk_gemv(vec, out, nullptr); }
        else {
            hipLaunchKernelGGL((symgemv_kernel<32>), dim3(sym_nwg(32)), dim3(256), 0, s, a.xx, vec, SP);
            hipLaunchKernelGGL((symgemv_sum_kernel<32>), dim3(q / 128), dim3(128), 0, s, SP, out);
        }
    };
    void (*lzk)(const double *, int, int, double *, double *, double *, double *, int) = nullptr;
    if (lz_fused) {
        if (q <= 512) lzk = lz_full ? lanczos_fused_kernel<8, true> : lanczos_fused_kernel<8, false>;
        else lzk = lz_full ? lanczos_fused_kernel<16, true> : lanczos_fused_kernel<16, false>;
    }
    int lblocks = (q + 3) / 4;
    if (lblocks > num_cu * 2) lblocks = num_cu * 2;
    const size_t lsh = sizeof(double) * (size_t)(q + 8);
    const bool d_given = a.d_fixed > 0.0;                // (the caller's d: no recurrence)
    if (d_given) { theta = a.d_fixed / 1.005; lz_capped = false; }
    else if (lz_fused) {
        hipLaunchKernelGGL(lanczos_init_kernel, dim3(1), dim3(1024), 0, s, q, Vc, Vp);
        int rc = launch_gemv(s, a.xx, q, Vc, Wb, nullptr, num_cu);           // w_0 = M v_0; every later product is fused
        if (rc) return rc;
    } else hipLaunchKernelGGL(lanczos_init_kernel, dim3(1), dim3(1024), 0, s, q, v, vp);
    while (m < mmax && !d_given) {
        const int chunk = (mmax - m) < 16 ? (mmax - m) : 16;     // a host look costs about one step
        for (int k = 0; k < chunk; ++k, ++m) {
            if (lz_fused) hipLaunchKernelGGL(lzk, dim3(lblocks), dim3(256), lsh, s, a.xx, q, m, Vc, Vp, Wb, T, m & 1);
            else {
                if (sym_ok || spk) sym_gemv(v, w);
                else {
                    int rc = launch_gemv(s, a.xx, q, v, w, nullptr, num_cu);
                    if (rc) return rc;
                }
                if (q > 1024 && q <= 8192) hipLaunchKernelGGL(lanczos_update_reg_kernel<8>, dim3(1), dim3(1024), 0, s, q, m, v, vp, w, T);
                else if (q > 8192 && q <= 12288) hipLaunchKernelGGL(lanczos_update_reg_kernel<12>, dim3(1), dim3(1024), 0, s, q, m, v, vp, w, T);
                else hipLaunchKernelGGL(lanczos_update_kernel, dim3(1), dim3(1024), 0, s, q, m, v, vp, w, T);
            }
        }
        OEM_HIP(hipGetLastError());
        OEM_HIP(hipMemcpyAsync(hT, T, sizeof(double) * 2 * MAXL, hipMemcpyDeviceToHost, s));
        OEM_HIP(hipStreamSynchronize(s));
        int mm = m;
        for (int k = 0; k < m; ++k)
            if (!(hT[MAXL + k] > 1e-13 * std::fabs(hT[k]))) { mm = k + 1; break; }     // breakdown: T is exact
        theta = tridiag_max_host(hT, hT + MAXL, mm);
        if (mm < m) { lz_capped = false; break; }
        // the stop rule of the register-resident engines (path_dev.hpp: lanczos_converged) on the top Ritz values of the leading
        // blocks T_{m-16}, T_{m-8}, T_m, which the host has for free: moved by <= 1e-14 relative over the last 8 steps, or two
        // successive moves that decay so fast that their geometric tail is <= 1e-12 relative
        if (m >= 24) {
            const double t1 = tridiag_max_host(hT, hT + MAXL, m - 8), t0 = tridiag_max_host(hT, hT + MAXL, m - 16);
            const double mv = theta - t1, mvp = t1 - t0, ath = std::fabs(theta);
            if (mv <= 1e-14 * ath || (mv < 0.01 * mvp && mv * mv <= OEM_LANCZOS_TAIL_TOL * ath * (mvp - mv))) { lz_capped = false; break; }
        }
        if (theta_prev > 0 && std::fabs(theta - theta_prev) <= 1e-12 * std::fabs(theta)) { lz_capped = false; break; }
        theta_prev = theta;
    }
    if (mmax >= q) lz_capped = false;                   // the whole Krylov space
    const double d = d_given ? a.d_fixed : theta * 1.005;     // ref src/oem_dense.h:498

    hipLaunchKernelGGL(path_init_kernel, dim3(1), dim3(1024), 0, s, a, st, beta, d, theta, m, lz_capped ? 1 : 0);
    OEM_HIP(hipGetLastError());
    if (a.npen == 0) return 0;

    // ---- fused engine when the operators are row-local and nothing needs a global sum per iteration
    const bool elementwise = a.ngroups == 0 && !a.accelerate && !a.compute_loss && !a.sinv;
    const bool fused_ok = elementwise && (q == 512 || q == 1024 || q == 2048 || q == 4096) &&
                          (((uintptr_t)a.xx) & 15) == 0 && !sw().OEM_NO_FUSED.set;
    if (fused_ok) {
        double *fbase = T + 2 * MAXL + 64;
        FState *S = reinterpret_cast<FState *>(fbase);
        int *fdone = reinterpret_cast<int *>(fbase + 8);
        double *Bv = fbase + 16;
        int *flags = reinterpret_cast<int *>(Bv + 2 * (size_t)(q + 8));
        int blocks = (q + 3) / 4;
        if (blocks > num_cu * 2) blocks = num_cu * 2;
        if (blocks > FMAXB) blocks = FMAXB;
        const bool sym = sym_ok;
        SState *SS = reinterpret_cast<SState *>(SP + 2 * (size_t)(q / SYM_TB) * q);      // behind the partial vectors (sym_part_doubles)
        static_assert(2 * sizeof(SState) <= 16 * sizeof(double), "SState[2] must fit the 16 spare doubles of the partial area");
        if (sym) hipLaunchKernelGGL(sym_init_kernel, dim3(1), dim3(1), 0, s, SS, a);
        else hipLaunchKernelGGL(fused_init_kernel, dim3(1), dim3(1), 0, s, S, a.npen);
        auto enq = [&](int count) {
            for (int k = 0; k < count; ++k) {
                const int par = k & 1;
                if (sym) hipLaunchKernelGGL((oem_symfused_kernel<32>), dim3(sym_nwg(32)), dim3(256), 0, s, a, SS, Bv, SP, flags, fdone, par, d);
                else if (q == 512) hipLaunchKernelGGL((oem_fused_kernel<8>), dim3(blocks), dim3(256), 0, s, a, S, Bv, flags, fdone, par, d);
                else if (q == 1024) hipLaunchKernelGGL((oem_fused_kernel<16>), dim3(blocks), dim3(256), 0, s, a, S, Bv, flags, fdone, par, d);
                else if (q == 2048) hipLaunchKernelGGL((oem_fused_kernel<32>), dim3(blocks), dim3(256), 0, s, a, S, Bv, flags, fdone, par, d);
                else hipLaunchKernelGGL((oem_fused_kernel<64>), dim3(blocks), dim3(256), 0, s, a, S, Bv, flags, fdone, par, d);
            }
        };
        return replay_batches(s, enq, fdone, reinterpret_cast<int *>(host_scratch), (long long)a.npen * a.nl * ((long long)a.maxit + 2) + 8, "fused engine");
    }

    // ---- element-wise penalties at every other q > 1024: (head, product) pairs over the packed triangle
    // (... and group operators whose groups are runs of <= 32 neighbouring coordinates, PathArgs::grp_head: the same pairs, sympk_head_kernel<true>)
    // (... and Nesterov's step / compute.loss: their sums over all coordinates are taken one launch later, sympk_head_kernel<.., true>)
    const bool grp_head = a.grp_head > 0 && a.grp_head <= 3 && a.ngroups > 0;
    const bool head_gen = a.accelerate || a.compute_loss;
    if (spk && (a.ngroups == 0 || grp_head) && spk_qpad / SPK_HC <= FMAXB && !sw().OEM_NO_FUSED.set) {
        spk_pack();
        int *flags = reinterpret_cast<int *>(spk_B + 2 * (size_t)spk_qpad);
        SState *SS = reinterpret_cast<SState *>(spk_B + 2 * (size_t)spk_qpad + FMAXB);
        int *fdone = reinterpret_cast<int *>(spk_B + 2 * (size_t)spk_qpad + FMAXB + 16);
        int *nz32 = reinterpret_cast<int *>(spk_B + 2 * (size_t)spk_qpad + FMAXB + 16 + 8);      // [2][qpad / 32]: which 32-coordinate pieces of B[parity] hold a non-zero
        double *gen = spk_B + 2 * (size_t)spk_qpad + FMAXB + 16 + 8 + spk_qpad / 32 + 8;          // the head's parts of the sums over all coordinates (zeroed with B)
        hipLaunchKernelGGL(sym_init_kernel, dim3(1), dim3(1), 0, s, SS, a);
        auto enq = [&](int count) {
            for (int k = 0; k < count; ++k) {
                const int par = k & 1;
                const int hb = grp_head ? a.grp_head : 0;
                void (*hk)(PathArgs, SState *, double *, const double *, int *, int *, int, double, int, int, int *, double *) =
                    head_gen ? (hb == 0 ? sympk_head_kernel<0, true> : hb == 1 ? sympk_head_kernel<1, true> : hb == 2 ? sympk_head_kernel<2, true> : sympk_head_kernel<3, true>)
                             : (hb == 0 ? sympk_head_kernel<0, false> : hb == 1 ? sympk_head_kernel<1, false> : hb == 2 ? sympk_head_kernel<2, false> : sympk_head_kernel<3, false>);
                hipLaunchKernelGGL(hk, dim3(spk_qpad / SPK_HC), dim3(256), 0, s, a, SS, spk_B, spk_P, flags, fdone, par, d, spk_nb, spk_qpad, nz32, head_gen ? gen : (double *)nullptr);
                hipLaunchKernelGGL(sympk_gemv_kernel, dim3(spk_nt), dim3(256), 0, s, a.sympk, q, spk_qpad, spk_B + (size_t)(par ^ 1) * spk_qpad, spk_P, fdone,
                                   (const int *)(nz32 + (size_t)(par ^ 1) * (spk_qpad / SPK_HC)));
            }
        };
        return replay_batches(s, enq, fdone, reinterpret_cast<int *>(host_scratch), (long long)a.npen * a.nl * ((long long)a.maxit + 2) + 8, "packed-triangle engine");
    }

    // ---- replicated-update fused engine: everything else up to q = 2048 (one launch per iteration, every workgroup thresholds the whole of u).
    // Beyond, the packed products + slot sum + update kernel are faster (tools/scattered_groups_time.py, grp.lasso with 60 scattered groups,
    // us per iteration: q = 3,000 65.9 -> 22.4, 4,096 47.0 -> 29.0; but 1,536 13.9 -> 15.7, 2,048 15.6 -> 15.8: three launches against one)
    const bool rep_ok = q <= 4096 && !sw().OEM_NO_FUSED.set && !(spk && q > 2048);
    if (rep_ok) {
        double *fbase = T + 2 * MAXL + 64;
        int *fdone = reinterpret_cast<int *>(fbase + 8);
        double *Bv = fbase + 16;
        double *gbase = Bv + 2 * (size_t)(q + 8) + FMAXB;
        GState *S = reinterpret_cast<GState *>(gbase);
        double *Uv = gbase + 16;
        int blocks = (q + 3) / 4;
        if (blocks > num_cu * 2) blocks = num_cu * 2;
        const size_t shb = sizeof(double) * (size_t)(2 * q + (a.ngroups > 0 ? a.ngroups : 0) + 8);
        const bool full = (q == 512 || q == 1024 || q == 2048 || q == 4096) && (((uintptr_t)a.xx) & 15) == 0;
        // kernel for this q: VPL = 8, 16, 32 or 64 columns per lane
        void (*kern)(PathArgs, GState *, double *, double *, int *, int, double);
        if (q <= 512) kern = full ? oem_fused_rep_kernel<8, true> : oem_fused_rep_kernel<8, false>;
        else if (q <= 1024) kern = full ? oem_fused_rep_kernel<16, true> : oem_fused_rep_kernel<16, false>;
        else if (q <= 2048) kern = full ? oem_fused_rep_kernel<32, true> : oem_fused_rep_kernel<32, false>;
        else kern = full ? oem_fused_rep_kernel<64, true> : oem_fused_rep_kernel<64, false>;
        if (shb > 64 * 1024) OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
        hipLaunchKernelGGL(fused_rep_init_kernel, dim3(1), dim3(1), 0, s, S, a.npen, a.penalty, a.lambda_out);
        auto enq = [&](int count) {
            for (int k = 0; k < count; ++k) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shb, s, a, S, Uv, Bv, fdone, k & 1, d);
        };
        return replay_batches(s, enq, fdone, reinterpret_cast<int *>(host_scratch), (long long)a.npen * a.nl * ((long long)a.maxit + 3) + 8, "fused engine");
    }

    // ---- path: (gemv, update) pairs replayed in batches from a hipGraph (eager launches are host-bound at ~3.5 us
    //      each); the host polls the done word once per batch.  Fusing the pair into one launch with a last-arriver
    //      hand-off was measured and is NOT faster: the agent-scope release + acquire cost what the boundary costs.
    double *uf = nullptr;
    const size_t sh = update_lds(a, &uf);                            // (U and F exist for group operators only)
    // (1024 < q <= 8192: the operands in registers, every independent load of the kernel issued at once)
    void (*updk)(PathArgs, LState *, double *, const double *, double *, int *) =
        (q > 1024 && q <= 4096) ? path_update_kernel<4> : (q > 4096 && q <= 8192) ? path_update_kernel<8> : path_update_kernel<0>;
    if (sh > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(updk),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS %zu): %s", sh, hipGetErrorString(e)); return OEMGPU_ERR_HIP; }
    }
    if (spk) spk_pack();
    auto enqueue = [&](int count) {
        for (int k = 0; k < count; ++k) {
            if (spk) spk_gemv(beta, g, &st->done, spk_nz);            // (packed before the capture, below; blocks of zeros skipped by the update kernel's flags)
            else (void)launch_gemv(s, a.xx, q, beta, g, &st->done, num_cu);
            hipLaunchKernelGGL(updk, dim3(1), dim3(1024), sh, s, a, st, beta, g, uf, spk_nz);
        }
    };
    return replay_batches(s, enqueue, &st->done, reinterpret_cast<int *>(host_scratch), (long long)a.npen * a.nl * ((long long)a.maxit + 2) + 8, "large-p engine");
}

// ================================================================================================
// p >= n: the reference's own iteration, without a Gram matrix (ref src/oem_dense.h:363-366, 476-482, 513-521):
//     u = Xs'(Ys - Xs beta)/n + d beta,      d = 1.005 lambda_max(Xs Xs'/n)
// Xs is the standardised copy (wide.hip): npad x p column-major, npad = 64 NR, padding rows zero.  A column is 8 n bytes and
// coordinate-local: ONE read of Xs per iteration does both products --
//   wide_cols_kernel<NR, W_OEM>   a wave owns columns j, j + 4, ...: x_j in registers (NR per lane), u_j = x_j . r / n + d beta_j
//                                 (wave sum), beta_j' = T(u_j) on the spot (the operator is coordinate-local), and the SAME
//                                 registers feed r' += x_j beta_j' (lane-local; skipped when beta_j' = 0: sparse paths stream
//                                 for the dot product only).  The workgroup's share of Xs beta' leaves as a partial vector.
//   wide_reduce_kernel<W_OEM>     r' = Ys - sum of the partial vectors, in workgroup order (fixed order, no atomics).
// 8 n p bytes per iteration where the reference's two GEMVs read 16 n p (and the Gram form 8 p^2).  Stop rule and lambda / penalty
// bookkeeping replicated one launch later, state by launch parity, exactly as the symmetric-tile engine above.
// Everything that needs the whole of u at once (group operators, Nesterov's step, compute.loss) runs the same two products as
// separate launches around path_update_kernel: W_XB (Xs beta as partial vectors), the reduction, W_XTV (g = Xs' t / n).
// The eigenvalue step is Lanczos on Xs Xs'/n with the same kernels (W_EIG: partial vectors of Xs (Xs' v)), n-vectors only.
// ================================================================================================
enum { W_EIG = 0, W_OEM = 1, W_XB = 2, W_XTV = 3 };

static int wide_nr(int n)
{
    static const int sizes[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    const int need = (n + 63) / 64;
    for (int v : sizes) if (v >= need) return v;
    return 0;
}
// waves per workgroup: as many as the combine buffer in LDS (NW x 64 NR doubles <= 64 KB) and the registers (3-4 NR doubles per
// lane) allow -- a wave walks its columns one after the other (dot product -> wave sum -> operator -> update is a serial chain of
// ~400 cycles per column), so the CU needs several waves per SIMD to keep loads in flight
__host__ __device__ constexpr int wide_nw(int nr) { return nr <= 8 ? 16 : (nr <= 16 ? 8 : 4); }
WideLayout wide_layout(int64_t n)
{
    WideLayout L;
    L.nb = n <= 2048 ? 1 : (int)((n + 2047) / 2048);
    L.rb = (int)((n + L.nb - 1) / L.nb);
    L.nr = wide_nr(L.rb);
    return L;
}
int wide_workgroups(int n, int p)
{
    const int nw = wide_nw(wide_layout(n).nr);
    int w = (p + nw - 1) / nw;                           // small p is latency-bound: a column per wave, as many CUs as that gives
    if (w > 256) w = 256;                                // one workgroup per CU: every partial vector is read back by the reduction
    return w < 1 ? 1 : w;
}
// P[W][npad] | r | t | v | vp | w (rows() each) | T[2 MAXL + 64] | SState[2] (16) | done (2) | flags[2][FMAXB] ints | gb[nb][p + 8]
size_t wide_scratch_doubles(int n, int p)
{
    const WideLayout L = wide_layout(n);
    const size_t own = (size_t)wide_workgroups(n, p) * L.npad() + 5 * (size_t)L.rows() + 2 * MAXL + 64 + 16 + 2 + FMAXB + 64 + (size_t)L.nb * (p + 8);
    size_t coop = path_wcoop_xchg_doubles(n, p);               // the persistent engines' exchange buffers live in the same scratch
    if (path_wstream_xchg_doubles((int)n) > coop) coop = path_wstream_xchg_doubles((int)n);
    if (path_wres_xchg_doubles(n, p) > coop) coop = path_wres_xchg_doubles(n, p);
    return own > coop ? own : coop;
}

// sum of p[w * stride] over w = first, first + step, ... < count, in that order, eight loads in flight (a loop of dependent-free
// loads whose trip count the compiler does not know is issued one load per memory round trip)
__device__ __forceinline__ double chain_sum(const double *__restrict__ p, int first, int count, int step, size_t stride)
{
    double s = 0.0;
    int w = first;
    for (; w + 7 * step < count; w += 8 * step) {
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = p[(size_t)(w + k * step) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += t[k];
    }
    for (; w < count; w += step) s += p[(size_t)w * stride];
    return s;
}

template <int I, int N, typename F> __device__ __forceinline__ void const_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); const_for<I + 1, N>(f); }
}

template <int NR> __device__ __forceinline__ void wide_load(double (&x)[NR], const double *__restrict__ col, int lane)
{
#pragma unroll
    for (int k = 0; k < NR; ++k) x[k] = col[lane + 64 * k];
}

template <int NR, int MODE>
__global__ __launch_bounds__(64 * wide_nw(NR)) void wide_cols_kernel(PathArgs A, const double *__restrict__ xs, const double *__restrict__ rin,
                                                         const double *__restrict__ ysv, double *__restrict__ P, double *__restrict__ beta,
                                                         double *__restrict__ outv, SState *__restrict__ S, int *__restrict__ flags,
                                                         int *__restrict__ fdone, const int *__restrict__ done, int par, double d,
                                                         int n, int cpw)
{
    extern __shared__ __attribute__((aligned(16))) double wsh[];     // [NW][64 NR] | beta of this workgroup's columns [cpw] | penalty factors [cpw]
    constexpr int NP = 64 * NR, NW = wide_nw(NR), NT = 64 * NW;
    double *bsh = wsh + NW * NP, *pfsh = bsh + cpw;
    const int q = A.p, nl = A.nl, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jbeg = blockIdx.x * cpw, jend = (jbeg + cpw < q) ? jbeg + cpw : q;
    const double rn = 1.0 / (double)n;
    // ---- head (W_OEM): state, flags -- one load away from the launch (SState carries the penalty code and lambda)
    SState st;
    int fl[(FMAXB + NT - 1) / NT];
    if (MODE == W_OEM) {
        st = S[par];
#pragma unroll
        for (int k = 0; k < (FMAXB + NT - 1) / NT; ++k) { const int t = tid + NT * k; fl[k] = flags[par * FMAXB + (t < (int)gridDim.x ? t : 0)]; }
    } else if (done && *done) return;
    // the workgroup's coefficients and penalty factors into LDS with one coalesced load each: a load per column inside the loop
    // would put a memory round trip on every column's chain
    if (MODE == W_OEM || MODE == W_XB)
        for (int t = tid; t < jend - jbeg; t += NT) { bsh[t] = beta[jbeg + t]; if (MODE == W_OEM) pfsh[t] = A.pf[jbeg + t]; }
    int j = jbeg + w;
    double xc[NR];
    if (MODE != W_XB && j < jend) wide_load<NR>(xc, xs + (size_t)j * NP, lane);
    if (MODE == W_XB) __syncthreads();
    int pp = 0, i = 0, it = 0, pen = 0, niter_fin = 0;
    double lam = 0.0;
    bool fresh = false, finalize = false, done_now = false, advanced = false;
    size_t kfin = 0;
    if (MODE == W_OEM) {
        if (st.done) {
            if (blockIdx.x == 0 && tid == 0) { S[par ^ 1].done = 1; *fdone = 1; }
            return;
        }
        int f = 0;
#pragma unroll
        for (int k = 0; k < (FMAXB + NT - 1) / NT; ++k) f |= (tid + NT * k < (int)gridDim.x) ? fl[k] : 0;
        const int any = __syncthreads_or(f);
        pp = st.pp; i = st.i; it = st.it; pen = st.pen; lam = st.lam; fresh = st.fresh != 0;
        if (!fresh) {
            const bool conv = !any;
            if (conv || it >= A.maxit) {
                finalize = true; kfin = (size_t)pp * nl + i; niter_fin = conv ? it : A.maxit + 1;     // ref src/oem_base.h:94-109
                const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
                if (i + 1 < nlam) { i = i + 1; advanced = true; }
                else if (pp + 1 < A.npen) { pp = pp + 1; i = 0; fresh = true; advanced = true; }
                else done_now = true;
                if (advanced) { pen = st.pen_next; lam = st.lam_next; }
                it = 0;
            }
        }
        if (blockIdx.x == 0 && tid == 0) {
            SState nx = st;
            nx.pp = pp; nx.i = i; nx.it = it + 1; nx.done = done_now ? 1 : 0; nx.fresh = 0; nx.pen = pen; nx.lam = lam;
            if (advanced) sym_successor(A, pp, i, pen, nx.pen_next, nx.lam_next);
            S[par ^ 1] = nx;
            if (finalize) { A.niter[kfin] = niter_fin; A.loss[kfin] = 1e99; }
        }
        if (done_now) {                                             // only the last lambda's coefficients are left to store
            for (int jj = jbeg + tid; jj < jend; jj += NT) A.beta[kfin * q + jj] = bsh[jj - jbeg];
            return;
        }
    }
    // ---- the n-vector this launch multiplies with (W_OEM: the residual; a fresh penalty starts from beta = 0, so r = Ys)
    double rr[NR], rp[NR];
    if (MODE != W_XB) wide_load<NR>(rr, (MODE == W_OEM && fresh) ? ysv : rin, lane);
#pragma unroll
    for (int k = 0; k < NR; ++k) rp[k] = 0.0;
    PenK K;
    double rD = 0.0, gammad = 0.0, dmg = 0.0, rdmg = 0.0, gm1 = 0.0, dsc = 0.0, rdsc = 0.0, rd = 0.0;
    if (MODE == W_OEM) {
        K = pen_consts(pen, lam / st.scaley, d, A.alpha, A.gamma, A.tau);
        rD = 1.0 / K.D; gammad = K.gamma * K.D; dmg = K.D - 1.0 / K.gamma; rdmg = 1.0 / dmg;
        gm1 = K.gamma - 1.0; dsc = gm1 * K.D - 1.0; rdsc = 1.0 / dsc; rd = 1.0 / d;
    }
    bool moving = false;
    // the wave's columns j, j + NW, ...: a ring of D columns in registers, each requested D columns ahead of its use (a column is a
    // memory round trip; with one column in flight per wave the small problems were nothing but that latency)
#ifndef OEM_WIDE_DEPTH_A
#define OEM_WIDE_DEPTH_A 2          // NR <= 4
#define OEM_WIDE_DEPTH_B 2          // NR <= 8 (128 VGPRs per wave at sixteen waves per workgroup: 4 x 8 doubles spill)
#define OEM_WIDE_DEPTH_C 2          // NR <= 16
#endif
    constexpr int D = (MODE == W_XB) ? 1 : (NR <= 4 ? OEM_WIDE_DEPTH_A : (NR <= 8 ? OEM_WIDE_DEPTH_B : (NR <= 16 ? OEM_WIDE_DEPTH_C : 1)));
    double xq[D][NR];
    if (MODE != W_XB) {
#pragma unroll
        for (int k = 0; k < NR; ++k) xq[0][k] = xc[k];               // (column j was requested in front of the head)
        const_for<1, D>([&](auto DD) __attribute__((always_inline)) {
            constexpr int dd = decltype(DD)::value;
            if (j + dd * NW < jend) wide_load<NR>(xq[dd], xs + (size_t)(j + dd * NW) * NP, lane);
        });
    }
    // one column (a macro, not a lambda: a closure that captures the accumulator arrays by reference had them spilt to scratch)
#define OEM_WIDE_COLUMN(X, JJ)                                                                                               \
    do {                                                                                                                     \
        const int jj__ = (JJ);                                                                                               \
        if (MODE == W_XB) {                                                                                                  \
            const double bj = bsh[jj__ - jbeg];                                                                              \
            if (bj != 0.0) {                       /* wave-uniform: a zero coefficient's column is never read */              \
                wide_load<NR>(X, xs + (size_t)jj__ * NP, lane);                                                              \
                _Pragma("unroll") for (int k = 0; k < NR; ++k) rp[k] = fma(X[k], bj, rp[k]);                                 \
            }                                                                                                                \
        } else {                                                                                                             \
            const double bo = (MODE == W_OEM) ? bsh[jj__ - jbeg] : 0.0, pfj = (MODE == W_OEM) ? pfsh[jj__ - jbeg] : 0.0;     \
            double a0 = 0.0, a1 = 0.0;                                                                                       \
            _Pragma("unroll") for (int k = 0; k < NR; k += 2) { a0 = fma(X[k], rr[k], a0); if (k + 1 < NR) a1 = fma(X[k + 1], rr[k + 1], a1); } \
            const double dot = wsum(a0 + a1);                                                                                \
            if (MODE == W_XTV) { if (lane == 0) outv[jj__] = dot * rn; }                                                     \
            else {                                                                                                           \
                double bn;                                                                                                   \
                if (MODE == W_EIG) bn = dot;                                                                                 \
                else {                                                                                                       \
                    const double b0 = fresh ? 0.0 : bo;                                                                      \
                    const double u = dot * rn + d * b0;        /* ref src/oem_dense.h:520: X'(Y - X beta)/n + d beta */       \
                    const double tp = pfj * K.L;                                                                             \
                    if (K.kind == K_SOFT) bn = cdiv(shrink(u, tp), K.D, rD);                                                 \
                    else if (K.kind == K_MCP) {                                                                              \
                        const bool big = fabs(u) > gammad * tp;                                                              \
                        bn = cdiv(big ? u : shrink(u, tp), big ? K.D : dmg, big ? rD : rdmg);                                \
                    } else if (K.kind == K_SCAD) {                                                                           \
                        const double au = fabs(u);                                                                           \
                        const bool big = au > gammad * tp, mid = !big && au > (K.D + 1.0) * tp;                              \
                        const double num = big ? u : (mid ? shrink(gm1 * u, K.gamma * tp) : shrink(u, tp));                  \
                        bn = cdiv(num, mid ? dsc : K.D, mid ? rdsc : rD);                                                    \
                    } else bn = cdiv(u, d, rd);                                                                              \
                    const double c = fabs(bn), qo = fabs(b0);                                                                \
                    const bool cn = c > 1e-13, qn = qo > 1e-13;                                                              \
                    moving |= (cn != qn) || (cn && qn && fabs(bn - b0) > A.tol * qo);                                        \
                    if (lane == 0) {                                                                                         \
                        if (finalize) A.beta[kfin * q + jj__] = bo;                                                          \
                        beta[jj__] = bn;                                                                                     \
                    }                                                                                                        \
                }                                                                                                            \
                if (bn != 0.0) { _Pragma("unroll") for (int k = 0; k < NR; ++k) rp[k] = fma(X[k], bn, rp[k]); }              \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
#define OEM_WIDE_SLOT(DD)                                                                                                    \
    if constexpr (D > (DD)) {                                                                                                \
        const int jj = j + (DD) * NW;                                                                                        \
        if (jj < jend) {                                                                                                     \
            OEM_WIDE_COLUMN(xq[(DD) < D ? (DD) : 0], jj);                                                                    \
            const int jn = jj + D * NW;                                                                                      \
            if (MODE != W_XB && jn < jend) wide_load<NR>(xq[(DD) < D ? (DD) : 0], xs + (size_t)jn * NP, lane);               \
        }                                                                                                                    \
    }
    for (; j < jend; j += D * NW) {
        OEM_WIDE_SLOT(0) OEM_WIDE_SLOT(1) OEM_WIDE_SLOT(2) OEM_WIDE_SLOT(3)
    }
#undef OEM_WIDE_SLOT
#undef OEM_WIDE_COLUMN
    if (MODE == W_OEM) {
        const int mv = __syncthreads_or(moving ? 1 : 0);
        if (tid == 0) flags[(par ^ 1) * FMAXB + blockIdx.x] = mv;
    }
    if (MODE == W_XTV) return;
    // ---- the workgroup's partial vector: the waves' sums added in wave order
#pragma unroll
    for (int k = 0; k < NR; ++k) wsh[w * NP + lane + 64 * k] = rp[k];
    __syncthreads();
    for (int r = tid; r < NP; r += NT) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) t += wsh[ww * NP + r];
        P[(size_t)blockIdx.x * NP + r] = t;
    }
}

// The fused iteration with GROUP operators (ref src/oem_dense.h:193-315, 513-521), for groups that are runs of neighbouring columns of
// at most WIDE_GRUN_MAX members: a run is coordinate-local to ONE wave, as a column is in wide_cols_kernel<W_OEM>.  The wave walks
// its runs (run w, w + NW, ... of the workgroup's):
//   pass 1   u_c = x_c . r / n + d beta_c for the members in member order (sparse group lasso: soft-thresholded), kept in LDS words
//            of the wave's own; the squared norm summed in member order like the reference; the group's factor
//   pass 2   beta_c' = u_c f / D, the stop rule; r' += x_c beta_c' -- the column is read AGAIN only where beta_c' != 0 (it came by a
//            moment ago: L2), so a sparse iterate streams Xs once per iteration like the element-wise form
// One launch + the reduction per iteration instead of four launches around the one-workgroup update kernel (500 x 30,000, 3,000
// groups of 10: 87 -> us per iteration; and no LDS limit on p + ngroups).  Element-wise penalties of the same call run through the
// same kernel (runs are just a way of dealing columns).  Stop rule / lambda bookkeeping replicated one launch later as in W_OEM.
template <int NR>
__global__ __launch_bounds__(64 * wide_nw(NR)) void wide_groups_kernel(PathArgs A, const double *__restrict__ xs, const double *__restrict__ rin,
                                                           const double *__restrict__ ysv, double *__restrict__ P, double *__restrict__ beta,
                                                           SState *__restrict__ S, int *__restrict__ flags, int *__restrict__ fdone, int par, double d,
                                                           int n, int cpw, const int *__restrict__ rstart, const int *__restrict__ rgid,
                                                           const int *__restrict__ wgrun)
{
    extern __shared__ __attribute__((aligned(16))) double wsh[];     // [NW][64 NR] | beta [cpw] | penalty factors [cpw] | u of a wave's run [NW][WIDE_GRUN_MAX]
    constexpr int NP = 64 * NR, NW = wide_nw(NR), NT = 64 * NW;
    double *bsh = wsh + NW * NP, *pfsh = bsh + cpw, *ush = pfsh + cpw;
    const int q = A.p, nl = A.nl, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r0 = wgrun[blockIdx.x], r1 = wgrun[blockIdx.x + 1];
    const int jbeg = rstart[r0], jend = rstart[r1];
    const double rn = 1.0 / (double)n;
    const SState st = S[par];
    int fl[(FMAXB + NT - 1) / NT];
#pragma unroll
    for (int k = 0; k < (FMAXB + NT - 1) / NT; ++k) { const int t = tid + NT * k; fl[k] = flags[par * FMAXB + (t < (int)gridDim.x ? t : 0)]; }
    for (int t = tid; t < jend - jbeg; t += NT) { bsh[t] = beta[jbeg + t]; pfsh[t] = A.pf[jbeg + t]; }
    if (st.done) {
        if (blockIdx.x == 0 && tid == 0) { S[par ^ 1].done = 1; *fdone = 1; }
        return;
    }
    int f = 0;
#pragma unroll
    for (int k = 0; k < (FMAXB + NT - 1) / NT; ++k) f |= (tid + NT * k < (int)gridDim.x) ? fl[k] : 0;
    const int any = __syncthreads_or(f);                             // (also the barrier behind bsh / pfsh)
    int pp = st.pp, i = st.i, it = st.it, pen = st.pen, niter_fin = 0;
    double lam = st.lam;
    bool fresh = st.fresh != 0, finalize = false, done_now = false, advanced = false;
    size_t kfin = 0;
    if (!fresh) {
        const bool conv = !any;
        if (conv || it >= A.maxit) {
            finalize = true; kfin = (size_t)pp * nl + i; niter_fin = conv ? it : A.maxit + 1;     // ref src/oem_base.h:94-109
            const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
            if (i + 1 < nlam) { i = i + 1; advanced = true; }
            else if (pp + 1 < A.npen) { pp = pp + 1; i = 0; fresh = true; advanced = true; }
            else done_now = true;
            if (advanced) { pen = st.pen_next; lam = st.lam_next; }
            it = 0;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        SState nx = st;
        nx.pp = pp; nx.i = i; nx.it = it + 1; nx.done = done_now ? 1 : 0; nx.fresh = 0; nx.pen = pen; nx.lam = lam;
        if (advanced) sym_successor(A, pp, i, pen, nx.pen_next, nx.lam_next);
        S[par ^ 1] = nx;
        if (finalize) { A.niter[kfin] = niter_fin; A.loss[kfin] = 1e99; }
    }
    if (finalize) for (int jj = jbeg + tid; jj < jend; jj += NT) A.beta[kfin * q + jj] = bsh[jj - jbeg];
    if (done_now) return;
    double rr[NR], rp[NR];
    wide_load<NR>(rr, fresh ? ysv : rin, lane);
#pragma unroll
    for (int k = 0; k < NR; ++k) rp[k] = 0.0;
    const PenK K = pen_consts(pen, lam / st.scaley, d, A.alpha, A.gamma, A.tau);
    const double rD = 1.0 / K.D, gammad = K.gamma * K.D, dmg = K.D - 1.0 / K.gamma, rdmg = 1.0 / dmg;
    const double gm1 = K.gamma - 1.0, dsc = gm1 * K.D - 1.0, rdsc = 1.0 / dsc, rd = 1.0 / d;
    const bool grp = K.kind >= K_GRP;
    bool moving = false;
    double *uw = ush + w * WIDE_GRUN_MAX;
    double xa[NR], xb[NR];
    int run = r0 + w;
    if (run < r1) wide_load<NR>(xa, xs + (size_t)rstart[run] * NP, lane);
    for (; run < r1; run += NW) {
        const int c0 = rstart[run], c1 = rstart[run + 1], g = rgid[run];
        const int cnext = run + NW < r1 ? rstart[run + NW] : -1;
        // ---- pass 1: u of the members (the next column is asked for before this one is consumed)
        double s2 = 0.0;
        for (int c = c0; c < c1; ++c) {
            const int cn = c + 1 < c1 ? c + 1 : cnext;
            if (cn >= 0) wide_load<NR>(xb, xs + (size_t)cn * NP, lane);
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int k = 0; k < NR; k += 2) { a0 = fma(xa[k], rr[k], a0); if (k + 1 < NR) a1 = fma(xa[k + 1], rr[k + 1], a1); }
            const double dot = wsum(a0 + a1);
            const double b0 = fresh ? 0.0 : bsh[c - jbeg];
            double u = dot * rn + d * b0;                            // ref src/oem_dense.h:520
            if (grp) {
                if (K.kind == K_SGL) u = soft1(u, pfsh[c - jbeg] * K.L1, 1.0);
                s2 += u * u;
                if (lane == 0) uw[c - c0] = u;
            } else {                                                 // element-wise operators: the column is done here, as in wide_cols_kernel
                const double tp = pfsh[c - jbeg] * K.L;
                double bn;
                if (K.kind == K_SOFT) bn = cdiv(shrink(u, tp), K.D, rD);
                else if (K.kind == K_MCP) {
                    const bool big = fabs(u) > gammad * tp;
                    bn = cdiv(big ? u : shrink(u, tp), big ? K.D : dmg, big ? rD : rdmg);
                } else if (K.kind == K_SCAD) {
                    const double au = fabs(u);
                    const bool big = au > gammad * tp, mid = !big && au > (K.D + 1.0) * tp;
                    const double num = big ? u : (mid ? shrink(gm1 * u, K.gamma * tp) : shrink(u, tp));
                    bn = cdiv(num, mid ? dsc : K.D, mid ? rdsc : rD);
                } else bn = cdiv(u, d, rd);
                const double cu = fabs(bn), qo = fabs(b0);
                const bool cnz = cu > 1e-13, qn = qo > 1e-13;
                moving |= (cnz != qn) || (cnz && qn && fabs(bn - b0) > A.tol * qo);
                if (lane == 0) beta[c] = bn;
                if (bn != 0.0) {
#pragma unroll
                    for (int k = 0; k < NR; ++k) rp[k] = fma(xa[k], bn, rp[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < NR; ++k) xa[k] = xb[k];
        }
        if (!grp) continue;
        // ---- the group's factor (the update kernel's arithmetic: path_update)
        double fct = 1.0;
        if (g < 0) fct = 0.0;
        else if (!A.gzero[g]) {
            const double s = sqrt(s2), pen_g = K.L * A.gw[g];
            if (K.kind == K_GRP || K.kind == K_SGL) { const double t = 1.0 - pen_g / s; fct = (0.0 < t) ? t : 0.0; }
            else if (K.kind == K_GRP_MCP) fct = mcp_norm(s, pen_g, K.D, K.gamma);
            else fct = scad_norm(s, pen_g, K.D, K.gamma);
        }
        // ---- pass 2: the members' coefficients; a column is read again only where its coefficient is not zero
        for (int c = c0; c < c1; ++c) {
            const double u = uw[c - c0], b0 = fresh ? 0.0 : bsh[c - jbeg];
            const double bn = (fct != 0.0) ? u * fct / K.D : 0.0;
            const double cu = fabs(bn), qo = fabs(b0);
            const bool cnz = cu > 1e-13, qn = qo > 1e-13;
            moving |= (cnz != qn) || (cnz && qn && fabs(bn - b0) > A.tol * qo);
            if (lane == 0) beta[c] = bn;
            if (bn != 0.0) {
                double xc[NR];
                wide_load<NR>(xc, xs + (size_t)c * NP, lane);
#pragma unroll
                for (int k = 0; k < NR; ++k) rp[k] = fma(xc[k], bn, rp[k]);
            }
        }
    }
    const int mv = __syncthreads_or(moving ? 1 : 0);
    if (tid == 0) flags[(par ^ 1) * FMAXB + blockIdx.x] = mv;
#pragma unroll
    for (int k = 0; k < NR; ++k) wsh[w * NP + lane + 64 * k] = rp[k];
    __syncthreads();
    for (int r = tid; r < NP; r += NT) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) t += wsh[ww * NP + r];
        P[(size_t)blockIdx.x * NP + r] = t;
    }
}

// out = Ys - sum_w P[w] (W_OEM: the residual) | sum / n (W_EIG: Xs Xs' v / n) | sum (W_XB: Xs beta).  64 rows per workgroup, sixteen
// interleaved chains over the workgroups' partial vectors (eight loads in flight per thread: a chain of W dependent-free loads
// issued one by one was the whole iteration's time), combined in a fixed order: bitwise reproducible.
template <int MODE>
__global__ __launch_bounds__(1024) void wide_reduce_kernel(const double *__restrict__ P, int W, long long npad, int n, const double *__restrict__ ysv,
                                                            double *__restrict__ out, const int *__restrict__ done)
{
    __shared__ double sh[16][64];
    if (done && *done) return;
    const int tid = threadIdx.x, l = tid & 63, part = tid >> 6;
    const long long i = (long long)blockIdx.x * 64 + l;
    sh[part][l] = chain_sum(P + i, part, W, 16, (size_t)npad);
    __syncthreads();
    if (part == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][l];
        out[i] = (MODE == W_OEM) ? ysv[i] - t : (MODE == W_EIG ? t * (1.0 / (double)n) : t);
    }
}

// g = sum over the row blocks of their g_b (block order)
__global__ __launch_bounds__(256) void wide_block_sum_kernel(const double *__restrict__ gb, int nb, int q, double *__restrict__ g, const int *__restrict__ done)
{
    if (done && *done) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= q) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += gb[(size_t)b * (q + 8) + j];
    g[j] = s;
}

// n > 2048: the rows in nb blocks of <= 2048 (a column of a block fits a wave's registers).  A dot product x_j . r now spans the
// blocks, so an iteration is the two products as passes of their own -- t = Xs beta block by block (W_XB + reduction), g = Xs' t / n
// as per-block partial g_b (W_XTV) added in block order -- around path_update_kernel: every penalty, accelerate, compute.loss; 16 n p
// bytes per iteration like the reference's two GEMVs.  The eigen step the same way on vectors of lay.rows() entries (the padding
// rows of Xs are zero, so the padded operator has the spectrum of Xs Xs'/n plus zeros).
template <int NR>
static int run_path_wide_blocks(hipStream_t s, const PathArgs &a, const WideArgs &wd, double *host_scratch)
{
    const int q = a.p, n = wd.n, W = wide_workgroups(n, q), cpw = (q + W - 1) / W, nb = wd.lay.nb;
    constexpr int NT = 64 * wide_nw(NR);
    const long long npad = 64 * NR, rows = wd.lay.rows();
    double *P = wd.scratch, *t = P + (size_t)W * npad + rows, *v = t + rows, *vp = v + rows, *w = vp + rows;
    double *T = w + rows;
    double *gb = T + 2 * MAXL + 64 + 16 + 2 + FMAXB + 64;
    LState *st = reinterpret_cast<LState *>(a.work);
    double *beta = a.work + STATE_DBL, *g = beta + (q + 8);
    OEM_HIP(hipMemsetAsync(a.work, 0, sizeof(double) * path_large_work_doubles(q, 0), s));
    OEM_HIP(hipMemsetAsync(wd.scratch, 0, sizeof(double) * wide_scratch_doubles(n, q), s));
    const size_t lds = sizeof(double) * ((size_t)wide_nw(NR) * (size_t)npad + 2 * (size_t)cpw);
    if (lds > 64 * 1024) {
        OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_cols_kernel<NR, W_XB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_cols_kernel<NR, W_XTV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int rblocks = (int)(npad / 64);
    auto xb = [&](double *coef, double *out, const int *done) {          // out (rows()) = Xs coef, block by block
        for (int b = 0; b < nb; ++b) {
            hipLaunchKernelGGL((wide_cols_kernel<NR, W_XB>), dim3(W), dim3(NT), lds, s, a, wd.xs + (size_t)b * npad * q, (const double *)nullptr, wd.ys, P, coef,
                               (double *)nullptr, (SState *)nullptr, (int *)nullptr, (int *)nullptr, done, 0, 0.0, n, cpw);
            hipLaunchKernelGGL((wide_reduce_kernel<W_XB>), dim3(rblocks), dim3(1024), 0, s, P, W, npad, n, wd.ys, out + (size_t)b * npad, done);
        }
    };
    auto xtv = [&](const double *vec, double *gout, const int *done) {   // gout (q) = Xs' vec / n
        for (int b = 0; b < nb; ++b)
            hipLaunchKernelGGL((wide_cols_kernel<NR, W_XTV>), dim3(W), dim3(NT), lds, s, a, wd.xs + (size_t)b * npad * q, vec + (size_t)b * npad, wd.ys, P,
                               (double *)nullptr, gb + (size_t)b * (q + 8), (SState *)nullptr, (int *)nullptr, (int *)nullptr, done, 0, 0.0, n, cpw);
        hipLaunchKernelGGL(wide_block_sum_kernel, dim3((q + 255) / 256), dim3(256), 0, s, gb, nb, q, gout, done);
    };
    // ---- d = 1.005 lambda_max(Xs Xs'/n)
    const int mmax = n < MAXL ? n : MAXL;
    double theta = 0.0, theta_prev = -1.0;
    bool lz_capped = true;
    double *hT = host_scratch;
    int m = 0;
    hipLaunchKernelGGL(lanczos_init_kernel, dim3(1), dim3(1024), 0, s, (int)rows, v, vp);
    while (m < mmax) {
        const int chunk = (mmax - m) < 16 ? (mmax - m) : 16;
        for (int k = 0; k < chunk; ++k, ++m) {
            xtv(v, g, nullptr);
            xb(g, w, nullptr);
            hipLaunchKernelGGL(lanczos_update_kernel, dim3(1), dim3(1024), 0, s, (int)rows, m, v, vp, w, T);
        }
        OEM_HIP(hipGetLastError());
        OEM_HIP(hipMemcpyAsync(hT, T, sizeof(double) * 2 * MAXL, hipMemcpyDeviceToHost, s));
        OEM_HIP(hipStreamSynchronize(s));
        int mm = m;
        for (int k = 0; k < m; ++k)
            if (!(hT[MAXL + k] > 1e-13 * std::fabs(hT[k]))) { mm = k + 1; break; }
        theta = tridiag_max_host(hT, hT + MAXL, mm);
        if (mm < m) { lz_capped = false; break; }
        if (m >= 24) {
            const double t1 = tridiag_max_host(hT, hT + MAXL, m - 8), t0 = tridiag_max_host(hT, hT + MAXL, m - 16);
            const double mv = theta - t1, mvp = t1 - t0, ath = std::fabs(theta);
            if (mv <= 1e-14 * ath || (mv < 0.01 * mvp && mv * mv <= OEM_LANCZOS_TAIL_TOL * ath * (mvp - mv))) { lz_capped = false; break; }
        }
        if (theta_prev > 0 && std::fabs(theta - theta_prev) <= 1e-12 * std::fabs(theta)) { lz_capped = false; break; }
        theta_prev = theta;
    }
    if (mmax >= n) lz_capped = false;
    const double d = theta * 1.005;                     // ref src/oem_dense.h:498
    OEM_HIP(hipMemsetAsync(g, 0, sizeof(double) * (size_t)(q + 8), s));
    hipLaunchKernelGGL(path_init_kernel, dim3(1), dim3(1024), 0, s, a, st, beta, d, theta, m, lz_capped ? 1 : 0);
    OEM_HIP(hipGetLastError());
    if (a.npen == 0) return 0;
    double *uf = nullptr;
    const size_t shu = update_lds(a, &uf);
    if (shu > 64 * 1024) OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&path_update_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shu));
    auto enq = [&](int count) {
        for (int k = 0; k < count; ++k) {
            xb(beta, t, (const int *)&st->done);
            xtv(t, g, (const int *)&st->done);
            hipLaunchKernelGGL(path_update_kernel<0>, dim3(1), dim3(1024), shu, s, a, st, beta, g, uf, (int *)nullptr);
        }
    };
    const int FB = 16;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        enq(FB);
        if (hipStreamEndCapture(s, &graph) != hipSuccess || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            graph = nullptr; exec = nullptr;
            (void)hipGetLastError();
        }
    } else (void)hipGetLastError();
    const long long max_it = (long long)a.npen * a.nl * ((long long)a.maxit + 3) + 8;
    long long launched = 0;
    int *hdone = reinterpret_cast<int *>(host_scratch);
    int rc = 0;
    for (;;) {
        if (exec) { if (hipGraphLaunch(exec, s) != hipSuccess) { set_error("hipGraphLaunch failed"); rc = OEMGPU_ERR_HIP; break; } }
        else enq(FB);
        if (hipGetLastError() != hipSuccess) { set_error("wide engine: launch failed"); rc = OEMGPU_ERR_HIP; break; }
        launched += FB;
        if (hipMemcpyAsync(hdone, &st->done, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { set_error("wide engine: device error"); rc = OEMGPU_ERR_HIP; break; }
        if (*hdone) break;
        if (caller_interrupted()) { set_error("interrupted by the caller"); rc = OEMGPU_ERR_INTERRUPTED; break; }
        if (launched > max_it) { set_error("wide engine did not finish within %lld iterations", max_it); rc = OEMGPU_ERR_INTERNAL; break; }
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
}

template <int NR>
static int run_path_wide_nr(hipStream_t s, const PathArgs &a, const WideArgs &wd, double *host_scratch)
{    if (wd.lay.nb > 1) return run_path_wide_blocks<NR>(s, a, wd, host_scratch);

    const int q = a.p, n = wd.n, W = wide_workgroups(n, q), cpw = (q + W - 1) / W;
    constexpr int NT = 64 * wide_nw(NR);
    const long long npad = 64 * NR;
    if (wd.lay.npad() != npad) { set_error("internal: wide engine padding"); return OEMGPU_ERR_INTERNAL; }
    double *P = wd.scratch, *r = P + (size_t)W * npad, *t = r + npad, *v = t + npad, *vp = v + npad, *w = vp + npad;
    double *T = w + npad;
    SState *SS = reinterpret_cast<SState *>(T + 2 * MAXL + 64);
    int *fdone = reinterpret_cast<int *>(T + 2 * MAXL + 64 + 16);
    int *flags = reinterpret_cast<int *>(T + 2 * MAXL + 64 + 16 + 2);
    LState *st = reinterpret_cast<LState *>(a.work);
    double *beta = a.work + STATE_DBL, *g = beta + (q + 8);
    OEM_HIP(hipMemsetAsync(a.work, 0, sizeof(double) * path_large_work_doubles(q, 0), s));
    OEM_HIP(hipMemsetAsync(wd.scratch, 0, sizeof(double) * wide_scratch_doubles(n, q), s));
    const size_t lds = sizeof(double) * ((size_t)wide_nw(NR) * (size_t)npad + 2 * (size_t)cpw);
#define OEM_WIDE_ATTR(MODE)                                                                                                       \
    if (lds > 64 * 1024) OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_cols_kernel<NR, MODE>),               \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
    OEM_WIDE_ATTR(W_EIG); OEM_WIDE_ATTR(W_OEM); OEM_WIDE_ATTR(W_XB); OEM_WIDE_ATTR(W_XTV);
#undef OEM_WIDE_ATTR
    const int rblocks = (int)(npad / 64);
    // ---- d = 1.005 lambda_max(Xs Xs'/n): Lanczos on n-vectors, the product as the two passes of ONE kernel (ref :476-498)
    const int mmax = n < MAXL ? n : MAXL;
    double theta = 0.0, theta_prev = -1.0;
    bool lz_capped = true;
    double *hT = host_scratch;
    int m = 0;
    hipLaunchKernelGGL(lanczos_init_kernel, dim3(1), dim3(1024), 0, s, n, v, vp);
    while (m < mmax) {
        const int chunk = (mmax - m) < 16 ? (mmax - m) : 16;
        for (int k = 0; k < chunk; ++k, ++m) {
            hipLaunchKernelGGL((wide_cols_kernel<NR, W_EIG>), dim3(W), dim3(NT), lds, s, a, wd.xs, v, wd.ys, P, (double *)nullptr, (double *)nullptr,
                               (SState *)nullptr, (int *)nullptr, (int *)nullptr, (const int *)nullptr, 0, 0.0, n, cpw);
            hipLaunchKernelGGL((wide_reduce_kernel<W_EIG>), dim3(rblocks), dim3(1024), 0, s, P, W, npad, n, wd.ys, w, (const int *)nullptr);
            hipLaunchKernelGGL(lanczos_update_kernel, dim3(1), dim3(1024), 0, s, n, m, v, vp, w, T);
        }
        OEM_HIP(hipGetLastError());
        OEM_HIP(hipMemcpyAsync(hT, T, sizeof(double) * 2 * MAXL, hipMemcpyDeviceToHost, s));
        OEM_HIP(hipStreamSynchronize(s));
        int mm = m;
        for (int k = 0; k < m; ++k)
            if (!(hT[MAXL + k] > 1e-13 * std::fabs(hT[k]))) { mm = k + 1; break; }     // breakdown: T is exact
        theta = tridiag_max_host(hT, hT + MAXL, mm);
        if (mm < m) { lz_capped = false; break; }
        if (m >= 24) {                                                  // the stop rule of run_path_large
            const double t1 = tridiag_max_host(hT, hT + MAXL, m - 8), t0 = tridiag_max_host(hT, hT + MAXL, m - 16);
            const double mv = theta - t1, mvp = t1 - t0, ath = std::fabs(theta);
            if (mv <= 1e-14 * ath || (mv < 0.01 * mvp && mv * mv <= OEM_LANCZOS_TAIL_TOL * ath * (mvp - mv))) { lz_capped = false; break; }
        }
        if (theta_prev > 0 && std::fabs(theta - theta_prev) <= 1e-12 * std::fabs(theta)) { lz_capped = false; break; }
        theta_prev = theta;
    }
    if (mmax >= n) lz_capped = false;
    const double d = theta * 1.005;                     // ref src/oem_dense.h:498
    hipLaunchKernelGGL(path_init_kernel, dim3(1), dim3(1024), 0, s, a, st, beta, d, theta, m, lz_capped ? 1 : 0);
    OEM_HIP(hipGetLastError());
    if (a.npen == 0) return 0;
    const bool fused = a.ngroups == 0 && !a.accelerate && !a.compute_loss && !a.sinv && W <= FMAXB;
    // group penalties whose groups are runs of neighbouring columns: the fused GROUP form (wide_groups_kernel), one launch + the reduction
    const bool gfused = !fused && a.ngroups > 0 && wd.grun_W > 0 && wd.grun_W <= W && wd.grun_W <= FMAXB && !a.accelerate && !a.compute_loss && !a.sinv &&
                        !sw().OEM_WIDE_NO_GROUP_FUSED.set;
    const size_t ldsg = sizeof(double) * ((size_t)wide_nw(NR) * (size_t)npad + 2 * (size_t)wd.grun_cpw + (size_t)wide_nw(NR) * WIDE_GRUN_MAX);
    if (gfused && ldsg > 64 * 1024) OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_groups_kernel<NR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg));
    double *uf = nullptr;
    const size_t shu = update_lds(a, &uf);                           // U[q] | F[ngroups]: group operators only
    if (!fused && shu > 64 * 1024) OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&path_update_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shu));
    if (fused || gfused) hipLaunchKernelGGL(sym_init_kernel, dim3(1), dim3(1), 0, s, SS, a);
    auto enq = [&](int count) {
        for (int k = 0; k < count; ++k) {
            if (gfused) {
                hipLaunchKernelGGL((wide_groups_kernel<NR>), dim3(wd.grun_W), dim3(NT), ldsg, s, a, wd.xs, r, wd.ys, P, beta, SS, flags, fdone, k & 1, d, n,
                                   wd.grun_cpw, wd.grun_start, wd.grun_gid, wd.grun_wg);
                hipLaunchKernelGGL((wide_reduce_kernel<W_OEM>), dim3(rblocks), dim3(1024), 0, s, P, wd.grun_W, npad, n, wd.ys, r, (const int *)fdone);
            } else if (fused) {
                hipLaunchKernelGGL((wide_cols_kernel<NR, W_OEM>), dim3(W), dim3(NT), lds, s, a, wd.xs, r, wd.ys, P, beta, (double *)nullptr, SS, flags,
                                   fdone, (const int *)nullptr, k & 1, d, n, cpw);
                hipLaunchKernelGGL((wide_reduce_kernel<W_OEM>), dim3(rblocks), dim3(1024), 0, s, P, W, npad, n, wd.ys, r, (const int *)fdone);
            } else {
                hipLaunchKernelGGL((wide_cols_kernel<NR, W_XB>), dim3(W), dim3(NT), lds, s, a, wd.xs, (const double *)nullptr, wd.ys, P, beta,
                                   (double *)nullptr, (SState *)nullptr, (int *)nullptr, (int *)nullptr, (const int *)&st->done, 0, d, n, cpw);
                hipLaunchKernelGGL((wide_reduce_kernel<W_XB>), dim3(rblocks), dim3(1024), 0, s, P, W, npad, n, wd.ys, t, (const int *)&st->done);
                hipLaunchKernelGGL((wide_cols_kernel<NR, W_XTV>), dim3(W), dim3(NT), lds, s, a, wd.xs, t, wd.ys, P, (double *)nullptr, g,
                                   (SState *)nullptr, (int *)nullptr, (int *)nullptr, (const int *)&st->done, 0, d, n, cpw);
                hipLaunchKernelGGL(path_update_kernel<0>, dim3(1), dim3(1024), shu, s, a, st, beta, g, uf, (int *)nullptr);
            }
        }
    };
    const int FB = (fused || gfused) ? 64 : 32;                      // even: every batch starts at parity 0
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        enq(FB);
        if (hipStreamEndCapture(s, &graph) != hipSuccess || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            graph = nullptr; exec = nullptr;
            (void)hipGetLastError();
        }
    } else (void)hipGetLastError();
    const long long max_it = (long long)a.npen * a.nl * ((long long)a.maxit + 3) + 8;
    long long launched = 0;
    int *hdone = reinterpret_cast<int *>(host_scratch);
    int rc = 0;
    for (;;) {
        if (exec) { if (hipGraphLaunch(exec, s) != hipSuccess) { set_error("hipGraphLaunch failed"); rc = OEMGPU_ERR_HIP; break; } }
        else enq(FB);
        if (hipGetLastError() != hipSuccess) { set_error("wide engine: launch failed"); rc = OEMGPU_ERR_HIP; break; }
        launched += FB;
        if (hipMemcpyAsync(hdone, (fused || gfused) ? fdone : &st->done, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { set_error("wide engine: device error"); rc = OEMGPU_ERR_HIP; break; }
        if (*hdone) break;
        if (caller_interrupted()) { set_error("interrupted by the caller"); rc = OEMGPU_ERR_INTERRUPTED; break; }
        if (launched > max_it) { set_error("wide engine did not finish within %lld iterations", max_it); rc = OEMGPU_ERR_INTERNAL; break; }
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
}

int run_path_wide(hipStream_t s, const PathArgs &a, const WideArgs &wd, double *host_scratch)
{
    switch (wd.lay.nr) {
    case 1: return run_path_wide_nr<1>(s, a, wd, host_scratch);
    case 2: return run_path_wide_nr<2>(s, a, wd, host_scratch);
    case 3: return run_path_wide_nr<3>(s, a, wd, host_scratch);
    case 4: return run_path_wide_nr<4>(s, a, wd, host_scratch);
    case 6: return run_path_wide_nr<6>(s, a, wd, host_scratch);
    case 8: return run_path_wide_nr<8>(s, a, wd, host_scratch);
    case 12: return run_path_wide_nr<12>(s, a, wd, host_scratch);
    case 16: return run_path_wide_nr<16>(s, a, wd, host_scratch);
    case 24: return run_path_wide_nr<24>(s, a, wd, host_scratch);
    case 32: return run_path_wide_nr<32>(s, a, wd, host_scratch);
    default: break;
    }
    set_error("p >= n: the wide engine holds n <= %d rows", WIDE_MAX_N);
    return OEMGPU_ERR_UNSUPPORTED;
}


}  // namespace oemgpu
