// xval.hip -- the device side of xval.oem (ref src/oem_xval_dense.{h,cpp}).
//
// The reference builds one Gram per fold from a gathered copy of the fold's rows (ref src/oem_xval_dense.h:358-484), fits
// K+1 models on sums of them (:733-853) and then walks the observations once more for the cross-validation error
// (ref src/oem_xval_dense.cpp:343-461).  Here:
//   * fold layout: rows are permuted ONCE into fold-contiguous order (stable inside a fold, so every run sums in the same
//     order): block histograms -> per-fold scan -> stable ranks by wave ballots -> a gather pass (coalesced reads, K write
//     streams).  Fold segments start on multiples of 16 rows, so each one meets the alignment the MFMA Gram kernels want;
//   * per-fold moments: the one-pass moment kernels of gram.hip on each segment (launched from api.hip);
//   * leave-one-fold-out sums: fold_sum_kernel, in fold order like the reference;
//   * CV error: predictions of a fold's rows under that fold's K-th fit for ALL lambdas are an (n_k x p) x (p x nlambda)
//     product -- as many flops as the Gram build -- so it runs on the FP64 MFMA pipe: a 16-row tile of X against 16-lambda
//     tiles of the coefficient matrix held in LDS, the intercept riding along as one more column of ones.  The epilogue
//     turns the accumulators into squared / absolute residuals and keeps per-lambda sums of the error and its square;
//     workgroup partials are combined in a fixed order.
#include "common.hpp"

namespace oemgpu {
namespace {

constexpr int FB = 1024;                 // rows per layout block

// ---------------------------------------------------------------------------------------- fold layout
__global__ __launch_bounds__(FB) void fold_count_kernel(const int *__restrict__ foldid, int64_t n, int K,
                                                        int *__restrict__ blockcnt, int *__restrict__ bad)
{
    extern __shared__ int hist[];
    for (int k = threadIdx.x; k < K; k += FB) hist[k] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x;
    if (i < n) {
        const int f = foldid[i];
        if (f < 1 || f > K) atomicOr(bad, 1);
        else atomicAdd(&hist[f - 1], 1);              // integer: the order of the adds does not matter
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += FB) blockcnt[(size_t)blockIdx.x * K + k] = hist[k];
}

// one wave per fold: exclusive prefix of the fold's block counts, 64 blocks per step
__global__ __launch_bounds__(64) void fold_scan_kernel(int *__restrict__ blockcnt, int nblk, int K, int64_t *__restrict__ fold_n)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    int64_t run = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int b = b0 + lane;
        const int c = b < nblk ? blockcnt[(size_t)b * K + k] : 0;
        int incl = c;
        for (int s = 1; s < 64; s <<= 1) { const int t = __shfl_up(incl, s, 64); if (lane >= s) incl += t; }
        if (b < nblk) blockcnt[(size_t)b * K + k] = (int)(run + incl - c);
        run += __shfl(incl, 63, 64);
    }
    if (lane == 0) fold_n[k] = run;
}

__global__ void fold_start_kernel(const int64_t *__restrict__ fold_n, int K, int64_t *__restrict__ fold_start)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t s = 0;
        for (int k = 0; k < K; ++k) { fold_start[k] = s; s += (fold_n[k] + 15) / 16 * 16; }
    }
}

__global__ __launch_bounds__(FB) void fold_pos_kernel(const int *__restrict__ foldid, int64_t n, int K,
                                                      const int *__restrict__ blockoff, const int64_t *__restrict__ fold_start,
                                                      int *__restrict__ pos)
{
    extern __shared__ int wcnt[];                     // [16][K]
    const int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int f = -1;
    if (i < n) { f = foldid[i] - 1; if (f < 0 || f >= K) f = -1; }
    int rank = 0;
    for (int k = 0; k < K; ++k) {
        const unsigned long long m = __ballot(f == k);
        if (lane == 0) wcnt[w * K + k] = __popcll(m);
        if (f == k) rank = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (f >= 0) {
        int pre = 0;
        for (int ww = 0; ww < w; ++ww) pre += wcnt[ww * K + f];
        pos[i] = (int)(fold_start[f] + blockoff[(size_t)blockIdx.x * K + f] + pre + rank);
    }
}

// rows into fold-contiguous order: reads coalesced along the rows of a column, writes in K streams
constexpr int GCB = 8;                   // columns per gather block
__global__ __launch_bounds__(256) void gather_rows_kernel(const double *__restrict__ x, int64_t n, int64_t ld, int p,
                                                          const double *__restrict__ y, const int *__restrict__ pos,
                                                          double *__restrict__ xo, int64_t ldo, double *__restrict__ yo)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t d = pos[i];
    const int c0 = blockIdx.y * GCB;
#pragma unroll
    for (int c = 0; c < GCB; ++c)
        if (c0 + c < p) xo[(size_t)(c0 + c) * ldo + d] = x[(size_t)(c0 + c) * ld + i];
    if (blockIdx.y == 0) yo[d] = y[i];
}

// out = sum of the fold moments except fold `skip` (1-based; 0: none), in fold order (ref src/oem_xval_dense.h:733-742, 801-811)
__global__ __launch_bounds__(256) void fold_sum_kernel(const double *__restrict__ M, int K, size_t len, int skip, double *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= len) return;
    double s = 0.0;
    for (int k = 1; k <= K; ++k) if (k != skip) s += M[(size_t)(k - 1) * len + t];
    out[t] = s;
}

// ---------------------------------------------------------------------------------------- observation weights
// xp: the fold-ordered copy with p + 1 columns, column 0 = w (gathered), columns 1..p = x (gathered); yp = y (gathered).
// Workgroup (j, k): column j (1..p: x_j; p + 1: y) over the rows of fold k -- the UNWEIGHTED sum of squares of an x column (the
// reference does not standardise with respect to the weights, ref src/oem_xval_dense.h:533-535, 614) in a fixed order, then the
// column times sqrt(w), in place.  csq: [K][p + 1], entry p of a fold = its number of rows (filled by the y workgroup).
__global__ __launch_bounds__(256) void weight_scale_kernel(double *__restrict__ xp, int64_t ldp, double *__restrict__ yp, int p,
                                                           const int64_t *__restrict__ fold_start, const int64_t *__restrict__ fold_n,
                                                           double *__restrict__ csq)
{
    __shared__ double sh[256];
    const int j = blockIdx.x + 1, k = blockIdx.y, tid = threadIdx.x;
    const int64_t st = fold_start[k], nk = fold_n[k];
    double *col = (j <= p) ? xp + (size_t)j * ldp + st : yp + st;
    const double *wv = xp + st;
    double s = 0.0;
    for (int64_t r = tid; r < nk; r += 256) {
        const double v = col[r];
        s = fma(v, v, s);
        col[r] = v * sqrt(wv[r]);
    }
    sh[tid] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if (tid < h) sh[tid] += sh[tid + h]; __syncthreads(); }
    if (tid == 0) csq[(size_t)k * (p + 1) + (j <= p ? j - 1 : p)] = (j <= p) ? sh[0] : (double)nk;
}
// column 0 becomes sqrt(w) once every other column has used w
__global__ __launch_bounds__(256) void weight_sqrt_kernel(double *__restrict__ xp, const int64_t *__restrict__ fold_start,
                                                          const int64_t *__restrict__ fold_n)
{
    const int k = blockIdx.y;
    const int64_t st = fold_start[k], nk = fold_n[k];
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nk; r += (int64_t)gridDim.x * 256) xp[st + r] = sqrt(xp[st + r]);
}

// Weighted fold sums -> XX, XY of oemXvalDense (ref src/oem_xval_dense.h:733-784 with XtWX_xval(_int), :486-623): Mw = moments of
// the p + 1 data columns [sqrt(w) | sqrt(w) X] and of sqrt(w) y ((p + 3)^2, lower triangle), cs = [sum x_j^2 (unweighted), rows].
// q = p + intercept.  stats as launch_finalize's (colsq_inv in the scale slots).
__global__ __launch_bounds__(256) void finalize_weighted_kernel(const double *__restrict__ Mw, const double *__restrict__ cs, int p,
                                                                int standardize, int intercept, double *__restrict__ xx,
                                                                double *__restrict__ xy, double *__restrict__ stats)
{
    const int q = p + (intercept ? 1 : 0), off = intercept ? 0 : 1, qm = p + 3;
    const double nobs = cs[p];
    auto cinv = [&](int c) -> double {                 // data column c of [sqrt(w) | X]: 1 for the intercept column
        if (c == 0 || !standardize) return 1.0;
        double v = cs[c - 1] / (nobs - 1.0);
        if (v == 0.0) v = 1.0;
        return 1.0 / sqrt(v);
    };
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int t = tid; t < q * q; t += nth) {
        const int a = t % q + off, b = t / q + off;    // data columns
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        xx[t] = cinv(a) * Mw[(size_t)lo * qm + hi] * cinv(b) / nobs;
    }
    for (int t = tid; t < q; t += nth) xy[t] = cinv(t + off) * Mw[(size_t)(t + off) * qm + (p + 1)] / nobs;
    if (tid == 0) {
        stats[0] = 0.0; stats[1] = 1.0; stats[2] = Mw[(size_t)(p + 1) * qm + (p + 1)]; stats[3] = nobs;
        stats[stats_shift_flag(p)] = 0.0; stats[stats_shift_flag(p) + 1] = 0.0;
    }
    for (int t = tid; t < p; t += nth) { stats[4 + t] = 0.0; stats[4 + p + t] = cinv(t + 1); }
}

// ---------------------------------------------------------------------------------------- CV error
typedef double d4 __attribute__((ext_vector_type(4)));

// grid (workgroups per fold, K folds, npen); 512 threads = 2 waves per SIMD sharing one copy of the coefficients in LDS.
// B: [K][npen][nl][p + 1], row 0 of each column the intercept.  part: [K * gridDim.x * CVW][npen][nl16][4] per WAVE: rows, centre c, sum (v - c), sum (v - c)^2 -- merged by Chan's formula in cv_finish_kernel
// KC: k-steps (4 columns each) whose X fragments a lane holds at once (SINGLE: p + 1 <= 4 KC, one chunk per row tile).
// CHUNK (the coefficient tile of all p + 1 rows does not fit LDS beside LT lambda tiles -- p beyond ~160 at 100 lambdas): LDS holds KCH
// coefficient rows at a time; the workgroup's eight waves take one 16-row tile of X each ("round"), keep its LT accumulator tiles
// over the chunks, and the chunk is staged again (from L2: the K x npen x nl x (p + 1) coefficients are a few MB) for every round.
// X is still read ONCE per pass over the lambdas, and p has no limit: the traffic added is 16 LT / 128 bytes of L2 per byte of X.
constexpr int CVW = 8;                   // waves per workgroup
constexpr int CV_KCH = 112;              // coefficient rows per LDS chunk (CHUNK): two fragment loads of KC = 14 k-steps
template <int LT, int KC, bool SINGLE, bool CHUNK>
__global__ __launch_bounds__(64 * CVW) void cv_error_kernel(const double *__restrict__ xp, int64_t ldp, const double *__restrict__ yp,
                                                            const int64_t *__restrict__ fold_start, const int64_t *__restrict__ fold_n,
                                                            int p, const double *__restrict__ B, int nl, int mae, int wmode,
                                                            double *__restrict__ part)
{
    // wmode (observation weights): xp has p + 1 data columns -- column 0 = sqrt(w), columns 1..p = sqrt(w) x -- and yp = sqrt(w) y, so
    // the squared residual of the scaled row IS w (y - yhat)^2 (ref src/oem_xval_dense.cpp:389-437); |.| takes one more sqrt(w).
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int k = blockIdx.y, pen = blockIdx.z, npen = gridDim.z, nwg = gridDim.x, wg = blockIdx.x;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int Kd = p + 1, K4 = (Kd + 3) & ~3, nl16 = (nl + 15) & ~15, ntile = nl16 >> 4;
    constexpr int LW = 16 * LT;                       // lambdas per pass
    double *Bl = lds;                                 // [K4][LW]
    const int64_t start = fold_start[k], nk = fold_n[k];
    const double *Bsrc = B + ((size_t)k * npen + pen) * nl * Kd;
    const double *xk = xp + start, *yk = yp + start;
    double *dst = part + ((((size_t)k * nwg + wg) * CVW + w) * npen + pen) * nl16 * 4;      // this wave's partial
    const int64_t stride = (int64_t)nwg * CVW;

    auto load_a = [&](double (&a)[KC], int64_t rt, int c0) {
        const int64_t arow = rt * 16 + l16;
        const bool avalid = arow < nk;
#pragma unroll
        for (int s_ = 0; s_ < KC; ++s_) {
            const int kk = c0 + 4 * s_ + g;
            double v = (!wmode && kk == p) ? 1.0 : 0.0;
            if (avalid && (wmode ? kk <= p : kk < p)) v = xk[(size_t)kk * ldp + arow];
            a[s_] = v;
        }
    };

    // coefficient rows [cc, cc + ncols) of the lambda tiles from l0 on -> Bl[ncols][LW]
    auto stage = [&](int l0, int cc, int ncols) {
        for (int idx = tid; idx < ncols * LW; idx += 64 * CVW) {
            const int c = cc + idx / LW, j = idx % LW, lam = l0 * 16 + j;
            double v = 0.0;
            if (lam < nl) {
                if (wmode) { if (c <= p) v = Bsrc[(size_t)lam * Kd + c]; }      // column 0 (sqrt(w)) meets the intercept
                else if (c < p) v = Bsrc[(size_t)lam * Kd + c + 1];
                else if (c == p) v = Bsrc[(size_t)lam * Kd];          // intercept: the column of ones
            }
            Bl[idx] = v;
        }
    };

    for (int l0 = 0; l0 < ntile; l0 += LT) {
        if constexpr (!CHUNK) {
            __syncthreads();
            stage(l0, 0, K4);
            __syncthreads();
        }
        // The reference runs Welford's update over the observations (ref src/oem_xval_dense.cpp:420-422,452-461).  Here every wave
        // accumulates sum (v - c) and sum (v - c)^2 about a centre c of its own -- the error of the FIRST row it meets, per lambda --
        // so nothing cancels however small the spread of the errors is next to their mean, and the wave partials
        // (rows, c, sums) are merged pairwise with Chan's formula.
        double s1[LT], s2[LT], cen[LT];
        double cnt = 0.0;
        bool have_c = false;
#pragma unroll
        for (int t = 0; t < LT; ++t) { s1[t] = 0.0; s2[t] = 0.0; cen[t] = 0.0; }
        d4 acc[LT];
        auto clear = [&]() {
#pragma unroll
            for (int t = 0; t < LT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        };
        auto mac = [&](const double (&a)[KC], int c0, int cc) {       // cc: the coefficient row Bl starts at
#pragma unroll
            for (int s_ = 0; s_ < KC; ++s_) {
                const int kk = c0 + 4 * s_ + g;
                if (SINGLE || c0 + 4 * s_ < K4) {
                    const double *bp = Bl + (size_t)(kk < K4 ? kk - cc : 0) * LW + l16;
                    const double av = (kk < K4) ? a[s_] : 0.0;
#pragma unroll
                    for (int t = 0; t < LT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[16 * t], acc[t], 0, 0, 0);
                }
            }
        };
        // accumulator layout: register r of lane (g, l16) is (row 4 r + g, lambda l16) of the tile
        auto err_of = [&](double yv, double a, double sw) { const double res = yv - a; return mae ? fabs(res) * sw : res * res; };
        auto finish = [&](int64_t rt) {
            if (!have_c) {                                      // the tile's first row always exists (rt * 16 < nk): lanes g = 0, r = 0 hold it
                const double y0 = yk[rt * 16], sw0 = (wmode && mae) ? xk[rt * 16] : 1.0;
#pragma unroll
                for (int t = 0; t < LT; ++t) cen[t] = __shfl(err_of(y0, acc[t][0], sw0), l16, 64);
                have_c = true;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = rt * 16 + 4 * r + g;
                const bool ok = row < nk;
                const double yv = ok ? yk[row] : 0.0;
                const double sw = (ok && wmode && mae) ? xk[row] : 1.0;
                if (l16 == 0) cnt += ok ? 1.0 : 0.0;
#pragma unroll
                for (int t = 0; t < LT; ++t) {
                    const double dv = ok ? err_of(yv, acc[t][r], sw) - cen[t] : 0.0;
                    s1[t] += dv; s2[t] = fma(dv, dv, s2[t]);
                }
            }
        };
        // all fragments of a chunk are requested before its first MFMA; the second wave of the SIMD covers the wait
        if constexpr (!CHUNK) {
            for (int64_t rt = (int64_t)wg * CVW + w; rt * 16 < nk; rt += stride) {
                clear();
                for (int c0 = 0; c0 < K4; c0 += 4 * KC) {
                    double a[KC];
                    load_a(a, rt, c0);
                    mac(a, c0, 0);
                }
                finish(rt);
            }
        } else {
            // rounds: wave w of the workgroup takes row tile r0 + w; the barriers of the staging are met by all eight waves
            for (int64_t r0 = (int64_t)wg * CVW; r0 * 16 < nk; r0 += stride) {
                const int64_t rt = r0 + w;
                const bool valid = rt * 16 < nk;
                clear();
                for (int cc = 0; cc < K4; cc += CV_KCH) {
                    const int ncols = K4 - cc < CV_KCH ? K4 - cc : CV_KCH;
                    double a[KC];
                    if (valid) load_a(a, rt, cc);                       // in flight while the chunk is staged
                    __syncthreads();
                    stage(l0, cc, ncols);
                    __syncthreads();
                    if (valid) {
                        mac(a, cc, cc);
                        for (int c0 = cc + 4 * KC; c0 < cc + ncols; c0 += 4 * KC) { load_a(a, rt, c0); mac(a, c0, cc); }
                    }
                }
                if (valid) finish(rt);
            }
        }
        // the four row groups of a wave (fixed order); every wave leaves its own partial
        cnt += __shfl_xor(cnt, 16, 64); cnt += __shfl_xor(cnt, 32, 64);
        const double rows = __shfl(cnt, 0, 64);
#pragma unroll
        for (int t = 0; t < LT; ++t) {
            s1[t] += __shfl_xor(s1[t], 16, 64); s1[t] += __shfl_xor(s1[t], 32, 64);
            s2[t] += __shfl_xor(s2[t], 16, 64); s2[t] += __shfl_xor(s2[t], 32, 64);
            const int lam = l0 * 16 + 16 * t + l16;
            if (g == 0 && lam < nl16) {
                double *q = dst + (size_t)lam * 4;
                q[0] = rows; q[1] = cen[t]; q[2] = s1[t]; q[3] = s2[t];
            }
        }
    }
}

// cvm = mean error, cvsd = sqrt(sample variance / n)  (ref src/oem_xval_dense.cpp:452-461).  One wave per (penalty, lambda):
// lanes stride over the workgroup partials, then a fixed butterfly -- reproducible.
// triples: out[npen][nl][3] <- (count, mean, M2) of the rows seen instead (row shards: the caller merges them, oemgpu_xval_merge).
__global__ __launch_bounds__(64) void cv_finish_kernel(const double *__restrict__ part, int nparts, int npen, int nl, double n,
                                                       double *__restrict__ out /* [npen][nl][2] */, int triples)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const int pen = t / nl, lam = t - pen * nl, nl16 = (nl + 15) & ~15;
    // (rows, mean, M2 = sum (v - mean)^2) of a set of observations; two sets merge by Chan, Golub & LeVeque's update
    double na = 0.0, ma = 0.0, qa = 0.0;
    auto merge = [&](double nb, double mb, double qb) {
        if (nb > 0.0) {
            if (na > 0.0) {
                const double nn = na + nb, dl = mb - ma;
                ma += dl * (nb / nn);
                qa += qb + dl * dl * (na * nb / nn);
                na = nn;
            } else { na = nb; ma = mb; qa = qb; }
        }
    };
    for (int b = lane; b < nparts; b += 64) {
        const double *q = part + (((size_t)b * npen + pen) * nl16 + lam) * 4;
        const double nb = q[0];
        if (nb > 0.0) { const double m1 = q[2] / nb; merge(nb, q[1] + m1, q[3] - q[2] * m1); }
    }
    for (int s_ = 32; s_ > 0; s_ >>= 1) {                       // fixed butterfly: reproducible
        const double nb = __shfl_xor(na, s_, 64), mb = __shfl_xor(ma, s_, 64), qb = __shfl_xor(qa, s_, 64);
        // both partners must end with the same numbers: merge in lane order (lower lane's set first)
        if (lane & s_) { const double n0 = na, m0 = ma, q0 = qa; na = nb; ma = mb; qa = qb; merge(n0, m0, q0); }
        else merge(nb, mb, qb);
    }
    if (lane == 0) {
        if (triples) { out[(size_t)t * 3] = na; out[(size_t)t * 3 + 1] = ma; out[(size_t)t * 3 + 2] = qa; }
        else {
            out[(size_t)t * 2] = ma;                            // all n observations are in: the mean
            out[(size_t)t * 2 + 1] = sqrt((qa < 0.0 ? 0.0 : qa) / (n - 1.0)) / sqrt(n);
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------- launchers
size_t fold_layout_ints(int64_t n, int K) { return (size_t)((n + FB - 1) / FB) * K + 8; }

int launch_fold_layout(hipStream_t s, const int *foldid, int64_t n, int K, int *blockcnt, int64_t *fold_n, int64_t *fold_start,
                       int *pos, int *bad)
{
    const int nblk = (int)((n + FB - 1) / FB);
    OEM_HIP(hipMemsetAsync(bad, 0, sizeof(int), s));
    hipLaunchKernelGGL(fold_count_kernel, dim3(nblk), dim3(FB), sizeof(int) * K, s, foldid, n, K, blockcnt, bad);
    hipLaunchKernelGGL(fold_scan_kernel, dim3(K), dim3(64), 0, s, blockcnt, nblk, K, fold_n);
    hipLaunchKernelGGL(fold_start_kernel, dim3(1), dim3(64), 0, s, fold_n, K, fold_start);
    hipLaunchKernelGGL(fold_pos_kernel, dim3(nblk), dim3(FB), sizeof(int) * 16 * K, s, foldid, n, K, blockcnt, fold_start, pos);
    OEM_HIP(hipGetLastError());
    return 0;
}

int launch_gather_rows(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, const int *pos,
                       double *xo, int64_t ldo, double *yo)
{
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256), (p + GCB - 1) / GCB), dim3(256), 0, s, x, n, ld, p,
                       y, pos, xo, ldo, yo);
    OEM_HIP(hipGetLastError());
    return 0;
}

int launch_weight_scale(hipStream_t s, double *xp, int64_t ldp, double *yp, int p, int K, const int64_t *fold_start, const int64_t *fold_n,
                        double *csq)
{
    hipLaunchKernelGGL(weight_scale_kernel, dim3(p + 1, K), dim3(256), 0, s, xp, ldp, yp, p, fold_start, fold_n, csq);
    hipLaunchKernelGGL(weight_sqrt_kernel, dim3(64, K), dim3(256), 0, s, xp, fold_start, fold_n);
    OEM_HIP(hipGetLastError());
    return 0;
}

int launch_finalize_weighted(hipStream_t s, const double *Mw, const double *cs, int p, int standardize, int intercept, double *xx,
                             double *xy, double *stats)
{
    const int q = p + (intercept ? 1 : 0);
    int blocks = (q * q + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(finalize_weighted_kernel, dim3(blocks), dim3(256), 0, s, Mw, cs, p, standardize, intercept, xx, xy, stats);
    OEM_HIP(hipGetLastError());
    return 0;
}

int launch_fold_sum(hipStream_t s, const double *M, int K, size_t len, int skip, double *out)
{
    hipLaunchKernelGGL(fold_sum_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, M, K, len, skip, out);
    OEM_HIP(hipGetLastError());
    return 0;
}

// workgroups per fold for the CV pass.  One workgroup fills a CU (LDS), and all of them carry the same work: the grid must not
// exceed the CU count by a few (260 workgroups on 256 CUs take two rounds), so round DOWN.
int cv_wg_per_fold(int64_t n, int K, int npen, int num_cu)
{
    int nwg = num_cu / (K * npen);
    const int64_t tiles = (n / K + 16 * CVW - 1) / (16 * CVW);
    if (nwg > tiles) nwg = (int)tiles;
    return nwg < 1 ? 1 : nwg;
}
size_t cv_part_doubles(int nwg, int K, int npen, int nl) { return (size_t)nwg * K * CVW * npen * ((nl + 15) & ~15) * 4; }

template <int LT>
static int launch_cv_lt(hipStream_t s, dim3 grid, size_t lds, int ksteps, bool chunk, const double *xp, int64_t ldp, const double *yp,
                        const int64_t *fold_start, const int64_t *fold_n, int p, const double *B, int nl, int mae, int wmode, double *part)
{
#define OEM_CVK(KC, SINGLE, CHUNK)                                                                                                   \
    do {                                                                                                                             \
        OEM_HIP(hipFuncSetAttribute((const void *)cv_error_kernel<LT, KC, SINGLE, CHUNK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((cv_error_kernel<LT, KC, SINGLE, CHUNK>), grid, dim3(64 * CVW), lds, s, xp, ldp, yp, fold_start, fold_n, p, B, nl, mae, wmode, part); \
    } while (0)
    if (chunk) OEM_CVK(14, false, true);
    else if (ksteps <= 14) OEM_CVK(14, true, false);
    else OEM_CVK(14, false, false);   // 28 fragments at once spill next to 7 accumulator tiles
#undef OEM_CVK
    OEM_HIP(hipGetLastError());
    return 0;
}

int launch_cv_error(hipStream_t s, const double *xp, int64_t ldp, const double *yp, const int64_t *fold_start, const int64_t *fold_n,
                    int K, int p, const double *B, int npen, int nl, int mae, int wmode, int nwg, double n, double *part, double *out, bool triples)
{
    const int K4 = (p + 1 + 3) & ~3, ntile = (nl + 15) >> 4;
    // lambdas per pass: at most 7 16-wide tiles (accumulator registers), in even passes; all of them in one pass whenever there are
    // <= 112 lambdas, so that X is read once.  The coefficient tile of those lambdas stays in LDS for the whole pass when it fits
    // 140 KB; beyond that (p + 1 > ~160 at 100 lambdas) it goes through LDS in chunks of CV_KCH rows (CHUNK above) -- no limit on p
    // (this used to shrink the lambda tile instead, re-reading X up to seven times, and to refuse p > 1,183).
    int lt = ntile < 7 ? ntile : 7;
    if (ntile > lt) lt = (ntile + (ntile + lt - 1) / lt - 1) / ((ntile + lt - 1) / lt);      // even passes
    const bool chunk = (size_t)K4 * 16 * lt * sizeof(double) > 140 * 1024;
    const size_t lds = (size_t)(chunk ? CV_KCH : K4) * 16 * lt * sizeof(double);
    dim3 grid(nwg, K, npen);
    int rc;
    switch (lt) {
    case 1: rc = launch_cv_lt<1>(s, grid, lds, K4 / 4, chunk, xp, ldp, yp, fold_start, fold_n, p, B, nl, mae, wmode, part); break;
    case 2: rc = launch_cv_lt<2>(s, grid, lds, K4 / 4, chunk, xp, ldp, yp, fold_start, fold_n, p, B, nl, mae, wmode, part); break;
    case 3: rc = launch_cv_lt<3>(s, grid, lds, K4 / 4, chunk, xp, ldp, yp, fold_start, fold_n, p, B, nl, mae, wmode, part); break;
    case 4: rc = launch_cv_lt<4>(s, grid, lds, K4 / 4, chunk, xp, ldp, yp, fold_start, fold_n, p, B, nl, mae, wmode, part); break;
    case 5: rc = launch_cv_lt<5>(s, grid, lds, K4 / 4, chunk, xp, ldp, yp, fold_start, fold_n, p, B, nl, mae, wmode, part); break;
    case 6: rc = launch_cv_lt<6>(s, grid, lds, K4 / 4, chunk, xp, ldp, yp, fold_start, fold_n, p, B, nl, mae, wmode, part); break;
    case 7: rc = launch_cv_lt<7>(s, grid, lds, K4 / 4, chunk, xp, ldp, yp, fold_start, fold_n, p, B, nl, mae, wmode, part); break;
    default: return OEMGPU_ERR_INTERNAL;
    }
    if (rc) return rc;
    hipLaunchKernelGGL(cv_finish_kernel, dim3(npen * nl), dim3(64), 0, s, part, nwg * K * CVW, npen, nl, n, out, triples ? 1 : 0);
    OEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace oemgpu
