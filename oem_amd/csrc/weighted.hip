// weighted.hip -- observation weights of oemDense (ref src/oem_dense.h:368-414, 466-483, 513-517, 699-707, 759-770; src/DataStd.h:94-202):
// what `.Call("oem_fit_dense", ..., weights_, ...)` computes.  The R front end stops with "weights not implemented yet"
// (R/oem.R:244), so nothing the package ships reaches this branch; it is built because the entry point takes the argument
// (SURVEY section 8 row f-3), as it is, inconsistencies included:
//   * DataStd::standardize(X, Y, wts): Y by sqrt(w)-weighted statistics in every flag; X by sd_n(x sqrt w) (flag 1), mean(x sqrt w)
//     (flag 2) and the UNWEIGHTED mean / norm (flag 3);
//   * XY = X'(Y w)/n, XX = X' diag(w) X / n: with Z = diag(sqrt w) Xs and yz = sqrt(w) Ys these are Z'yz/n and Z'Z/n -- the
//     ordinary moment pass (FP64 MFMA, gram.hip) over Z and yz with no further standardisation; get_loss = sum w (Ys - Xs beta)^2
//     = |yz - Z beta|^2, the engines' Gram identity on the same moments;
//   * nobs <= nvars: d from (sqrt(w) Xs)(sqrt(w) Xs)'/n -- the spectrum of Z'Z/n -- but next_u = Xs'((Ys - Xs beta) w^2)/n + d beta
//     (w SQUARED): the Gram form on Z2 = diag(w) Xs, with d and the lambda grid handed over from the first pass (api.hip).
// Two small kernels (one workgroup per column, two sweeps: the centred sums need the means) and the scaled copy; a rare path,
// not a tuned one.
#include "common.hpp"
#include "path_dev.hpp"

namespace oemgpu {

namespace {

__device__ __forceinline__ double wt_block_sum(double v, double *sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double s = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[w] = s;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ws: [0] meanY [1] scaleY [2 .. 2 + p) meanX [2 + p .. 2 + 2p) scaleX      (block j < p: column j; block p: y)
__global__ __launch_bounds__(256) void wstd_stats_kernel(const double *__restrict__ x, int64_t n, int64_t ld, int p, const double *__restrict__ y,
                                                         const double *__restrict__ w, int flag, double *__restrict__ ws)
{
    __shared__ double sh[4];
    const int j = blockIdx.x, tid = threadIdx.x;
    const bool isy = j == p;
    const double *v = isy ? y : x + (size_t)j * ld;
    double s1 = 0.0, u1 = 0.0;                                   // sum sqrt(w) v, sum v
    for (int64_t i = tid; i < n; i += 256) { const double t = v[i]; s1 = fma(sqrt(w[i]), t, s1); u1 += t; }
    s1 = wt_block_sum(s1, sh); u1 = wt_block_sum(u1, sh);
    const double msw = s1 / (double)n, mu = u1 / (double)n, rsn = 1.0 / sqrt((double)n);
    double mean = 0.0, scale = 1.0, c2 = 0.0;
    if (isy) {
        if (flag == 1) {                                         // scaleY = sd_n(Y sqrt w)
            for (int64_t i = tid; i < n; i += 256) { const double t = v[i] * sqrt(w[i]) - msw; c2 = fma(t, t, c2); }
            scale = sqrt(wt_block_sum(c2, sh)) / sqrt((double)n);
        } else if (flag >= 2) {                                  // meanY = mean(Y sqrt w); scaleY = |(Y - meanY) sqrt w| / sqrt n
            mean = msw;
            for (int64_t i = tid; i < n; i += 256) { const double t = (v[i] - mean) * sqrt(w[i]); c2 = fma(t, t, c2); }
            scale = sqrt(wt_block_sum(c2, sh)) * rsn;
        }
        if (tid == 0) { ws[0] = mean; ws[1] = scale; }
        return;
    }
    if (flag == 1) {                                             // scaleX = sd_n(x sqrt w), 0 -> 1
        for (int64_t i = tid; i < n; i += 256) { const double t = v[i] * sqrt(w[i]) - msw; c2 = fma(t, t, c2); }
        scale = sqrt(wt_block_sum(c2, sh)) / sqrt((double)n);
        if (scale == 0.0) scale = 1.0;
    } else if (flag == 2) mean = msw;                            // meanX = mean(x sqrt w)
    else if (flag == 3) {                                        // the UNWEIGHTED mean and norm (ref src/DataStd.h:160-196)
        mean = mu;
        for (int64_t i = tid; i < n; i += 256) { const double t = v[i] - mean; c2 = fma(t, t, c2); }
        scale = sqrt(wt_block_sum(c2, sh)) * rsn;
        if (scale == 0.0) scale = 1.0;
    }
    if (tid == 0) { ws[2 + j] = mean; ws[2 + p + j] = scale; }
}

// z[i, j] = w_i^POW (x[i, j] - meanX_j) / scaleX_j ; yz[i] = w_i^POW (y_i - meanY) / scaleY      (POW = 1/2 or 1)
__global__ __launch_bounds__(256) void wstd_apply_kernel(const double *__restrict__ x, int64_t n, int64_t ld, int p, const double *__restrict__ y,
                                                         const double *__restrict__ w, int flag, const double *__restrict__ ws, int squared,
                                                         double *__restrict__ z, int64_t ldz, double *__restrict__ yz)
{
    const int j = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double wi = squared ? w[i] : sqrt(w[i]);
    if (j == p) { yz[i] = wi * ((y[i] - ws[0]) / ws[1]); return; }
    const double m = ws[2 + j], s = ws[2 + p + j];
    const double t = x[(size_t)j * ld + i] - m;
    z[(size_t)j * ldz + i] = wi * (flag == 1 ? t * (1.0 / s) : t / s);
}

__global__ void wstd_patch_stats_kernel(const double *__restrict__ ws, int p, double *__restrict__ stats)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) { stats[0] = ws[0]; stats[1] = ws[1]; }
    if (j < p) { stats[4 + j] = ws[2 + j]; stats[4 + p + j] = ws[2 + p + j]; }
}

}  // namespace

int launch_weighted_stats(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, const double *w, int flag, double *ws)
{
    hipLaunchKernelGGL(wstd_stats_kernel, dim3(p + 1), dim3(256), 0, s, x, n, ld, p, y, w, flag, ws);
    OEM_HIP(hipGetLastError());
    return 0;
}

int launch_weighted_apply(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, const double *w, int flag, const double *ws,
                          int squared, double *z, int64_t ldz, double *yz)
{
    hipLaunchKernelGGL(wstd_apply_kernel, dim3((unsigned)((n + 255) / 256), p + 1), dim3(256), 0, s, x, n, ld, p, y, w, flag, ws, squared, z, ldz, yz);
    OEM_HIP(hipGetLastError());
    return 0;
}

int launch_weighted_patch_stats(hipStream_t s, const double *ws, int p, double *stats)
{
    hipLaunchKernelGGL(wstd_patch_stats_kernel, dim3((p + 255) / 256), dim3(256), 0, s, ws, p, stats);
    OEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace oemgpu
