// wide.hip -- the p >= n branch of oem_fit_dense, first stage: DataStd on the data itself.
//
// With n > p everything the solver needs is in the (p+2)^2 moment buffer and X is read once (gram.hip).  With p >= n the
// reference does NOT form X'X (p^2 doubles: 3.2 GB at p = 20,000): it keeps the standardised X and iterates through it,
//   u = X'(Y - X beta)/n + d beta,   d = 1.005 lambda_max(X X'/n)          (ref src/oem_dense.h:363-366, 476-482, 513-521)
// so the standardised copy is materialised here exactly as the reference does (ref src/oem_dense.cpp:61-67 copies, DataStd
// standardises in place, src/DataStd.h:94-267): one wave per column, the column held in registers / re-read from L2,
// two-pass mean and centred norm like the reference's own loops, then XY = X'Y / n (ref src/oem_dense.h:699-707).
// The iteration itself is path_large.hip: run_path_wide.
#include "common.hpp"

namespace oemgpu {

namespace {

template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dppw(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int rlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int rhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dppw<0x111, 0xf>(v); v += dppw<0x112, 0xf>(v); v += dppw<0x114, 0xf>(v); v += dppw<0x118, 0xf>(v);
    v += dppw<0x142, 0xa>(v); v += dppw<0x143, 0xc>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double block_sum1024(double v, double *sh)
{
    const int w = threadIdx.x >> 6;
    const double s = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = s;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[k];
    return t;
}

// Y: flag 1 scale by the centred sd_n; flags 2 and 3 (2 falls through into 3, quirk Q1): centre, scaleY = |Yc| / sqrt(n), divide
// (ref src/DataStd.h:112-138).  ys is written in the blocked row order of WideLayout (its padding was zeroed by the caller).
// stats: [0] meanY [1] scaleY [2] sum ys^2 [3] n.
__global__ __launch_bounds__(1024) void wide_y_kernel(const double *__restrict__ y, long long n, WideLayout lay, int flag,
                                                       double *__restrict__ ys, double *__restrict__ stats)
{
    __shared__ double sh[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    double meany = 0.0, scaley = 1.0;
    if (flag != 0) {
        double s = 0.0;
        for (long long i = tid; i < n; i += nt) s += y[i];
        const double mean = block_sum1024(s, sh) / (double)n;
        double c2 = 0.0;
        for (long long i = tid; i < n; i += nt) { const double c = y[i] - mean; c2 = fma(c, c, c2); }
        c2 = block_sum1024(c2, sh);
        scaley = sqrt(c2) * (1.0 / sqrt((double)n));           // sd_n(y) and |Yc| / sqrt(n) are the same number
        if (flag >= 2) meany = mean;
    }
    double yy = 0.0;
    for (long long i = tid; i < n; i += nt) {
        const double v = (flag == 0) ? y[i] : (flag == 1 ? y[i] / scaley : (y[i] - meany) / scaley);
        const long long b = i / lay.rb;
        ys[b * lay.npad() + (i - b * lay.rb)] = v;
        yy = fma(v, v, yy);
    }
    yy = block_sum1024(yy, sh);
    if (tid == 0) { stats[0] = meany; stats[1] = scaley; stats[2] = yy; stats[3] = (double)n; }
}

// X: one wave per column.  flag 1: scale by sd_n (about the mean, the column is NOT centred); flag 2: centre; flag 3: centre,
// scale = |Xc_j| / sqrt(n) (zero -> 1), divide (ref src/DataStd.h:203-265).  xs in the blocked layout (padding zeroed by the caller).
// Then XY_j = xs_j . ys / n.  A column constant to within 32 eps of its mean is the exact constant it is (as gram.hip: Mom::flat).
__global__ __launch_bounds__(256) void wide_x_kernel(const double *__restrict__ x, long long n, long long ld, int p, int flag,
                                                      const double *__restrict__ ys, WideLayout lay, double *__restrict__ xs,
                                                      double *__restrict__ xy, double *__restrict__ stats)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= p) return;
    const double *c = x + (size_t)j * ld;
    double mean = 0.0, scale = 1.0;
    bool flat = false;
    if (flag != 0) {
        double s = 0.0;
        for (long long i = lane; i < n; i += 64) s += c[i];
        mean = wave_sum(s) / (double)n;
        if (flag != 2) {
            double c2 = 0.0;
            for (long long i = lane; i < n; i += 64) { const double t = c[i] - mean; c2 = fma(t, t, c2); }
            c2 = wave_sum(c2);
            const double tol = 32.0 * 2.220446049250313e-16 * fabs(mean);
            flat = c2 <= tol * tol * (double)n;
            scale = flat ? 0.0 : sqrt(c2) * (1.0 / sqrt((double)n));
            if (scale == 0.0) scale = 1.0;
        }
    }
    const double rs = 1.0 / scale;
    const long long npb = lay.npad();
    double dot = 0.0;
    for (long long i = lane; i < n; i += 64) {
        const double t = c[i];
        double v;
        if (flag == 0) v = t;
        else if (flag == 1) v = t * rs;                           // ref :211-214 multiplies by the reciprocal
        else if (flag == 2) v = t - mean;
        else v = flat ? 0.0 : (t - mean) / scale;
        const long long b = i / lay.rb, off = b * npb + (i - b * lay.rb);
        dot = fma(v, ys[off], dot);
        xs[((size_t)b * p + j) * npb + (i - b * lay.rb)] = v;
    }
    dot = wave_sum(dot);
    if (lane == 0) {
        xy[j] = dot / (double)n;
        stats[4 + j] = (flag >= 2) ? mean : 0.0;
        stats[4 + p + j] = (flag & 1) ? scale : 1.0;
    }
}

}  // namespace

// big.oem / sparse x with p >= n (no intercept; ref src/oem_big.h:743-764, 810-826): colsq_inv_j = 1 / sqrt(sum x_j^2 / (n - 1)) (a zero
// column: 1) into the scale slots of `stats` -- get_beta multiplies the coefficients by it -- and xy_std = xy colsq_inv, which the
// reference uses for lambda_zero alone (its iteration runs on the data as they are).  One wave per column.
__global__ __launch_bounds__(256) void big_wide_scales_kernel(const double *__restrict__ x, long long n, long long ld, int p, const double *__restrict__ xy,
                                                               double *__restrict__ stats, double *__restrict__ xy_std)
{
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= p) return;
    const double *c = x + (size_t)j * ld;
    double s = 0.0;
    for (long long i = lane; i < n; i += 64) s = fma(c[i], c[i], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    double cs = s / ((double)n - 1.0);
    if (cs == 0.0) cs = 1.0;
    const double inv = 1.0 / sqrt(cs);
    if (lane == 0) { stats[4 + p + j] = inv; xy_std[j] = xy[j] * inv; }
}
// The same constants from the Gram of the data as they are (xx = X'X / n, stats[3] = n; fit_big_gram_dev): sum x_j^2 = n xx_jj.
// standardize == 0: scales 1, xy_std = xy.
__global__ __launch_bounds__(256) void big_gram_scales_kernel(const double *__restrict__ xx, const double *__restrict__ xy, int p, int standardize,
                                                               double *__restrict__ stats, double *__restrict__ xy_std)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    const double n = stats[3];
    double inv = 1.0;
    if (standardize) {
        double cs = xx[(size_t)j * p + j] * n / (n - 1.0);
        if (cs == 0.0) cs = 1.0;
        inv = 1.0 / sqrt(cs);
    }
    stats[4 + p + j] = inv;
    xy_std[j] = xy[j] * inv;
}
int launch_big_gram_scales(hipStream_t s, const double *xx, const double *xy, int p, int standardize, double *stats, double *xy_std)
{
    hipLaunchKernelGGL(big_gram_scales_kernel, dim3((p + 255) / 256), dim3(256), 0, s, xx, xy, p, standardize, stats, xy_std);
    OEM_HIP(hipGetLastError());
    return 0;
}
int launch_big_wide_scales(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *xy, double *stats, double *xy_std)
{
    hipLaunchKernelGGL(big_wide_scales_kernel, dim3((p + 3) / 4), dim3(256), 0, s, x, (long long)n, (long long)ld, p, xy, stats, xy_std);
    OEM_HIP(hipGetLastError());
    return 0;
}

int launch_wide_standardize(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, int standardize, int intercept,
                            const WideLayout &lay, double *xs, double *ys, double *xy, double *stats)
{
    const int flag = (standardize ? 1 : 0) + 2 * (intercept ? 1 : 0);
    OEM_HIP(hipMemsetAsync(stats + stats_shift_flag(p), 0, 2 * sizeof(double), s));      // no shift machinery on this branch
    // the padding rows of every block are zero: no kernel of the wide engine has bounds logic
    OEM_HIP(hipMemsetAsync(xs, 0, sizeof(double) * (size_t)lay.rows() * p, s));
    OEM_HIP(hipMemsetAsync(ys, 0, sizeof(double) * (size_t)lay.rows(), s));
    hipLaunchKernelGGL(wide_y_kernel, dim3(1), dim3(1024), 0, s, y, (long long)n, lay, flag, ys, stats);
    hipLaunchKernelGGL(wide_x_kernel, dim3((p + 3) / 4), dim3(256), 0, s, x, (long long)n, (long long)ld, p, flag, ys, lay, xs, xy, stats);
    OEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace oemgpu
