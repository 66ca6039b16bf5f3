// sparse.hip -- the moment buffer of a compressed-sparse-column x WITHOUT densifying it (ref src/oem_sparse.h:493-615: XtX() of a
// dgCMatrix through Eigen's sparse product, column sums and X'y by column loops).
//
// The dense FP64-MFMA pass over zero-filled row tiles (api.hip: oemgpu_fit_sparse) costs n p^2 flops whatever the density.  Here the
// cost follows the non-zeros: for a column a the other columns' non-zeros are multiplied against a DENSE copy of a row chunk of
// column a held in LDS,
//     G[a][b] = sum over row chunks c of  sum_{k in column b, chunk c} val[k] * dense_a_c[row[k]],        b >= a,
// i.e. p * nnz / 2 LDS gathers in all (config of man/oem.Rd: 2.5e5 x 200 at 1 %: 5e7 gathers against 2e10 dense flops).
//   * csc_chunk_ptr_kernel: where every column enters every 8192-row chunk (one lower_bound per (column, chunk));
//   * csc_gram_kernel: workgroup w (1024 threads) owns columns a = w and a' = p - 1 - w (their work adds up to p + 1 columns:
//     balanced), walks the chunks in order, scatters the column's chunk into LDS, and each of its 64 sixteen-lane groups takes the
//     columns b = a + g, a + g + 64, ...: lanes over the non-zeros of b in the chunk, a four-stage DPP row sum, accumulated in an
//     LDS word only that group touches -- every sum has one fixed order, so the result is bitwise reproducible.  (Sixty-four
//     columns in flight per workgroup: a wave per column, one after the other, spent a memory round trip per column and chunk and
//     ran 1.3 ms on the man/oem.Rd example, three times the dense pass it was meant to beat.)
//   * csc_stats_kernel: column sums, X'y (y gathered at the column's rows), sum y, sum y^2, n.
// The result is the same (p + 2)^2 moment buffer about 0 that the MFMA kernels produce (include/oemgpu.h).
#include "common.hpp"
#include "path_dev.hpp"

namespace oemgpu {

namespace {

constexpr int SRC = 8192;           // rows per chunk: 64 KB of LDS

__global__ __launch_bounds__(256) void csc_chunk_ptr_kernel(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx, int p,
                                                            int nchunk, int32_t *__restrict__ cptr /* [nchunk + 1][p] */)
{
    const int j = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c > nchunk) return;
    const int64_t lo0 = colptr[j], hi0 = colptr[j + 1];
    const int64_t r0 = (int64_t)c * SRC;
    int64_t lo = lo0, hi = hi0;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)rowidx[mid] < r0) lo = mid + 1; else hi = mid; }
    cptr[(size_t)c * p + j] = (int32_t)(lo - lo0);
}

// M: the (p + 2)^2 moment buffer; writes M[i][j] for i, j < p (both triangles)
constexpr int GT = 1024;            // threads per workgroup: 64 sixteen-lane groups
__global__ __launch_bounds__(GT) void csc_gram_kernel(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                      const double *__restrict__ val, const int32_t *__restrict__ cptr, int p, int nchunk,
                                                      int nrange, double *__restrict__ part /* [nrange][p][p]: row a holds G[a][b], b >= a */)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *dense = lds;                                         // [SRC]
    double *acc = lds + SRC;                                     // [p]: G[a][b], b >= a
    int32_t *cp0 = reinterpret_cast<int32_t *>(acc + p);          // [p] chunk start of every column (offset inside the column)
    int32_t *cp1 = cp0 + p;                                      // [p] chunk end
    const int tid = threadIdx.x, grp = tid >> 4, l16 = tid & 15;
    // blockIdx.y: a contiguous range of chunks (more workgroups than CUs hide each other's memory round trips; the range sums are
    // added in range order by csc_finish_kernel)
    const int cper = (nchunk + nrange - 1) / nrange, c_lo = blockIdx.y * cper, c_hi = c_lo + cper < nchunk ? c_lo + cper : nchunk;
    double *out = part + (size_t)blockIdx.y * p * p;
    for (int k = tid; k < SRC; k += GT) dense[k] = 0.0;
    for (int half = 0; half < 2; ++half) {
        const int a = half == 0 ? (int)blockIdx.x : p - 1 - (int)blockIdx.x;
        if (half == 1 && a <= (int)blockIdx.x) break;            // the middle column of an odd p: done in the first half
        for (int b = tid; b < p; b += GT) acc[b] = 0.0;
        const int64_t ca = colptr[a];
        for (int c = c_lo; c < c_hi; ++c) {
            __syncthreads();                                     // the previous chunk's dense copy is clean again, cp0 / cp1 free
            for (int b = a + tid; b < p; b += GT) { cp0[b] = cptr[(size_t)c * p + b]; cp1[b] = cptr[(size_t)(c + 1) * p + b]; }
            __syncthreads();
            const int ka0 = cp0[a], ka1 = cp1[a];
            if (ka0 == ka1) continue;                            // column a has nothing in this chunk (uniform)
            const int base = c * SRC;
            for (int k = ka0 + tid; k < ka1; k += GT) dense[rowidx[ca + k] - base] = val[ca + k];
            __syncthreads();
            for (int b = a + grp; b < p; b += GT / 16) {
                const int kb0 = cp0[b], kb1 = cp1[b];
                const int64_t cb = colptr[b];
                double s = 0.0;
                for (int k = kb0 + l16; k < kb1; k += 16) s = fma(val[cb + k], dense[rowidx[cb + k] - base], s);
                // sum over the sixteen lanes of the group (DPP row_shr 1, 2, 4, 8: lane 15 of the row ends with the total)
                s += dpp_mov<0x111, 0xf>(s, 0.0);
                s += dpp_mov<0x112, 0xf>(s, 0.0);
                s += dpp_mov<0x114, 0xf>(s, 0.0);
                s += dpp_mov<0x118, 0xf>(s, 0.0);
                if (l16 == 15 && kb1 > kb0) acc[b] += s;         // only this group ever touches acc[b]
            }
            __syncthreads();
            for (int k = ka0 + tid; k < ka1; k += GT) dense[rowidx[ca + k] - base] = 0.0;
        }
        __syncthreads();
        for (int b = a + tid; b < p; b += GT) out[(size_t)a * p + b] = acc[b];
        __syncthreads();
    }
}

// range sums -> M (both triangles), in range order; the y partials -> M's y entries
__global__ __launch_bounds__(256) void csc_finish_kernel(const double *__restrict__ part, int nrange, const double *__restrict__ ypart, int ny,
                                                         int64_t n, int p, double *__restrict__ M)
{
    const int q = p + 2;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < (size_t)p * p) {
        const int a = (int)(t / p), b = (int)(t % p);
        if (b >= a) {
            double g = 0.0;
            for (int s_ = 0; s_ < nrange; ++s_) g += part[(size_t)s_ * p * p + t];
            M[(size_t)a * q + b] = g;
            M[(size_t)b * q + a] = g;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s0 = 0.0, s1 = 0.0;
        for (int k = 0; k < ny; ++k) { s0 += ypart[2 * k]; s1 += ypart[2 * k + 1]; }
        M[(size_t)p * q + (p + 1)] = s0; M[(size_t)(p + 1) * q + p] = s0;      // sum y
        M[(size_t)p * q + p] = s1;                                             // sum y^2
        M[(size_t)(p + 1) * q + (p + 1)] = (double)n;
    }
}

// one workgroup per column: sum x_j, sum x_j y; workgroups p .. p + NY - 1: a slice of y each (sum y, sum y^2 -> ypart)
constexpr int NY = 64;
__global__ __launch_bounds__(256) void csc_stats_kernel(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                        const double *__restrict__ val, const double *__restrict__ y, int64_t n, int p,
                                                        double *__restrict__ M, double *__restrict__ ypart)
{
    __shared__ double sh[2][256];
    const int j = blockIdx.x, tid = threadIdx.x, q = p + 2;
    double s0 = 0.0, s1 = 0.0;
    if (j < p) {
        for (int64_t k = colptr[j] + tid; k < colptr[j + 1]; k += 256) { const double v = val[k]; s0 += v; s1 = fma(v, y[rowidx[k]], s1); }
    } else {
        const int64_t per = (n + NY - 1) / NY, r0 = (int64_t)(j - p) * per, r1 = r0 + per < n ? r0 + per : n;
        for (int64_t r = r0 + tid; r < r1; r += 256) { const double v = y[r]; s0 += v; s1 = fma(v, v, s1); }
    }
    sh[0][tid] = s0; sh[1][tid] = s1;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if (tid < h) { sh[0][tid] += sh[0][tid + h]; sh[1][tid] += sh[1][tid + h]; } __syncthreads(); }
    if (tid == 0) {
        if (j < p) {
            M[(size_t)j * q + (p + 1)] = sh[0][0]; M[(size_t)(p + 1) * q + j] = sh[0][0];      // sum x_j
            M[(size_t)j * q + p] = sh[1][0];       M[(size_t)p * q + j] = sh[1][0];            // sum x_j y
        } else { ypart[2 * (j - p)] = sh[0][0]; ypart[2 * (j - p) + 1] = sh[1][0]; }
    }
}

// compute.loss of oemSparse where the path engine cannot take it itself (the in-place rescale of the intercept slot, ref
// src/oem_sparse.h:897-900, changes the member before get_loss): sum (Y - X beta)^2 of the coefficients in the coordinates of the
// iteration through the Gram identity yy + n (beta' XX beta - 2 beta' XY) (ref src/oem_sparse.h:919-944 takes the residual of the same
// fitted values).  One workgroup per (penalty, lambda); only the rows of non-zero coefficients are read.
__global__ __launch_bounds__(256) void gram_loss_kernel(const double *__restrict__ xx, const double *__restrict__ xy, const double *__restrict__ stats, int q,
                                                        const double *__restrict__ beta, const double *__restrict__ sinv, const int *__restrict__ niter,
                                                        double *__restrict__ loss)
{
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    __shared__ double part[4];
    const double *b = beta + (size_t)k * q;
    if (niter[k] == 0) { if (tid == 0) loss[k] = 1e99; return; }
    double t = 0.0;
    for (int i = w; i < q; i += 4) {
        const double bi = sinv ? b[i] / sinv[i] : b[i];
        if (bi == 0.0) continue;                                     // (wave-uniform)
        const double *row = xx + (size_t)i * q;                      // XX is symmetric: row i = column i, contiguous
        double g = 0.0;
        for (int j = lane; j < q; j += 64) {
            const double bj = sinv ? b[j] / sinv[j] : b[j];
            g = fma(row[j], bj, g);
        }
        g = wave_sum(g);
        t += bi * (g - 2.0 * xy[i]);
    }
    if (lane == 0) part[w] = t;
    __syncthreads();
    if (tid == 0) loss[k] = stats[2] + stats[3] * ((part[0] + part[1]) + (part[2] + part[3]));
}

// the same for p >= n (no Gram matrix): the residual itself, Y - X beta of the RETURNED coefficients on the data as they are
// (ref src/oem_sparse.h:932-941: with standardize the member times colsq_inv -- what get_beta returns).  2048 rows per workgroup,
// the chunks of a (penalty, lambda) added in chunk order by resid_loss_sum_kernel.
__global__ __launch_bounds__(256) void resid_loss_kernel(const double *__restrict__ x, int64_t n, int64_t ld, int p, const double *__restrict__ y,
                                                         const double *__restrict__ beta, int rows, double *__restrict__ part)
{
    const int c = blockIdx.x, k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    __shared__ double ws[4];
    const double *b = beta + (size_t)k * rows + 1;                   // slot 0: the intercept (none here)
    double r[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { const int64_t i = (int64_t)c * 2048 + m * 256 + tid; r[m] = i < n ? y[i] : 0.0; }
    for (int j = 0; j < p; ++j) {
        const double bj = b[j];
        if (bj == 0.0) continue;                                     // (uniform)
        const double *col = x + (size_t)j * ld;
#pragma unroll
        for (int m = 0; m < 8; ++m) { const int64_t i = (int64_t)c * 2048 + m * 256 + tid; if (i < n) r[m] = fma(-col[i], bj, r[m]); }
    }
    double t = 0.0;
#pragma unroll
    for (int m = 0; m < 8; ++m) t = fma(r[m], r[m], t);
    t = wave_sum(t);
    if (lane == 0) ws[w] = t;
    __syncthreads();
    if (tid == 0) part[(size_t)k * gridDim.x + c] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ void resid_loss_sum_kernel(const double *__restrict__ part, int nchunk, int nk, double *__restrict__ loss)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    double t = 0.0;
    for (int c = 0; c < nchunk; ++c) t += part[(size_t)k * nchunk + c];
    loss[k] = t;
}

}  // namespace

int launch_gram_loss(hipStream_t s, const double *xx, const double *xy, const double *stats, int q, const double *beta, const double *sinv,
                     const int *niter, double *loss, int nk)
{
    hipLaunchKernelGGL(gram_loss_kernel, dim3(nk), dim3(256), 0, s, xx, xy, stats, q, beta, sinv, niter, loss);
    OEM_HIP(hipGetLastError());
    return 0;
}

// part: nk * ceil(n / 2048) doubles of scratch; loss: nk doubles (device)
int launch_resid_loss(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, const double *beta, int rows, int nk,
                      double *part, double *loss)
{
    const int nchunk = (int)((n + 2047) / 2048);
    hipLaunchKernelGGL(resid_loss_kernel, dim3(nchunk, nk), dim3(256), 0, s, x, n, ld, p, y, beta, rows, part);
    OEM_HIP(hipGetLastError());
    hipLaunchKernelGGL(resid_loss_sum_kernel, dim3((nk + 255) / 256), dim3(256), 0, s, part, nchunk, nk, loss);
    OEM_HIP(hipGetLastError());
    return 0;
}

static int csc_ranges(int64_t n, int p)
{
    const int nchunk = (int)((n + SRC - 1) / SRC), half = (p + 1) / 2;
    int nr = 1024 / (half > 0 ? half : 1);             // ~1024 workgroups in all
    if (nr > nchunk) nr = nchunk;
    // the range sums cost nr * p^2 doubles: keep them under 256 MB
    while (nr > 1 && (double)nr * p * p * 8.0 > 256e6) --nr;
    return nr < 1 ? 1 : nr;
}

size_t csc_moments_work_bytes(int64_t n, int p)
{
    return sizeof(int32_t) * ((size_t)((n + SRC - 1) / SRC) + 1) * (size_t)p + 256 + sizeof(double) * ((size_t)csc_ranges(n, p) * p * p + 2 * NY) + 256;
}

// can the compressed-column kernel take this matrix (LDS: the dense chunk, p accumulators, 2 p chunk pointers)?
bool csc_moments_fits(int p) { return (size_t)SRC * 8 + (size_t)p * 16 + 64 <= 160 * 1024; }

int launch_csc_moments(hipStream_t s, const int64_t *colptr, const int32_t *rowidx, const double *val, const double *y, int64_t n, int p,
                       void *work, double *moments)
{
    const int nchunk = (int)((n + SRC - 1) / SRC), nrange = csc_ranges(n, p);
    int32_t *cptr = reinterpret_cast<int32_t *>(work);
    const size_t cbytes = (sizeof(int32_t) * ((size_t)nchunk + 1) * (size_t)p + 255) / 256 * 256;
    double *part = reinterpret_cast<double *>((char *)work + cbytes), *ypart = part + (size_t)nrange * p * p;
    hipLaunchKernelGGL(csc_chunk_ptr_kernel, dim3(p, (nchunk + 1 + 255) / 256), dim3(256), 0, s, colptr, rowidx, p, nchunk, cptr);
    const size_t sh = (size_t)SRC * 8 + (size_t)p * 16 + 64;
    if (sh > 64 * 1024) OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&csc_gram_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(csc_gram_kernel, dim3((p + 1) / 2, nrange), dim3(GT), sh, s, colptr, rowidx, val, cptr, p, nchunk, nrange, part);
    hipLaunchKernelGGL(csc_stats_kernel, dim3(p + NY), dim3(256), 0, s, colptr, rowidx, val, y, n, p, moments, ypart);
    hipLaunchKernelGGL(csc_finish_kernel, dim3((unsigned)(((size_t)p * p + 255) / 256)), dim3(256), 0, s, part, nrange, ypart, NY, n, p, moments);
    OEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace oemgpu
