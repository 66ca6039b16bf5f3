// Determines the operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 by brute force, and times the
// mixed MFMA stream the strip form of the Gram kernel would issue (21 x 16x16x4 + 14 x 4x4x4_4b per k-step).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma444_layout tools/mfma444_layout.hip && ./tools/mfma444_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

__global__ void one(const double *a, const double *b, double *d)
{
    double av = a[threadIdx.x], bv = b[threadIdx.x], acc = 0.0;
    asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "v"(av), "v"(bv));
    d[threadIdx.x] = acc;
}

// indicator probe: block (la, lb) runs A = e_la, B = e_lb and stores the 64 results
__global__ void indic(double *d)
{
    const int la = blockIdx.x, lb = blockIdx.y;
    double av = (int)threadIdx.x == la ? 1.0 : 0.0, bv = (int)threadIdx.x == lb ? 1.0 : 0.0, acc = 0.0;
    asm volatile("s_nop 4\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "v"(av), "v"(bv));
    d[((size_t)la * 64 + lb) * 64 + threadIdx.x] = acc;
}

typedef double v4d __attribute__((ext_vector_type(4)));
template <int N16, int N4>
__global__ __launch_bounds__(256) void mixed(double *out, int iters, unsigned long long *cyc)
{
    double a = threadIdx.x * 1e-3 + 1.0, b = 0.5 - threadIdx.x * 1e-4;
    v4d acc16[N16 ? N16 : 1];
    double acc4[N4 ? N4 : 1];
    for (int i = 0; i < N16; ++i) acc16[i] = v4d{0, 0, 0, 0};
    for (int i = 0; i < N4; ++i) acc4[i] = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < N16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc16[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < N4; ++i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acc4[i]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    double s = 0;
    for (int i = 0; i < N16; ++i) s += acc16[i].x + acc16[i].y + acc16[i].z + acc16[i].w;
    for (int i = 0; i < N4; ++i) s += acc4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int N16, int N4> void run_mixed(const char *name)
{
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, sizeof(double) * 256 * 256); (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipLaunchKernelGGL((mixed<N16, N4>), dim3(256), dim3(256), 0, 0, out, 100, cyc);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((mixed<N16, N4>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s cycles per trip = %8.1f   (sum of parts: %d x 64 + %d x 16.5 = %.0f)\n", name, (double)h / iters, N16, N4,
           N16 * 64.0 + N4 * 16.5);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    srand(7);
    for (int l = 0; l < 64; ++l) { ha[l] = (rand() % 1000) / 100.0 + 1.0; hb[l] = (rand() % 1000) / 100.0 - 3.0; }
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    // candidates: within a block of 16 lanes, t = lane % 16; A: (i, k) = (t%4, t/4) [ma=0] or (t/4, t%4) [ma=1];
    // B: (j, k) = (t%4, t/4) [mb=0] or (t/4, t%4) [mb=1]; D: (i, j) = (t/4, t%4) [md=0] or (t%4, t/4) [md=1]
    for (int ma = 0; ma < 2; ++ma) for (int mb = 0; mb < 2; ++mb) for (int md = 0; md < 2; ++md) {
        double err = 0;
        for (int blk = 0; blk < 4; ++blk) {
            double A[4][4], B[4][4];
            for (int t = 0; t < 16; ++t) {
                const int x = t % 4, y = t / 4;
                if (ma == 0) A[x][y] = ha[16 * blk + t]; else A[y][x] = ha[16 * blk + t];       // A[i][k]
                if (mb == 0) B[y][x] = hb[16 * blk + t]; else B[x][y] = hb[16 * blk + t];       // B[k][j]
            }
            for (int t = 0; t < 16; ++t) {
                const int i = md == 0 ? t / 4 : t % 4, j = md == 0 ? t % 4 : t / 4;
                double s = 0;
                for (int k = 0; k < 4; ++k) s += A[i][k] * B[k][j];
                err = fmax(err, fabs(s - hd[16 * blk + t]));
            }
        }
        printf("A:%s  B:%s  D:%s  max err %.3g%s\n", ma ? "(i=t/4,k=t%4)" : "(i=t%4,k=t/4)", mb ? "(j=t/4,k=t%4)" : "(j=t%4,k=t/4)",
               md ? "(i=t%4,j=t/4)" : "(i=t/4,j=t%4)", err, err < 1e-9 ? "   <== MATCH" : "");
    }
    {
        double *dd2; static double h[64 * 64 * 64];
        (void)hipMalloc(&dd2, sizeof(h));
        hipLaunchKernelGGL(indic, dim3(64, 64), dim3(64), 0, 0, dd2);
        (void)hipMemcpy(h, dd2, sizeof(h), hipMemcpyDeviceToHost);
        // for every result lane: the (A lane, B lane) pairs whose product lands there
        for (int ld = 0; ld < 64; ++ld) {
            printf("D lane %2d <-", ld);
            int cnt = 0;
            for (int la = 0; la < 64; ++la)
                for (int lb = 0; lb < 64; ++lb)
                    if (h[((size_t)la * 64 + lb) * 64 + ld] != 0.0 && cnt++ < 8) printf(" A%d*B%d", la, lb);
            printf("  (%d pairs)\n", cnt);
        }
    }
    run_mixed<21, 14>("21 x 16x16x4 + 14 x 4x4x4_4b");
    run_mixed<28, 1>("28 x 16x16x4 + 1 x 4x4x4");
    run_mixed<1, 28>("1 x 16x16x4 + 28 x 4x4x4_4b");
    return 0;
}
