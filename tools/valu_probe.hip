// Latency / issue-rate probe for the FP64 VALU and LDS instructions the small-p path kernel is made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe tools/valu_probe.hip && ./tools/valu_probe
// One wave per SIMD (256 threads, one workgroup), timed with s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH>
__global__ __launch_bounds__(512) void fma_chain(double *out, int iters, unsigned long long *cyc)
{
    double a[CH];
    const double m = out[0], c = out[1];
#pragma unroll
    for (int k = 0; k < CH; ++k) a[k] = threadIdx.x + k;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
            for (int k = 0; k < CH; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) s += a[k];
    out[2 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// LDS round trip: write own value, read a broadcast value, dependent
__global__ __launch_bounds__(256) void lds_roundtrip(double *out, int iters, unsigned long long *cyc)
{
    __shared__ double sh[512];
    double v = threadIdx.x;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        sh[threadIdx.x] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        v = sh[(threadIdx.x & 192) + ((it * 7) & 63)] + 1.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[2 + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// barrier round: LDS write, s_barrier, LDS read of another wave's value
__global__ __launch_bounds__(512) void barrier_round(double *out, int iters, unsigned long long *cyc)
{
    __shared__ double sh[2][512];
    double v = threadIdx.x;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        sh[it & 1][threadIdx.x] = v;
        __syncthreads();
        v = sh[it & 1][(threadIdx.x + 64) & (blockDim.x - 1)] + 1.0;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[2 + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// DPP wave reduction of a double (the path kernel's wave_sum)
__device__ __forceinline__ double dpp_sum(double x)
{
#define STEP(ctrl, rm)                                                                                  \
    {                                                                                                    \
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, rm, 0xf, false);               \
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, rm, 0xf, false);               \
        x += __hiloint2double(hi, lo);                                                                   \
    }
    STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
#undef STEP
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
__global__ __launch_bounds__(256) void dpp_round(double *out, int iters, unsigned long long *cyc)
{
    double v = threadIdx.x;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) v = dpp_sum(v) * 1e-3 + threadIdx.x;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[2 + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- prototype of a "sliced" OEM round (p = 100: 4 waves x 26 columns, rows lane / lane+64):
//      threshold on the wave's own 26 entries (2 per lane, replicated in the four 16-lane rows), GEMV with
//      v_fmac_f64_dpp row_newbcast (no LDS broadcast reads), partials through LDS + one barrier.
template <int K> struct Bc {
    static __device__ __forceinline__ void fmac(double &acc, const double &b, const double &a)
    {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(a), "n"(K));
    }
};
template <int CW, int J, int K> struct Row {
    static __device__ __forceinline__ void run(double (&acc)[2][2], const double (&B)[2], const double (&a)[2][CW])
    {
        if constexpr (16 * J + K < CW) {
            Bc<K>::fmac(acc[0][K & 1], B[J], a[0][16 * J + K]);
            Bc<K>::fmac(acc[1][K & 1], B[J], a[1][16 * J + K]);
            if constexpr (K + 1 < 16) Row<CW, J, K + 1>::run(acc, B, a);
            else if constexpr (J + 1 < 2) Row<CW, J + 1, 0>::run(acc, B, a);
        }
    }
};
template <int MASK>
__global__ __launch_bounds__(256) void sliced_round(double *out, int iters, unsigned long long *cyc)
{
    constexpr int CW = 26, NW = 4;
    __shared__ double P[2][NW][128];
    __shared__ int F[2][NW];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15;
    double a[2][CW];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < CW; ++k) a[r][k] = 1e-3 * ((lane + 64 * r) == (w * CW + k) ? 50.0 : out[2 + ((lane * 7 + k * 3 + r) & 127)]);
    int e[2]; e[0] = w * CW + l16; e[1] = (16 + l16 < CW) ? w * CW + 16 + l16 : 127;
    double xy[2] = {out[2 + e[0]], out[2 + e[1]]}, B[2] = {0.0, 0.0}, ab[2] = {0.0, 0.0}, bold[2];
    const double t = 1e-4, D = 1.7, rD = 1.0 / D, tol = 1e-300;
    int buf = 0, it = 0;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (;;) {
        bool bad = false;
        if (MASK & 1)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bold[j] = B[j];
            const double u = ab[j] + xy[j];
            const double m = fmax(fabs(u) - t, 0.0);
            const double num = copysign(m, u);
            double q = num * rD;
            const double rr = fma(-q, D, num);
            q = fma(rr, rD, q);
            B[j] = q;
            const double c = fabs(B[j]), qq = fabs(bold[j]);
            const bool cn = c > 1e-13, qn = qq > 1e-13;
            bad |= (cn != qn);
            bad |= (cn && qn && fabs(B[j] - bold[j]) > tol * qq);
        }
        ++it;
        double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        asm volatile("s_nop 1" ::: "memory");
        if (MASK & 2) Row<CW, 0, 0>::run(acc, B, a);
        else { acc[0][0] = B[0]; acc[1][0] = B[1]; }
        if (!(MASK & 4)) { ab[0] = acc[0][0] + acc[0][1]; ab[1] = acc[1][0] + acc[1][1]; if (it >= iters) break; continue; }
        P[buf][w][lane] = acc[0][0] + acc[0][1];
        P[buf][w][lane + 64] = acc[1][0] + acc[1][1];
        const unsigned long long bl = __ballot(bad);
        if (lane == 0) F[buf][w] = (bl != 0ull);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double p0 = P[buf][0][e[j]], p1 = P[buf][1][e[j]], p2 = P[buf][2][e[j]], p3 = P[buf][3][e[j]];
            ab[j] = (p0 + p1) + (p2 + p3);
        }
        const int f = F[buf][0] | F[buf][1] | F[buf][2] | F[buf][3];
        buf ^= 1;
        if (__builtin_amdgcn_readfirstlane(f) == 12345 || it >= iters) break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[256 + tid] = B[0] + B[1] + ab[0];
    if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = it; }
}

template <typename K>
static void run(const char *name, K kern, int threads, int iters, double per)
{
    double *out; unsigned long long *cyc, h = 0;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 16); hipMemset(out, 0, 8 * 1024);
    double init[1024];
    for (int k = 0; k < 1024; ++k) init[k] = ((k * 2654435761u) >> 8 & 0xffff) / 65536.0 - 0.5;
    init[0] = 0.999999; init[1] = 1e-9;
    hipMemcpy(out, init, sizeof(init), hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s threads=%4d  cycles per unit = %8.2f\n", name, threads, (double)h / ((double)iters * per));
    hipFree(out); hipFree(cyc);
}

int main()
{
    const int it = 20000;
    run("v_fma_f64 dependent chain (1 chain)", fma_chain<1>, 256, it, 16);
    run("v_fma_f64 2 chains (per FMA)", fma_chain<2>, 256, it, 32);
    run("v_fma_f64 4 chains (per FMA)", fma_chain<4>, 256, it, 64);
    run("v_fma_f64 8 chains (per FMA)", fma_chain<8>, 256, it, 128);
    run("v_fma_f64 8 chains, 2 waves/SIMD (per FMA)", fma_chain<8>, 512, it, 128);
    run("LDS write -> read round trip", lds_roundtrip, 256, it, 1);
    run("LDS write + s_barrier + read (4 waves)", barrier_round, 256, it, 1);
    run("LDS write + s_barrier + read (8 waves)", barrier_round, 512, it, 1);
    run("DPP wave_sum + fma", dpp_round, 256, it, 1);
    run("sliced OEM round prototype (p=100)", sliced_round<7>, 256, it, 1);
    run("  threshold + stop flags only", sliced_round<1>, 256, it, 1);
    run("  52 v_fmac_f64_dpp only", sliced_round<2>, 256, it, 1);
    run("  exchange only (write, barrier, 8+4 reads)", sliced_round<4>, 256, it, 1);
    run("  threshold + FMAs", sliced_round<3>, 256, it, 1);
    run("  FMAs + exchange", sliced_round<6>, 256, it, 1);
    return 0;
}

