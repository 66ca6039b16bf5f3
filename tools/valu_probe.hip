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
    __shared__ int F[2][NW][16];
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
        if (!(MASK & 1)) { B[0] = ab[0] * 0.5; B[1] = ab[1] * 0.5; }       // keep the round a dependent chain
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
        if (!(MASK & 8)) {
            const unsigned long long bl = __ballot(bad);
            if (MASK & 16) F[buf][w][lane & 15] = (bl != 0ull);
            else if (lane == 0) F[buf][w][0] = (bl != 0ull);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double p0 = P[buf][0][e[j]], p1 = P[buf][1][e[j]], p2 = P[buf][2][e[j]], p3 = P[buf][3][e[j]];
            ab[j] = (p0 + p1) + (p2 + p3);
        }
        int f = 0;
        if (!(MASK & 8)) f = F[buf][0][0] | F[buf][1][0] | F[buf][2][0] | F[buf][3][0];
        else if (it < 0) f = 12345;
        buf ^= 1;
        if (__builtin_amdgcn_readfirstlane(f) == 12345 || it >= iters) break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[256 + tid] = B[0] + B[1] + ab[0];
    if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = it; }
}


// barrier round with NRD dependent b64 reads per lane (stride: other waves' slots), as in the partial-sum exchange
template <int NRD, int REPL>
__global__ __launch_bounds__(256) void barrier_reads(double *out, int iters, unsigned long long *cyc)
{
    __shared__ double sh[2][NRD > 4 ? NRD : 4][128];
    double v = threadIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = REPL ? (w * 16 + (lane & 15)) : lane;                 // REPL: the four 16-lane rows read the same words
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        sh[it & 1][w][lane] = v;
        sh[it & 1][w][lane + 64] = v + 1.0;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < NRD; ++k) t += sh[it & 1][k & 3][e + 64 * (k >> 2)];
        v = t * 0.25;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[2 + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}


// ---- prototype of a "row-split" OEM round (p = 100, 8 waves): wave w owns rows [13 w, 13 w + 13) (one per lane of a
//      16-lane row, replicated in the four row groups), row group g multiplies the column slice [25 g, 25 g + 25);
//      the four slice sums meet through v_permlane16/32_swap; what crosses waves is the NEW beta (1 LDS write,
//      barrier, 2 reads per lane), not partial sums.
__device__ __forceinline__ double rowgroup_allreduce(double x)
{
    unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    auto l1 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double y = __hiloint2double((int)h1[0], (int)l1[0]) + __hiloint2double((int)h1[1], (int)l1[1]);
    lo = (unsigned)__double2loint(y); hi = (unsigned)__double2hiint(y);
    auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}
template <int CG, int C> struct GroupFma {
    static __device__ __forceinline__ void run(double (&acc)[4], const double (&B)[2], const double (&a)[CG])
    {
        if constexpr (C < CG) {
            Bc<(C & 15)>::fmac(acc[C & 3], B[C >> 4], a[C]);
            GroupFma<CG, C + 1>::run(acc, B, a);
        }
    }
};
template <int NW, int RW, int CG, int MODE>
__global__ __launch_bounds__(NW * 64) void rowsplit_round(double *out, int iters, unsigned long long *cyc)
{
    __shared__ double Gv[2][160];
    __shared__ int F[2][8];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    double a[CG];
#pragma unroll
    for (int k = 0; k < CG; ++k) a[k] = 1e-3 * ((w * RW + l16) == (g * CG + k) ? 50.0 : out[2 + ((lane * 7 + k * 3 + w) & 127)]);
    const bool rowok = l16 < RW;
    const int row = rowok ? w * RW + l16 : 159;
    const int wslot = (g == 0 && rowok) ? row : 159;                      // replicas and padding write a dummy word
    int e[2]; e[0] = g * CG + l16; e[1] = (16 + l16 < CG) ? g * CG + 16 + l16 : 159;
    double xy = out[2 + row], beta = 0.0, ab = 0.0, bold;
    const double t = 1e-4, D = 1.7, rD = 1.0 / D, tol = 1e-300;
    int buf = 0, it = 0, facc = 0;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (;;) {
        bold = beta;
        const double u = ab + xy;
        const double m = fmax(fabs(u) - t, 0.0);
        const double num = copysign(m, u);
        double q = num * rD;
        const double rr = fma(-q, D, num);
        q = fma(rr, rD, q);
        beta = rowok ? q : 0.0;
        const double c = fabs(beta), qq = fabs(bold);
        const bool cn = c > 1e-13, qn = qq > 1e-13;
        const bool bad = (cn != qn) || (cn && qn && fabs(beta - bold) > tol * qq);
        ++it;
        Gv[buf][wslot] = beta;
        if (MODE == 0 || MODE == 2 || MODE == 3) {
            const unsigned long long bl = __ballot(bad);
            if (MODE == 3) { if (lane < 8) F[buf][lane] = (lane == w) ? (bl != 0ull) : F[buf][lane]; }
            else if (lane == 0) F[buf][w] = (bl != 0ull);
        }
        __syncthreads();
        int4 f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0};
        if (MODE == 0 || MODE == 2 || MODE == 3) { f0 = *reinterpret_cast<const int4 *>(&F[buf][0]); f1 = *reinterpret_cast<const int4 *>(&F[buf][4]); }
        double B[2];
        B[0] = Gv[buf][e[0]]; B[1] = Gv[buf][e[1]];
        __builtin_amdgcn_sched_barrier(0);
        const int f = (f0.x | f0.y | f0.z | f0.w) | (NW > 4 ? (f1.x | f1.y | f1.z | f1.w) : 0);
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        asm volatile("s_nop 1" ::: "memory");
        if (MODE != 4) GroupFma<CG, 0>::run(acc, B, a);
        else { acc[0] = B[0]; acc[1] = B[1]; }
        const double part = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        ab = (MODE == 5) ? part : rowgroup_allreduce(part);
        buf ^= 1;
        if (MODE == 2) { facc |= f; if (it >= iters) break; }
        else if (__builtin_amdgcn_readfirstlane(f) == 12345 || it >= iters) break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[256 + tid] = beta + ab + facc;
    if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = it; }
}

template <typename K>
static void run(const char *name, K kern, int threads, int iters, double per)
{
    double *out; unsigned long long *cyc, h = 0;
    hipMalloc(&out, 16 * 1024); hipMalloc(&cyc, 16); hipMemset(out, 0, 16 * 1024);
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
    run("2 writes + barrier + 1 read", barrier_reads<1, 0>, 256, it, 1);
    run("2 writes + barrier + 2 reads", barrier_reads<2, 0>, 256, it, 1);
    run("2 writes + barrier + 4 reads", barrier_reads<4, 0>, 256, it, 1);
    run("2 writes + barrier + 8 reads", barrier_reads<8, 0>, 256, it, 1);
    run("2 writes + barrier + 8 reads, 4x replicated", barrier_reads<8, 1>, 256, it, 1);
    run("2 writes + barrier + 4 reads, 4x replicated", barrier_reads<4, 1>, 256, it, 1);
    run("sliced OEM round prototype (p=100)", sliced_round<7>, 256, it, 1);
    run("  threshold + stop flags only", sliced_round<1>, 256, it, 1);
    run("  52 v_fmac_f64_dpp only", sliced_round<2>, 256, it, 1);
    run("  exchange only (write, barrier, 8+4 reads)", sliced_round<4>, 256, it, 1);
    run("  threshold + FMAs", sliced_round<3>, 256, it, 1);
    run("  FMAs + exchange", sliced_round<6>, 256, it, 1);
    run("  all, no flags", sliced_round<15>, 256, it, 1);
    run("row-split round, 8 waves (p=100)", rowsplit_round<8, 13, 25, 0>, 512, it, 1);
    run("  no flags", rowsplit_round<8, 13, 25, 1>, 512, it, 1);
    run("  flags read, branch independent", rowsplit_round<8, 13, 25, 2>, 512, it, 1);
    run("  no flags, no FMAs", rowsplit_round<8, 13, 25, 4>, 512, it, 1);
    run("  no flags, no allreduce", rowsplit_round<8, 13, 25, 5>, 512, it, 1);
    run("row-split round, 4 waves p=64", rowsplit_round<4, 16, 16, 0>, 256, it, 1);
    run("  no flags", rowsplit_round<4, 16, 16, 1>, 256, it, 1);
    run("row-split round, 8 waves (p=128)", rowsplit_round<8, 16, 32, 0>, 512, it, 1);
    run("  all, flags without exec branch", sliced_round<23>, 256, it, 1);
    return 0;
}

