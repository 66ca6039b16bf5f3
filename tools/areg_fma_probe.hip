// What a tile product of path_symcoop.hip issues at, alone ("measure what an areg_rd-fed two-FMA chain issues at").
//   hipcc --offload-arch=gfx950 -O3 -o tools/areg_fma_probe tools/areg_fma_probe.hip && ./tools/areg_fma_probe
// One workgroup of four waves (one per SIMD), each lane with 64 doubles of "tile" -- in VGPRs, or in AGPRs a0..a127 read with two
// v_accvgpr_read_b32 per double as sx_tile does -- and per element the two FMAs of the symmetric product (g_I += t b_J, g_J += t b_I:
// eight + eight accumulators).  Prints cycles per tile of: the FMAs alone (VGPR tile), the reads alone, reads + FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int IDX> __device__ __forceinline__ double areg_rd()
{
    unsigned l, h;
    asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\tv_accvgpr_read_b32 %1, a[%3]" : "=v"(l), "=v"(h) : "n"(2 * IDX), "n"(2 * IDX + 1));
    return __hiloint2double((int)h, (int)l);
}
template <int IDX> __device__ __forceinline__ void areg_wr(double x)
{
    asm volatile("v_accvgpr_write_b32 a[%2], %0\n\tv_accvgpr_write_b32 a[%3], %1" ::"v"(__double2loint(x)), "v"(__double2hiint(x)), "n"(2 * IDX), "n"(2 * IDX + 1));
}
template <int N, typename F> __device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (N > 0) { sfor<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// MODE 0: VGPR tile, FMAs; 1: AGPR tile, reads only (summed into one register so that they are not dead); 2: AGPR tile, reads + FMAs
// DIAG: the 64 elements in diagonal order (j = (i + s) & 7): sixteen consecutive FMAs go to sixteen different accumulators, where the
// row-major order of round 4 returns to the same row accumulator every other instruction
template <int MODE, bool DIAG>
__global__ __launch_bounds__(256) void tile_probe(double *out, int iters, unsigned long long *cyc)
{
    asm volatile("" ::: "a127");
    double t[64], bj[8], bi[8], ad[8], at[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { bj[k] = out[k] + threadIdx.x; bi[k] = out[8 + k] - threadIdx.x; ad[k] = 0.0; at[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < 64; ++k) t[k] = out[16 + (k & 7)] * (k + 1);
    if (MODE != 0) sfor<64>([&](auto K_) { constexpr int k = decltype(K_)::value; areg_wr<k>(t[k]); });
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc32 = 0;
    for (int it = 0; it < iters; ++it) {
        sfor<8>([&](auto I_) {
            sfor<8>([&](auto J_) {
                constexpr int i = DIAG ? decltype(J_)::value : decltype(I_)::value, j = DIAG ? ((decltype(J_)::value + decltype(I_)::value) & 7) : decltype(J_)::value;
                double x;
                if constexpr (MODE == 0) { x = t[i * 8 + j]; asm volatile("" : "+v"(x)); }
                else x = areg_rd<i * 8 + j>();
                if constexpr (MODE == 1) { acc32 ^= (unsigned)__double2loint(x) ^ (unsigned)__double2hiint(x); }
                else { ad[i] = fma(x, bj[j], ad[i]); at[j] = fma(x, bi[i], at[j]); }
            });
        });
        if (MODE != 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(ad[k]), "+v"(at[k]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = (double)acc32;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += ad[k] + at[k];
    out[32 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 4096 * sizeof(double)); hipMalloc(&cyc, 64);
    double h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 1.0 / (i + 3);
    hipMemcpy(out, h, sizeof h, hipMemcpyHostToDevice);
    const int iters = 2000;
    const char *names[3] = {"VGPR tile: 128 FMAs per lane                                   ", "AGPR tile: 128 v_accvgpr_read_b32 per lane, no FMA             ",
                            "AGPR tile: 128 v_accvgpr_read_b32 + 128 FMAs per lane (sx_tile)"};
    for (int m = 0; m < 6; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL((tile_probe<0, false>), dim3(1), dim3(256), 0, 0, out, iters, cyc);
            if (m == 1) hipLaunchKernelGGL((tile_probe<1, false>), dim3(1), dim3(256), 0, 0, out, iters, cyc);
            if (m == 2) hipLaunchKernelGGL((tile_probe<2, false>), dim3(1), dim3(256), 0, 0, out, iters, cyc);
            if (m == 3) hipLaunchKernelGGL((tile_probe<0, true>), dim3(1), dim3(256), 0, 0, out, iters, cyc);
            if (m == 4) hipLaunchKernelGGL((tile_probe<1, true>), dim3(1), dim3(256), 0, 0, out, iters, cyc);
            if (m == 5) hipLaunchKernelGGL((tile_probe<2, true>), dim3(1), dim3(256), 0, 0, out, iters, cyc);
            hipDeviceSynchronize();
        }
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%s %s: %8.1f cycles per tile (s_memtime)\n", names[m % 3], m < 3 ? "row-major order" : "diagonal order ", (double)c / iters);
    }
    return 0;
}
