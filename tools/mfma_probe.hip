// mfma_probe.hip -- what does the FP64 matrix pipe of gfx950 sustain?  (calibrates roofline.peak in DESIGN.md)
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip     run: tools/mfma_probe
// Variants: bare v_mfma_f64_16x16x4_f64 stream over NACC independent accumulators, 1 or 2 waves per SIMD,
// optionally with NV independent v_fma_f64 between consecutive MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#include <type_traits>
#include "../oem_amd/csrc/gen/acc_tiles.inc"      // asm-owned AGPR tiles (hipcc shuttles "+a" accumulators through VGPRs)
template <int N, typename F> __device__ __forceinline__ void static_for(F &&fn)
{
    if constexpr (N > 0) { static_for<N - 1>(fn); fn(std::integral_constant<int, N - 1>{}); }
}

template <int I, int NACC, int NV> __device__ __forceinline__ void body(double a, double b, double (&f)[8])
{
    if constexpr (I < NACC) {
        AccTile<I>::mfma(a, b);
#pragma unroll
        for (int k = 0; k < NV; ++k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[(I * NV + k) & 7]) : "v"(a), "v"(b));
        body<I + 1, NACC, NV>(a, b, f);
    }
}
template <int I, int NACC> __device__ __forceinline__ void zero_all()
{
    if constexpr (I < NACC) { AccTile<I>::zero(); zero_all<I + 1, NACC>(); }
}
template <int I, int NACC> __device__ __forceinline__ double sum_all()
{
    if constexpr (I < NACC) return AccTile<I>::template read<0>() + AccTile<I>::template read<3>() + sum_all<I + 1, NACC>();
    else return 0.0;
}

template <int NACC, int NV>
__global__ __launch_bounds__(256) void probe(double *out, int iters, unsigned long long *cyc)
{
    zero_all<0, NACC>();
    double a = threadIdx.x * 1e-3 + 1.0, b = 0.5 - threadIdx.x * 1e-4;
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = a + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) body<0, NACC, NV>(a, b, f);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    double s = sum_all<0, NACC>();
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int NV> void run(const char *name, int blocks, int threads)
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, NV>), dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * NACC;                       // MFMAs per wave
    const double waves = (double)blocks * threads / 64;
    const double tf = nm * waves * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-34s blocks=%4d thr=%4d  cycles/MFMA/wave=%7.2f  clock=%.2f GHz  MFMA TFLOP/s=%6.1f  ms=%.3f\n", name, blocks, threads,
           (double)h / nm, (double)h / (ms * 1e-3) / 1e9, tf, ms);
    hipFree(out); hipFree(cyc);
}

// v_mfma_f64_4x4x4_4b_f64: 4 blocks of 4x4x4 (512 flop), one f64 accumulator per lane
template <int NACC>
__global__ __launch_bounds__(256) void probe444(double *out, int iters, unsigned long long *cyc)
{
    double a = threadIdx.x * 1e-3 + 1.0, b = 0.5 - threadIdx.x * 1e-4;
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC> void run444(const char *name, int blocks, int threads)
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe444<NACC>), dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe444<NACC>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * NACC, waves = (double)blocks * threads / 64;
    printf("%-34s blocks=%4d thr=%4d  cycles/MFMA/wave=%7.2f  clock=%.2f GHz  MFMA TFLOP/s=%6.1f  ms=%.3f\n", name, blocks, threads,
           (double)h / nm, (double)h / (ms * 1e-3) / 1e9, nm * waves * 512.0 / (ms * 1e-3) / 1e12, ms);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run444<32>("4x4x4_4b, 32 acc, 1 wave/SIMD", 256, 256);
    run444<4>("4x4x4_4b, 4 acc, 1 wave/SIMD", 256, 256);
    run<28, 0>("bare, 28 acc, 1 wave/SIMD", 256, 256);
    run<21, 0>("bare, 21 acc, 1 wave/SIMD", 256, 256);
    run<16, 0>("bare, 16 acc, 2 waves/SIMD", 512, 256);
    run<4, 0>("bare, 4 acc, 1 wave/SIMD", 256, 256);
    run<21, 2>("21 acc + 2 v_fma_f64 per MFMA", 256, 256);
    run<21, 4>("21 acc + 4 v_fma_f64 per MFMA", 256, 256);
    run<21, 8>("21 acc + 8 v_fma_f64 per MFMA", 256, 256);
    run<28, 0>("bare, 28 acc, ONE CU only", 1, 256);
    run<28, 0>("bare, 28 acc, 1 wave on 1 CU", 1, 64);
    return 0;
}
