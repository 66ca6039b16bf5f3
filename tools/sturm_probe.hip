// What one top-Ritz evaluation (path_dev.hpp: tridiag_max) costs, by size and by bracket: cycles per call, per round, per step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/sturm_probe tools/sturm_probe.hip && ./tools/sturm_probe
// One wave; T = the Lanczos tridiagonal of a Marchenko-Pastur-like matrix (alpha ~ 9, beta ~ 0.09: config 1's scale).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../oem_amd/csrc/path_dev.hpp"

using namespace oemgpu;

__global__ __launch_bounds__(64) void probe(const double *al_g, const double *be_g, int m, double hint, double *out, unsigned long long *cyc)
{
    __shared__ double al[320], be[320];
    __shared__ __attribute__((aligned(16))) double sab[2 * (320 + 16)];
    const int lane = threadIdx.x;
    for (int j = lane; j < m; j += 64) { al[j] = al_g[j]; be[j] = be_g[j]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int nr = 0;
    const double th = tridiag_max(al, be, m, lane, sab, hint, &nr);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = th; out[1] = nr; cyc[0] = t1 - t0; }
}

int main()
{
    const int M = 288;
    std::vector<double> al(M), be(M);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0 - 0.5; };
    // Lanczos tridiagonal of a random Gram matrix (p = 288, n = 40 p; entries ~ N(0, 9)): what the engines really see
    {
        const int p = M, n = 40 * M;
        std::vector<double> X((size_t)n * p), A((size_t)p * p, 0.0);
        for (auto &x : X) { double u = 0; for (int k = 0; k < 12; ++k) u += rnd(); x = 3.0 * u; }
        for (int i = 0; i < p; ++i) for (int j = 0; j <= i; ++j) { double a = 0; for (int r = 0; r < n; ++r) a += X[(size_t)r * p + i] * X[(size_t)r * p + j]; A[(size_t)i * p + j] = A[(size_t)j * p + i] = a / n; }
        std::vector<double> v(p, 0.0), vp(p, 0.0), w(p), wv(p);
        for (int i = 0; i < p; ++i) w[i] = rnd();
        double bb = 0;
        for (int j = 0; j < M; ++j) {
            double nb = 0; for (int i = 0; i < p; ++i) nb += w[i] * w[i]; nb = std::sqrt(nb);
            if (j > 0) be[j - 1] = nb;
            bb = j > 0 ? nb : 0.0;
            for (int i = 0; i < p; ++i) { vp[i] = v[i]; v[i] = w[i] / nb; }
            double a = 0;
            for (int i = 0; i < p; ++i) { double t = 0; for (int k = 0; k < p; ++k) t += A[(size_t)i * p + k] * v[k]; wv[i] = t; a += t * v[i]; }
            al[j] = a;
            for (int i = 0; i < p; ++i) w[i] = wv[i] - a * v[i] - bb * vp[i];
        }
        be[M - 1] = 0.0;
    }
    double *dal, *dbe, *dout; unsigned long long *dcyc;
    hipMalloc(&dal, M * 8); hipMalloc(&dbe, M * 8); hipMalloc(&dout, 64); hipMalloc(&dcyc, 64);
    hipMemcpy(dal, al.data(), M * 8, hipMemcpyHostToDevice); hipMemcpy(dbe, be.data(), M * 8, hipMemcpyHostToDevice);
    for (int m : {16, 32, 48, 64, 128, 256}) {
        double th = 0, th2 = 0; unsigned long long c0 = 0, c1 = 0, c2 = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dal, dbe, m, -INFINITY, dout, dcyc);
            hipMemcpy(&th, dout, 8, hipMemcpyDeviceToHost); hipMemcpy(&c0, dcyc, 8, hipMemcpyDeviceToHost);
        }
        double r0 = 0, r1 = 0, r2 = 0;
        hipMemcpy(&r0, dout + 1, 8, hipMemcpyDeviceToHost);
        // hinted from 1e-6 below (a mid-convergence look) and from 1e-15 below (the last look)
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dal, dbe, m, th * (1 - 1e-6), dout, dcyc);
        hipMemcpy(&th2, dout, 8, hipMemcpyDeviceToHost); hipMemcpy(&c1, dcyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&r1, dout + 1, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dal, dbe, m, th * (1 - 2e-15), dout, dcyc);
        hipMemcpy(&th2, dout, 8, hipMemcpyDeviceToHost); hipMemcpy(&c2, dcyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&r2, dout + 1, 8, hipMemcpyDeviceToHost);
        printf("m=%3d theta=%.15g  cycles (rounds): no hint %llu (%d)   hint 1e-6 below %llu (%d)   hint 2e-15 below %llu (%d)  (same value: %d)\n", m, th, c0, (int)r0, c1, (int)r1, c2, (int)r2, th == th2);
    }
    return 0;
}
