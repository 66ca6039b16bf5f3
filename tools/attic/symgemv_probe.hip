// symgemv_probe.hip -- launch-level timing of the symmetric-tile product (path_large.hip: symgemv_kernel<32>, q = 4096) against the
// row-streaming gemv_sym_kernel, with the grid cut short (no diagonal workgroups: results wrong by construction) to see what the
// dispatch tail costs.  Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DOEM_SYM_MINWG=3] tools/symgemv_probe.hip oem_amd/csrc/api.hip ... is not
// needed: the probe includes the engine's source and stubs what it does not use.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/symgemv_probe tools/symgemv_probe.hip
#include "../oem_amd/csrc/path_large.hip"
#include <cstdio>
#include <vector>
namespace oemgpu { void set_error(const char *, ...) {} bool caller_interrupted() { return false; } }
using namespace oemgpu;
int main()
{
    const int q = 4096, NB = 32;
    double *xx, *vec, *P, *out;
    hipMalloc(&xx, sizeof(double) * q * q); hipMalloc(&vec, sizeof(double) * q); hipMalloc(&P, sizeof(double) * 2 * NB * q); hipMalloc(&out, sizeof(double) * q);
    std::vector<double> h((size_t)q * q), hv(q);
    for (int i = 0; i < q; ++i) { hv[i] = (i % 7) * 0.25 - 0.5; for (int j = 0; j <= i; ++j) { const double v = ((i * 31 + j * 17) % 23) * 0.01 - 0.1; h[(size_t)i * q + j] = v; h[(size_t)j * q + i] = v; } }
    hipMemcpy(xx, h.data(), sizeof(double) * q * q, hipMemcpyHostToDevice); hipMemcpy(vec, hv.data(), sizeof(double) * q, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto &&launch, double bytes) {
        hipGraph_t g; hipGraphExec_t ex;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < 128; ++k) launch();
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        hipGraphLaunch(ex, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < 8; ++r) hipGraphLaunch(ex, s);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = 1e3 * ms / (8 * 128);
        printf("%-52s %7.2f us per launch (graph, back to back)  %6.2f TB/s of %.1f MB\n", name, us, bytes / us / 1e6, bytes / 1e6);
    };
    const double tri = 8.0 * (496.0 + 32.0) * 128 * 128, full = 8.0 * q * q;
    time("symgemv_kernel<32>, 528 workgroups", [&] { hipLaunchKernelGGL((symgemv_kernel<32>), dim3(528), dim3(256), 0, s, xx, vec, P); }, tri);
    time("symgemv_kernel<32>, 496 (no diagonal blocks)", [&] { hipLaunchKernelGGL((symgemv_kernel<32>), dim3(496), dim3(256), 0, s, xx, vec, P); }, 8.0 * 496 * 128 * 128);
    time("symgemv_kernel<32>, 512", [&] { hipLaunchKernelGGL((symgemv_kernel<32>), dim3(512), dim3(256), 0, s, xx, vec, P); }, 8.0 * 512 * 128 * 128);
    time("symgemv_kernel<32>, 256", [&] { hipLaunchKernelGGL((symgemv_kernel<32>), dim3(256), dim3(256), 0, s, xx, vec, P); }, 8.0 * 256 * 128 * 128);
    time("symgemv + sum", [&] { hipLaunchKernelGGL((symgemv_kernel<32>), dim3(528), dim3(256), 0, s, xx, vec, P); hipLaunchKernelGGL((symgemv_sum_kernel<32>), dim3(32), dim3(128), 0, s, P, out); }, tri);
    time("gemv_sym_kernel (all of XX)", [&] { launch_gemv(s, xx, q, vec, out, nullptr, 256); }, full);
    // check the product against the row-streaming kernel
    std::vector<double> a(q), b(q);
    launch_gemv(s, xx, q, vec, out, nullptr, 256); hipMemcpyAsync(a.data(), out, sizeof(double) * q, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    hipLaunchKernelGGL((symgemv_kernel<32>), dim3(528), dim3(256), 0, s, xx, vec, P); hipLaunchKernelGGL((symgemv_sum_kernel<32>), dim3(32), dim3(128), 0, s, P, out);
    hipMemcpyAsync(b.data(), out, sizeof(double) * q, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
    double md = 0; for (int i = 0; i < q; ++i) md = fmax(md, fabs(a[i] - b[i]));
    printf("max |sym - row| = %.3g\n", md);
    return 0;
}
