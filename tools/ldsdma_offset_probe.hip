// Does the instruction offset of global_load_lds_dwordx4 also move the LDS destination?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ldsdma_offset_probe tools/ldsdma_offset_probe.hip && ./tools/ldsdma_offset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const double __attribute__((address_space(1))) *gptr_t;
template <int OFF>
__global__ void k(const double *x, double *o)
{
    __shared__ __attribute__((aligned(16))) double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.0;
    __syncthreads();
    gptr_t p = (gptr_t)x + 512 + threadIdx.x * 2;                 // global address BEFORE the instruction offset
    unsigned base = (unsigned)(size_t)lds + 2048;                  // M0: 2 KiB into the array
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%2\n\ts_waitcnt vmcnt(0)"
                 :: "s"(base), "v"(p), "i"(OFF) : "m0", "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) o[i] = lds[i];
}
template <int OFF> void run()
{
    double *x, *o, hx[2048], ho[1024];
    for (int i = 0; i < 2048; ++i) hx[i] = i;
    (void)hipMalloc(&x, sizeof(hx)); (void)hipMalloc(&o, sizeof(ho));
    (void)hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<OFF>, dim3(1), dim3(64), 0, 0, x, o);
    (void)hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int first = -1; double val = 0;
    for (int i = 0; i < 1024; ++i) if (ho[i] != -1.0) { first = i; val = ho[i]; break; }
    printf("offset %5d: first written LDS double index %d (M0 points at 256), value %.0f (pointer without offset -> 512)\n", OFF, first, val);
}
int main() { run<0>(); run<1024>(); run<-1024>(); return 0; }
