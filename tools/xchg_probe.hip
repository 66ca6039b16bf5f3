// xchg_probe.hip -- the floor of one all-gather between cooperating workgroups on MI355X (path_coop.hip / path_wcoop.hip pay one or
// two per iteration): G workgroups of 256 threads, N rows in all, every workgroup publishes its N / G rows as data-tagged granule
// pairs {tag, lo} {tag, hi} and gathers everybody else's; nothing else in the loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/xchg_probe tools/xchg_probe.hip && tools/xchg_probe
// Modes: 0  two 8-byte atomic stores / loads per row, three sweeps in flight (what the engines do)
//        1  the same, one sweep in flight
//        2  one 16-byte store / load per row (buffer_*_dwordx4 sc1), three sweeps in flight
//        3  the same, one sweep in flight
//        5 / 6  as 3, the first sweep delayed by s_sleep 6 / 12 (384 / 768 cycles): polls issued before anything can have landed are
//           wasted traffic, and the one in flight when the data lands costs a whole round trip (measured: 0-8 % in this symmetric
//           loop; NOT adopted in the engines -- there a workgroup that arrives late would sleep on everybody's critical path)
//        7 / 8  as 3 / 6 with FOUR replicas of the published rows (a reader takes replica wg % 4): fewer pollers per cache line --
//           round 4, for path_wres_kernel's 209 workgroups
//        9 / 10  as 3 / 2 with PLAIN stores (wave scope: the line stays in the writer XCD's L2) and sc1 loads -- only meaningful between
//           workgroups of ONE XCD (stride 8); round 5: what an engine confined to one XCD would pay per exchange
//        11  as 9 with nt stores and nt loads
//        4  16-byte rows + a compact flag word per workgroup (stored after its rows): poll the G flags, then read the rows once
//           (every row still validated by its own tags, re-read if a flag overtook it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
constexpr int NTH = 256, ROUNDS = 3000, EPT = 2;              // N <= 512 rows

template <int MODE>
__global__ __launch_bounds__(NTH) void allgather(unsigned long long *buf, unsigned long long *flags, int N, unsigned long long *out, int stride)
{
    if (blockIdx.x % stride) return;
    const int tid = threadIdx.x, wg = blockIdx.x / stride, G = gridDim.x / stride, RW = (N + G - 1) / G;
    __shared__ double sh[512];
    constexpr int REP = (MODE == 7 || MODE == 8) ? 4 : 1;
    constexpr int STAUX = (MODE == 9 || MODE == 10) ? 0 : (MODE == 11 ? 2 : 16), LDAUX = MODE == 11 ? 2 : 16;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, 2 * N * 16 * REP, 0x00020000);
    const int myrep = wg % REP;
    bool need[EPT], own[EPT];
    for (int k = 0; k < EPT; ++k) { const int row = tid + NTH * k; own[k] = row < N && row / RW == wg; need[k] = row < N && !own[k]; }
    unsigned long long bad = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned ep = 1; ep <= (unsigned)ROUNDS; ++ep) {
        const int par = ep & 1;
        gu64 *base = (gu64 *)buf + (size_t)par * N * 2;
        // publish
#pragma unroll
        for (int k = 0; k < EPT; ++k) if (own[k]) {
            const int row = tid + NTH * k;
            const double val = (double)ep * 0.5 + row;
            const unsigned lo = (unsigned)__double2loint(val), hi = (unsigned)__double2hiint(val);
            if (MODE <= 1) {
                __hip_atomic_store(base + (size_t)row * 2, ((unsigned long long)ep << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(base + (size_t)row * 2 + 1, ((unsigned long long)ep << 32) | hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                v4u v; v.x = lo; v.y = ep; v.z = hi; v.w = ep;
#pragma unroll
                for (int r = 0; r < REP; ++r) __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((par * REP + r) * N + row) * 16, 0, STAUX);
            }
        }
        if (MODE == 4) {
            // the flag of this workgroup after its rows (no fence: the rows validate themselves)
            if (tid == 0) __hip_atomic_store((gu64 *)flags + par * 256 + wg, (unsigned long long)ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            for (;;) {
                const unsigned long long f = tid < G ? __hip_atomic_load((gu64 *)flags + par * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ep;
                if (__syncthreads_and(f == ep)) break;
                if (++spins > 1000000u) { bad = 1; break; }
            }
        }
        // gather
        constexpr int NS = (MODE == 0 || MODE == 2 || MODE == 10) ? 3 : 1;
        if (MODE == 5) __builtin_amdgcn_s_sleep(6);
        if (MODE == 6 || MODE == 8) __builtin_amdgcn_s_sleep(12);
        v4u pv[NS][EPT];
        auto issue = [&](int s) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                pv[s][k] = v4u{0, 0, 0, 0};
                if (need[k]) {
                    const int row = tid + NTH * k;
                    if (MODE <= 1) {
                        const unsigned long long a = __hip_atomic_load(base + (size_t)row * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long b = __hip_atomic_load(base + (size_t)row * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pv[s][k] = v4u{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
                    } else pv[s][k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((par * REP + myrep) * N + row) * 16, 0, LDAUX);
                }
            }
        };
        v4u got[EPT];
        unsigned spins = 0;
        bool ok = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) issue(s);
        while (!ok) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (ok) break;
                bool all = true;
#pragma unroll
                for (int k = 0; k < EPT; ++k) if (need[k]) all &= (pv[s][k].y == ep) & (pv[s][k].w == ep);
                if (__all(all)) {
#pragma unroll
                    for (int k = 0; k < EPT; ++k) got[k] = pv[s][k];
                    ok = true;
                } else {
                    if (++spins > 1000000u) { bad = 1; ok = true; }
                    issue(s);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) if (need[k]) {
            const int row = tid + NTH * k;
            const double v = __hiloint2double((int)got[k].z, (int)got[k].x);
            if (v != (double)ep * 0.5 + row) bad |= 2;
            sh[row] = v;
        }
        __syncthreads();
        if (bad) break;                                                // (a mode that cannot work here: do not spin through every round)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (wg == 0 && tid == 0) out[0] = t1 - t0;
    if (bad) atomicOr(out + 1, bad);
}

template <int MODE> void run(unsigned long long *buf, unsigned long long *flags, unsigned long long *out, int G, int N, int stride)
{
    CK(hipMemset(buf, 0, 2 * 512 * 16 * 4)); CK(hipMemset(flags, 0, 2 * 256 * 8)); CK(hipMemset(out, 0, 16));
    hipLaunchKernelGGL((allgather<MODE>), dim3(G * stride), dim3(NTH), 0, 0, buf, flags, N, out, stride);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    printf("  G=%3d N=%3d stride %d mode %d: %7.1f ns per all-gather%s\n", G, N, stride, MODE, (double)h[0] * 10.0 / ROUNDS, h[1] ? "  (TIMEOUT / WRONG DATA)" : "");
}

int main()
{
    unsigned long long *buf, *flags, *out;
    CK(hipMalloc(&buf, 2 * 512 * 16 * 4)); CK(hipMalloc(&flags, 2 * 256 * 8)); CK(hipMalloc(&out, 16));
    const int gs[] = {2, 4, 8, 16, 32, 64};
    for (int N : {512, 128})
        for (int G : gs) {
            run<0>(buf, flags, out, G, N, 1); run<1>(buf, flags, out, G, N, 1); run<2>(buf, flags, out, G, N, 1);
            run<3>(buf, flags, out, G, N, 1); run<5>(buf, flags, out, G, N, 1); run<6>(buf, flags, out, G, N, 1); run<4>(buf, flags, out, G, N, 1);
            if (G <= 32) { run<0>(buf, flags, out, G, N, 8); run<2>(buf, flags, out, G, N, 8); run<3>(buf, flags, out, G, N, 8); run<9>(buf, flags, out, G, N, 8); run<10>(buf, flags, out, G, N, 8); run<11>(buf, flags, out, G, N, 8); }
        }
    // round 4: the workgroup counts of path_wres_kernel (p >= n with columns in the accumulator file)
    for (int N : {500, 128})
        for (int G : {64, 128, 167, 209, 240}) { run<3>(buf, flags, out, G, N, 1); run<6>(buf, flags, out, G, N, 1); run<7>(buf, flags, out, G, N, 1); run<8>(buf, flags, out, G, N, 1); }
    return 0;
}
