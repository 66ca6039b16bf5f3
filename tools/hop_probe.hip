// hop_probe.hip -- what one flag hop between two workgroups costs on MI355X, by cache-policy bits of the store and of the polling load,
// for a partner on the same XCD and on another one (the cooperating-workgroup engines pay one or two such hops per iteration).
//   hipcc --offload-arch=gfx950 -O3 -o hop_probe tools/hop_probe.hip && ./hop_probe
// Ping-pong: A stores k to fa, B polls fa == k then stores k to fb, A polls fb == k; ROUNDS times; one hop = time / (2 ROUNDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); exit(1); } } while (0)
constexpr int ROUNDS = 2000, NWG = 64;

#define LOADER(NAME, MODS)                                                                                             \
    __device__ __forceinline__ unsigned NAME(const unsigned *p)                                                        \
    {                                                                                                                  \
        unsigned v;                                                                                                    \
        asm volatile("global_load_dword %0, %1, off " MODS "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");     \
        return v;                                                                                                      \
    }
#define STORER(NAME, MODS)                                                                                             \
    __device__ __forceinline__ void NAME(unsigned *p, unsigned v)                                                      \
    {                                                                                                                  \
        asm volatile("global_store_dword %0, %1, off " MODS : : "v"(p), "v"(v) : "memory");                             \
    }
LOADER(ld_plain, "") LOADER(ld_sc0, "sc0") LOADER(ld_sc1, "sc1") LOADER(ld_sc01, "sc0 sc1") LOADER(ld_nt, "nt")
STORER(st_plain, "") STORER(st_sc0, "sc0") STORER(st_sc1, "sc1") STORER(st_sc01, "sc0 sc1") STORER(st_nt, "nt")

template <int L> __device__ __forceinline__ unsigned ld(const unsigned *p)
{
    if constexpr (L == 0) return ld_plain(p); else if constexpr (L == 1) return ld_sc0(p); else if constexpr (L == 2) return ld_sc1(p);
    else if constexpr (L == 3) return ld_sc01(p); else return ld_nt(p);
}
template <int S> __device__ __forceinline__ void st(unsigned *p, unsigned v)
{
    if constexpr (S == 0) st_plain(p, v); else if constexpr (S == 1) st_sc0(p, v); else if constexpr (S == 2) st_sc1(p, v);
    else if constexpr (S == 3) st_sc01(p, v); else st_nt(p, v);
}

template <int L, int S>
__global__ void pingpong(unsigned *flags, int a, int b, unsigned long long *out, int *xcc)
{
    const int wg = blockIdx.x;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[wg] = (int)(id & 0xf);
    }
    if (wg != a && wg != b) return;
    if (threadIdx.x != 0) return;
    unsigned *fa = flags, *fb = flags + 64;                  // different 128-byte lines
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned fails = 0;
    for (unsigned k = 1; k <= (unsigned)ROUNDS; ++k) {
        if (wg == a) {
            st<S>(fa, k);
            unsigned spins = 0;
            while (ld<L>(fb) != k) if (++spins > 200000u) { ++fails; break; }
        } else {
            unsigned spins = 0;
            while (ld<L>(fa) != k) if (++spins > 200000u) { ++fails; break; }
            st<S>(fb, k);
        }
        if (fails) break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (wg == a) { out[0] = t1 - t0; out[1] = fails; }
}

template <int L, int S> void run(const char *ln, const char *sn, unsigned *flags, unsigned long long *out, int *xcc, int a, int b, const char *what)
{
    CK(hipMemset(flags, 0, 1024));
    CK(hipMemset(out, 0, 16));
    hipLaunchKernelGGL((pingpong<L, S>), dim3(NWG), dim3(64), 0, 0, flags, a, b, out, xcc);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    if (h[1]) printf("  load %-8s store %-8s %-12s: no progress (stale line)\n", ln, sn, what);
    else printf("  load %-8s store %-8s %-12s: %7.1f ns per hop\n", ln, sn, what, (double)h[0] * 10.0 / (2.0 * ROUNDS));
}

int main()
{
    unsigned *flags; unsigned long long *out; int *xcc;
    CK(hipMalloc(&flags, 1024)); CK(hipMalloc(&out, 16)); CK(hipMalloc(&xcc, NWG * sizeof(int)));
    CK(hipMemset(xcc, 0xff, NWG * sizeof(int)));
    run<3, 3>("sc0 sc1", "sc0 sc1", flags, out, xcc, 0, 1, "warm-up");
    std::vector<int> hx(NWG);
    CK(hipMemcpy(hx.data(), xcc, NWG * sizeof(int), hipMemcpyDeviceToHost));
    printf("XCC of workgroups 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
    printf("\n");
    int same = -1, other = -1;
    for (int i = 1; i < NWG; ++i) { if (hx[i] == hx[0] && same < 0) same = i; if (hx[i] != hx[0] && other < 0) other = i; }
    printf("partner on the same XCD: workgroup %d; on another XCD: workgroup %d\n", same, other);
    const int pr[2] = {same, other};
    const char *nm[2] = {"same XCD", "other XCD"};
    for (int w = 0; w < 2; ++w) {
        if (pr[w] < 0) continue;
        const int b = pr[w];
#define RUN(L, S, LN, SN) run<L, S>(LN, SN, flags, out, xcc, 0, b, nm[w])
        RUN(2, 2, "sc1", "sc1"); RUN(3, 3, "sc0 sc1", "sc0 sc1"); RUN(2, 0, "sc1", "plain"); RUN(2, 1, "sc1", "sc0");
        RUN(1, 1, "sc0", "sc0"); RUN(1, 0, "sc0", "plain"); RUN(1, 2, "sc0", "sc1");
        RUN(4, 4, "nt", "nt"); RUN(4, 0, "nt", "plain"); RUN(0, 0, "plain", "plain"); RUN(3, 0, "sc0 sc1", "plain");
    }
    // what the engines use today: relaxed agent-scope atomics
    return 0;
}
