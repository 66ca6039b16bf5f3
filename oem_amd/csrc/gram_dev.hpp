// gram_dev.hpp -- device-side pieces shared by the moment kernels of gram.hip (triangle / ring / block forms) and gram_sb.hip (the
// shared-slab form): the shift predicate, the asm-owned accumulator tiles, slab registers, LDS-DMA and counted waits.  Everything here is
// __forceinline__ device code or a template: each translation unit gets its own copy (no relocatable device code in this build).
#pragma once
#include <type_traits>
#include "common.hpp"

namespace oemgpu {

// ------------------------------------------------------------------------------------------------
// Is the provisional shift worth its price?  Every x - c is an FP64 VALU op, and FP64 VALU shares the DP units with
// the FP64 MFMA (tools/mfma_probe.hip: each v_fma_f64 between MFMAs costs 4.5-9 MFMA cycles).  Un-shifted
// accumulation loses (mean/sd)^2 * eps of relative accuracy on the centred moments, so the shift is applied only
// when some sampled column has |mean| / sd > 16 (worst un-shifted loss 2^8 eps ~ 6e-14: the level of the
// summation rounding itself).  The decision is a pure function of the (all-reduced) sample sums, so every kernel, workgroup
// and rank takes it identically.  sums layout: [0..p] sum z_j (x columns, then y), [p+1] sample count,
// [p+2 .. 2p+2] sum z_j^2.
__device__ __forceinline__ bool column_needs_shift(const double *__restrict__ sums, int p, int j)
{
    const double cnt = sums[p + 1], m = sums[j] / cnt;
    double var = sums[p + 2 + j] / cnt - m * m;
    if (!(var > 0.0)) var = 0.0;
    return m * m > 256.0 * var;
}
__device__ __forceinline__ bool shift_needed_wave(const double *__restrict__ sums, int p)      // wave-uniform result
{
    if (!sums) return false;
    bool need = false;
    for (int j = threadIdx.x & 63; j <= p; j += 64) need |= column_needs_shift(sums, p, j);
    return __any(need);
}

// ------------------------------------------------------------------------------------------------
// MFMA Gram body
// ------------------------------------------------------------------------------------------------
#include "gen/acc_tiles.inc"

template <int N, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Explicit global address space: through the per-fragment pointer arrays hipcc otherwise falls back to flat_load,
// whose out-of-order return makes it wait vmcnt(0) + lgkmcnt(0) and drains the prefetch pipeline.
typedef const double __attribute__((address_space(1))) *gptr_t;
typedef const v2d __attribute__((address_space(1))) *gptr2_t;

// One 8-row slab of one wave: lane (i, q) holds rows r+2q, r+2q+1 of each fragment's column.
template <int NF> struct Slab {
    v2d v[NF];
    v2d y;
};

template <int NF, bool ALIGNED, bool MASKED, bool LOADY>
__device__ __forceinline__ void load_slab(Slab<NF> &s, const gptr_t (&ptr)[NF], gptr_t y, int64_t r, int64_t n)
{
    if (!MASKED) {
        if (ALIGNED) {
#pragma unroll
            for (int f = 0; f < NF; ++f) s.v[f] = *(gptr2_t)(ptr[f] + r);
            if (LOADY) s.y = *(gptr2_t)(y + r);
        } else {
#pragma unroll
            for (int f = 0; f < NF; ++f) { s.v[f].x = ptr[f][r]; s.v[f].y = ptr[f][r + 1]; }
            if (LOADY) { s.y.x = y[r]; s.y.y = y[r + 1]; }
        }
    } else {
        const int64_t r0 = r < n ? r : n - 1, r1 = r + 1 < n ? r + 1 : n - 1;
#pragma unroll
        for (int f = 0; f < NF; ++f) { s.v[f].x = ptr[f][r0]; s.v[f].y = ptr[f][r1]; }
        if (LOADY) { s.y.x = y[r0]; s.y.y = y[r1]; }
    }
}

// Hot-loop loads as inline asm: hipcc then keeps no count of them, and the waits below are exact.  (With compiler
// loads it folds every loop form tried back into one with conditional prefetches and then waits for all but the
// newest slab, which halves the prefetch distance.)  Form (ii) of the guide's asm rules: "=v" loads, then one wait
// statement naming every destination "+v" before the first consumer.
template <int OFF> __device__ __forceinline__ void gload16(v2d &dst, gptr_t p)
{
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "i"(OFF * 8) : "memory");
}
template <int OFF> __device__ __forceinline__ double gload8(gptr_t p)
{
    double dst;
    asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "i"(OFF * 8) : "memory");
    return dst;
}
// loads issued per slab (for the vmcnt arithmetic)
template <int NF, bool ALIGNED, bool LOADY> struct SlabLoads { static constexpr int N = (ALIGNED ? 1 : 2) * (NF + (LOADY ? 1 : 0)); };

template <int NF, bool ALIGNED, bool LOADY, int OFF>
__device__ __forceinline__ void load_slab_asm(Slab<NF> &s, const gptr_t (&ptr)[NF], gptr_t y)
{
    if (ALIGNED) {
#pragma unroll
        for (int f = 0; f < NF; ++f) gload16<OFF>(s.v[f], ptr[f]);
        if (LOADY) gload16<OFF>(s.y, y);
    } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) { s.v[f].x = gload8<OFF>(ptr[f]); s.v[f].y = gload8<OFF + 1>(ptr[f]); }
        if (LOADY) { s.y.x = gload8<OFF>(y); s.y.y = gload8<OFF + 1>(y); }
    }
}
// wait until at most PENDING younger loads are outstanding; ties the slab's registers to the wait
template <int NF, int PENDING> __device__ __forceinline__ void wait_slab(Slab<NF> &s)
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PENDING) : "memory");
#pragma unroll
    for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(s.v[f]));
    asm volatile("" : "+v"(s.y));
    __builtin_amdgcn_sched_barrier(0);
}

// LDS-DMA (global_load_lds_dwordx4): 64 lanes x 16 B land at (wave-uniform LDS address in M0) + 16 lane, with no VGPR
// destination, so prefetch depth is bounded by LDS, not by registers.
// Steady-state form: M0 is set ONCE per slab and the seven fragments are told apart by the instruction offset, which
// (tools/ldsdma_offset_probe.hip) moves the global address AND the LDS destination by the same number of bytes -- so
// the source pointer is pre-decremented by it.  The M0 save / set / restore dance per DMA cost 12 cycles each.
__device__ __forceinline__ void set_m0(unsigned lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_byte_addr) : "memory");
}
#ifndef OEM_GLDS_POLICY
#define OEM_GLDS_POLICY ""                  // cache policy of the slab DMAs (" nt": experiment knob)
#endif
template <int OFFB> __device__ __forceinline__ void glds_v(gptr_t src)               // 64-bit per-lane address
{
    asm volatile("global_load_lds_dwordx4 %0, off offset:%1" OEM_GLDS_POLICY ::"v"(src), "i"(OFFB) : "memory");
}
template <int OFFB> __device__ __forceinline__ void glds_s(unsigned voff, gptr_t sbase)   // scalar base + 32-bit lane offset
{
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" OEM_GLDS_POLICY ::"v"(voff), "s"(sbase), "i"(OFFB) : "memory");
}
template <int PENDING> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PENDING) : "memory");
}

// VALU-side accumulators (block kernel only; the triangle kernel gets X'y and the column sums from the MFMAs
// by treating y and a column of ones as columns p and p+1 of the matrix).
template <int NF> struct VecAcc {
    double sx[NF], sxy[NF];
    double sy, syy;
};

// Per-lane load-time transform: v = x * m + o.  Ordinary column: m = 1, o = -c (exact x - c).
// Ones column (AUG, col == p+1): m = 0, o = 1.
template <int NF> struct LaneXf {
    double c[NF];       // shift of the fragment's column (0 for y/ones handled through m_last/o_last)
    double m_last, o_last;
};

// Columns beyond the last valid one need no masking: a fragment lane only feeds the tile rows / columns of
// its own column index, and entries outside the matrix are dropped by moments_reduce_kernel.
struct NoHook { template <typename M> __device__ __forceinline__ void operator()(M) const {} };


__device__ __forceinline__ double shfl_xor_d(double v, int m)
{
    return __shfl_xor(v, m, 64);
}

// Pointers are separate kernel parameters (not struct members): only then does hipcc know they are global
// and emit global_load (counted vmcnt) instead of flat_load (vmcnt(0) + lgkmcnt(0) drains the prefetch).
struct GramDims {
    int64_t n; int64_t ld; int p;
    int ntc, ntile, nblk, nchunk, steps;
};

// the shared-slab launch (gram_sb.hip) behind launch_gram
int launch_gram_sb(hipStream_t s, const GramPlan &pl, const double *x, const double *y, const double *sums, double *tpart, double *vpart, const GramDims &a);
int launch_gram_wd(hipStream_t s, const GramPlan &pl, const double *x, const double *y, const double *sums, double *tpart, double *vpart, const GramDims &a);   // gram_wd.hip: 15-16 tile columns, one read of X
constexpr int SB_KINDS = 7;                      // kinds of super-block, in launch order (gram_sb_kernel)

}  // namespace oemgpu
