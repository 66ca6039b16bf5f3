// gram.hip -- one-pass moment build for tall column-major X on gfx950 (CDNA4).
//
// Replaces, with ONE streaming pass over X:
//   DataStd::standardize   ref src/DataStd.h:203-265   (column means / scales, never materialised)
//   XY = X'Y / n           ref src/oem_dense.h:699-707
//   XtX()                  ref src/oem_dense.h:318-361  (lower rank-n update), row slices ref src/oem_big.h:319-361
//   per-column sums        ref src/oem_big.h:743-841
//
// Layout trick (no LDS, no transpose): v_mfma_f64_16x16x4_f64 takes A[i][k] from lane i+16k and
// B[k][j] from lane j+16k, one f64 per lane.  For a Gram tile G[I][J] = sum_rows x[:,16I+i] x[:,16J+j]
// the SAME register fragment f_T(lane) = x[row(k)][16T + (lane&15)] serves as the A operand of every
// tile in tile-row T and the B operand of every tile in tile-column T, and the identity of the four
// rows occupying the k slots of one MFMA is irrelevant as long as all fragments of that step agree.
// So lane (i, q=lane>>4) loads 16 B = rows {r0+2q, r0+2q+1} of column 16T+i: the four q-lanes of a
// column read 64 contiguous bytes, every fetched byte is used, and the two halves of the dwordx4
// feed two consecutive MFMA k-steps.  Each 4-wave workgroup walks 64-row steps (16 rows per wave, so
// a workgroup touches 512 contiguous bytes of every column per step).
//
// Centring: data are shifted by a provisional mean c (from a strided row sample) while loading, and the
// exact centred moments follow from  C_ij = M_ij - s_i s_j / n  (M, s = shifted moments / sums).  The
// correction is O(sample error^2), so there is no cancellation even when |mean| >> sd.
#include <cstdlib>
#include "common.hpp"
#include "gram_dev.hpp"
#include <vector>

#include <type_traits>

namespace oemgpu {

// ------------------------------------------------------------------------------------------------
// provisional shift: sums over <= 256 evenly spaced 16-row chunks (all rows when n <= 4096)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shift_sums_kernel(const double *__restrict__ x, int64_t n, int64_t ld, int p,
                                                          const double *__restrict__ y, double *__restrict__ sums)
{
    // one workgroup per column: 256 threads x 16 consecutive rows (a full chunk is two 64-byte loads per thread, all in
    // flight at once), then ONE fixed-order tree for the three sums together (sum, sum of squares, count)
    __shared__ double sh[3][256];
    const int j = blockIdx.x;
    const double *col = (j < p) ? x + (size_t)j * ld : y;
    const int64_t nch = (n + 15) / 16;
    const int64_t nsamp = nch < 256 ? nch : 256;
    const int k = threadIdx.x;
    double s = 0.0, ss = 0.0, cnt = 0.0;
    if (k < nsamp) {
        const int64_t c = (nsamp > 1) ? ((int64_t)k * (nch - 1)) / (nsamp - 1) : 0;
        const int64_t r0 = c * 16;
        if (r0 + 16 <= n) {
            double v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = col[r0 + r];
#pragma unroll
            for (int r = 0; r < 16; ++r) { s += v[r]; ss = fma(v[r], v[r], ss); }
            cnt = 16.0;
        } else {
            for (int64_t r = r0; r < n; ++r) { const double v = col[r]; s += v; ss = fma(v, v, ss); cnt += 1.0; }
        }
    }
    sh[0][k] = s; sh[1][k] = ss; sh[2][k] = cnt;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (k < w) { sh[0][k] += sh[0][k + w]; sh[1][k] += sh[1][k + w]; sh[2][k] += sh[2][k + w]; }
        __syncthreads();
    }
    if (k == 0) {
        sums[j] = sh[0][0];
        sums[p + 2 + j] = sh[1][0];
        if (j == 0) { sums[p + 1] = sh[2][0]; sums[2 * p + 3] = 0.0; }
    }
}

int launch_shift_sums(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, double *sums)
{
    hipLaunchKernelGGL(shift_sums_kernel, dim3(p + 1), dim3(256), 0, s, x, n, ld, p, y, sums);
    OEM_HIP(hipGetLastError());
    return 0;
}

// hook(integral_constant<m>) runs right after the m-th MFMA of the slab (m = 0 .. 2 NTILES - 1): the place for
// scalar / VMEM / LDS instructions, which issue for free in the 64-cycle shadow of an FP64 MFMA.
#ifndef OEM_GRAM_EXP
#define OEM_GRAM_EXP 0
#endif
template <int NR, int NC, bool DIAG, bool MASKED, bool VEC, bool AUG, bool XF = true, typename Hook = NoHook>
__device__ __forceinline__ void consume_slab(VecAcc<DIAG ? NR : NR + NC> &V, Slab<DIAG ? NR : NR + NC> &s,
                                             const LaneXf<DIAG ? NR : NR + NC> &X, double cy, int64_t r, int64_t n,
                                             Hook &&hook = NoHook())
{
    constexpr int NF = DIAG ? NR : NR + NC;
    double m0 = 1.0, m1 = 1.0;
    if (MASKED) {
        m0 = (r < n) ? 1.0 : 0.0;
        m1 = (r + 1 < n) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (!XF || OEM_GRAM_EXP == 5) {
            // un-shifted form: the operands are used as loaded (the ones column is read from a constant)
        } else if (AUG && f == NF - 1) {
            s.v[f].x = fma(s.v[f].x, X.m_last, X.o_last);
            s.v[f].y = fma(s.v[f].y, X.m_last, X.o_last);
        } else {
            s.v[f].x -= X.c[f];
            s.v[f].y -= X.c[f];
        }
        if (MASKED) { s.v[f].x *= m0; s.v[f].y *= m1; }
    }
    if (VEC && OEM_GRAM_EXP != 5) {
        double y0 = s.y.x - cy, y1 = s.y.y - cy;
        if (MASKED) { y0 *= m0; y1 *= m1; }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            V.sx[f] = (V.sx[f] + s.v[f].x) + s.v[f].y;
            V.sxy[f] = fma(s.v[f].x, y0, V.sxy[f]);
            V.sxy[f] = fma(s.v[f].y, y1, V.sxy[f]);
        }
        V.sy = (V.sy + y0) + y1;
        V.syy = fma(y0, y0, V.syy);
        V.syy = fma(y1, y1, V.syy);
    }
    // All VALU writes of MFMA operands are above this point.  The MFMAs are inline asm on asm-owned AGPR tiles
    // (gen/acc_tiles.inc); hipcc pads no hazards for asm, so the VALU-write -> MFMA-read wait states are explicit.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3" ::: "memory");
    static_for<2>([&](auto E) {
        constexpr int e = decltype(E)::value;
        if constexpr (DIAG) {
            static_for<NR>([&](auto I_) {
                constexpr int I = decltype(I_)::value;
                static_for<I + 1>([&](auto J_) {
                    constexpr int J = decltype(J_)::value;
                    AccTile<I *(I + 1) / 2 + J>::mfma(s.v[I][e], s.v[J][e]);
                    hook(std::integral_constant<int, e * (NR * (NR + 1) / 2) + I * (I + 1) / 2 + J>{});
                });
            });
        } else {
            static_for<NR>([&](auto I_) {
                constexpr int I = decltype(I_)::value;
                static_for<NC>([&](auto J_) {
                    constexpr int J = decltype(J_)::value;
                    AccTile<I * NC + J>::mfma(s.v[I][e], s.v[NR + J][e]);
                    hook(std::integral_constant<int, e * (NR * NC) + I * NC + J>{});
                });
            });
        }
    });
    __builtin_amdgcn_sched_barrier(0);
}

// One workgroup (4 waves) builds the tiles {(I0+a, J0+b)} over `steps` 64-row steps starting at row_begin.
// AUG: the matrix is Z = [X | y | 1] (p+2 columns); otherwise X only, with X'y / sums on the VALU (VEC).
// tdst: this chunk's tile partials [ntile_total][256]; vdst: this chunk's vector partials.
template <int NR, int NC, bool DIAG, bool ALIGNED, bool VEC, bool AUG>
__device__ __forceinline__ void gram_body(const double *__restrict__ x, int64_t n, int64_t ld, int p,
                                          const double *__restrict__ y, const double *__restrict__ sums, int ntc,
                                          int I0, int J0, int64_t row_begin, int steps, bool do_scalar,
                                          double *__restrict__ tdst, double *__restrict__ vdst, double *lds)
{
    constexpr int NF = DIAG ? NR : NR + NC;
    constexpr int NTILES = DIAG ? NR * (NR + 1) / 2 : NR * NC;
    // w must be provably wave-uniform or every loop below turns into exec-masked vector control flow
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, q = lane >> 4;

    const gptr_t xg = (gptr_t)x, yg = (gptr_t)y;
    gptr_t ptr[NF];
    LaneXf<NF> X;
    X.m_last = 1.0; X.o_last = 0.0;
    const bool use_shift = shift_needed_wave(sums, p);
    const double inv_cnt = use_shift ? 1.0 / sums[p + 1] : 0.0;
    const double cy = use_shift ? sums[p] * inv_cnt : 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int T = (DIAG || f < NR) ? I0 + f : J0 + (f - NR);
        const int col = 16 * T + i;
        double cf;
        if (AUG) {
            if (col < p) { ptr[f] = xg + (size_t)col * ld; cf = use_shift ? sums[col] * inv_cnt : 0.0; }
            else { ptr[f] = yg; cf = cy; }
            if (f == NF - 1) {
                X.m_last = (col == p + 1) ? 0.0 : 1.0;
                X.o_last = (col == p + 1) ? 1.0 : -cf;
            }
        } else {
            const int cc = col < p ? col : p - 1;
            ptr[f] = xg + (size_t)cc * ld;
            cf = use_shift ? sums[cc] * inv_cnt : 0.0;
        }
        X.c[f] = cf;
    }

    VecAcc<NF> V;
#pragma unroll
    for (int f = 0; f < NF; ++f) { V.sx[f] = 0.0; V.sxy[f] = 0.0; }
    V.sy = 0.0; V.syy = 0.0;
    static_for<NTILES>([&](auto T_) { AccTile<decltype(T_)::value>::zero(); });
    asm volatile("s_nop 7" ::: "memory");

    // Row ownership: the workgroup walks 32-row steps; wave w owns the 8-row slab 8 w of each, so a workgroup
    // touches 256 contiguous bytes of every column per step and a wave's slabs are evenly strided (32 rows):
    // one pointer bump per three slabs, everything else is an immediate offset.
    // Full slabs (row + 8 <= n) form a prefix; at most one ragged slab follows.
    const int nslab = 2 * steps;
    const int64_t w0 = row_begin + 8 * w;
    int ns = 0;
    if (w0 + 8 <= n) {
        int64_t k = (n - w0 - 8) / 32 + 1;
        ns = k < nslab ? (int)k : nslab;
    }
    gptr_t cur[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) cur[f] = ptr[f] + w0 + 2 * q;
    gptr_t ycur = yg + w0 + 2 * q;
    Slab<NF> s0, s1, s2;
    s0.y = v2d{0.0, 0.0}; s1.y = s0.y; s2.y = s0.y;
    constexpr int LPS = SlabLoads<NF, ALIGNED, VEC>::N;        // vmcnt units per slab
    static_assert(2 * LPS <= 63, "vmcnt field is 6 bits");
    int k = 0;
    if (ns > 0) load_slab_asm<NF, ALIGNED, VEC, 0>(s0, cur, ycur);
    if (ns > 1) load_slab_asm<NF, ALIGNED, VEC, 32>(s1, cur, ycur);
    // steady state: slab k is consumed with slabs k+1 and k+2 in flight (two slabs = 2 x 3584 MFMA cycles of cover)
    while (k + 5 <= ns) {
        load_slab_asm<NF, ALIGNED, VEC, 64>(s2, cur, ycur);
        wait_slab<NF, 2 * LPS>(s0);
        consume_slab<NR, NC, DIAG, false, VEC, AUG>(V, s0, X, cy, 0, n);
        load_slab_asm<NF, ALIGNED, VEC, 96>(s0, cur, ycur);
        wait_slab<NF, 2 * LPS>(s1);
        consume_slab<NR, NC, DIAG, false, VEC, AUG>(V, s1, X, cy, 0, n);
        load_slab_asm<NF, ALIGNED, VEC, 128>(s1, cur, ycur);
        wait_slab<NF, 2 * LPS>(s2);
        consume_slab<NR, NC, DIAG, false, VEC, AUG>(V, s2, X, cy, 0, n);
#pragma unroll
        for (int f = 0; f < NF; ++f) cur[f] += 96;
        ycur += 96;
        k += 3;
    }
    // drain: s0 = slab k and s1 = slab k+1 are in flight (when they exist); at most two more follow
    {
        const int rem = ns - k;
        if (rem >= 3) load_slab_asm<NF, ALIGNED, VEC, 64>(s2, cur, ycur);
        if (rem >= 1) { wait_slab<NF, 0>(s0); consume_slab<NR, NC, DIAG, false, VEC, AUG>(V, s0, X, cy, 0, n); }
        if (rem >= 4) load_slab_asm<NF, ALIGNED, VEC, 96>(s0, cur, ycur);
        if (rem >= 2) { wait_slab<NF, 0>(s1); consume_slab<NR, NC, DIAG, false, VEC, AUG>(V, s1, X, cy, 0, n); }
        if (rem >= 3) { wait_slab<NF, 0>(s2); consume_slab<NR, NC, DIAG, false, VEC, AUG>(V, s2, X, cy, 0, n); }
        if (rem >= 4) { wait_slab<NF, 0>(s0); consume_slab<NR, NC, DIAG, false, VEC, AUG>(V, s0, X, cy, 0, n); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ns < nslab && w0 + 32 * (int64_t)ns < n) {
        const int64_t r = w0 + 32 * (int64_t)ns + 2 * q;
        load_slab<NF, false, true, VEC>(s0, ptr, yg, r, n);
        consume_slab<NR, NC, DIAG, true, VEC, AUG>(V, s0, X, cy, r, n);
    }

    // MFMA result -> any other reader: the 16-pass DGEMM needs up to 18 wait states, invisible to hipcc (asm)
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // ---- vector partials: reduce over the 4 q-lanes of a column, then over the 4 waves (fixed order)
    if (VEC) {
        constexpr int VW = 2 * 16 * NF + 4;     // doubles per wave
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            V.sx[f] += shfl_xor_d(V.sx[f], 16);  V.sx[f] += shfl_xor_d(V.sx[f], 32);
            V.sxy[f] += shfl_xor_d(V.sxy[f], 16); V.sxy[f] += shfl_xor_d(V.sxy[f], 32);
        }
        V.sy += shfl_xor_d(V.sy, 16);   V.sy += shfl_xor_d(V.sy, 32);
        V.syy += shfl_xor_d(V.syy, 16); V.syy += shfl_xor_d(V.syy, 32);
        if (q == 0) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                lds[w * VW + f * 16 + i] = V.sx[f];
                lds[w * VW + 16 * NF + f * 16 + i] = V.sxy[f];
            }
            if (i == 0) { lds[w * VW + 32 * NF] = V.sy; lds[w * VW + 32 * NF + 1] = V.syy; }
        }
        __syncthreads();
        if (w == 0 && q == 0) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int T = I0 + f;
                if (T < ntc) {
                    double a = 0.0, b = 0.0;
                    for (int ww = 0; ww < 4; ++ww) { a += lds[ww * VW + f * 16 + i]; b += lds[ww * VW + 16 * NF + f * 16 + i]; }
                    vdst[16 * T + i] = a;
                    vdst[16 * ntc + 16 * T + i] = b;
                }
            }
            if (i == 0 && do_scalar) {
                double a = 0.0, b = 0.0;
                for (int ww = 0; ww < 4; ++ww) { a += lds[ww * VW + 32 * NF]; b += lds[ww * VW + 32 * NF + 1]; }
                int64_t rows = n - row_begin;
                if (rows < 0) rows = 0;
                if (rows > (int64_t)steps * 64) rows = (int64_t)steps * 64;
                vdst[32 * ntc] = a; vdst[32 * ntc + 1] = b; vdst[32 * ntc + 2] = (double)rows; vdst[32 * ntc + 3] = 0.0;
            }
        }
        __syncthreads();
    }

    // ---- tile partials: ((w3 + w2) + w1) + w0, summed in place in LDS, then one coalesced store per register
    if (w == 3) {
        static_for<NTILES>([&](auto T_) {
            constexpr int tt = decltype(T_)::value;
            static_for<4>([&](auto R_) {
                constexpr int r = decltype(R_)::value;
                lds[(tt * 4 + r) * 64 + lane] = AccTile<tt>::template read<r>();
            });
        });
    }
    __syncthreads();
    for (int src = 2; src >= 1; --src) {
        if (w == src) {
            static_for<NTILES>([&](auto T_) {
                constexpr int tt = decltype(T_)::value;
                static_for<4>([&](auto R_) {
                    constexpr int r = decltype(R_)::value;
                    lds[(tt * 4 + r) * 64 + lane] += AccTile<tt>::template read<r>();
                });
            });
        }
        __syncthreads();
    }
    if (w == 0) {
        static_for<NR>([&](auto I_) {
            constexpr int I = decltype(I_)::value;
            static_for<(DIAG ? I + 1 : NC)>([&](auto J_) {
                constexpr int J = decltype(J_)::value;
                const int gi = I0 + I, gj = J0 + J;
                if (gi < ntc && gj < ntc) {
                    constexpr int tt = DIAG ? I * (I + 1) / 2 + J : I * NC + J;
                    double *dst = tdst + (size_t)(gi * (gi + 1) / 2 + gj) * 256;
                    static_for<4>([&](auto R_) {
                        constexpr int r = decltype(R_)::value;
                        dst[r * 64 + lane] = lds[(tt * 4 + r) * 64 + lane] + AccTile<tt>::template read<r>();
                    });
                }
            });
        });
    }
}

// ------------------------------------------------------------------------------------------------
// Triangle kernel, LDS-DMA ring form (the c1 path).  Measured on MI355X (tools/mfma_probe.hip): a bare stream of
// v_mfma_f64_16x16x4_f64 issues every 64.00 cycles (77 TFLOP/s at the clock held), and EVERY interleaved v_fma_f64
// costs 4.5-9 of those cycles: FP64 VALU and FP64 MFMA share the DP units.  So the kernel wants (a) as little FP64
// VALU as possible -- moving the 6-column remainder strip of p = 100 to VALU FMAs gained nothing -- and (b) no cycle
// of the MFMA stream spent waiting for memory.  (b): slabs arrive by LDS-DMA into a wave-private ring of NSLOT
// slots (no VGPR destination, so depth is bounded by LDS: 4 slabs = 28 KiB per wave in flight), a slab is copied to
// registers one slab ahead of its use, and the only waits are exact vmcnt counts.
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) const double g_ones[2] = {1.0, 1.0};

#ifdef OEM_GRAM_DIAG
__device__ unsigned long long g_gram_diag[8];
#define GSTAMP(slot)                                                                       \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        unsigned long long t__;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        gd[slot] += t__ - gl;                                                              \
        gl = t__;                                                                          \
    } while (0)
#else
#define GSTAMP(slot) do { } while (0)
#endif

// The strip form (SB > 0; un-shifted variant only).  When the last tile row holds at most 4 SB <= 8 real columns
// (p = 100: y, the ones and four x columns -- 6 of 16), its NT tiles cost a quarter of the MFMA work for 3/8 of a
// tile row.  v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks, 16.5 cycles: the same FLOP per cycle) has the
// k index on lane / 16 like the 16x16x4 and the block index on (lane % 16) / 4 (tools/mfma444_layout.hip), so a
// column fragment already IS a valid B operand holding four 4-column sub-blocks; the A operand -- one 4-column
// sub-block of the strip replicated into all four block slots -- is a second read of the ring slot with a permuted
// lane address.  SB NT strip MFMAs replace the NT tile MFMAs of the last row: 3584 -> 3150 cycles per slab at p = 100.
// The ragged last slab keeps the plain tiles; the epilogue adds both into the same partial tiles.
template <int NT, int SB, typename Hook = NoHook>
__device__ __forceinline__ void consume_ring_strip(Slab<NT> &s, const v2d (&ua)[SB > 0 ? SB : 1], Hook &&hook = NoHook())
{
    constexpr int NM16 = (NT - 1) * NT / 2;
    __builtin_amdgcn_sched_barrier(0);
    static_for<2>([&](auto E) {
        constexpr int e = decltype(E)::value;
        static_for<NT - 1>([&](auto I_) {
            constexpr int I = decltype(I_)::value;
            static_for<I + 1>([&](auto J_) {
                constexpr int J = decltype(J_)::value;
                AccTile<I *(I + 1) / 2 + J>::mfma(s.v[I][e], s.v[J][e]);
                hook(std::integral_constant<int, e * NM16 + I * (I + 1) / 2 + J>{});
            });
        });
        static_for<SB>([&](auto A_) {
            constexpr int a = decltype(A_)::value;
            static_for<NT>([&](auto T_) {
                constexpr int T = decltype(T_)::value;
                AccStrip<a * NT + T>::mfma(ua[a][e], s.v[T][e]);
            });
        });
    });
    __builtin_amdgcn_sched_barrier(0);
}

// Diagonal tiles as 4x4 sub-blocks (un-shifted ring form, NT >= 6).  A 16x16x4 MFMA on a diagonal tile computes 256 entries of which 136
// are wanted.  The tile is a 4 x 4 grid of 4x4 sub-blocks; v_mfma_f64_4x4x4_4b_f64 multiplies four independent block pairs in a
// quarter of the time, and with the B operand the plain column fragment (block slot b = columns 4b .. 4b + 3) and the A operand the
// SAME fragment rotated left by 4 r lanes inside each 16-lane row (slot b then holds columns 4 ((b + r) mod 4) ...), rotation r yields
// the sub-blocks ((b + r) mod 4, b): r = 0 the four diagonal ones, r = 1 (1,0) (2,1) (3,2) and (0,3) = (3,0) transposed, r = 2 (2,0)
// (3,1) and two duplicates -- all ten lower sub-blocks in 3 x 16.5 cycles instead of 64.  The rotated operands are two more reads of
// the ring slot with a permuted lane address; the three accumulators live in the tile's own AGPRs (a[8T .. 8T + 5]).  In the result
// layout of the 16x16x4 tile (row = (lane >> 4) + 4 reg, column = lane & 15) the value of rotation r in lane L belongs to register
// ((L & 15) / 4 + r) mod 4 of the SAME lane (the epilogue stores it there; the (0,3) block goes transposed into register 3).
// Rotation 2 fills only two of its four block slots with wanted sub-blocks, so two diagonal tiles share one MFMA: block slots 0, 1 take
// tile 2P's (2,0) and (3,1), slots 2, 3 tile 2P+1's (0,2) and (1,3), their transposes -- A lanes i < 8 <- fragment 2P lanes i + 8, lanes
// i >= 8 <- fragment 2P+1 lanes i - 8; B lanes i < 8 <- fragment 2P lanes i, lanes i >= 8 <- fragment 2P+1 lanes i (two ring reads with
// lane-dependent addresses, as many as the two rotated reads they replace; this assignment keeps them free of bank conflicts).  p = 100: 3,141 -> 2,976 (rotations per tile) -> 2,877 (paired) MFMA issue cycles per slab.
template <int NT, int SB, bool LAST_PLAIN, typename Hook = NoHook>
__device__ __forceinline__ void consume_ring_dg(Slab<NT> &s, const v2d (&ua)[SB > 0 ? SB : 1], const v2d (&r1)[NT - 1],
                                                const v2d (&r2)[NT - 1], Hook &&hook = NoHook())
{
    constexpr int NM16 = (NT - 1) * NT / 2, NP = (NT - 1) / 2;      // NP pairs of diagonal tiles share their rotation-2 MFMA
    __builtin_amdgcn_sched_barrier(0);
    static_for<2>([&](auto E) {
        constexpr int e = decltype(E)::value;
        static_for<NT - 1>([&](auto I_) {
            constexpr int I = decltype(I_)::value;
            static_for<I>([&](auto J_) {
                constexpr int J = decltype(J_)::value;
                AccTile<I *(I + 1) / 2 + J>::mfma(s.v[I][e], s.v[J][e]);
                hook(std::integral_constant<int, e * NM16 + I * (I + 1) / 2 + J>{});
            });
            AccTile<I *(I + 1) / 2 + I>::template mfma444<0>(s.v[I][e], s.v[I][e]);
            AccTile<I *(I + 1) / 2 + I>::template mfma444<1>(r1[I][e], s.v[I][e]);
            // rotation 2 wants only two of its four block slots per tile: tiles 2P and 2P + 1 share ONE MFMA (operands mixed
            // from both fragments by the ring reads: r2[P] = A, r2[NP + P] = B), accumulated in tile 2P's third register pair
            if constexpr (I < 2 * NP) {
                if constexpr (I % 2 == 0) AccTile<I *(I + 1) / 2 + I>::template mfma444<2>(r2[I / 2][e], r2[NP + I / 2][e]);
            } else AccTile<I *(I + 1) / 2 + I>::template mfma444<2>(r2[2 * NP][e], s.v[I][e]);
            hook(std::integral_constant<int, e * NM16 + I * (I + 1) / 2 + I>{});
        });
        if constexpr (LAST_PLAIN) {                             // the last tile row as plain tiles (ragged slab; SB == 0)
            static_for<NT>([&](auto J_) {
                constexpr int J = decltype(J_)::value;
                AccTile<(NT - 1) * NT / 2 + J>::mfma(s.v[NT - 1][e], s.v[J][e]);
            });
        } else {
            static_for<SB>([&](auto A_) {
                constexpr int a = decltype(A_)::value;
                static_for<NT>([&](auto T_) {
                    constexpr int T = decltype(T_)::value;
                    AccStrip<a * NT + T>::mfma(ua[a][e], s.v[T][e]);
                });
            });
        }
    });
    __builtin_amdgcn_sched_barrier(0);
}

// SADDR: fragments 0 .. NT-2 (whole X columns) are addressed as one scalar base (advanced by a scalar add per slab) plus
// a 32-bit per-lane offset; only the last fragment (y, ones, another allocation) keeps a 64-bit pointer per lane.
// A 64-bit VALU add runs on the DP units the MFMAs need: seven pointer bumps cost 75 cycles per slab (measured).
template <int NT, bool SHIFT, int SB = 0, bool SADDR = false>
__device__ __forceinline__ void gram_tri_ring_body(const double *__restrict__ x, int64_t n, int64_t ld, int p,
                                                   const double *__restrict__ y, const double *__restrict__ sums,
                                                   int64_t row_begin, int steps, double *__restrict__ tdst, double *lds)
{
    constexpr int NF = NT, NTILES = NT * (NT + 1) / 2;
    constexpr bool DG = !SHIFT && NT >= 6;                          // diagonal tiles as rotated 4x4x4 sub-blocks (consume_ring_dg)
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, q = lane >> 4;
    const gptr_t xg = (gptr_t)x, yg = (gptr_t)y;
    gptr_t cur[NF];
    LaneXf<NF> X;
    X.m_last = 1.0; X.o_last = 0.0;
    const double inv_cnt = (SHIFT && sums) ? 1.0 / sums[p + 1] : 0.0;
    const double cy = (SHIFT && sums) ? sums[p] * inv_cnt : 0.0;
    // SHIFT: every operand is x - c (FP64 VALU) and the ones column is made by the same fma (m = 0, o = 1).
    // !SHIFT: no FP64 VALU at all -- lanes of the ones column (and the padding lanes after it, whose tile entries
    // are dropped) DMA a constant {1, 1} with a pointer stride of 0.
    bool const_lane = false;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int col = 16 * f + i;
        double cf;
        if (col < p) { cur[f] = xg + (size_t)col * ld; cf = (SHIFT && sums) ? sums[col] * inv_cnt : 0.0; }
        else { cur[f] = yg; cf = cy; }
        if (f == NF - 1) {
            X.m_last = (col == p + 1) ? 0.0 : 1.0;
            X.o_last = (col == p + 1) ? 1.0 : -cf;
            if (!SHIFT && col > p) { const_lane = true; cur[f] = (gptr_t)g_ones; }
        }
        X.c[f] = cf;
    }
    const int64_t inc_last = const_lane ? 0 : 32;
    VecAcc<NF> V;          // unused (AUG): kept for consume_slab's signature
    V.sy = 0.0; V.syy = 0.0;
    static_assert(SB == 0 || !SHIFT, "the strip form takes its operands straight from the ring");
    static_assert(SB * NT <= 14, "strip accumulators");
    static_for<NTILES>([&](auto T_) { AccTile<decltype(T_)::value>::zero(); });
    static_for<SB * NT>([&](auto S_) { AccStrip<decltype(S_)::value>::zero(); });
    asm volatile("s_nop 7" ::: "memory");

    const int nslab = 2 * steps;
    const int64_t w0 = row_begin + 8 * w;
    int ns = 0;
    if (w0 + 8 <= n) {
        int64_t k = (n - w0 - 8) / 32 + 1;
        ns = k < nslab ? (int)k : nslab;
    }
    // instruction offsets (f - CEN) KiB select the fragment's part of the ring slot; sources are pre-decremented
    constexpr int CEN = NF / 2;
    unsigned voff[NF];
    gptr_t sbase = xg + w0;                                          // wave-uniform
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int col = 16 * f + i;
        voff[f] = (unsigned)(((int64_t)(col < p ? col : 0) * ld + 2 * q) * 8 - (f - CEN) * 1024);
        cur[f] += ((f == NF - 1 && const_lane) ? 0 : w0 + 2 * q) - (f - CEN) * 128;
    }

    constexpr int NSLOT = 5, SLOT_B = NF * 1024;
    static_assert((NSLOT - 1) * NF <= 63, "vmcnt field is 6 bits");
    const unsigned ring = (unsigned)(size_t)lds + (unsigned)w * (NSLOT * SLOT_B);     // LDS byte address of this wave's ring
    const v2d *ring_rd = reinterpret_cast<const v2d *>(reinterpret_cast<const char *>(lds) + w * (NSLOT * SLOT_B)) + lane;
    auto dma = [&](auto F_) {                                      // fragment f of the next slab (M0 set by the caller)
        constexpr int f = decltype(F_)::value;
        if constexpr (SADDR && f < NF - 1) glds_s<(f - CEN) * 1024>(voff[f], sbase);
        else glds_v<(f - CEN) * 1024>(cur[f]);
    };
    auto bump = [&]() {
        if constexpr (SADDR) { sbase += 32; cur[NF - 1] += inc_last; }
        else {
#pragma unroll
            for (int f = 0; f < NF; ++f) cur[f] += (f == NF - 1) ? inc_last : 32;
        }
    };
    auto issue = [&](int slot) {                                   // DMA the next slab into ring slot `slot`
        set_m0(ring + (unsigned)slot * SLOT_B + CEN * 1024);
        static_for<NF>(dma);
        bump();
    };
    // strip A operands: lane (q, blk, x) reads the 16 B of lane (q, a, x) of the last fragment (both k halves)
    const v2d *ring_rdA = reinterpret_cast<const v2d *>(reinterpret_cast<const char *>(lds) + w * (NSLOT * SLOT_B)) +
                          (NF - 1) * 64 + 16 * q + (lane & 3);
    // rotated A operands of the diagonal tiles: lane (q, i) reads the 16 B of lane (q, (i + 4 r) mod 16), r = 1, 2
    const v2d *ring_rd1 = reinterpret_cast<const v2d *>(reinterpret_cast<const char *>(lds) + w * (NSLOT * SLOT_B)) + (16 * q + ((i + 4) & 15));
    const v2d *ring_rd2 = reinterpret_cast<const v2d *>(reinterpret_cast<const char *>(lds) + w * (NSLOT * SLOT_B)) + (16 * q + ((i + 8) & 15));
    // paired rotation 2 (tiles 2P, 2P + 1; add P * 2048 bytes): A lanes i < 8 <- fragment 2P lane i + 8, i >= 8 <- fragment 2P + 1 lane i - 8;
    //                                                          B lanes i < 8 <- fragment 2P lane i,     i >= 8 <- fragment 2P + 1 lane i
    // (the two halves of a 16-lane row then read different halves of the 64 LDS banks; tile 2P + 1's blocks come out transposed)
    const v2d *ring_rdPA = reinterpret_cast<const v2d *>(reinterpret_cast<const char *>(lds) + w * (NSLOT * SLOT_B)) + (i < 8 ? 0 : 64) + (16 * q + (i ^ 8));
    const v2d *ring_rdPB = reinterpret_cast<const v2d *>(reinterpret_cast<const char *>(lds) + w * (NSLOT * SLOT_B)) + (i < 8 ? 0 : 64) + (16 * q + i);
    constexpr int NR = DG ? NT - 1 : 1, NP = (NT - 1) / 2;
    // r2[] entry e of a slab: e < NP pair A operands, NP <= e < 2 NP pair B operands, e = 2 NP (odd tile count) the last tile's own rotation
    auto r2_src = [&](auto E_) -> const v2d * {
        constexpr int e = decltype(E_)::value;
        if constexpr (e < NP) return ring_rdPA + (e * 2048) / 16;
        else if constexpr (e < 2 * NP) return ring_rdPB + ((e - NP) * 2048) / 16;
        else return ring_rd2 + ((NT - 2) * 1024) / 16;
    };
    auto fetch = [&](Slab<NF> &s, v2d (&ua)[SB > 0 ? SB : 1], v2d (&r1)[NR], v2d (&r2)[NR], int slot) {   // ring slot -> registers (a lane reads back its own 16 B)
#pragma unroll
        for (int f = 0; f < NF; ++f) s.v[f] = ring_rd[(slot * SLOT_B + f * 1024) / 16];
#pragma unroll
        for (int a = 0; a < SB; ++a) ua[a] = ring_rdA[(slot * SLOT_B) / 16 + 4 * a];
        if constexpr (DG) {
#pragma unroll
            for (int f = 0; f < NT - 1; ++f) r1[f] = ring_rd1[(slot * SLOT_B + f * 1024) / 16];
            static_for<NT - 1>([&](auto E_) { r2[decltype(E_)::value] = r2_src(E_)[(slot * SLOT_B) / 16]; });
        }
    };
    auto next = [](int v) { return v + 1 == NSLOT ? 0 : v + 1; };
    Slab<NF> sa, sb;
    v2d ua[SB > 0 ? SB : 1], ub[SB > 0 ? SB : 1];
    v2d ra1[NR], ra2[NR], rb1[NR], rb2[NR];
    sa.y = v2d{0.0, 0.0}; sb.y = sa.y;
#ifdef OEM_GRAM_DIAG
    unsigned long long gd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gl = __builtin_amdgcn_s_memtime();
#endif
    // prologue: NSLOT-1 slabs in flight, slab 0 in registers
    const int npre = ns < NSLOT - 1 ? ns : NSLOT - 1;
    for (int j = 0; j < npre; ++j) issue(j);
    int slot = 0, islot = npre % NSLOT, issued = npre, k = 0;
    if (ns > 0) {
        if (npre == NSLOT - 1) wait_vm<(NSLOT - 2) * NF>(); else wait_vm<0>();
        fetch(sa, ua, ra1, ra2, 0);
        slot = next(slot);
    }
    // steady state, two slabs per trip (sa / sb alternate as "in registers" and "being fetched"):
    //   issue slab k+NSLOT-1; wait until slab k+1 has landed; copy it to registers; MFMAs of slab k
    GSTAMP(0);                                   // prologue
    // steady state: while the MFMAs of slab k run, the hook (a) issues the DMA of slab k+NSLOT-1 after MFMAs 1..NF,
    // (b) bumps the pointers, (c) waits for slab k+1 (exact vmcnt) and (d) copies it from the ring to the other
    // register slab after MFMAs NF+2 .. 2NF+1.  None of it is FP64 VALU, so it rides in the MFMA shadows.
    auto steady = [&](Slab<NF> &use, Slab<NF> &nxt, v2d (&uuse)[SB > 0 ? SB : 1], v2d (&unxt)[SB > 0 ? SB : 1],
                      v2d (&r1use)[NR], v2d (&r2use)[NR], v2d (&r1nxt)[NR], v2d (&r2nxt)[NR]) {
        const unsigned dst = ring + (unsigned)islot * SLOT_B;
        const v2d *src = ring_rd + (slot * SLOT_B) / 16;
        const v2d *srcA = ring_rdA + (slot * SLOT_B) / 16;
        const v2d *src1 = ring_rd1 + (slot * SLOT_B) / 16;
        const int sl16 = (slot * SLOT_B) / 16;
        // OEM_GRAM_EXP (timing experiments in diagnostic builds only; results are wrong): 1 = no DMA issue,
        // 2 = no ring reads, 3 = no vmcnt wait, 4 = no pointer bumps, 5 = block kernel without shift / VALU sums
        auto hook = [&](auto M_) {
            constexpr int m = decltype(M_)::value;
            if constexpr (OEM_GRAM_EXP != 1 && m == 1) set_m0(dst + CEN * 1024);
            if constexpr (OEM_GRAM_EXP != 1 && m >= 1 && m <= NF) dma(std::integral_constant<int, m - 1>{});
            if constexpr (m == NF + 1) {
                if constexpr (OEM_GRAM_EXP != 4) bump();
                if constexpr (OEM_GRAM_EXP != 3 && OEM_GRAM_EXP != 1) wait_vm<(NSLOT - 2) * NF>();
            }
            if constexpr (OEM_GRAM_EXP != 2 && m >= NF + 2 && m <= 2 * NF + 1) nxt.v[m - NF - 2] = src[((m - NF - 2) * 1024) / 16];
            if constexpr (OEM_GRAM_EXP != 2 && SB > 0 && m >= 2 * NF + 2 && m < 2 * NF + 2 + SB) unxt[m - 2 * NF - 2] = srcA[4 * (m - 2 * NF - 2)];
            // rotated operands of the next slab's diagonal tiles: two reads per hook
            if constexpr (DG && m >= 2 * NF + 2 + SB && m < 2 * NF + 2 + SB + (NT - 1)) {
                constexpr int f = m - (2 * NF + 2 + SB);
                r1nxt[DG ? f : 0] = src1[(f * 1024) / 16]; r2nxt[DG ? f : 0] = r2_src(std::integral_constant<int, DG ? f : 0>{})[sl16];
            }
        };
        if constexpr (DG) consume_ring_dg<NT, SB, (SB == 0)>(use, uuse, r1use, r2use, hook);
        else if constexpr (SB > 0) consume_ring_strip<NT, SB>(use, uuse, hook);
        else consume_slab<NT, NT, true, false, false, true, SHIFT>(V, use, X, cy, 0, n, hook);
        islot = next(islot); slot = next(slot);
    };
    static_assert(SB == 0 || 2 * NF + 2 + SB <= (NT - 1) * NT, "not enough 16x16x4 MFMAs per slab to carry the hooks");
    static_assert(2 * NF + 1 < NT * (NT + 1), "not enough MFMAs per slab to carry the hooks");
    static_assert(!DG || 2 * NF + 2 + SB + (NT - 1) <= (NT - 1) * NT, "not enough hook slots per slab for the rotated reads");
    while (k + NSLOT + 1 <= ns) {
        steady(sa, sb, ua, ub, ra1, ra2, rb1, rb2);
        GSTAMP(4);
        steady(sb, sa, ub, ua, rb1, rb2, ra1, ra2);
        GSTAMP(4);
        issued += 2; k += 2;
    }
    // drain: slab k is in sa; issue what is left, then walk the ring with everything landed
    while (issued < ns) { issue(islot); islot = next(islot); ++issued; }
    wait_vm<0>();
    for (; k < ns; ++k) {
        if (k + 1 < ns) { fetch(sb, ub, rb1, rb2, slot); slot = next(slot); }
        if constexpr (DG) consume_ring_dg<NT, SB, (SB == 0)>(sa, ua, ra1, ra2);
        else if constexpr (SB > 0) consume_ring_strip<NT, SB>(sa, ua);
        else consume_slab<NT, NT, true, false, false, true, SHIFT>(V, sa, X, cy, 0, n);
        if (k + 1 < ns) {
#pragma unroll
            for (int f = 0; f < NF; ++f) sa.v[f] = sb.v[f];
#pragma unroll
            for (int a = 0; a < SB; ++a) ua[a] = ub[a];
            if constexpr (DG) {
#pragma unroll
                for (int f = 0; f < NT - 1; ++f) { ra1[f] = rb1[f]; ra2[f] = rb2[f]; }
            }
        }
    }
    GSTAMP(5);                                   // drain
    __syncthreads();                                               // the ring memory is reused by the epilogue below
    if (ns < nslab && w0 + 32 * (int64_t)ns < n) {
        // ragged slab, addressed relative to the running pointers (cur = column base + w0 + 2q + 32 ns)
        const int64_t r = w0 + 32 * (int64_t)ns + 2 * q;
        const int64_t o0 = (r < n ? r : n - 1) - r, o1 = (r + 1 < n ? r + 1 : n - 1) - r;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (f == NF - 1 && const_lane) { sa.v[f] = v2d{1.0, 1.0}; continue; }
            // undo the pre-decrement; SADDR fragments are rebuilt from the scalar base and the lane offset
            const gptr_t pf = (SADDR && f < NF - 1)
                                  ? (gptr_t)((const char __attribute__((address_space(1))) *)sbase + voff[f]) + (f - CEN) * 128
                                  : cur[f] + (f - CEN) * 128;
            sa.v[f].x = pf[o0]; sa.v[f].y = pf[o1];
        }
        if constexpr (DG) {
            // the diagonal tiles' AGPRs hold 4x4x4 accumulators: the ragged slab takes the same form.  Rows past n are masked to
            // zero, the fragments go through (free) ring slot 0 of this wave to come back rotated, the last tile row stays plain.
            const double m0 = (r < n) ? 1.0 : 0.0, m1 = (r + 1 < n) ? 1.0 : 0.0;
#pragma unroll
            for (int f = 0; f < NF; ++f) { sa.v[f].x *= m0; sa.v[f].y *= m1; }
            v2d *ring_wr = reinterpret_cast<v2d *>(reinterpret_cast<char *>(lds) + w * (NSLOT * SLOT_B)) + lane;
#pragma unroll
            for (int f = 0; f < NT - 1; ++f) ring_wr[(f * 1024) / 16] = sa.v[f];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int f = 0; f < NT - 1; ++f) ra1[f] = ring_rd1[(f * 1024) / 16];
            static_for<NT - 1>([&](auto E_) { ra2[decltype(E_)::value] = r2_src(E_)[0]; });
            asm volatile("s_nop 3" ::: "memory");
            consume_ring_dg<NT, SB, true>(sa, ua, ra1, ra2);
        } else consume_slab<NT, NT, true, true, false, true, SHIFT>(V, sa, X, cy, r, n);
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // ---- tile partials: all four waves write their tiles to LDS at once (two passes of <= 14 tiles: 4 x 28 KiB),
    //      then every wave sums a quarter of the pass ((w0 + w1) + (w2 + w3), fixed order) and stores it coalesced.
    constexpr int TPP = (NTILES + 1) / 2;                                   // tiles per pass
    static_for<2>([&](auto P_) {
        constexpr int pass = decltype(P_)::value, t0 = pass * TPP, t1 = (t0 + TPP < NTILES) ? t0 + TPP : NTILES;
        static_for<(t1 - t0)>([&](auto T_) {
            constexpr int tt = t0 + decltype(T_)::value;
            // is tt a diagonal tile (I, I) with I < NT - 1 ?  tt = I (I + 3) / 2
            constexpr bool isdiag = DG && [] { for (int I = 0; I < NT - 1; ++I) if (I * (I + 3) / 2 == tt) return true; return false; }();
            if constexpr (isdiag) {
                double *tl = lds + (size_t)w * (TPP * 256) + (size_t)(tt - t0) * 256;
                static_for<4>([&](auto R_) { tl[decltype(R_)::value * 64 + lane] = 0.0; });
                const int b4 = i >> 2;
                constexpr int I = [] { for (int I2 = 0; I2 < NT - 1; ++I2) if (I2 * (I2 + 3) / 2 == tt) return I2; return 0; }();
                static_for<2>([&](auto R_) {
                    constexpr int r = decltype(R_)::value;
                    const double v = AccTile<tt>::template read<r>();           // registers 2r, 2r + 1 of the tile: rotation r
                    int reg = (b4 + r) & 3, ln = lane;
                    if (r == 1 && b4 == 3) { reg = 3; ln = ((lane & 3) << 4) | q; }   // (0,3) -> (3,0) transposed
                    tl[reg * 64 + ln] = v;
                });
                // rotation 2: a pair's accumulator sits in tile 2P (third register pair); block slots 0, 1 are tile 2P's (2,0), (3,1)
                // (register b4 + 2, same lane), slots 2, 3 tile 2P + 1's transposed ones (register b4, lane 16 j + 4 (b4 - 2) + row); an unpaired last tile
                // keeps all four slots ((0,2) and (1,3) are duplicates in the upper triangle, which nobody reads)
                if constexpr (I >= 2 * ((NT - 1) / 2)) tl[((b4 + 2) & 3) * 64 + lane] = AccTile<tt>::template read<2>();
                else if constexpr (I % 2 == 0) { const double v = AccTile<tt>::template read<2>(); if (b4 < 2) tl[(b4 + 2) * 64 + lane] = v; }
                else {
                    constexpr int ttl = (I - 1) * (I + 2) / 2;                  // the pair's first tile
                    const double v = AccTile<ttl>::template read<2>();
                    if (b4 >= 2) tl[b4 * 64 + (((lane & 3) << 4) | ((b4 - 2) << 2) | q)] = v;     // (0,2) -> (2,0), (1,3) -> (3,1): transposed
                }
            } else {
                static_for<4>([&](auto R_) {
                    constexpr int r = decltype(R_)::value;
                    lds[(size_t)w * (TPP * 256) + ((tt - t0) * 4 + r) * 64 + lane] = AccTile<tt>::template read<r>();
                });
            }
        });
        // strip accumulator (a, T): lane 16 i + c holds element (row 4 a + i, column c) of tile (NT-1, T); the 16x16x4
        // layout keeps (row, c) in register row / 4 of lane 16 (row % 4) + c: register a of the same lane
        static_for<SB * NT>([&](auto S_) {
            constexpr int sidx = decltype(S_)::value, a = sidx / NT, T = sidx % NT, tt = (NT - 1) * NT / 2 + T;
            if constexpr (tt >= t0 && tt < t1)
                lds[(size_t)w * (TPP * 256) + ((tt - t0) * 4 + a) * 64 + lane] += AccStrip<sidx>::read();
        });
        __syncthreads();
        for (int e = tid; e < (t1 - t0) * 256; e += 256) {
            const double a0 = lds[e], a1 = lds[TPP * 256 + e], a2 = lds[2 * TPP * 256 + e], a3 = lds[3 * TPP * 256 + e];
            tdst[(size_t)t0 * 256 + e] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
    });
#ifdef OEM_GRAM_DIAG
    GSTAMP(6);                                   // epilogue
    if (blockIdx.x == 7 && tid == 0) { gd[7] = (unsigned long long)ns; for (int j = 0; j < 8; ++j) g_gram_diag[j] = gd[j]; }
#endif
}

#ifdef OEM_GRAM_DIAG
extern "C" __attribute__((visibility("default"))) int oemgpu_gram_diag_read(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gram_diag), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif


// whole lower triangle of Z = [X | y | 1] in one wave (p + 2 <= 112)
template <int NT, bool ALIGNED>
__global__ __launch_bounds__(256) void gram_tri_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                        const double *__restrict__ sums, double *__restrict__ tpart,
                                                        double *__restrict__ vpart, GramDims a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int chunk = blockIdx.x;
    gram_body<NT, NT, true, ALIGNED, false, true>(x, a.n, a.ld, a.p, y, sums, a.ntc, 0, 0,
                                                  (int64_t)chunk * a.steps * 64, a.steps, true,
                                                  tpart + (size_t)chunk * a.ntile * 256, vpart, lds);
}

// whole lower triangle of Z = [X | y | 1] in one wave, slabs through the LDS-DMA ring (16-byte aligned X)
template <int NT, int SB, bool SADDR>
__global__ __launch_bounds__(256) void gram_ring_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ sums, double *__restrict__ tpart,
                                                         double *__restrict__ vpart, GramDims a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    (void)vpart;
    const int chunk = blockIdx.x;
    if (shift_needed_wave(sums, a.p))
        gram_tri_ring_body<NT, true, 0, SADDR>(x, a.n, a.ld, a.p, y, sums, (int64_t)chunk * a.steps * 64, a.steps,
                                               tpart + (size_t)chunk * a.ntile * 256, lds);
    else
        gram_tri_ring_body<NT, false, SB, SADDR>(x, a.n, a.ld, a.p, y, sums, (int64_t)chunk * a.steps * 64, a.steps,
                                                 tpart + (size_t)chunk * a.ntile * 256, lds);
}

// 4x4 tile blocks of X'X; blockIdx -> (row chunk, tile block) so that the tile blocks of one row chunk run
// back to back on one XCD (blocks b and b+8 share an XCD) and re-read that chunk's rows from its L2.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void gram_blk_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                        const double *__restrict__ sums, double *__restrict__ tpart,
                                                        double *__restrict__ vpart, GramDims a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int L = blockIdx.x, xcd = L & 7, s = L >> 3;
    const int tb = s % a.nblk, chunk = (s / a.nblk) * 8 + xcd;
    int BI = (int)((sqrtf(8.0f * (float)tb + 1.0f) - 1.0f) * 0.5f);
    while (BI * (BI + 1) / 2 > tb) --BI;
    while ((BI + 1) * (BI + 2) / 2 <= tb) ++BI;
    const int BJ = tb - BI * (BI + 1) / 2;
    const int64_t row_begin = (int64_t)chunk * a.steps * 64;
    double *tdst = tpart + (size_t)chunk * a.ntile * 256;
    double *vdst = vpart + (size_t)chunk * (32 * a.ntc + 4);
    if (BI == BJ)
        gram_body<4, 4, true, ALIGNED, true, false>(x, a.n, a.ld, a.p, y, sums, a.ntc, 4 * BI, 4 * BI, row_begin,
                                                    a.steps, BI == 0, tdst, vdst, lds);
    else
        gram_body<4, 4, false, ALIGNED, false, false>(x, a.n, a.ld, a.p, y, sums, a.ntc, 4 * BI, 4 * BJ,
                                                      row_begin, a.steps, false, tdst, vdst, lds);
}

// The deal of ntc tile columns into n8 super-block rows of eight, n6 <= 1 of six and n4 <= 1 of four: the one with the least
// multiply time (in tile units: an off-diagonal super-block of h1 x h2 tiles costs h1 h2; a diagonal one 36 (eight), 24 (six:
// 6 + 5 + 5 + 5 tiles over the four waves) or 12 (four: 3 + 3 + 2 + 2)); ties go to fewer super-blocks.  One six and one four next to
// the eights cover every remainder of ntc mod 8 with at most one tile column of padding (two at 8 k + 1 ... never more), which no
// deal with more sixes or fours beats -- and it keeps the kinds of super-block at seven.
void gram_sb_deal(int ntc, int *n8_out, int *n6_out, int *n4_out)
{
    int best8 = (ntc + 7) / 8, best6 = 0, best4 = 0;
    long best = -1;
    for (int n4 = 0; n4 <= 1; ++n4)
        for (int n6 = 0; n6 <= 1; ++n6) {
            const int rest = ntc - 6 * n6 - 4 * n4, n8 = rest > 0 ? (rest + 7) / 8 : 0;
            if (n8 + n6 + n4 == 0 || 8 * n8 + 6 * n6 + 4 * n4 - ntc >= 4) continue;      // (no super-block row that is all padding)
            const long cost = 64L * n8 * (n8 - 1) / 2 + 48L * n8 * n6 + 32L * n8 * n4 + 24L * n6 * n4 + 36L * n8 + 24L * n6 + 12L * n4;
            if (best < 0 || cost < best || (cost == best && n8 + n6 + n4 < best8 + best6 + best4)) { best = cost; best8 = n8; best6 = n6; best4 = n4; }
        }
    *n8_out = best8; *n6_out = best6; *n4_out = best4;
}

constexpr int GRAM_MAX_ROUNDS = 12;
// (gram_wd.hip's workgroups are all alike: more than one round per CU never won -- profiles/r5_gram_one_read.txt -- and the scratch that
// holds 'any smaller row count' is sized by the largest count the search may return)
static inline int gram_max_rounds(const GramPlan &pl) { return pl.wd ? (pl.wd_units == 1 ? 2 : 6) : GRAM_MAX_ROUNDS; }      // (units: two kinds of workgroup of almost one length)

static GramPlan gram_plan_compute(int64_t n, int p, int num_cu)
{
    GramPlan pl;
    pl.p = p;
    pl.tri = (p + 2 + 15) / 16 <= 7;                     // Z = [X | y | 1] fits one wave's triangle
    pl.n8 = pl.n6 = pl.n4 = 0; pl.wd = 0; pl.wd_units = 0;
    pl.ntc = pl.tri ? (p + 2 + 15) / 16 : (p + 15) / 16;
    pl.ntile = pl.ntc * (pl.ntc + 1) / 2;
    const int64_t nsteps = (n + 63) / 64;
    if (pl.tri) {
        pl.nblk = 1;
        int64_t c = nsteps < num_cu ? nsteps : num_cu;
        if (c < 1) c = 1;
        pl.steps = (int)((nsteps + c - 1) / c);
        if (pl.steps < 1) pl.steps = 1;
        pl.nchunk = (int)((nsteps + pl.steps - 1) / pl.steps);
        if (pl.nchunk < 1) pl.nchunk = 1;
    } else {
        const int nb = (pl.ntc + 3) / 4;
        pl.nblk = nb * (nb + 1) / 2;
        int n8 = 0, n6 = 0, n4 = 0;
        gram_sb_deal(pl.ntc, &n8, &n6, &n4);                             // the shared-slab kernel's super-block rows
        pl.n8 = n8; pl.n6 = n6; pl.n4 = n4;
        // 15 or 16 tile columns (225 <= p <= 256: config 5): ONE workgroup of eight waves per row chunk multiplies the whole triangle
        // from one read of X (gram_wd.hip) instead of three super-blocks that each stream the rows
        // (11 or 12 tile columns, 161 <= p <= 192: the same with groups of three tile columns -- 78 tiles, nine or ten per wave)
        pl.wd = (pl.ntc == 15 || pl.ntc == 16) ? 4 : ((pl.ntc == 11 || pl.ntc == 12) ? 3 : 0);
        pl.wd_units = pl.wd ? 1 : 0;
        // 16 k tile columns (k >= 2: p = 512, 1,024, ... and the fifteen columns below each): k such diagonal units and, between every
        // two of them, two off-diagonal blocks of 8 x 16 tiles on eight waves -- one launch, 80 fragment reads per slab at p = 512
        // where the super-blocks make 128
        if (pl.ntc >= 31 && (pl.ntc % 16 == 0 || pl.ntc % 16 == 15)) { pl.wd = 4; pl.wd_units = (pl.ntc + 15) / 16; }
        const int nsb = n8 + n6 + n4, nsblk = pl.wd ? pl.wd_units * pl.wd_units : nsb * (nsb + 1) / 2;
        // 1-12 rounds of one workgroup per CU: the count whose launch ends soonest when the workgroups are handed out longest
        // first (gram_sb_kernel) -- a greedy replay with the measured costs (tools/gram_diag.py: 2,136 cycles per 8-row slab off
        // the diagonal at 32 MFMAs per wave, 1,284 on it at 18: ~66 per MFMA + ~100; ~25 k per workgroup: ring fill, 72-128 KB of
        // partials written) plus the partials the reduction has to read (2 KB per tile and chunk).  Until round 5 the search began
        // at 8 rounds, which is right for the configurations' row counts (millions) and wrong below: at n = 1e5 a workgroup then
        // has 3 steps of rows and spends more time filling its ring than multiplying (p = 128: 82 us, with ONE round 42;
        // p = 256: 178 -> 141 with three; profiles/r5_gram_rounds_experiment.txt holds the grid this replay was checked against).
        const int64_t cnt[SB_KINDS] = {(int64_t)n8 * (n8 - 1) / 2, (int64_t)n6 * n8, n8, (int64_t)n4 * n8, (int64_t)n4 * n6, n6, n4};
        const double per_slab[SB_KINDS] = {2136.0, 1680.0, 1284.0, 1160.0, 890.0, 890.0, 520.0};      // (the kernel's launch order)
        int64_t c = 0;
        double best = 0.0;
        for (int rounds = 1; rounds <= gram_max_rounds(pl); ++rounds) {
            int64_t cc = ((int64_t)num_cu * rounds) / nsblk;
            if (cc > nsteps) cc = nsteps;
            if (cc < 1) cc = 1;
            cc = (cc + 7) / 8 * 8;
            const double slabs = (double)((nsteps + cc - 1) / cc) * 8.0;
            // CUs that fall free at the same time form a group: (time, CUs), ascending; a kind's workgroups go to the earliest group
            // first, whole groups at a time -- the same schedule as a heap of 256 finish times in ~rounds x kinds steps
            std::vector<std::pair<double, int64_t>> grp{{0.0, (int64_t)num_cu}};
            double end = 0.0;
            for (int kind = 0; kind < SB_KINDS; ++kind) {
                // (gram_wd.hip: one kind of workgroup, 68 MFMAs per SIMD and slab -- 4,352 cycles of issue, ~4,500 measured)
                // (units: kind 0 = the diagonal units, kind 1 = the off-diagonal 8 x 16-tile blocks, 64 MFMAs per SIMD and slab)
                const double d = slabs * (pl.wd == 4 ? (kind == 0 ? 4500.0 : 4250.0) : (pl.wd == 3 ? 2700.0 : per_slab[kind])) + 25000.0;
                for (int64_t m = pl.wd ? (kind == 0 ? cc * pl.wd_units : (kind == 1 ? cc * pl.wd_units * (pl.wd_units - 1) : 0)) : cc * cnt[kind]; m > 0;) {
                    const double t = grp.front().first + d;
                    const int64_t take = grp.front().second < m ? grp.front().second : m;
                    if ((grp.front().second -= take) == 0) grp.erase(grp.begin());
                    size_t at = grp.size();
                    while (at > 0 && grp[at - 1].first > t) --at;
                    if (at > 0 && grp[at - 1].first == t) grp[at - 1].second += take; else grp.insert(grp.begin() + at, {t, take});
                    m -= take;
                    if (t > end) end = t;
                }
            }
            const double cost = end + (double)cc * pl.ntile * 2048.0 / 3.5e12 * 2.1e9;
            if (c == 0 || cost < best * 0.995) { c = cc; best = cost; }      // a later (larger) count has to win by more than noise
            if (cc >= nsteps) break;
        }
        pl.steps = (int)((nsteps + c - 1) / c);
        if (pl.steps < 1) pl.steps = 1;
        pl.nchunk = (int)c;
    }
    pl.tpart_doubles = (size_t)pl.nchunk * pl.ntile * 256;
    pl.vpart_doubles = (size_t)pl.nchunk * (32 * pl.ntc + 4);
    return pl;
}

// The plan of the last (n, p, device) of this thread is kept: the replay above is ~0.1-0.3 ms of host time, and callers come back
// with the same sizes (a bench loop, the folds of xval.oem, the row blocks of a host-resident call).
GramPlan gram_plan(int64_t n, int p, int num_cu)
{
    struct Memo { int64_t n; int p, cu; unsigned gen; GramPlan pl; };
    static thread_local Memo memo[4] = {{-1, 0, 0, 0u, {}}, {-1, 0, 0, 0u, {}}, {-1, 0, 0, 0u, {}}, {-1, 0, 0, 0u, {}}};
    static thread_local unsigned next = 0;
    const unsigned gen = sw().generation;
    for (const Memo &m : memo) if (m.n == n && m.p == p && m.cu == num_cu && m.gen == gen) return m.pl;
    Memo &m = memo[next++ & 3];
    m.n = n; m.p = p; m.cu = num_cu; m.gen = gen; m.pl = gram_plan_compute(n, p, num_cu);
    return m.pl;
}

// tpart / vpart sizes that hold the plan of ANY row count up to nmax (scratch shared by the folds of xval.oem or the row tiles of
// a sparse x: the chunk count is not monotone in n)
GramPlan gram_plan_bound(int64_t nmax, int p, int num_cu)
{
    GramPlan pl = gram_plan(nmax, p, num_cu);
    if (!pl.tri) {
        const int nsb = pl.n8 + pl.n6 + pl.n4, nsblk = pl.wd ? pl.wd_units * pl.wd_units : nsb * (nsb + 1) / 2;
        const int64_t nsteps = (nmax + 63) / 64;
        int64_t cc = ((int64_t)num_cu * gram_max_rounds(pl)) / nsblk;
        if (cc > nsteps) cc = nsteps;
        if (cc < 1) cc = 1;
        cc = (cc + 7) / 8 * 8;
        if (cc > pl.nchunk) pl.nchunk = (int)cc;
    } else {
        const int64_t nsteps = (nmax + 63) / 64;                      // one chunk per CU, fewer when there are fewer steps
        const int64_t cc = nsteps < num_cu ? (nsteps < 1 ? 1 : nsteps) : num_cu;
        if (cc > pl.nchunk) pl.nchunk = (int)cc;
    }
    pl.tpart_doubles = (size_t)pl.nchunk * pl.ntile * 256;
    pl.vpart_doubles = (size_t)pl.nchunk * (32 * pl.ntc + 4);
    return pl;
}

template <bool ALIGNED>
static int launch_gram_t(hipStream_t s, const GramPlan &pl, const double *x, const double *y, const double *sums,
                         double *tpart, double *vpart, const GramDims &a)
{
    const size_t tile_bytes = 256 * sizeof(double);
#define OEM_TRI(NT)                                                                                       \
    case NT: {                                                                                            \
        size_t sh = (size_t)(NT * (NT + 1) / 2) * tile_bytes;                                             \
        size_t vb = (size_t)4 * (2 * 16 * NT + 4) * sizeof(double);                                       \
        if (sh < vb) sh = vb;                                                                             \
        hipLaunchKernelGGL((gram_tri_kernel<NT, ALIGNED>), dim3(pl.nchunk), dim3(256), sh, s, x, y, sums, tpart, vpart, a);          \
        break;                                                                                            \
    }
    // 16-byte aligned X with at least 4 tile columns: the LDS-DMA ring form
    // (not for a handful of rows: the ring's pre-decremented 32-bit lane offsets assume 16 columns x ld x 8 bytes >= 1 KiB per
    // fragment, i.e. ld >= 8 -- a 1-row shard with ld = 2 wrapped them around and read 4 GB away)
    if (pl.tri && ALIGNED && pl.ntc >= 4 && a.n >= 64) {
        size_t sh = (size_t)pl.ntile * tile_bytes;
        const size_t rb = (size_t)4 * 5 * pl.ntc * 1024;                         // 4 waves x NSLOT x NF KiB ring
        if (sh < rb) sh = rb;
        // real columns in the last tile row -> strip sub-blocks (0: the last row stays on 16x16x4 tiles)
        const int rem = pl.p + 2 - 16 * (pl.ntc - 1);
        // scalar-base addressing needs fragments 0 .. ntc-2 to be whole X columns and every lane offset to fit 32 bits
        const bool saddr = pl.p >= 16 * (pl.ntc - 1) && (double)pl.p * (double)a.ld * 8.0 + 65536.0 < 4294967296.0;
        const int sb = !saddr ? 0 : (rem <= 4 ? 1 : (rem <= 8 ? 2 : 0));
#define OEM_RING(NT, SB, SA)                                                                                                   \
    if (pl.ntc == NT && sb == SB && saddr == SA) {                                                                             \
        if (sh > 64 * 1024 && lds_limit_once(reinterpret_cast<const void *>(&gram_ring_kernel<NT, SB, SA>), sh)) return OEMGPU_ERR_HIP; \
        hipLaunchKernelGGL((gram_ring_kernel<NT, SB, SA>), dim3(pl.nchunk), dim3(256), sh, s, x, y, sums, tpart, vpart, a);   \
        OEM_HIP(hipGetLastError());                                                                                            \
        return 0;                                                                                                              \
    }
        OEM_RING(4, 0, true) OEM_RING(5, 0, true) OEM_RING(6, 0, true) OEM_RING(7, 0, true)
        OEM_RING(4, 1, true) OEM_RING(5, 1, true) OEM_RING(6, 1, true) OEM_RING(7, 1, true)
        OEM_RING(4, 2, true) OEM_RING(5, 2, true) OEM_RING(6, 2, true) OEM_RING(7, 2, true)
        OEM_RING(4, 0, false) OEM_RING(5, 0, false) OEM_RING(6, 0, false) OEM_RING(7, 0, false)
#undef OEM_RING
    }
    if (pl.tri) {
        switch (pl.ntc) {
            OEM_TRI(1) OEM_TRI(2) OEM_TRI(3) OEM_TRI(4) OEM_TRI(5) OEM_TRI(6) OEM_TRI(7)
        default: set_error("gram: bad tile count %d", pl.ntc); return OEMGPU_ERR_INTERNAL;
        }
    } else {
        if (ALIGNED && a.n >= 64 && (double)a.ld * 16.0 * 8.0 < 4294967296.0) {   // 32-bit lane offsets within a tile
            if (pl.wd) return launch_gram_wd(s, pl, x, y, sums, tpart, vpart, a);
            return launch_gram_sb(s, pl, x, y, sums, tpart, vpart, a);
        }
        size_t sh = 16 * tile_bytes;
        size_t vb = (size_t)4 * (2 * 16 * 4 + 4) * sizeof(double);
        if (sh < vb) sh = vb;
        hipLaunchKernelGGL((gram_blk_kernel<ALIGNED>), dim3(pl.nchunk * pl.nblk), dim3(256), sh, s, x, y, sums, tpart, vpart, a);
    }
#undef OEM_TRI
    OEM_HIP(hipGetLastError());
    return 0;
}

int launch_gram(hipStream_t s, const GramPlan &pl, const double *x, int64_t n, int64_t ld, const double *y,
                const double *sums, double *tpart, double *vpart)
{
    GramDims a;
    a.n = n; a.ld = ld; a.p = pl.p;
    a.ntc = pl.ntc; a.ntile = pl.ntile; a.nblk = pl.nblk; a.nchunk = pl.nchunk; a.steps = pl.steps;
    const bool aligned = (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (ld & 1) == 0;
    return aligned ? launch_gram_t<true>(s, pl, x, y, sums, tpart, vpart, a)
                   : launch_gram_t<false>(s, pl, x, y, sums, tpart, vpart, a);
}

// ------------------------------------------------------------------------------------------------
// partial reduction: sums the per-chunk partials in chunk order (bitwise reproducible) and scatters the
// MFMA accumulator layout (row = (lane>>4) + 4 reg, col = lane & 15) into the q x q moment buffer.
// ------------------------------------------------------------------------------------------------
// A tile's chunk partials are summed by four thread groups (chunks c = g mod 4, each in ascending order, eight loads in
// flight) and combined as (g0 + g1) + (g2 + g3): a fixed order, so the result is reproducible; with one group the
// kernel was a 15 us latency chain of 253 dependent steps on 28 workgroups.
// The same sum for few tiles (p + 2 <= 112: at most 28): SPL workgroups per tile, each EB = 256 / SPL elements of the tile over
// NG = 4 SPL groups of chunks -- 28 workgroups of the form below keep 28 of 256 CUs busy (config 1: 11 us for 14.5 MB).
template <int SPL>
__global__ __launch_bounds__(1024) void moments_reduce_split_kernel(const double *__restrict__ tpart, int p, int ntile, int nchunk,
                                                                     double *__restrict__ M)
{
    constexpr int EB = 256 / SPL, NG = 4 * SPL;
    __shared__ double part[NG][EB];
    const int q = p + 2;
    const int tile = blockIdx.x / SPL, e = threadIdx.x % EB, grp = threadIdx.x / EB, el = (blockIdx.x % SPL) * EB + e;
    int I = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
    while (I * (I + 1) / 2 > tile) --I;
    while ((I + 1) * (I + 2) / 2 <= tile) ++I;
    const int J = tile - I * (I + 1) / 2;
    double s = 0.0;
    const double *src = tpart + (size_t)tile * 256 + el;
    const size_t stride = (size_t)ntile * 256;
    int c = grp;
    for (; c + 7 * NG < nchunk; c += 8 * NG) {                // 8 independent loads in flight, summed in chunk order
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = src[(size_t)(c + NG * k) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += t[k];
    }
    for (; c < nchunk; c += NG) s += src[(size_t)c * stride];
    part[grp][e] = s;
    __syncthreads();
    if (grp == 0) {
        double t[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) t[g] = part[g][e];
#pragma unroll
        for (int h = 1; h < NG; h <<= 1)                       // fixed pairwise tree
#pragma unroll
            for (int g = 0; g + h < NG; g += 2 * h) t[g] += t[g + h];
        const int reg = el >> 6, lane = el & 63;
        const int row = 16 * I + (lane >> 4) + 4 * reg, col = 16 * J + (lane & 15);
        if (row < q && col < q && row >= col) {
            M[(size_t)col * q + row] = t[0];
            M[(size_t)row * q + col] = t[0];
        }
    }
}

__global__ __launch_bounds__(1024) void moments_reduce_kernel(const double *__restrict__ tpart,
                                                               const double *__restrict__ vpart, int p, int ntc,
                                                               int ntile, int nchunk, int nchunk_v, int aug, double *__restrict__ M)
{
    __shared__ double part[4][256];
    const int q = p + 2;
    const int lim = aug ? q : p;      // tiles cover Z = [X | y | 1] (aug) or X only
    const int b = blockIdx.x, e = threadIdx.x & 255, grp = threadIdx.x >> 8;
    if (b < ntile) {
        int I = (int)((sqrtf(8.0f * (float)b + 1.0f) - 1.0f) * 0.5f);
        while (I * (I + 1) / 2 > b) --I;
        while ((I + 1) * (I + 2) / 2 <= b) ++I;
        const int J = b - I * (I + 1) / 2;
        double s = 0.0;
        const double *src = tpart + (size_t)b * 256 + e;
        const size_t stride = (size_t)ntile * 256;
        int c = grp;
        for (; c + 28 < nchunk; c += 32) {                // 8 independent loads in flight, summed in chunk order
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = src[(size_t)(c + 4 * k) * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += t[k];
        }
        for (; c < nchunk; c += 4) s += src[(size_t)c * stride];
        part[grp][e] = s;
        __syncthreads();
        s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
        const int reg = e >> 6, lane = e & 63;
        const int row = 16 * I + (lane >> 4) + 4 * reg, col = 16 * J + (lane & 15);
        if (grp == 0 && row < lim && col < lim && row >= col) {
            M[(size_t)col * q + row] = s;
            M[(size_t)row * q + col] = s;
        }
    } else if (!aug) {
        // vector partials (X'y, column sums, y sums, count: vw doubles per chunk), 256 elements per extra workgroup, summed
        // like a tile: four thread groups over the chunks, eight loads in flight.  (One thread walking all chunks with
        // dependent loads made this tail the whole kernel: 0.25 ms at p = 512, 0.4 ms at p = 256 with 688 chunks; a first stage
        // over contiguous tile runs, tried on the theory that the 1 MB stride was the problem, made it slower: 76 vs 59 us.)
        const int vw = 32 * ntc + 4;
        const int idx = (b - ntile) * 256 + e;
        double s = 0.0;
        if (idx < vw) {
            const double *src = vpart + idx;
            int c = grp;
            for (; c + 28 < nchunk_v; c += 32) {
                double t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = src[(size_t)(c + 4 * k) * vw];
#pragma unroll
                for (int k = 0; k < 8; ++k) s += t[k];
            }
            for (; c < nchunk_v; c += 4) s += src[(size_t)c * vw];
        }
        part[grp][e] = s;
        __syncthreads();
        s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
        if (grp == 0 && idx < vw) {
            if (idx < 16 * ntc) {                      // sum (x_j - c_j)
                if (idx < p) { M[(size_t)idx * q + (p + 1)] = s; M[(size_t)(p + 1) * q + idx] = s; }
            } else if (idx < 32 * ntc) {               // sum (x_j - c_j)(y - c_y)
                const int j = idx - 16 * ntc;
                if (j < p) { M[(size_t)j * q + p] = s; M[(size_t)p * q + j] = s; }
            } else if (idx == 32 * ntc) { M[(size_t)p * q + (p + 1)] = s; M[(size_t)(p + 1) * q + p] = s; }      // sum (y - c_y)
            else if (idx == 32 * ntc + 1) M[(size_t)p * q + p] = s;                                              // sum (y - c_y)^2
            else if (idx == 32 * ntc + 2) M[(size_t)(p + 1) * q + (p + 1)] = s;                                  // rows
        }
    }
}

int launch_moments_reduce(hipStream_t s, const GramPlan &pl, const double *tpart, const double *vpart, double *moments)
{
    const int nvblk = pl.tri ? 0 : (32 * pl.ntc + 4 + 255) / 256;
    if (pl.tri && pl.nchunk >= 64) {        // all tiles cover Z: no vector partials
        hipLaunchKernelGGL(moments_reduce_split_kernel<4>, dim3(pl.ntile * 4), dim3(1024), 0, s, tpart, pl.p, pl.ntile, pl.nchunk, moments);
        OEM_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(moments_reduce_kernel, dim3(pl.ntile + nvblk), dim3(1024), 0, s, tpart, vpart, pl.p,
                       pl.ntc, pl.ntile, pl.nchunk, pl.nchunk, pl.tri, moments);
    OEM_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// finalize: shifted moments -> standardisation constants, XX, XY
//   sem 0 (DataStd + oemDense): ref src/DataStd.h:94-267, src/oem_dense.h:483,704-707
//   sem 1 (oemBig):             ref src/oem_big.h:731-842, 469-545
// ------------------------------------------------------------------------------------------------
struct Mom {
    const double *M; const double *sums; int p, q; double n;
    bool shift;
    __device__ double c(int j) const { return shift ? sums[j] / sums[p + 1] : 0.0; }
    __device__ double s(int j) const { return M[(size_t)j * q + (p + 1)]; }              // sum (z_j - c_j), j <= p
    __device__ double mu(int j) const { return c(j) + s(j) / n; }
    __device__ double cen(int i, int j) const { return M[(size_t)j * q + i] - s(i) * s(j) / n; }   // centred cross product
    // A column whose entries are all the same value is EXACTLY zero once the reference has centred it (src/DataStd.h:219-262), and
    // its scale then falls back to 1 (:237-240).  One-pass moments leave rounding noise instead: about the shift (which such a
    // column always triggers) the noise is ~ n (eps mean)^2, far below anything a column that really varies can produce.
    __device__ bool flat(int i) const { const double t = 32.0 * 2.220446049250313e-16 * fabs(mu(i)); return cen(i, i) <= n * t * t; }
    __device__ double raw(int i, int j) const { return cen(i, j) + n * mu(i) * mu(j); }            // sum z_i z_j
};

__global__ __launch_bounds__(256) void finalize_kernel(const double *__restrict__ Mbuf, const double *__restrict__ sums,
                                                        int p, int sem, int standardize, int intercept,
                                                        double *__restrict__ xx, double *__restrict__ xy,
                                                        double *__restrict__ stats)
{
    Mom m; m.M = Mbuf; m.sums = sums; m.p = p; m.q = p + 2; m.n = Mbuf[(size_t)(p + 1) * (p + 2) + (p + 1)];
    {
        __shared__ int need_sh;
        if (threadIdx.x == 0) need_sh = 0;
        __syncthreads();
        if (sums) for (int j = threadIdx.x; j <= p; j += blockDim.x) if (column_needs_shift(sums, p, j)) need_sh = 1;
        __syncthreads();
        m.shift = need_sh != 0;
    }
    const double n = m.n;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    // Moments about 0 (no sums) that DataStd will centre: say so if some column has mean^2 > 2^8 var -- the predicate of
    // oemgpu.h on the full data instead of a sample -- so that the caller can redo the pass about a shift.  oemBig / oemXvalDense
    // use the raw moments themselves (the reference accumulates them about 0 too): nothing to advise.
    __shared__ int advise;
    if (threadIdx.x == 0) advise = 0;
    __syncthreads();
    if (!sums && sem == OEMGPU_SEM_DENSE && blockIdx.x == 0) {
        const int q = p + 2;
        for (int j = threadIdx.x; j <= p; j += blockDim.x) {
            const double mean = Mbuf[(size_t)j * q + (p + 1)] / n;
            double var = Mbuf[(size_t)j * q + j] / n - mean * mean;
            if (!(var > 0.0)) var = 0.0;
            if (mean * mean > 256.0 * var) advise = 1;
        }
        __syncthreads();
    }
    if (tid == 0) { stats[stats_shift_flag(p)] = m.shift ? 1.0 : 0.0; stats[stats_shift_flag(p) + 1] = advise ? 1.0 : 0.0; }
    if (sem == OEMGPU_SEM_DENSE) {
        const int flag = (standardize ? 1 : 0) + 2 * (intercept ? 1 : 0);
        // y
        double meany = 0.0, scaley = 1.0, yy;
        const double cyy = fmax(m.cen(p, p), 0.0);
        if (flag == 1) { scaley = sqrt(cyy) / sqrt(n); yy = m.raw(p, p) / (scaley * scaley); }
        else if (flag >= 2) { meany = m.mu(p); scaley = sqrt(cyy) * (1.0 / sqrt(n)); yy = cyy / (scaley * scaley); }
        else yy = m.raw(p, p);
        if (tid == 0) { stats[0] = meany; stats[1] = scaley; stats[2] = yy; stats[3] = n; }
        for (int idx = tid; idx < p * p; idx += nth) {
            const int i = idx % p, j = idx / p;
            double si = 1.0, sj = 1.0;
            const bool fi = m.flat(i), fj = m.flat(j);
            if (flag & 1) {
                si = (flag == 1) ? sqrt(fmax(m.cen(i, i), 0.0)) / sqrt(n) : sqrt(fmax(m.cen(i, i), 0.0)) * (1.0 / sqrt(n));
                sj = (flag == 1) ? sqrt(fmax(m.cen(j, j), 0.0)) / sqrt(n) : sqrt(fmax(m.cen(j, j), 0.0)) * (1.0 / sqrt(n));
                if (si == 0.0 || fi) si = 1.0;
                if (sj == 0.0 || fj) sj = 1.0;
            }
            const double g = (flag >= 2) ? ((fi || fj) ? 0.0 : m.cen(i, j)) : m.raw(i, j);
            xx[(size_t)j * p + i] = g / (si * sj) / n;
            if (i == 0) {
                const double gy = (flag >= 2) ? (fj ? 0.0 : m.cen(j, p)) : m.raw(j, p);
                xy[j] = gy / (sj * scaley) / n;
                stats[4 + j] = (flag >= 2) ? m.mu(j) : 0.0;
                stats[4 + p + j] = sj;
            }
        }
    } else {
        const int off = intercept ? 1 : 0, qq = p + off;
        if (tid == 0) { stats[0] = 0.0; stats[1] = 1.0; stats[2] = m.raw(p, p); stats[3] = n; }
        for (int idx = tid; idx < p * p; idx += nth) {
            const int i = idx % p, j = idx / p;
            double ci = 1.0, cj = 1.0;
            if (standardize) {
                double a = m.raw(i, i) / (n - 1.0), b = m.raw(j, j) / (n - 1.0);
                if (a == 0.0) a = 1.0;
                if (b == 0.0) b = 1.0;
                ci = 1.0 / sqrt(a); cj = 1.0 / sqrt(b);
            }
            xx[(size_t)(j + off) * qq + (i + off)] = ci * m.raw(i, j) * cj / n;
            if (i == 0) {
                xy[j + off] = m.raw(j, p) * cj / n;
                stats[4 + j] = 0.0;
                stats[4 + p + j] = cj;
                if (intercept) {
                    const double cs = n * m.mu(j) * cj / n;     // colsums * colsq_inv / nobs
                    xx[(size_t)(j + 1) * qq] = cs;
                    xx[(size_t)(j + 1)] = cs;
                }
            }
        }
        if (tid == 0 && intercept) { xx[0] = 1.0; xy[0] = n * m.mu(p) / n; }
    }
}

int launch_finalize(hipStream_t s, const double *moments, const double *sums, int p, int sem, int standardize,
                    int intercept, double *xx, double *xy, double *stats)
{
    int blocks = (p * p + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(256), 0, s, moments, sums, p, sem, standardize, intercept, xx,
                       xy, stats);
    OEM_HIP(hipGetLastError());
    return 0;
}

// oem.xtx: XY = xty / s, XX = S^-1 xtx S^-1  (ref src/oem_xtx.h:347-356, 520-531)
__global__ __launch_bounds__(256) void xtx_prepare_kernel(const double *__restrict__ xtx, const double *__restrict__ xty,
                                                           const double *__restrict__ sinv, int p,
                                                           double *__restrict__ xx, double *__restrict__ xy,
                                                           double *__restrict__ stats)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = tid; idx < (size_t)p * p; idx += nth) {
        const int i = (int)(idx % p), j = (int)(idx / p);
        xx[idx] = sinv ? sinv[i] * xtx[idx] * sinv[j] : xtx[idx];
        if (i == 0) xy[j] = sinv ? xty[j] * sinv[j] : xty[j];
    }
    if (tid == 0) { stats[0] = 0.0; stats[1] = 1.0; stats[2] = 0.0; stats[3] = 1.0; stats[stats_shift_flag(p)] = 0.0; stats[stats_shift_flag(p) + 1] = 0.0; }
}

int launch_xtx_prepare(hipStream_t s, const double *xtx, const double *xty, const double *sf_inv, int p, double *xx,
                       double *xy, double *stats)
{
    size_t blocks = ((size_t)p * p + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(xtx_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, s, xtx, xty, sf_inv, p, xx, xy, stats);
    OEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace oemgpu
