// gram_sb.hip -- the shared-slab moment kernel (p + 2 > 112, 16-byte aligned X): a workgroup owns a super-block of tiles over all rows of
// its chunk, the column fragments of an 8-row slab are DMA'd once into a workgroup-shared LDS ring and every wave multiplies its own tile
// block from them.  A translation unit of its own since round 5: its (kind of super-block, wave, shift) bodies are 32 straight-line
// instantiations, two thirds of gram.hip's compile time.  Plan (gram_plan, gram_sb_deal) and launcher of the other forms: gram.hip.
// References: ref src/oem_dense.h:316-366 (XtX), src/oem_base.h:90-110; DESIGN.md section 3.1.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "gram_dev.hpp"

namespace oemgpu {

#ifdef OEM_GRAM_DIAG
__device__ unsigned long long g_gram_diag[8];
extern "C" __attribute__((visibility("default"))) int oemgpu_gram_sb_diag_read(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gram_diag), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif

// ------------------------------------------------------------------------------------------------
// Shared-slab kernel (p + 2 > 112, 16-byte aligned X).  profiles/r1_pmc_blk_p512.txt: gram_blk_kernel is bound by
// RE-READS -- every 4x4 tile block streams its own 8 column fragments, 26.8 GB reach the fabric for 4.1 GB of X.  Here a
// workgroup owns a SUPER-BLOCK of 2x2 tile blocks (8 x 8 tiles) over ALL rows of its chunk: the 16 (diagonal: 8)
// column fragments of an 8-row slab are DMA'd ONCE into a workgroup-shared LDS ring (each wave issues a quarter of
// them), and every wave multiplies a different 4x4 tile block from the same slab -- half the fragment traffic per tile.
// One s_barrier per slab hands a slot over: a wave waits for its OWN DMAs of slab k+1 (exact vmcnt), the barrier then
// makes everybody's visible; the slot of slab k-1 is refilled after that same barrier, by which time every wave has
// consumed it.  Tiles are wave-private, so the epilogue stores accumulators straight to the partial buffer.
// Diagonal super-blocks: wave 0 -> block (0,0) (10 tiles), wave 2 -> (1,0) (16 tiles), wave 3 -> (1,1) (10 tiles),
// wave 1 only helps with the DMA; X'y and the column sums ride on the VALU of the two diagonal blocks as in
// gram_blk_kernel, with y DMA'd beside the fragments.
// ------------------------------------------------------------------------------------------------
#ifndef OEM_SB_NSLOT
#define OEM_SB_NSLOT 9
#endif
constexpr int SB_NSLOT = OEM_SB_NSLOT;           // ring depth: the prefetch distance is (NSLOT - 3) slabs

// Super-block heights.  A super-block row is HI = 8, 6 or 4 tile columns high (gram_plan deals the ntc tile columns into eights,
// at most one six and one four so that little of the last one is padding: with eights alone p = 160 multiplied 136 tiles' worth for
// 55 real ones, p = 300 300 for 190 -- tools/gram_band.sh, round 5); the eights come first, then the six, then the four, so an
// off-diagonal super-block (SI > SJ) is 8 x 8, 6 x 8, 4 x 8 or 4 x 6 tiles and its four waves take (HI / 2) x (HJ / 2) tiles each.
// row of the lower triangle that holds row-major index tt (tt = I (I + 1) / 2 + J, J <= I)
constexpr int tri_row(int tt) { int I = 0; while ((I + 1) * (I + 2) / 2 <= tt) ++I; return I; }
// diagonal super-block of height H: wave W multiplies tiles sb_diag_t0(H, W) .. sb_diag_t0(H, W + 1) - 1 of the row-major triangle
// (H = 8: 36 tiles, 9 each; H = 6: 21 tiles, 6 + 5 + 5 + 5; H = 4: 10 tiles, 3 + 3 + 2 + 2)
constexpr int sb_diag_t0(int H, int W) { return H == 8 ? 9 * W : H == 6 ? (W == 0 ? 0 : 1 + 5 * W) : (W < 2 ? 3 * W : 2 + 2 * W); }

// One slab of one wave of a super-block.  Up to eight fragments in registers.
//   off-diagonal super-block: s.v[0 .. NR-1] = the wave's tile rows, s.v[NR .. NR+NC-1] = its tile columns, NR x NC tiles (I, J);
//   diagonal super-block (H = NR = NC; H (H + 1) / 2 tiles): s.v[f] = fragment f, wave W multiplies its share of the row-major
//   triangle and carries X'y / column sums of fragments 2 W, 2 W + 1 (where those exist) on the VALU.
template <bool DIAGSB, int W, bool XF, bool MASKED, int NR, int NC, typename Hook = NoHook>
__device__ __forceinline__ void sb_consume(Slab<8> &s, const double (&c)[8], double cy, double (&sx)[2], double (&sxy)[2],
                                           double &sy, double &syy, int64_t r, int64_t n, Hook &&hook = NoHook())
{
    constexpr int NFR = DIAGSB ? NR : NR + NC;
    double m0 = 1.0, m1 = 1.0;
    if (MASKED) { m0 = (r < n) ? 1.0 : 0.0; m1 = (r + 1 < n) ? 1.0 : 0.0; }
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
        if (XF) { s.v[f].x -= c[f]; s.v[f].y -= c[f]; }
        if (MASKED) { s.v[f].x *= m0; s.v[f].y *= m1; }
    }
    if (DIAGSB) {
        double y0 = s.y.x - cy, y1 = s.y.y - cy;
        if (MASKED) { y0 *= m0; y1 *= m1; }
        static_for<2>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            if constexpr (2 * W + k < NR) {
                const v2d v = s.v[2 * W + k];
                sx[k] = (sx[k] + v.x) + v.y;
                sxy[k] = fma(v.x, y0, sxy[k]);
                sxy[k] = fma(v.y, y1, sxy[k]);
            }
        });
        if (W == 0) { sy = (sy + y0) + y1; syy = fma(y0, y0, syy); syy = fma(y1, y1, syy); }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3" ::: "memory");                          // VALU write -> MFMA read (hipcc pads nothing for asm)
    static_for<2>([&](auto E) {
        constexpr int e = decltype(E)::value;
        if constexpr (DIAGSB) {
            constexpr int T0 = sb_diag_t0(NR, W), NT = sb_diag_t0(NR, W + 1) - T0;
            static_for<NT>([&](auto T_) {
                constexpr int t = decltype(T_)::value, tt = T0 + t, I = tri_row(tt), J = tt - I * (I + 1) / 2;
                AccTile<t>::mfma(s.v[I][e], s.v[J][e]);
                hook(std::integral_constant<int, e * NT + t>{});
            });
        } else {
            static_for<NR * NC>([&](auto T_) {
                constexpr int t = decltype(T_)::value;
                AccTile<t>::mfma(s.v[t / NC][e], s.v[NR + t % NC][e]);
                hook(std::integral_constant<int, e * NR * NC + t>{});
            });
        }
    });
    __builtin_amdgcn_sched_barrier(0);
}

template <bool DIAGSB, int W, bool XF, int HI, int HJ>
__device__ __forceinline__ void gram_sb_body(const double *__restrict__ x, int64_t n, int64_t ld, int p,
                                             const double *__restrict__ y, const double *__restrict__ sums, int ntc, int TI,
                                             int TJ /* first tile column of the row / column group */, int64_t row_begin, int steps, double *__restrict__ tdst,
                                             double *__restrict__ vdst, double *lds)
{
    static_assert(HI <= HJ && (HI == 4 || HI == 6 || HI == 8) && (HJ == 4 || HJ == 6 || HJ == 8) && (!DIAGSB || HI == HJ), "super-block heights");
    constexpr int NR = DIAGSB ? HI : HI / 2, NC = DIAGSB ? HI : HJ / 2;   // the wave's tile block (diagonal: the whole triangle's fragments)
    constexpr int NFR = DIAGSB ? HI : NR + NC;         // fragments in this wave's registers
    constexpr int F = DIAGSB ? HI : HI + HJ;           // x fragments per slab
    constexpr int NDMA = (F + 3) / 4;                  // ... dealt to the four waves (a ragged deal repeats the last fragment)
    constexpr int DPW = NDMA + (DIAGSB ? 1 : 0);       // DMAs per wave per slab (diagonal: + its own copy of y)
    constexpr int NFETCH = NFR + (DIAGSB ? 1 : 0);     // ring reads per wave per slab
    constexpr int NTW = DIAGSB ? sb_diag_t0(HI, W + 1) - sb_diag_t0(HI, W) : NR * NC;   // tiles of this wave
    constexpr int NMFMA = 2 * NTW;
    constexpr int NACT = 1 + NFETCH + DPW;             // hand-over + ring reads + DMAs, one after each of the first MFMAs
    constexpr int SLOT_B = (F + (DIAGSB ? 4 : 0)) * 1024, NSLOT = SB_NSLOT;
    static_assert((NSLOT - 2) * DPW <= 63, "vmcnt field is 6 bits");
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, q = lane >> 4;
    const gptr_t xg = (gptr_t)x, yg = (gptr_t)y;
    const double inv_cnt = XF ? 1.0 / sums[p + 1] : 0.0;
    const double cy = XF ? sums[p] * inv_cnt : 0.0;
    // tile column of fragment f of this super-block
    auto frag_tile = [&](int f) { return DIAGSB ? TI + f : (f < HI ? TI + f : TJ + (f - HI)); };
    auto frag_col = [&](int f) { const int col = 16 * frag_tile(f) + i; return col < p ? col : p - 1; };
    // ---- DMA duty of this wave: fragments w, w + 4 (, w + 8, w + 12) (+ its own copy of y in a diagonal super-block).
    // Addressing: one scalar base per fragment (first column of its tile at the chunk's first row, bumped by a scalar
    // add per slab) + a 32-bit lane offset (column within the tile, row pair) -- 64-bit VALU adds run on the DP units.
    gptr_t dbase[NDMA];
    unsigned doff[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
        const int fr = w + 4 * k < F ? w + 4 * k : F - 1;
        int t0 = 16 * frag_tile(fr);                              // wave-uniform
        if (t0 > p - 1) t0 = p - 1;
        dbase[k] = xg + (size_t)t0 * ld + row_begin;
        doff[k] = (unsigned)(((int64_t)(frag_col(fr) - t0) * ld + 2 * q) * 8);
    }
    gptr_t ybase = yg + row_begin;
    const unsigned yoff = (unsigned)(2 * q * 8);
    // ---- the fragments this wave multiplies: registers 0..7 <- ring fragments rf[0..7]
    int rb = 0, cb = 0;                                            // off-diagonal: first fragment of the row / column group
    if (!DIAGSB) { rb = __builtin_amdgcn_readfirstlane(NR * (w >> 1)); cb = __builtin_amdgcn_readfirstlane(HI + NC * (w & 1)); }
    auto reg_frag = [&](int f) { return DIAGSB ? f : (f < NR ? rb + f : cb + f - NR); };
    double c[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) c[f] = (XF && f < NFR) ? sums[frag_col(reg_frag(f))] * inv_cnt : 0.0;
    double sx[2] = {0.0, 0.0}, sxy[2] = {0.0, 0.0}, sy = 0.0, syy = 0.0;
    static_for<16>([&](auto T_) { AccTile<decltype(T_)::value>::zero(); });
    asm volatile("s_nop 7" ::: "memory");

    // full slabs of this chunk (8 rows each; every wave walks all of them)
    const int64_t rows_chunk = (int64_t)steps * 64;
    int64_t rows = n - row_begin; if (rows > rows_chunk) rows = rows_chunk; if (rows < 0) rows = 0;
    const int ns = (int)(rows / 8);
    const unsigned ring = (unsigned)(size_t)lds;
    const v2d *rd = reinterpret_cast<const v2d *>(lds) + lane;
    auto issue1 = [&](int slot, auto K_) {                         // one DMA of this wave's share of the next slab
        constexpr int k = decltype(K_)::value;
        const unsigned dst = ring + (unsigned)slot * SLOT_B;
        if constexpr (k < NDMA) {
            unsigned fr = (unsigned)(w + 4 * k);
            if constexpr (4 * k + 3 >= F) fr = fr < (unsigned)F ? fr : (unsigned)(F - 1);   // the repeat lands where the original does
            set_m0(dst + fr * 1024); glds_s<0>(doff[k], dbase[k]); dbase[k] += 8;
        }
        else { set_m0(dst + (unsigned)(F + w) * 1024); glds_s<0>(yoff, ybase); ybase += 8; }
    };
    auto issue = [&](int slot) { static_for<DPW>([&](auto K_) { issue1(slot, K_); }); };
    Slab<8> sa, sb;
    sa.y = v2d{0.0, 0.0}; sb.y = sa.y;
    auto fetch1 = [&](Slab<8> &s, int slot, auto J_) {
        constexpr int j = decltype(J_)::value;
        const v2d *b = rd + (slot * SLOT_B) / 16;
        if constexpr (j < NFR) s.v[j] = b[reg_frag(j) * 64];
        else s.y = b[(F + w) * 64];
    };
    auto fetch = [&](Slab<8> &s, int slot) { static_for<NFETCH>([&](auto J_) { fetch1(s, slot, J_); }); };
    auto next = [](int v) { return v + 1 == NSLOT ? 0 : v + 1; };
    // prologue: NSLOT - 2 slabs in flight
    const int npre = ns < NSLOT - 2 ? ns : NSLOT - 2;
    for (int j = 0; j < npre; ++j) issue(j);
    int islot = npre % NSLOT, rslot = 0, issued = npre;
    if (ns > 0) {
        if (npre == NSLOT - 2) wait_vm<(NSLOT - 3) * DPW>(); else wait_vm<0>();
        __syncthreads();
        fetch(sa, 0);
        rslot = 1;
    }
    // Steady state, two slabs per trip and not a single conditional inside.  The hand-over (exact vmcnt wait: this
    // wave's DMAs of slab k+1 have landed; barrier: everybody's have, and everybody is done with slab k-1), the ring
    // reads of slab k+1 and this wave's DMAs of slab k+NSLOT-2 (into the slot of slab k-2) sit BETWEEN the MFMAs of
    // slab k (hook after MFMA m), where scalar, LDS and VMEM instructions issue for free.
    int k = 0;
    auto steady = [&](Slab<8> &use, Slab<8> &nxt) {
        const int rs = rslot, is = islot;
        sb_consume<DIAGSB, W, XF, false, NR, NC>(use, c, cy, sx, sxy, sy, syy, 0, n, [&](auto M_) {
            constexpr int m = decltype(M_)::value;
            // action a goes after MFMA min(a, NMFMA - 1): a wave with fewer MFMAs than actions does the rest after its last one
            static_for<NACT>([&](auto A_) {
                constexpr int act = decltype(A_)::value, at = act < NMFMA ? act : NMFMA - 1;
                if constexpr (at == m) {
                    if constexpr (act == 0) { wait_vm<(NSLOT - 4) * DPW>(); __syncthreads(); }
                    else if constexpr (act <= NFETCH) fetch1(nxt, rs, std::integral_constant<int, act - 1>{});
                    else issue1(is, std::integral_constant<int, act - NFETCH - 1>{});
                }
            });
        });
        rslot = next(rslot); islot = next(islot);
    };
    while (issued + 2 <= ns) {
        steady(sa, sb);
        steady(sb, sa);
        issued += 2; k += 2;
    }
    // drain
    auto step = [&](Slab<8> &use, Slab<8> &nxt) {
        if (k + 1 < ns) {
            if (issued - (k + 2) >= NSLOT - 4) wait_vm<(NSLOT - 4) * DPW>(); else wait_vm<0>();
        }
        __syncthreads();
        if (k + 1 < ns) { fetch(nxt, rslot); rslot = next(rslot); }
        if (issued < ns) { issue(islot); islot = next(islot); ++issued; }
        sb_consume<DIAGSB, W, XF, false, NR, NC>(use, c, cy, sx, sxy, sy, syy, 0, n);
        ++k;
    };
    while (k < ns) {
        step(sa, sb);
        if (k < ns) step(sb, sa);
    }
    wait_vm<0>();
    // ragged tail of the data set (fewer than 8 rows left): masked loads straight from global memory
    if (rows - 8 * (int64_t)ns > 0) {
        const int64_t r = row_begin + 8 * (int64_t)ns + 2 * q;
        const int64_t r0 = r < n ? r : n - 1, r1 = r + 1 < n ? r + 1 : n - 1;
        Slab<8> t;
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
            const gptr_t pf = xg + (size_t)frag_col(reg_frag(f)) * ld;
            t.v[f].x = pf[r0]; t.v[f].y = pf[r1];
        }
        t.y.x = yg[r0]; t.y.y = yg[r1];
        sb_consume<DIAGSB, W, XF, true, NR, NC>(t, c, cy, sx, sxy, sy, syy, r, n);
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // ---- results: tiles are wave-private -> straight to the partial buffer; vector sums of the diagonal super-blocks
    if constexpr (DIAGSB) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            sx[k2] += shfl_xor_d(sx[k2], 16);  sx[k2] += shfl_xor_d(sx[k2], 32);
            sxy[k2] += shfl_xor_d(sxy[k2], 16); sxy[k2] += shfl_xor_d(sxy[k2], 32);
        }
        sy += shfl_xor_d(sy, 16);   sy += shfl_xor_d(sy, 32);
        syy += shfl_xor_d(syy, 16); syy += shfl_xor_d(syy, 32);
        if (q == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int T = TI + 2 * W + k2;
                if (2 * W + k2 < HI && T < ntc) { vdst[16 * T + i] = sx[k2]; vdst[16 * ntc + 16 * T + i] = sxy[k2]; }
            }
            if (W == 0 && i == 0 && TI == 0) {
                vdst[32 * ntc] = sy; vdst[32 * ntc + 1] = syy; vdst[32 * ntc + 2] = (double)rows; vdst[32 * ntc + 3] = 0.0;
            }
        }
        static_for<NTW>([&](auto T_) {
            constexpr int t = decltype(T_)::value, tt = sb_diag_t0(HI, W) + t, I = tri_row(tt), J = tt - I * (I + 1) / 2;
            const int gi = TI + I, gj = TI + J;
            if (gi < ntc && gj < ntc) {
                double *dst = tdst + (size_t)(gi * (gi + 1) / 2 + gj) * 256;
                static_for<4>([&](auto R_) { constexpr int r = decltype(R_)::value; dst[r * 64 + lane] = AccTile<t>::template read<r>(); });
            }
        });
    } else {
        const int bi = TI + NR * (w >> 1), bj = TJ + NC * (w & 1);
        static_for<NR * NC>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            const int gi = bi + t / NC, gj = bj + t % NC;
            if (gi < ntc && gj < ntc) {
                double *dst = tdst + (size_t)(gi * (gi + 1) / 2 + gj) * 256;
                static_for<4>([&](auto R_) { constexpr int r = decltype(R_)::value; dst[r * 64 + lane] = AccTile<t>::template read<r>(); });
            }
        });
    }
}

__global__ __launch_bounds__(256) void gram_sb_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                       const double *__restrict__ sums, double *__restrict__ tpart,
                                                       double *__restrict__ vpart, GramDims a, int n8, int n6, int n4 /* super-block rows of height 8 / 6 / 4 */)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // Launch order = longest first: the super-blocks of all row chunks by the MFMAs a wave issues per slab -- off the diagonal 8 x 8
    // tiles 32, 6 x 8 24, diagonal eight 18, 4 x 8 16, 4 x 6 12, diagonal six 12, diagonal four 6.  Workgroups are handed
    // to CUs as CUs fall free, so the launch ends one (partial) workgroup after the work runs out: with the short ones last that
    // tail is short (measured at p = 256: 11 % of the launch in mixed order; gram_plan also picks the chunk count whose simulated
    // tail is smallest).  Blocks of one row chunk still share blockIdx % 8 (one XCD).
    const int L = blockIdx.x, xcd = L & 7, ng = a.nchunk / 8;
    int s = L >> 3, SI = 0, SJ = 0, kind = 0;
    auto tri_decode = [](int ob, int &I, int &J) {                 // ob = I (I - 1) / 2 + J, J < I
        I = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)ob)) * 0.5f);
        while (I * (I - 1) / 2 > ob) --I;
        while ((I + 1) * I / 2 <= ob) ++I;
        J = ob - I * (I - 1) / 2;
    };
    const int cnt[SB_KINDS] = {n8 * (n8 - 1) / 2, n6 * n8, n8, n4 * n8, n4 * n6, n6, n4};       // (n6, n4 <= 1: gram_sb_deal)
#pragma unroll
    for (int k = 0; k < SB_KINDS - 1; ++k) {
        if (kind == k) { if (s < cnt[k] * ng) break; s -= cnt[k] * ng; kind = k + 1; }
    }
    const int ck = cnt[kind] > 0 ? cnt[kind] : 1, ob = s % ck, chunk = (s / ck) * 8 + xcd;
    switch (kind) {
    case 0: tri_decode(ob, SI, SJ); break;                         // 8 x 8
    case 1: SI = n8 + ob / n8; SJ = ob % n8; break;                // 6 x 8
    case 2: SI = SJ = ob; break;                                   // diagonal eight
    case 3: SI = n8 + n6 + ob / n8; SJ = ob % n8; break;           // 4 x 8
    case 4: SI = n8 + n6 + ob / n6; SJ = n8 + ob % n6; break;      // 4 x 6
    case 5: SI = SJ = n8 + ob; break;                              // diagonal six
    default: SI = SJ = n8 + n6 + ob; break;                        // diagonal four
    }
    auto first_tile = [&](int S) { return S < n8 ? 8 * S : (S < n8 + n6 ? 8 * n8 + 6 * (S - n8) : 8 * n8 + 6 * n6 + 4 * (S - n8 - n6)); };
    const int TI = first_tile(SI), TJ = first_tile(SJ);            // first tile columns of the row / column group
#ifdef OEM_SB_EXP_SAMEROWS                                  // experiment (never the product): every workgroup multiplies the rows of its XCD's first chunk -- the same
    const int64_t row_begin = (int64_t)(chunk & 7) * a.steps * 64;      // MFMA work with (almost) no HBM traffic: what would one read of X be worth?
#else
    const int64_t row_begin = (int64_t)chunk * a.steps * 64;
#endif
    double *tdst = tpart + (size_t)chunk * a.ntile * 256;
    double *vdst = vpart + (size_t)chunk * (32 * a.ntc + 4);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef OEM_GRAM_DIAG
    const unsigned long long dg_c0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // every (kind of super-block, wave, shift) combination is its own straight-line body: a taken scalar branch per slab
    // costs ~80 cycles (measured on the path kernels), a dispatch up here costs nothing
#define OEM_SB(D, W, XF, HI, HJ) gram_sb_body<D, W, XF, HI, HJ>(x, a.n, a.ld, a.p, y, sums, a.ntc, TI, TJ, row_begin, a.steps, tdst, vdst, lds)
#define OEM_SB_DIAG(XF, H)                                                                          \
    do {                                                                                            \
        if (w == 0) OEM_SB(true, 0, XF, H, H); else if (w == 1) OEM_SB(true, 1, XF, H, H);          \
        else if (w == 2) OEM_SB(true, 2, XF, H, H); else OEM_SB(true, 3, XF, H, H);                 \
    } while (0)
#define OEM_SB_ALL(XF)                                                                              \
    do {                                                                                            \
        if (kind == 0) OEM_SB(false, 0, XF, 8, 8);                                                  \
        else if (kind == 1) OEM_SB(false, 0, XF, 6, 8);                                             \
        else if (kind == 2) OEM_SB_DIAG(XF, 8);                                                     \
        else if (kind == 3) OEM_SB(false, 0, XF, 4, 8);                                             \
        else if (kind == 4) OEM_SB(false, 0, XF, 4, 6);                                             \
        else if (kind == 5) OEM_SB_DIAG(XF, 6);                                                     \
        else OEM_SB_DIAG(XF, 4);                                                                    \
    } while (0)
    if (shift_needed_wave(sums, a.p)) OEM_SB_ALL(true); else OEM_SB_ALL(false);
#undef OEM_SB_ALL
#undef OEM_SB_DIAG
#undef OEM_SB
#ifdef OEM_GRAM_DIAG
    // one diagonal and one off-diagonal workgroup of the first row chunk: shader cycles, 100 MHz ticks (-> the clock held), 8-row slabs
    const int sbk = (SI == 0 && SJ == 0) ? 0 : ((SI == 1 && SJ == 0) ? 1 : -1);
    if (chunk == 0 && sbk >= 0 && threadIdx.x == 0) {
        int64_t rows = a.n - row_begin; if (rows > (int64_t)a.steps * 64) rows = (int64_t)a.steps * 64;
        g_gram_diag[4 * sbk + 0] = __builtin_amdgcn_s_memtime() - dg_c0;
        g_gram_diag[4 * sbk + 1] = __builtin_amdgcn_s_memrealtime() - dg_r0;
        g_gram_diag[4 * sbk + 2] = (unsigned long long)(rows / 8);
        g_gram_diag[4 * sbk + 3] = (unsigned long long)(SI == SJ);
    }
#endif
}

int launch_gram_sb(hipStream_t s, const GramPlan &pl, const double *x, const double *y, const double *sums, double *tpart, double *vpart, const GramDims &a)
{
    const int nsb = pl.n8 + pl.n6 + pl.n4, nsblk = nsb * (nsb + 1) / 2;
    const size_t shb = (size_t)SB_NSLOT * 16 * 1024;                // NSLOT x 16 KiB slots
    if (lds_limit_once(reinterpret_cast<const void *>(&gram_sb_kernel), shb)) return OEMGPU_ERR_HIP;
    hipLaunchKernelGGL(gram_sb_kernel, dim3(pl.nchunk * nsblk), dim3(256), shb, s, x, y, sums, tpart, vpart, a, pl.n8, pl.n6, pl.n4);
    OEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace oemgpu
