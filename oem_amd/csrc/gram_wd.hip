// gram_wd.hip -- the moment kernel for 225 <= p <= 256 (config 5's p = 256) and 161 <= p <= 192: ONE read of X.
//
// gram_sb_kernel covers 16 tile columns with three workgroups per row chunk (two diagonal super-blocks of 8 x 8 tiles and the
// off-diagonal one between them), each streaming the chunk's rows by itself: 32 fragment reads per 8-row slab for 16 distinct
// fragments -- the counters show 2.0 x the algorithmic HBM bytes at p = 256 (profiles/r5_c5_pmc_gram_sb.json), and the same MFMA work
// with the traffic taken away runs 15 % faster (tools/gram_samerows_ab.sh, profiles/r5_gram_one_read.txt).  Here ONE workgroup of EIGHT
// waves (two per SIMD, 256 registers each) owns all 136 tiles of the 16 x 16-tile triangle over the rows of its chunk: the 16
// fragments of a slab are DMA'd once into the workgroup's LDS ring (two per wave; wave 0 also brings y), every wave copies the eight
// fragments it multiplies to registers one slab ahead and issues its 17 tiles = 34 MFMAs per slab -- 68 per SIMD, the same issue
// floor as the three super-blocks (4,352 cycles per slab and chunk), with the second wave of a SIMD filling the first one's gaps.
//
//   waves 0 / 1: the two pairs of diagonal 4 x 4 tile blocks (fragments 0-7 / 8-15), minus three diagonal tiles each;
//   waves 2 .. 7: one off-diagonal 4 x 4 tile block each (tile rows x tile columns: 4-7 x 0-3, 8-11 x 0-3, 12-15 x 0-3, 8-11 x 4-7,
//                 12-15 x 4-7, 12-15 x 8-11) plus ONE of those diagonal tiles -- a diagonal tile needs one fragment, which the
//                 taker already holds.  17 tiles and 8 fragments per wave, 136 accumulator registers.
//   X'y and the column sums ride on the VALU, two fragments per wave (every fragment once), y from the ring.
// Groups of THREE tile columns (78 tiles, nine or ten per wave) serve 11-12 tile columns (161 <= p <= 192) the same way.
// UNITS: for 16 k tile columns, k >= 2 (p = 512, 1,024, ...), one launch runs per row chunk k such diagonal units and, between every two of
// them, two off-diagonal blocks of 8 x 16 tiles on eight waves (gram_od_body below) -- 80 fragment reads per slab at p = 512 where
// gram_sb_kernel's ten super-blocks make 128.
// Same partial layout as gram_sb_kernel (tiles by their global index, vector partials), same reduction behind it.
// References: ref src/oem_dense.h:316-366 (XtX), src/oem_big.h:455-534 (row blocks); DESIGN.md section 3.1 (docs/history.md section 3.1c for how it got there).
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "gram_dev.hpp"

namespace oemgpu {

namespace {

constexpr int WD_NSLOT = 9;                       // ring depth (slots of 4 B + 1 KiB: the fragments of a slab + y)

// ---- the deal (compile-time), for 4 x 4 groups of B tile columns (B = 4: 16 tile columns, 136 tiles, 17 per wave; B = 3: 12 tile
// columns, 78 tiles, 9 / 10 per wave).  Fragment f of wave W's registers is tile column wd_frag(W, f); tile t of wave W is (row
// fragment, column fragment) in REGISTER indices.
template <int B> constexpr int wd_frag(int W, int f)
{
    constexpr int rows[8] = {0, 2, 1, 2, 3, 2, 3, 3}, cols[8] = {1, 3, 0, 0, 0, 1, 1, 2};      // in groups of B tile columns
    return f < B ? B * rows[W] + f : B * cols[W] + (f - B);         // registers 0 .. B-1: the block's tile rows, B .. 2B-1: its tile columns
}
template <int B> constexpr int wd_ntiles(int W) { return W < 2 ? B * (B + 1) - 3 : B * B + 1; }
struct WdTile { int a, b; };                                        // register indices: tile (wd_frag(W, a), wd_frag(W, b)), a's column >= b's
template <int B> constexpr WdTile wd_tile(int W, int t)
{
    if (W >= 2) {
        if (t < B * B) return WdTile{t / B, B + t % B};            // the off-diagonal block
        // the diagonal tile taken over from waves 0 / 1: tile columns 0, 1, 2 (waves 2, 3, 4), 2B (wave 5), 3B (wave 6), 2B + 1 (wave 7)
        constexpr int reg[8] = {0, 0, B, B + 1, B + 2, 0, 0, B + 1};
        return WdTile{reg[W], reg[W]};
    }
    // waves 0 / 1: the lower triangles of two B x B tile blocks (registers 0 .. B-1 and B .. 2B-1) without three diagonal tiles:
    // wave 0 gives away (0,0) (1,1) (2,2); wave 1 (2B,2B) (2B+1,2B+1) (3B,3B) = registers (0,0) (1,1) (B,B)
    int k = 0;
    for (int blk = 0; blk < 2; ++blk)
        for (int I = 0; I < B; ++I)
            for (int J = 0; J <= I; ++J) {
                const int ra = B * blk + I, rb = B * blk + J;
                const bool given = I == J && (W == 0 ? (blk == 0 && I < 3) : ((blk == 0 && I < 2) || (blk == 1 && I == 0)));
                if (given) continue;
                if (k == t) return WdTile{ra, rb};
                ++k;
            }
    return WdTile{0, 0};
}
// the REGISTER fragments whose X'y / column sums wave W carries (every tile column once over the eight waves), -1: none
template <int B> constexpr int wd_sum_reg(int W, int k)
{
    if (B == 4) {      // tile columns: W0: 0,1  W1: 8,9  W2: 2,3  W3: 10,11  W4: 12,13  W5: 4,5  W6: 6,7  W7: 14,15
        constexpr int reg[8][2] = {{0, 1}, {0, 1}, {6, 7}, {2, 3}, {0, 1}, {4, 5}, {6, 7}, {2, 3}};
        return reg[W][k];
    }
    // B = 3, tile columns: W0: 0,1  W1: 6,7  W2: 2  W3: 8  W4: 9,10  W5: 3,4  W6: 5  W7: 11
    constexpr int reg[8][2] = {{0, 1}, {0, 1}, {5, -1}, {2, -1}, {0, 1}, {3, 4}, {5, -1}, {2, -1}};
    return reg[W][k];
}
// DMA duty of wave W: fragments (tile columns) W and W + 8 where those exist
template <int B> constexpr int wd_ndma(int W) { return W + 8 < 4 * B ? 2 : 1; }

typedef double v4d __attribute__((ext_vector_type(4)));            // the 17th tile of a wave: eight VGPRs (with two waves per SIMD the
                                                                    // accumulator file ends at a127 = sixteen tiles)
__device__ __forceinline__ void wd_mfma_v(v4d &acc, double a, double b)
{
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// shift: the column means live in LDS (cs: [16 tile columns][16] doubles behind the ring) and are subtracted as a slab's fragments arrive
// in registers -- eight constants per lane would not fit beside two slabs and a tile in 128 VGPRs
template <int B, int W, bool XF, bool MASKED, typename Hook = NoHook>
__device__ __forceinline__ void wd_consume(Slab<8> &s, v4d &acc16, const double *cs, double cy, double (&sx)[2], double (&sxy)[2],
                                           double &sy, double &syy, int64_t r, int64_t n, Hook &&hook = NoHook())
{
    double m0 = 1.0, m1 = 1.0;
    if (MASKED) { m0 = (r < n) ? 1.0 : 0.0; m1 = (r + 1 < n) ? 1.0 : 0.0; }
    if (XF || MASKED) {
        static_for<2 * B>([&](auto F_) {
            constexpr int f = decltype(F_)::value;
            if (XF) { const double c = cs[wd_frag<B>(W, f) * 16]; s.v[f].x -= c; s.v[f].y -= c; }
            if (MASKED) { s.v[f].x *= m0; s.v[f].y *= m1; }
        });
    }
    double y0 = s.y.x - cy, y1 = s.y.y - cy;
    if (MASKED) { y0 *= m0; y1 *= m1; }
    static_for<2>([&](auto K_) {
        constexpr int k = decltype(K_)::value;
        if constexpr (wd_sum_reg<B>(W, k) >= 0) {
            const v2d v = s.v[wd_sum_reg<B>(W, k)];
            sx[k] = (sx[k] + v.x) + v.y;
            sxy[k] = fma(v.x, y0, sxy[k]);
            sxy[k] = fma(v.y, y1, sxy[k]);
        }
    });
    if (W == 0) { sy = (sy + y0) + y1; syy = fma(y0, y0, syy); syy = fma(y1, y1, syy); }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3" ::: "memory");                          // VALU write -> MFMA read (hipcc pads nothing for asm)
    static_for<2>([&](auto E) {
        constexpr int e = decltype(E)::value;
        static_for<wd_ntiles<B>(W)>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            constexpr WdTile tl = wd_tile<B>(W, t);
            if constexpr (t < 16) AccTile<t>::mfma(s.v[tl.a][e], s.v[tl.b][e]);
            else wd_mfma_v(acc16, s.v[tl.a][e], s.v[tl.b][e]);
            hook(std::integral_constant<int, e * wd_ntiles<B>(W) + t>{});
        });
    });
    __builtin_amdgcn_sched_barrier(0);
}

template <int B, int W, bool XF>
__device__ __forceinline__ void gram_wd_body(const double *__restrict__ x, int64_t n, int64_t ld, int p, const double *__restrict__ y,
                                             const double *__restrict__ sums, int ntc, int TB /* first tile column of this unit */, int64_t row_begin, int steps,
                                             double *__restrict__ tdst, double *__restrict__ vdst, double *lds)
{
    constexpr int NF = 2 * B, NT = wd_ntiles<B>(W), SLOT_B = (4 * B + 1) * 1024;
    constexpr int NDMA = wd_ndma<B>(W), DPW = NDMA + (W == 0 ? 1 : 0), NFETCH = NF + 1, NMFMA = 2 * NT, NACT = 1 + NFETCH + DPW, NSLOT = WD_NSLOT;
    static_assert((NSLOT - 2) * DPW <= 63, "vmcnt field is 6 bits");
    static_assert(NACT <= NMFMA, "more hook actions than MFMAs");
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const gptr_t xg = (gptr_t)x, yg = (gptr_t)y;
    const double inv_cnt = XF ? 1.0 / sums[p + 1] : 0.0;
    const double cy = XF ? sums[p] * inv_cnt : 0.0;
    auto tile_col = [&](int T) { const int col = 16 * (TB + T) + i; return col < p ? col : p - 1; };      // T: tile column within the unit
    // ---- DMA duty: tile columns W and W + 8 (scalar base per fragment + 32-bit lane offset), wave 0 also y
    gptr_t dbase[NDMA];
    unsigned doff[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
        int t0 = 16 * (TB + W + 8 * k);
        if (t0 > p - 1) t0 = p - 1;
        dbase[k] = xg + (size_t)t0 * ld + row_begin;
        doff[k] = (unsigned)(((int64_t)(tile_col(W + 8 * k) - t0) * ld + 2 * q) * 8);
    }
    gptr_t ybase = yg + row_begin;
    const unsigned yoff = (unsigned)(2 * q * 8);
    // the shift table behind the ring: entry [tile column][column within the tile]; this lane reads [.][i]
    double *cst = lds + (size_t)WD_NSLOT * SLOT_B / 8;
    if (XF) { if (tid < 64 * B) cst[tid] = sums[(16 * TB + tid < p) ? 16 * TB + tid : p - 1] * inv_cnt; __syncthreads(); }      // (a chunk of fewer than 8 rows meets no other barrier before it reads the table)
    const double *cs = cst + i;
    double sx[2] = {0.0, 0.0}, sxy[2] = {0.0, 0.0}, sy = 0.0, syy = 0.0;
    v4d acc16 = {0.0, 0.0, 0.0, 0.0};
    static_for<(NT < 16 ? NT : 16)>([&](auto T_) { AccTile<decltype(T_)::value>::zero(); });
    asm volatile("s_nop 7" ::: "memory");

    const int64_t rows_chunk = (int64_t)steps * 64;
    int64_t rows = n - row_begin; if (rows > rows_chunk) rows = rows_chunk; if (rows < 0) rows = 0;
    const int ns = (int)(rows / 8);
    const unsigned ring = (unsigned)(size_t)lds;
    const v2d *rd = reinterpret_cast<const v2d *>(lds) + lane;
    auto issue1 = [&](int slot, auto K_) {
        constexpr int k = decltype(K_)::value;
        const unsigned dst = ring + (unsigned)slot * SLOT_B;
        if constexpr (k < NDMA) { set_m0(dst + (unsigned)(W + 8 * k) * 1024); glds_s<0>(doff[k], dbase[k]); dbase[k] += 8; }
        else { set_m0(dst + (unsigned)(4 * B) * 1024u); glds_s<0>(yoff, ybase); ybase += 8; }
    };
    auto issue = [&](int slot) { static_for<DPW>([&](auto K_) { issue1(slot, K_); }); };
    Slab<8> sa, sb;
    sa.y = v2d{0.0, 0.0}; sb.y = sa.y;
    auto fetch1 = [&](Slab<8> &s, int slot, auto J_) {
        constexpr int j = decltype(J_)::value;
        const v2d *b = rd + (slot * SLOT_B) / 16;
        if constexpr (j < NF) s.v[j] = b[wd_frag<B>(W, j) * 64];
        else s.y = b[4 * B * 64];
    };
    auto fetch = [&](Slab<8> &s, int slot) { static_for<NFETCH>([&](auto J_) { fetch1(s, slot, J_); }); };
    auto next = [](int v) { return v + 1 == NSLOT ? 0 : v + 1; };
    const int npre = ns < NSLOT - 2 ? ns : NSLOT - 2;
    for (int j = 0; j < npre; ++j) issue(j);
    int islot = npre % NSLOT, rslot = 0, issued = npre;
    if (ns > 0) {
        if (npre == NSLOT - 2) wait_vm<(NSLOT - 3) * DPW>(); else wait_vm<0>();
        __syncthreads();
        fetch(sa, 0);
        rslot = 1;
    }
    int k = 0;
    auto steady = [&](Slab<8> &use, Slab<8> &nxt) {
        const int rs = rslot, is = islot;
        wd_consume<B, W, XF, false>(use, acc16, cs, cy, sx, sxy, sy, syy, 0, n, [&](auto M_) {
            constexpr int m = decltype(M_)::value;
            if constexpr (m == 0) { wait_vm<(NSLOT - 4) * DPW>(); __syncthreads(); }
            else if constexpr (m <= NFETCH) fetch1(nxt, rs, std::integral_constant<int, m - 1>{});
            else if constexpr (m <= NFETCH + DPW) issue1(is, std::integral_constant<int, m - NFETCH - 1>{});
        });
        rslot = next(rslot); islot = next(islot);
    };
    while (issued + 2 <= ns) {
        steady(sa, sb);
        steady(sb, sa);
        issued += 2; k += 2;
    }
    auto step = [&](Slab<8> &use, Slab<8> &nxt) {
        if (k + 1 < ns) {
            if (issued - (k + 2) >= NSLOT - 4) wait_vm<(NSLOT - 4) * DPW>(); else wait_vm<0>();
        }
        __syncthreads();
        if (k + 1 < ns) { fetch(nxt, rslot); rslot = next(rslot); }
        if (issued < ns) { issue(islot); islot = next(islot); ++issued; }
        wd_consume<B, W, XF, false>(use, acc16, cs, cy, sx, sxy, sy, syy, 0, n);
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
        for (int f = 0; f < NF; ++f) {
            const gptr_t pf = xg + (size_t)tile_col(wd_frag<B>(W, f)) * ld;
            t.v[f].x = pf[r0]; t.v[f].y = pf[r1];
        }
        t.y.x = yg[r0]; t.y.y = yg[r1];
        wd_consume<B, W, XF, true>(t, acc16, cs, cy, sx, sxy, sy, syy, r, n);
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // ---- results: tiles are wave-private -> straight to the partial buffer; the vector sums of this wave's two tile columns
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        sx[k2] += shfl_xor_d(sx[k2], 16);  sx[k2] += shfl_xor_d(sx[k2], 32);
        sxy[k2] += shfl_xor_d(sxy[k2], 16); sxy[k2] += shfl_xor_d(sxy[k2], 32);
    }
    sy += shfl_xor_d(sy, 16);   sy += shfl_xor_d(sy, 32);
    syy += shfl_xor_d(syy, 16); syy += shfl_xor_d(syy, 32);
    if (q == 0) {
        static_for<2>([&](auto K_) {
            constexpr int k2 = decltype(K_)::value;
            if constexpr (wd_sum_reg<B>(W, k2) >= 0) {
                const int T = TB + wd_frag<B>(W, wd_sum_reg<B>(W, k2));
                if (T < ntc) { vdst[16 * T + i] = sx[k2]; vdst[16 * ntc + 16 * T + i] = sxy[k2]; }
            }
        });
        if (W == 0 && i == 0 && TB == 0) {
            vdst[32 * ntc] = sy; vdst[32 * ntc + 1] = syy; vdst[32 * ntc + 2] = (double)rows; vdst[32 * ntc + 3] = 0.0;
        }
    }
    static_for<NT>([&](auto T_) {
        constexpr int t = decltype(T_)::value;
        constexpr WdTile tl = wd_tile<B>(W, t);
        static_assert(wd_frag<B>(W, tl.a) >= wd_frag<B>(W, tl.b), "a tile of the lower triangle");
        const int gi = TB + wd_frag<B>(W, tl.a), gj = TB + wd_frag<B>(W, tl.b);
        if (gi < ntc && gj < ntc) {
            double *dst = tdst + (size_t)(gi * (gi + 1) / 2 + gj) * 256;
            if constexpr (t < 16) static_for<4>([&](auto R_) { constexpr int r = decltype(R_)::value; dst[r * 64 + lane] = AccTile<t>::template read<r>(); });
            else { dst[lane] = acc16.x; dst[64 + lane] = acc16.y; dst[128 + lane] = acc16.z; dst[192 + lane] = acc16.w; }
        }
    });
}


// ------------------------------------------------------------------------------------------------
// Off-diagonal blocks between two units of sixteen tile columns (p > 256 with 16 k tile columns, round 5): 8 tile rows x 16 tile
// columns = 128 tiles, sixteen per wave -- at two waves per SIMD exactly a wave's accumulator file.  Wave W takes tile rows
// 4 (W >> 2) .. + 3 and tile columns 4 (W & 3) .. + 3 of the block; the 24 fragments of a slab (ring index 0-7: the tile rows, 8-23:
// the tile columns) are DMA'd once, three per wave, into a six-slot ring of 24 KiB slots.  24 fragment reads for 128 tiles where two
// 8 x 8 super-blocks of gram_sb_kernel read 32; no vector sums (the units' diagonal workgroups carry them).
// ------------------------------------------------------------------------------------------------
constexpr int OD_NSLOT = 6, OD_SLOT_B = 24 * 1024;
constexpr int od_frag(int W, int f) { return f < 4 ? 4 * (W >> 2) + f : 8 + 4 * (W & 3) + (f - 4); }      // ring index of register f

template <int W, bool XF, bool MASKED, typename Hook = NoHook>
__device__ __forceinline__ void od_consume(Slab<8> &s, const double *cs, int64_t r, int64_t n, Hook &&hook = NoHook())
{
    if (XF || MASKED) {
        double m0 = 1.0, m1 = 1.0;
        if (MASKED) { m0 = (r < n) ? 1.0 : 0.0; m1 = (r + 1 < n) ? 1.0 : 0.0; }
        static_for<8>([&](auto F_) {
            constexpr int f = decltype(F_)::value;
            if (XF) { const double c = cs[od_frag(W, f) * 16]; s.v[f].x -= c; s.v[f].y -= c; }
            if (MASKED) { s.v[f].x *= m0; s.v[f].y *= m1; }
        });
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 3" ::: "memory");                      // VALU write -> MFMA read
    }
    static_for<2>([&](auto E) {
        constexpr int e = decltype(E)::value;
        static_for<16>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            AccTile<t>::mfma(s.v[t / 4][e], s.v[4 + t % 4][e]);
            hook(std::integral_constant<int, e * 16 + t>{});
        });
    });
    __builtin_amdgcn_sched_barrier(0);
}

template <int W, bool XF>
__device__ __forceinline__ void gram_od_body(const double *__restrict__ x, int64_t n, int64_t ld, int p, const double *__restrict__ sums, int ntc,
                                             int TI /* first of the 8 tile rows */, int TJ /* first of the 16 tile columns */, int64_t row_begin,
                                             int steps, double *__restrict__ tdst, double *lds)
{
    constexpr int NDMA = 3, DPW = 3, NFETCH = 8, NMFMA = 32, NSLOT = OD_NSLOT, SLOT_B = OD_SLOT_B;
    static_assert((NSLOT - 2) * DPW <= 63 && 1 + NFETCH + DPW <= NMFMA, "hooks");
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const gptr_t xg = (gptr_t)x;
    const double inv_cnt = XF ? 1.0 / sums[p + 1] : 0.0;
    auto ring_tile = [&](int rf) { return rf < 8 ? TI + rf : TJ + (rf - 8); };                      // tile column of ring fragment rf
    auto ring_col = [&](int rf) { const int col = 16 * ring_tile(rf) + i; return col < p ? col : p - 1; };
    gptr_t dbase[NDMA];
    unsigned doff[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
        int t0 = 16 * ring_tile(W + 8 * k);
        if (t0 > p - 1) t0 = p - 1;
        dbase[k] = xg + (size_t)t0 * ld + row_begin;
        doff[k] = (unsigned)(((int64_t)(ring_col(W + 8 * k) - t0) * ld + 2 * q) * 8);
    }
    double *cst = lds + (size_t)NSLOT * SLOT_B / 8;                 // shift table [24 ring fragments][16]
    if (XF) {
        if (tid < 384) { int col = 16 * ring_tile(tid >> 4) + (tid & 15); if (col > p - 1) col = p - 1; cst[tid] = sums[col] * inv_cnt; }
        __syncthreads();
    }
    const double *cs = cst + i;
    static_for<16>([&](auto T_) { AccTile<decltype(T_)::value>::zero(); });
    asm volatile("s_nop 7" ::: "memory");

    const int64_t rows_chunk = (int64_t)steps * 64;
    int64_t rows = n - row_begin; if (rows > rows_chunk) rows = rows_chunk; if (rows < 0) rows = 0;
    const int ns = (int)(rows / 8);
    const unsigned ring = (unsigned)(size_t)lds;
    const v2d *rd = reinterpret_cast<const v2d *>(lds) + lane;
    auto issue1 = [&](int slot, auto K_) {
        constexpr int k = decltype(K_)::value;
        set_m0(ring + (unsigned)slot * SLOT_B + (unsigned)(W + 8 * k) * 1024); glds_s<0>(doff[k], dbase[k]); dbase[k] += 8;
    };
    auto issue = [&](int slot) { static_for<DPW>([&](auto K_) { issue1(slot, K_); }); };
    Slab<8> sa, sb;
    auto fetch1 = [&](Slab<8> &s, int slot, auto J_) {
        constexpr int j = decltype(J_)::value;
        s.v[j] = (rd + (slot * SLOT_B) / 16)[od_frag(W, j) * 64];
    };
    auto fetch = [&](Slab<8> &s, int slot) { static_for<NFETCH>([&](auto J_) { fetch1(s, slot, J_); }); };
    auto next = [](int v) { return v + 1 == NSLOT ? 0 : v + 1; };
    const int npre = ns < NSLOT - 2 ? ns : NSLOT - 2;
    for (int j = 0; j < npre; ++j) issue(j);
    int islot = npre % NSLOT, rslot = 0, issued = npre;
    if (ns > 0) {
        if (npre == NSLOT - 2) wait_vm<(NSLOT - 3) * DPW>(); else wait_vm<0>();
        __syncthreads();
        fetch(sa, 0);
        rslot = 1;
    }
    int k = 0;
    auto steady = [&](Slab<8> &use, Slab<8> &nxt) {
        const int rs = rslot, is = islot;
        od_consume<W, XF, false>(use, cs, 0, n, [&](auto M_) {
            constexpr int m = decltype(M_)::value;
            if constexpr (m == 0) { wait_vm<(NSLOT - 4) * DPW>(); __syncthreads(); }
            else if constexpr (m <= NFETCH) fetch1(nxt, rs, std::integral_constant<int, m - 1>{});
            else if constexpr (m <= NFETCH + DPW) issue1(is, std::integral_constant<int, m - NFETCH - 1>{});
        });
        rslot = next(rslot); islot = next(islot);
    };
    while (issued + 2 <= ns) {
        steady(sa, sb);
        steady(sb, sa);
        issued += 2; k += 2;
    }
    auto step = [&](Slab<8> &use, Slab<8> &nxt) {
        if (k + 1 < ns) {
            if (issued - (k + 2) >= NSLOT - 4) wait_vm<(NSLOT - 4) * DPW>(); else wait_vm<0>();
        }
        __syncthreads();
        if (k + 1 < ns) { fetch(nxt, rslot); rslot = next(rslot); }
        if (issued < ns) { issue(islot); islot = next(islot); ++issued; }
        od_consume<W, XF, false>(use, cs, 0, n);
        ++k;
    };
    while (k < ns) {
        step(sa, sb);
        if (k < ns) step(sb, sa);
    }
    wait_vm<0>();
    if (rows - 8 * (int64_t)ns > 0) {                               // ragged tail: masked loads straight from global memory
        const int64_t r = row_begin + 8 * (int64_t)ns + 2 * q;
        const int64_t r0 = r < n ? r : n - 1, r1 = r + 1 < n ? r + 1 : n - 1;
        Slab<8> t;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const gptr_t pf = xg + (size_t)ring_col(od_frag(W, f)) * ld;
            t.v[f].x = pf[r0]; t.v[f].y = pf[r1];
        }
        od_consume<W, XF, true>(t, cs, r, n);
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    static_for<16>([&](auto T_) {
        constexpr int t = decltype(T_)::value;
        const int gi = TI + 4 * (W >> 2) + t / 4, gj = TJ + 4 * (W & 3) + t % 4;
        if (gi < ntc && gj < ntc) {
            double *dst = tdst + (size_t)(gi * (gi + 1) / 2 + gj) * 256;
            static_for<4>([&](auto R_) { constexpr int r = decltype(R_)::value; dst[r * 64 + lane] = AccTile<t>::template read<r>(); });
        }
    });
}

}  // namespace

// One launch: per row chunk nu diagonal units of sixteen (B = 3: twelve) tile columns and, between every two units, two off-diagonal
// blocks of 8 x 16 tiles.  Order: all diagonal units of all chunks (68 MFMAs per SIMD and slab), then the off-diagonal blocks (64);
// the workgroups of one row chunk share blockIdx % 8 (one XCD).  nu = 1 (p <= 256): blockIdx.x is the chunk.
template <int B>
__device__ __forceinline__ void gram_wd_kernel_body(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ sums,
                                                    double *__restrict__ tpart, double *__restrict__ vpart, const GramDims &a, int nu)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int L = blockIdx.x, xcd = L & 7, ng = a.nchunk / 8;
    int s = L >> 3, chunk, unit = 0, od = -1;
    if (s < nu * ng) { unit = s % nu; chunk = (s / nu) * 8 + xcd; }
    else { s -= nu * ng; const int nod = nu * (nu - 1); od = s % nod; chunk = (s / nod) * 8 + xcd; }
    const int64_t row_begin = (int64_t)chunk * a.steps * 64;
    double *tdst = tpart + (size_t)chunk * a.ntile * 256;
    double *vdst = vpart + (size_t)chunk * (32 * a.ntc + 4);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool xf = shift_needed_wave(sums, a.p);
    if (od < 0) {
        const int TB = 4 * B * unit;
#define OEM_WD(W, XF) gram_wd_body<B, W, XF>(x, a.n, a.ld, a.p, y, sums, a.ntc, TB, row_begin, a.steps, tdst, vdst, lds)
#define OEM_WD_ALL(XF)                                                                                                   \
    do {                                                                                                                 \
        if (w == 0) OEM_WD(0, XF); else if (w == 1) OEM_WD(1, XF); else if (w == 2) OEM_WD(2, XF); else if (w == 3) OEM_WD(3, XF);   \
        else if (w == 4) OEM_WD(4, XF); else if (w == 5) OEM_WD(5, XF); else if (w == 6) OEM_WD(6, XF); else OEM_WD(7, XF);          \
    } while (0)
        if (xf) OEM_WD_ALL(true); else OEM_WD_ALL(false);
#undef OEM_WD_ALL
#undef OEM_WD
    } else if constexpr (B == 4) {
        // pair (I > J) of units and which half of unit I's tile rows: od = 2 (I (I - 1) / 2 + J) + half
        const int pi = od >> 1, half = od & 1;
        int I = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)pi)) * 0.5f);
        while (I * (I - 1) / 2 > pi) --I;
        while ((I + 1) * I / 2 <= pi) ++I;
        const int J = pi - I * (I - 1) / 2, TI = 16 * I + 8 * half, TJ = 16 * J;
#define OEM_OD(W, XF) gram_od_body<W, XF>(x, a.n, a.ld, a.p, sums, a.ntc, TI, TJ, row_begin, a.steps, tdst, lds)
#define OEM_OD_ALL(XF)                                                                                                   \
    do {                                                                                                                 \
        if (w == 0) OEM_OD(0, XF); else if (w == 1) OEM_OD(1, XF); else if (w == 2) OEM_OD(2, XF); else if (w == 3) OEM_OD(3, XF);   \
        else if (w == 4) OEM_OD(4, XF); else if (w == 5) OEM_OD(5, XF); else if (w == 6) OEM_OD(6, XF); else OEM_OD(7, XF);          \
    } while (0)
        if (xf) OEM_OD_ALL(true); else OEM_OD_ALL(false);
#undef OEM_OD_ALL
#undef OEM_OD
    }
}

__global__ __launch_bounds__(512) void gram_wd_kernel(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ sums,
                                                       double *__restrict__ tpart, double *__restrict__ vpart, GramDims a, int nu)
{
    gram_wd_kernel_body<4>(x, y, sums, tpart, vpart, a, nu);         // units of 15-16 tile columns
}
__global__ __launch_bounds__(512) void gram_wd3_kernel(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ sums,
                                                        double *__restrict__ tpart, double *__restrict__ vpart, GramDims a)
{
    gram_wd_kernel_body<3>(x, y, sums, tpart, vpart, a, 1);          // 11-12 tile columns
}

int launch_gram_wd(hipStream_t s, const GramPlan &pl, const double *x, const double *y, const double *sums, double *tpart, double *vpart, const GramDims &a)
{
    const int B = pl.wd, nu = pl.wd_units;
    size_t shb = (size_t)WD_NSLOT * (4 * B + 1) * 1024 + (size_t)64 * B * sizeof(double);      // the ring + the shift table
    if (B == 4) {
        const size_t sho = (size_t)OD_NSLOT * OD_SLOT_B + 384 * sizeof(double);
        if (nu > 1 && sho > shb) shb = sho;
        if (lds_limit_once(reinterpret_cast<const void *>(&gram_wd_kernel), shb)) return OEMGPU_ERR_HIP;
        hipLaunchKernelGGL(gram_wd_kernel, dim3(pl.nchunk * nu * nu), dim3(512), shb, s, x, y, sums, tpart, vpart, a, nu);
    } else if (B == 3 && nu == 1) {
        if (lds_limit_once(reinterpret_cast<const void *>(&gram_wd3_kernel), shb)) return OEMGPU_ERR_HIP;
        hipLaunchKernelGGL(gram_wd3_kernel, dim3(pl.nchunk), dim3(512), shb, s, x, y, sums, tpart, vpart, a);
    } else { set_error("internal: gram_wd block size %d, %d units", B, nu); return OEMGPU_ERR_INTERNAL; }
    OEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace oemgpu
