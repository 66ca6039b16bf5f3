// path_symcoop.hip -- eigenvalue + penalty x lambda path for 1024 < q <= 4096 (config 4: oem.xtx at p = 4096): ONE persistent launch
// of cooperating workgroups with the LOWER TRIANGLE of XX resident in the register files of the chip.
//
// Replaces, for element-wise penalties at these sizes, the launch-per-iteration engines of path_large.hip, which stream the matrix
// (oem_symfused_kernel: its lower triangle, 69 MB at q = 4096) from the Infinity Cache once per OEM iteration: 15.8 us per iteration,
// 3,504 of them in config 4.  The lower triangle of a 4096 x 4096 FP64 matrix is 67 MB; the register files of 176 CUs hold 88 MB.
// So nothing is streamed at all.  Same arithmetic as the other engines (ref src/oem_xtx.h:378-381, src/oem_dense.h:501-524, 76-149,
// src/oem_base.h:90-110, src/utils.cpp:537-549, src/oem_dense.cpp:175-297):
//
//   tiles      XX in 64 x 64 tiles, only I >= J.  A wave keeps NT of them (1, 2 or 3: 128 NT registers per lane, the compiler's
//              VGPR + AGPR file of a one-wave-per-SIMD kernel), a workgroup 4 NT -- a patch of a few tile rows x tile columns.
//              Lane (rl, cl) of a tile holds the 8 x 8 sub-block of rows 16 (i >> 1) + 2 rl + (i & 1), columns likewise with cl.
//   products   a tile feeds BOTH products it holds: g_I += T beta_J ("direct") and g_J += T' beta_I ("transposed"; not for the
//              diagonal tiles, which are stored whole): 128 plain FMAs per lane and tile against 8 + 8 vector entries read from
//              LDS.  The eight column lanes of a row (and the eight row lanes of a column) meet in a transposed butterfly:
//              v_permlane32_swap / v_permlane16_swap for the first halving (both directions in one instruction, no select),
//              two DPP stages after it; the lane bits are dealt so that every partner holds the same rows (or columns).
//   exchange   what crosses workgroups is an ALL-REDUCE of the q-vector in two SPARSE exchanges through the memory side, as
//              data-tagged 16-byte pairs {lo, tag, hi, tag} (path_coop.hip's recipe: the data is the flag, sc1 stores and polls,
//              one poll sweep in flight, two buffers by parity):
//                1. a workgroup adds its waves' partial vectors per 64-block in LDS (fixed order) and sends each block to the
//                   owners of its coordinates: block B receives one partial from every workgroup whose patch touches tile row
//                   or tile column B (16-27 of them, not G); the owner of a slice of q / G coordinates adds them in sender
//                   order (interleaved chains combined by DPP: bitwise reproducible), applies the operator -- it is
//                   coordinate-local -- and the stop rule to ITS coordinates;
//                2. the owners publish their coordinates of the new beta and every workgroup gathers the blocks its patch
//                   touches (448 values for a 3 x 4 patch, not q).
//   stop rule  the "still moving" bit of a coordinate rides in the tag of its pair in exchange 2; a workgroup ORs what it gathers
//              and sends that bit in the tags of the NEXT exchange 1.  For any two blocks some patch touches both (the tile
//              (max, min) exists), so after that exchange every owner holds the OR over all coordinates: the decision about
//              iteration t is taken by everybody, identically, at iteration t + 1 -- exactly the "replicated one launch later"
//              bookkeeping of oem_fused_kernel / oem_symfused_kernel (u does not depend on lambda, so the iteration that detects
//              convergence at lambda_i already is the first one of lambda_{i+1}).  No flag or scalar of its own crosses.
//   Lanczos    on the same registers with the same two exchanges; alpha = v'XXv as the sum of the workgroups' v . partial (one
//              more pair per workgroup next to exchange 1) and ||w'||^2 as the owners' parts next to exchange 2, both added in
//              workgroup order by everybody: two hops per step.  Top Ritz value by the Sturm multisection of path_dev.hpp.
// Every spin is bounded; a timeout poisons d_out[6] and the host makes the call again on the launch-per-iteration engine.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "penalty_ops.hpp"
#include "path_dev.hpp"

#ifndef OEM_SX_VOTE_BARRIER
#define OEM_SX_VOTE_BARRIER 0     // 1 (A/B builds): the workgroup vote + barrier behind gather 1 also in the path phase of the element-wise form (rounds 4-5)
#endif
#ifndef OEM_XCHG_SLEEP
#define OEM_XCHG_SLEEP 12         // s_sleep units (64 cycles) before the first poll sweep of a gather (tools/xchg_sleep_ab.sh)
#endif
namespace oemgpu {

namespace {

constexpr int SNTH = 256;         // threads per workgroup: one wave per SIMD (512 registers per lane)
constexpr int SNB = 16;           // 64-blocks of the vector a workgroup's patch may touch
constexpr int SE1 = 4;            // senders per owned coordinate / 8, at most
constexpr int SE2 = (SNB * 64) / SNTH;
constexpr int SCML = 512;         // Lanczos steps kept
constexpr int SSL = SNTH / 8;     // coordinates a workgroup may own
// per-workgroup plan record (ints)
enum { SW_NB = 0, SW_C0 = 1, SW_NSL = 2, SW_BLK = 4, SW_RANK = SW_BLK + SNB, SW_TILE = SW_RANK + SNB, SW_CST = SW_TILE + 12,
       SW_TPOS = SW_CST + SNB + 1, SW_INTS = SW_TPOS + 12 + 19 };          // TPOS: row of Wp of a tile's direct product | of its transposed one << 8
static_assert(SW_INTS == 96, "plan record");
constexpr int SPLAN_HEAD = 80;    // blkbase[T + 1], T <= 64
// rows (of the eight of a lane's sub-block) of a wave's THIRD tile that stay in VGPRs; the others live in LDS -- two, or one in the
// general form (GEN), whose group operators, Nesterov step and loss cost the compiler 50 more registers than the element-wise form
__host__ __device__ constexpr int sx_svr(bool gen) { return gen ? 1 : 2; }
__host__ __device__ constexpr int sx_slt(bool gen) { return (8 - sx_svr(gen)) * 4 * 64 * 2; }      // doubles of LDS per wave for them

#ifdef OEM_PATH_DIAG
__device__ unsigned long long g_diag_symcoop[16];
__device__ unsigned long long g_diag_symcoop_waves[4 * 16];      // the path phase of every wave of workgroup 0: acc[0..10] (tools/symcoop_diag.py)
#define SX_STAMP(slot)                                                                     \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        unsigned long long t__;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        X.acc[slot] += t__ - X.last;                                                       \
        X.last = t__;                                                                      \
    } while (0)
#else
#define SX_STAMP(slot) do { } while (0)
#endif

typedef unsigned sx_v4u __attribute__((ext_vector_type(4)));
struct SymX {
#ifdef OEM_PATH_DIAG
    unsigned long long acc[16], last;
#endif
    __amdgpu_buffer_rsrc_t rs;    // exchange 1 at 0: [2 parities][nsum * 64] pairs; o2: exchange 2, [2][64 T] pairs; o3 / o4: the workgroups'
    int o2, o3, o4;               // parts of alpha / of ||w'||^2 (Lanczos), [2][G] pairs each (byte offsets)
    int s1, s2;                   // bytes per parity of exchange 1 / 2
    unsigned epoch;               // all-reduce counter, never 0; identical in every workgroup
    int wg, G;
    bool failed;                  // an exchange timed out, or the host's abort word was seen: nobody waits any more
    // PathArgs::abort_word without a register of its own (path_rowcoop_kernel has none to spare: one more live value and hipcc parks
    // its own in the accumulator file the inline asm owns -- oem_amd/build.py audits that): the pointer sits in an LDS slot and is read
    // where it is needed, and "seen" is an LDS word (0 / 2) that is read next to the votes behind their barrier
    const int *const *abortp;
    int *aflag;
};
// the host's abort word, looked at by the whole wave: seen => nobody waits any more, and the workgroup's next vote says "leave"
__device__ __forceinline__ bool sx_abort_seen(SymX &X)
{
    if (!path_abort_asked(*X.abortp)) return false;
    X.failed = true;
    *X.aflag = 2;
    return true;
}

__device__ __forceinline__ void sx_publish(__amdgpu_buffer_rsrc_t rs, int off, double val, unsigned tag)
{
    sx_v4u v;
    v.x = (unsigned)__double2loint(val); v.y = tag; v.z = (unsigned)__double2hiint(val); v.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);       // aux 16: sc1 (device scope)
}

// E pairs per thread at byte offsets off[k] (need bit k), polled until both tags carry this epoch; the low tag bits are OR-ed into
// `flags`.  One 16-byte load per pair, one sweep in flight, a pair that has arrived is not asked for again (tools/xchg_probe.hip).
// (DS >= 0, diagnostic builds only: acc[DS] += cycles from the first sweep until the wave's FIRST pair is there, acc[DS + 1] += from then until
// its last one -- how far apart a wave's pairs arrive is what "multiply a block as soon as ITS coefficients are there" could overlap)
template <int E, int DS = -1>
__device__ __forceinline__ void sx_gather(const int (&off)[E], unsigned need, double (&out)[E], int &flags, SymX &X)
{
    sx_v4u pv[E];
    unsigned miss = need;
#pragma unroll
    for (int k = 0; k < E; ++k) pv[k] = sx_v4u{0u, 0u, 0u, 0u};
    if (E > 1 && OEM_XCHG_SLEEP > 0) __builtin_amdgcn_s_sleep(OEM_XCHG_SLEEP);      // (path_wcoop.hip: wc_gather has the measurement; E = 1: scalars that landed a hop ago)
#ifdef OEM_PATH_DIAG
    unsigned long long dg0 = 0, dg1 = 0;
    if (DS >= 0) dg0 = __builtin_amdgcn_s_memtime();
#endif
    // ONE counter in the sweep loop: it runs out once per 1,024 sweeps (~1 ms), and only then are the abort word and the timeout looked at
    // (~1 s = 1,000 such rounds: a partner is gone; after one timeout -- or the abort word -- nobody waits again: one sweep each)
    unsigned left = X.failed ? 1u : PATH_ABORT_SPINS, rounds = 0u;
    bool ok = true;
    while (__any(miss != 0u)) {
#pragma unroll
        for (int k = 0; k < E; ++k)
            if ((miss >> k) & 1u) pv[k] = __builtin_amdgcn_raw_buffer_load_b128(X.rs, off[k], 0, 16);
#pragma unroll
        for (int k = 0; k < E; ++k)
            if (((miss >> k) & 1u) && (pv[k].y >> 1) == X.epoch && (pv[k].w >> 1) == X.epoch) miss &= ~(1u << k);
#ifdef OEM_PATH_DIAG
        if (DS >= 0 && dg1 == 0 && __any(miss != need)) dg1 = __builtin_amdgcn_s_memtime();
#endif
        if (--left == 0u && __any(miss != 0u)) {
            if (X.failed || ++rounds >= PATH_TIMEOUT_ROUNDS) { ok = false; break; }
            if (sx_abort_seen(X)) break;
            left = PATH_ABORT_SPINS;
        }
    }
#ifdef OEM_PATH_DIAG
    if (DS >= 0) { const unsigned long long dg2 = __builtin_amdgcn_s_memtime(); if (dg1 == 0) dg1 = dg2; X.acc[DS >= 0 ? DS : 0] += dg1 - dg0; X.acc[DS >= 0 ? DS + 1 : 0] += dg2 - dg1; }
#endif
    if (!ok) X.failed = true;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const bool nd = ((need >> k) & 1u) != 0 && ((miss >> k) & 1u) == 0;
        out[k] = nd ? __hiloint2double((int)pv[k].z, (int)pv[k].x) : 0.0;
        if (nd) flags |= (int)(pv[k].y & 1u);
    }
}

__device__ __forceinline__ double sx_block_sum(double v, double *red, int &rpar, int w, int lane)
{
    const double s = wave_sum(v);
    double *r = red + 4 * rpar;
    if (lane == 0) r[w] = s;
    __syncthreads();
    const double t = (r[0] + r[1]) + (r[2] + r[3]);
    rpar ^= 1;                    // the next call writes the other half: no second barrier needed
    return t;
}
__device__ __forceinline__ void sx_vote(int *words, int w, int lane, int bit)
{
    const int wb = __ballot(bit != 0) != 0ull ? 1 : 0;
    if (lane == 0) words[w] = wb;
}

// both halves of a v_permlane{16,32}_swap of two doubles, added: lanes whose bit is 0 get a(own) + a(partner), the others b(partner) + b(own)
template <bool SW32> __device__ __forceinline__ double sx_swap_add(double a, double b)
{
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if constexpr (SW32) {
        auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    } else {
        auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
}
template <int CTRL> __device__ __forceinline__ double sx_dpp_stage(double lo, double hi, bool sel)
{
    const double keep = sel ? hi : lo, send = sel ? lo : hi;
    return keep + dpp_mov<CTRL, 0xf>(send, 0.0);
}

// Where a wave keeps its tiles.  The register file belongs to the matrix, and hipcc's own allocation is not to be trusted with it
// (three tiles as plain arrays: 512 registers, 57 spilled into the loop): tiles 0 and 1 live in AGPRs a0..a255 that ONLY the inline
// asm below names -- the compiler's own values fit the 256 architectural VGPRs, so it never touches the accumulator file
// (oem_amd/build.py: audit_symcoop_isa proves that on the emitted ISA, as for the Gram kernels) -- and the third tile of a wave
// (q > 3456) a quarter in VGPRs (rows 0, 1 of the lane's 8 x 8 sub-block), the rest in LDS (24 16-byte reads per lane,
// [read][lane], a wave's read 1 KiB contiguous: 96 KB per workgroup).
// The products of ONE tile with the vector blocks at Bsh + oI (rows) and Bsh + oJ (columns), reduced over the lanes.
// DD: the direct product (T vec_J) -> Wd[64]; TT: the transposed one (T' vec_I) -> Wt[64].  One pass over the tile feeds both.
// ST: 0 / 1 = the tile in AGPRs a[128 ST ..]; 2 = rows 0..SVR-1 in vlo, the others in LDS at lt.
// Lane bits: rl = (b4, b3, b2), cl = (b5, b1 ^ b2, b0 ^ b2) -- the partners of the direct reduction (lane ^ 32, ^ 2, ^ 1) keep rl, those
// of the transposed one (lane ^ 16, ^ 8, ^ 7) keep cl.
template <bool DD, bool TT, int ST, int SVR>
__device__ __forceinline__ void sx_tile(const double (&vlo)[8 * SVR], const double *lt, const double *Bsh, int oI, int oJ, unsigned mI, unsigned mJ,
                                        double *Wd, double *Wt, int lane, int rlv, int clv)
{
    double bj[8], bi[8], ad[8], at[8];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if (DD) { const v2d cj = *reinterpret_cast<const v2d *>(Bsh + oJ + 16 * h + 2 * clv); bj[2 * h] = cj.x; bj[2 * h + 1] = cj.y; }
        if (TT) { const v2d ci = *reinterpret_cast<const v2d *>(Bsh + oI + 16 * h + 2 * rlv); bi[2 * h] = ci.x; bi[2 * h + 1] = ci.y; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { ad[k] = 0.0; at[k] = 0.0; }
    // rows i, columns 2 jj and 2 jj + 1 of the lane's sub-block
    auto cell = [&](auto I_, auto J_) {
        constexpr int i = decltype(I_)::value, jj = decltype(J_)::value;
        double x0, x1;
        if constexpr (ST < 2) { x0 = areg_rd<ST * 64 + i * 8 + 2 * jj>(); x1 = areg_rd<ST * 64 + i * 8 + 2 * jj + 1>(); }
        else if constexpr (i < SVR) { x0 = vlo[i * 8 + 2 * jj]; x1 = vlo[i * 8 + 2 * jj + 1]; }
        else { const v2d t = *reinterpret_cast<const v2d *>(lt + (((i - SVR) * 4 + jj) * 64 + lane) * 2); x0 = t.x; x1 = t.y; }
        if (DD) { ad[i] = fma(x0, bj[2 * jj], ad[i]); }
        if (TT) { at[2 * jj] = fma(x0, bi[i], at[2 * jj]); }
        if (DD) { ad[i] = fma(x1, bj[2 * jj + 1], ad[i]); }
        if (TT) { at[2 * jj + 1] = fma(x1, bi[i], at[2 * jj + 1]); }
    };
    if constexpr (DD && TT) {
        static_for_dev<8>([&](auto I_) { static_for_dev<4>([&](auto J_) { cell(I_, J_); }); });
    } else if constexpr (DD) {
        // only the 16-column groups of the vector block that hold a non-zero (a lasso iterate is sparse inside its blocks too)
        static_for_dev<4>([&](auto J_) {
            if ((mJ >> decltype(J_)::value) & 1u) static_for_dev<8>([&](auto I_) { cell(I_, J_); });
        });
    } else {
        static_for_dev<4>([&](auto H_) {                             // ... the 16-row groups
            constexpr int ii = decltype(H_)::value;
            if ((mI >> ii) & 1u) static_for_dev<4>([&](auto J_) {
                cell(std::integral_constant<int, 2 * ii>{}, J_); cell(std::integral_constant<int, 2 * ii + 1>{}, J_);
            });
        });
    }
    if (DD) {   // over cl -- lane ^ 32, lane ^ 2, lane ^ 1; this lane ends with row index i = clv
        const bool sc1 = ((clv >> 1) & 1) != 0, sc0 = (clv & 1) != 0;
        double n1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) n1[i] = sx_swap_add<true>(ad[i], ad[4 + i]);
        const double n20 = sx_dpp_stage<0x4E>(n1[0], n1[2], sc1), n21 = sx_dpp_stage<0x4E>(n1[1], n1[3], sc1);
        Wd[16 * (clv >> 1) + 2 * rlv + (clv & 1)] = sx_dpp_stage<0xB1>(n20, n21, sc0);
    }
    if (TT) {   // over rl -- lane ^ 16, lane ^ 8 (row_ror:8), lane ^ 7 (row_half_mirror); this lane ends with column index j = rlv
        const bool sb3 = (lane & 8) != 0, sb2 = (lane & 4) != 0;
        double m1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m1[j] = sx_swap_add<false>(at[j], at[4 + j]);
        const double m20 = sx_dpp_stage<0x128>(m1[0], m1[2], sb3), m21 = sx_dpp_stage<0x128>(m1[1], m1[3], sb3);
        Wt[16 * (rlv >> 1) + 2 * clv + (rlv & 1)] = sx_dpp_stage<0x141>(m20, m21, sb2);
    }
}

// element-wise operators (ref src/oem_dense.h:76-149), branch-free inside a kind.  The constants of the lambda in use live in LDS
// (the register file belongs to the matrix): kind, L, D, 1/D, gamma D, D - 1/gamma and its reciprocal, gamma - 1, gamma,
// (gamma - 1) D - 1 and its reciprocal, d, 1/d
enum { TH_L = 0, TH_D, TH_RD, TH_GAMMAD, TH_DMG, TH_RDMG, TH_GM1, TH_GAMMA, TH_DSC, TH_RDSC, TH_D0, TH_RD0, TH_L1, TH_N };
__device__ __forceinline__ void sx_thr_store(double *th, int *kind, const PenK &K, double d)
{
    *kind = K.kind;
    th[TH_L] = K.L; th[TH_D] = K.D; th[TH_RD] = 1.0 / K.D; th[TH_GAMMAD] = K.gamma * K.D;
    const double dmg = K.D - 1.0 / K.gamma, gm1 = K.gamma - 1.0, dsc = gm1 * K.D - 1.0;
    th[TH_DMG] = dmg; th[TH_RDMG] = 1.0 / dmg; th[TH_GM1] = gm1; th[TH_GAMMA] = K.gamma; th[TH_DSC] = dsc; th[TH_RDSC] = 1.0 / dsc;
    th[TH_D0] = d; th[TH_RD0] = 1.0 / d; th[TH_L1] = K.L1;
}
__device__ __forceinline__ double sx_op(double u, double pf, int kind, const double (&th)[TH_N])
{
    const double tp = pf * th[TH_L];
    if (kind == K_SOFT) return cdiv(shrink(u, tp), th[TH_D], th[TH_RD]);
    if (kind == K_MCP) {
        const bool big = fabs(u) > th[TH_GAMMAD] * tp;
        return cdiv(big ? u : shrink(u, tp), big ? th[TH_D] : th[TH_DMG], big ? th[TH_RD] : th[TH_RDMG]);
    }
    if (kind == K_SCAD) {
        const double au = fabs(u), D = th[TH_D];
        const bool big = au > th[TH_GAMMAD] * tp, mid = !big && au > (D + 1.0) * tp;
        const double num = big ? u : (mid ? shrink(th[TH_GM1] * u, th[TH_GAMMA] * tp) : shrink(u, tp));
        return cdiv(num, mid ? th[TH_DSC] : D, mid ? th[TH_RDSC] : th[TH_RD]);
    }
    return cdiv(u, th[TH_D0], th[TH_RD0]);
}

// wave-uniform doubles into SGPRs (the spilled ones cost a lane of a VGPR, not a VGPR pair: the register file belongs to the matrix)
__device__ __forceinline__ double sx_uni(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// GEN: what needs more than a coordinate of its own (template parameter: the element-wise form pays nothing for it) --
//   group operators (ref src/oem_dense.h:193-315): the owners' slices are cut at group boundaries by the host (every group a run of
//       neighbouring coordinates), an owner's u goes through 32 LDS words and every coordinate sums the squares of ITS group in
//       member order like the reference, forms the group's factor and its coefficient; a group of more than 32 members lies in several
//       owners' slices -- their parts of the sum cross in a third tagged exchange and are added in owner order (gsplit);
//   Nesterov's step (ref :633-651): its restart test is a sum over ALL coordinates -- the owners' parts ride next to exchange 2 as
//       Lanczos' norm parts do, added in workgroup order by everybody;
//   compute.loss (ref :759-770, Gram identity): when a lambda ends, g = XX beta of the finished iterate is in hand; the owners' parts of
//       beta'(g - 2 XY) go to workgroup 0 through the scalar area of exchange 1.
template <int NT, bool GEN>
__global__ __launch_bounds__(SNTH) void path_symcoop_kernel(PathArgs A, const int *__restrict__ plan, unsigned long long *xchg, int T, int nsum, int e1n, int gsplit)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int q = A.p, wg = blockIdx.x, G = gridDim.x;
    const unsigned long long t_cyc0 = __builtin_amdgcn_s_memtime(), t_rt0 = __builtin_amdgcn_s_memrealtime();
    double *Bsh = lds;                                   // the vector going into the product, by block slot [SNB][64]
    double *Wp = Bsh + SNB * 64;                         // the waves' reduced tile products [8 NT][64]
    double *Tal = Wp + (8 * NT + 1) * 64, *Tbe = Tal + SCML;   // (a row of zeros behind Wp;) Lanczos alpha, beta
    double *sturm = Tbe + SCML;                          // Sturm scratch 2 (SCML + 16)
    double *red = sturm + 2 * (SCML + 16);               // block reductions [2][4], theta slot [8], lmax words [12..16)
    double *thr = red + 16;                              // operator constants of the lambda in use [TH_N <= 16]
    int *votes = reinterpret_cast<int *>(thr + 16);      // [8] votes, [8] kind
    int *nzs = votes + 16;                               // [SNB] which 16-coordinate groups of this slot of Bsh hold a non-zero (4 bits)
    int *wv = nzs + SNB;                                 // (32 spare words; [0..1]: PathArgs::abort_word, SymX::abortp, [2]: SymX::aflag)
    double *uo = reinterpret_cast<double *>(wv + 32);     // GEN: u of this owner's coordinates [SSL]
    int *rec = wv + 32 + 2 * SSL;                        // this workgroup's plan record [SW_INTS], then P1[SNB] (pairs index of exchange 1 per slot)
    int *P1 = rec + SW_INTS;
    constexpr int SVR = sx_svr(GEN), SLT = sx_slt(GEN);
    int *gfr = P1 + SNB + 8;                             // GEN, split groups: per owned coordinate (first owner of its group) * 2 + slot, number of owners [2 SSL]
    double *gwl = reinterpret_cast<double *>(gfr + 2 * SSL);      // GEN: the weight of an owned coordinate's group [SSL] (read where it is used: two registers the NT = 3 form does not have)
    double *Lt = gwl + SSL + w * SLT;                    // NT == 3: rows SVR..7 of every lane's part of this wave's third tile
    const bool writer = wg == 0;

    const int *__restrict__ blkbase = plan;
    for (int k = tid; k < SW_INTS; k += SNTH) rec[k] = plan[SPLAN_HEAD + wg * SW_INTS + k];
    if (tid == 0) { *reinterpret_cast<const int **>(wv) = A.abort_word; wv[2] = 0; }
    for (int k = tid; k < SNB * 64; k += SNTH) Bsh[k] = 0.0;
    if (tid < 64) Wp[8 * NT * 64 + tid] = 0.0;
    __syncthreads();
    const int nb = rec[SW_NB], c0 = rec[SW_C0], nsl = rec[SW_NSL];
    if (tid < SNB) P1[tid] = tid < nb ? (blkbase[rec[SW_BLK + tid]] + rec[SW_RANK + tid]) * 64 : 0;

    // ---- the reduced tile products that make up block slot s sit in rows cst[s] .. cst[s + 1] - 1 of Wp (the host deals the rows
    // slot by slot); this wave adds up slots w, w + 4, w + 8, w + 12
    int erow[4], ecnt[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int sl = w + 4 * m;
        erow[m] = __builtin_amdgcn_readfirstlane(sl < nb ? rec[SW_CST + sl] : 0);
        ecnt[m] = __builtin_amdgcn_readfirstlane(sl < nb ? rec[SW_CST + sl + 1] - rec[SW_CST + sl] : 0);
    }
    const int nslw = __builtin_amdgcn_readfirstlane(nb > w ? (nb - w + 3) / 4 : 0);       // block slots of this wave
    // ---- this wave's tiles: lane (rl, cl) holds the 8 x 8 sub-block of rows 16 (i >> 1) + 2 rl + (i & 1), columns alike with cl
    const int rlv = (lane >> 2) & 7, clv = ((lane >> 5) << 2) | ((((lane >> 1) ^ (lane >> 2)) & 1) << 1) | ((lane ^ (lane >> 2)) & 1);
    const double *xxp = A.xx;
    double vlo[8 * SVR];                                 // NT == 3: rows 0..SVR-1 of this lane's part of the wave's third tile
    int tI[NT], tJ[NT], tflag[NT], tpos[NT];             // block slots of the tile's rows / columns; 0: no tile, 1: off-diagonal, 2: diagonal; rows of Wp
    if constexpr (NT == 1) asm volatile("" ::: "a127"); else asm volatile("" ::: "a255");      // the accumulator file is in use (by the asm alone)
#pragma unroll
    for (int k = 0; k < 8 * SVR; ++k) vlo[k] = 0.0;
    static_for_dev<NT>([&](auto K_) {
        constexpr int k = decltype(K_)::value;
        const int t = __builtin_amdgcn_readfirstlane(rec[SW_TILE + w * NT + k]);
        const int I = t & 0xff, J = (t >> 8) & 0xff;
        tflag[k] = t < 0 ? 0 : (I == J ? 2 : 1);
        tI[k] = t < 0 ? 0 : ((t >> 16) & 0xff); tJ[k] = t < 0 ? 0 : ((t >> 24) & 0x7f);
        tpos[k] = __builtin_amdgcn_readfirstlane(rec[SW_TPOS + w * NT + k]);
        static_for_dev<8>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            const int row = 64 * I + 16 * (i >> 1) + 2 * rlv + (i & 1);
            static_for_dev<8>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                const int col = 64 * J + 16 * (j >> 1) + 2 * clv + (j & 1);
                const double x = (t >= 0 && row < q && col < q) ? xxp[(size_t)col * q + row] : 0.0;
                if constexpr (k < 2) areg_wr<k * 64 + i * 8 + j>(x);
                else if constexpr (i < SVR) vlo[i * 8 + j] = x;
                else Lt[(((i - SVR) * 4 + (j >> 1)) * 64 + lane) * 2 + (j & 1)] = x;
            });
        });
    });
    // ---- the coordinate this thread owns (eight lanes per coordinate: the interleaved chains of the sender sum)
    const int ocl = tid >> 3, part = tid & 7;
    const bool own = ocl < nsl;
    const int cg = c0 + (own ? ocl : 0);
    const double xyc = own ? A.xy[cg] : 0.0, pfc = own ? A.pf[cg] : 0.0;
    // GEN: this coordinate's group -- its members are the slice-local coordinates [gm0, gm1) (a run, inside this owner's slice)
    // gsplit > 0: a group of more than SSL members lies in the slices of k > 1 (<= gsplit) neighbouring owners, the first of them a (gfr[2 ocl] =
    // 2 a + 1 if its fragment does not start that owner's slice, gfr[2 ocl + 1] = k -- in LDS: the NT = 3 form has no register for them);
    // [gm0, gm1) then reaches beyond this slice and is clamped where it is used
    int gm0 = 0, gm1 = 0;
    bool gzr = true;
    if (GEN && part == 0) gwl[ocl] = 0.0;
    if (GEN && A.ngroups > 0 && own) {
        const int g = A.gid[cg];
        if (g >= 0) {
            const int m0 = A.gstart[g];
            gm0 = A.gidx[m0] - c0; gm1 = gm0 + (A.gstart[g + 1] - m0);
            gzr = A.gzero[g] != 0;
            if (part == 0) gwl[ocl] = A.gw[g];
        } else { gm0 = 0; gm1 = -1; }                                // (in no group: its coefficient stays 0, as path_update has it)
    }
    if (GEN && tid < SSL) {
        const bool sp = gsplit > 0 && A.ngroups > 0 && tid < nsl && A.gid[c0 + tid] >= 0;
        const int *ft = plan + SPLAN_HEAD + G * SW_INTS + 2 * (c0 + (sp ? tid : 0));
        gfr[2 * tid] = sp ? ft[0] : 0; gfr[2 * tid + 1] = sp ? ft[1] : 1;
    }
    double ak = 1.0;                                                 // Nesterov's sequence (ref src/oem_dense.h:529, 633-651)
    const bool want_loss = GEN && A.compute_loss != 0, accel = GEN && A.accelerate != 0;
    // GEN: oem.xtx's scale.factor (and oemSparse's intercept slot): oemXTX::get_beta rescales the iterate IN PLACE when a lambda ends (ref
    // src/oem_xtx.h:576-581, quirk Q5), so the next lambda starts from beta / s -- whose product is not the one in hand.  The owners
    // rescale, publish the rescaled coordinates in a round of their own (no operator, not an iteration) and the path goes on from there.
    // (1 / s of this coordinate is read where it is used, once per lambda: the NT = 3 general form has no register left for it)
    bool resc = false;
    const double yy = want_loss ? A.stats[2] : 0.0, nobs = want_loss ? A.stats[3] : 0.0;
    int goff[SE1];                                       // byte offsets (inside a parity) of this thread's senders of coordinate cg
    unsigned need1 = 0;
    {
        const int B = cg >> 6, nsB = blkbase[B + 1] - blkbase[B];
#pragma unroll
        for (int e = 0; e < SE1; ++e) {
            const int r = part + 8 * e;
            // (the first lane group of a wave that owns no coordinate gathers the pairs of coordinate c0 as well -- for their tags alone: every
            // wave -- and a workgroup that owns nothing, general form with a few large groups -- then holds the "still moving" OR over ALL
            // coordinates by itself, see the vote below)
            const bool ok = (own || lane < 8) && e < e1n && r < nsB;
            goff[e] = ok ? ((blkbase[B] + r) * 64 + (cg & 63)) * 16 : 0;
            if (ok) need1 |= 1u << e;
        }
    }
    __syncthreads();                                     // P1
    int p1m[4];                                          // exchange 1: where this wave's block slots go (pairs index inside a parity)
#pragma unroll
    for (int m = 0; m < 4; ++m) p1m[m] = __builtin_amdgcn_readfirstlane(w + 4 * m < nb ? P1[w + 4 * m] : 0);
    const double tol = sx_uni(A.tol);
    const int maxit = A.maxit, npen = A.npen;
    int g2off[SE2];                                      // exchange 2: the coordinates of the blocks this workgroup touches (slot w + 4 k: one slot per wave and k)
    unsigned need2 = 0;
#pragma unroll
    for (int k = 0; k < SE2; ++k) {
        const int s = w + 4 * k;
        const int j = (s < nb ? rec[SW_BLK + s] : 0) * 64 + lane;
        const bool ok = s < nb && j < q;
        g2off[k] = ok ? j * 16 : 0;
        if (ok) need2 |= 1u << k;
    }
    SymX X;
    X.s1 = nsum * 64 * 16; X.s2 = T * 64 * 16;
    X.o2 = 2 * X.s1; X.o3 = X.o2 + 2 * X.s2; X.o4 = X.o3 + 2 * G * 16;
    const int o5 = X.o4 + 2 * G * 16;                    // the owners' parts of split groups' squared norms, [2 parities][G][2] pairs
    X.rs = __builtin_amdgcn_make_buffer_rsrc((void *)xchg, 0, o5 + 4 * G * 16, 0x00020000);
    X.epoch = 0; X.wg = wg; X.G = G; X.failed = false; X.abortp = reinterpret_cast<const int *const *>(wv); X.aflag = wv + 2;
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;
    X.last = __builtin_amdgcn_s_memtime();
#endif
    int rpar = 0;

    // ---- Lanczos start vector (every owner keeps its coordinates of v, v_prev), the blocks of v into Bsh
    auto start_v = [](unsigned j) { const unsigned h = j * 2654435761u + 12345u; return (double)(h >> 8) * (1.0 / 16777216.0) - 0.5; };
    double ca, cb = 0.0;                                 // Lanczos: this coordinate of v, v_prev; path: beta_t, -
    {
        double nn = 0.0;
        for (int j = tid; j < q; j += SNTH) { const double x = start_v((unsigned)j); nn = fma(x, x, nn); }
        nn = 1.0 / sqrt(sx_block_sum(nn, red, rpar, w, lane));
        ca = own ? start_v((unsigned)cg) * nn : 0.0;
#pragma unroll
        for (int k = 0; k < SE2; ++k) {
            const bool nd = ((need2 >> k) & 1u) != 0;
            if (nd) Bsh[(w + 4 * k) * 64 + lane] = start_v((unsigned)(g2off[k] >> 4)) * nn;
            if (lane == 0 && w + 4 * k < SNB) nzs[w + 4 * k] = nd ? 15 : 0;
        }
        __syncthreads();
    }
    int msteps = A.lanczos_steps > SCML ? SCML : A.lanczos_steps;
    if (q < SCML) msteps = msteps < q ? msteps : q;
    if (msteps < 1) msteps = 1;
    double *theta_slot = red + 8;
    auto top_ritz = [&](int m, double hint) {
        if (w == 0) {
            const double th = tridiag_max(Tal, Tbe, m, lane, sturm, hint);
            if (lane == 0) theta_slot[0] = th;
        }
        __syncthreads();
        const double th = theta_slot[0];
        __syncthreads();
        return th;
    };
    int nst = 0;
    double bprev = 0.0, theta_prev = -__builtin_inf(), mv_prev = __builtin_inf(), d = 0.0;
    // path state (identical in every workgroup)
    int phase = 0, pp = 0, i = 0, it = 0, pen = 0, mw = 0;
    bool fresh = true;
    const int nl = A.nl;
    double scaley = 1.0, llo = 0.0, lhi = 0.0, lstep = 0.0;
    bool lflip = false;
    auto lambda_of = [&](int pq, int iq) {
        if (A.user_lambda) return A.lambda_user[(size_t)pq * nl + iq];
        double lv;
        if (nl == 1) lv = lhi;
        else if (lflip) lv = (iq == 0) ? llo : lhi - (double)(nl - 1 - iq) * lstep;
        else lv = (iq == nl - 1) ? lhi : llo + (double)iq * lstep;
        double lam = exp(lv);
        if (pen_is_net(A.penalty[pq])) lam = lam / A.alpha;
        return lam;
    };
    auto set_lambda = [&]() {                                        // (uniform: every thread of every workgroup takes it together)
        const double lam = lambda_of(pp, i);
        __syncthreads();
        if (tid == 0) sx_thr_store(thr, votes + 8, pen_consts(pen, lam / scaley, d, A.alpha, A.gamma, A.tau), d);
        __syncthreads();
    };

    // ONE loop for the Lanczos steps (phase 0) and the OEM iterations (phase 1): both are product -> exchange 1 -> the owners'
    // arithmetic -> exchange 2.
    unsigned tick = 0u;
    for (;;) {
        const bool lz = phase == 0;
        if ((tick++ & 127u) == 0u) (void)sx_abort_seen(X);           // (PathArgs::abort_word: the caller's interrupt)
        // ---- products of this wave's tiles (those whose vector block holds a non-zero), reduced over the lanes, into Wp
        SX_STAMP(0);
        int fJ[NT], fI[NT];                                          // (all flag words asked for at once: one LDS latency, not six)
#pragma unroll
        for (int k = 0; k < NT; ++k) { fJ[k] = nzs[tJ[k]]; fI[k] = nzs[tI[k]]; }
        static_for_dev<NT>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            if (tflag[k] == 0) return;
            const unsigned mJ = (unsigned)__builtin_amdgcn_readfirstlane(fJ[k]), mI = tflag[k] == 1 ? (unsigned)__builtin_amdgcn_readfirstlane(fI[k]) : 0u;
            const bool dj = mJ != 0u, di = mI != 0u;
            double *Wd = Wp + (tpos[k] & 0xff) * 64, *Wt = Wp + ((tpos[k] >> 8) & 0xff) * 64;
            if (dj && di) sx_tile<true, true, k, SVR>(vlo, Lt, Bsh, tI[k] * 64, tJ[k] * 64, mI, mJ, Wd, Wt, lane, rlv, clv);
            else if (dj) sx_tile<true, false, k, SVR>(vlo, Lt, Bsh, tI[k] * 64, tJ[k] * 64, mI, mJ, Wd, Wt, lane, rlv, clv);
            else if (di) sx_tile<false, true, k, SVR>(vlo, Lt, Bsh, tI[k] * 64, tJ[k] * 64, mI, mJ, Wd, Wt, lane, rlv, clv);
            // a product that was skipped leaves zeros (the block sums read every row of their slot)
            if (!dj) Wd[lane] = 0.0;
            if (!di && tflag[k] == 1) Wt[lane] = 0.0;
        });
        SX_STAMP(1);
        __syncthreads();                                             // Wp is complete
        SX_STAMP(9);                                                 // (the wait behind the slowest wave's products)
        // ---- exchange 1: this workgroup's partial vector per block slot (wave w adds the entries of slots w, w + 4, ... in list
        // order) to the owners; Lanczos: its part of v'XXv as well
        ++X.epoch;
        const int par = (int)(X.epoch & 1u);
        {
            const unsigned tag1 = (X.epoch << 1) | (unsigned)(lz ? 0 : mw);
            double apart = 0.0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m >= nslw) continue;
                // eight reads, none depending on another; rows beyond the slot's own read the zero row behind Wp
                // (round 5: four reads where the slot has no more entries -- most slots -- measured no faster, 22.60 -> 22.66 ms: the second
                // code path cost more in scalar spills than the reads saved)
                double rr[8];
                int er = erow[m], ec = ecnt[m];
                asm volatile("" : "+s"(er), "+s"(ec));               // (recomputed addresses, not 32 loop-invariant address registers)
#pragma unroll
                for (int j = 0; j < 8; ++j) rr[j] = Wp[(j < ec ? er + j : 8 * NT) * 64 + lane];
                const double t = ((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]));
                sx_publish(X.rs, par * X.s1 + (p1m[m] + lane) * 16, t, tag1);
                if (lz) apart = fma(Bsh[(w + 4 * m) * 64 + lane], t, apart);
            }
            if (lz) {
                apart = sx_block_sum(apart, red, rpar, w, lane);
                if (tid == 0) sx_publish(X.rs, X.o3 + (par * G + wg) * 16, apart, X.epoch << 1);
            }
        }
        SX_STAMP(2);
        // (the operator's constants, asked for now: they arrive while the owners poll)
        double thc[TH_N];
#pragma unroll
        for (int k = 0; k < TH_N; ++k) thc[k] = thr[k];
        int thkind = votes[8];
        // ---- the owners: the partials of coordinate cg in sender order (senders part, part + 8, ...: one chain per lane, then the eight lanes)
        int bits = 0;
        double gsum;
        {
            double g[SE1];
            int off[SE1];
#pragma unroll
            for (int e = 0; e < SE1; ++e) off[e] = par * X.s1 + goff[e];
            sx_gather<SE1, 11>(off, need1, g, bits, X);
            double t = (g[0] + g[1]) + (g[2] + g[3]);
            t += dpp_mov<0xB1, 0xf>(t, 0.0);
            t += dpp_mov<0x4E, 0xf>(t, 0.0);
            t += dpp_mov<0x141, 0xf>(t, 0.0);
            gsum = t;
        }
        SX_STAMP(3);
        double alpha = 0.0;
        if (lz) {                                                    // thread t holds workgroup t's part: a fixed order
            int off[1] = {X.o3 + (par * G + (tid < G ? tid : 0)) * 16};
            double v[1];
            int fl = 0;
            sx_gather<1>(off, tid < G ? 1u : 0u, v, fl, X);
            alpha = sx_block_sum(v[0], red, rpar, w, lane);
        }
        // The OR of "beta_t moved against beta_{t-1}" over ALL coordinates.  The senders of ONE coordinate already carry it: for any other block B'
        // some workgroup's patch holds the tile (max(B, B'), min(B, B')), touches both blocks, sends to this coordinate's block B and has
        // OR-ed B' into its tag -- so in the path phase of the element-wise form every wave takes the OR of its own lanes' tags (every wave
        // gathers at least one coordinate, above) and there is NO barrier behind gather 1.  Lanczos and the general
        // form keep the workgroup vote (alpha's block sum / the owners' LDS words need the barrier anyway).
        int any;
        if (GEN || lz || OEM_SX_VOTE_BARRIER) {
            sx_vote(votes, w, lane, bits);
            __syncthreads();
            any = votes[0] | votes[1] | votes[2] | votes[3];
        } else any = __ballot(bits != 0) != 0ull ? 1 : 0;
        SX_STAMP(4);

        // ---- the owners' arithmetic
        double val, npart = 0.0, acc_akn = 1.0;
        int mybit = 0;
        if (lz) {
            val = (gsum - alpha * ca) - bprev * cb;                  // this coordinate of w' (eight lanes hold the same)
            npart = (own && part == 0) ? val * val : 0.0;
        } else {
            // iteration t + 1 of every workgroup: the state transition is taken identically everywhere (oem_symfused_kernel's)
            bool done_now = false;
            if (!fresh) {
                const bool conv = !any;
                if (conv || it >= maxit) {
                    const size_t kfin = (size_t)pp * nl + i;
                    if (GEN && A.sinv) {
                        int cgo = cg;
                        asm volatile("" : "+v"(cgo));                // (opaque: an address or a load hipcc can hoist out of the loop is a register pair the NT = 3 general form does not have)
                        ca *= A.sinv[cgo];
                    }
                    if (own && part == 0) A.beta[kfin * q + cg] = ca;
                    if (tid == 0 && writer) { A.niter[kfin] = conv ? it : maxit + 1; if (!want_loss) A.loss[kfin] = 1e99; }      // ref src/oem_base.h:94-109
                    if (want_loss) {
                        // sum (Y - X beta)^2 = yy + n beta'(XX beta - 2 XY) (ref src/oem_dense.h:759-770): gsum IS XX beta of the finished iterate
                        const double lp = sx_block_sum((own && part == 0) ? ca * (gsum - 2.0 * xyc) : 0.0, red, rpar, w, lane);
                        if (tid == 0) sx_publish(X.rs, X.o3 + (par * G + wg) * 16, lp, X.epoch << 1);
                        if (writer) {                                // (workgroup 0 adds the parts in workgroup order; nobody else waits)
                            int offl[1] = {X.o3 + (par * G + (tid < G ? tid : 0)) * 16};
                            double vl[1];
                            int fll = 0;
                            sx_gather<1>(offl, tid < G ? 1u : 0u, vl, fll, X);
                            const double tl = sx_block_sum(vl[0], red, rpar, w, lane);
                            if (tid == 0) A.loss[kfin] = yy + nobs * tl;
                        }
                    }
                    const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
                    if (i + 1 < nlam) { i = i + 1; resc = GEN && A.sinv != nullptr; }
                    else if (pp + 1 < npen) { pp = pp + 1; i = 0; fresh = true; pen = A.penalty[pp]; ak = 1.0; }
                    else done_now = true;
                    if (!done_now) {
                        set_lambda();
#pragma unroll
                        for (int k = 0; k < TH_N; ++k) thc[k] = thr[k];
                        thkind = votes[8];
                    }
                    it = 0;
                }
            }
            if (done_now) break;
            if (GEN && resc) {
                // the round of the rescaled iterate (above): beta / s goes out as it is, marked "moving" so that nobody takes this round
                // for a converged iteration; it is not counted
                val = ca; mybit = 1; resc = false;
            } else {
                // beta_{t+1} of this coordinate: u = d beta - g + XY, the operator, the stop rule (ref src/utils.cpp:537-549)
                const double b0 = fresh ? 0.0 : ca;
                double u = (d * b0 - (fresh ? 0.0 : gsum)) + xyc;
                double bn;
                if (GEN && thkind >= K_GRP) {
                    // group operators: u of this owner's coordinates through LDS, the squared norm of this coordinate's group in member order
                    if (thkind == K_SGL) u = soft1(u, pfc * thc[TH_L1], 1.0);
                    if (own && part == 0) uo[ocl] = u;
                    __syncthreads();
                    double s2 = 0.0;
                    const int ma = gm0 > 0 ? gm0 : 0, mb = gm1 < nsl ? gm1 : nsl;      // (whole groups: gm0, gm1)
                    for (int m = ma; m < mb; ++m) { const double x = uo[m]; s2 += x * x; }
                    if (gsplit > 0) {
                        // groups in several owners' slices: the owners' parts (each in member order) through one more tagged exchange -- the
                        // part of the fragment that holds a slice's first coordinate in the owner's pair 0, the other one's in pair 1 --, added
                        // in owner order by the eight lanes of every member (lane k: owners k, k + 8, ...; then the lanes): the same bits everywhere
                        const int gfa = gfr[2 * ocl], gfk = gfr[2 * ocl + 1];
                        const bool sp = gfk > 1 && own;
                        if (sp && part == 0 && ocl == ma) sx_publish(X.rs, o5 + ((par * G + wg) * 2 + (gm0 > 0 ? 1 : 0)) * 16, s2, X.epoch << 1);
                        double t = 0.0;
                        for (int f0 = 0; f0 < gsplit; f0 += 16) {
                            double gp[2];
                            int offp[2], flp = 0;
                            unsigned needp = 0;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int fr = f0 + part + 8 * e;
                                const bool okp = sp && fr < gfk;
                                offp[e] = okp ? o5 + (par * G * 2 + (fr == 0 ? gfa : ((gfa >> 1) + fr) * 2)) * 16 : 0;
                                if (okp) needp |= 1u << e;
                            }
                            sx_gather<2>(offp, needp, gp, flp, X);
                            t += gp[0]; t += gp[1];
                        }
                        t += dpp_mov<0xB1, 0xf>(t, 0.0);
                        t += dpp_mov<0x4E, 0xf>(t, 0.0);
                        t += dpp_mov<0x141, 0xf>(t, 0.0);
                        if (sp) s2 = t;
                    }
                    double f = 1.0;
                    if (gm1 < gm0) f = 0.0;
                    else if (!gzr) {
                        const double sn = sqrt(s2), pen_g = thc[TH_L] * gwl[ocl];
                        if (thkind == K_GRP || thkind == K_SGL) { const double t = 1.0 - pen_g / sn; f = (0.0 < t) ? t : 0.0; }     // (quirk Q6: 0 / 0 -> NaN -> 0)
                        else if (thkind == K_GRP_MCP) f = mcp_norm(sn, pen_g, thc[TH_D], thc[TH_GAMMA]);
                        else f = scad_norm(sn, pen_g, thc[TH_D], thc[TH_GAMMA]);
                    }
                    bn = (own && f != 0.0) ? u * f / thc[TH_D] : 0.0;
                    __syncthreads();                                     // (uo is written again next iteration)
                } else bn = own ? sx_op(u, pfc, thkind, thc) : 0.0;
                double adp_part = 0.0, akn = 1.0;
                if (accel) {                                             // ref src/oem_dense.h:633-651
                    akn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * ak * ak));
                    const double ratio = (ak - 1.0) / akn, upd = bn, diff = upd - b0;
                    bn = upd + ratio * diff;
                    adp_part = (own && part == 0) ? (bn - upd) * diff : 0.0;
                }
                const double cu = fabs(bn), qo = fabs(b0);
                const bool cn = cu > 1e-13, qn = qo > 1e-13;
                mybit = (own && ((cn != qn) || (cn && qn && fabs(bn - b0) > tol * qo))) ? 1 : 0;
                ca = bn; fresh = false; ++it;
                val = bn;
                if (accel) { npart = adp_part; acc_akn = akn; }
            }
        }

        // ---- exchange 2: the owners' values out, the blocks this workgroup touches in (Lanczos: scaled by 1 / ||w'||, whose
        // squared parts ride along)
        if (own && part == 0) sx_publish(X.rs, X.o2 + par * X.s2 + cg * 16, val, (X.epoch << 1) | (unsigned)mybit);
        const bool scal = lz || (accel && !lz);                     // a sum over all coordinates rides along: ||w'||^2 (Lanczos) or Nesterov's restart test
        if (scal) {
            const double np = sx_block_sum(npart, red, rpar, w, lane);
            if (tid == 0) sx_publish(X.rs, X.o4 + (par * G + wg) * 16, np, X.epoch << 1);
        }
        SX_STAMP(5);
        int bits2 = 0;
        double r[SE2];
        {
            int off[SE2];
#pragma unroll
            for (int k = 0; k < SE2; ++k) off[k] = X.o2 + par * X.s2 + g2off[k];
            sx_gather<SE2, 13>(off, need2, r, bits2, X);
        }
        SX_STAMP(6);
        double bb = 0.0, ib = 1.0;
        if (scal) {
            int off[1] = {X.o4 + (par * G + (tid < G ? tid : 0)) * 16};
            double v[1];
            int fl = 0;
            sx_gather<1>(off, tid < G ? 1u : 0u, v, fl, X);
            const double tot = sx_block_sum(v[0], red, rpar, w, lane);
            if (lz) sqrt_rsqrt(tot, bb, ib);
            else ak = (tot > 0.0) ? 1.0 : acc_akn;                   // (the extrapolated beta is kept; only the momentum counter restarts)
        }
        // (a pair this thread did not ask for -- a slot beyond the workgroup's blocks, a coordinate beyond q -- came back as 0.0: stored like
        // the others, into slots nobody reads / over the zeros the ragged end of the last block holds anyway.  No predicate, no masked store:
        // at one wave per SIMD an instruction is ~5 cycles, and this tail was 300 of them -- 1,570 cycles, profiles/r5_symcoop_stamped.txt)
        static_assert(SE2 * 4 <= SNB, "every slot w + 4 k is a row of Bsh");
        if (lz) {
#pragma unroll
            for (int k = 0; k < SE2; ++k) r[k] *= ib;
        }
#pragma unroll
        for (int k = 0; k < SE2; ++k) {
            const double x = r[k];
            Bsh[(w + 4 * k) * 64 + lane] = x;
            const unsigned long long nzb = __ballot(x != 0.0);             // bit h of the word: 16-coordinate group h of the block holds a non-zero
            const unsigned nlo = (unsigned)nzb, nhi = (unsigned)(nzb >> 32);
            const int nz4 = ((nlo & 0xffffu) ? 1 : 0) | ((nlo >> 16) ? 2 : 0) | ((nhi & 0xffffu) ? 4 : 0) | ((nhi >> 16) ? 8 : 0);
            if (lane == 0) nzs[w + 4 * k] = nz4;
        }
        SX_STAMP(10);                                                // (LDS stores and ballots; what follows is the wait behind the slowest wave's gather)
        sx_vote(votes + 4, w, lane, bits2);
        __syncthreads();
        SX_STAMP(7);
#ifdef OEM_PATH_DIAG
        X.acc[8] += 1;
#endif
        {
            const int v2 = votes[4] | votes[5] | votes[6] | votes[7] | X.aflag[0];
            if (v2 & 2) break;                                       // somebody here has seen the host's abort word: nobody waits any more, leave
            if (!lz) { mw = v2; continue; }
        }

        // ---- Lanczos bookkeeping (replicated): T, the stop rule, and at the end d and the hand-over to the path
        if (tid == 0) { Tal[nst] = alpha; Tbe[nst] = bb; }
        ++nst;
        bool fin = !(bb > 1e-13 * fabs(alpha)) || nst >= msteps;     // invariant subspace reached (T is exact), or the step cap
        bool have_theta = false;
        double theta = 0.0;
        if (!fin && lanczos_check_due(nst)) {
            theta = top_ritz(nst, theta_prev);                       // its first barrier publishes Tal / Tbe
            if (lanczos_converged(theta, theta_prev, mv_prev)) { fin = true; have_theta = true; }
            theta_prev = sx_uni(theta_prev); mv_prev = sx_uni(mv_prev);
        }
        if (!fin) { cb = ca; ca = val * ib; bprev = sx_uni(bb); continue; }
        if (!have_theta) { __syncthreads(); theta = top_ritz(nst, theta_prev); }
        d = sx_uni(theta * 1.005);                                   // ref src/oem_dense.h:498, src/oem_xtx.h:369
        if (tid == 0 && writer) {
            A.d_out[0] = d; A.d_out[1] = theta; A.d_out[4] = (double)nst;
            A.d_out[5] = (!have_theta && nst >= msteps && nst < q && bb > 1e-13 * fabs(alpha)) ? 1.0 : 0.0;
        }
#ifdef OEM_PATH_DIAG
        SX_STAMP(0);
        if (tid == 0 && writer) { for (int k = 0; k < 8; ++k) g_diag_symcoop[k] = X.acc[k]; }
        for (int k = 0; k < 16; ++k) X.acc[k] = 0;
#endif
        if (A.npen == 0) break;
        // lambda grid constants (ref src/oem_dense.cpp:175-192), replicated
        scaley = sx_uni(A.yscale ? A.stats[1] : 1.0);
        {
            double m = 0.0;
            for (int j = tid; j < q; j += SNTH) {
                const double xl = A.lmax_xy ? A.lmax_xy[j] : A.xy[j];
                m = fmax(m, j >= A.lmax_from ? fabs(xl) : 0.0);
            }
            m = wave_max(m);
            if (lane == 0) red[12 + w] = m;
            __syncthreads();
            const double lmax = fmax(fmax(red[12], red[13]), fmax(red[14], red[15])) * scaley;
            llo = sx_uni(log(lmax)); lhi = sx_uni(log(A.lambda_min_ratio * lmax));
            lstep = sx_uni(nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0);
            lflip = fabs(lhi) < fabs(llo);
        }
        if (writer) for (int idx = tid; idx < A.npen * nl; idx += SNTH) A.lambda_out[idx] = lambda_of(idx / nl, idx % nl);
        // cold start (ref src/oem_dense.cpp:243-244): beta = 0 in Bsh, nothing to multiply
        phase = 1; pp = 0; i = 0; it = 0; pen = A.penalty[0]; fresh = true; mw = 0; ca = 0.0; cb = 0.0;
        set_lambda();
        for (int k = tid; k < SNB * 64; k += SNTH) Bsh[k] = 0.0;
        if (tid < SNB) nzs[tid] = 0;
        __syncthreads();
    }
#ifdef OEM_PATH_DIAG
    if (tid == 0 && writer) { for (int k = 0; k < 8; ++k) g_diag_symcoop[8 + k] = X.acc[k]; g_diag_symcoop[7] = X.acc[8]; }
    if (lane == 0 && writer) { for (int k = 0; k < 16; ++k) g_diag_symcoop_waves[w * 16 + k] = X.acc[k]; }
#endif
    if (tid == 0 && writer) {
        A.d_out[2] = (double)(__builtin_amdgcn_s_memtime() - t_cyc0);
        A.d_out[3] = (double)(__builtin_amdgcn_s_memrealtime() - t_rt0);
    }
    if (__syncthreads_or(X.failed ? 1 : 0) && tid == 0) A.d_out[6] = 1.0;       // exchange timeout: poison (api.hip: run_paths falls back)
}

template <int NT, bool GEN> constexpr size_t symcoop_lds_bytes()
{
    return sizeof(double) * (size_t)(SNB * 64 + (8 * NT + 1) * 64 + 2 * SCML + 2 * (SCML + 16) + 16 + 16) + sizeof(int) * (size_t)(16 + SNB + 32 + 2 * SSL + SW_INTS + SNB + 8 + 4 * SSL) +
           (NT == 3 ? sizeof(double) * 4 * sx_slt(GEN) : 0);
}

// ================================================================================================================================
// 1024 < q <= 2048, element-wise penalties: the ROW-SPLIT form -- ONE exchange per iteration.
//
// At these sizes the WHOLE matrix (32 MB at q = 2048) fits the accumulator files of q / 16 <= 128 CUs: workgroup g keeps rows
// 16 g .. 16 g + 15 -- a 16-lane row group of a wave 128 columns of them, 128 doubles per lane = a0..a255, named by inline asm alone
// (path_dev.hpp: areg_rd) -- multiplies them with v_fmac_f64_dpp row_newbcast as path_coop.hip does (each lane supplies eight vector
// entries to its row, the four slices of a row meet through v_permlane16/32_swap, the four waves through LDS) and so holds u of ITS
// rows complete: no partial vectors, no reduce-scatter.  The operator runs on the sixteen owner lanes, and the one exchange is the
// all-gather of the new coefficients (tagged pairs, the "still moving" bit of a coordinate in its tag: every workgroup gathers all of
// them, so the stop decision is everybody's in the SAME iteration).  path_coop.hip extended to these sizes does not close its
// register budget (its replicated update at eight coordinates per thread: DESIGN section 0, item 4); here a thread owns at most one
// coordinate.  16-column groups of the vector that are all zero are skipped (a lasso iterate is sparse).  Lanczos the same way, the
// vector updates replicated from the gathered product (LDS-resident v, v_prev).  compute.loss and (round 6, template parameter ACC) Nesterov's
// step run here; group operators and scale.factor go to the symmetric engine above / the launches.
// ================================================================================================================================
constexpr int RQ = 2048;          // columns a workgroup covers: 4 waves x 4 row groups x 128
constexpr int RE = RQ / SNTH;     // pairs per thread in the all-gather

#ifdef OEM_PATH_DIAG
__device__ unsigned long long g_diag_rowcoop[16];
#endif

constexpr int RVG = 48;           // the first RVG of a lane's 128 matrix entries live in VGPRs (an accumulator value costs two v_accvgpr_read_b32
                                  // per use: 8 of the 12 cycles of a column; the kernel has the registers for three groups of sixteen)
template <int C> struct RowFma {
    static __device__ __forceinline__ void run(double (&acc)[4], const double (&B)[8], const double (&xv)[RVG], unsigned nzmask)
    {
        if constexpr (C < 128) {
            if constexpr ((C & 15) == 0) {
                // sixteen columns at a time; a group whose vector entries are zero in every lane of the wave is skipped (wave-uniform)
                if ((nzmask >> (C >> 4)) & 1u) RowFma16<C>(acc, B, xv);
                RowFma<C + 16>::run(acc, B, xv, nzmask);
            }
        }
    }
    template <int C0> static __device__ __forceinline__ void RowFma16(double (&acc)[4], const double (&B)[8], const double (&xv)[RVG])
    {
        static_for_dev<16>([&](auto K_) {
            constexpr int c = C0 + decltype(K_)::value;
            if constexpr (c < RVG) BcFma<(c & 15)>::fmac(acc[c & 3], B[c >> 4], xv[c]);
            else {
                const double x = areg_rd<c>();
                BcFma<(c & 15)>::fmac(acc[c & 3], B[c >> 4], x);
            }
        });
    }
};

// ACC: Nesterov's step (ref src/oem_dense.h:633-651) -- beta = T(u) + r (T(u) - beta_prev); its restart test, the sign of
// sum_j (beta_j - T(u)_j)(T(u)_j - beta_prev_j) over ALL coordinates, as the workgroups' parts: one more tagged pair per workgroup published next to
// the coefficients and gathered behind them (thread t holds workgroup t's part: a fixed order) -- the ONE exchange per iteration stays one.
template <bool ACC>
__global__ __launch_bounds__(SNTH) void path_rowcoop_kernel(PathArgs A, unsigned long long *xchg)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15, grp = lane >> 4;
    const int q = A.p, wg = blockIdx.x, G = gridDim.x;
    const unsigned long long t_cyc0 = __builtin_amdgcn_s_memtime(), t_rt0 = __builtin_amdgcn_s_memrealtime();
    double *Bsh = lds;                                   // the vector going into the product [RQ + 16] (zero words behind it)
    double *Vp = Bsh + RQ + 16;                          // Lanczos: v_prev [RQ]
    double *Wv = Vp + RQ;                                // Lanczos: the gathered product [RQ]
    double *Tal = Wv + RQ, *Tbe = Tal + SCML;
    double *sturm = Tbe + SCML;
    double *red = sturm + 2 * (SCML + 16);               // block reductions [2][4], theta slot [8], lmax words [12..16)
    double *thr = red + 16;
    double *Pc = thr + 16;                               // the waves' parts of the sixteen row sums [4][16]
    int *votes = reinterpret_cast<int *>(Pc + 64);       // [8] votes, [8] kind, [2] PathArgs::abort_word (SymX::abortp), [1] SymX::aflag
    const bool writer = wg == 0;
    asm volatile("" ::: "a255");                         // the accumulator file is in use (by the asm alone)
    if (tid == 0) { *reinterpret_cast<const int **>(votes + 16) = A.abort_word; votes[18] = 0; }      // (read behind the barriers of the set-up below)

    // ---- this lane's 128 matrix entries: row 16 wg + l16, columns 512 w + 128 grp + k
    const int row = 16 * wg + l16, cbase = 512 * w + 128 * grp;
    const bool rowok = row < q;
    double xv[RVG];
    static_for_dev<128>([&](auto K_) {
        constexpr int k = decltype(K_)::value;
        const int col = cbase + k;
        const double t = (rowok && col < q) ? A.xx[(size_t)col * q + row] : 0.0;
        if constexpr (k < RVG) xv[k] = t; else areg_wr<k>(t);
    });
    int bidx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int col = cbase + 16 * j + l16; bidx[j] = col < q ? col : RQ + (lane & 7); }
    for (int j = tid; j < RQ + 16; j += SNTH) Bsh[j] = 0.0;
    for (int j = tid; j < 2 * RQ; j += SNTH) Vp[j] = 0.0;
    // the coordinate this thread owns: lanes 0..15 of wave 0
    const bool own = w == 0 && lane < 16 && rowok;
    const double xyc = own ? A.xy[row] : 0.0, pfc = own ? A.pf[row] : 0.0;
    SymX X;
    X.s1 = 0; X.s2 = RQ * 16; X.o2 = 0; X.o3 = 0; X.o4 = 0;
    X.rs = __builtin_amdgcn_make_buffer_rsrc((void *)xchg, 0, 2 * RQ * 16 + 4 * (RQ / 16) * 16, 0x00020000);      // (+ compute.loss, + Nesterov's restart test: the workgroups' parts, [2][RQ / 16] pairs each)
    X.epoch = 0; X.wg = wg; X.G = G; X.failed = false; X.abortp = reinterpret_cast<const int *const *>(votes + 16); X.aflag = votes + 18;
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;
    X.last = __builtin_amdgcn_s_memtime();
#endif
    int g2off[RE];
    unsigned need2 = 0;
#pragma unroll
    for (int k = 0; k < RE; ++k) { const int j = tid + SNTH * k; g2off[k] = j < q ? j * 16 : 0; if (j < q) need2 |= 1u << k; }
    int rpar = 0;
    const double tol = sx_uni(A.tol);
    const int maxit = A.maxit, npen = A.npen, nl = A.nl;
    __syncthreads();

    // (M vec)[row] for the sixteen rows of this workgroup, complete, in lanes 0..15 of wave 0 (the other lanes: garbage)
    auto product = [&]() -> double {
        SX_STAMP(0);
        double B[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) B[j] = Bsh[bidx[j]];
        unsigned nz = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) nz |= (__ballot(B[j] != 0.0) != 0ull ? 1u : 0u) << j;
        dpp_hazard_fence(B);
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        RowFma<0>::run(acc, B, xv, nz);
        const double s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        // the four row groups (column slices) of the wave: v_permlane16_swap, then v_permlane32_swap
        const double t = sx_swap_add<false>(s, s);
        const double g4 = sx_swap_add<true>(t, t);
        if (lane < 16) Pc[w * 16 + lane] = g4;
        SX_STAMP(1);
        __syncthreads();
        return (Pc[l16] + Pc[16 + l16]) + (Pc[32 + l16] + Pc[48 + l16]);
    };
    // the owners' values out, all q coordinates in: Dst[j] (behind a barrier); returns the OR of the bits in the tags
    // ACC, xsum != null: *xsum = the sum of the workgroups' parts of Nesterov's restart test (rpart: this lane's part), thread t holding
    // workgroup t's -- a fixed order.  The parts go out BEHIND the coefficients and their load is issued when the coefficients are all there,
    // in front of the LDS stores; its wave sums meet behind the gather's own barrier: no round trip and no barrier of their own
    auto all_gather = [&](double val, int mybit, double *Dst, double rpart = 0.0, double *xsum = nullptr) -> int {
        ++X.epoch;
        const int par = (int)(X.epoch & 1u);
        if (own) sx_publish(X.rs, par * X.s2 + row * 16, val, (X.epoch << 1) | (unsigned)mybit);
        if constexpr (ACC) {
            // Nesterov: this workgroup's part of the restart test (its sixteen owner lanes are lanes 0..15 of wave 0, the others hold 0),
            // published behind the coefficients
            if (xsum && w == 0) {
                const double part = wave_sum(rpart);
                if (tid == 0) sx_publish(X.rs, 2 * RQ * 16 + 2 * (RQ / 16) * 16 + par * (RQ / 16) * 16 + wg * 16, part, X.epoch << 1);
            }
        }
        SX_STAMP(2);
        int bits = 0;
        double r[RE];
        int off[RE];
#pragma unroll
        for (int k = 0; k < RE; ++k) off[k] = par * X.s2 + g2off[k];
        sx_gather<RE>(off, need2, r, bits, X);
        SX_STAMP(3);
        sx_v4u xp = sx_v4u{0u, 0u, 0u, 0u};
        const int xoff = 2 * RQ * 16 + 2 * (RQ / 16) * 16 + par * (RQ / 16) * 16 + (tid < G ? tid : 0) * 16;
        if constexpr (ACC) { if (xsum && tid < G) xp = __builtin_amdgcn_raw_buffer_load_b128(X.rs, xoff, 0, 16); }
        // (a coordinate beyond q came back as 0.0 and is stored like the others, over the zeros that sit there: no masked stores)
#pragma unroll
        for (int k = 0; k < RE; ++k) Dst[tid + SNTH * k] = r[k];
        sx_vote(votes, w, lane, bits);
        double *xr = red + 4 * rpar;
        if constexpr (ACC) {
            if (xsum) {
                const bool xn = tid < G;
                double xv = __hiloint2double((int)xp.z, (int)xp.x);
                if (__any(xn && ((xp.y >> 1) != X.epoch || (xp.w >> 1) != X.epoch))) {      // (a part that is not there yet -- rare: it left before the last coefficient arrived)
                    int offa[1] = {xoff};
                    double va[1];
                    int fla = 0;
                    sx_gather<1>(offa, xn ? 1u : 0u, va, fla, X);
                    xv = va[0];
                }
                const double xs = wave_sum(xn ? xv : 0.0);
                if (lane == 0) xr[w] = xs;
            }
        }
        __syncthreads();
        if constexpr (ACC) { if (xsum) { *xsum = (xr[0] + xr[1]) + (xr[2] + xr[3]); rpar ^= 1; } }
        SX_STAMP(4);
#ifdef OEM_PATH_DIAG
        X.acc[8] += 1;
#endif
        return votes[0] | votes[1] | votes[2] | votes[3] | X.aflag[0];      // (bit 1: the host's abort word was seen in this workgroup)
    };

    // ---- eigenvalue step: Lanczos, the vector updates replicated from the gathered product (v in Bsh, v_prev in Vp)
    auto start_v = [](unsigned j) { const unsigned h = j * 2654435761u + 12345u; return (double)(h >> 8) * (1.0 / 16777216.0) - 0.5; };
    {
        double nn = 0.0;
        for (int j = tid; j < q; j += SNTH) { const double x = start_v((unsigned)j); nn = fma(x, x, nn); }
        nn = 1.0 / sqrt(sx_block_sum(nn, red, rpar, w, lane));
        for (int j = tid; j < q; j += SNTH) Bsh[j] = start_v((unsigned)j) * nn;
        __syncthreads();
    }
    int msteps = q < SCML ? q : SCML;
    double *theta_slot = red + 8;
    auto top_ritz = [&](int m, double hint) {
        if (w == 0) {
            const double th = tridiag_max(Tal, Tbe, m, lane, sturm, hint);
            if (lane == 0) theta_slot[0] = th;
        }
        __syncthreads();
        const double th = theta_slot[0];
        __syncthreads();
        return th;
    };
    int nst = 0;
    double bprev = 0.0, theta = 0.0, theta_prev = -__builtin_inf(), mv_prev = __builtin_inf();
    bool have_theta = false;
    for (int js = 0; js < msteps; ++js) {
        const double g = product();
        (void)all_gather(g, 0, Wv);
        double al = 0.0;
        for (int j = tid; j < q; j += SNTH) al = fma(Bsh[j], Wv[j], al);
        al = sx_block_sum(al, red, rpar, w, lane);
        double bb2 = 0.0;
        for (int j = tid; j < q; j += SNTH) { const double x = (Wv[j] - al * Bsh[j]) - bprev * Vp[j]; Wv[j] = x; bb2 = fma(x, x, bb2); }
        double bb, ib;
        sqrt_rsqrt(sx_block_sum(bb2, red, rpar, w, lane), bb, ib);
        if (tid == 0) { Tal[js] = al; Tbe[js] = bb; }
        nst = js + 1;
        if (!(bb > 1e-13 * fabs(al))) break;                         // invariant subspace reached: T is exact
        if (lanczos_check_due(nst) && nst < msteps) {
            const double th = top_ritz(nst, theta_prev);
            if (lanczos_converged(th, theta_prev, mv_prev)) { theta = th; have_theta = true; break; }
            theta_prev = sx_uni(theta_prev); mv_prev = sx_uni(mv_prev);
        }
        for (int j = tid; j < q; j += SNTH) { Vp[j] = Bsh[j]; Bsh[j] = Wv[j] * ib; }        // (each thread its own entries: no barrier between)
        bprev = sx_uni(bb);
        __syncthreads();
    }
    if (!have_theta) { __syncthreads(); theta = top_ritz(nst, theta_prev); }
    const double d = sx_uni(theta * 1.005);                          // ref src/oem_dense.h:498, src/oem_xtx.h:369
    if (tid == 0 && writer) { A.d_out[0] = d; A.d_out[1] = theta; A.d_out[4] = (double)nst; A.d_out[5] = (!have_theta && nst >= msteps && nst < q) ? 1.0 : 0.0; }
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;
#endif

    // ---- lambda grid constants (ref src/oem_dense.cpp:175-192), replicated
    const double scaley = sx_uni(A.yscale ? A.stats[1] : 1.0);
    double lmax = 0.0;
    {
        double m = 0.0;
        for (int j = tid; j < q; j += SNTH) {
            const double xl = A.lmax_xy ? A.lmax_xy[j] : A.xy[j];
            m = fmax(m, j >= A.lmax_from ? fabs(xl) : 0.0);
        }
        m = wave_max(m);
        if (lane == 0) red[12 + w] = m;
        __syncthreads();
        lmax = fmax(fmax(red[12], red[13]), fmax(red[14], red[15])) * scaley;
    }
    const double llo = sx_uni(log(lmax)), lhi = sx_uni(log(A.lambda_min_ratio * lmax));
    const double lstep = sx_uni(nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0);
    const bool lflip = fabs(lhi) < fabs(llo);
    auto lambda_of = [&](int pq, int iq) {
        if (A.user_lambda) return A.lambda_user[(size_t)pq * nl + iq];
        double lv;
        if (nl == 1) lv = lhi;
        else if (lflip) lv = (iq == 0) ? llo : lhi - (double)(nl - 1 - iq) * lstep;
        else lv = (iq == nl - 1) ? lhi : llo + (double)iq * lstep;
        double lam = exp(lv);
        if (pen_is_net(A.penalty[pq])) lam = lam / A.alpha;
        return lam;
    };
    if (writer) for (int idx = tid; idx < npen * nl; idx += SNTH) A.lambda_out[idx] = lambda_of(idx / nl, idx % nl);

    // ---- the penalty x lambda path (ref src/oem_base.h:90-110): product, the operator on the owner lanes, ONE all-gather
    unsigned lep = 0;                                            // compute.loss: the epoch of its own little exchange
    for (int pp = 0; pp < npen; ++pp) {
        const int pen = A.penalty[pp];
        const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
        __syncthreads();
        for (int j = tid; j < RQ; j += SNTH) Bsh[j] = 0.0;           // cold start (ref src/oem_dense.cpp:243-244)
        double bc = 0.0;                                             // this owner lane's coefficient
        double ak = 1.0;                                             // Nesterov's sequence, reset at every penalty's cold start (ref src/oem_dense.h:744), not between lambdas
        __syncthreads();
        for (int i = 0; i < nlam; ++i) {
            const double lam = lambda_of(pp, i);
            __syncthreads();
            if (tid == 0) sx_thr_store(thr, votes + 8, pen_consts(pen, lam / scaley, d, A.alpha, A.gamma, A.tau), d);
            __syncthreads();
            double thc[TH_N];
#pragma unroll
            for (int k = 0; k < TH_N; ++k) thc[k] = thr[k];
            const int thkind = votes[8];
            int it = 0;
            bool conv = false;
            while (it < maxit) {
                // (Nesterov's sequence depends on the previous restart test alone: its square root and division go in front of the product,
                // into the shadow of its LDS reads, not between the operator and the publication of the coefficients)
                double akn = 1.0, aratio = 0.0, rpart = 0.0;
                if (ACC) { akn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * ak * ak)); aratio = (ak - 1.0) / akn; }
                const double g = product();
                const double u = (d * bc - g) + xyc;                 // ref src/oem_dense.h:512
                double bn = own ? sx_op(u, pfc, thkind, thc) : 0.0;
                if (ACC) {
                    const double upd = bn, diff = upd - bc;
                    bn = upd + aratio * diff;
                    rpart = own ? (bn - upd) * diff : 0.0;           // (its wave sum and its pair go out BEHIND the coefficient, inside all_gather: nobody's product waits for them)
                }
                const double cu = fabs(bn), qo = fabs(bc);
                const bool cn = cu > 1e-13, qn = qo > 1e-13;         // ref src/utils.cpp:537-549
                const int moving = (own && ((cn != qn) || (cn && qn && fabs(bn - bc) > tol * qo))) ? 1 : 0;
                bc = bn;
                ++it;
                // PathArgs::abort_word (the caller's interrupt), looked at once per lambda and per 128 iterations; whoever sees it waits for
                // nobody any more, and the bit rides in this workgroup's vote: every loop is left (pp = npen says so -- no flag of its own:
                // this kernel has no scalar register to spare)
                if ((it & 127) == 1) (void)sx_abort_seen(X);
                double tot = 0.0;
                const int any = all_gather(bn, moving, Bsh, rpart, ACC ? &tot : nullptr);
                if (ACC) ak = (tot > 0.0) ? 1.0 : akn;               // (the extrapolated beta is kept; only the momentum counter restarts)
                if (any & 2) { pp = npen; break; }
                if (!any) { conv = true; break; }
            }
            if (pp >= npen) break;
            const size_t kfin = (size_t)pp * nl + i;
            if (own) A.beta[kfin * q + row] = bc;
            if (tid == 0 && writer) { A.niter[kfin] = conv ? it : maxit + 1; if (!A.compute_loss) A.loss[kfin] = 1e99; }      // ref src/oem_base.h:94-109
            if (A.compute_loss) {
                // sum (Y - X beta)^2 = yy + n beta'(XX beta - 2 XY) (ref src/oem_dense.h:759-770): one more product, of the FINISHED iterate (Bsh holds
                // it: the last all-gather), the workgroups' parts to workgroup 0 through pairs of their own -- once per lambda, nobody else waits
                const double gfin = product();
                const double lp = sx_block_sum(own ? bc * (gfin - 2.0 * xyc) : 0.0, red, rpar, w, lane);
                ++lep;
                const int lpar = (int)(lep & 1u), lbase = 2 * RQ * 16 + lpar * (RQ / 16) * 16;
                if (tid == 0) sx_publish(X.rs, lbase + wg * 16, lp, lep << 1);
                if (writer) {
                    const unsigned keep = X.epoch;
                    X.epoch = lep;                               // (sx_gather matches tags against X.epoch)
                    int offl[1] = {lbase + (tid < G ? tid : 0) * 16};
                    double vl[1];
                    int fll = 0;
                    sx_gather<1>(offl, tid < G ? 1u : 0u, vl, fll, X);
                    X.epoch = keep;
                    const double tl = sx_block_sum(vl[0], red, rpar, w, lane);      // thread t holds workgroup t's part: a fixed order
                    if (tid == 0) A.loss[kfin] = A.stats[2] + A.stats[3] * tl;
                }
            }
        }
        if (pp >= npen) break;                                       // (left on the abort word)
        if (tid == 0 && writer) for (int r = nlam; r < nl; ++r) { A.niter[(size_t)pp * nl + r] = 0; A.loss[(size_t)pp * nl + r] = 1e99; }
    }
#ifdef OEM_PATH_DIAG
    if (tid == 0 && writer) { for (int k = 0; k < 9; ++k) g_diag_rowcoop[k] = X.acc[k]; }
#endif
    if (tid == 0 && writer) {
        A.d_out[2] = (double)(__builtin_amdgcn_s_memtime() - t_cyc0);
        A.d_out[3] = (double)(__builtin_amdgcn_s_memrealtime() - t_rt0);
    }
    if (__syncthreads_or(X.failed ? 1 : 0) && tid == 0) A.d_out[6] = 1.0;       // exchange timeout: poison (api.hip: run_paths falls back)
}

constexpr size_t rowcoop_lds_bytes() { return sizeof(double) * (size_t)(RQ + 16 + 2 * RQ + 2 * SCML + 2 * (SCML + 16) + 16 + 16 + 64) + sizeof(int) * 20; }

}  // namespace

#ifdef OEM_PATH_DIAG
extern "C" __attribute__((visibility("default"))) int oemgpu_diag_read_rowcoop(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_rowcoop), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

// 1024 < q <= 2048, element-wise penalties: the one-exchange row-split form (OEM_NO_ROWCOOP=1: the symmetric engine)
bool path_rowcoop_eligible(const PathArgs &a, bool group_penalty)
{
    if (sw().OEM_NO_ROWCOOP.set || sw().OEM_NO_SYMCOOP.set || sw().OEM_NO_COOP.set) return false;
    if (a.p <= 1024 || a.p > RQ || a.nbatch > 1 || a.pen_split) return false;
    return !(a.sinv || group_penalty);
}
int path_rowcoop_workgroups(int q) { return (q + 15) / 16; }
size_t path_rowcoop_xchg_bytes() { return (size_t)2 * RQ * 16 + (size_t)4 * (RQ / 16) * 16 + 256; }
int launch_path_rowcoop(hipStream_t s, const PathArgs &a, void *xchg)
{
    OEM_HIP(hipMemsetAsync(xchg, 0, path_rowcoop_xchg_bytes(), s));            // the tags must start at 0
    OEM_HIP(hipMemsetAsync(a.d_out, 0, sizeof(double) * D_OUT_LEN, s));        // [6]: only a timed-out workgroup writes it
    const size_t sh = rowcoop_lds_bytes();
    void (*kern)(PathArgs, unsigned long long *) = a.accelerate ? path_rowcoop_kernel<true> : path_rowcoop_kernel<false>;
    if (int rc = lds_limit_once(reinterpret_cast<const void *>(kern), sh)) return rc;
    hipLaunchKernelGGL(kern, dim3(path_rowcoop_workgroups(a.p)), dim3(SNTH), sh, s, a, reinterpret_cast<unsigned long long *>(xchg));
    OEM_HIP(hipGetLastError());
    if (sw().OEM_WCOOP_FAKE_TIMEOUT.set) OEM_HIP(hipMemsetAsync(a.d_out + 6, 0xFF, sizeof(double), s));      // tests: the host's fallback
    return 0;
}

#ifdef OEM_PATH_DIAG
extern "C" __attribute__((visibility("default"))) int oemgpu_diag_read_symcoop(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_symcoop), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int oemgpu_diag_read_symcoop_waves(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_symcoop_waves), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

// ---- the plan: which tiles a wave holds, which blocks a workgroup touches, who sends to whom.  Pure host arithmetic.
// Tiles are enumerated in strips of `a` tile rows, column by column inside a strip, and dealt 4 NT at a time: a workgroup is a patch
// of a tile rows x 4 NT / a tile columns (3 x 4 at NT = 3), a wave one column of the strip (its tiles share their column block).
static bool symcoop_plan_per(int q, int gmax, SymcoopPlan &P, const int *runs, int nruns, int per_want);
bool symcoop_plan(int q, int gmax, SymcoopPlan &P, const int *runs, int nruns)
{
    // per_want = 0: 4 NT tiles per workgroup.  With group runs the owners' slices want slack (whole runs, <= SSL coordinates): start from
    // about 18 coordinates per owner and give the workgroups more tiles until every limit of the plan holds
    if (!runs) return symcoop_plan_per(q, gmax, P, nullptr, 0, 0);
    // A run longer than an owner holds (a group of more than SSL members) may be cut after every fourth of its coordinates: the owners'
    // slices end at run boundaries or at these cuts, and such a group's squared norm is then the sum of its owners' parts (one more hop of
    // the kernel, its fragment table behind the workgroups' records)
    std::vector<int> cuts;
    bool longrun = false;
    for (int r = 0; r < nruns; ++r) {
        const int len = runs[r + 1] - runs[r];
        longrun = longrun || len > SSL;
        for (int k = 0; k < (len > SSL ? len : 1); k += 4) cuts.push_back(runs[r] + k);
    }
    cuts.push_back(q);
    bool ok = false;
    for (int per = 1; per <= 12 && !ok; ++per) ok = symcoop_plan_per(q, gmax, P, cuts.data(), (int)cuts.size() - 1, per);
    if (!ok) return false;
    if (!longrun) return true;
    // per coordinate: (first owner of its group) * 2 + (1: that owner's fragment does not start its slice), the number of owners
    const int G = P.G;
    std::vector<int> c0(G + 1, q);
    for (int g = 0; g < G; ++g) {
        const int *r = P.tab.data() + SPLAN_HEAD + (size_t)g * SW_INTS;
        if (r[SW_NSL] > 0) c0[g] = r[SW_C0];
    }
    for (int g = G - 1; g >= 0; --g) if (c0[g] > c0[g + 1]) c0[g] = c0[g + 1];      // (owners of nothing sit behind the others)
    const size_t base = P.tab.size();
    P.tab.resize(base + 2 * (size_t)q, 0);
    for (int r = 0; r < nruns; ++r) {
        const int gs = runs[r], ge = runs[r + 1];
        const int a = (int)(std::upper_bound(c0.begin(), c0.begin() + G, gs) - c0.begin()) - 1;
        const int b = (int)(std::upper_bound(c0.begin(), c0.begin() + G, ge - 1) - c0.begin()) - 1;
        if (a < 0 || b < a) return false;
        if (b > a && b - a + 1 > P.split) P.split = b - a + 1;
        for (int j = gs; j < ge; ++j) { P.tab[base + 2 * (size_t)j] = 2 * a + (gs == c0[a] ? 0 : 1); P.tab[base + 2 * (size_t)j + 1] = b - a + 1; }
    }
    return true;
}
static bool symcoop_plan_per(int q, int gmax, SymcoopPlan &P, const int *runs, int nruns, int per_want)
{
    P = SymcoopPlan();
    if (q <= 1024 || q > 4096) return false;
    const int T = (q + 63) / 64, ntile = T * (T + 1) / 2;
    int NT = 0;
    for (int nt = 1; nt <= 3; ++nt) if ((ntile + 4 * nt - 1) / (4 * nt) <= gmax) { NT = nt; break; }
    if (!NT) return false;
    const int a = NT == 3 ? 3 : 2;
    int per = 4 * NT;                                                // tiles per workgroup
    if (per_want > 0) {
        // group operators: more workgroups with fewer tiles each (some waves then hold fewer than NT) -- never fewer tiles than give about
        // 18 coordinates per owner
        const int gt = std::min(gmax, (q + 17) / 18), pmin = std::min(4 * NT, std::max(1, (ntile + gt - 1) / gt));
        if (per_want < pmin || per_want > 4 * NT) return false;
        per = per_want;
    }
    const int G = (ntile + per - 1) / per;
    if (G > gmax || q / G < 1 || (q + G - 1) / G > SSL) return false;
    std::vector<int> tiles;                                          // I | J << 8
    for (int r0 = 0; r0 < T;) {                                      // (a short strip comes FIRST, where it holds a handful of tiles: a last strip of one
        int r1 = r0 + ((r0 == 0 && T % a) ? T % a : a);              //  tile row would be 64 tiles of one row -- twelve entries in one block sum)
        if (r1 > T) r1 = T;
        for (int j = 0; j < r1; ++j) for (int i = (j > r0 ? j : r0); i < r1; ++i) tiles.push_back(i | (j << 8));
        r0 = r1;
    }
    std::vector<std::vector<int>> blocks(G), senders(T);
    for (int g = 0; g < G; ++g) {
        std::vector<int> &b = blocks[g];
        for (int k = g * per; k < (g + 1) * per && k < ntile; ++k) { b.push_back(tiles[k] & 0xff); b.push_back(tiles[k] >> 8); }
        std::sort(b.begin(), b.end());
        b.erase(std::unique(b.begin(), b.end()), b.end());
        if ((int)b.size() > SNB) return false;
        for (int B : b) senders[B].push_back(g);
    }
    int maxns = 0, nsum = 0;
    std::vector<int> tab(SPLAN_HEAD + (size_t)G * SW_INTS, 0);
    for (int B = 0; B < T; ++B) { tab[B] = nsum; nsum += (int)senders[B].size(); maxns = std::max(maxns, (int)senders[B].size()); }
    tab[T] = nsum;
    if (maxns > 8 * SE1) return false;
    // the owners' slices: q / G coordinates each -- or, with group operators (runs != null: every group a run of neighbouring
    // coordinates, runs[0 .. nruns] their starts), whole runs, so that a group's norm never needs a value from another owner
    std::vector<int> cut((size_t)G + 1, 0);
    if (runs) {
        // every owner whole runs, 1 .. SSL coordinates: the partition of the nruns runs into G consecutive pieces with the smallest sum of
        // squared piece sizes (dynamic programming over (owners, runs); a piece ends at most SSL coordinates after it starts)
        // (fewer runs than workgroups -- a few large groups: the first K = nruns owners take a run each, the others own nothing: an owner's
        //  coordinates are worked on by eight lanes each, all at once, so an owner with 30 of them is no slower than one with 20)
        const int K = std::min(G, nruns);
        const double INF = 1e300;
        std::vector<double> cost((size_t)(K + 1) * (nruns + 1), INF);
        std::vector<int> from((size_t)(K + 1) * (nruns + 1), -1);
        cost[0] = 0.0;
        for (int g = 1; g <= K; ++g)
            for (int r = g; r <= nruns - (K - g); ++r) {
                double best = INF; int bi = -1;
                for (int r0 = r - 1; r0 >= g - 1 && runs[r] - runs[r0] <= SSL; --r0) {
                    const double c = cost[(size_t)(g - 1) * (nruns + 1) + r0];
                    if (c >= INF) continue;
                    const double len = (double)(runs[r] - runs[r0]), t = c + len * len;
                    if (t < best) { best = t; bi = r0; }
                }
                cost[(size_t)g * (nruns + 1) + r] = best; from[(size_t)g * (nruns + 1) + r] = bi;
            }
        if (cost[(size_t)K * (nruns + 1) + nruns] >= INF) return false;      // (a run longer than an owner holds, or no such partition)
        int r = nruns;
        for (int g = G; g > K; --g) cut[g] = q;
        for (int g = K; g >= 1; --g) { cut[g] = runs[r]; r = from[(size_t)g * (nruns + 1) + r]; }
        cut[0] = 0;
    } else {
        const int base = q / G, rem = q % G;
        for (int g = 0; g <= G; ++g) cut[g] = g * base + (g < rem ? g : rem);
    }
    for (int g = 0; g < G; ++g) {
        int *r = tab.data() + SPLAN_HEAD + (size_t)g * SW_INTS;
        const std::vector<int> &b = blocks[g];
        r[SW_NB] = (int)b.size(); r[SW_C0] = cut[g] < q ? cut[g] : q - 1; r[SW_NSL] = cut[g + 1] - cut[g];      // (an owner of nothing points at a valid coordinate: it reads that one's tags, path_symcoop_kernel)
        if (r[SW_NSL] < (runs ? 0 : 1) || r[SW_NSL] > SSL) return false;
        auto slot = [&](int B) { return (int)(std::lower_bound(b.begin(), b.end(), B) - b.begin()); };
        for (int s = 0; s < (int)b.size(); ++s) {
            r[SW_BLK + s] = b[s];
            r[SW_RANK + s] = (int)(std::lower_bound(senders[b[s]].begin(), senders[b[s]].end(), g) - senders[b[s]].begin());
        }
        for (int wk = 0; wk < 12; ++wk) r[SW_TILE + wk] = -1;
        // wave w holds tiles w, w + 4, w + 8 of the workgroup's list -- three different tile columns: a lasso iterate's non-zeros
        // cluster (config 4: all in block 0), the products of a zero block are skipped, and the busiest WAVE sets the pace
        auto tile_of = [&](int w, int k) { return (k * 4 + w < per) ? g * per + k * 4 + w : ntile; };      // (ntile: no tile in this slot)
        for (int w = 0; w < 4; ++w)
            for (int k = 0; k < NT; ++k) {
                const int idx = tile_of(w, k);
                if (idx >= ntile) continue;
                const int I = tiles[idx] & 0xff, J = tiles[idx] >> 8;
                r[SW_TILE + w * NT + k] = I | (J << 8) | (slot(I) << 16) | (slot(J) << 24);
            }
        int ne = 0;
        for (int wk = 0; wk < 12; ++wk) r[SW_TPOS + wk] = 0xffff;
        for (int s = 0; s < (int)b.size(); ++s) {                    // the rows of Wp of slot s: direct products of tiles in tile row b[s], transposed ones of tile column b[s]
            r[SW_CST + s] = ne;
            for (int wk = 0; wk < 4 * NT; ++wk) {
                const int idx = tile_of(wk / NT, wk % NT);
                if (idx >= ntile) continue;
                const int I = tiles[idx] & 0xff, J = tiles[idx] >> 8;
                if (I == b[s]) r[SW_TPOS + wk] = (r[SW_TPOS + wk] & 0xff00) | ne++;
                if (J == b[s] && I != J) r[SW_TPOS + wk] = (r[SW_TPOS + wk] & 0x00ff) | (ne++ << 8);
            }
            if (ne - r[SW_CST + s] > 8) return false;                // (the block sums read eight rows)
        }
        for (int s = (int)b.size(); s <= SNB; ++s) r[SW_CST + s] = ne;
        if (ne > 8 * NT) return false;
    }
    P.q = q; P.T = T; P.NT = NT; P.G = G; P.nsum = nsum; P.e1n = (maxns + 7) / 8; P.runs = runs != nullptr;
    P.tab.swap(tab);
    return true;
}

// Host-only self-check (include/oemgpu.h: oemgpu_selftest_symcoop_owners): the owners' slices and the fragment table the plan deals for these runs
int symcoop_plan_owners(int q, int gmax, const int *runs, int nruns, int *owner_c0, int *owner_n, int *frag, int *G_out, int *split_out)
{
    SymcoopPlan P;
    if (!symcoop_plan(q, gmax, P, runs, nruns)) return 0;
    for (int g = 0; g < P.G; ++g) {
        const int *r = P.tab.data() + SPLAN_HEAD + (size_t)g * SW_INTS;
        owner_c0[g] = r[SW_C0]; owner_n[g] = r[SW_NSL];
    }
    const size_t base = SPLAN_HEAD + (size_t)P.G * SW_INTS;
    for (int j = 0; j < 2 * q; ++j) frag[j] = P.tab.size() > base ? P.tab[base + j] : (j & 1);
    *G_out = P.G; *split_out = P.split;
    return 1;
}

size_t symcoop_xchg_bytes(const SymcoopPlan &P) { return 2 * ((size_t)P.nsum * 64 * 16) + 2 * ((size_t)P.T * 64 * 16) + 8 * ((size_t)P.G * 16) + 256; }
size_t symcoop_work_bytes(const SymcoopPlan &P) { return (symcoop_xchg_bytes(P) + 255) / 256 * 256 + (sizeof(PathArgs) + 255) / 256 * 256; }      // + the kernel's arguments
// an upper bound for any q the engine takes (workspace reservation): every block receives at most 8 SE1 partials
size_t symcoop_xchg_bytes_max(int q)
{
    if (q <= 1024 || q > 4096) return 0;
    const size_t T = ((size_t)q + 63) / 64;
    return 2 * (T * 8 * SE1 * 64 * 16) + 2 * (T * 64 * 16) + 8 * ((size_t)WCOOP_GMAX * 16) + 1024 + sizeof(PathArgs);
}

// OEM_NO_SYMCOOP=1: the launch-per-iteration engines
bool path_symcoop_eligible(const PathArgs &a, bool group_penalty, bool plan_has_runs)
{
    if (sw().OEM_NO_SYMCOOP.set || sw().OEM_NO_COOP.set) return false;
    if (a.p <= 1024 || a.p > 4096 || a.nbatch > 1 || a.pen_split) return false;
    if (group_penalty && !plan_has_runs) return false;   // (groups that are not runs of <= 32 neighbouring coordinates: the same)
    if ((group_penalty || a.accelerate || a.compute_loss || a.sinv) && sw().OEM_SYMCOOP_NO_GENERAL.set) return false;
    return true;
}

int launch_path_symcoop(hipStream_t s, const PathArgs &a_, const SymcoopPlan &P, const int *plan_dev, void *xchg)
{
    PathArgs a = a_;
    a.lanczos_steps = a.p < SCML ? a.p : SCML;
    OEM_HIP(hipMemsetAsync(xchg, 0, symcoop_xchg_bytes(P), s));                 // the tags must start at 0
    OEM_HIP(hipMemsetAsync(a.d_out, 0, sizeof(double) * D_OUT_LEN, s));        // [6]: only a timed-out workgroup writes it
    unsigned long long *x = reinterpret_cast<unsigned long long *>(xchg);
    const bool gen = a.ngroups > 0 || a.accelerate || a.compute_loss || a.sinv;      // (group tables present: a group penalty is in the call; sinv: the in-place rescale)
#define SX_LAUNCH(NT_, GEN_)                                                                                                     \
    do {                                                                                                                         \
        const size_t sh = symcoop_lds_bytes<NT_, GEN_>();                                                                              \
        if (int rc = lds_limit_once(reinterpret_cast<const void *>(&path_symcoop_kernel<NT_, GEN_>), sh)) return rc;             \
        hipLaunchKernelGGL((path_symcoop_kernel<NT_, GEN_>), dim3(P.G), dim3(SNTH), sh, s, a, plan_dev, x, P.T, P.nsum, P.e1n, P.split);  \
    } while (0)
    switch (P.NT) {
    case 1: if (gen) SX_LAUNCH(1, true); else SX_LAUNCH(1, false); break;
    case 2: if (gen) SX_LAUNCH(2, true); else SX_LAUNCH(2, false); break;
    default: if (gen) SX_LAUNCH(3, true); else SX_LAUNCH(3, false); break;
    }
#undef SX_LAUNCH
    OEM_HIP(hipGetLastError());
    if (sw().OEM_WCOOP_FAKE_TIMEOUT.set) OEM_HIP(hipMemsetAsync(a.d_out + 6, 0xFF, sizeof(double), s));      // tests: the host's fallback
    return 0;
}

}  // namespace oemgpu
