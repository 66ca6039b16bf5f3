// path_coop.hip -- eigenvalue + penalty x lambda path for 288 < q <= 1024: ONE persistent launch of cooperating workgroups.
//
// Replaces, for these sizes, the launch-per-iteration engines of path_large.hip (config 3, p = 512: 873 OEM iterations at
// 5.7 us per launch + ~2 ms of Lanczos launches = 6.0 ms of a 12 ms solve).  Same arithmetic as those engines
// (ref src/oem_base.h:90-110, src/oem_dense.h:485-653, src/utils.cpp:537-549, src/oem_dense.cpp:206-297):
//
//   * XX stays in REGISTERS for the whole call, rows split over W = ceil(q / RW) workgroups, one per CU.  Inside a
//     workgroup the rows are laid out as in path_small.hip's row-split kernel: the 16-lane row group g of a wave holds a
//     slice of CG = 64 columns of 16 rows, multiplies it with v_fmac_f64_dpp row_newbcast (each lane supplies 4 vector
//     entries to its row), and the four slices of a row meet through v_permlane16/32_swap -- no LDS in the product.
//     A wave covers 256 columns; CH = 2 (q <= 512) or 4 (q <= 1024) waves share a row set and combine through LDS.
//   * what crosses workgroups is the PRODUCT: u = d beta - XX beta + XY (or XX v during Lanczos), one all-gather per
//     iteration through the memory side as data-tagged 16-byte pairs {lo, epoch, hi, epoch} (guide recipe R2: the data is the
//     flag, each 8-byte half carries its own tag, no fence; device-scope stores and polls; two buffers by parity).  Every workgroup
//     then thresholds the WHOLE u itself (q / 256 coordinates per thread: group norms, Nesterov step, stop rule, the
//     lambda / penalty state machine), identically everywhere, so no flag or scalar ever crosses workgroups.
//   * u does not depend on lambda: the product that follows convergence at lambda_i is the warm start of lambda_{i+1} and
//     carries the loss of lambda_i (Gram identity).
//   * the eigenvalue step is Lanczos on the same registers with the same exchange (the vector updates and both inner
//     products replicated per workgroup), the top Ritz value by the Sturm multisection of path_dev.hpp.
// Every spin is bounded; a timeout poisons the result (theta = -1) and the host reports it.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "penalty_ops.hpp"
#include "path_dev.hpp"

#ifndef OEM_XCHG_SLEEP
#define OEM_XCHG_SLEEP 12         // s_sleep units (64 cycles) before the first poll sweep of the gather (tools/coop_sleep_ab.sh: config 3 path 2.23 -> 2.17 ms, p = 1024 3.99 -> 3.75)
#endif
#ifndef OEM_XCHG_SLEEP_LOCAL
#define OEM_XCHG_SLEEP_LOCAL 2    // the same inside one XCD, where a pair is there after ~0.25 us (tools/coop_sleep_local_ab.sh: 0 / 2 / 5 within noise, 8 +4 %, 12 +10 %)
#endif
namespace oemgpu {

namespace {

constexpr int CG = 64;            // columns per 16-lane row group, in registers
constexpr int NTH = 256;          // threads per workgroup (one wave per SIMD: 512 registers per lane, 256 of them VGPRs)
constexpr int CML = 256;          // Lanczos steps kept

template <int CH> struct CoopCfg {
    static constexpr int RW = 64 / CH;          // rows per workgroup (16 per row set, 4 / CH row sets)
    static constexpr int QMAX = 256 * CH;       // columns covered: CH waves x 4 groups x CG
    static constexpr int EPT = QMAX / NTH;      // coordinates per thread in the replicated vector work
    // LDS carve (doubles)
    static constexpr int OFF_U = 0;                         // gathered product / u [QMAX + 8]
    static constexpr int OFF_B = OFF_U + QMAX + 8;          // vector going into the product (beta / Lanczos v) [QMAX + 8]
    static constexpr int OFF_F = OFF_B + QMAX + 8;          // group factors [QMAX]
    static constexpr int OFF_GW = OFF_F + QMAX;             // group weights [QMAX]
    static constexpr int OFF_T = OFF_GW + QMAX;             // Lanczos alpha [CML], beta [CML]
    static constexpr int OFF_S = OFF_T + 2 * CML;           // Sturm scratch 2 (CML + 16)
    static constexpr int OFF_R = OFF_S + 2 * (CML + 16);    // block reductions [2][4], theta slot, ...
    static constexpr int OFF_P = OFF_R + 16;                // partial products of the column parts [CH][16 * 4 / CH ... ] = [64]
    static constexpr int OFF_I = OFF_P + 64;                // ints: gid[QMAX], gstart[QMAX + 1], gidx[QMAX], gzero[QMAX]
    static constexpr int N_DBL = OFF_I + (4 * QMAX + 4) / 2 + 2;
};

// -DOEM_PATH_DIAG: cycles of wave 0 of workgroup 0 by segment (fenced stamps: read the SHARES)
#ifdef OEM_PATH_DIAG
__device__ unsigned long long g_diag_coop[16];
#define COOP_STAMP(slot)                                                                   \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        unsigned long long t__;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        X.acc[slot] += t__ - X.last;                                                       \
        X.last = t__;                                                                      \
    } while (0)
#else
#define COOP_STAMP(slot) do { } while (0)
#endif

typedef unsigned coop_v4u __attribute__((ext_vector_type(4)));
struct CoopX {
    __amdgpu_buffer_rsrc_t rs;    // [2 parities][QMAX rows] pairs of 16 bytes: {lo, epoch, hi, epoch}
    unsigned epoch;               // exchange counter, never 0; identical in every workgroup
    int wg, qmax;
    int failed;                   // 0; PATH_FAILED_TIMEOUT: an exchange timed out; PATH_FAILED_ABORT: the host's abort word was seen (common.hpp)
    const int *abortw;            // PathArgs::abort_word
#ifdef OEM_PATH_DIAG
    unsigned long long acc[16], last;
#endif
};

// sum over the 256 threads, identical in every thread and every workgroup (per-wave DPP sums, four words in fixed order)
__device__ __forceinline__ double coop_block_sum(double v, double *red, int &rpar, int w, int lane)
{
    const double s = wave_sum(v);
    double *r = red + 4 * rpar;
    if (lane == 0) r[w] = s;
    __syncthreads();
    const double t = (r[0] + r[1]) + (r[2] + r[3]);
    rpar ^= 1;                    // the next call writes the other half: no second barrier needed
    return t;
}

template <int C> struct CoopFma {
    static __device__ __forceinline__ void run(double (&acc)[4], const double (&B)[CG / 16], const double (&a)[CG])
    {
        if constexpr (C < CG) {
            BcFma<(C & 15)>::fmac(acc[C & 3], B[C >> 4], a[C]);
            CoopFma<C + 1>::run(acc, B, a);
        }
    }
};

// (M vec)[row of this lane] over this wave's 256 columns: every lane of the 16-lane column position gets the sum of the four groups
__device__ __forceinline__ double coop_product(const double (&a)[CG], const double *Bsh, const int (&bidx)[CG / 16])
{
    double B[CG / 16];
#pragma unroll
    for (int j = 0; j < CG / 16; ++j) B[j] = Bsh[bidx[j]];
    asm volatile("s_nop 1" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]));      // LDS/VALU write -> DPP read (path_dev.hpp)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    CoopFma<0>::run(acc, B, a);
    const double s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    // sum over the four row groups: v_permlane16_swap (rows 1, 3 <-> 0, 2), then v_permlane32_swap (rows 2, 3 <-> 0, 1)
    const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
    auto l1 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double t = __hiloint2double((int)h1[0], (int)l1[0]) + __hiloint2double((int)h1[1], (int)l1[1]);
    const unsigned lo2 = (unsigned)__double2loint(t), hi2 = (unsigned)__double2hiint(t);
    auto l2 = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);
    auto h2 = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
    return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}

// One product + all-gather.  In: the vector in Bsh (complete, behind a barrier).  Out: Ush[j] for every j < q, behind a
// barrier: OEM ? (d vec_j - (M vec)_j) + xy_j : (M vec)_j.
// LOCAL: all workgroups of the instance sit on ONE XCD (proved at the start of the kernel), whose L2 is the point of coherence for its
// CUs: plain stores and sc1 loads then make the exchange -- the pair never leaves the XCD (tools/xchg_probe.hip modes 3 / 9: 16
// workgroups, 512 rows, 1.04 -> 0.65 us per all-gather; between XCDs a plain store is never seen)
template <int CH, bool OEM, bool LOCAL>
__device__ __forceinline__ void coop_round(const double (&a)[CG], const int (&bidx)[CG / 16], double *Ush, const double *Bsh, double *Pc,
                                           int q, int row, bool rowok, bool publisher, double d, double xyR, CoopX &X, int w, int lane, int tid)
{
    typedef CoopCfg<CH> C;
    COOP_STAMP(0);                                              // everything between rounds (threshold, stop rule, Lanczos vector work)
    double g = coop_product(a, Bsh, bidx);
    COOP_STAMP(1);                                              // vector reads, FMAs, row-group sum
    if (CH > 1) {                                               // the CH column parts of a row set meet in LDS
        const int part = w % CH, rset = w / CH;                 // waves rset * CH .. rset * CH + CH - 1 share 16 rows
        if (part != 0 && lane < 16) Pc[(rset * CH + part) * 16 + lane] = g;
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int k = 1; k < CH; ++k) g += Pc[(rset * CH + k) * 16 + (lane & 15)];
        }
    }
    COOP_STAMP(2);                                              // column parts through LDS
    ++X.epoch;
    const int base = (int)(X.epoch & 1u) * X.qmax * 16;        // byte offset of this parity's pairs
    if (publisher) {                                            // lanes 0..15 of the first wave of each row set
        const double out = OEM ? (d * Bsh[row] - g) + xyR : g;
        if (rowok) {
            coop_v4u pr;
            pr.x = (unsigned)__double2loint(out); pr.y = X.epoch; pr.z = (unsigned)__double2hiint(out); pr.w = X.epoch;
            __builtin_amdgcn_raw_buffer_store_b128(pr, X.rs, base + row * 16, 0, LOCAL ? 0 : 16);       // aux 16: sc1 (device scope)
            Ush[row] = out;
        }
    }
    COOP_STAMP(3);                                              // publish
    // Gather the rows of the other workgroups: thread t polls rows t, t + 256, ... -- a row is ONE 16-byte load (each 8-byte half
    // carries its own tag, so a torn pair is only ever seen as "not there yet"), ONE sweep in flight, and a row that has arrived
    // is not asked for again.  (Round 2 kept three sweeps of 8-byte atomic loads in flight: tools/xchg_probe.hip shows that the
    // polls then flood the fabric and every exchange gets SLOWER -- 16 workgroups, 512 rows: 1.46 us per all-gather against 1.15.)
    coop_v4u pv[C::EPT];
    unsigned miss = 0;
#pragma unroll
    for (int k = 0; k < C::EPT; ++k) {
        const int j = tid + NTH * k;
        if (j < q && j / C::RW != X.wg) miss |= 1u << k;
        pv[k] = coop_v4u{0u, 0u, 0u, 0u};
    }
    const unsigned need = miss;
    if (OEM_XCHG_SLEEP > 0) __builtin_amdgcn_s_sleep(LOCAL ? OEM_XCHG_SLEEP_LOCAL : OEM_XCHG_SLEEP);      // (nothing of this epoch can have landed yet: path_wcoop.hip, wc_gather)
    // ONE counter in the sweep loop: it runs out once per 1,024 sweeps (~1 ms), and only then are the abort word and the timeout looked at
    // (~1 s = 1,000 such rounds: a partner is gone; after one timeout -- or the abort word -- nobody waits again: one sweep each)
    unsigned left = X.failed ? 1u : PATH_ABORT_SPINS, rounds = 0u;
    bool ok = true;
    while (__any(miss != 0u)) {
#pragma unroll
        for (int k = 0; k < C::EPT; ++k)
            if ((miss >> k) & 1u) pv[k] = __builtin_amdgcn_raw_buffer_load_b128(X.rs, base + (tid + NTH * k) * 16, 0, 16);
#pragma unroll
        for (int k = 0; k < C::EPT; ++k)
            if (((miss >> k) & 1u) && pv[k].y == X.epoch && pv[k].w == X.epoch) miss &= ~(1u << k);
        if (--left == 0u && __any(miss != 0u)) {
            if (X.failed || ++rounds >= PATH_TIMEOUT_ROUNDS) { ok = false; break; }
            if (path_abort_asked(X.abortw)) { X.failed = PATH_FAILED_ABORT; break; }
            left = PATH_ABORT_SPINS;
        }
    }
    if (!ok && X.failed == 0) X.failed = PATH_FAILED_TIMEOUT;
    COOP_STAMP(4);                                              // polling
#ifdef OEM_PATH_DIAG
    X.acc[8] += 1;
#endif
#pragma unroll
    for (int k = 0; k < C::EPT; ++k)
        if ((need >> k) & 1u) Ush[tid + NTH * k] = ((miss >> k) & 1u) ? 0.0 : __hiloint2double((int)pv[k].z, (int)pv[k].x);
    __syncthreads();
    COOP_STAMP(5);                                              // LDS stores + barrier (waits for the slowest wave's poll)
}

// element-wise operators (ref src/oem_dense.h:76-149), branch-free; only the reciprocals the operator uses
struct ThrC { double D, rD, gammad, dmg, rdmg, gm1, gamma, dsc, rdsc, d, rd; };
template <int KIND> __device__ __forceinline__ ThrC thr_c(const PenK &K, double d)
{
    ThrC c = {};
    c.D = K.D; c.gammad = K.gamma * K.D; c.gamma = K.gamma; c.gm1 = K.gamma - 1.0; c.d = d;
    if (KIND != K_OLS) c.rD = 1.0 / K.D;
    if (KIND == K_MCP) { c.dmg = K.D - 1.0 / K.gamma; c.rdmg = 1.0 / c.dmg; }
    if (KIND == K_SCAD) { c.dsc = c.gm1 * K.D - 1.0; c.rdsc = 1.0 / c.dsc; }
    if (KIND == K_OLS) c.rd = 1.0 / d;
    return c;
}
template <int KIND> __device__ __forceinline__ double thr1(double u, double tp, const ThrC &c)
{
    if (KIND == K_SOFT) return cdiv(shrink(u, tp), c.D, c.rD);
    if (KIND == K_MCP) {
        const bool big = fabs(u) > c.gammad * tp;
        return cdiv(big ? u : shrink(u, tp), big ? c.D : c.dmg, big ? c.rD : c.rdmg);
    }
    if (KIND == K_SCAD) {
        const double au = fabs(u);
        const bool big = au > c.gammad * tp, mid = !big && au > (c.D + 1.0) * tp;
        const double num = big ? u : (mid ? shrink(c.gm1 * u, c.gamma * tp) : shrink(u, tp));
        return cdiv(num, mid ? c.dsc : c.D, mid ? c.rdsc : c.rD);
    }
    return cdiv(u, c.d, c.rd);
}

template <int CH, bool LOCAL>
__global__ __launch_bounds__(NTH) void path_coop_kernel(PathArgs A_)
{
    // LOCAL: eight times the workgroups were launched; workgroup ids go round the XCDs, so those with the same id mod 8 share one --
    // instance y takes the ones of XCD (xcd_base + y) mod 8, the others leave at once
    if (LOCAL && (int)(blockIdx.x & 7u) != ((A_.xcd_base + (int)blockIdx.y) & 7)) return;
    const PathArgs A = path_instance(A_);
    typedef CoopCfg<CH> C;
    constexpr int EPT = C::EPT;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15, grp = lane >> 4;
    const int q = A.p, wg = LOCAL ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const unsigned long long t_cyc0 = __builtin_amdgcn_s_memtime(), t_rt0 = __builtin_amdgcn_s_memrealtime();
    double *Ush = lds + C::OFF_U, *Bsh = lds + C::OFF_B, *F = lds + C::OFF_F, *GW = lds + C::OFF_GW;
    double *Tal = lds + C::OFF_T, *Tbe = Tal + CML, *red = lds + C::OFF_R, *Pc = lds + C::OFF_P;
    int *gid = reinterpret_cast<int *>(lds + C::OFF_I), *gstart = gid + C::QMAX, *gidx = gstart + C::QMAX + 1, *gzero = gidx + C::QMAX;
    const bool writer = wg == 0;

    // ---- this lane's part of the matrix: row set w / CH (16 rows), column part w % CH, group slice grp
    const int part = w % CH, rset = w / CH;
    const int row = wg * C::RW + rset * 16 + l16;
    const bool rowok = rset * 16 + l16 < C::RW && row < q;
    const int cbase = part * 256 + grp * CG;
    double a[CG];
#pragma unroll
    for (int k = 0; k < CG; ++k) {
        const int col = cbase + k;
        a[k] = (rowok && col < q) ? A.xx[(size_t)col * q + row] : 0.0;
    }
    int bidx[CG / 16];
#pragma unroll
    for (int j = 0; j < CG / 16; ++j) {
        const int col = cbase + 16 * j + l16;
        bidx[j] = col < q ? col : C::QMAX + (lane & 7);         // a zero word behind the vector
    }
    const bool publisher = part == 0 && lane < 16;
    const double xyR = rowok ? A.xy[row] : 0.0;
    // ---- this thread's coordinates of the replicated vector work
    double xyE[EPT], pfE[EPT], sinvE[EPT];
    bool valid[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int j = tid + NTH * k;
        valid[k] = j < q;
        xyE[k] = valid[k] ? A.xy[j] : 0.0;
        pfE[k] = valid[k] ? A.pf[j] : 0.0;
        sinvE[k] = (valid[k] && A.sinv) ? A.sinv[j] : 1.0;
    }
    const int ng = A.ngroups;
    for (int j = tid; j < C::QMAX + 8; j += NTH) { Ush[j] = 0.0; Bsh[j] = 0.0; }
    for (int j = tid; j < q; j += NTH) gid[j] = ng > 0 ? A.gid[j] : -1;
    if (ng > 0) {
        for (int g = tid; g <= ng; g += NTH) gstart[g] = A.gstart[g];
        for (int g = tid; g < ng; g += NTH) { gzero[g] = A.gzero[g]; GW[g] = A.gw[g]; }
        const int nm = A.gstart[ng];
        for (int m = tid; m < nm; m += NTH) gidx[m] = A.gidx[m];
    }
    __syncthreads();
    // this thread's group (tid < ng; groups beyond 256 take the LDS loop) with its first eight members, and the groups of its coordinates
    int gm[8], gcnt = 0, gme = 0;
    bool gz = true;
    double gwt = 0.0;
    if (tid < ng) {
        const int m0 = gstart[tid];
        gme = gstart[tid + 1]; gcnt = gme - m0; gz = gzero[tid] != 0; gwt = GW[tid];
#pragma unroll
        for (int k = 0; k < 8; ++k) gm[k] = (k < gcnt) ? gidx[m0 + k] : C::QMAX + k;       // zero words behind the vector
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) gm[k] = C::QMAX + k;
    }
    int gidE[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) gidE[k] = valid[k] ? gid[tid + NTH * k] : -1;
    // Groups that are aligned blocks of eight consecutive coordinates (config 3: rep(1:64, each = 8)) sit in eight neighbouring
    // lanes of one register (coordinate j = tid + 256 k), so their norms are three DPP butterfly stages and every lane forms its
    // group's factor itself: no LDS list walk, no factor exchange, no barrier (1,340 of the 7,000 cycles of a config 3 round).
    bool aligned8 = false;
    double gwE[EPT];
    bool gzE[EPT];
    {
        bool mine = ng > 0 && (q & 7) == 0 && ng * 8 == q;
#pragma unroll
        for (int k = 0; k < EPT; ++k) if (valid[k]) mine = mine && gidE[k] == ((tid + NTH * k) >> 3);
        if (mine) for (int g = tid; g < ng; g += NTH) mine = mine && (gstart[g + 1] - gstart[g] == 8);
#ifdef OEM_COOP_NO_ALIGNED                                   // timing experiments: the general group stage for every layout
        mine = false;
#endif
        aligned8 = __syncthreads_and(mine ? 1 : 0) != 0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            gwE[k] = (aligned8 && gidE[k] >= 0) ? GW[gidE[k]] : 0.0;
            gzE[k] = (aligned8 && gidE[k] >= 0) ? gzero[gidE[k]] != 0 : true;
        }
    }
    CoopX X;
    X.rs = __builtin_amdgcn_make_buffer_rsrc((void *)A.work, 0, 2 * C::QMAX * 16 + (LOCAL ? 1024 : 0), 0x00020000); X.epoch = 0; X.wg = wg; X.qmax = C::QMAX; X.failed = 0; X.abortw = A.abort_word;
    if (LOCAL) {
        // The proof of placement, before anything relies on it: every workgroup publishes the XCD it runs on (device scope: this
        // exchange has to work ACROSS XCDs) behind the pairs, reads everybody's and compares.  All workgroups reach the same verdict
        // (somebody differs or nobody does), so on a mismatch they all leave without waiting for each other; the host then makes
        // the call again with the exchange at device scope.
        constexpr unsigned MAGIC = 0x58434431u;                  // "XCD1"
        const int W = (q + C::RW - 1) / C::RW, xoff = 2 * C::QMAX * 16;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc = (xcc & 0xfu) + 1u;
        if (A.one_xcd == 2 && wg == 1) xcc += 1u;                 // tests (OEM_FAKE_XCD_MISMATCH): a placement that is not what the launch assumed
        if (tid == 0) {
            coop_v4u pr; pr.x = xcc; pr.y = MAGIC; pr.z = xcc; pr.w = MAGIC;
            __builtin_amdgcn_raw_buffer_store_b128(pr, X.rs, xoff + wg * 16, 0, 16);
        }
        coop_v4u got = coop_v4u{xcc, MAGIC, xcc, MAGIC};
        bool there = tid >= W;
        unsigned spin_left = PATH_ABORT_SPINS, spin_rounds = 0u;
        bool timed_out = false;
        while (__any(!there)) {
            if (!there) {
                got = __builtin_amdgcn_raw_buffer_load_b128(X.rs, xoff + tid * 16, 0, 16);
                there = got.y == MAGIC && got.w == MAGIC && got.x == got.z;
            }
            if (--spin_left == 0u && __any(!there)) {
                if (++spin_rounds >= PATH_TIMEOUT_ROUNDS || path_abort_asked(X.abortw)) { timed_out = true; break; }
                spin_left = PATH_ABORT_SPINS;
            }
        }
        const bool lost = __syncthreads_or(timed_out ? 1 : 0) != 0;                                   // a partner never came: the usual poison
        const bool apart = __syncthreads_or((there && tid < W && got.x != xcc) ? 1 : 0) != 0;        // not one XCD
        if (lost || apart) {
            if (tid == 0) A.d_out[6] = lost ? 3.0 : 2.0;       // (3: a partner never came to the PROOF -- the instance was not co-resident on its XCD; 2: not one XCD)
            return;
        }
    }
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;
    X.last = __builtin_amdgcn_s_memtime();
#endif
    int rpar = 0;

    // ---- eigenvalue step: Lanczos on XX, the vector updates replicated per workgroup
    double v[EPT], vp[EPT], wv[EPT];
    {
        double nn = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const unsigned j = tid + NTH * k;
            const unsigned h = j * 2654435761u + 12345u;                 // deterministic non-structured start
            v[k] = valid[k] ? ((double)(h >> 8) * (1.0 / 16777216.0) - 0.5) : 0.0;
            vp[k] = 0.0;
            nn = fma(v[k], v[k], nn);
        }
        nn = 1.0 / sqrt(coop_block_sum(nn, red, rpar, w, lane));
#pragma unroll
        for (int k = 0; k < EPT; ++k) v[k] *= nn;
    }
    int msteps = A.lanczos_steps > CML ? CML : A.lanczos_steps;
    if (q < CML) msteps = msteps < q ? msteps : q;
    if (msteps < 1) msteps = 1;
    double *theta_slot = red + 8;
    auto top_ritz = [&](int m, double hint) {
        if (w == 0) {
            const double th = tridiag_max(Tal, Tbe, m, lane, lds + C::OFF_S, hint);
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
    for (int j = 0; j < msteps; ++j) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) if (valid[k]) Bsh[tid + NTH * k] = v[k];
        __syncthreads();
        coop_round<CH, false, LOCAL>(a, bidx, Ush, Bsh, Pc, q, row, rowok, publisher, 0.0, 0.0, X, w, lane, tid);
        double al = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) { wv[k] = valid[k] ? Ush[tid + NTH * k] : 0.0; al = fma(v[k], wv[k], al); }
        al = coop_block_sum(al, red, rpar, w, lane);
        double bb = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            wv[k] = (wv[k] - al * v[k]) - bprev * vp[k];
            bb = fma(wv[k], wv[k], bb);
        }
        double ib;
        sqrt_rsqrt(coop_block_sum(bb, red, rpar, w, lane), bb, ib);
        if (tid == 0) { Tal[j] = al; Tbe[j] = bb; }
        nst = j + 1;
        if (!(bb > 1e-13 * fabs(al))) break;                        // invariant subspace reached: T is exact
        if (lanczos_check_due(nst) && nst < msteps) {
            const double th = top_ritz(nst, theta_prev);            // its first barrier publishes Tal / Tbe
            if (lanczos_converged(th, theta_prev, mv_prev)) { theta = th; have_theta = true; break; }
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) { vp[k] = v[k]; v[k] = wv[k] * ib; }
        bprev = bb;
    }
    if (!have_theta) { __syncthreads(); theta = top_ritz(nst, theta_prev); }
    const double d = theta * 1.005;                                  // ref src/oem_dense.h:498
    if (tid == 0 && writer) { A.d_out[0] = d; A.d_out[1] = theta; A.d_out[4] = (double)nst; A.d_out[5] = (!have_theta && nst >= msteps && nst < q) ? 1.0 : 0.0; }
#ifdef OEM_PATH_DIAG
    COOP_STAMP(0);
    if (tid == 0 && writer) { for (int k = 0; k < 6; ++k) { g_diag_coop[k] = X.acc[k]; } g_diag_coop[6] = X.acc[8]; g_diag_coop[7] = (unsigned long long)nst; }
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;
#endif

    // ---- lambda grid constants (ref src/oem_dense.cpp:175-192)
    const double scaley = A.yscale ? A.stats[1] : 1.0;
    const double yy = A.stats[2], nobs = A.stats[3];
    const int nl = A.nl;
    double lmax = 0.0;
    {
        double m = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int j = tid + NTH * k;
            const double xl = (A.lmax_xy && valid[k]) ? A.lmax_xy[j] : xyE[k];
            m = fmax(m, (valid[k] && j >= A.lmax_from) ? fabs(xl) : 0.0);
        }
        m = wave_max(m);
        if (lane == 0) red[12 + w] = m;
        __syncthreads();
        lmax = fmax(fmax(red[12], red[13]), fmax(red[14], red[15])) * scaley;
    }
    const double llo = log(lmax), lhi = log(A.lambda_min_ratio * lmax);
    const double lstep = nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0;
    const bool lflip = fabs(lhi) < fabs(llo);
    const double tol = A.tol;

    int *flagw = reinterpret_cast<int *>(red + 10);                 // "some coefficient still moving", one word per wave
    // One lambda loop per (operator family, accelerate) pair: the dispatch happens once per penalty, the serial loop carries only
    // the arithmetic of the operator in use.  KIND == K_GRP covers every group operator (K.kind selects inside).
    bool left = false;                                           // the host's abort word was seen (PathArgs::abort_word): every loop is left
    unsigned tick = 0u;
    auto lambda_loop = [&](auto KIND_, auto ACC_, int pp, int pen) {
        constexpr int KIND = decltype(KIND_)::value;
        constexpr bool ACC = decltype(ACC_)::value;
        const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
        const bool isnet = pen_is_net(pen);
        const int maxit = A.maxit;
        const bool want_loss = A.compute_loss != 0, has_sinv = A.sinv != nullptr;
        double beta[EPT], bold[EPT];
        // cold start (ref src/oem_dense.cpp:243-244): beta = 0, so u = XY
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            beta[k] = 0.0;
            if (valid[k]) { Ush[tid + NTH * k] = xyE[k]; Bsh[tid + NTH * k] = 0.0; }
        }
        __syncthreads();
        double ak = 1.0;
        double lam_next = A.user_lambda ? A.lambda_user[(size_t)pp * nl] : 0.0;
        for (int i = 0; i < nl; ++i) {
            double lam;
            if (A.user_lambda) {
                lam = lam_next;
                if (i + 1 < nl) lam_next = A.lambda_user[(size_t)pp * nl + i + 1];
            } else {
                double lv;
                if (nl == 1) lv = lhi;
                else if (lflip) lv = (i == 0) ? llo : lhi - (double)(nl - 1 - i) * lstep;
                else lv = (i == nl - 1) ? lhi : llo + (double)i * lstep;
                lam = exp(lv);
                if (isnet) lam = lam / A.alpha;
            }
            const size_t orow = (size_t)pp * nl + i;
            if (tid == 0 && writer) A.lambda_out[orow] = lam;
            if (i >= nlam) continue;
            const PenK K = pen_consts(pen, lam / scaley, d, A.alpha, A.gamma, A.tau);       // ref src/oem_dense.cpp:241
            const ThrC c = thr_c<KIND>(K, d);
            double tp[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) tp[k] = pfE[k] * K.L;
            int it = 0;
            for (;;) {
                if ((tick++ & 127u) == 0u && path_abort_asked(X.abortw)) X.failed = PATH_FAILED_ABORT;
                // ---- beta = T(u) for every coordinate (replicated), acceleration, stop rule
                double u[EPT];
#pragma unroll
                for (int k = 0; k < EPT; ++k) { bold[k] = beta[k]; u[k] = valid[k] ? Ush[tid + NTH * k] : 0.0; }
                double fE[EPT];
                if constexpr (KIND == K_GRP) {                      // group operators, ref src/oem_dense.h:193-315
                    // factor of a group from its squared norm: 1 - pen / ||u_g|| with the root and its reciprocal from v_rsq_f64 +
                    // Goldschmidt (~10 dependent FP64 ops; sqrt followed by a division is ~60), the quotient refined like cdiv;
                    // ||u_g|| = 0 => f = 0 (quirk Q6); unpenalised groups keep f = 1
                    auto factor = [&](double s2, double pen_g) {
                        if (K.kind == K_GRP || K.kind == K_SGL) {
                            double nrm, rn;
                            sqrt_rsqrt_lane(s2, nrm, rn);
                            const double t = 1.0 - cdiv(pen_g, nrm, rn);
                            return (s2 > 0.0 && 0.0 < t) ? t : 0.0;
                        }
                        const double nr = sqrt(s2);
                        return (K.kind == K_GRP_MCP) ? mcp_norm(nr, pen_g, K.D, K.gamma) : scad_norm(nr, pen_g, K.D, K.gamma);
                    };
                    if (aligned8) {
                        double s2[EPT];
#pragma unroll
                        for (int k = 0; k < EPT; ++k) {
                            if (K.kind == K_SGL) u[k] = soft1(u[k], pfE[k] * K.L1, 1.0);    // the soft-thresholded u feeds the norms
                            s2[k] = u[k] * u[k];
                        }
#pragma unroll
                        for (int k = 0; k < EPT; ++k) s2[k] += dpp_mov<0xB1, 0xf>(s2[k], 0.0);      // quad_perm [1,0,3,2]
#pragma unroll
                        for (int k = 0; k < EPT; ++k) s2[k] += dpp_mov<0x4E, 0xf>(s2[k], 0.0);      // quad_perm [2,3,0,1]
#pragma unroll
                        for (int k = 0; k < EPT; ++k) s2[k] += dpp_mov<0x141, 0xf>(s2[k], 0.0);     // row_half_mirror: the eight lanes of the group
                        bool plain = K.kind == K_GRP || K.kind == K_SGL;
#pragma unroll
                        for (int k = 0; k < EPT; ++k) plain = plain && ((s2[k] > 1e-200 && s2[k] < 1e200) || s2[k] == 0.0);
                        if (__builtin_expect(__all(plain), 1)) {
                            // the EPT chains side by side, no branch inside: y = rsq(s2), two coupled Goldschmidt steps, one correction of
                            // the root (path_dev.hpp: sqrt_rsqrt), then t = 1 - pen / nrm through the corrected reciprocal (cdiv)
                            double g[EPT], h[EPT], e[EPT];
#pragma unroll
                            for (int k = 0; k < EPT; ++k) { const double x = s2[k] == 0.0 ? 1.0 : s2[k]; const double y = __builtin_amdgcn_rsq(x); g[k] = x * y; h[k] = 0.5 * y; }
#pragma unroll
                            for (int it2 = 0; it2 < 2; ++it2) {
#pragma unroll
                                for (int k = 0; k < EPT; ++k) e[k] = fma(-h[k], g[k], 0.5);
#pragma unroll
                                for (int k = 0; k < EPT; ++k) { g[k] = fma(g[k], e[k], g[k]); h[k] = fma(h[k], e[k], h[k]); }
                            }
#pragma unroll
                            for (int k = 0; k < EPT; ++k) {
                                const double x = s2[k] == 0.0 ? 1.0 : s2[k];
                                e[k] = fma(-g[k], g[k], x);
                                g[k] = fma(e[k], h[k], g[k]);                               // nrm
                            }
#pragma unroll
                            for (int k = 0; k < EPT; ++k) {
                                const double t = 1.0 - cdiv(K.L * gwE[k], g[k], 2.0 * h[k]);
                                fE[k] = gzE[k] ? 1.0 : ((s2[k] > 0.0 && 0.0 < t) ? t : 0.0);
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < EPT; ++k) fE[k] = gzE[k] ? 1.0 : factor(s2[k], K.L * gwE[k]);
                        }
                    } else {
                    if (K.kind == K_SGL) {                          // sparse group lasso: the soft-thresholded u feeds the norms
                        __syncthreads();                            // everybody has read u before it is overwritten
#pragma unroll
                        for (int k = 0; k < EPT; ++k) {
                            u[k] = soft1(u[k], pfE[k] * K.L1, 1.0);
                            if (valid[k]) Ush[tid + NTH * k] = u[k];
                        }
                        __syncthreads();
                    }
                    for (int gi = tid; gi < ng; gi += NTH) {
                        double f = 1.0;
                        const bool first = gi == tid;               // this thread's cached group
                        if (first ? !gz : !gzero[gi]) {
                            // summed in member order like the reference; the first eight member indices sit in registers (eight
                            // independent LDS reads: one latency), longer groups and groups beyond the 256th walk the LDS lists
                            double s2 = 0.0;
                            int m, me;
                            if (first) {
                                double x[8];
#pragma unroll
                                for (int k = 0; k < 8; ++k) x[k] = Ush[gm[k]];              // padding slots read zero words
#pragma unroll
                                for (int k = 0; k < 8; ++k) s2 += x[k] * x[k];
                                m = gme - gcnt + 8; me = gme;
                            } else { m = gstart[gi]; me = gstart[gi + 1]; }
                            for (; m < me; ++m) { const double x = Ush[gidx[m]]; s2 += x * x; }
                            f = factor(s2, K.L * (first ? gwt : GW[gi]));
                        }
                        F[gi] = f;
                    }
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < EPT; ++k) fE[k] = gidE[k] >= 0 ? F[gidE[k]] : 0.0;
                    }
                }
                COOP_STAMP(9);                                      // u read (+ group norms and factors)
                bool bad = false;
                double adp = 0.0, akn = 1.0, ratio = 0.0;
                if (ACC) { akn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * ak * ak)); ratio = (ak - 1.0) / akn; }     // ref src/oem_dense.h:633-651
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    const int j = tid + NTH * k;
                    double bn;
                    if constexpr (KIND == K_GRP) {
                        const double f = fE[k];
                        bn = (f != 0.0) ? cdiv(u[k] * f, K.D, c.rD) : 0.0;
                    } else bn = thr1<KIND>(u[k], tp[k], c);
                    bn = valid[k] ? bn : 0.0;
                    if (ACC) {
                        const double upd = bn, diff = upd - bold[k];
                        bn = upd + ratio * diff;
                        adp += (bn - upd) * diff;
                    }
                    const double cu = fabs(bn), qo = fabs(bold[k]);
                    const bool cn = cu > 1e-13, qn = qo > 1e-13;    // ref src/utils.cpp:537-549
                    bad |= (cn != qn) | (cn & qn & (fabs(bn - bold[k]) > tol * qo));
                    beta[k] = bn;
                    if (valid[k]) Bsh[j] = bn;
                }
                if (ACC) {
                    adp = coop_block_sum(adp, red, rpar, w, lane);
                    ak = (adp > 0.0) ? 1.0 : akn;
                }
                COOP_STAMP(10);                                     // threshold, stop rule
                // one barrier: Bsh is complete, and every wave's "still moving" word is in place
                const int mine = ((__ballot(bad) != 0ull) ? 1 : 0) | (X.failed == PATH_FAILED_ABORT ? 2 : 0);      // (bit 1: the abort word was seen)
                if (lane == 0) flagw[w] = mine;
                __syncthreads();
                const int anybad = flagw[0] | flagw[1] | flagw[2] | flagw[3];
                COOP_STAMP(11);                                     // the barrier
                ++it;
                if (anybad & 2) { left = true; break; }
                const bool conv = !(anybad & 1);
                const bool fin = conv || it >= maxit;
                if (__builtin_expect(fin, 0)) {
                    if (has_sinv) {                                 // oemXTX::get_beta rescales the member in place (quirk Q5)
#pragma unroll
                        for (int k = 0; k < EPT; ++k) { beta[k] *= sinvE[k]; if (valid[k]) Bsh[tid + NTH * k] = beta[k]; }
                        __syncthreads();
                    }
                    if (writer) {
#pragma unroll
                        for (int k = 0; k < EPT; ++k) if (valid[k]) A.beta[orow * q + tid + NTH * k] = beta[k];
                        if (tid == 0) {
                            A.niter[orow] = conv ? it : maxit + 1;                      // ref src/oem_base.h:94-109
                            if (!want_loss) A.loss[orow] = 1e99;
                        }
                    }
                }
                // ---- u = d beta - XX beta + XY: the next iteration's input, or the warm start of the next lambda
                coop_round<CH, true, LOCAL>(a, bidx, Ush, Bsh, Pc, q, row, rowok, publisher, d, xyR, X, w, lane, tid);
                if (__builtin_expect(fin, 0)) {
                    if (want_loss) {
                        // sum (Y - X beta)^2 through the Gram identity (ref src/oem_dense.h:759-770): XX beta = d beta - u + XY
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < EPT; ++k) {
                            const double uu = valid[k] ? Ush[tid + NTH * k] : 0.0;
                            const double gg = (d * beta[k] - uu) + xyE[k];
                            t += beta[k] * (gg - 2.0 * xyE[k]);
                        }
                        t = coop_block_sum(t, red, rpar, w, lane);
                        if (tid == 0 && writer) A.loss[orow] = yy + nobs * t;
                    }
                    break;
                }
            }
            if (left) break;
        }
    };
    for (int pp = A.pen_lo; pp < A.pen_hi; ++pp) {
        const int pen = A.penalty[pp];
        const int kind = pen_consts(pen, 1.0, d, A.alpha, A.gamma, A.tau).kind;
        using T = std::true_type; using Fa = std::false_type;
        if (A.accelerate) {
            switch (kind) {
            case K_SOFT: lambda_loop(std::integral_constant<int, K_SOFT>{}, T{}, pp, pen); break;
            case K_MCP: lambda_loop(std::integral_constant<int, K_MCP>{}, T{}, pp, pen); break;
            case K_SCAD: lambda_loop(std::integral_constant<int, K_SCAD>{}, T{}, pp, pen); break;
            case K_OLS: lambda_loop(std::integral_constant<int, K_OLS>{}, T{}, pp, pen); break;
            default: lambda_loop(std::integral_constant<int, K_GRP>{}, T{}, pp, pen); break;
            }
        } else {
            switch (kind) {
            case K_SOFT: lambda_loop(std::integral_constant<int, K_SOFT>{}, Fa{}, pp, pen); break;
            case K_MCP: lambda_loop(std::integral_constant<int, K_MCP>{}, Fa{}, pp, pen); break;
            case K_SCAD: lambda_loop(std::integral_constant<int, K_SCAD>{}, Fa{}, pp, pen); break;
            case K_OLS: lambda_loop(std::integral_constant<int, K_OLS>{}, Fa{}, pp, pen); break;
            default: lambda_loop(std::integral_constant<int, K_GRP>{}, Fa{}, pp, pen); break;
            }
        }
        if (left) break;
    }
#ifdef OEM_PATH_DIAG
    if (tid == 0 && writer) {
        for (int k = 0; k < 6; ++k) g_diag_coop[8 + k] = X.acc[k];
        g_diag_coop[14] = X.acc[8];
        g_diag_coop[0] = X.acc[9]; g_diag_coop[1] = X.acc[10]; g_diag_coop[2] = X.acc[11];      // (overwrites the Lanczos slots 0..2)
    }
#endif
    if (tid == 0 && writer) {
        A.d_out[2] = (double)(__builtin_amdgcn_s_memtime() - t_cyc0);
        A.d_out[3] = (double)(__builtin_amdgcn_s_memrealtime() - t_rt0);
    }
    // exchange timeout: poison (the host turns it into an error).  X.failed is per-wave state: combined over the workgroup, and
    // stored to a slot of its own that the host cleared before the launch (d_out[6]; every failing workgroup stores the same 1)
    if (__syncthreads_or(X.failed ? 1 : 0) && tid == 0) A.d_out[6] = 1.0;
}

}  // namespace

#ifdef OEM_PATH_DIAG
extern "C" __attribute__((visibility("default"))) int oemgpu_diag_read_coop(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_coop), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

size_t path_coop_xchg_bytes() { return (size_t)2 * 1024 * 2 * sizeof(unsigned long long); }

// Smallest q the cooperating engine takes.  It used to start where one workgroup's registers end (289); measured against the
// engines below (tools/coop_small_time.py, eigen + path of a 100-lambda path): element-wise penalties at q = 209 1.86 vs 2.70 ms
// (the four-workgroup replicated kernel), q = 288 2.14 vs 4.34 ms, config 5's q = 257 1.59 vs 2.96 ms -- but q = 200 1.80 vs
// 0.94 ms (the row-split kernel, up to 208, stays); with a group penalty in the call q = 192 2.25 vs 2.71 ms on the replicated kernel,
// but 1.83 ms on the row-split kernel now that it has the group operators: the same bound for both.
int path_coop_min_q(bool has_groups)
{
    return has_groups ? COOP_MIN_Q_GROUPS : COOP_MIN_Q;
}

bool path_coop_eligible(int q, bool has_sinv, bool compute_loss, int ngroups, int nbatch)
{
    if (sw().OEM_NO_COOP.set || q < path_coop_min_q(ngroups > 0) || q > 1024) return false;      // OEM_NO_COOP: the launch-per-iteration engines
    if (has_sinv && compute_loss) return false;          // the loss of the un-rescaled member would need a product of its own
    if (ngroups > (q <= 512 ? 512 : 1024)) return false;
    (void)nbatch;                                        // nbatch > 1: the caller checks that all workgroup sets fit (api.hip: run_paths)
    return true;
}

int path_coop_workgroups(int q) { return q <= 512 ? (q + 31) / 32 : (q + 15) / 16; }

int launch_path_coop(hipStream_t s, const PathArgs &a_)
{
    PathArgs a = a_;
    const int q = a.p;
    a.lanczos_steps = q < CML ? q : CML;
    const int ninst = (a.nbatch > 1 ? a.nbatch : 1) * (a.pen_split ? a.npen : 1);
    if (ninst > 1 && (size_t)a.bs_work * 8 < path_coop_xchg_bytes()) { set_error("internal: coop work stride"); return OEMGPU_ERR_INTERNAL; }
    OEM_HIP(hipMemsetAsync(a.work, 0, (ninst > 1 ? (size_t)a.bs_work * 8 : path_coop_xchg_bytes()) * ninst, s));     // granule tags must start at 0
    {   // the poison slot of every instance (common.hpp: d_out[6]) starts at 0; only a timed-out workgroup writes it
        const int nb = a.nbatch > 1 ? a.nbatch : 1;
        OEM_HIP(hipMemset2DAsync(a.d_out + 6, nb > 1 ? (size_t)a.bs_out : sizeof(double), 0, sizeof(double), (size_t)nb, s));
    }
    if (q <= 512) {
        typedef CoopCfg<2> C;
        const int W = (q + C::RW - 1) / C::RW;
        const size_t sh = (size_t)C::N_DBL * sizeof(double);
        if (a.one_xcd) {
            if (sh > 64 * 1024) OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&path_coop_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
            hipLaunchKernelGGL((path_coop_kernel<2, true>), dim3(8 * W, ninst), dim3(NTH), sh, s, a);
        } else {
            if (sh > 64 * 1024) OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&path_coop_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
            hipLaunchKernelGGL((path_coop_kernel<2, false>), dim3(W, ninst), dim3(NTH), sh, s, a);
        }
    } else {
        typedef CoopCfg<4> C;
        const int W = (q + C::RW - 1) / C::RW;
        const size_t sh = (size_t)C::N_DBL * sizeof(double);
        if (a.one_xcd) { set_error("internal: the one-XCD form of the cooperating engine stops at q = 512"); return OEMGPU_ERR_INTERNAL; }
        if (sh > 64 * 1024) OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&path_coop_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL((path_coop_kernel<4, false>), dim3(W, ninst), dim3(NTH), sh, s, a);
    }
    OEM_HIP(hipGetLastError());
    if (sw().OEM_WCOOP_FAKE_TIMEOUT.set && ninst == 1) OEM_HIP(hipMemsetAsync(a.d_out + 6, 0xFF, sizeof(double), s));      // tests: the host's fallback
    return 0;
}

}  // namespace oemgpu
