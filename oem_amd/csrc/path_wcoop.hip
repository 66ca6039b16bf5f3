// path_wcoop.hip -- p >= n where the standardised X fits the register files of <= 192 CUs: eigenvalue + penalty x lambda path in
// ONE persistent launch of cooperating workgroups (n = 500, p = 2,000: 14,009 iterations at two launches = 11.3 us each were the
// whole 158 ms of the launch-per-iteration wide engine, path_large.hip: run_path_wide, on 8 MB of data).
//
// Same arithmetic as that engine (ref src/oem_dense.h:363-366, 476-482, 513-521; operators :76-149; stop rule src/utils.cpp:537-549;
// lambda grid src/oem_dense.cpp:175-227):       u = Xs'(Ys - Xs beta)/n + d beta,   d = 1.005 lambda_max(Xs Xs'/n)
//
//   * Xs stays in REGISTERS for the whole call, COLUMNS split over G = ceil(p / (4 CW)) workgroups (one per CU, four waves): a wave
//     owns CW columns, lane l their rows l, l + 64, ... (NR per column; CW NR <= 64 doubles per lane).  A column is coordinate-local,
//     so one pass over the registers does both products (as wide_cols_kernel does from memory):
//       dot_c = x_c . r     lane-local FMAs, then ONE transposed butterfly for all CW columns of the wave (row_mirror,
//                           row_half_mirror, quad_perm: each stage halves the number of live values, CW - 1 adds instead of
//                           6 CW), the four rows through v_permlane16/32_swap; lane l ends with the column (l & 15) >> SH
//       beta_c' = T(dot_c / n + d beta_c)     one operator per lane (its own column), the coefficient lives in that lane
//       t += x_c beta_c'    v_fmac_f64_dpp row_newbcast straight from the lane that holds beta_c' (zero coefficients skipped)
//   * what crosses workgroups is the n-vector Xs beta' (or Xs Xs'v during Lanczos) as G partial vectors: an ALL-REDUCE in two
//     exchanges of n values each, through the memory side as data-tagged 16-byte pairs {lo, tag, hi, tag} (path_coop.hip's recipe:
//     the data is the flag, each 8-byte half carries its own tag, no fence; device-scope (sc1) stores and polls, ONE poll sweep in
//     flight, buffers by parity):
//       1. workgroup g stores its partial rows into the slice owner's area; owner h (rows [h SL, (h + 1) SL)) adds the G
//          partials of its rows in workgroup order (eight interleaved chains per row, combined by DPP: fixed order, bitwise
//          reproducible) and forms r = Ys - sum (or sum / n);
//       2. the owners publish their slices of r and everybody gathers all n rows.
//     The "some coefficient still moving" bit of every workgroup rides in the tag of its granules (tag = epoch << 1 | bit): the
//     owners OR what they gathered into the tags of exchange 2, so after the all-reduce every workgroup holds the same stop
//     decision and the lambda / penalty state machine runs replicated, with no flag or scalar of its own crossing workgroups.
//   * the eigenvalue step is Lanczos on n-vectors with the same registers and the same all-reduce (vector updates and both inner
//     products replicated per workgroup), the top Ritz value by the Sturm multisection of path_dev.hpp.
// Element-wise operators as above; group operators with ONE more exchange (u of the members of a workgroup's own groups; template
// parameter GEN); Nesterov's step with its inner product summed over the workgroups next to the all-reduce; compute.loss is the
// squared norm of the residual every workgroup holds.  Every spin is bounded; a timeout poisons the result (d_out[6]) and the host reports it.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "penalty_ops.hpp"
#include "path_dev.hpp"

namespace oemgpu {

namespace {

#ifndef OEM_XCHG_SLEEP
#define OEM_XCHG_SLEEP 12         // s_sleep units (64 cycles) before the first poll sweep of a gather (tools/xchg_sleep_ab.sh)
#endif
#ifndef OEM_XCHG_SLEEP2
#define OEM_XCHG_SLEEP2 0         // ... and between sweeps: measured WORSE (tools/xchg_sleep2_ab.sh, us per iteration at s_sleep 0 / 1 / 3 / 6: 500 x 20,000
                                  // 7.00 / 7.07 / 7.12 / 7.21, 500 x 2,000 3.84 / 3.96 / 4.07 / 4.22) -- once something can have landed, ask
#endif
constexpr int WNTH = 256;         // threads per workgroup: one wave per SIMD
constexpr int WCML = 256;         // Lanczos steps kept

__host__ __device__ constexpr int wc_cw(int nr) { return nr <= 4 ? 16 : (nr <= 8 ? 8 : (nr <= 16 ? 4 : 2)); }
__host__ __device__ constexpr int wc_log2(int v) { return v <= 1 ? 0 : 1 + wc_log2(v >> 1); }

constexpr int WRES_GMAX = 240;    // the resident form with columns in the accumulator file too (path_wres_kernel): up to 240 workgroups

template <int NR, int GM = WCOOP_GMAX> struct WCfg {
    static constexpr int NP = 64 * NR;                       // padded rows
    static constexpr int CW = wc_cw(NR);                     // columns per wave
    static constexpr int LG = wc_log2(CW);
    static constexpr int SH = 4 - LG;                        // lane l of a row holds column (l & 15) >> SH
    static constexpr int CPG = 4 * CW;                       // columns per workgroup
    static constexpr int E2 = (NP + WNTH - 1) / WNTH;        // rows per thread in the replicated vector work
    static constexpr int GS = NP + GM;                       // >= G SL: what a slice owner gathers
    static constexpr int E1 = (GS + WNTH - 1) / WNTH;
    // LDS carve (doubles)
    static constexpr int OFF_R = 0;                          // the n-vector of the product (residual / Lanczos v) [NP + 8]
    static constexpr int OFF_Y = OFF_R + NP + 8;             // Ys [NP]
    static constexpr int OFF_P = OFF_Y + NP;                 // the four waves' partial vectors [4][NP]
    static constexpr int OFF_G = OFF_P + 4 * NP;             // gathered partials of this workgroup's slice [GS]
    static constexpr int OFF_T = OFF_G + GS;                 // Lanczos alpha [WCML], beta [WCML]
    static constexpr int OFF_S = OFF_T + 2 * WCML;           // Sturm scratch 2 (WCML + 16)
    static constexpr int OFF_X = OFF_S + 2 * (WCML + 16);    // block reductions [2][4], theta slot, lmax words [16 + 8], votes [16 ints]
    static constexpr int N_DBL = OFF_X + 32;
};

// -DOEM_PATH_DIAG: cycles of wave 0 of workgroup 0 by segment (fenced stamps: read the SHARES)
#ifdef OEM_PATH_DIAG
__device__ unsigned long long g_diag_wcoop[16];
#define WC_STAMP(slot)                                                                     \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        unsigned long long t__;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        X.acc[slot] += t__ - X.last;                                                       \
        X.last = t__;                                                                      \
    } while (0)
#else
#define WC_STAMP(slot) do { } while (0)
#endif

struct WX {
#ifdef OEM_PATH_DIAG
    unsigned long long acc[16], last;
#endif
    __amdgpu_buffer_rsrc_t rs;         // ONE descriptor over this set's exchange buffers (four of them were sixteen SGPRs of a file that spills):
    int o2, o3, o4;                    // exchange 1 at 0: [2 parities][G owners][G senders][SL] pairs of 16 bytes; exchange 2 at o2: [2][NP]
                                       // pairs; o3: group operators, the exchange of u, [2][qpad] pairs; o4: Nesterov's step, the
                                       // workgroups' parts of its inner product, [2][G] pairs (byte offsets)
    int qpad;
    unsigned epoch;               // all-reduce counter, never 0; identical in every workgroup
    int wg, G, SL, n, row0, nsl;  // this workgroup's slice: rows [row0, row0 + nsl)
    int stride1;                  // pairs per parity of exchange 1
    int failed;                   // 0; PATH_FAILED_TIMEOUT: an exchange timed out (a partner is gone); PATH_FAILED_ABORT: the host asked to stop
    const int *abortw;            // PathArgs::abort_word
};

__device__ __forceinline__ double wc_block_sum(double v, double *red, int &rpar, int w, int lane)
{
    const double s = wave_sum(v);
    double *r = red + 4 * rpar;
    if (lane == 0) r[w] = s;
    __syncthreads();
    const double t = (r[0] + r[1]) + (r[2] + r[3]);
    rpar ^= 1;                    // the next call writes the other half: no second barrier needed
    return t;
}

// ---- the CW column sums of a wave at once.  In: s[c] = this lane's part of column c.  Out: the whole sum of column
// (lane & 15) >> SH, in every lane.  Stage on row-lane bit b (partner: xor 15 / 7 / 2 / 1, all involutions of the DPP network):
// a lane keeps the half of the live values its bit b selects and receives the partner's part of the same half.
template <int B> __device__ __forceinline__ double wc_xchg(double v)
{
    if constexpr (B == 3) return dpp_mov<0x140, 0xf>(v, 0.0);        // row_mirror: lane ^ 15
    else if constexpr (B == 2) return dpp_mov<0x141, 0xf>(v, 0.0);   // row_half_mirror: lane ^ 7
    else if constexpr (B == 1) return dpp_mov<0x4E, 0xf>(v, 0.0);    // quad_perm [2,3,0,1]: lane ^ 2
    else return dpp_mov<0xB1, 0xf>(v, 0.0);                          // quad_perm [1,0,3,2]: lane ^ 1
}
template <int M, int B, int CW> __device__ __forceinline__ void wc_stage(double (&s)[CW], int lane)
{
    if constexpr (B >= 0) {
        if constexpr (M > 1) {
            constexpr int H = M / 2;
            const bool hb = ((lane >> B) & 1) != 0;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const double keep = hb ? s[H + i] : s[i], send = hb ? s[i] : s[H + i];
                s[i] = keep + wc_xchg<B>(send);
            }
            wc_stage<H, B - 1, CW>(s, lane);
        } else {
            s[0] += wc_xchg<B>(s[0]);
            wc_stage<1, B - 1, CW>(s, lane);
        }
    }
}
template <int CW> __device__ __forceinline__ double wc_colsum(double (&s)[CW], int lane)
{
    wc_stage<CW, 3, CW>(s, lane);
    // the four rows: v_permlane16_swap (rows 1, 3 <-> 0, 2), then v_permlane32_swap (rows 2, 3 <-> 0, 1)
    const double a = s[0];
    const unsigned lo = (unsigned)__double2loint(a), hi = (unsigned)__double2hiint(a);
    auto l1 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double t = __hiloint2double((int)h1[0], (int)l1[0]) + __hiloint2double((int)h1[1], (int)l1[1]);
    const unsigned lo2 = (unsigned)__double2loint(t), hi2 = (unsigned)__double2hiint(t);
    auto l2 = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);
    auto h2 = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
    return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}

// t[k] += x[c][k] * (bn of the lane that holds column c), c = 0 .. CW - 1; columns with a zero coefficient skipped (wave-uniform)
template <int C, int NR, int CW, int SH> struct WcUpd {
    static __device__ __forceinline__ void run(double (&acc)[NR], const double &bn, const double (&x)[CW][NR], unsigned long long nz)
    {
        if constexpr (C < CW) {
            if (NR <= 2 || ((nz >> (C << SH)) & 1ull)) {
#pragma unroll
                for (int k = 0; k < NR; ++k) BcFma<(C << SH)>::fmac(acc[k], bn, x[C][k]);
            }
            WcUpd<C + 1, NR, CW, SH>::run(acc, bn, x, nz);
        }
    }
};

// Gather E granule pairs per thread (pair tid + 256 k from byte offset off0 of the buffer, where `need` has bit k) whose two tags
// carry this epoch; the low tag bits are OR-ed into `flags`.  A pair is ONE 16-byte load (each 8-byte half validates itself, so a
// torn pair is only ever seen as "not there yet"), ONE sweep in flight, and a pair that has arrived is not asked for again:
// tools/xchg_probe.hip -- three sweeps in flight (path_coop.hip's first recipe) flood the fabric with polls and make every
// exchange SLOWER (64 workgroups, 512 rows: 1.85 us per all-gather against 1.17 us).
typedef unsigned wc_v4u __attribute__((ext_vector_type(4)));
template <int E>
__device__ __forceinline__ void wc_gather(__amdgpu_buffer_rsrc_t rs, int off0, unsigned need, double (&out)[E], int &flags, WX &X, int tid)
{
    wc_v4u pv[E];
    unsigned miss = need;                                       // pairs still to come
#pragma unroll
    for (int k = 0; k < E; ++k) pv[k] = wc_v4u{0u, 0u, 0u, 0u};
    // Nothing published in this epoch can be visible yet (a store needs ~0.8 us to reach the other workgroups), and polls sent before then
    // are traffic in the way of those very stores: tools/xchg_probe.hip modes 3 / 6 -- 209 workgroups, 500 rows: 1.72 us per
    // all-gather polling at once, 1.20 us with the first sweep 768 cycles late (64 workgroups, 128 rows: 1.36 -> 0.83).  A
    // workgroup that arrives LAST sleeps too, but its partners cannot see its rows before it wakes.
    if (OEM_XCHG_SLEEP > 0) __builtin_amdgcn_s_sleep(OEM_XCHG_SLEEP);
    // ONE counter in the sweep loop: it runs out once per 1,024 sweeps (~1 ms), and only then are the abort word and the timeout looked at
    // (~1 s = 1,000 such rounds: a partner is gone; after one timeout -- or the abort word -- nobody waits again: one sweep each)
    unsigned left = X.failed ? 1u : PATH_ABORT_SPINS, rounds = 0u;
    bool ok = true;
    while (__any(miss != 0u)) {
#pragma unroll
        for (int k = 0; k < E; ++k)
            if ((miss >> k) & 1u) pv[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off0 + (tid + WNTH * k) * 16, 0, 16);      // aux 16: sc1
#pragma unroll
        for (int k = 0; k < E; ++k)
            if (((miss >> k) & 1u) && (pv[k].y >> 1) == X.epoch && (pv[k].w >> 1) == X.epoch) miss &= ~(1u << k);
        if (--left == 0u && __any(miss != 0u)) {
            if (X.failed || ++rounds >= PATH_TIMEOUT_ROUNDS) { ok = false; break; }
            if (path_abort_asked(X.abortw)) { X.failed = PATH_FAILED_ABORT; break; }
            left = PATH_ABORT_SPINS;
        }
#if OEM_XCHG_SLEEP2 > 0
        if (__any(miss != 0u)) __builtin_amdgcn_s_sleep(OEM_XCHG_SLEEP2);
#endif
    }
    if (!ok && X.failed == 0) X.failed = PATH_FAILED_TIMEOUT;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const bool nd = ((need >> k) & 1u) != 0 && ((miss >> k) & 1u) == 0;
        out[k] = nd ? __hiloint2double((int)pv[k].z, (int)pv[k].x) : 0.0;
        if (nd) flags |= (int)(pv[k].y & 1u);
    }
}

__device__ __forceinline__ void wc_publish(__amdgpu_buffer_rsrc_t rs, int off, double val, unsigned tag)
{
    wc_v4u v;
    v.x = (unsigned)__double2loint(val); v.y = tag; v.z = (unsigned)__double2hiint(val); v.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
}

// Group operators: exchange of u, one value per column.  A workgroup only needs u of the members of its own columns' groups: the
// columns on `list` (with groups of neighbouring columns a handful of values from the next workgroup, or none).  In: the column
// owners' values (one lane per column: `storer`); everybody publishes all its columns.  Out: Ush[j] for the own columns and the
// listed ones, behind a barrier.  Tagged with the epoch of the all-reduce that follows.
__device__ __forceinline__ void wc_gather_u_list(double *Ush, double u_own, int mycol, bool storer, const int *list, int nlist, bool publish, WX &X, int tid)
{
    const unsigned ep = X.epoch + 1;
    const int off3 = (int)(ep & 1u) * X.qpad * 16;
    if (storer) { if (publish) wc_publish(X.rs, X.o3 + off3 + mycol * 16, u_own, ep << 1); Ush[mycol] = u_own; }      // (publish: somebody may ask)
    const int nk = (nlist + WNTH - 1) / WNTH;
    unsigned miss = 0;
    for (int k = 0; k < nk; ++k) if (tid + WNTH * k < nlist) miss |= 1u << k;
    unsigned left = X.failed ? 1u : PATH_ABORT_SPINS, rounds = 0u;
    while (__any(miss != 0u)) {
        for (int k0 = 0; k0 < nk; k0 += 4) {
            wc_v4u pv[4];
            int col[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pv[i] = wc_v4u{0u, 0u, 0u, 0u};
                col[i] = 0;
                if ((miss >> (k0 + i)) & 1u) {
                    col[i] = list[tid + WNTH * (k0 + i)];
                    pv[i] = __builtin_amdgcn_raw_buffer_load_b128(X.rs, X.o3 + off3 + col[i] * 16, 0, 16);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (((miss >> (k0 + i)) & 1u) && (pv[i].y >> 1) == ep && (pv[i].w >> 1) == ep) {
                    Ush[col[i]] = __hiloint2double((int)pv[i].z, (int)pv[i].x);
                    miss &= ~(1u << (k0 + i));
                }
        }
        if (--left == 0u && __any(miss != 0u)) {
            if (X.failed || ++rounds >= PATH_TIMEOUT_ROUNDS) { if (X.failed == 0) X.failed = PATH_FAILED_TIMEOUT; break; }
            if (path_abort_asked(X.abortw)) { X.failed = PATH_FAILED_ABORT; break; }
            left = PATH_ABORT_SPINS;
        }
    }
    __syncthreads();
}

// Nesterov's step (ref src/oem_dense.h:633-651) restarts its sequence when sum_j (beta_j+ - beta_j')(beta_j' - beta_j) > 0 -- a sum over
// ALL coordinates.  Every workgroup publishes its part next to exchange 1 of the all-reduce (wc_adp_publish) and reads the G parts
// back after it (wc_adp_total: they landed a hop ago), adding them in workgroup order: the same bits everywhere.
__device__ __forceinline__ void wc_adp_publish(double part, WX &X, int tid)
{
    const unsigned ep = X.epoch + 1;
    if (tid == 0) wc_publish(X.rs, X.o4 + ((int)(ep & 1u) * X.G + X.wg) * 16, part, ep << 1);
}
__device__ __forceinline__ double wc_adp_total(double own, double *red, int &rpar, WX &X, int tid, int w, int lane)
{
    const unsigned ep = X.epoch;                                 // (the all-reduce in between has counted)
    const int off4 = (int)(ep & 1u) * X.G * 16;
    const bool mine = tid < X.G && tid != X.wg;
    double v = (tid == X.wg) ? own : 0.0;
    unsigned left = X.failed ? 1u : PATH_ABORT_SPINS, rounds = 0u;
    bool miss = mine;
    while (__any(miss)) {
        wc_v4u pv = wc_v4u{0u, 0u, 0u, 0u};
        if (miss) pv = __builtin_amdgcn_raw_buffer_load_b128(X.rs, X.o4 + off4 + tid * 16, 0, 16);
        if (miss && (pv.y >> 1) == ep && (pv.w >> 1) == ep) { v = __hiloint2double((int)pv.z, (int)pv.x); miss = false; }
        if (--left == 0u && __any(miss)) {
            if (X.failed || ++rounds >= PATH_TIMEOUT_ROUNDS) { if (X.failed == 0) X.failed = PATH_FAILED_TIMEOUT; break; }
            if (path_abort_asked(X.abortw)) { X.failed = PATH_FAILED_ABORT; break; }
            left = PATH_ABORT_SPINS;
        }
    }
    return wc_block_sum(v, red, rpar, w, lane);                  // thread t holds workgroup t's part: a fixed order
}

// OR of one bit per thread over the workgroup through four LDS words and ONE barrier (the caller's: `words` is read behind it)
__device__ __forceinline__ void wc_vote(int *words, int w, int lane, int bit, int extra = 0)
{
    const int wb = (__ballot(bit != 0) != 0ull ? 1 : 0) | extra;      // (extra: wave-uniform bits -- 2: this wave has seen the host's abort word)
    if (lane == 0) words[w] = wb;
}

// All-reduce of the workgroups' partial vectors.  In: the four waves' parts in Pc (no barrier yet), this thread's "moving" bit.
// Out (behind a barrier): Rsh[i] = OEM ? Ys[i] - sum_g part_g[i] : sum_g part_g[i] / n for every row i < n, and the OR of every
// thread's bit in every workgroup (the same value everywhere).  Three barriers.
template <int NR, bool OEM, int GM = WCOOP_GMAX>
__device__ __forceinline__ int wc_allreduce(double *Rsh, const double *Ysh, const double *Pc, double *Gsh, int *votes, const int (&pub)[WCfg<NR>::E2],
                                            unsigned need1, unsigned need2, double rn, int mybit, WX &X, int tid, int w, int lane)
{
    typedef WCfg<NR, GM> C;
    WC_STAMP(0);                                                 // the product (and everything between all-reduces)
    wc_vote(votes, w, lane, mybit);
    __syncthreads();                                             // Pc is complete
    const int wgbit = votes[0] | votes[1] | votes[2] | votes[3];
    WC_STAMP(1);
    ++X.epoch;
    const int par = (int)(X.epoch & 1u);
    const int off1 = par * X.stride1 * 16, off2 = par * C::NP * 16;       // (bytes; stride1 counts pairs)
    // ---- exchange 1: this workgroup's partial rows to the slice owners
    const unsigned tag1 = (X.epoch << 1) | (unsigned)wgbit;
#pragma unroll
    for (int k = 0; k < C::E2; ++k) {
        const int row = tid + WNTH * k;
        if (pub[k] != -1) {
            const double t = (Pc[row] + Pc[C::NP + row]) + (Pc[2 * C::NP + row] + Pc[3 * C::NP + row]);
            if (pub[k] < -1) Gsh[-2 - pub[k]] = t;               // a row of the own slice
            else wc_publish(X.rs, off1 + pub[k] * 16, t, tag1);
        }
    }
    WC_STAMP(2);                                                 // publish 1
    int bits = 0;
    {
        double g[C::E1];
        wc_gather<C::E1>(X.rs, off1 + X.wg * X.G * X.SL * 16, need1, g, bits, X, tid);
#pragma unroll
        for (int k = 0; k < C::E1; ++k) if ((need1 >> k) & 1u) Gsh[tid + WNTH * k] = g[k];
    }
    WC_STAMP(3);                                                 // gather 1
    wc_vote(votes + 4, w, lane, bits);
    __syncthreads();
    const int any1 = wgbit | votes[4] | votes[5] | votes[6] | votes[7];  // owners: the OR over all workgroups (every one of them sent a row)
    WC_STAMP(4);
    // ---- the slice: G partials per row in workgroup order, eight interleaved chains per row (sixteen for path_wres_kernel's
    // > 192 workgroups: a chain of 27 dependent adds per row became the longest thing between the two gathers)
    constexpr int PARTS = GM > WCOOP_GMAX ? 16 : 8;
    const unsigned tag2 = (X.epoch << 1) | (unsigned)any1;
    for (int idx = tid; idx < X.nsl * PARTS; idx += WNTH) {
        const int s = idx / PARTS, part = idx % PARTS;
        double t = 0.0;
        for (int g0 = part; g0 < X.G; g0 += 8 * PARTS) {
            double a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int g = g0 + PARTS * j; a[j] = Gsh[(g < X.G ? g : 0) * X.SL + s]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) t += (g0 + PARTS * j < X.G) ? a[j] : 0.0;
        }
        t += dpp_mov<0xB1, 0xf>(t, 0.0);
        t += dpp_mov<0x4E, 0xf>(t, 0.0);
        t += dpp_mov<0x141, 0xf>(t, 0.0);
        if constexpr (PARTS == 16) t += dpp_mov<0x140, 0xf>(t, 0.0);
        if (part == 0) {
            const int row = X.row0 + s;
            const double out = OEM ? Ysh[row] - t : t * rn;
            Rsh[row] = out;
            wc_publish(X.rs, X.o2 + off2 + row * 16, out, tag2);
        }
    }
    WC_STAMP(5);                                                 // slice sums, publish 2
    // ---- exchange 2: everybody gathers the n rows
    int bits2 = 0;
    {
        double r[C::E2];
        wc_gather<C::E2>(X.rs, X.o2 + off2, need2, r, bits2, X, tid);
#pragma unroll
        for (int k = 0; k < C::E2; ++k) if ((need2 >> k) & 1u) Rsh[tid + WNTH * k] = r[k];
    }
    WC_STAMP(6);                                                 // gather 2
    // (bit 1 of the result: somebody in this workgroup has seen the host's abort word -- the caller leaves its loops; nobody waits any more)
    wc_vote(votes + 8, w, lane, bits2, X.failed == PATH_FAILED_ABORT ? 2 : 0);
    __syncthreads();
    const int any = any1 | votes[8] | votes[9] | votes[10] | votes[11];
    WC_STAMP(7);
#ifdef OEM_PATH_DIAG
    X.acc[8] += 1;
#endif
    return any;
}

struct WThr { int kind; double L, D, rD, gammad, dmg, rdmg, gm1, gamma, dsc, rdsc, d, rd; };
__device__ __forceinline__ WThr wc_thr(const PenK &K, double d)
{
    WThr c;
    c.kind = K.kind; c.L = K.L; c.D = K.D; c.rD = 1.0 / K.D; c.gammad = K.gamma * K.D; c.gamma = K.gamma; c.gm1 = K.gamma - 1.0;
    c.dmg = K.D - 1.0 / K.gamma; c.rdmg = 1.0 / c.dmg; c.dsc = c.gm1 * K.D - 1.0; c.rdsc = 1.0 / c.dsc; c.d = d; c.rd = 1.0 / d;
    return c;
}
// element-wise operators (ref src/oem_dense.h:76-149), branch-free inside a kind (path_large.hip: wide_cols_kernel has the same)
__device__ __forceinline__ double wc_op(double u, double tp, const WThr &c)
{
    if (c.kind == K_SOFT) return cdiv(shrink(u, tp), c.D, c.rD);
    if (c.kind == K_MCP) {
        const bool big = fabs(u) > c.gammad * tp;
        return cdiv(big ? u : shrink(u, tp), big ? c.D : c.dmg, big ? c.rD : c.rdmg);
    }
    if (c.kind == K_SCAD) {
        const double au = fabs(u);
        const bool big = au > c.gammad * tp, mid = !big && au > (c.D + 1.0) * tp;
        const double num = big ? u : (mid ? shrink(c.gm1 * u, c.gamma * tp) : shrink(u, tp));
        return cdiv(num, mid ? c.dsc : c.D, mid ? c.rdsc : c.rD);
    }
    return cdiv(u, c.d, c.rd);
}

// ACC: Nesterov's step.  GEN: a group penalty in the call.  A group's norm needs u of all its members: one more exchange per iteration, of the members of this
// workgroup's own groups (wc_gather_u_list); every lane then forms the factor of its column's group.  The group tables live in LDS.
// cstart (GEN only, or null): the columns of workgroup g are [cstart[g], cstart[g + 1]) -- at most 4 CW of them, cut at group
// boundaries by the host (api.hip) when every group is a run of neighbouring columns: then no group reaches into another workgroup
// and the exchange of u has nobody to serve.  Null: 4 CW columns each.
template <int NR, bool GEN, bool ACC>
__global__ __launch_bounds__(WNTH) void path_wcoop_kernel(PathArgs A, const double *__restrict__ xs, const double *__restrict__ ysv, int n,
                                                           unsigned long long *xchg, long long set_stride, const int *__restrict__ cstart)
{
    typedef WCfg<NR> C;
    constexpr int NP = C::NP, CW = C::CW, SH = C::SH, E2 = C::E2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15;
    const int q = A.p, wg = blockIdx.x, G = gridDim.x;
    const unsigned long long t_cyc0 = __builtin_amdgcn_s_memtime(), t_rt0 = __builtin_amdgcn_s_memrealtime();
    double *Rsh = lds + C::OFF_R, *Ysh = lds + C::OFF_Y, *Pc = lds + C::OFF_P, *Gsh = lds + C::OFF_G;
    double *Tal = lds + C::OFF_T, *Tbe = Tal + WCML, *red = lds + C::OFF_X;
    int *votes = reinterpret_cast<int *>(red + 24);
    // blockIdx.y: one set of G workgroups per penalty (independent cold starts, ref src/oem_dense.cpp:243-244), each with the whole
    // of Xs in its registers and exchange buffers of its own; set y walks penalties y, y + sets, ...
    const int set = blockIdx.y, nsets = gridDim.y;
    xchg += (size_t)set * (size_t)set_stride;
    const bool writer = wg == 0;
    const double rn = 1.0 / (double)n;

    // ---- this wave's CW columns in registers; this lane's own column (the one whose sum and coefficient it ends up with)
    const int c0 = (GEN && cstart) ? cstart[wg] : wg * C::CPG;
    const int c1 = (GEN && cstart) ? cstart[wg + 1] : (c0 + C::CPG < q ? c0 + C::CPG : q);
    const int cbase = c0 + w * CW;
    double x[CW][NR];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const bool ok = cbase + c < c1;
        const double *col = xs + (size_t)(ok ? cbase + c : 0) * NP;
#pragma unroll
        for (int k = 0; k < NR; ++k) { const double t = col[lane + 64 * k]; x[c][k] = ok ? t : 0.0; }
    }
    const int mycol = cbase + (l16 >> SH);
    const bool colok = mycol < c1;
    const bool storer = colok && lane < 16 && (l16 & ((1 << SH) - 1)) == 0;       // one lane per column writes its coefficient out
    const double pfj = colok ? A.pf[mycol] : 0.0;
    for (int j = tid; j < NP + 8; j += WNTH) Rsh[j] = 0.0;
    for (int j = tid; j < NP; j += WNTH) Ysh[j] = j < n ? ysv[j] : 0.0;
    for (int j = tid; j < C::GS; j += WNTH) Gsh[j] = 0.0;

    // ---- the exchange: slices, what this thread publishes and gathers (fixed for the whole call)
    WX X;
    X.G = G; X.wg = wg; X.n = n; X.SL = (n + G - 1) / G; X.row0 = wg * X.SL;
    X.nsl = n - X.row0 < 0 ? 0 : (n - X.row0 < X.SL ? n - X.row0 : X.SL);
    X.stride1 = G * G * X.SL;
    X.qpad = G * C::CPG;
    X.o2 = 2 * X.stride1 * 16; X.o3 = X.o2 + 2 * NP * 16; X.o4 = X.o3 + 2 * X.qpad * 16;
    X.rs = __builtin_amdgcn_make_buffer_rsrc((void *)xchg, 0, X.o4 + 2 * G * 16, 0x00020000);
    // group operators: u of the columns this workgroup's groups touch (indexed by column), the group tables
    const int ng = GEN ? A.ngroups : 0, qp = (q + 8 + 1) & ~1, ngp = (ng + 2) & ~1;
    double *Ush = lds + C::N_DBL, *GWsh = Ush + qp;
    int *gidL = reinterpret_cast<int *>(GWsh + ngp), *gstartL = gidL + qp, *gidxL = gstartL + ngp + 2, *gzeroL = gidxL + qp;
    int *needL = gzeroL + ngp, *nneedL = needL + qp;             // the columns of other workgroups this one's groups reach into
    if (GEN) {
        for (int j = tid; j < qp; j += WNTH) Ush[j] = 0.0;
        for (int j = tid; j < qp; j += WNTH) gidL[j] = (ng > 0 && j < q) ? A.gid[j] : -1;
        if (ng > 0) {
            for (int g = tid; g <= ng; g += WNTH) gstartL[g] = A.gstart[g];
            for (int g = tid; g < ng; g += WNTH) { gzeroL[g] = A.gzero[g]; GWsh[g] = A.gw[g]; }
            const int nm = A.gstart[ng];
            for (int m = tid; m < nm; m += WNTH) gidxL[m] = A.gidx[m];
        }
    }
    X.epoch = 0; X.failed = 0; X.abortw = A.abort_word;
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;
#endif
    int pub[E2];                                                // -1: nothing; <= -2: own slice, Gsh index -2 - pub; else pair index in exchange 1
    unsigned need1 = 0, need2 = 0;
#pragma unroll
    for (int k = 0; k < E2; ++k) {
        const int row = tid + WNTH * k;
        pub[k] = -1;
        if (row < n) {
            const int h = row / X.SL, s = row - h * X.SL;
            pub[k] = (h == wg) ? -2 - (wg * X.SL + s) : (h * G + wg) * X.SL + s;
            if (h != wg) need2 |= 1u << k;
        }
    }
#pragma unroll
    for (int k = 0; k < C::E1; ++k) {
        const int e = tid + WNTH * k;
        if (e < G * X.SL) {
            const int g = e / X.SL, s = e - g * X.SL;
            if (g != wg && s < X.nsl) need1 |= 1u << k;
        }
    }
    __syncthreads();
    int nneed = 0;
    if (GEN && ng > 0) {
        if (tid == 0) nneedL[0] = 0;
        __syncthreads();
        const int cme = c0 + tid;
        if (cme < c1) {
            const int gi = gidL[cme];
            bool first = gi >= 0;
            for (int cc = c0; cc < cme && first; ++cc) first = gidL[cc] != gi;      // one own column per group walks its members
            if (first)
                for (int m = gstartL[gi]; m < gstartL[gi + 1]; ++m) {
                    const int j = gidxL[m];
                    if (j < c0 || j >= c1) needL[atomicAdd(&nneedL[0], 1)] = j;
                }
        }
        __syncthreads();
        nneed = nneedL[0];
    }
    // this lane's column: its group, fixed for the whole call -- weight, member range and the first eight member indices in registers
    // (per iteration the walk is then ONE round of LDS reads for groups of up to eight)
    int mygi = -1, gm0 = 0, gm1 = 0, gix[8];
    bool mygz = true;
    double mygw = 0.0;
#pragma unroll
    for (int t = 0; t < 8; ++t) gix[t] = q;                      // slot q: zero words
    if (GEN && ng > 0 && colok) {
        mygi = gidL[mycol];
        if (mygi >= 0) {
            mygz = gzeroL[mygi] != 0; mygw = GWsh[mygi]; gm0 = gstartL[mygi]; gm1 = gstartL[mygi + 1];
#pragma unroll
            for (int t = 0; t < 8; ++t) if (gm0 + t < gm1) gix[t] = gidxL[gm0 + t];
        }
    }
    int rpar = 0;

    // both products over the registers: col_dots() = x_c . Rsh for this lane's column (every lane of the column gets it);
    // col_update(b) = this wave's part of sum_c x_c b_c into Pc (b: the value of this lane's column)
    double bcur = 0.0;                                          // (element-wise form) the coefficient of this lane's column
    auto col_dots = [&]() __attribute__((always_inline)) {
        double rr[NR], s[CW];
#pragma unroll
        for (int k = 0; k < NR; ++k) rr[k] = Rsh[lane + 64 * k];
        WC_STAMP(9);                                             // (state machine, operator constants; the vector's LDS reads issued)
#pragma unroll
        for (int cc = 0; cc < CW; ++cc) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int k = 0; k < NR; k += 2) { a0 = fma(x[cc][k], rr[k], a0); if (k + 1 < NR) a1 = fma(x[cc][k + 1], rr[k + 1], a1); }
            s[cc] = a0 + a1;
        }
        WC_STAMP(10);                                            // dot products
        const double dot = wc_colsum<CW>(s, lane);
        WC_STAMP(11);                                            // column sums
        return dot;
    };
    auto col_update = [&](double b) __attribute__((always_inline)) {
        double bn[1] = {b};
        const unsigned long long nz = __ballot(bn[0] != 0.0);
        dpp_hazard_fence(bn);
        double acc[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) acc[k] = 0.0;
        WcUpd<0, NR, CW, SH>::run(acc, bn[0], x, nz);
        WC_STAMP(13);                                            // update
#pragma unroll
        for (int k = 0; k < NR; ++k) Pc[w * NP + lane + 64 * k] = acc[k];
    };

    // ---- eigenvalue step: Lanczos on Xs Xs'/n (ref src/oem_dense.h:476-498), the vector updates replicated per workgroup
    double v[E2], vp[E2], wv[E2];
    bool rowok[E2];
    {
        double nn = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) {
            const unsigned j = tid + WNTH * k;
            rowok[k] = (int)j < n;
            const unsigned h = j * 2654435761u + 12345u;                 // deterministic non-structured start
            v[k] = rowok[k] ? ((double)(h >> 8) * (1.0 / 16777216.0) - 0.5) : 0.0;
            vp[k] = 0.0;
            nn = fma(v[k], v[k], nn);
        }
        nn = 1.0 / sqrt(wc_block_sum(nn, red, rpar, w, lane));
#pragma unroll
        for (int k = 0; k < E2; ++k) v[k] *= nn;
    }
    int msteps = n < WCML ? n : WCML;
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
        for (int k = 0; k < E2; ++k) if (rowok[k]) Rsh[tid + WNTH * k] = v[k];
        __syncthreads();
        col_update(col_dots());                                   // z = Xs'v, this workgroup's part of Xs z
        (void)wc_allreduce<NR, false>(Rsh, Ysh, Pc, Gsh, votes, pub, need1, need2, rn, 0, X, tid, w, lane);
        double al = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) { wv[k] = rowok[k] ? Rsh[tid + WNTH * k] : 0.0; al = fma(v[k], wv[k], al); }
        al = wc_block_sum(al, red, rpar, w, lane);
        double bb = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) {
            wv[k] = (wv[k] - al * v[k]) - bprev * vp[k];
            bb = fma(wv[k], wv[k], bb);
        }
        double ib;
        sqrt_rsqrt(wc_block_sum(bb, red, rpar, w, lane), bb, ib);
        if (tid == 0) { Tal[j] = al; Tbe[j] = bb; }
        nst = j + 1;
        if (!(bb > 1e-13 * fabs(al))) break;                        // invariant subspace reached: T is exact
        if (lanczos_check_due(nst) && nst < msteps) {
            const double th = top_ritz(nst, theta_prev);            // its first barrier publishes Tal / Tbe
            if (lanczos_converged(th, theta_prev, mv_prev)) { theta = th; have_theta = true; break; }
        }
#pragma unroll
        for (int k = 0; k < E2; ++k) { vp[k] = v[k]; v[k] = wv[k] * ib; }
        bprev = bb;
    }
    if (!have_theta) { __syncthreads(); theta = top_ritz(nst, theta_prev); }
    const double d = theta * 1.005;                                  // ref src/oem_dense.h:498
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;                       // (the stamps below cover the lambda path only)
    X.last = __builtin_amdgcn_s_memtime();
#endif
    if (tid == 0 && writer && set == 0) { A.d_out[0] = d; A.d_out[1] = theta; A.d_out[4] = (double)nst; A.d_out[5] = (!have_theta && nst >= msteps && nst < n) ? 1.0 : 0.0; }

    // ---- lambda grid constants (ref src/oem_dense.cpp:175-192)
    const double scaley = A.yscale ? A.stats[1] : 1.0;
    const int nl = A.nl;
    double lmax = 0.0;
    {
        double m = 0.0;
        for (int j0 = 0; j0 < q; j0 += 8 * WNTH) {
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = j0 + tid + WNTH * k; t[k] = (j < q && j >= A.lmax_from) ? fabs((A.lmax_xy ? A.lmax_xy : A.xy)[j]) : 0.0; }
#pragma unroll
            for (int k = 0; k < 8; ++k) m = fmax(m, t[k]);
        }
        m = wave_max(m);
        if (lane == 0) red[12 + w] = m;
        __syncthreads();
        lmax = fmax(fmax(red[12], red[13]), fmax(red[14], red[15])) * scaley;
    }
    const double llo = log(lmax), lhi = log(A.lambda_min_ratio * lmax);
    const double lstep = nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0;
    const bool lflip = fabs(lhi) < fabs(llo);

    bool left = false;                                           // the host's abort word was seen (PathArgs::abort_word): every loop is left
    unsigned tick = 0u;
    for (int pp = A.pen_lo + set; pp < A.pen_hi; pp += nsets) {
        const int pen = A.penalty[pp];
        const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
        const bool isnet = pen_is_net(pen);
        const int maxit = A.maxit;
        // cold start (ref src/oem_dense.cpp:243-244): beta = 0, so the residual is Ys
        bcur = 0.0;
        double ak = 1.0;                                          // Nesterov's sequence (ref src/oem_dense.h:633-651), restarted per penalty
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E2; ++k) if (rowok[k]) Rsh[tid + WNTH * k] = Ysh[tid + WNTH * k];
        __syncthreads();
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
            const WThr c = wc_thr(K, d);
            const double tp = pfj * K.L;
            int it = 0;
            for (;;) {
                if ((tick++ & 127u) == 0u && path_abort_asked(X.abortw)) X.failed = PATH_FAILED_ABORT;
                int any = 0;
                const bool grp = GEN && K.kind >= K_GRP;
                // 1 - pen / ||u_g|| etc. from the squared norm of a group (ref src/oem_dense.h:193-315; quirk Q6: ||u_g|| = 0 => 0)
                auto group_factor = [&](double s2, double pen_g) {
                    if (K.kind == K_GRP || K.kind == K_SGL) {
                        double nrm, rnm;                                 // root and reciprocal from v_rsq_f64 + Goldschmidt, the quotient
                        sqrt_rsqrt_lane(s2, nrm, rnm);                   // refined like cdiv (path_coop.hip has the same)
                        const double t = 1.0 - cdiv(pen_g, nrm, rnm);
                        return (s2 > 0.0 && 0.0 < t) ? t : 0.0;
                    }
                    const double nrm = sqrt(s2);
                    return (K.kind == K_GRP_MCP) ? mcp_norm(nrm, pen_g, K.D, K.gamma) : scad_norm(nrm, pen_g, K.D, K.gamma);
                };
                // Nesterov's step (ref src/oem_dense.h:633-651): beta+ = beta' + ratio (beta' - beta), its inner product summed over all
                // workgroups next to the all-reduce; the stop rule looks at beta+ against beta
                constexpr bool acc = ACC;                            // (a template parameter: the plain iteration carries none of it)
                const double akn = acc ? 0.5 * (1.0 + sqrt(1.0 + 4.0 * ak * ak)) : 1.0, ratio = acc ? (ak - 1.0) / akn : 0.0;
                auto finish_iteration = [&](double upd) __attribute__((always_inline)) {
                    double b = upd, part = 0.0;
                    if (acc) {
                        const double diff = upd - bcur;
                        b = upd + ratio * diff;
                        part = wc_block_sum(storer ? (b - upd) * diff : 0.0, red, rpar, w, lane);
                        wc_adp_publish(part, X, tid);
                    }
                    const double cu = fabs(b), qo = fabs(bcur);
                    const bool cn = cu > 1e-13, qn = qo > 1e-13;      // ref src/utils.cpp:537-549
                    const bool moving = (cn != qn) || (cn && qn && fabs(b - bcur) > A.tol * qo);
                    bcur = b;
                    WC_STAMP(12);                                    // operator, stop rule
                    col_update(b);
                    // r' = Ys - Xs beta': the next iteration's input, or the warm start of the next lambda
                    const int mv = wc_allreduce<NR, true>(Rsh, Ysh, Pc, Gsh, votes, pub, need1, need2, rn, moving ? 1 : 0, X, tid, w, lane);
                    if (acc) ak = (wc_adp_total(part, red, rpar, X, tid, w, lane) > 0.0) ? 1.0 : akn;
                    return mv;
                };
                if (!grp) {
                    const double dot = col_dots();
                    const double u = dot * rn + d * bcur;            // ref src/oem_dense.h:520: X'(Y - X beta)/n + d beta
                    any = finish_iteration(colok ? wc_op(u, tp, c) : 0.0);
                } else if constexpr (GEN) {
                    // ---- a group operator: u of the members of this workgroup's groups (its own columns, and what the list names),
                    // then every lane forms the factor of ITS column's group and the coefficient of its column
                    const double dot = col_dots();
                    const double uo = colok ? dot * rn + d * bcur : 0.0;
                    wc_gather_u_list(Ush, uo, mycol, storer, needL, nneed, cstart == nullptr, X, tid);
                    WC_STAMP(14);                                    // exchange of u
                    const bool sgl = K.kind == K_SGL;                // sparse group lasso: the soft-thresholded u feeds the norms
                    double f = 0.0;
                    if (mygi >= 0) {
                        f = 1.0;
                        if (!mygz) {
                            double s2 = 0.0;                         // summed in member order like the reference
                            {
                                double xv[8], pv8[8];
#pragma unroll
                                for (int t = 0; t < 8; ++t) { xv[t] = Ush[gix[t]]; pv8[t] = (sgl && gix[t] < q) ? A.pf[gix[t]] : 0.0; }      // (sparse group lasso only: from L2)
#pragma unroll
                                for (int t = 0; t < 8; ++t) { const double xs1 = sgl ? soft1(xv[t], pv8[t] * K.L1, 1.0) : xv[t]; s2 += xs1 * xs1; }
                            }
                            for (int m = gm0 + 8; m < gm1; m += 8) {     // longer groups: eight members per trip, the index reads together
                                int ix[8];
                                double xv[8], pv8[8];
#pragma unroll
                                for (int t = 0; t < 8; ++t) ix[t] = (m + t < gm1) ? gidxL[m + t] : q;
#pragma unroll
                                for (int t = 0; t < 8; ++t) { xv[t] = Ush[ix[t]]; pv8[t] = (sgl && ix[t] < q) ? A.pf[ix[t]] : 0.0; }
#pragma unroll
                                for (int t = 0; t < 8; ++t) { const double xs1 = sgl ? soft1(xv[t], pv8[t] * K.L1, 1.0) : xv[t]; s2 += xs1 * xs1; }
                            }
                            f = group_factor(s2, K.L * mygw);
                        }
                    }
                    WC_STAMP(15);                                    // group norm and factor
                    const double us = sgl ? soft1(uo, pfj * K.L1, 1.0) : uo;
                    any = finish_iteration((colok && f != 0.0) ? cdiv(us * f, K.D, c.rD) : 0.0);
                }
                ++it;
                if (any & 2) { left = true; break; }
                const bool conv = !(any & 1);
                if (conv || it >= maxit) {
                    if (storer) A.beta[orow * q + mycol] = bcur;
                    // compute.loss (ref src/oem_dense.h:759-770): sum (Ys - Xs beta)^2 -- the residual every workgroup already holds
                    double loss = 1e99;
                    if (A.compute_loss) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < E2; ++k) { const double r = rowok[k] ? Rsh[tid + WNTH * k] : 0.0; t = fma(r, r, t); }
                        loss = wc_block_sum(t, red, rpar, w, lane);
                    }
                    if (tid == 0 && writer) { A.niter[orow] = conv ? it : maxit + 1; A.loss[orow] = loss; }      // ref src/oem_base.h:94-109
                    break;
                }
            }
            if (left) break;
        }
        if (left) break;
    }
#ifdef OEM_PATH_DIAG
    if (tid == 0 && writer) for (int k = 0; k < 16; ++k) g_diag_wcoop[k] = X.acc[k];
#endif
    if (tid == 0 && writer && set == 0) {
        A.d_out[2] = (double)(__builtin_amdgcn_s_memtime() - t_cyc0);
        A.d_out[3] = (double)(__builtin_amdgcn_s_memrealtime() - t_rt0);
    }
    // exchange timeout: poison (the host turns it into an error); a slot of its own that the host cleared before the launch
    if (__syncthreads_or(X.failed ? 1 : 0) && tid == 0) A.d_out[6] = 1.0;
}

// ================================================================================================
// The resident form BEYOND the VGPR tile (round 4; VERDICT r3 item 5: "a persistent form for the 3 M - 50 M-entry band"): the same
// persistent launch with NX more column sets of every wave in the ACCUMULATOR file -- a0..a255, 128 doubles per lane, named by inline
// asm alone (areg_rd / areg_wr; oem_amd/build.py audits that hipcc itself emits no v_accvgpr and no scratch here) -- so a wave owns
// (1 + NX) CW columns instead of CW and Xs up to ~11 M entries stays in registers (n = 500, p = 20,000: 209 workgroups x 96 columns;
// the launch-per-iteration engine re-reads those 82 MB every iteration at 20 us, the streamed persistent form at 19.6).  An
// accumulator value costs two v_accvgpr_read_b32 on its way into an FMA, in each of the two products; everything else -- the
// all-reduce of the n-vector, the state machine, the eigenvalue step -- is path_wcoop_kernel's.  Element-wise operators and
// compute.loss; one workgroup set (the penalties in turn); up to WRES_GMAX workgroups, i.e. MORE than three quarters of the CUs:
// the exchanges are bounded and a timeout sends the call to the launch-per-iteration engine as everywhere (api.hip: run_paths).
__device__ __forceinline__ double wc_uni(double v)              // a wave-uniform value into scalar registers
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int NR> struct WRes {
    static constexpr int CW = wc_cw(NR), TILE = CW * NR;
    static constexpr int NX = 128 / TILE, NS = 1 + NX;          // accumulator sets, sets in all
    static constexpr int CPG = 4 * CW * NS;                      // columns per workgroup
};
// s[cc] = sum_k a[BASE + cc NR + k] rr[k]: the lane-local part of the column dots of an accumulator set
template <int BASE, int NR, int CW, int CC = 0> struct WrDots {
    static __device__ __forceinline__ void run(double (&s)[CW], const double (&rr)[NR])
    {
        if constexpr (CC < CW) {
            double a0 = 0.0, a1 = 0.0;
            static_for_dev<NR>([&](auto K_) {
                constexpr int k = decltype(K_)::value;
                const double xv = areg_rd<BASE + CC * NR + k>();
                if constexpr ((k & 1) == 0) a0 = fma(xv, rr[k], a0); else a1 = fma(xv, rr[k], a1);
            });
            s[CC] = a0 + a1;
            WrDots<BASE, NR, CW, CC + 1>::run(s, rr);
        }
    }
};
// acc[k] += a[BASE + c NR + k] * (bn of the lane that holds column c); columns with a zero coefficient skipped (wave-uniform)
template <int BASE, int NR, int CW, int SH, int C = 0> struct WrUpd {
    static __device__ __forceinline__ void run(double (&acc)[NR], const double &bn, unsigned long long nz)
    {
        if constexpr (C < CW) {
            if (NR <= 2 || ((nz >> (C << SH)) & 1ull)) {
                static_for_dev<NR>([&](auto K_) {
                    constexpr int k = decltype(K_)::value;
                    const double xv = areg_rd<BASE + C * NR + k>();
                    BcFma<(C << SH)>::fmac(acc[k], bn, xv);
                });
            }
            WrUpd<BASE, NR, CW, SH, C + 1>::run(acc, bn, nz);
        }
    }
};

template <int NR>
__global__ __launch_bounds__(WNTH) void path_wres_kernel(PathArgs A, const double *__restrict__ xs, const double *__restrict__ ysv, int n,
                                                          unsigned long long *xchg)
{
    typedef WCfg<NR, WRES_GMAX> C;
    typedef WRes<NR> R;
    constexpr int NP = C::NP, CW = C::CW, SH = C::SH, E2 = C::E2, NX = R::NX, NS = R::NS, TILE = R::TILE;
    asm volatile("" ::: "a255");                                 // the accumulator file is in use (by the asm alone)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15;
    const int q = A.p, wg = blockIdx.x, G = gridDim.x;
    const unsigned long long t_cyc0 = __builtin_amdgcn_s_memtime(), t_rt0 = __builtin_amdgcn_s_memrealtime();
    double *Rsh = lds + C::OFF_R, *Ysh = lds + C::OFF_Y, *Pc = lds + C::OFF_P, *Gsh = lds + C::OFF_G;
    double *Tal = lds + C::OFF_T, *Tbe = Tal + WCML, *red = lds + C::OFF_X;
    int *votes = reinterpret_cast<int *>(red + 24);
    const bool writer = wg == 0;
    const double rn = wc_uni(1.0 / (double)n);

    // ---- the lambda grid (ref src/oem_dense.cpp:175-227) FIRST, while no tile is live: log / exp take more registers than the tiles
    // leave.  Every workgroup writes the same values (the same bits) to lambda_out and reads them back in the path loop.
    const double scaley = wc_uni(A.yscale ? A.stats[1] : 1.0);
    const int nl = A.nl;
    {
        double m = 0.0;
        for (int j0 = 0; j0 < q; j0 += 8 * WNTH) {
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = j0 + tid + WNTH * k; t[k] = (j < q && j >= A.lmax_from) ? fabs((A.lmax_xy ? A.lmax_xy : A.xy)[j]) : 0.0; }
#pragma unroll
            for (int k = 0; k < 8; ++k) m = fmax(m, t[k]);
        }
        m = wave_max(m);
        if (lane == 0) red[12 + w] = m;
        __syncthreads();
        const double lmax = fmax(fmax(red[12], red[13]), fmax(red[14], red[15])) * scaley;
        const double llo = log(lmax), lhi = log(A.lambda_min_ratio * lmax);
        const double lstep = nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0;
        const bool lflip = fabs(lhi) < fabs(llo);
        for (int idx = tid; idx < (A.pen_hi - A.pen_lo) * nl; idx += WNTH) {
            const int pp = A.pen_lo + idx / nl, i = idx % nl;
            double lam;
            if (A.user_lambda) lam = A.lambda_user[(size_t)pp * nl + i];
            else {
                double lv;
                if (nl == 1) lv = lhi;
                else if (lflip) lv = (i == 0) ? llo : lhi - (double)(nl - 1 - i) * lstep;
                else lv = (i == nl - 1) ? lhi : llo + (double)i * lstep;
                lam = exp(lv);
                if (pen_is_net(A.penalty[pp])) lam = lam / A.alpha;
            }
            A.lambda_out[(size_t)pp * nl + i] = lam;
        }
        __syncthreads();
    }

    // ---- this wave's NS CW columns: set 0 in x[][], sets 1 .. NX in a[(s - 1) TILE + c NR + k]
    const int c0 = wg * R::CPG, c1 = c0 + R::CPG < q ? c0 + R::CPG : q;
    const int cbase = c0 + w * (CW * NS);                       // set s: columns cbase + s CW ...
    double x[CW][NR];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const bool ok = cbase + c < c1;
        const double *col = xs + (size_t)(ok ? cbase + c : 0) * NP;
#pragma unroll
        for (int k = 0; k < NR; ++k) { const double t = col[lane + 64 * k]; x[c][k] = ok ? t : 0.0; }
    }
    static_for_dev<NX * CW>([&](auto I_) {
        constexpr int i = decltype(I_)::value;                   // (set - 1) CW + column
        const int colj = cbase + CW + i;
        const bool ok = colj < c1;
        const double *col = xs + (size_t)(ok ? colj : 0) * NP;
        double t[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) t[k] = col[lane + 64 * k];
        static_for_dev<NR>([&](auto K_) { constexpr int k = decltype(K_)::value; areg_wr<i * NR + k>(ok ? t[k] : 0.0); });
    });
    const int mycol0 = cbase + (l16 >> SH);                      // set s: mycol0 + s CW
    const bool storer = lane < 16 && (l16 & ((1 << SH) - 1)) == 0;       // one lane per column writes its coefficient out
    for (int j = tid; j < NP + 8; j += WNTH) Rsh[j] = 0.0;
    for (int j = tid; j < NP; j += WNTH) Ysh[j] = j < n ? ysv[j] : 0.0;
    for (int j = tid; j < C::GS; j += WNTH) Gsh[j] = 0.0;

    // ---- the exchange (as path_wcoop_kernel)
    WX X;
    X.G = G; X.wg = wg; X.n = n; X.SL = (n + G - 1) / G; X.row0 = wg * X.SL;
    X.nsl = n - X.row0 < 0 ? 0 : (n - X.row0 < X.SL ? n - X.row0 : X.SL);
    X.stride1 = G * G * X.SL;
    X.qpad = 0;
    X.o2 = 2 * X.stride1 * 16; X.o3 = X.o2 + 2 * NP * 16; X.o4 = X.o3;
    X.rs = __builtin_amdgcn_make_buffer_rsrc((void *)xchg, 0, X.o3, 0x00020000);
    X.epoch = 0; X.failed = 0; X.abortw = A.abort_word;
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;
    X.last = __builtin_amdgcn_s_memtime();
#endif
    int pub[E2];
    unsigned need1 = 0, need2 = 0;
#pragma unroll
    for (int k = 0; k < E2; ++k) {
        const int row = tid + WNTH * k;
        pub[k] = -1;
        if (row < n) {
            const int h = row / X.SL, s = row - h * X.SL;
            pub[k] = (h == wg) ? -2 - (wg * X.SL + s) : (h * G + wg) * X.SL + s;
            if (h != wg) need2 |= 1u << k;
        }
    }
#pragma unroll
    for (int k = 0; k < C::E1; ++k) {
        const int e = tid + WNTH * k;
        if (e < G * X.SL) {
            const int g = e / X.SL, s = e - g * X.SL;
            if (g != wg && s < X.nsl) need1 |= 1u << k;
        }
    }
    __syncthreads();
    int rpar = 0;

    // the column dots of every set against Rsh: dots[s] = x_c . Rsh for this lane's column of set s
    auto all_dots = [&](double (&dots)[NS]) __attribute__((always_inline)) {
        double rr[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) rr[k] = Rsh[lane + 64 * k];
        WC_STAMP(9);
        {
            double s[CW];
#pragma unroll
            for (int cc = 0; cc < CW; ++cc) {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int k = 0; k < NR; k += 2) { a0 = fma(x[cc][k], rr[k], a0); if (k + 1 < NR) a1 = fma(x[cc][k + 1], rr[k + 1], a1); }
                s[cc] = a0 + a1;
            }
            dots[0] = wc_colsum<CW>(s, lane);
        }
        static_for_dev<NX>([&](auto S_) {
            constexpr int sx = decltype(S_)::value;
            double s[CW];
            WrDots<sx * TILE, NR, CW>::run(s, rr);
            dots[1 + sx] = wc_colsum<CW>(s, lane);
        });
        WC_STAMP(10);                                            // dot products and column sums
    };
    // this wave's part of sum_c x_c b_c over every set into Pc (b[s]: the value of this lane's column of set s)
    auto all_update = [&](const double (&b)[NS]) __attribute__((always_inline)) {
        double acc[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) acc[k] = 0.0;
        {
            double bn[1] = {b[0]};
            const unsigned long long nz = __ballot(bn[0] != 0.0);
            dpp_hazard_fence(bn);
            WcUpd<0, NR, CW, SH>::run(acc, bn[0], x, nz);
        }
        static_for_dev<NX>([&](auto S_) {
            constexpr int sx = decltype(S_)::value;
            double bn[1] = {b[1 + sx]};
            const unsigned long long nz = __ballot(bn[0] != 0.0);
            dpp_hazard_fence(bn);
            WrUpd<sx * TILE, NR, CW, SH>::run(acc, bn[0], nz);
        });
        WC_STAMP(13);                                            // update
#pragma unroll
        for (int k = 0; k < NR; ++k) Pc[w * NP + lane + 64 * k] = acc[k];
    };

    // ---- eigenvalue step: Lanczos on Xs Xs'/n (ref src/oem_dense.h:476-498), the vector updates replicated per workgroup
    double v[E2], vp[E2], wv[E2];
    bool rowok[E2];
    {
        double nn = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) {
            const unsigned j = tid + WNTH * k;
            rowok[k] = (int)j < n;
            const unsigned h = j * 2654435761u + 12345u;                 // deterministic non-structured start
            v[k] = rowok[k] ? ((double)(h >> 8) * (1.0 / 16777216.0) - 0.5) : 0.0;
            vp[k] = 0.0;
            nn = fma(v[k], v[k], nn);
        }
        nn = 1.0 / sqrt(wc_block_sum(nn, red, rpar, w, lane));
#pragma unroll
        for (int k = 0; k < E2; ++k) v[k] *= nn;
    }
    int msteps = n < WCML ? n : WCML;
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
        for (int k = 0; k < E2; ++k) if (rowok[k]) Rsh[tid + WNTH * k] = v[k];
        __syncthreads();
        {
            double z[NS];
            all_dots(z);                                          // z = Xs'v, then this workgroup's part of Xs z
            all_update(z);
        }
        (void)wc_allreduce<NR, false, WRES_GMAX>(Rsh, Ysh, Pc, Gsh, votes, pub, need1, need2, rn, 0, X, tid, w, lane);
        double al = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) { wv[k] = rowok[k] ? Rsh[tid + WNTH * k] : 0.0; al = fma(v[k], wv[k], al); }
        al = wc_block_sum(al, red, rpar, w, lane);
        double bb = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) {
            wv[k] = (wv[k] - al * v[k]) - bprev * vp[k];
            bb = fma(wv[k], wv[k], bb);
        }
        double ib;
        sqrt_rsqrt(wc_block_sum(bb, red, rpar, w, lane), bb, ib);
        if (tid == 0) { Tal[j] = al; Tbe[j] = bb; }
        nst = j + 1;
        if (!(bb > 1e-13 * fabs(al))) break;                        // invariant subspace reached: T is exact
        if (lanczos_check_due(nst) && nst < msteps) {
            const double th = top_ritz(nst, theta_prev);
            if (lanczos_converged(th, theta_prev, mv_prev)) { theta = th; have_theta = true; break; }
        }
#pragma unroll
        for (int k = 0; k < E2; ++k) { vp[k] = v[k]; v[k] = wv[k] * ib; }
        bprev = bb;
    }
    if (!have_theta) { __syncthreads(); theta = top_ritz(nst, theta_prev); }
    const double d = wc_uni(theta * 1.005);                          // ref src/oem_dense.h:498 (uniform values in scalar registers: the vector file holds a tile)
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;
    X.last = __builtin_amdgcn_s_memtime();
#endif
    if (tid == 0 && writer) { A.d_out[0] = d; A.d_out[1] = theta; A.d_out[4] = (double)nst; A.d_out[5] = (!have_theta && nst >= msteps && nst < n) ? 1.0 : 0.0; }

    unsigned colok = 0;                                          // bit s: this lane's column of set s exists
    double bcur[NS];                                             // (declared behind the eigenvalue step: nothing of the path lives across it)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        colok |= (mycol0 + s * CW < c1) ? 1u << s : 0u;
        bcur[s] = 0.0;
    }
    bool left = false;                                           // the host's abort word was seen (PathArgs::abort_word): every loop is left
    unsigned tick = 0u;
    for (int pp = A.pen_lo; pp < A.pen_hi; ++pp) {
        const int pen = A.penalty[pp];
        const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
        const int maxit = A.maxit;
        // cold start (ref src/oem_dense.cpp:243-244): beta = 0, so the residual is Ys
#pragma unroll
        for (int s = 0; s < NS; ++s) bcur[s] = 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E2; ++k) if (rowok[k]) Rsh[tid + WNTH * k] = Ysh[tid + WNTH * k];
        __syncthreads();
        for (int i = 0; i < nl; ++i) {
            const size_t orow = (size_t)pp * nl + i;
            if (i >= nlam) continue;
            const double lam = wc_uni(A.lambda_out[orow]);
            double al_ = A.alpha, ga_ = A.gamma, ta_ = A.tau;
            asm volatile("" : "+s"(al_), "+s"(ga_), "+s"(ta_));      // (opaque per lambda: what hipcc hoists out of these loops it parks in the accumulator file)
            const PenK K = pen_consts(pen, lam / scaley, d, al_, ga_, ta_);                  // ref src/oem_dense.cpp:241
            WThr c = wc_thr(K, d);
            c.L = wc_uni(c.L); c.D = wc_uni(c.D); c.rD = wc_uni(c.rD); c.gammad = wc_uni(c.gammad); c.dmg = wc_uni(c.dmg); c.rdmg = wc_uni(c.rdmg);
            c.gm1 = wc_uni(c.gm1); c.gamma = wc_uni(c.gamma); c.dsc = wc_uni(c.dsc); c.rdsc = wc_uni(c.rdsc); c.d = wc_uni(c.d); c.rd = wc_uni(c.rd);
            const double tol = wc_uni(A.tol);
            double tp[NS];                                       // penalty factor x lambda of this lane's columns (read per lambda: short live ranges
#pragma unroll                                                   // around the set-up code are what keeps hipcc out of the accumulator file)
            for (int s = 0; s < NS; ++s) tp[s] = ((colok >> s) & 1u) ? A.pf[mycol0 + s * CW] * c.L : 0.0;
            int it = 0;
            for (;;) {
                if ((tick++ & 127u) == 0u && path_abort_asked(X.abortw)) X.failed = PATH_FAILED_ABORT;
                double bn[NS];
                all_dots(bn);
                bool moving = false;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double u = bn[s] * rn + d * bcur[s];       // ref src/oem_dense.h:520: X'(Y - X beta)/n + d beta
                    const double b = ((colok >> s) & 1u) ? wc_op(u, tp[s], c) : 0.0;
                    const double cu = fabs(b), qo = fabs(bcur[s]);
                    const bool cn = cu > 1e-13, qn = qo > 1e-13;      // ref src/utils.cpp:537-549
                    moving |= (cn != qn) || (cn && qn && fabs(b - bcur[s]) > tol * qo);
                    bcur[s] = b; bn[s] = b;
                }
                WC_STAMP(12);                                        // operator, stop rule
                all_update(bn);
                // r' = Ys - Xs beta': the next iteration's input, or the warm start of the next lambda
                const int any = wc_allreduce<NR, true, WRES_GMAX>(Rsh, Ysh, Pc, Gsh, votes, pub, need1, need2, rn, moving ? 1 : 0, X, tid, w, lane);
                ++it;
                if (any & 2) { left = true; break; }
                const bool conv = !(any & 1);
                if (conv || it >= maxit) {
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        if (storer && ((colok >> s) & 1u)) A.beta[orow * q + mycol0 + s * CW] = bcur[s];
                    // compute.loss (ref src/oem_dense.h:759-770): sum (Ys - Xs beta)^2 -- the residual every workgroup already holds
                    double loss = 1e99;
                    if (A.compute_loss) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < E2; ++k) { const double r = rowok[k] ? Rsh[tid + WNTH * k] : 0.0; t = fma(r, r, t); }
                        loss = wc_block_sum(t, red, rpar, w, lane);
                    }
                    if (tid == 0 && writer) { A.niter[orow] = conv ? it : maxit + 1; A.loss[orow] = loss; }      // ref src/oem_base.h:94-109
                    break;
                }
            }
            if (left) break;
        }
        if (left) break;
    }
#ifdef OEM_PATH_DIAG
    if (tid == 0 && writer) for (int k = 0; k < 16; ++k) g_diag_wcoop[k] = X.acc[k];
#endif
    if (tid == 0 && writer) {
        A.d_out[2] = (double)(__builtin_amdgcn_s_memtime() - t_cyc0);
        A.d_out[3] = (double)(__builtin_amdgcn_s_memrealtime() - t_rt0);
    }
    if (__syncthreads_or(X.failed ? 1 : 0) && tid == 0) A.d_out[6] = 1.0;          // exchange timeout: poison (api.hip: run_paths falls back)
}

// ================================================================================================
// The STREAMED form: the same persistent launch where Xs does NOT fit the registers (n p beyond 3 M entries, e.g. 500 x 20,000).
// G workgroups (three quarters of the CUs) stay for the whole call; a wave walks its columns in chunks of CW (chunk ch of wave
// (wg, w): columns ((ch G + wg) 4 + w) CW ...: every round of chunks is one contiguous sweep over Xs), re-reading the tile from
// L2 / Infinity Cache / HBM in every pass -- ONE read of Xs per iteration, as wide_cols_kernel -- with the all-reduce of the n-vector
// in-kernel instead of a second launch: what the launch-per-iteration engine pays per iteration in launch boundaries and in its
// reduction kernel (5 us of 21 at 500 x 20,000) becomes the two tagged exchanges (2.4 us), and because Xs never changes the first
// tile of the next pass is requested BEFORE the all-reduce and arrives behind it.  Coefficients and penalty factors of a wave's
// chunks live in LDS.  Element-wise operators, compute.loss, columns of <= 512 rows; where it pays: path_wstream_eligible.
template <int NR>
__global__ __launch_bounds__(WNTH) void path_wstream_kernel(PathArgs A, const double *__restrict__ xs, const double *__restrict__ ysv, int n,
                                                             unsigned long long *xchg, int nch)
{
    typedef WCfg<NR> C;
    constexpr int NP = C::NP, CW = C::CW, SH = C::SH, E2 = C::E2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15;
    const int q = A.p, wg = blockIdx.x, G = gridDim.x;
    const unsigned long long t_cyc0 = __builtin_amdgcn_s_memtime(), t_rt0 = __builtin_amdgcn_s_memrealtime();
    double *Rsh = lds + C::OFF_R, *Ysh = lds + C::OFF_Y, *Pc = lds + C::OFF_P, *Gsh = lds + C::OFF_G;
    double *Tal = lds + C::OFF_T, *Tbe = Tal + WCML, *red = lds + C::OFF_X;
    int *votes = reinterpret_cast<int *>(red + 24);
    double *Bst = lds + C::N_DBL, *Pst = Bst + nch * 64;        // [chunk][wave][16 lanes of a row]: coefficient, penalty factor
    const bool writer = wg == 0;
    const double rn = 1.0 / (double)n;
    const int cl = l16 >> SH;
    const bool storer = lane < 16 && (l16 & ((1 << SH) - 1)) == 0;
    auto chunk_base = [&](int ch) { return ((ch * G + wg) * 4 + w) * CW; };
    double x[CW][NR];
    auto load_tile = [&](int ch) __attribute__((always_inline)) {
        const int cb = chunk_base(ch);
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const bool ok = cb + c < q;
            const double *col = xs + (size_t)(ok ? cb + c : 0) * NP;
#pragma unroll
            for (int k = 0; k < NR; ++k) { const double t = col[lane + 64 * k]; x[c][k] = ok ? t : 0.0; }
        }
    };
    load_tile(0);
    for (int ch = 0; ch < nch; ++ch) {
        const int col = chunk_base(ch) + cl;
        if (lane < 16) { Pst[(ch * 4 + w) * 16 + l16] = col < q ? A.pf[col] : 0.0; Bst[(ch * 4 + w) * 16 + l16] = 0.0; }
    }
    for (int j = tid; j < NP + 8; j += WNTH) Rsh[j] = 0.0;
    for (int j = tid; j < NP; j += WNTH) Ysh[j] = j < n ? ysv[j] : 0.0;
    for (int j = tid; j < C::GS; j += WNTH) Gsh[j] = 0.0;

    // ---- the exchange (as path_wcoop_kernel)
    WX X;
    X.G = G; X.wg = wg; X.n = n; X.SL = (n + G - 1) / G; X.row0 = wg * X.SL;
    X.nsl = n - X.row0 < 0 ? 0 : (n - X.row0 < X.SL ? n - X.row0 : X.SL);
    X.stride1 = G * G * X.SL;
    X.qpad = 0;
    X.o2 = 2 * X.stride1 * 16; X.o3 = X.o2 + 2 * NP * 16; X.o4 = X.o3;
    X.rs = __builtin_amdgcn_make_buffer_rsrc((void *)xchg, 0, X.o3, 0x00020000);
    X.epoch = 0; X.failed = 0; X.abortw = A.abort_word;
#ifdef OEM_PATH_DIAG
    for (int k = 0; k < 16; ++k) X.acc[k] = 0;
    X.last = __builtin_amdgcn_s_memtime();
#endif
    int pub[E2];
    unsigned need1 = 0, need2 = 0;
#pragma unroll
    for (int k = 0; k < E2; ++k) {
        const int row = tid + WNTH * k;
        pub[k] = -1;
        if (row < n) {
            const int h = row / X.SL, s = row - h * X.SL;
            pub[k] = (h == wg) ? -2 - (wg * X.SL + s) : (h * G + wg) * X.SL + s;
            if (h != wg) need2 |= 1u << k;
        }
    }
#pragma unroll
    for (int k = 0; k < C::E1; ++k) {
        const int e = tid + WNTH * k;
        if (e < G * X.SL) {
            const int g = e / X.SL, s = e - g * X.SL;
            if (g != wg && s < X.nsl) need1 |= 1u << k;
        }
    }
    __syncthreads();
    int rpar = 0;

    // one pass over this wave's chunks: both products of every chunk from ONE read of its tile.  EIG: z = Xs'v, part of Xs z.
    // OEM: beta' = T(Xs'r / n + d beta), part of Xs beta'.  Returns "some coefficient of this lane's columns still moving".
    auto pass = [&](auto OEM_, const WThr &c, double d) __attribute__((always_inline)) {
        constexpr bool OEM = decltype(OEM_)::value;
        double rr[NR], acc[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) { rr[k] = Rsh[lane + 64 * k]; acc[k] = 0.0; }
        bool moving = false;
        for (int ch = 0; ch < nch; ++ch) {
            if (ch > 0) load_tile(ch);                           // (chunk 0 came in behind the previous all-reduce)
            double s[CW];
#pragma unroll
            for (int cc = 0; cc < CW; ++cc) {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int k = 0; k < NR; k += 2) { a0 = fma(x[cc][k], rr[k], a0); if (k + 1 < NR) a1 = fma(x[cc][k + 1], rr[k + 1], a1); }
                s[cc] = a0 + a1;
            }
            const double dot = wc_colsum<CW>(s, lane);
            double bn[1];
            if (OEM) {
                const int idx = (ch * 4 + w) * 16 + l16;
                const double bo = Bst[idx], tp = Pst[idx] * c.L;
                const double u = dot * rn + d * bo;              // ref src/oem_dense.h:520: X'(Y - X beta)/n + d beta
                const double b = (chunk_base(ch) + cl < q) ? wc_op(u, tp, c) : 0.0;
                const double cu = fabs(b), qo = fabs(bo);
                const bool cn = cu > 1e-13, qn = qo > 1e-13;      // ref src/utils.cpp:537-549
                moving |= (cn != qn) || (cn && qn && fabs(b - bo) > A.tol * qo);
                Bst[idx] = b;                                    // (the lanes that share a column store the same value)
                bn[0] = b;
            } else bn[0] = dot;
            const unsigned long long nz = __ballot(bn[0] != 0.0);
            dpp_hazard_fence(bn);
            WcUpd<0, NR, CW, SH>::run(acc, bn[0], x, nz);
        }
        if (nch > 1) load_tile(0);                               // Xs never changes: the next pass's first tile arrives behind the all-reduce
#pragma unroll
        for (int k = 0; k < NR; ++k) Pc[w * NP + lane + 64 * k] = acc[k];
        return moving;
    };

    // ---- eigenvalue step: Lanczos on Xs Xs'/n (ref src/oem_dense.h:476-498), the vector updates replicated per workgroup
    double v[E2], vp[E2], wv[E2];
    bool rowok[E2];
    {
        double nn = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) {
            const unsigned j = tid + WNTH * k;
            rowok[k] = (int)j < n;
            const unsigned h = j * 2654435761u + 12345u;                 // deterministic non-structured start
            v[k] = rowok[k] ? ((double)(h >> 8) * (1.0 / 16777216.0) - 0.5) : 0.0;
            vp[k] = 0.0;
            nn = fma(v[k], v[k], nn);
        }
        nn = 1.0 / sqrt(wc_block_sum(nn, red, rpar, w, lane));
#pragma unroll
        for (int k = 0; k < E2; ++k) v[k] *= nn;
    }
    int msteps = n < WCML ? n : WCML;
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
    const WThr nothr = {};
    int nst = 0;
    double bprev = 0.0, theta = 0.0, theta_prev = -__builtin_inf(), mv_prev = __builtin_inf();
    bool have_theta = false;
    for (int j = 0; j < msteps; ++j) {
#pragma unroll
        for (int k = 0; k < E2; ++k) if (rowok[k]) Rsh[tid + WNTH * k] = v[k];
        __syncthreads();
        (void)pass(std::false_type{}, nothr, 0.0);
        (void)wc_allreduce<NR, false>(Rsh, Ysh, Pc, Gsh, votes, pub, need1, need2, rn, 0, X, tid, w, lane);
        double al = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) { wv[k] = rowok[k] ? Rsh[tid + WNTH * k] : 0.0; al = fma(v[k], wv[k], al); }
        al = wc_block_sum(al, red, rpar, w, lane);
        double bb = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) {
            wv[k] = (wv[k] - al * v[k]) - bprev * vp[k];
            bb = fma(wv[k], wv[k], bb);
        }
        double ib;
        sqrt_rsqrt(wc_block_sum(bb, red, rpar, w, lane), bb, ib);
        if (tid == 0) { Tal[j] = al; Tbe[j] = bb; }
        nst = j + 1;
        if (!(bb > 1e-13 * fabs(al))) break;                        // invariant subspace reached: T is exact
        if (lanczos_check_due(nst) && nst < msteps) {
            const double th = top_ritz(nst, theta_prev);
            if (lanczos_converged(th, theta_prev, mv_prev)) { theta = th; have_theta = true; break; }
        }
#pragma unroll
        for (int k = 0; k < E2; ++k) { vp[k] = v[k]; v[k] = wv[k] * ib; }
        bprev = bb;
    }
    if (!have_theta) { __syncthreads(); theta = top_ritz(nst, theta_prev); }
    const double d = theta * 1.005;                                  // ref src/oem_dense.h:498
    if (tid == 0 && writer) { A.d_out[0] = d; A.d_out[1] = theta; A.d_out[4] = (double)nst; A.d_out[5] = (!have_theta && nst >= msteps && nst < n) ? 1.0 : 0.0; }

    // ---- lambda grid constants (ref src/oem_dense.cpp:175-192)
    const double scaley = A.yscale ? A.stats[1] : 1.0;
    const int nl = A.nl;
    double lmax = 0.0;
    {
        double m = 0.0;
        for (int j0 = 0; j0 < q; j0 += 8 * WNTH) {
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = j0 + tid + WNTH * k; t[k] = (j < q && j >= A.lmax_from) ? fabs((A.lmax_xy ? A.lmax_xy : A.xy)[j]) : 0.0; }
#pragma unroll
            for (int k = 0; k < 8; ++k) m = fmax(m, t[k]);
        }
        m = wave_max(m);
        if (lane == 0) red[12 + w] = m;
        __syncthreads();
        lmax = fmax(fmax(red[12], red[13]), fmax(red[14], red[15])) * scaley;
    }
    const double llo = log(lmax), lhi = log(A.lambda_min_ratio * lmax);
    const double lstep = nl > 1 ? (lhi - llo) / (double)(nl - 1) : 0.0;
    const bool lflip = fabs(lhi) < fabs(llo);

    bool left = false;                                           // the host's abort word was seen (PathArgs::abort_word): every loop is left
    unsigned tick = 0u;
    for (int pp = A.pen_lo; pp < A.pen_hi; ++pp) {
        const int pen = A.penalty[pp];
        const int nlam = (pen == OEMGPU_OLS) ? 1 : nl;
        const bool isnet = pen_is_net(pen);
        const int maxit = A.maxit;
        // cold start (ref src/oem_dense.cpp:243-244): beta = 0, so the residual is Ys
        __syncthreads();
        for (int ch = 0; ch < nch; ++ch) if (lane < 16) Bst[(ch * 4 + w) * 16 + l16] = 0.0;
#pragma unroll
        for (int k = 0; k < E2; ++k) if (rowok[k]) Rsh[tid + WNTH * k] = Ysh[tid + WNTH * k];
        __syncthreads();
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
            const WThr c = wc_thr(K, d);
            int it = 0;
            for (;;) {
                if ((tick++ & 127u) == 0u && path_abort_asked(X.abortw)) X.failed = PATH_FAILED_ABORT;
                const bool moving = pass(std::true_type{}, c, d);
                const int any = wc_allreduce<NR, true>(Rsh, Ysh, Pc, Gsh, votes, pub, need1, need2, rn, moving ? 1 : 0, X, tid, w, lane);
                ++it;
                if (any & 2) { left = true; break; }
                const bool conv = !(any & 1);
                if (conv || it >= maxit) {
                    if (storer)
                        for (int ch = 0; ch < nch; ++ch) {
                            const int col = chunk_base(ch) + cl;
                            if (col < q) A.beta[orow * q + col] = Bst[(ch * 4 + w) * 16 + l16];
                        }
                    double loss = 1e99;
                    if (A.compute_loss) {                           // ref src/oem_dense.h:759-770: sum (Ys - Xs beta)^2
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < E2; ++k) { const double r = rowok[k] ? Rsh[tid + WNTH * k] : 0.0; t = fma(r, r, t); }
                        loss = wc_block_sum(t, red, rpar, w, lane);
                    }
                    if (tid == 0 && writer) { A.niter[orow] = conv ? it : maxit + 1; A.loss[orow] = loss; }      // ref src/oem_base.h:94-109
                    break;
                }
            }
            if (left) break;
        }
        if (left) break;
    }
    if (tid == 0 && writer) {
        A.d_out[2] = (double)(__builtin_amdgcn_s_memtime() - t_cyc0);
        A.d_out[3] = (double)(__builtin_amdgcn_s_memrealtime() - t_rt0);
    }
    if (__syncthreads_or(X.failed ? 1 : 0) && tid == 0) A.d_out[6] = 1.0;
}

template <int NR> int wstream_launch(hipStream_t s, const PathArgs &a, const WideArgs &wd, int G, int nch)
{
    typedef WCfg<NR> C;
    const size_t sh = ((size_t)C::N_DBL + 2 * (size_t)nch * 64) * sizeof(double);
    if (sh > 64 * 1024 && lds_limit_once(reinterpret_cast<const void *>(&path_wstream_kernel<NR>), sh)) return OEMGPU_ERR_HIP;
    hipLaunchKernelGGL((path_wstream_kernel<NR>), dim3(G), dim3(WNTH), sh, s, a, wd.xs, wd.ys, wd.n, reinterpret_cast<unsigned long long *>(wd.scratch), nch);
    OEM_HIP(hipGetLastError());
    if (sw().OEM_WCOOP_FAKE_TIMEOUT.set) OEM_HIP(hipMemsetAsync(a.d_out + 6, 0xFF, sizeof(double), s));      // tests: the host's fallback
    return 0;
}

static size_t wcoop_gen_lds_doubles(int q, int ng)
{
    const size_t qp = (size_t)((q + 8 + 1) & ~1), ngp = (size_t)((ng + 2) & ~1);
    return qp + ngp + (qp + ngp + 2 + qp + ngp + qp + 2 + 1) / 2 + 2;       // Ush | GW | ints: gid, gstart, gidx, gzero, need list + count
}
template <int NR, bool GEN, bool ACC> int wcoop_launch_as(hipStream_t s, const PathArgs &a, const WideArgs &wd, int G, int sets, size_t set_stride, const int *cstart)
{
    typedef WCfg<NR> C;
    const size_t sh = ((size_t)C::N_DBL + (GEN ? wcoop_gen_lds_doubles(a.p, a.ngroups) : 0)) * sizeof(double);
    if (sh > 64 * 1024 && lds_limit_once(reinterpret_cast<const void *>(&path_wcoop_kernel<NR, GEN, ACC>), sh)) return OEMGPU_ERR_HIP;
    hipLaunchKernelGGL((path_wcoop_kernel<NR, GEN, ACC>), dim3(G, sets), dim3(WNTH), sh, s, a, wd.xs, wd.ys, wd.n,
                       reinterpret_cast<unsigned long long *>(wd.scratch), (long long)set_stride, GEN ? cstart : (const int *)nullptr);
    OEM_HIP(hipGetLastError());
    if (sw().OEM_WCOOP_FAKE_TIMEOUT.set) OEM_HIP(hipMemsetAsync(a.d_out + 6, 0xFF, sizeof(double), s));      // tests: the host's fallback
    return 0;
}
static bool wcoop_general(const PathArgs &a) { return a.ngroups != 0; }
template <int NR> int wcoop_launch(hipStream_t s, const PathArgs &a, const WideArgs &wd, int G, int sets, size_t set_stride, const int *cstart)
{
    if (wcoop_general(a))
        return a.accelerate ? wcoop_launch_as<NR, true, true>(s, a, wd, G, sets, set_stride, cstart) : wcoop_launch_as<NR, true, false>(s, a, wd, G, sets, set_stride, cstart);
    return a.accelerate ? wcoop_launch_as<NR, false, true>(s, a, wd, G, sets, set_stride, cstart) : wcoop_launch_as<NR, false, false>(s, a, wd, G, sets, set_stride, cstart);
}
// dynamic LDS of the kernel for this call (bytes)
template <int NR> size_t wcoop_lds_bytes(const PathArgs &a)
{
    return ((size_t)WCfg<NR>::N_DBL + (wcoop_general(a) ? wcoop_gen_lds_doubles(a.p, a.ngroups) : 0)) * sizeof(double);
}

}  // namespace

#ifdef OEM_PATH_DIAG
extern "C" __attribute__((visibility("default"))) int oemgpu_diag_read_wcoop(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_wcoop), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

int path_wcoop_workgroups(int n, int p)
{
    const WideLayout L = wide_layout(n);
    if (L.nb != 1 || L.nr < 1 || L.nr > 16) return 0;          // (taller columns: two per wave, p <= 1024 < n -- never p >= n)
    return (p + 4 * wc_cw(L.nr) - 1) / (4 * wc_cw(L.nr));
}

int path_wcoop_cpg(int n)                                        // columns per workgroup
{
    const WideLayout L = wide_layout(n);
    return (L.nb != 1 || L.nr < 1 || L.nr > 16) ? 0 : 4 * wc_cw(L.nr);
}
// a partition of the columns cut at group boundaries may need a few more workgroups than p / (4 CW): at most this many
int path_wcoop_max_workgroups(int n, int p)
{
    const int G = path_wcoop_workgroups(n, p);
    if (G < 1) return 0;
    const int g = G + (G / 4 > 2 ? G / 4 : 2);
    return g < WCOOP_GMAX ? g : WCOOP_GMAX;
}
// the exchanges of ONE set of G workgroups, both parities (16-byte pairs), as doubles of scratch
static size_t wcoop_set_doubles(int n, int p, int G)
{
    if (G < 1 || G > WCOOP_GMAX) return 0;
    const WideLayout L = wide_layout(n);
    const size_t SL = ((size_t)n + G - 1) / G;
    return 2 * ((size_t)G * G * SL * 2) + 2 * ((size_t)L.npad() * 2) + 2 * ((size_t)G * 4 * wc_cw(L.nr) * 2) + 2 * ((size_t)G * 2) + 64;
}
// What the scratch is sized for (wide_scratch_doubles): every partition the launch may pick -- G' workgroups for G <= G' <= Gx (a
// partition cut at group boundaries needs a few more than p / (4 CW)) -- with as many sets as 192 CUs hold.  A set's size is NOT
// monotone in G' (2 G'^2 ceil(n / G') pairs: the slice length drops when G' passes a divisor of n), so the bound is the maximum
// over the range, not the value at Gx (n = 100, p = 6,300 asked for 104,724 doubles of the 93,072 sized at Gx and the
// launch refused itself).
static int wcoop_sized_sets(int G) { int s = 192 / G; if (s > WCOOP_MAX_SETS) s = WCOOP_MAX_SETS; return s < 1 ? 1 : s; }
// what a launch of `sets` sets of G workgroups needs (oemgpu_selftest_plan holds it against the scratch the callers allocate)
size_t path_wcoop_launch_doubles(int n, int p, int G, int sets) { return wcoop_set_doubles(n, p, G) * (size_t)(sets > 0 ? sets : 1); }
size_t path_wcoop_xchg_doubles(int n, int p)
{
    const int G = path_wcoop_workgroups(n, p), Gx = path_wcoop_max_workgroups(n, p);
    if (G < 1 || G > WCOOP_GMAX) return 0;
    size_t m = 0;
    for (int g = G; g <= Gx; ++g) {
        const size_t b = wcoop_set_doubles(n, p, g) * (size_t)wcoop_sized_sets(g);
        if (b > m) m = b;
    }
    return m;
}
// workgroup sets (one per penalty) that may run side by side: all of them resident at once, on three quarters of the CUs at most,
// and never more than the scratch was sized for (a device with more than 256 CUs would otherwise ask for more: ADVICE r3).
// 0: not even one set of G workgroups fits the scratch (the caller then takes the launch-per-iteration engine).
int path_wcoop_sets(int n, int p, int npen, int num_cu, int G)
{
    if (G < 1) G = path_wcoop_workgroups(n, p);
    if (G < 1) return 1;
    int s = (num_cu * 3 / 4) / G;
    if (s > WCOOP_MAX_SETS) s = WCOOP_MAX_SETS;
    if (s > npen) s = npen;
    if (sw().OEM_WCOOP_ONE_SET.set) s = 1;
    if (s < 1) s = 1;
    const size_t have = path_wcoop_xchg_doubles(n, p), one = wcoop_set_doubles(n, p, G);
    if (one == 0 || one > have) return 0;
    while (s > 1 && one * (size_t)s > have) --s;
    return s;
}
// host-only check of the two functions above against the launch's own test (tests/test_host_api.py sweeps it without a GPU):
// 0 if every partition size the launch may pick gets at least one set and never more than were sized
extern "C" __attribute__((visibility("default"))) int oemgpu_selftest_wcoop_sizing(int32_t n, int32_t p, int32_t npen, int32_t num_cu)
{
    const int G = path_wcoop_workgroups(n, p), Gx = path_wcoop_max_workgroups(n, p);
    if (G < 1 || G > WCOOP_GMAX) return 0;
    const size_t have = path_wcoop_xchg_doubles(n, p);
    for (int g = G; g <= Gx; ++g) {
        const int s = path_wcoop_sets(n, p, npen, num_cu, g);
        if (s < 1) return g;                                         // a partition the launch would have to refuse
        if (wcoop_set_doubles(n, p, g) * (size_t)s > have) return -g;
    }
    return 0;
}

// OEM_NO_WCOOP=1: the launch-per-iteration engine
bool path_wcoop_eligible(const PathArgs &a, const WideArgs &wd)
{
    const bool off = sw().OEM_NO_WCOOP.set;           // (read per call: the tests hold the two engines against each other)
    const int maxg = WCOOP_GMAX;
    if (off || wd.lay.nb != 1) return false;
    if (a.sinv || a.nbatch > 1 || a.pen_split) return false;
    const int G = path_wcoop_workgroups(wd.n, a.p);
    if (G < 1 || G > maxg) return false;
    if (sw().OEM_WCOOP_NO_GENERAL.set && (wcoop_general(a) || a.accelerate)) return false;
    if (wcoop_general(a)) {                                      // group operators keep u (by column) and the group tables in LDS
        if (a.p > 8192) return false;                            // (the gather masks hold 32 x 256 columns)
        size_t lds = 0;
        switch (wd.lay.nr) {
        case 1: lds = wcoop_lds_bytes<1>(a); break; case 2: lds = wcoop_lds_bytes<2>(a); break; case 3: lds = wcoop_lds_bytes<3>(a); break;
        case 4: lds = wcoop_lds_bytes<4>(a); break; case 6: lds = wcoop_lds_bytes<6>(a); break; case 8: lds = wcoop_lds_bytes<8>(a); break;
        case 12: lds = wcoop_lds_bytes<12>(a); break; case 16: lds = wcoop_lds_bytes<16>(a); break; default: return false;
        }
        if (lds > 150 * 1024) return false;
    }
    return true;
}

int launch_path_wcoop(hipStream_t s, const PathArgs &a, const WideArgs &wd, int sets, const int *cstart, int G)
{
    if (!cstart) G = path_wcoop_workgroups(wd.n, a.p);
    if (G < 1 || G > WCOOP_GMAX || G > path_wcoop_max_workgroups(wd.n, a.p)) { set_error("internal: wide cooperating engine asked for %d workgroups", G); return OEMGPU_ERR_INTERNAL; }
    const size_t set_stride = wcoop_set_doubles(wd.n, a.p, G);
    if (sets < 1 || set_stride * (size_t)sets > path_wcoop_xchg_doubles(wd.n, a.p)) { set_error("internal: wide cooperating engine, %d sets", sets); return OEMGPU_ERR_INTERNAL; }
    OEM_HIP(hipMemsetAsync(wd.scratch, 0, sizeof(double) * set_stride * (size_t)sets, s));                 // the tags must start at 0
    OEM_HIP(hipMemsetAsync(a.d_out, 0, sizeof(double) * D_OUT_LEN, s));                                  // [6]: only a timed-out workgroup writes it
    switch (wd.lay.nr) {
    case 1: return wcoop_launch<1>(s, a, wd, G, sets, set_stride, cstart);
    case 2: return wcoop_launch<2>(s, a, wd, G, sets, set_stride, cstart);
    case 3: return wcoop_launch<3>(s, a, wd, G, sets, set_stride, cstart);
    case 4: return wcoop_launch<4>(s, a, wd, G, sets, set_stride, cstart);
    case 6: return wcoop_launch<6>(s, a, wd, G, sets, set_stride, cstart);
    case 8: return wcoop_launch<8>(s, a, wd, G, sets, set_stride, cstart);
    case 12: return wcoop_launch<12>(s, a, wd, G, sets, set_stride, cstart);
    case 16: return wcoop_launch<16>(s, a, wd, G, sets, set_stride, cstart);
    default: break;
    }
    set_error("internal: wide cooperating engine, nr = %d", wd.lay.nr);
    return OEMGPU_ERR_INTERNAL;
}

// ---- the resident form with columns in the accumulator file too (path_wres_kernel)
static int wres_cpg(int nr) { return 4 * wc_cw(nr) * (1 + 128 / (wc_cw(nr) * nr)); }
static bool wres_nr_built(int nr) { return nr == 1 || nr == 2 || nr == 3 || nr == 4 || nr == 6 || nr == 8 || nr == 12 || nr == 16; }
int path_wres_workgroups(int n, int p)
{
    const WideLayout L = wide_layout(n);
    if (L.nb != 1 || !wres_nr_built(L.nr)) return 0;
    return (p + wres_cpg(L.nr) - 1) / wres_cpg(L.nr);
}
size_t path_wres_xchg_doubles(int n, int p)
{
    const int G = path_wres_workgroups(n, p);
    if (G < 1 || G > WRES_GMAX) return 0;
    const WideLayout L = wide_layout(n);
    const size_t SL = ((size_t)n + G - 1) / G;
    return 2 * ((size_t)G * G * SL * 2) + 2 * ((size_t)L.npad() * 2) + 64;
}
// Where the VGPR-only form (path_wcoop_kernel) cannot hold Xs: element-wise operators, one instance.  max_wg: what this device may
// keep resident at once (api.hip: all CUs but a few).  OEM_NO_WRES=1: the streamed / launch-per-iteration engines; OEM_WRES=1:
// also where path_wcoop_kernel would have run (tests).
bool path_wres_eligible(const PathArgs &a, const WideArgs &wd, int max_wg)
{
    if (sw().OEM_NO_WRES.set || sw().OEM_NO_WCOOP.set || wd.lay.nb != 1 || !wres_nr_built(wd.lay.nr)) return false;
    if (a.ngroups != 0 || a.accelerate || a.sinv || a.nbatch > 1 || a.pen_split) return false;
    const int G = path_wres_workgroups(wd.n, a.p);
    return G >= 1 && G <= WRES_GMAX && G <= max_wg;
}
template <int NR> static int wres_launch(hipStream_t s, const PathArgs &a, const WideArgs &wd, int G)
{
    typedef WCfg<NR, WRES_GMAX> C;
    const size_t sh = (size_t)C::N_DBL * sizeof(double);
    if (sh > 64 * 1024 && lds_limit_once(reinterpret_cast<const void *>(&path_wres_kernel<NR>), sh)) return OEMGPU_ERR_HIP;
    hipLaunchKernelGGL((path_wres_kernel<NR>), dim3(G), dim3(WNTH), sh, s, a, wd.xs, wd.ys, wd.n, reinterpret_cast<unsigned long long *>(wd.scratch));
    OEM_HIP(hipGetLastError());
    if (sw().OEM_WCOOP_FAKE_TIMEOUT.set) OEM_HIP(hipMemsetAsync(a.d_out + 6, 0xFF, sizeof(double), s));      // tests: the host's fallback
    return 0;
}
int launch_path_wres(hipStream_t s, const PathArgs &a, const WideArgs &wd)
{
    const int G = path_wres_workgroups(wd.n, a.p);
    const size_t need = path_wres_xchg_doubles(wd.n, a.p);
    if (G < 1 || need == 0) { set_error("internal: resident wide engine asked for %d workgroups", G); return OEMGPU_ERR_INTERNAL; }
    OEM_HIP(hipMemsetAsync(wd.scratch, 0, sizeof(double) * need, s));                                    // the tags must start at 0
    OEM_HIP(hipMemsetAsync(a.d_out, 0, sizeof(double) * D_OUT_LEN, s));                                  // [6]: only a timed-out workgroup writes it
    switch (wd.lay.nr) {
    case 1: return wres_launch<1>(s, a, wd, G);
    case 2: return wres_launch<2>(s, a, wd, G);
    case 3: return wres_launch<3>(s, a, wd, G);
    case 4: return wres_launch<4>(s, a, wd, G);
    case 6: return wres_launch<6>(s, a, wd, G);
    case 8: return wres_launch<8>(s, a, wd, G);
    case 12: return wres_launch<12>(s, a, wd, G);
    case 16: return wres_launch<16>(s, a, wd, G);
    default: break;
    }
    set_error("internal: resident wide engine, nr = %d", wd.lay.nr);
    return OEMGPU_ERR_INTERNAL;
}

// ---- the streamed form (path_wstream_kernel): G workgroups whatever p is
static size_t wstream_xchg_doubles_for(int n, int G, int nr)
{
    const size_t SL = ((size_t)n + G - 1) / G;
    return 2 * ((size_t)G * G * SL * 2) + 2 * ((size_t)64 * nr * 2) + 64;
}
size_t path_wstream_xchg_doubles(int n)
{
    const WideLayout L = wide_layout(n);
    if (L.nb != 1 || L.nr < 1 || L.nr > 8) return 0;
    size_t m = 0;                                                // (G^2 ceil(n / G) is not monotone in G: a device with fewer CUs runs fewer workgroups)
    for (int g = 1; g <= WCOOP_GMAX; ++g) { const size_t b = wstream_xchg_doubles_for(n, g, L.nr); if (b > m) m = b; }
    return m;
}
static int wstream_chunks(int p, int G, int nr) { const int per = 4 * G * wc_cw(nr); return (p + per - 1) / per; }
// Where it is taken is a measurement (tools/wstream_time.py, streamed against the launch-per-iteration engine, us per iteration): columns
// of <= 128 rows 128 x 40,000 12.2 / 16.8, 64 x 100,000 18.0 / 23.8, 128 x 200,000 35.5 / 46.3 (there the launches are latency-bound:
// a column is one or two registers per lane); taller columns only while Xs is small -- 500 x 8,000 10.9 / 12.3, 250 x 16,000 11.9 /
// 13.0 -- because four persistent waves with a 64-double tile keep fewer loads in flight than the sixteen waves per CU of
// wide_cols_kernel: 200 x 30,000 18.4 / 17.4, 200 x 100,000 52 / 37, and 1,000-row columns spill (1,000 x 8,000 37 / 19: not built).
// OEM_NO_WSTREAM=1: the launch-per-iteration engine; OEM_WSTREAM=1: wherever it can run (tests).
bool path_wstream_eligible(const PathArgs &a, const WideArgs &wd, int G)
{
    if (sw().OEM_NO_WSTREAM.set || wd.lay.nb != 1 || wd.lay.nr < 1 || wd.lay.nr > 8 || G < 1 || G > WCOOP_GMAX) return false;
    if (a.ngroups != 0 || a.accelerate || a.sinv || a.nbatch > 1 || a.pen_split) return false;
    const int nch = wstream_chunks(a.p, G, wd.lay.nr);
    if ((size_t)nch * 64 * 2 * sizeof(double) > 64 * 1024) return false;      // coefficients and penalty factors of the chunks in LDS
    if (sw().OEM_WSTREAM.set) return true;
    return wd.lay.nr <= 2 || (long long)wd.lay.npad() * a.p <= 4500000LL;
}
int launch_path_wstream(hipStream_t s, const PathArgs &a, const WideArgs &wd, int G)
{
    const int nch = wstream_chunks(a.p, G, wd.lay.nr);
    OEM_HIP(hipMemsetAsync(wd.scratch, 0, sizeof(double) * wstream_xchg_doubles_for(wd.n, G, wd.lay.nr), s));      // the tags must start at 0
    OEM_HIP(hipMemsetAsync(a.d_out, 0, sizeof(double) * D_OUT_LEN, s));
    switch (wd.lay.nr) {
    case 1: return wstream_launch<1>(s, a, wd, G, nch);
    case 2: return wstream_launch<2>(s, a, wd, G, nch);
    case 3: return wstream_launch<3>(s, a, wd, G, nch);
    case 4: return wstream_launch<4>(s, a, wd, G, nch);
    case 6: return wstream_launch<6>(s, a, wd, G, nch);
    case 8: return wstream_launch<8>(s, a, wd, G, nch);
    default: break;
    }
    set_error("internal: streamed cooperating engine, nr = %d", wd.lay.nr);
    return OEMGPU_ERR_INTERNAL;
}

}  // namespace oemgpu
