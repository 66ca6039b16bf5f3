// path_wcoop_dev.hpp -- what the persistent p >= n engines share (path_wcoop.hip: Xs in vector registers; path_wres.hip: also in the
// accumulator file, and the streamed form): configuration, the tagged all-reduce, the transposed butterflies, the operator constants.
// Internal to those two translation units (an anonymous namespace each: the file was one 1,764-line unit and the long pole of the build).
#pragma once
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
}  // namespace

}  // namespace oemgpu
