// path_dev.hpp -- device helpers shared by the register-resident path engines (path_small.hip, path_coop.hip):
// DPP wave reductions, the Sturm multisection of the Lanczos tridiagonal, the DPP-broadcast FMA and its hazard fence.
#pragma once

#include "common.hpp"

namespace oemgpu {
namespace {

// Wave-wide reductions on the DPP network (no LDS round trips: __shfl_xor lowers to ds_bpermute, ~100 cycles per
// step on this serial path).  Classic GFX9 scan: row_shr 1/2/4/8 inside each 16-lane row, then row_bcast15 and
// row_bcast31 across rows; lane 63 holds the total, returned wave-uniform through an SGPR.  Every wave runs
// the same instructions on the same data, so the result is bit-identical across waves.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v, double identity)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int ilo = __double2loint(identity), ihi = __double2hiint(identity);
    const int rlo = __builtin_amdgcn_update_dpp(ilo, lo, CTRL, ROW_MASK, 0xf, false);
    const int rhi = __builtin_amdgcn_update_dpp(ihi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ double wave_uniform_lane63(double v)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_mov<0x111, 0xf>(v, 0.0);      // row_shr:1
    v += dpp_mov<0x112, 0xf>(v, 0.0);      // row_shr:2
    v += dpp_mov<0x114, 0xf>(v, 0.0);      // row_shr:4
    v += dpp_mov<0x118, 0xf>(v, 0.0);      // row_shr:8
    v += dpp_mov<0x142, 0xa>(v, 0.0);      // row_bcast:15 -> rows 1, 3
    v += dpp_mov<0x143, 0xc>(v, 0.0);      // row_bcast:31 -> rows 2, 3
    return wave_uniform_lane63(v);
}
__device__ __forceinline__ double wave_max(double v)
{
    const double ninf = -__builtin_inf();
    v = fmax(v, dpp_mov<0x111, 0xf>(v, ninf));
    v = fmax(v, dpp_mov<0x112, 0xf>(v, ninf));
    v = fmax(v, dpp_mov<0x114, 0xf>(v, ninf));
    v = fmax(v, dpp_mov<0x118, 0xf>(v, ninf));
    v = fmax(v, dpp_mov<0x142, 0xa>(v, ninf));
    v = fmax(v, dpp_mov<0x143, 0xc>(v, ninf));
    return wave_uniform_lane63(v);
}
// Largest eigenvalue of the symmetric tridiagonal (al[0..m), be[0..m-1)) by 64-way multisection of the Sturm count;
// every lane returns the same value (an upper bracket end, so d never undershoots).  ONE wave runs it.
//   * The count uses the determinant recurrence p_k = (a_k - t) p_{k-1} - b_{k-1}^2 p_{k-2} (sign agreements of
//     consecutive p_k) on T scaled by 1/Gershgorin, renormalised by exponent every 4 steps: one dependent FMA per
//     step.  The pivot form q_k = (a_k - t) - b^2 / q_{k-1} costs a ~200-cycle FP64 division per step, which made this
//     routine (9 rounds x 100 steps) a fifth of config 1's whole path kernel.
//   * sab: LDS scratch for the scaled, interleaved coefficients (2 (m + 16) doubles).
//   * lo_hint: a known lower bound of the answer (the value at an earlier Lanczos step; Ritz values only grow), or
//     -inf.  With a hint the first round places its 64 probes geometrically above it, so a nearly converged value is
//     bracketed to a factor of two at once and two or three uniform rounds finish the job.
__device__ __forceinline__ double tridiag_max(const double *al, const double *be, int m, int lane, double *sab, double lo_hint)
{
    if (m == 1) return al[0];
    double lo = -1e300, hi = -1e300, nrm = 0.0;
    for (int j = lane; j < m; j += 64) {
        const double bl = j > 0 ? fabs(be[j - 1]) : 0.0, br = j < m - 1 ? fabs(be[j]) : 0.0;
        lo = fmax(lo, al[j]);
        hi = fmax(hi, al[j] + bl + br);
        nrm = fmax(nrm, fabs(al[j]) + bl + br);
    }
    lo = wave_max(lo); hi = wave_max(hi); nrm = wave_max(nrm);
    const double sc = (nrm > 0.0 && nrm < 1e300) ? 1.0 / nrm : 1.0;
    // scaled coefficients, interleaved {a_k, b_{k-1}^2}, padded to a multiple of sixteen steps with identity steps
    // (a_k = 2^100, b^2 = 0: p_k = 2^100 p_{k-1} keeps the sign of p_{k-1}; the renormalisation absorbs the factor)
    const int mp = 1 + (m - 1 + 15) / 16 * 16;
    v2d *co = reinterpret_cast<v2d *>(sab);
    for (int j = lane; j < mp; j += 64) {
        const double b = (j > 0 && j < m) ? be[j - 1] * sc : 0.0;
        co[j] = v2d{j < m ? al[j] * sc : 0x1p+100, b * b};
    }
    const bool hinted = lo_hint > lo;
    if (hinted) lo = lo_hint;
    if (!(hi > lo)) return lo;                              // the hint already is the top of the bracket
    for (int round = 0; round < 14; ++round) {
        const double w = hi - lo;
        if (!(w > 4.0e-16 * fabs(hi))) break;
        const bool geo = hinted && round == 0;
        // probe positions lo < th_0 < ... < th_63 < hi: uniform, or (first hinted round) lo + w 2^(lane - 64)
        const double frac = geo ? ldexp(1.0, lane - 64) : (double)(lane + 1) / 65.0;
        const double t = (lo + w * frac) * sc;
        double pm2 = 1.0, pm1 = co[0].x - t;
        // sign history: one v_alignbit per step shifts the sign bit of p_k into a 32-bit register; sign changes are
        // counted eight steps at a time with a popcount.  An exact zero counts as positive, which keeps the count
        // right: p_k = +0 gives p_{k+1} = -b^2 p_{k-1}, one change over the two steps whichever sign p_{k-1} has.
        unsigned hist = (unsigned)__double2hiint(pm1) >> 31;             // bit 0 = sign(p_1); sign(p_0) = 0
        int neg = hist;
        auto eight = [&](const v2d (&c)[8]) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double pn = fma(c[q].x - t, pm1, -(c[q].y * pm2));
                hist = __builtin_amdgcn_alignbit(hist, (unsigned)__double2hiint(pn), 31);   // (hist << 1) | sign(pn)
                pm2 = pm1; pm1 = pn;
            }
            neg += __popc((hist ^ (hist >> 1)) & 0xffu);                // changes between p_{k-1} .. p_{k+7}
            // renormalise by exponent: sign counts are scale-free; |a - t| <= 2 (2^100 in the padding), b^2 <= 1
            const int e1 = (__double2hiint(pm1) >> 20) & 0x7ff, e2 = (__double2hiint(pm2) >> 20) & 0x7ff;
            const int e = 1023 - (e1 > e2 ? e1 : e2);
            pm1 = ldexp(pm1, e); pm2 = ldexp(pm2, e);
        };
        // sixteen steps per trip on two register sets that take turns: the other set's coefficients are in flight while
        // this one's dependent FMAs run, and nothing is copied (one taken branch per sixteen steps)
        v2d ca[8], cb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ca[q] = co[1 + q];
        for (int k = 1; k < mp; k += 16) {
#pragma unroll
            for (int q = 0; q < 8; ++q) cb[q] = co[k + 8 + q];
            eight(ca);
            if (k + 16 < mp) {
#pragma unroll
                for (int q = 0; q < 8; ++q) ca[q] = co[k + 16 + q];
            }
            eight(cb);
        }
        const int above = neg < m;                         // an eigenvalue >= th exists
        const int kk = __popcll(__ballot(above));          // monotone in the lane index
        const double flo = kk == 0 ? 0.0 : (geo ? ldexp(1.0, kk - 1 - 64) : (double)kk / 65.0);
        const double fhi = kk == 64 ? 1.0 : (geo ? ldexp(1.0, kk - 64) : (double)(kk + 1) / 65.0);
        const double nlo = kk == 0 ? lo : lo + w * flo;
        const double nhi = kk == 64 ? hi : lo + w * fhi;
        lo = nlo; hi = nhi;
    }
    return hi;
}

// A VALU write of a VGPR needs two wait states before a DPP instruction reads it, and hipcc pads no hazards for
// inline asm.  The nop must be TIED to the registers: a bare asm volatile("s_nop") only orders memory operations, so
// the compiler may sink the producing VALU instruction below it, straight in front of the DPP read (seen as a
// run-to-run varying eigenvalue in a diagnostic build).  The "+v" operands make the producers precede the nop and the
// consumers follow it.
template <int N> __device__ __forceinline__ void dpp_hazard_fence(double (&B)[N])
{
    if constexpr (N == 1) asm volatile("s_nop 1" : "+v"(B[0]));
    else if constexpr (N == 2) asm volatile("s_nop 1" : "+v"(B[0]), "+v"(B[1]));
    else if constexpr (N == 3) asm volatile("s_nop 1" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]));
    else { static_assert(N == 4, "extend dpp_hazard_fence"); asm volatile("s_nop 1" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3])); }
}
template <int K> struct BcFma {
    static __device__ __forceinline__ void fmac(double &acc, const double &b, const double &a)
    {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(a), "n"(K));
    }
};
// s = sqrt(x), r = 1 / sqrt(x) for the Lanczos normalisation: v_rsq_f64 + two coupled Goldschmidt steps (~10
// dependent FP64 ops; the library sqrt followed by a division is ~45).  Outside [1e-200, 1e200] the slow pair.
__device__ __forceinline__ void sqrt_rsqrt(double x, double &s, double &r)
{
    // x is the same in every lane; the test is made wave-uniform so that it is ONE untaken scalar branch
    if (__builtin_expect(__all(x > 1e-200 && x < 1e200), 1)) {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = 0.5 * y;
        double e = fma(-h, g, 0.5);
        g = fma(g, e, g); h = fma(h, e, h);
        e = fma(-h, g, 0.5);
        g = fma(g, e, g); h = fma(h, e, h);
        e = fma(-g, g, x);                                         // last correction of the root
        g = fma(e, h, g);
        s = g; r = 2.0 * h;
    } else {
        s = sqrt(x); r = 1.0 / s;
    }
}
// the same per lane (x differs between lanes: no wave-uniform fast-path test)
__device__ __forceinline__ void sqrt_rsqrt_lane(double x, double &s, double &r)
{
    if (x > 1e-200 && x < 1e200) {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = 0.5 * y;
        double e = fma(-h, g, 0.5);
        g = fma(g, e, g); h = fma(h, e, h);
        e = fma(-h, g, 0.5);
        g = fma(g, e, g); h = fma(h, e, h);
        e = fma(-g, g, x);
        g = fma(e, h, g);
        s = g; r = 2.0 * h;
    } else {
        s = sqrt(x); r = 1.0 / s;
    }
}

}  // namespace
}  // namespace oemgpu
