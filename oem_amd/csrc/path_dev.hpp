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
template <int N, typename F> __device__ __forceinline__ void static_for_dev(F &&f)
{
    if constexpr (N > 0) { static_for_dev<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}
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
    else if constexpr (N == 4) asm volatile("s_nop 1" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]));
    else { static_assert(N == 8, "extend dpp_hazard_fence"); asm volatile("s_nop 1" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), "+v"(B[4]), "+v"(B[5]), "+v"(B[6]), "+v"(B[7])); }
}
// A double kept in the accumulator file: AGPRs a[2 IDX], a[2 IDX + 1], named by inline asm alone (the compiler's own values must fit
// the architectural VGPRs of such a kernel: oem_amd/build.py audits that it never emits a v_accvgpr of its own there).  `a[%n]`, not
// `a%n`: the printer writes immediates above 64 in hex, and `a0x41` does not assemble.
template <int IDX> __device__ __forceinline__ double areg_rd()
{
    unsigned l, h;
    asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\tv_accvgpr_read_b32 %1, a[%3]" : "=v"(l), "=v"(h) : "n"(2 * IDX), "n"(2 * IDX + 1));
    return __hiloint2double((int)h, (int)l);
}
template <int IDX> __device__ __forceinline__ void areg_wr(double x)
{
    asm volatile("v_accvgpr_write_b32 a[%2], %0\n\tv_accvgpr_write_b32 a[%3], %1" ::"v"(__double2loint(x)), "v"(__double2hiint(x)), "n"(2 * IDX), "n"(2 * IDX + 1));
}
template <int K> struct BcFma {
    static __device__ __forceinline__ void fmac(double &acc, const double &b, const double &a)
    {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(a), "n"(K));
    }
};
// Largest eigenvalue of the symmetric tridiagonal (al[0..m), be[0..m-1)) by 64-way multisection of the Sturm count;
// every lane returns the same value (an upper bracket end, so d never undershoots).  ONE wave runs it.
//   * The count uses the determinant recurrence p_k = (a_k - t) p_{k-1} - b_{k-1}^2 p_{k-2} (sign agreements of
//     consecutive p_k) on T scaled by 1/Gershgorin, renormalised by exponent every 8 steps: three FP64 operations per
//     step.  The pivot form q_k = (a_k - t) - b^2 / q_{k-1} costs a ~200-cycle FP64 division per step, which made this
//     routine (9 rounds x 100 steps) a fifth of config 1's whole path kernel.
//   * sab: LDS scratch for the scaled, interleaved coefficients (2 (m + 16) doubles).
//   * lo_hint: a known lower bound of the answer (the value at an earlier Lanczos step; Ritz values only grow), or
//     -inf.  With a hint the first round places its 64 probes geometrically above it, so a nearly converged value is
//     bracketed to a factor of two at once; the following rounds cluster their probes around the secant estimate.
__device__ __forceinline__ double tridiag_max(const double *al, const double *be, int m, int lane, double *sab, double lo_hint,
                                              int *rounds_out = nullptr)
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
    // scale by a power of two (exact: the scaled matrix has the same eigenvalues times sc, bit for bit), norm in [0.5, 1)
    const int enrm = (__double2hiint(nrm) >> 20) & 0x7ff;
    const double sc = (enrm > 30 && enrm < 2000) ? __hiloint2double((2045 - enrm) << 20, 0) : 1.0;
    // scaled coefficients, interleaved {a_k, -b_{k-1}^2}, padded to a multiple of sixteen steps with identity steps
    // (a_k = 2^100, b^2 = 0: p_k = 2^100 p_{k-1} keeps the sign of p_{k-1}; the renormalisation absorbs the factor)
    const int mp = 1 + (m - 1 + 15) / 16 * 16;
    v2d *co = reinterpret_cast<v2d *>(sab);
    for (int j = lane; j < mp; j += 64) {
        const double b = (j > 0 && j < m) ? be[j - 1] * sc : 0.0;
        co[j] = v2d{j < m ? al[j] * sc : 0x1p+100, -(b * b)};
    }
    const bool hinted = lo_hint > lo;
    if (hinted) lo = lo_hint;
    if (!(hi > lo)) return lo;                              // the hint already is the top of the bracket
    // The coefficients are the same for every probe, so they are not loaded per lane: lane L keeps entry 1 + 16 J + (L & 15) of
    // block J (ONE ds_read_b128 per sixteen steps) and step k takes it from lane k of the row through the DPP broadcast of
    // v_fmac_f64_dpp -- p_k = a_k p_{k-1} + (-t p_{k-1} + (-b^2) p_{k-2}): one FMA and two broadcast FMAs per step, two of them dependent.  (Sixteen
    // coefficient pairs per lane and trip, double-buffered, were 64 VGPRs that the compiler parked in AGPRs: 64 v_accvgpr moves
    // per trip, 40 % of the sweep.)  -t p + a p differs from (a - t) p by one rounding of size eps |t p|: a perturbation of
    // a_k by eps ||T||, which is what the bracket's final width allows anyway.
    const v2d *cl = co + 1 + (lane & 15);
    // Probe placement.  A round costs a sweep whatever it learns, so after the first one the probes go where the root is
    // expected: p_m(t) at the two bracket ends (the sweep's last value, with the exponents the renormalisation took out)
    // has opposite signs exactly when the bracket holds one eigenvalue, and then the secant through them is the estimate
    // ts; 32 probes approach ts geometrically from below (ts - dl 2^-(i+1)) and 32 from above, so the new bracket is as
    // narrow as the secant was good (down to 2^-32 of the old one) and never worse than half of it.  A secant round
    // that gains less than 8x is followed by a uniform one.  Brackets come from Sturm counts only: the secant decides
    // where to look, never what is true.  Config 1's five looks: 29 rounds with uniform probes, 17 with these.
    double f_lo = 1.0, f_hi = 1.0;                          // p_m at lo / hi: mantissas ...
    int e_lo = 0, e_hi = 0;                                 // ... and the exponents taken out of them
    bool have_lo = false, have_hi = false, uniform_next = false;
    auto pow2 = [](int e) { return __hiloint2double((1023 + e) << 20, 0); };
    // this lane's probe fractions for the three placements (a taken branch is ~80 cycles: the round below selects, it does not branch)
    const double fac_u = (double)(lane + 1) * (1.0 / 65.0), fac_g = pow2(lane - 64);
    const double fac_s = lane < 32 ? -pow2(-(lane + 1)) : fac_g;
    for (int round = 0; round < 20; ++round) {
        const double w = hi - lo;
        if (!(w > 4.0e-16 * fabs(hi))) break;
        const bool geo = hinted && round == 0;
        int de = e_lo - e_hi;                                // f = mantissa 2^-e
        de = de > 1000 ? 1000 : (de < -1000 ? -1000 : de);
        // f(hi) / f(lo): negative around a simple root (v_rcp_f64 instead of divisions: the result only places probes)
        const double r = (f_hi * __builtin_amdgcn_rcp(f_lo)) * pow2(de);
        const double ts = lo + w * __builtin_amdgcn_rcp(1.0 - r), dl = ts - lo, dh = hi - ts;
        const bool sec = !geo && !uniform_next && have_lo && have_hi && r < 0.0 && r > -1e300 && dl >= 0.0 && dh >= 0.0;
        uniform_next = false;
        // probe positions lo <= th_0 <= ... <= th_63 <= hi
        const double tpos = sec ? ts + (lane < 32 ? dl : dh) * fac_s : lo + w * (geo ? fac_g : fac_u);
        const double t = tpos * sc, nt = -t;
        double pm2 = 1.0, pm1 = co[0].x - t;
        // sign history: one v_alignbit per step shifts the sign bit of p_k into a 32-bit register; sign changes are
        // counted eight steps at a time with a popcount.  An exact zero counts as positive, which keeps the count
        // right: p_k = +0 gives p_{k+1} = -b^2 p_{k-1}, one change over the two steps whichever sign p_{k-1} has.
        unsigned hist = (unsigned)__double2hiint(pm1) >> 31;             // bit 0 = sign(p_1); sign(p_0) = 0
        int neg = hist, esum = 0;
        auto renorm = [&]() {
            neg += __popc((hist ^ (hist >> 1)) & 0xffu);                // changes between p_{k-1} .. p_{k+7}
            // renormalise by exponent: sign counts are scale-free; |a - t| <= 2 (2^100 in the padding), b^2 <= 1
            const int e1 = (__double2hiint(pm1) >> 20) & 0x7ff, e2 = (__double2hiint(pm2) >> 20) & 0x7ff;
            const int e = 1023 - (e1 > e2 ? e1 : e2);
            pm1 = ldexp(pm1, e); pm2 = ldexp(pm2, e);
            esum += e;
        };
        v2d cur = cl[0];
        for (int k = 1; k < mp; k += 16) {
            const v2d nxt = cl[k + 16 < mp ? k + 15 : 0];              // the next block's entries, in flight during this one
            double cc[2] = {cur.x, cur.y};
            dpp_hazard_fence(cc);
            static_for_dev<16>([&](auto Q_) {
                constexpr int q = decltype(Q_)::value;
                double wk = 0.0;
                BcFma<q>::fmac(wk, cc[1], pm2);                          // -b^2 p_{k-2}: off the chain (p_{k-2} is one step old)
                double pn = fma(nt, pm1, wk);                            // the chain: two dependent operations per step
                BcFma<q>::fmac(pn, cc[0], pm1);
                hist = __builtin_amdgcn_alignbit(hist, (unsigned)__double2hiint(pn), 31);   // (hist << 1) | sign(pn)
                pm2 = pm1; pm1 = pn;
                if constexpr (q == 7 || q == 15) renorm();
            });
            cur = nxt;
        }
        const int above = neg < m;                         // an eigenvalue >= th exists
        const int kk = __popcll(__ballot(above));          // monotone in the lane index
        // the new ends are the probes of lanes kk - 1 and kk, and p_m there is what those lanes ended with (an end that did not
        // move keeps what it had)
        const int la = kk > 0 ? kk - 1 : 0, lb = kk < 64 ? kk : 63;
        auto bcast = [](double v, int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
        const double ta = bcast(tpos, la), tb = bcast(tpos, lb), fa = bcast(pm1, la), fb = bcast(pm1, lb);
        const int ea = __builtin_amdgcn_readlane(esum, la), eb = __builtin_amdgcn_readlane(esum, lb);
        const double nlo = kk == 0 ? lo : ta, nhi = kk == 64 ? hi : tb;
        f_lo = kk > 0 ? fa : f_lo; e_lo = kk > 0 ? ea : e_lo; have_lo = have_lo || kk > 0;
        f_hi = kk < 64 ? fb : f_hi; e_hi = kk < 64 ? eb : e_hi; have_hi = have_hi || kk < 64;
        if (sec && !((nhi - nlo) * 8.0 <= w)) uniform_next = true;
        lo = nlo; hi = nhi;
        if (rounds_out) ++*rounds_out;
    }
    return hi;
}

// When the register-resident engines look at the top Ritz value, and when they stop the recurrence.
//   * every 8 steps from step 16 on (an evaluation costs about six Lanczos steps);
//   * stop when the value has moved by <= 1e-14 relative since the last look, or when two successive moves decay so fast that
//     the geometric tail they imply, mv^2 / (mv_prev - mv), is <= OEM_LANCZOS_TAIL_TOL (1e-12) relative (the Ritz value of a Krylov method converges
//     superlinearly, so the tail estimate is on the safe side: on config 1's matrix it says 3e-13 at step 32 where the true
//     error is 1e-14; config 1 stops at 32 steps instead of 56).  The reference's own tolerance is 1e-10
//     (src/oem_dense.h:485-498), and d only sets the step length: a relative 1e-12 in d moves nothing.
__device__ __forceinline__ bool lanczos_check_due(int nst) { return nst >= 16 && (nst & 7) == 0; }
__device__ __forceinline__ bool lanczos_converged(double th, double &theta_prev, double &mv_prev)
{
    const double mv = th - theta_prev, ath = fabs(th);
    bool stop = mv <= 1e-14 * ath;
    if (mv_prev < 1e300 && mv < 0.01 * mv_prev && mv * mv <= OEM_LANCZOS_TAIL_TOL * ath * (mv_prev - mv)) stop = true;
    mv_prev = mv; theta_prev = th;
    return stop;
}

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
