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
#include "path_wcoop_dev.hpp"

namespace oemgpu {

namespace {

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

}  // namespace oemgpu
