// path_wres.hip -- p >= n, the persistent engines beyond the vector registers (split from path_wcoop.hip, whose header describes the common
// design: columns over workgroups, the n-vector all-reduced in two tagged exchanges, the replicated lambda / penalty state machine):
//   path_wres_kernel     Xs up to ~11 M entries resident: more column sets of every wave in the ACCUMULATOR file
//   path_wstream_kernel  short columns beyond that: the same persistent launch re-reading its column tiles every iteration
#include "path_wcoop_dev.hpp"

namespace oemgpu {

namespace {

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

}  // namespace

#ifdef OEM_PATH_DIAG
extern "C" __attribute__((visibility("default"))) int oemgpu_diag_read_wres(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_wcoop), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

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