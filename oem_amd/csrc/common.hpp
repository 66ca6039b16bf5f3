// common.hpp -- shared declarations of liboemgpu (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/oemgpu.h"
#include "switches.hpp"

namespace oemgpu {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ error plumbing
void set_error(const char *fmt, ...);
#define OEM_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            oemgpu::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return OEMGPU_ERR_HIP;                                                            \
        }                                                                                     \
    } while (0)

// ------------------------------------------------------------------ Gram / moments (gram.hip)
// Tiles are 16x16 (one v_mfma_f64_16x16x4_f64 accumulator); only tiles I >= J are built.
struct GramPlan {
    int p;          // columns of X
    int ntc;        // tile columns = ceil(p/16)
    int ntile;      // ntc*(ntc+1)/2
    int tri;        // 1: one wave holds the whole lower triangle (ntc <= 7); 0: 4x4 tile blocks
    int nblk;       // tile blocks per row chunk (1 when tri)
    int wd_units;   // gram_wd.hip: diagonal units of 16 tile columns per row chunk (1: p <= 256; k: 16 k tile columns, with k (k - 1) off-diagonal 8 x 16-tile blocks)
    int wd;         // 4 / 3: 15-16 / 11-12 tile columns, one eight-wave workgroup per row chunk (gram_wd.hip, groups of 4 / 3 tile columns) instead of the super-blocks below; 0: those
    int n8, n6, n4; // shared-slab kernel: super-block rows of 8, 6 and (at most one) 4 tile columns (gram_sb_deal; 0 when tri)
    int nchunk;     // row chunks (workgroups along rows)
    int steps;      // 64-row steps per chunk
    size_t tpart_doubles;   // nchunk * ntile * 256
    size_t vpart_doubles;   // nchunk * (2*16*ntc + 4)
};
GramPlan gram_plan(int64_t n, int p, int num_cu);
GramPlan gram_plan_bound(int64_t nmax, int p, int num_cu);   // sizes that hold the plan of any n <= nmax
void gram_sb_deal(int ntc, int *n8, int *n6, int *n4);

int launch_shift_sums(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, double *sums);
int launch_gram(hipStream_t s, const GramPlan &pl, const double *x, int64_t n, int64_t ld, const double *y,
                const double *sums /* p+2 or null */, double *tpart, double *vpart);
int launch_moments_reduce(hipStream_t s, const GramPlan &pl, const double *tpart, const double *vpart, double *moments);

// stats layout written by finalize (doubles): [0] meanY [1] scaleY [2] yy (sum of squared standardised y)
// [3] nobs [4..4+p) meanX [4+p..4+2p) scaleX (dense) or colsq_inv (big) [4+2p] 1 if the moments were read as shifted [4+2p+1] 0
__host__ __device__ static inline int stats_len(int p) { return 4 + 2 * p + 2; }
__host__ __device__ static inline int stats_shift_flag(int p) { return 4 + 2 * p; }
int launch_finalize(hipStream_t s, const double *moments, const double *sums, int p, int sem, int standardize,
                    int intercept, double *xx /* q x q */, double *xy /* q */, double *stats);
int launch_xtx_prepare(hipStream_t s, const double *xtx, const double *xty, const double *sf_inv /* or null */, int p,
                       double *xx, double *xy, double *stats);

// ------------------------------------------------------------------ sparse x (sparse.hip): moments of a compressed-sparse-column matrix
size_t csc_moments_work_bytes(int64_t n, int p);
bool csc_moments_fits(int p);
int launch_csc_moments(hipStream_t s, const int64_t *colptr, const int32_t *rowidx, const double *val, const double *y, int64_t n, int p,
                       void *work, double *moments);
// observation weights of oemDense (weighted.hip): DataStd's weighted statistics, the scaled copy, its constants into `stats`
int launch_weighted_stats(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, const double *w, int flag, double *ws);
int launch_weighted_apply(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, const double *w, int flag, const double *ws,
                          int squared, double *z, int64_t ldz, double *yz);
int launch_weighted_patch_stats(hipStream_t s, const double *ws, int p, double *stats);
int launch_gram_loss(hipStream_t s, const double *xx, const double *xy, const double *stats, int q, const double *beta, const double *sinv,
                     const int *niter, double *loss, int nk);                       // oemSparse's compute.loss behind the larger engines
int launch_resid_loss(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, const double *beta, int rows, int nk,
                      double *part, double *loss);                                  // ... and with p >= n

// ------------------------------------------------------------------ xval.oem (xval.hip)
size_t fold_layout_ints(int64_t n, int K);
int launch_fold_layout(hipStream_t s, const int *foldid, int64_t n, int K, int *blockcnt, int64_t *fold_n, int64_t *fold_start,
                       int *pos, int *bad);
int launch_gather_rows(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, const int *pos,
                       double *xo, int64_t ldo, double *yo);
int launch_fold_sum(hipStream_t s, const double *M, int K, size_t len, int skip /* 1-based, 0: none */, double *out);
int cv_wg_per_fold(int64_t n, int K, int npen, int num_cu);
size_t cv_part_doubles(int nwg, int K, int npen, int nl);
int launch_cv_error(hipStream_t s, const double *xp, int64_t ldp, const double *yp, const int64_t *fold_start, const int64_t *fold_n,
                    int K, int p, const double *B, int npen, int nl, int mae, int wmode, int nwg, double n, double *part, double *out, bool triples = false);
// observation weights of xval.oem (ref src/oem_xval_dense.h:486-623): unweighted column sums of squares per fold, then the
// fold-ordered copy times sqrt(w); and the weighted counterpart of launch_finalize
int launch_weight_scale(hipStream_t s, double *xp, int64_t ldp, double *yp, int p, int K, const int64_t *fold_start, const int64_t *fold_n,
                        double *csq);
int launch_finalize_weighted(hipStream_t s, const double *Mw, const double *cs, int p, int standardize, int intercept, double *xx,
                             double *xy, double *stats);

// ------------------------------------------------------------------ eigen + path
struct PathArgs {
    int p;                   // dimension of beta (q)
    int npen, nl, user_lambda, maxit, accelerate, compute_loss;
    int ngroups;             // 0 when no group data
    int lanczos_steps;       // eigen step: number of Lanczos steps (<= p)
    int yscale;              // 1: ilambda = lambda / scaleY and lmax *= scaleY (dense); 0: xtx / big
    int lmax_from;           // lambda_zero = max |xy[j]| over j >= lmax_from (1: xval.oem leaves the intercept slot out, ref src/oem_xval_dense.h:1025-1032)
    double alpha, gamma, tau, tol, lambda_min_ratio;
    const double *xx;        // q x q col-major, ld = q
    const double *xy;        // q
    const double *stats;     // see stats layout
    const int *penalty;      // npen
    const double *lambda_user;   // npen * nl or null
    const double *pf;        // q penalty factors
    const double *sinv;      // q: oemXTX::get_beta in-place rescale (ref src/oem_xtx.h:576-581) or null
    const int *gid;          // q: index of the column's group in unique_groups, -1 if none
    const int *gstart;       // ngroups+1
    const int *gidx;         // members by group, increasing
    const double *gw;        // ngroups weights
    const int *gzero;        // ngroups: unique_groups[g] == 0
    // grp_head = 1, 2, 3 (the launch engines on the packed triangle, path_large.hip: sympk_head_kernel<HB>): every group is a run of <= 32 HB
    // neighbouring coordinates and nothing needs a sum over all coordinates -- the group operators then run in the head of the (head, product)
    // pairs like the element-wise ones.  Per coordinate: grun[2 j] = its group's first coordinate, grun[2 j + 1] = one past its last |
    // (unique_groups[g] == 0) << 30 -- (1, 0): in no group --, gwc[j] = the group's weight
    int grp_head;
    const int *grun;
    const double *gwc;
    // outputs (device)
    double *beta;            // npen * nl * q (standardised scale)
    double *lambda_out;      // npen * nl (unscaled lambda actually used)
    int *niter;              // npen * nl
    double *loss;            // npen * nl
    double *d_out;           // D_OUT_LEN doubles: [0] = d, [1] = lambda_max, [2] shader cycles, [3] 100 MHz ticks of the fused kernel,
                             // [4] Lanczos steps taken, [5] 1 if the step cap ended the recurrence (neither stop rule nor breakdown),
                             // [6] exchange poison: cleared by the host before the launch, set to 1 by ANY workgroup whose exchange
                             //     timed out (a slot nobody else writes: the writer's stores cannot cover it up -- ADVICE r2)
    // workspace for the large-p engine
    double *work;
    // q > 4096 on the launch-per-iteration Gram engine: sympk_doubles(q) doubles for the packed lower triangle of XX and the partial
    // vectors of its products (path_large.hip: sympk_*; a buffer of the context, outside the workspace frame); null: the row-streaming product
    double *sympk;
    // nbatch > 1 (p <= SMALL_P_MAX only): blockIdx.y selects one of nbatch independent problems that share everything above
    // except xx, xy, stats (element strides bs_xx, bs_xy, bs_stats), the outputs (byte stride bs_out) and work (bs_work)
    int nbatch;
    long long bs_xx, bs_xy, bs_stats, bs_out, bs_work;
    // pen_split (p <= SMALL_P_MAX only): every penalty of every instance gets a workgroup (set) of its own -- penalties are
    // independent cold starts (ref src/oem_dense.cpp:243-244) -- so blockIdx.y = instance + nbatch * penalty; pen_lo / pen_hi are
    // the penalties this workgroup walks ([0, npen) without the split)
    int pen_split, pen_lo, pen_hi;
    const double *lmax_xy;   // not null: every instance takes lambda_zero from THIS xy (xval.oem: the grid of the full-data fit, ref src/oem_xval_dense.cpp:177-194)
    // not null (row-split kernel only): the outputs above point into PINNED HOST memory the device writes directly (the kernel only
    // ever stores to them), and the kernel leaves a copy of `stats` here as well -- no device-to-host copy node behind the kernel
    double *stats_out;
    int stats_n;             // doubles of stats to copy
    double d_fixed;          // > 0 (launch-per-iteration Gram engine only): d handed over, no eigenvalue step (weighted oemDense with nobs <= nvars:
                             // the reference takes d from one matrix and iterates on another, ref src/oem_dense.h:466-483, 513-517)
    // The persistent engines only (one launch for the whole penalty x lambda path): a word in host-coherent pinned memory that the host
    // sets when the caller's interrupt callback fires while it waits for the launch (ref src/oem_dense.cpp:235-238: the reference polls
    // every third lambda).  Every workgroup reads it once per 128 iterations and inside any exchange spin that lasts; whoever sees it
    // stops waiting for anybody, the "leave" bit rides in the workgroup's next vote and the loops are left.  Null: never asked.
    const int *abort_word;
    // path_coop.hip, q <= 512: the cooperating workgroups of an instance all on ONE XCD (8 x as many workgroups are launched and only
    // those with blockIdx.x % 8 == (xcd_base + blockIdx.y) % 8 take part; the kernel proves the placement before it relies on it)
    int one_xcd, xcd_base;
};

// the read of PathArgs::abort_word (system scope: the word lives in host memory)
// (wave-uniform by construction -- every lane reads the same word in the same instruction -- and said so: the engines keep what
// follows from it in scalar registers)
__device__ __forceinline__ bool path_abort_asked(const int *w)
{
    return w && __builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0;
}
constexpr unsigned PATH_ABORT_SPINS = 1024u;   // an exchange spin looks at the word (and at its timeout) every 1,024 sweeps (~1 ms): never on the way of an exchange that arrives
constexpr unsigned PATH_TIMEOUT_ROUNDS = 1000u; // ... and gives up after that many rounds (~1 s): a partner is gone
constexpr int PATH_FAILED_TIMEOUT = 1, PATH_FAILED_ABORT = 2;

// the problem instance of this workgroup (see PathArgs::nbatch); all scalar arithmetic
__device__ __forceinline__ PathArgs path_instance(PathArgs A)
{
    const int nb = A.nbatch > 1 ? A.nbatch : 1;
    const long long y = blockIdx.y;
    if (A.pen_split) { A.pen_lo = (int)(y / nb); A.pen_hi = A.pen_lo + 1; }
    A.work += y * A.bs_work;                                   // exchange granules: one set per workgroup set
    if (A.nbatch > 1) {
        const long long b = y % nb;
        A.xx += b * A.bs_xx; A.xy += b * A.bs_xy; A.stats += b * A.bs_stats;
        const long long ob = b * A.bs_out;
        A.beta = (double *)((char *)A.beta + ob); A.lambda_out = (double *)((char *)A.lambda_out + ob);
        A.loss = (double *)((char *)A.loss + ob); A.d_out = (double *)((char *)A.d_out + ob);
        A.niter = (int *)((char *)A.niter + ob);
    }
    return A;
}

static const int D_OUT_LEN = 7;
// Stop rule of every Lanczos recurrence here: the geometric tail implied by two successive moves of the top Ritz value, relative.
// The reference's own tolerance on this eigenvalue is 1e-10 (Spectra, ref src/oem_dense.h:494-498); the estimate is conservative
// (config 1: it says 3e-13 where the true error is 1e-14), and the tests hold d to 1e-10 against LAPACK and the CPU restatement.  (1e-11 was tried in round 3: config 1 still takes 40 steps, config 4 still 144, and one xval fit moved: no gain.)
#ifndef OEM_LANCZOS_TAIL_TOL
#define OEM_LANCZOS_TAIL_TOL 1e-12
#endif
static const int COOP_MIN_Q = 209, COOP_MIN_Q_GROUPS = 209;      // from here the cooperating-workgroup engine (path_coop.hip) takes the path when it is eligible (element-wise penalties only / a group penalty in the call)
int path_coop_min_q(bool has_groups);
static const int SMALL_P_MAX = 288;     // one workgroup where the matrix fits its registers (+ LDS); else four cooperating workgroups (big.oem's p + 1 = 257 included)
size_t path_small_xchg_bytes();
int launch_path_small(hipStream_t s, const PathArgs &a);          // p <= SMALL_P_MAX: one fused launch
bool path_small_takes_rows(const PathArgs &a);                    // will launch_path_small use the row-split kernel (p <= 208)?
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size): the call costs microseconds, and config 1 made it
// in front of every Gram launch with the GPU idle
int lds_limit_once(const void *fn, size_t bytes);
int run_path_large(hipStream_t s, const PathArgs &a, double *host_scratch);   // any p: multi-launch engine
size_t path_large_work_doubles(int p, int nsteps);
size_t sympk_doubles(int q);                                      // 0 for q <= 4096
int sympk_gemv_probe(hipStream_t s, const double *xx, int q, double *pk, const double *vec, double *out, int reps, double *us_per_product);
// 288 < p <= 1024: one persistent launch of cooperating workgroups (path_coop.hip)
bool path_coop_eligible(int q, bool has_sinv, bool compute_loss, int ngroups, int nbatch);
int path_coop_workgroups(int q);
size_t path_coop_xchg_bytes();
int launch_path_coop(hipStream_t s, const PathArgs &a);

// 1024 < p <= 4096, element-wise penalties: one persistent launch with the lower triangle of XX in the register files of <= 192 CUs
// (path_symcoop.hip).  The plan (tiles per wave, blocks per workgroup, senders per block) is pure host arithmetic.
struct SymcoopPlan {
    int q = 0, T = 0, NT = 0, G = 0, nsum = 0, e1n = 0;    // 64-blocks, tiles per wave, workgroups, sender rows of exchange 1, sender sweeps per owner
    bool runs = false;                                      // the owners' slices are cut at group-run boundaries
    int split = 0;                                          // ... and some group (of more than 32 members) lies in several owners' slices: the most owners of one
    std::vector<int> tab;                                   // blkbase[80] then G records (uploaded as is); split groups: + the fragment table [2 q]
};
bool symcoop_plan(int q, int gmax, SymcoopPlan &P, const int *runs = nullptr, int nruns = 0);   // runs: group runs (starts, nruns + 1) the owners' slices are cut at
int symcoop_plan_owners(int q, int gmax, const int *runs, int nruns, int *owner_c0, int *owner_n, int *frag, int *G_out, int *split_out);
size_t symcoop_xchg_bytes(const SymcoopPlan &P);
size_t symcoop_work_bytes(const SymcoopPlan &P);    // ... plus the kernel's copy of its arguments
size_t symcoop_xchg_bytes_max(int q);
bool path_symcoop_eligible(const PathArgs &a, bool group_penalty, bool plan_has_runs);
int launch_path_symcoop(hipStream_t s, const PathArgs &a, const SymcoopPlan &P, const int *plan_dev, void *xchg);

// 1024 < p <= 2048, element-wise penalties: the row-split form, ONE exchange per iteration (path_symcoop.hip: path_rowcoop_kernel)
bool path_rowcoop_eligible(const PathArgs &a, bool group_penalty);
int path_rowcoop_workgroups(int q);
size_t path_rowcoop_xchg_bytes();
int launch_path_rowcoop(hipStream_t s, const PathArgs &a, void *xchg);

// ------------------------------------------------------------------ p >= n (wide.hip, path_large.hip: run_path_wide)
// The reference's own iteration for p >= n: no Gram, two products with the standardised X per iteration
// (ref src/oem_dense.h:363-366, 476-482, 513-521).  Layout of the standardised copy: `nb` row blocks of `rb` rows each (the last
// may hold fewer), every block a column-major (64 nr) x p matrix of its own with zero padding rows -- a column of a block lives in
// the registers of one wave (nr <= 32 doubles per lane).  n <= 2048: one block.
struct WideLayout {
    int nb;                  // row blocks
    int rb;                  // data rows per block (block b holds rows [b rb, min(n, (b + 1) rb)))
    int nr;                  // 64-row register slices per block: a block has 64 nr rows
    __host__ __device__ long long npad() const { return 64LL * nr; }              // rows of one block
    __host__ __device__ long long rows() const { return (long long)nb * 64 * nr; }  // rows of all blocks (vectors of this length)
};
WideLayout wide_layout(int64_t n);
struct WideArgs {
    const double *xs;        // standardised X: block b at xs + b * npad() * p
    const double *ys;        // standardised y, rows() entries in the same blocked order (padding zero)
    WideLayout lay;
    int n;
    double *scratch;         // wide_scratch_doubles(n, p) doubles
    // group penalties where every group is a run of neighbouring columns (the usual case): the runs in column order, dealt to the
    // workgroups of the fused group kernel (path_large.hip: wide_groups_kernel) in whole runs.  grun_W = 0: not available.
    const int *grun_start = nullptr;   // device: nruns + 1 column indices
    const int *grun_gid = nullptr;     // device: group index of each run
    const int *grun_wg = nullptr;      // device: grun_W + 1 run indices
    int grun_W = 0, grun_cpw = 0;      // workgroups; most columns in one workgroup
};
static const int WIDE_GRUN_MAX = 64;  // longest run (group) the fused group kernel takes
static const int WIDE_MAX_N = 32768;         // 16 row blocks of 2048 rows
int wide_workgroups(int n, int p);
size_t wide_scratch_doubles(int n, int p);
int launch_wide_standardize(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *y, int standardize, int intercept,
                            const WideLayout &lay, double *xs, double *ys, double *xy, double *stats);
int launch_big_wide_scales(hipStream_t s, const double *x, int64_t n, int64_t ld, int p, const double *xy, double *stats, double *xy_std);
int launch_big_gram_scales(hipStream_t s, const double *xx, const double *xy, int p, int standardize, double *stats, double *xy_std);
int run_path_wide(hipStream_t s, const PathArgs &a, const WideArgs &w, double *host_scratch);
// the same iteration as ONE persistent launch of cooperating workgroups with Xs in registers (path_wcoop.hip): element-wise
// penalties, one row block, p <= 4 CW WCOOP_GMAX columns (CW = 16 / 8 / 4 by column height); WCOOP_GMAX = three quarters of the CUs
static const int WCOOP_GMAX = 192, WCOOP_MAX_SETS = 8;
int path_wcoop_workgroups(int n, int p);
int path_wcoop_cpg(int n);                                        // columns per workgroup (0: not this engine)
int path_wcoop_max_workgroups(int n, int p);                     // ... of a partition cut at group boundaries, at most
int path_wcoop_sets(int n, int p, int npen, int num_cu, int G = 0);   // workgroup sets side by side, one penalty each (G: of a cut partition)
size_t path_wcoop_xchg_doubles(int n, int p);
size_t path_wcoop_launch_doubles(int n, int p, int G, int sets);
bool path_wcoop_eligible(const PathArgs &a, const WideArgs &w);
int launch_path_wcoop(hipStream_t s, const PathArgs &a, const WideArgs &w, int sets, const int *cstart = nullptr, int G = 0);   // cstart (device, G + 1 ints): columns cut at group boundaries
// the same where Xs does not fit the registers: G persistent workgroups re-read their column tiles every iteration (path_wcoop.hip: path_wstream_kernel)
size_t path_wstream_xchg_doubles(int n);
// ... and with more column sets of every wave in the accumulator file (path_wres_kernel): Xs up to ~11 M entries in registers
int path_wres_workgroups(int n, int p);
size_t path_wres_xchg_doubles(int n, int p);
bool path_wres_eligible(const PathArgs &a, const WideArgs &wd, int max_wg);
int launch_path_wres(hipStream_t s, const PathArgs &a, const WideArgs &wd);
bool path_wstream_eligible(const PathArgs &a, const WideArgs &w, int G);
int launch_path_wstream(hipStream_t s, const PathArgs &a, const WideArgs &w, int G);

// opts->interrupt of the call in progress on this thread (api.hip: run_paths sets it around the engines); false if none
bool caller_interrupted();

int launch_eig_small(hipStream_t s, const double *a, int p, int steps, double *out /* [d, lambda_max] */);

__host__ __device__ static inline bool pen_is_net(int pen)
{
    return pen == OEMGPU_ELASTIC_NET || pen == OEMGPU_MCP_NET || pen == OEMGPU_SCAD_NET ||
           pen == OEMGPU_GRP_LASSO_NET || pen == OEMGPU_GRP_MCP_NET || pen == OEMGPU_GRP_SCAD_NET;
}
__host__ __device__ static inline bool pen_is_grp(int pen) { return pen >= OEMGPU_GRP_LASSO; }

}  // namespace oemgpu
