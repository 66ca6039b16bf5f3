/*
 * oemgpu.h -- C ABI of the MI355X-native Orthogonalizing-EM solver (liboemgpu.so).
 *
 * Drop-in boundary for the dense Gaussian hot path of jaredhuling/oem (reference @ 2024_08_07):
 * the three entry points below are what the reference's `.Call` targets would bind instead of
 * their RcppEigen bodies (INTEGRATION.md shows the R-side shim).  Plain pointers and sizes only;
 * everything is IEEE fp64, matrices are column-major, outputs are caller-allocated, inputs are
 * never written.  All "ref:" citations are paths under the reference tree.
 *
 * Return value: 0 on success, <0 on error; oemgpu_last_error() gives the message
 * (thread-local).  There is NO CPU fallback: without a gfx950 device every compute entry
 * point fails with OEMGPU_ERR_NO_DEVICE.
 */
#ifndef OEMGPU_H
#define OEMGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OEMGPU_OK              0
#define OEMGPU_ERR_ARG        -1   /* invalid argument (the R front ends stop() on these: ref R/oem.R:215-431) */
#define OEMGPU_ERR_NO_DEVICE  -2
#define OEMGPU_ERR_HIP        -3   /* a HIP runtime call failed */
#define OEMGPU_ERR_UNSUPPORTED -4  /* outside the restated path (e.g. big.oem with p >= n, ref src/oem_big.h:547-551) */
#define OEMGPU_ERR_INTERNAL   -5
#define OEMGPU_ERR_INTERRUPTED -6  /* opts->interrupt asked to stop (the R shim then raises the pending user interrupt,
                                     ref src/oem_dense.cpp:235-238 Rcpp::checkUserInterrupt); every buffer is already released */

/* penalty codes = position in the R default vector (ref R/oem.R:165-173) */
enum {
    OEMGPU_ELASTIC_NET = 0, OEMGPU_LASSO = 1, OEMGPU_OLS = 2, OEMGPU_MCP = 3, OEMGPU_SCAD = 4,
    OEMGPU_MCP_NET = 5, OEMGPU_SCAD_NET = 6, OEMGPU_GRP_LASSO = 7, OEMGPU_GRP_LASSO_NET = 8,
    OEMGPU_GRP_MCP = 9, OEMGPU_GRP_SCAD = 10, OEMGPU_GRP_MCP_NET = 11, OEMGPU_GRP_SCAD_NET = 12,
    OEMGPU_SPARSE_GRP_LASSO = 13, OEMGPU_NPENALTIES = 14
};

/* Arguments shared by the three entry points: the scalar/vector arguments of
 * oem_fit_dense (ref src/oem_dense.cpp:30-48), oem_xtx (ref src/oem_xtx.cpp:29-44) and
 * oem_fit_big (ref src/oem_big.cpp:30-48) after R's coercions (ref R/oem.R:411-445). */
typedef struct oemgpu_opts {
    int32_t        npen;             /* length(penalty) >= 1 */
    const int32_t *penalty;          /* npen codes */
    int32_t        nlambda;          /* nlambda_, used when no lambda is supplied */
    double         lambda_min_ratio; /* lmin_ratio_ */
    const double  *lambda_user;      /* lambda_: npen x nlambda_user, one row per penalty, each sorted
                                        decreasing (ref R/oem.R:366-404); NULL => generated grid */
    int32_t        nlambda_user;
    double         alpha, gamma, tau;
    double         tol;              /* opts$tol */
    int32_t        maxit;            /* opts$maxit */
    int32_t        accelerate;       /* opts$accelerate (dense only, ref src/oem_dense.h:633-651) */
    int32_t        compute_loss;     /* compute_loss_ (dense only) */
    const double  *penalty_factor;   /* p values */
    const int32_t *groups;           /* ngroupvars values or NULL; big.oem with intercept passes p+1
                                        values with a leading 0 (ref R/big_oem.R:254-257) */
    int32_t        ngroupvars;
    const int32_t *unique_groups;    /* ngroups values, sorted (ref R/oem.R:292) */
    int32_t        ngroups;
    const double  *group_weights;    /* n_group_weights values; 0 => sqrt(group size) (ref src/oem_dense.h:447-454) */
    int32_t        n_group_weights;
    int32_t        device;           /* HIP device ordinal; -1 => current device */
    /* ---- host-resident entry points only (oemgpu_fit_dense, oemgpu_fit_big); zero / NULL = the defaults ---- */
    int32_t        ngpus;            /* G > 1: the rows are split over G devices inside the library, floor(n / G) rows each and
                                        the remainder on the last (as the reference's row blocks, ref src/oem_dense.h:328,343,
                                        src/oem_big.h:329-358); every device builds the moments of its rows and they are summed
                                        in device order on the first one (peer copies over xGMI), which solves.  0 or 1: `device` */
    const int32_t *devices;          /* ngpus ordinals, or NULL => device, device + 1, ... (device = -1 => 0, 1, ...) */
    int32_t        upload_threads;   /* host staging threads per device (pageable rows -> pinned bounce slots -> H2D); 0 => 8 */
    int          (*interrupt)(void *);   /* polled between row blocks and between penalties / batches of iterations;
                                            non-zero => OEMGPU_ERR_INTERRUPTED after cleanup.  NULL => never polled */
    void          *interrupt_arg;
} oemgpu_opts;

/* -------------------------------------------------------------------------------------------
 * Drop-in entry points (host buffers in, host buffers out; exactly the data the .Call carries).
 *
 * nl below = nlambda_user if lambda_user != NULL else nlambda.
 * beta:       npen * nl * (p+1) doubles; beta[(k*nl + i)*(p+1) + j] = coefficient j (0 = intercept)
 *             of penalty k at lambda i, i.e. each penalty's block is the reference's (p+1) x nl
 *             column-major matrix (ref src/oem_dense.cpp:194,252-254).  For "ols" only i = 0 is
 *             meaningful (ref :208-211,282-288).  oemgpu_fit_xtx: p rows, no intercept row
 *             (ref src/oem_xtx.cpp:129).
 * lambda_out: npen * nl, the unscaled lambda actually used (ref src/oem_dense.cpp:278)
 * niter:      npen * nl (maxit+1 when the loop ran out, ref src/oem_base.h:94-109)
 * loss:       npen * nl, 1e99 unless compute_loss (ref src/oem_dense.cpp:229-230,256-260)
 * d:          1.005 * lambda_max(X'X/n) (ref src/oem_dense.h:498)
 * ------------------------------------------------------------------------------------------- */

/* replaces oem_fit_dense, ref src/oem_dense.cpp:30-309 (family "gaussian", weights empty).
 * Both branches of ref src/oem_dense.h:476-482: n > p, and p >= n, where the reference iterates through X twice
 * (u = X'(Y - X b)/n + d b, d from XXt/n, ref :363-366, 513-521).  The library runs that very form -- a standardised copy of x
 * on the device, one read of it per iteration, no p x p matrix -- where it pays (n <= 32768, p > 1024 and 2 n < p: p = 20,000
 * needs 80 MB instead of 3.2 GB), and the same iteration written on the Gram elsewhere. */
int oemgpu_fit_dense(const double *x, int64_t n, int32_t p, const double *y,
                     int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                     double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);

/* replaces oem_xtx, ref src/oem_xtx.cpp:29-219.  scale_factor: p values or NULL. */
int oemgpu_fit_xtx(const double *xtx, const double *xty, int32_t p, const double *scale_factor,
                   const oemgpu_opts *o,
                   double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);

/* replaces oem_fit_big / oem_fit_fb_big, ref src/oem_big.cpp:30-258, src/oem_fb_big.cpp:30-258.
 * The big.matrix is handed over as row shards (the reference itself slices rows,
 * ref src/oem_big.h:319-361): shard s holds n_shard[s] rows, column-major with leading
 * dimension n_shard[s].  One shard of n rows is the plain big.matrix buffer. */
int oemgpu_fit_big(const double *const *x_shards, const int64_t *n_shard, int32_t nshards, int32_t p,
                   const double *const *y_shards,
                   int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                   double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);

/* oem_fit_dense with a non-empty `weights_` (ref src/oem_dense.cpp:34,75,152,162; src/oem_dense.h:368-414, 699-707, 759-770;
 * src/DataStd.h:94-202): the R front end never sends one ("weights not implemented yet", R/oem.R:244), the compiled entry takes it.
 * Computed as the reference computes it (sqrt(w)-weighted DataStd statistics -- unweighted for x under flag 3 --, X'WX / n, X'(Yw) / n,
 * loss = sum w r^2; with nobs <= nvars d from (sqrt(w) X)(sqrt(w) X)'/n but the iteration X'((Y - X beta) w^2)/n + d beta, w SQUARED, as
 * src/oem_dense.h:513-517 has it -- there through the p x p Gram forms on the launch-per-iteration engine).  weights: n values, finite, >= 0. */
int oemgpu_fit_dense_weighted(const double *x, int64_t n, int32_t p, const double *y, const double *weights,
                              int32_t standardize, int32_t intercept, const oemgpu_opts *opts,
                              double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);

/* replaces oem_fit_sparse, ref src/oem_sparse.cpp:30-267 (family "gaussian", weights empty, n > p): oem() on a dgCMatrix.
 * colptr[p + 1], rowidx[nnz], values[nnz]: the compressed sparse column slots @p, @i, @x (row indices increasing inside a
 * column).  Outputs as oemgpu_fit_dense.  The semantics are oemSparse's, not oemDense's: no centring, columns scaled by
 * sqrt(sum x^2 / (n - 1)), the intercept as a Gram column of value sqrt(mean diag / n) whose coefficient is rescaled in
 * place after every lambda (ref src/oem_sparse.h:493-615, 897-917), lambda_zero without the intercept slot (:854-863).
 * compute_loss (ref :919-944) for p + intercept <= 288.  The Gram is built by the dense FP64-MFMA pass over zero-filled row tiles of x (<= 2 GiB each). */
int oemgpu_fit_sparse(int64_t n, int32_t p, const int64_t *colptr, const int32_t *rowidx, const double *values, const double *y,
                      int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                      double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);

/* replaces oem_xval_dense, ref src/oem_xval_dense.cpp:31-482 (family "gaussian", weights empty): xval.oem's fast
 * cross-validation.  foldid: n values in 1..nfolds.  type_measure: 0 "mse", 1 "mae" (ref :378-411).
 * beta, lambda_out, niter, loss, d: the fit on ALL rows, laid out as in oemgpu_fit_dense (loss only if compute_loss,
 * ref :296-301).  cvm, cvsd: npen * nl each -- mean over the n observations of the error of row i under the fit that
 * left row i's fold out, and sqrt(sample variance / n) of the same (ref :452-461; 0 beyond the single "ols" entry). */
int oemgpu_xval_dense(const double *x, int64_t n, int32_t p, const double *y, const double *weights, const int32_t *foldid, int32_t nfolds,
                      int32_t standardize, int32_t intercept, int32_t type_measure, const oemgpu_opts *o,
                      double *beta, double *lambda_out, int32_t *niter, double *loss, double *d,
                      double *cvm, double *cvsd);

/* -------------------------------------------------------------------------------------------
 * Device-resident / staged interface.  Used when X already lives in HBM (bench.py, repeated
 * solves) and by the one-process-per-GPU row-sharded driver (oem_amd/distributed.py), which
 * all-reduces the moment buffer between oemgpu_moments_dev and oemgpu_solve_moments_dev.
 * All *_dev pointers are device pointers on the context's device.
 * ------------------------------------------------------------------------------------------- */
typedef struct oemgpu_ctx oemgpu_ctx;

/* stream: a hipStream_t to run on, or NULL for a stream owned by the context */
oemgpu_ctx *oemgpu_create(int32_t device, void *stream);
void        oemgpu_destroy(oemgpu_ctx *ctx);
int         oemgpu_synchronize(oemgpu_ctx *ctx);

/* Moment buffer of the augmented, shifted data Z = [X - 1 c_x' | y - c_y | 1]:
 * q = p + 2; M is q x q column-major, lower triangle valid:
 *   M[i,j] (i>=j, i,j<p) = sum_r (x_ri - c_i)(x_rj - c_j)      M[p,j]   = sum_r (y_r - c_y)(x_rj - c_j)
 *   M[p,p] = sum_r (y_r - c_y)^2     M[p+1,j] = sum_r (x_rj - c_j)     M[p+1,p] = sum_r (y_r - c_y)
 *   M[p+1,p+1] = number of rows.
 * Moments add over row shards that use the same shift, which is what the RCCL all-reduce sums. */
static inline int64_t oemgpu_moments_len(int32_t p) { return (int64_t)(p + 2) * (p + 2); }

/* Sample sums for the provisional shift (oemgpu_sums_len(p) doubles): sums_dev[0..p-1] = sum over the sampled
 * rows of x_j, sums_dev[p] = same for y, sums_dev[p+1] = number of sampled rows, sums_dev[p+2 .. 2p+2] = the
 * sampled sums of squares of the same p+1 columns, sums_dev[2p+3] = 0.  All-reduce the whole buffer across row
 * shards.  The shift in effect is then a pure function of the buffer (identical in every kernel and rank):
 *   c_j = sums[j] / sums[p+1]  if ANY column has mean_j^2 > 2^8 var_j (sample mean / variance), else c = 0
 * -- un-shifted accumulation costs (mean/sd)^2 eps of relative accuracy on the centred moments, and is used when
 * that is < 6e-14 because every x - c is an FP64 VALU op competing with the FP64 MFMA for the DP units. */
static inline int64_t oemgpu_sums_len(int32_t p) { return 2 * (int64_t)(p + 1) + 2; }
int oemgpu_shift_sums_dev(oemgpu_ctx *ctx, const double *x_dev, int64_t n, int64_t ld, int32_t p,
                          const double *y_dev, double *sums_dev);

/* moments_dev <- moments of rows [0,n) about the shift c defined above (sums_dev: the (all-reduced)
 * output of oemgpu_shift_sums_dev; NULL => c = 0).
 * Replaces DataStd's passes + X'Y + XtX (ref src/DataStd.h:203-265, src/oem_dense.h:318-361,704-707;
 * src/oem_big.h:743-841) with ONE pass over X. */
int oemgpu_moments_dev(oemgpu_ctx *ctx, const double *x_dev, int64_t n, int64_t ld, int32_t p,
                       const double *y_dev, const double *sums_dev, double *moments_dev);

/* out_dev[i] = ((parts[0][i] + parts[1][i]) + parts[2][i]) + ...: `nparts` buffers of `len` doubles, contiguous one behind the other
 * (an all-gather of the ranks' moment buffers), added in shard order by ONE kernel -- the order the in-library multi-GPU path
 * (opts.ngpus) adds its devices' buffers in, so both forms return the same bits whatever algorithm a collective library would
 * choose for an all-reduce.  The reference's own order is the arrival order of its threads at a critical section
 * (ref src/oem_dense.h:328-358).  Asynchronous on the context's stream. */
int oemgpu_sum_in_order_dev(oemgpu_ctx *ctx, const double *parts_dev, int32_t nparts, int64_t len, double *out_dev);

/* semantics selector for oemgpu_solve_moments_dev */
#define OEMGPU_SEM_DENSE 0   /* DataStd + oemDense (ref src/DataStd.h, src/oem_dense.h) */
#define OEMGPU_SEM_BIG   1   /* oemBig: (n-1)-scaling, intercept as Gram row/column (ref src/oem_big.h:731-842,469-566) */
#define OEMGPU_SEM_XVAL  3   /* oemXvalDense: oemBig's algebra (ref src/oem_xval_dense.h:745-789), lambda_zero without the intercept
                              * slot (:1025-1032), groups scanned over all p + 1 slots (:636), compute_loss allowed (:1088-1117) */

/* From (all-reduced) moments to the full result: standardisation constants, XX, XY, d, lambda grid,
 * penalty x lambda loops, recover.  Outputs are HOST buffers as in oemgpu_fit_dense. */
int oemgpu_solve_moments_dev(oemgpu_ctx *ctx, const double *moments_dev, const double *sums_dev, int32_t p,
                             int32_t semantics, int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                             double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);

/* oemgpu_fit_dense with X (n x p, leading dimension ld >= n) and y already on the device. */
int oemgpu_fit_dense_dev(oemgpu_ctx *ctx, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev,
                         int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                         double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);
/* oemgpu_fit_dense_weighted with X, y and the weights already on the device. */
int oemgpu_fit_dense_weighted_dev(oemgpu_ctx *ctx, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev,
                                  const double *weights_dev, int32_t standardize, int32_t intercept, const oemgpu_opts *opts,
                                  double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);

/* oemgpu_fit_xtx with xtx (p x p) / xty on the device. */
int oemgpu_fit_xtx_dev(oemgpu_ctx *ctx, const double *xtx_dev, const double *xty_dev, int32_t p,
                       const double *scale_factor, const oemgpu_opts *o,
                       double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);

/* 1 if the most recent oemgpu_solve_moments_dev (OEMGPU_SEM_DENSE) on this context was given moments about 0
 * (sums_dev == NULL) of data in which some column has mean^2 > 2^8 var -- the shift predicate above, evaluated on the
 * full-data moments.  The coefficients just returned may then have lost (mean/sd)^2 eps of relative accuracy to
 * cancellation: redo the pass about a shift (oemgpu_shift_sums_dev, oemgpu_moments_dev and this call with those sums).
 * 0 otherwise, -1 for a NULL context.  oemgpu_fit_dense(_dev) and the row-sharded driver do exactly this, so that the
 * usual data costs one pass and one collective, with no sample pass in front. */
int oemgpu_last_shift_advised(oemgpu_ctx *ctx);

/* oemgpu_xval_dense with X (n x p, leading dimension ld >= n), y, weights and foldid already on the device. */
int oemgpu_xval_dense_dev(oemgpu_ctx *ctx, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev,
                          const double *weights_dev /* or NULL */, const int32_t *foldid_dev, int32_t nfolds, int32_t standardize, int32_t intercept,
                          int32_t type_measure, const oemgpu_opts *o,
                          double *beta, double *lambda_out, int32_t *niter, double *loss, double *d,
                          double *cvm, double *cvsd);

/* xval.oem over row shards (one process per GPU; oem_amd/distributed.py: xval_oem_sharded): the three phases of
 * oemgpu_xval_dense_dev as separate calls, with the caller's collectives between them.  The cross-validation of the
 * reference is additive in exactly the way its fit is: per-fold Gram matrices are sums over rows
 * (ref src/oem_xval_dense.h:358-484), the K + 1 fits need nothing but those sums (ref src/oem_xval_dense.cpp:213-340),
 * and the CV error is a mean and a variance over observations (ref :343-461).  All three calls of one fit take the same
 * (n_local, p, nfolds, weighted, o); the fold-ordered copy of the local rows stays in the context between them.
 *   1. fold_moments_dev[oemgpu_xval_moments_len(p, nfolds, weighted)] <- per-fold moments of the LOCAL rows (device
 *      buffer: all-reduce it, sum), fold_n[nfolds] <- local fold sizes (host: all-reduce, sum);
 *   2. the summed moments and fold sizes in, the replicated fits out (beta ... d as in oemgpu_xval_dense);
 *   3. triples[npen * nl * 3] <- (count, mean, M2 = sum (v - mean)^2) of the LOCAL rows' errors (host: all-gather);
 *   oemgpu_xval_merge: the triples of all ranks, in rank order -> cvm, cvsd (Chan, Golub & LeVeque; pure host code). */
int64_t oemgpu_xval_moments_len(int32_t p, int32_t nfolds, int32_t weighted);
int oemgpu_xval_fold_moments_dev(oemgpu_ctx *ctx, const double *x_dev, int64_t n_local, int64_t ld, int32_t p, const double *y_dev,
                                 const double *weights_dev /* or NULL */, const int32_t *foldid_dev, int32_t nfolds, const oemgpu_opts *o,
                                 double *fold_moments_dev, int64_t *fold_n);
int oemgpu_xval_solve_folds_dev(oemgpu_ctx *ctx, const double *fold_moments_dev, const int64_t *fold_n_total, int64_t n_local, int32_t p,
                                int32_t nfolds, int32_t weighted, int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                                double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);
int oemgpu_xval_cv_triples_dev(oemgpu_ctx *ctx, int64_t n_local, int32_t p, int32_t nfolds, int32_t weighted, int32_t type_measure,
                               const oemgpu_opts *o, double *triples);
int oemgpu_xval_merge(const double *triples, int32_t nsets, const oemgpu_opts *o, double *cvm, double *cvsd);

/* The eigenvalue step of the most recent solve (or oemgpu_eig_max_dev) on this context: *steps = Lanczos steps taken, *capped = 1
 * if the recurrence ran into its step cap (min(2 q, 288) for q <= 288, 256 / 512 on the larger engines) instead of stopping by its
 * rule or by breakdown -- d = 1.005 x the last Ritz value is then a lower estimate (the reference's Spectra call, tol 1e-10 and up
 * to 10000 restarts, ref src/oem_dense.h:485-498, has no such cap; OEM converges for any d > lambda_max / 2).  -1 for a NULL context. */
int oemgpu_last_eigen_info(oemgpu_ctx *ctx, int32_t *steps, int32_t *capped);

/* Which kernel family ran the most recent penalty x lambda path on this context (diagnostics and tests: the engines are chosen by
 * size and options, api.hip: run_paths), and how many calls of a persistent engine so far timed out in their exchanges (somebody
 * else held the CUs) and were made again on the launch-per-iteration engines.  -1 for a NULL context. */
enum {
    OEMGPU_ENGINE_NONE = 0,
    OEMGPU_ENGINE_ROWS = 1,      /* p <= 288: one workgroup per penalty (path_small.hip) */
    OEMGPU_ENGINE_COOP = 2,      /* <= 1024: cooperating workgroups, one exchange per iteration (path_coop.hip) */
    OEMGPU_ENGINE_ROWCOOP = 3,   /* <= 2048, element-wise: the matrix in the accumulator files, one exchange (path_symcoop.hip) */
    OEMGPU_ENGINE_SYMCOOP = 4,   /* <= 4096: the lower triangle in registers, two exchanges (path_symcoop.hip) */
    OEMGPU_ENGINE_LAUNCHES = 5,  /* any p: launch-per-iteration engines on the Gram (path_large.hip) */
    OEMGPU_ENGINE_WCOOP = 6,     /* p >= n: the standardised X in vector registers (path_wcoop.hip) */
    OEMGPU_ENGINE_WRES = 7,      /* p >= n: ... and in the accumulator file (path_wcoop.hip: path_wres_kernel) */
    OEMGPU_ENGINE_WSTREAM = 8,   /* p >= n: persistent, X re-read every iteration (path_wcoop.hip: path_wstream_kernel) */
    OEMGPU_ENGINE_WLAUNCHES = 9  /* p >= n: launch-per-iteration (path_large.hip: run_path_wide) */
};
int oemgpu_last_path_engine(oemgpu_ctx *ctx, int32_t *engine, int32_t *persistent_fallbacks);
/* OEMGPU_ENGINE_COOP with q <= 512 puts the cooperating workgroups of an instance on ONE XCD where the device's layout allows (the
 * exchange then stays in that XCD's L2).  Of the most recent path launch on this context: 0 not asked for, 1 ran on one XCD, 2 asked for,
 * refused by the launch's own proof of placement and made again with the exchange at device scope (this context does not ask again),
 * 3 asked for, but an instance's workgroups were not all resident on its XCD (somebody else holds CUs there): made again once with the
 * workgroups anywhere on the device. */
int oemgpu_last_placement(oemgpu_ctx *ctx);

/* 1 if the most recent oemgpu_solve_moments_dev on this context found the shift predicate above true for its
 * sums_dev (and so read moments_dev as accumulated about c), 0 if not, -1 for a NULL context. */
int oemgpu_last_shift_in_effect(oemgpu_ctx *ctx);

/* lambda_max of a symmetric p x p device matrix (the Spectra call of ref src/oem_dense.h:485-498). */
int oemgpu_eig_max_dev(oemgpu_ctx *ctx, const double *a_dev, int32_t p, double *lambda_max);

/* Time of the most recent kernels on this context, measured with HIP events on the context's
 * stream (milliseconds; 0 if that stage has not run).  Stages: */
#define OEMGPU_T_SHIFT   0
#define OEMGPU_T_MOMENTS 1   /* Gram/moment build (the MFMA kernel + its partial reduction) */
#define OEMGPU_T_FINAL   2   /* moments -> XX, XY, standardisation constants */
#define OEMGPU_T_EIGPATH 3   /* eigenvalue + penalty x lambda loops */
#define OEMGPU_T_GRAMK   4   /* the MFMA Gram kernel alone */
#define OEMGPU_T_PATHCYC 6   /* not a time: shader cycles of the last fused eigen+path kernel (p <= 192) */
#define OEMGPU_T_PATHTICKS 7 /* not a time: the same span in 100 MHz ticks (cycles / ticks * 100 MHz = clock held) */
#define OEMGPU_NTIMERS   8
int oemgpu_last_timings(oemgpu_ctx *ctx, double *ms /* OEMGPU_NTIMERS */);
/* enable (1) / disable (0) event timing of the stages (off by default: events cost a few us) */
int oemgpu_set_timing(oemgpu_ctx *ctx, int32_t on);

/* Row range [*r0, *r1) of device g of G for n rows: floor(n / G) each, the remainder on the last
 * (ref src/oem_dense.h:328,343).  Pure host arithmetic. */
void oemgpu_row_split(int64_t n, int32_t G, int32_t g, int64_t *r0, int64_t *r1);

/* What the most recent host-resident call (oemgpu_fit_dense / oemgpu_fit_big) of THIS thread did, for bench.py and the tests:
 * [0] wall milliseconds of the whole call  [1] of the upload + moment passes  [2] of the solve(s)  [3] bytes staged to the
 * devices  [4] devices used  [5] row blocks streamed (all devices)  [6] 1 if the rows stayed resident in HBM, 0 if two block
 * buffers were recycled  [7] hipMalloc / hipHostMalloc calls made inside the call (0 in the steady state of repeated calls)
 * [8] cross-device hand-overs of the moment buffers that were staged through the host because the two devices cannot access each
 *     other (hipDeviceCanAccessPeer; OEMGPU_NO_PEER=1 forces that route) -- 0 when every pair used one peer copy over xGMI. */
#define OEMGPU_NHOSTSTATS 9
int oemgpu_last_host_stats(double *out /* OEMGPU_NHOSTSTATS */);

/* Frees every cached context (streams, workspaces, pinned staging).  The host-resident entry points keep theirs between
 * calls; nothing else needs this.  Contexts in use by another thread are left alone. */
void oemgpu_release_cache(void);

/* Host-only self-check of the persistent p >= n engine's scratch sizing (pure arithmetic, runs without a GPU): for an n x p problem
 * with npen penalties on a device of num_cu CUs, 0 if every column partition the launch may choose gets at least one workgroup set
 * and never more sets than the exchange scratch was sized for; otherwise +/- the offending workgroup count. */
int oemgpu_selftest_wcoop_sizing(int32_t n, int32_t p, int32_t npen, int32_t num_cu);

/* Host-only self-check of the engine PLAN (pure arithmetic, runs without a GPU): what api.hip: plan_paths decides for a call with
 * these sizes and options on a device of num_cu CUs -- *engine = the OEMGPU_ENGINE_* of the first attempt (+ 256 where the cooperating
 * engine is planned with every instance on ONE XCD: run time still asks the device for its layout; + 512 k, k = 1, 2, 3, where the
 * launch engines would run group operators in the head of their (head, product) pairs with k blocks of 32 coordinates on either side of
 * a workgroup's own: every group a run of <= 32 k neighbouring coordinates) -- and whether the
 * buffers the callers size hold what the launch will carve: *frame_bytes (outputs + parameter blob + engine workspace) against
 * *reserved_bytes, and for p >= n (wide_n > 0 rows, no Gram matrix) the persistent engine's exchange buffers against the scratch
 * (*scratch_need_doubles <= *scratch_have_doubles).  p: columns of x; q: dimension of beta (p + 1 with big.oem's intercept);
 * semantics: OEMGPU_SEM_* (2: oem.xtx). */
int oemgpu_selftest_plan(int32_t p, int32_t q, int32_t semantics, int32_t intercept, const oemgpu_opts *o, int32_t has_scale, int32_t nbatch,
                         int64_t wide_n, int32_t num_cu, int32_t *engine, int64_t *frame_bytes, int64_t *reserved_bytes,
                         int64_t *scratch_need_doubles, int64_t *scratch_have_doubles);

/* Host-only self-check of the MOMENT plan (gram.hip: gram_plan; pure arithmetic, runs without a GPU) for n rows and p columns on
 * a device of num_cu CUs: out[0] = tile columns of 16, out[1] / out[2] / out[7] = super-block rows of eight / six / four tile columns
 * the shared-slab kernel deals them into (all 0: p + 2 <= 112, one wave holds the triangle; all -1: 11-12, 15-16 or 16 k (- 1) tile columns: eight-wave
 * workgroups hold whole units of the triangle, gram_wd.hip), out[3] = row chunks, out[4] = 64-row
 * steps per chunk, out[5] = the multiply time of that deal in tile units (an h1 x h2 off-diagonal super-block h1 h2, a diagonal
 * one 36 / 24 / 12), out[6] = real tiles of the lower triangle.  Also checks that the partial-sum scratch sized for "any row count up
 * to n" (the folds of xval.oem, the row tiles of a sparse x) holds the plans of smaller row counts: OEMGPU_ERR_INTERNAL if not. */
int oemgpu_selftest_gram_plan(int64_t n, int32_t p, int32_t num_cu, int64_t *out /* 8 */);

/* Self-test aid (tests/test_gpu_host.py): enqueue, on the context's stream, `blocks` workgroups that each occupy a whole CU and
 * spin for `ms` milliseconds -- "somebody else holds the CUs", for the fallback of the persistent engines.  Asynchronous. */
int oemgpu_selftest_hold_cus(oemgpu_ctx *ctx, int32_t blocks, double ms);

/* Host-only self-check of the group reordering (api.hip: group_run_permutation; pure arithmetic, runs without a GPU): for the groups of `o`
 * over q coordinates, the permutation (new position -> old position) that makes every group a run of neighbouring coordinates -- groups in
 * the order of their first member, members in their own order (the order the reference sums their squares in, ref src/oem_dense.h:193-315) --
 * which lets the register-resident engine at 1024 < q <= 4096 take group penalties whatever the layout.  Returns q and fills perm[0..q), or 0
 * when no reordering applies: the groups are runs already, a variable is listed in two groups, or the group vector does not cover q
 * coordinates.  (Groups of more than 32 members -- an owner's slice -- are reordered like the others: the engine sums their norms over
 * several owners.) */
int oemgpu_selftest_group_permutation(const oemgpu_opts *o, int32_t q, int32_t *perm);

/* Host-only self-check of how the register-resident engine at 1024 < q <= 4096 deals group runs to its workgroups (path_symcoop.hip:
 * symcoop_plan; pure arithmetic, runs without a GPU).  runs[0 .. nruns]: the starts of the runs of neighbouring coordinates (one per group,
 * ungrouped coordinates runs of one), runs[nruns] = q.  On return *nowners workgroups (0: no plan, the launch-per-iteration engines take the
 * call), workgroup g owning the owner_n[g] coordinates from owner_c0[g] on (<= 32; slices end at run boundaries, or after every fourth
 * coordinate inside a run of more than 32), and per coordinate j frag[2 j] = 2 (first owner of j's run) + (1 if the run does not start that
 * owner's slice), frag[2 j + 1] = the number of owners the run lies in; *split = the most owners of one run (0: no run is split).  The arrays
 * hold 192 / 192 / 2 q entries. */
int oemgpu_selftest_symcoop_owners(int32_t q, int32_t num_cu, const int32_t *runs, int32_t nruns, int32_t *owner_c0, int32_t *owner_n, int32_t *frag,
                                   int32_t *nowners, int32_t *split);

/* Host-only self-check of the CU-slot book of the persistent engines (pure arithmetic, runs without a GPU): `calls` concurrent callers
 * each place `ninst` instances of W cooperating workgroups with every instance on ONE XCD of a device with num_cu CUs (path_coop.hip,
 * q <= 512).  bases[k] = the XCD of call k's first instance (chosen where the XCDs are emptiest), *peak = the most CUs any XCD was
 * booked for while all calls were in flight (<= num_cu / 8 whenever that is possible).  OEMGPU_ERR_ARG if the calls would have to
 * wait for each other (more than 3/4 of the CUs). */
int oemgpu_selftest_coop_slots(int32_t num_cu, int32_t W, int32_t ninst, int32_t calls, int32_t *bases, int32_t *peak);

/* Self-test / measurement aid: out = XX vec for a symmetric q x q matrix (q > 4096, column-major, device) through the packed lower
 * triangle the launch-per-iteration Gram engine streams beyond q = 4096 (path_large.hip: sympk_*; replaces the GEMV of
 * ref src/oem_xtx.h:378-381 / src/oem_dense.h:508-512): the pack, then `reps` products back to back between two HIP events on the
 * context's stream.  *us_per_product = the product kernel's own duration (bench.py prices 4 q^2 + 512 q + 8 q ceil(q / 128) bytes
 * against it).  Synchronises the stream. */
int oemgpu_selftest_sympk_gemv(oemgpu_ctx *ctx, const double *xx_dev, int32_t q, const double *vec_dev, double *out_dev, int32_t reps,
                               double *us_per_product);

/* The OEM_* / OEMGPU_* environment switches (engine selection for tests, knobs of the host-resident upload; none is needed in
 * production: DESIGN.md section 7b) are parsed ONCE, at the first call into the library.  oemgpu_reload_switches() parses the
 * environment again (tests); oemgpu_switch_names() is the space-separated list of every name the library reads. */
void        oemgpu_reload_switches(void);
const char *oemgpu_switch_names(void);

const char *oemgpu_last_error(void);
const char *oemgpu_version(void);
int         oemgpu_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* OEMGPU_H */
