/*
 * oem_shim.c -- the R-side binding a maintainer of jaredhuling/oem would add.
 *
 * Exports the SAME unmangled symbols with the SAME all-SEXP signatures as the reference's RcppExport entry
 * points (ref src/oem_dense.cpp:30-48, src/oem_xtx.cpp:29-44), so the unchanged R front ends
 * (R/oem.R:556-575, R/oem_xtx.R:389-406: `.Call("oem_fit_dense", ..., PACKAGE = "oem")`) resolve to it through
 * the package's dynamic symbol lookup (ref src/oem_init.c:167-171).  R C API only (no Rcpp); all numerics are in
 * liboemgpu.so behind include/oemgpu.h.
 *
 * Build (inside the package's src/, replacing oem_dense.cpp / oem_xtx.cpp):
 *     PKG_CPPFLAGS = -I<repo>/include        PKG_LIBS = -L<repo>/oem_amd -loemgpu
 * NOT compiled in this repository's image: there is no R installation.
 */
#include <R.h>
#include <Rinternals.h>
#include <Rinterface.h>      /* Rf_onintr */
#include <R_ext/Rdynload.h>  /* DllInfo (R_unload_oem) */
#include <string.h>

#include "oemgpu.h"

static const char *PENALTIES[OEMGPU_NPENALTIES] = {      /* ref R/oem.R:165-173 */
    "elastic.net", "lasso", "ols", "mcp", "scad", "mcp.net", "scad.net", "grp.lasso", "grp.lasso.net",
    "grp.mcp", "grp.scad", "grp.mcp.net", "grp.scad.net", "sparse.grp.lasso"};

/* User interrupts (ref src/oem_dense.cpp:235-238: Rcpp::checkUserInterrupt() every third lambda).  R_CheckUserInterrupt()
 * longjmps out of the caller, which would leak whatever the library holds; under R_ToplevelExec the jump is caught, so the
 * library's callback only learns "an interrupt is pending", unwinds itself (OEMGPU_ERR_INTERRUPTED after releasing every
 * buffer) and the shim raises the condition once the call has returned.  Polled on the R main thread only. */
static void shim_check_interrupt(void *unused) { (void)unused; R_CheckUserInterrupt(); }
static int shim_interrupted(void *unused) { (void)unused; return R_ToplevelExec(shim_check_interrupt, NULL) == FALSE; }

/* opts$name or R_NilValue (ngpus / devices are additions of this binding: absent => one GPU) */
static SEXP list_opt(SEXP list, const char *name)
{
    SEXP names = Rf_getAttrib(list, R_NamesSymbol);
    for (R_xlen_t i = 0; i < XLENGTH(list); i++)
        if (strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return VECTOR_ELT(list, i);
    return R_NilValue;
}

static SEXP list_elt(SEXP list, const char *name)
{
    SEXP names = Rf_getAttrib(list, R_NamesSymbol);
    for (R_xlen_t i = 0; i < XLENGTH(list); i++)
        if (strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return VECTOR_ELT(list, i);
    Rf_error("opts$%s is missing", name);
    return R_NilValue;
}

/* fills o from the SEXP arguments; buffers that need a C layout are R_alloc'd (freed by R at the end of .Call) */
static void fill_opts(oemgpu_opts *o, SEXP penalty_, SEXP groups_, SEXP unique_groups_, SEXP group_weights_,
                      SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_, SEXP gamma_, SEXP tau_,
                      SEXP penalty_factor_, SEXP compute_loss_, SEXP opts_, int dense)
{
    memset(o, 0, sizeof *o);
    o->npen = (int32_t)XLENGTH(penalty_);
    int32_t *pen = (int32_t *)R_alloc(o->npen, sizeof(int32_t));
    for (int k = 0; k < o->npen; k++) {
        pen[k] = -1;
        for (int c = 0; c < OEMGPU_NPENALTIES; c++)
            if (strcmp(CHAR(STRING_ELT(penalty_, k)), PENALTIES[c]) == 0) pen[k] = c;
        if (pen[k] < 0) Rf_error("unknown penalty '%s'", CHAR(STRING_ELT(penalty_, k)));
    }
    o->penalty = pen;
    o->nlambda = Rf_asInteger(nlambda_);
    o->lambda_min_ratio = Rf_asReal(lmin_ratio_);
    /* lambda_: list of one numeric vector per penalty, possibly length 0 (ref R/oem.R:366-404) */
    R_xlen_t nlu = XLENGTH(VECTOR_ELT(lambda_, 0));
    if (nlu > 0) {
        double *lam = (double *)R_alloc((size_t)o->npen * nlu, sizeof(double));
        for (int k = 0; k < o->npen; k++) memcpy(lam + (size_t)k * nlu, REAL(VECTOR_ELT(lambda_, k)), sizeof(double) * nlu);
        o->lambda_user = lam;
        o->nlambda_user = (int32_t)nlu;
    }
    o->alpha = Rf_asReal(alpha_); o->gamma = Rf_asReal(gamma_); o->tau = Rf_asReal(tau_);
    o->tol = Rf_asReal(list_elt(opts_, "tol"));
    o->maxit = Rf_asInteger(list_elt(opts_, "maxit"));
    o->accelerate = dense ? (Rf_asReal(list_elt(opts_, "accelerate")) != 0.0) : 0;     /* read as<double> (quirk Q12) */
    o->compute_loss = compute_loss_ != R_NilValue ? Rf_asLogical(compute_loss_) : 0;
    o->penalty_factor = REAL(penalty_factor_);
    o->groups = XLENGTH(groups_) ? INTEGER(groups_) : NULL;            o->ngroupvars = (int32_t)XLENGTH(groups_);
    o->unique_groups = XLENGTH(unique_groups_) ? INTEGER(unique_groups_) : NULL;   o->ngroups = (int32_t)XLENGTH(unique_groups_);
    o->group_weights = XLENGTH(group_weights_) ? REAL(group_weights_) : NULL;      o->n_group_weights = (int32_t)XLENGTH(group_weights_);
    o->device = -1;
    o->interrupt = shim_interrupted;
    {                                            /* options(oem.ngpus = G) style additions, passed through opts */
        SEXP g = list_opt(opts_, "ngpus"), dv = list_opt(opts_, "devices");
        if (g != R_NilValue) o->ngpus = Rf_asInteger(g);
        if (dv != R_NilValue && XLENGTH(dv) > 0) {
            SEXP di = PROTECT(Rf_coerceVector(dv, INTSXP));
            int32_t *d = (int32_t *)R_alloc(XLENGTH(di), sizeof(int32_t));
            for (R_xlen_t i = 0; i < XLENGTH(di); i++) d[i] = INTEGER(di)[i];
            o->devices = d; o->ngpus = (int32_t)XLENGTH(di);
            UNPROTECT(1);
        }
    }
}

/* non-zero return of the library -> R condition (the library has already released what it held) */
static void raise(int rc)
{
    if (rc == OEMGPU_ERR_INTERRUPTED) Rf_onintr();        /* the user's interrupt, re-raised outside the library */
    Rf_error("%s", oemgpu_last_error());
}

/* List(beta = list(...), lambda = list(...), niter = list(...), loss = list(...), d = d)  (ref src/oem_dense.cpp:280-307) */
static SEXP pack(const oemgpu_opts *o, int rows, int nl, const double *beta, const double *lam, const int32_t *niter,
                 const double *loss, double d)
{
    SEXP res = PROTECT(Rf_allocVector(VECSXP, 5)), names = PROTECT(Rf_allocVector(STRSXP, 5));
    const char *nm[5] = {"beta", "lambda", "niter", "loss", "d"};
    for (int i = 0; i < 5; i++) SET_STRING_ELT(names, i, Rf_mkChar(nm[i]));
    SEXP lb = PROTECT(Rf_allocVector(VECSXP, o->npen)), ll = PROTECT(Rf_allocVector(VECSXP, o->npen));
    SEXP ln = PROTECT(Rf_allocVector(VECSXP, o->npen)), lo = PROTECT(Rf_allocVector(VECSXP, o->npen));
    for (int k = 0; k < o->npen; k++) {
        const int ols = o->penalty[k] == OEMGPU_OLS;                    /* vector / scalars for "ols" (ref :282-288) */
        const int nlam = ols ? 1 : nl;
        SEXP b = ols ? Rf_allocVector(REALSXP, rows) : Rf_allocMatrix(REALSXP, rows, nl);
        SET_VECTOR_ELT(lb, k, b);
        memcpy(REAL(b), beta + (size_t)k * nl * rows, sizeof(double) * (size_t)rows * nlam);
        SEXP l = Rf_allocVector(REALSXP, nl);  SET_VECTOR_ELT(ll, k, l);  memcpy(REAL(l), lam + (size_t)k * nl, sizeof(double) * nl);
        SEXP it = Rf_allocVector(INTSXP, nlam); SET_VECTOR_ELT(ln, k, it); memcpy(INTEGER(it), niter + (size_t)k * nl, sizeof(int) * nlam);
        SEXP ls = Rf_allocVector(REALSXP, nlam); SET_VECTOR_ELT(lo, k, ls); memcpy(REAL(ls), loss + (size_t)k * nl, sizeof(double) * nlam);
    }
    SET_VECTOR_ELT(res, 0, lb); SET_VECTOR_ELT(res, 1, ll); SET_VECTOR_ELT(res, 2, ln); SET_VECTOR_ELT(res, 3, lo);
    SET_VECTOR_ELT(res, 4, Rf_ScalarReal(d));
    Rf_setAttrib(res, R_NamesSymbol, names);
    UNPROTECT(6);
    return res;
}

SEXP oem_fit_dense(SEXP x_, SEXP y_, SEXP family_, SEXP penalty_, SEXP weights_, SEXP groups_, SEXP unique_groups_,
                   SEXP group_weights_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_, SEXP gamma_,
                   SEXP tau_, SEXP penalty_factor_, SEXP standardize_, SEXP intercept_, SEXP compute_loss_, SEXP opts_)
{
    if (strcmp(CHAR(STRING_ELT(family_, 0)), "gaussian") != 0)
        Rf_error("binomial not available for oem_fit_dense, use oem_fit_logistic_dense");     /* ref src/oem_dense.cpp:168 */
    /* a non-empty weights_ (R's oem() never sends one: R/oem.R:244) takes the weighted entry, as the reference's compiled code does
     * (ref src/oem_dense.cpp:75,152,162) */
    SEXP dim = Rf_getAttrib(x_, R_DimSymbol);
    const int64_t n = INTEGER(dim)[0];
    const int p = INTEGER(dim)[1];
    oemgpu_opts o;
    fill_opts(&o, penalty_, groups_, unique_groups_, group_weights_, lambda_, nlambda_, lmin_ratio_, alpha_, gamma_, tau_,
              penalty_factor_, compute_loss_, opts_, 1);
    const int nl = o.nlambda_user > 0 ? o.nlambda_user : o.nlambda;
    const size_t nk = (size_t)o.npen * nl;
    double *beta = (double *)R_alloc(nk * (p + 1), sizeof(double)), *lam = (double *)R_alloc(nk, sizeof(double));
    double *loss = (double *)R_alloc(nk, sizeof(double)), d = 0.0;
    int32_t *niter = (int32_t *)R_alloc(nk, sizeof(int32_t));
    /* x and y are read-only for the library: no copy (the reference copies, ref src/oem_dense.cpp:61-67) */
    if (XLENGTH(weights_) > 0 && XLENGTH(weights_) != n) Rf_error("length of weights not same as number of observations in x");
    const int rc = XLENGTH(weights_) > 0
        ? oemgpu_fit_dense_weighted(REAL(x_), n, p, REAL(y_), REAL(weights_), Rf_asLogical(standardize_), Rf_asLogical(intercept_), &o,
                                    beta, lam, niter, loss, &d)
        : oemgpu_fit_dense(REAL(x_), n, p, REAL(y_), Rf_asLogical(standardize_), Rf_asLogical(intercept_), &o,
                           beta, lam, niter, loss, &d);
    if (rc != 0) raise(rc);
    return pack(&o, p + 1, nl, beta, lam, niter, loss, d);
}

SEXP oem_xtx(SEXP xtx_, SEXP xty_, SEXP family_, SEXP penalty_, SEXP groups_, SEXP unique_groups_, SEXP group_weights_,
             SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_, SEXP gamma_, SEXP tau_, SEXP scale_factor_,
             SEXP penalty_factor_, SEXP opts_)
{
    if (strcmp(CHAR(STRING_ELT(family_, 0)), "gaussian") != 0)
        Rf_error("binomial not available for oem_fit_dense, use oem_fit_logistic_dense");
    const int p = INTEGER(Rf_getAttrib(xtx_, R_DimSymbol))[1];
    oemgpu_opts o;
    fill_opts(&o, penalty_, groups_, unique_groups_, group_weights_, lambda_, nlambda_, lmin_ratio_, alpha_, gamma_, tau_,
              penalty_factor_, R_NilValue, opts_, 0);
    const int nl = o.nlambda_user > 0 ? o.nlambda_user : o.nlambda;
    const size_t nk = (size_t)o.npen * nl;
    double *beta = (double *)R_alloc(nk * p, sizeof(double)), *lam = (double *)R_alloc(nk, sizeof(double));
    double *loss = (double *)R_alloc(nk, sizeof(double)), d = 0.0;
    int32_t *niter = (int32_t *)R_alloc(nk, sizeof(int32_t));
    const int rc = oemgpu_fit_xtx(REAL(xtx_), REAL(xty_), p, XLENGTH(scale_factor_) ? REAL(scale_factor_) : NULL, &o,
                                  beta, lam, niter, loss, &d);
    if (rc != 0) raise(rc);
    return pack(&o, p, nl, beta, lam, niter, loss, d);
}

/* oem_fit_big / oem_fit_fb_big (ref src/oem_big.cpp:30, src/oem_fb_big.cpp:30): x_ is the big.matrix external pointer, a C++
 * object of the bigmemory package -- r/oem_shim_big.cpp (a C++ TU against bigmemory's BigMatrix.h, as the reference's own
 * LinkingTo) unwraps it and calls oem_shim_fit_big below with the plain column-major buffer. */
SEXP oem_shim_fit_big(const double *x, int64_t n, int p, SEXP y_, SEXP family_, SEXP penalty_, SEXP weights_, SEXP groups_,
                      SEXP unique_groups_, SEXP group_weights_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_,
                      SEXP gamma_, SEXP tau_, SEXP penalty_factor_, SEXP standardize_, SEXP intercept_, SEXP compute_loss_, SEXP opts_)
{
    if (strcmp(CHAR(STRING_ELT(family_, 0)), "gaussian") != 0)
        Rf_error("binomial not available for oem_fit_big");                                   /* ref src/oem_big.cpp:152 */
    if (XLENGTH(weights_) > 0) Rf_error("weights not implemented yet.");
    oemgpu_opts o;
    /* groups_ arrives with its leading 0 for the intercept (ref R/big_oem.R:254-257) and is passed through unchanged;
     * opts$gigs (the reference's row-slice size, ref src/oem_big.h:738-741) has no meaning here: the library cuts its own
     * row blocks (oem_amd/csrc/hoststream.hip) */
    fill_opts(&o, penalty_, groups_, unique_groups_, group_weights_, lambda_, nlambda_, lmin_ratio_, alpha_, gamma_, tau_,
              penalty_factor_, compute_loss_, opts_, 0);
    const int nl = o.nlambda_user > 0 ? o.nlambda_user : o.nlambda;
    const size_t nk = (size_t)o.npen * nl;
    double *beta = (double *)R_alloc(nk * (p + 1), sizeof(double)), *lam = (double *)R_alloc(nk, sizeof(double));
    double *loss = (double *)R_alloc(nk, sizeof(double)), d = 0.0;
    int32_t *niter = (int32_t *)R_alloc(nk, sizeof(int32_t));
    const double *xs[1] = {x}, *ys[1] = {REAL(y_)};
    const int64_t ns[1] = {n};
    const int rc = oemgpu_fit_big(xs, ns, 1, p, ys, Rf_asLogical(standardize_), Rf_asLogical(intercept_), &o, beta, lam, niter, loss, &d);
    if (rc != 0) raise(rc);
    return pack(&o, p + 1, nl, beta, lam, niter, loss, d);
}

/* oem_xval_dense (ref src/oem_xval_dense.cpp:31-52, called from R/oem_xval.R:497-521): the list of oem_fit_dense plus cvm and cvsd
 * (ref :470-478).  R/oem_xval.R computes lambda.min, lambda.1se, model.min, cvup, cvlo, nzero from it, unchanged. */
SEXP oem_xval_dense(SEXP x_, SEXP y_, SEXP family_, SEXP penalty_, SEXP weights_, SEXP groups_, SEXP unique_groups_,
                    SEXP group_weights_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_, SEXP gamma_,
                    SEXP tau_, SEXP penalty_factor_, SEXP standardize_, SEXP intercept_, SEXP nfolds_, SEXP foldid_,
                    SEXP compute_loss_, SEXP type_measure_, SEXP opts_)
{
    if (strcmp(CHAR(STRING_ELT(family_, 0)), "gaussian") != 0)
        Rf_error("binomial not available for oem_xval_dense, use oem_xval_logistic_dense");    /* ref src/oem_xval_dense.cpp:157 */
    if (XLENGTH(weights_) > 0 && Rf_asLogical(compute_loss_))
        Rf_error("compute.loss with observation weights is outside the GPU path (the reference's loss is unweighted)");
    SEXP dim = Rf_getAttrib(x_, R_DimSymbol);
    const int64_t n = INTEGER(dim)[0];
    const int p = INTEGER(dim)[1];
    oemgpu_opts o;
    fill_opts(&o, penalty_, groups_, unique_groups_, group_weights_, lambda_, nlambda_, lmin_ratio_, alpha_, gamma_, tau_,
              penalty_factor_, compute_loss_, opts_, 0);
    const int nl = o.nlambda_user > 0 ? o.nlambda_user : o.nlambda;
    const size_t nk = (size_t)o.npen * nl;
    double *beta = (double *)R_alloc(nk * (p + 1), sizeof(double)), *lam = (double *)R_alloc(nk, sizeof(double));
    double *loss = (double *)R_alloc(nk, sizeof(double)), *cvm = (double *)R_alloc(nk, sizeof(double)),
           *cvsd = (double *)R_alloc(nk, sizeof(double)), d = 0.0;
    int32_t *niter = (int32_t *)R_alloc(nk, sizeof(int32_t));
    const int mae = strcmp(CHAR(STRING_ELT(type_measure_, 0)), "mae") == 0;                    /* ref :378-411 */
    const int rc = oemgpu_xval_dense(REAL(x_), n, p, REAL(y_), XLENGTH(weights_) > 0 ? REAL(weights_) : NULL, INTEGER(foldid_),
                                     Rf_asInteger(nfolds_), Rf_asLogical(standardize_),
                                     Rf_asLogical(intercept_), mae, &o, beta, lam, niter, loss, &d, cvm, cvsd);
    if (rc != 0) raise(rc);
    SEXP base = PROTECT(pack(&o, p + 1, nl, beta, lam, niter, loss, d));
    /* beta, lambda, niter, loss, cvm, cvsd, d */
    SEXP res = PROTECT(Rf_allocVector(VECSXP, 7)), names = PROTECT(Rf_allocVector(STRSXP, 7));
    const char *nm[7] = {"beta", "lambda", "niter", "loss", "cvm", "cvsd", "d"};
    for (int i = 0; i < 7; i++) SET_STRING_ELT(names, i, Rf_mkChar(nm[i]));
    for (int i = 0; i < 4; i++) SET_VECTOR_ELT(res, i, VECTOR_ELT(base, i));
    SEXP lm = PROTECT(Rf_allocVector(VECSXP, o.npen)), ls = PROTECT(Rf_allocVector(VECSXP, o.npen));
    for (int k = 0; k < o.npen; k++) {
        const int nlam = (o.penalty[k] == OEMGPU_OLS) ? 1 : nl;
        SEXP m = Rf_allocVector(REALSXP, nlam); SET_VECTOR_ELT(lm, k, m); memcpy(REAL(m), cvm + (size_t)k * nl, sizeof(double) * nlam);
        SEXP s = Rf_allocVector(REALSXP, nlam); SET_VECTOR_ELT(ls, k, s); memcpy(REAL(s), cvsd + (size_t)k * nl, sizeof(double) * nlam);
    }
    SET_VECTOR_ELT(res, 4, lm); SET_VECTOR_ELT(res, 5, ls); SET_VECTOR_ELT(res, 6, VECTOR_ELT(base, 4));
    Rf_setAttrib(res, R_NamesSymbol, names);
    UNPROTECT(5);
    return res;
}

/* oem_fit_sparse (ref src/oem_sparse.cpp:30-50, called from R/oem.R:534-553): x_ is a dgCMatrix; its slots are the compressed
 * sparse column arrays the library takes.  (The column pointers of a dgCMatrix are 32-bit: widened here.) */
SEXP oem_fit_sparse(SEXP x_, SEXP y_, SEXP family_, SEXP penalty_, SEXP weights_, SEXP groups_, SEXP unique_groups_,
                    SEXP group_weights_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_, SEXP gamma_,
                    SEXP tau_, SEXP penalty_factor_, SEXP standardize_, SEXP intercept_, SEXP compute_loss_, SEXP opts_)
{
    if (strcmp(CHAR(STRING_ELT(family_, 0)), "gaussian") != 0)
        Rf_error("binomial not available for oem_fit_sparse, use oem_fit_logistic_sparse");    /* ref src/oem_sparse.cpp:150 */
    if (XLENGTH(weights_) > 0) Rf_error("weights not implemented yet.");
    SEXP dim = R_do_slot(x_, Rf_install("Dim")), ip = R_do_slot(x_, Rf_install("p")), ii = R_do_slot(x_, Rf_install("i")),
         xv = R_do_slot(x_, Rf_install("x"));
    const int64_t n = INTEGER(dim)[0];
    const int p = INTEGER(dim)[1];
    int64_t *colptr = (int64_t *)R_alloc((size_t)p + 1, sizeof(int64_t));
    for (int j = 0; j <= p; j++) colptr[j] = INTEGER(ip)[j];
    oemgpu_opts o;
    fill_opts(&o, penalty_, groups_, unique_groups_, group_weights_, lambda_, nlambda_, lmin_ratio_, alpha_, gamma_, tau_,
              penalty_factor_, compute_loss_, opts_, 0);
    const int nl = o.nlambda_user > 0 ? o.nlambda_user : o.nlambda;
    const size_t nk = (size_t)o.npen * nl;
    double *beta = (double *)R_alloc(nk * (p + 1), sizeof(double)), *lam = (double *)R_alloc(nk, sizeof(double));
    double *loss = (double *)R_alloc(nk, sizeof(double)), d = 0.0;
    int32_t *niter = (int32_t *)R_alloc(nk, sizeof(int32_t));
    const int rc = oemgpu_fit_sparse(n, p, colptr, INTEGER(ii), REAL(xv), REAL(y_), Rf_asLogical(standardize_),
                                     Rf_asLogical(intercept_), &o, beta, lam, niter, loss, &d);
    if (rc != 0) raise(rc);
    return pack(&o, p + 1, nl, beta, lam, niter, loss, d);
}

/* The host-resident entry points keep contexts (streams, pinned staging lanes, workspaces; the device copy of the rows only up to
 * OEMGPU_CACHE_KEEP_BYTES, default an eighth of the device's memory) in a process-wide cache so that repeated calls allocate
 * nothing.  `.Call("oem_gpu_release_cache", PACKAGE = "oem")` frees all of it on demand, and unloading the package's shared
 * object (library.dynam.unload / detach(unload = TRUE)) does the same through R's unload hook. */
SEXP oem_gpu_release_cache(void)
{
    oemgpu_release_cache();
    return R_NilValue;
}

void R_unload_oem(DllInfo *info)
{
    (void)info;
    oemgpu_release_cache();
}
