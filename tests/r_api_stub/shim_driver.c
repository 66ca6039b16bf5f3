/* tests/r_api_stub/shim_driver.c -- TESTS ONLY.  Calls the `.Call` entry points of r/oem_shim.c and r/oem_shim_big.cpp the way
 * R/oem.R:556-575, R/oem_xtx.R:389-406, R/big_oem.R:447-491, R/oem_xval.R:497-521 and R/oem.R:534-553 call the reference's, over
 * the stand-in runtime of this directory and the recording fake of liboemgpu, and checks
 *   (1) what reaches the C ABI (argument order of the 19 / 16 / 22 SEXPs -> fields of oemgpu_opts, no copies of x),
 *   (2) what comes back (the list of ref src/oem_dense.cpp:280-307: names, storage modes, dimensions, "ols" as a vector),
 *   (3) that the protect stack is where it was on every exit path, under collect-on-every-allocation,
 *   (4) errors and user interrupts (Rf_error with the library's text; Rf_onintr after OEMGPU_ERR_INTERRUPTED).
 * Prints "shim driver: N checks passed" or aborts. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "R.h"
#include "R_ext/Rdynload.h"
#include "fake_oemgpu.h"
#include "r_stub_runtime.h"

SEXP oem_fit_dense(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP oem_xtx(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP oem_fit_big(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP oem_fit_fb_big(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP oem_fit_sparse(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP oem_xval_dense(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP,
                    SEXP, SEXP, SEXP);
SEXP oem_gpu_release_cache(void);
void R_unload_oem(DllInfo *);
void *driver_big_matrix(double *data, long nrow, long ncol, int type, int sepcols, long total_rows, long row_offset);   /* big_matrix_maker.cpp */

static int checks;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "shim driver: line %d: %s\n", __LINE__, #c); abort(); } checks++; } while (0)

static SEXP R(SEXP x) { stub_root(x); return x; }          /* an argument of the running .Call */
static SEXP real1(double v) { return R(stub_real(&v, 1)); }
static SEXP int1(int v) { return R(stub_int(&v, 1)); }
static SEXP str1(const char *s) { return R(stub_str(&s, 1)); }
static SEXP empty_real(void) { return R(stub_real(NULL, 0)); }
static SEXP empty_int(void) { return R(stub_int(NULL, 0)); }

enum { N = 6, P = 3, NL = 4 };
static double X[N * P], Y[N], PF[P] = {1.0, 0.5, 2.0};

/* opts as R/oem.R:430-445 builds it (+ the two additions of this binding when asked) */
static SEXP make_opts(int ngpus, const double *devices, int ndev)
{
    const char *names[10] = {"maxit", "tol", "irls_maxit", "irls_tol", "accelerate", "ncores", "hessian.type", "gigs", "ngpus", "devices"};
    const int nbase = 8, n = nbase + (ngpus > 0) + (ndev > 0);
    const char *use[10];
    SEXP o = R(stub_list(n));
    int maxit = 321, irls = 100, nc = 1, acc = 1;
    double tol = 1e-9, irls_tol = 1e-3, gigs = 4.0;
    const char *hess = "upper.bound";
    int k = 0;
    use[k] = names[0]; SET_VECTOR_ELT(o, k++, stub_int(&maxit, 1));
    use[k] = names[1]; SET_VECTOR_ELT(o, k++, stub_real(&tol, 1));
    use[k] = names[2]; SET_VECTOR_ELT(o, k++, stub_int(&irls, 1));
    use[k] = names[3]; SET_VECTOR_ELT(o, k++, stub_real(&irls_tol, 1));
    use[k] = names[4]; SET_VECTOR_ELT(o, k++, stub_lgl(acc));
    use[k] = names[5]; SET_VECTOR_ELT(o, k++, stub_int(&nc, 1));
    use[k] = names[6]; SET_VECTOR_ELT(o, k++, stub_str(&hess, 1));
    use[k] = names[7]; SET_VECTOR_ELT(o, k++, stub_real(&gigs, 1));
    if (ngpus > 0) { use[k] = names[8]; SET_VECTOR_ELT(o, k++, stub_int(&ngpus, 1)); }
    if (ndev > 0) { use[k] = names[9]; SET_VECTOR_ELT(o, k++, stub_real(devices, ndev)); }      /* c(2, 5) is a double vector in R */
    stub_set_names(o, use);
    return o;
}

static SEXP lambda_list(int npen, int nlu)
{
    SEXP l = R(stub_list(npen));
    for (int k = 0; k < npen; k++) {
        double v[8];
        for (int i = 0; i < nlu; i++) v[i] = 1.0 + k - 0.1 * i;
        SET_VECTOR_ELT(l, k, stub_real(v, nlu));
    }
    return l;
}

static int has_names(SEXP list, const char *const *nm, int n)
{
    SEXP names = Rf_getAttrib(list, R_NamesSymbol);
    if (TYPEOF(names) != STRSXP || XLENGTH(names) != n) return 0;
    for (int i = 0; i < n; i++) if (strcmp(CHAR(STRING_ELT(names, i)), nm[i]) != 0) return 0;
    return 1;
}

/* the list of ref src/oem_dense.cpp:280-307 for penalties pen[0..npen) (code 2 = "ols": a plain vector and scalars) */
static void check_fit_list(SEXP res, const int *pen, int npen, int rows, int nl, int compute_loss)
{
    const char *nm[5] = {"beta", "lambda", "niter", "loss", "d"};
    CHECK(TYPEOF(res) == VECSXP && XLENGTH(res) == 5 && has_names(res, nm, 5));
    for (int f = 0; f < 4; f++) CHECK(TYPEOF(VECTOR_ELT(res, f)) == VECSXP && XLENGTH(VECTOR_ELT(res, f)) == npen);
    CHECK(TYPEOF(VECTOR_ELT(res, 4)) == REALSXP && XLENGTH(VECTOR_ELT(res, 4)) == 1 && REAL(VECTOR_ELT(res, 4))[0] == FAKE_D);
    for (int k = 0; k < npen; k++) {
        const int ols = pen[k] == OEMGPU_OLS, nlam = ols ? 1 : nl;
        SEXP b = VECTOR_ELT(VECTOR_ELT(res, 0), k), l = VECTOR_ELT(VECTOR_ELT(res, 1), k), it = VECTOR_ELT(VECTOR_ELT(res, 2), k),
             ls = VECTOR_ELT(VECTOR_ELT(res, 3), k);
        CHECK(TYPEOF(b) == REALSXP && XLENGTH(b) == (R_xlen_t)rows * nlam);
        SEXP dim = Rf_getAttrib(b, R_DimSymbol);
        if (ols) CHECK(dim == R_NilValue);
        else CHECK(TYPEOF(dim) == INTSXP && XLENGTH(dim) == 2 && INTEGER(dim)[0] == rows && INTEGER(dim)[1] == nl);
        for (int i = 0; i < nlam; i++)
            for (int j = 0; j < rows; j++) CHECK(REAL(b)[(size_t)i * rows + j] == fake_beta(k, i, j));
        CHECK(TYPEOF(l) == REALSXP && XLENGTH(l) == nl);
        for (int i = 0; i < nl; i++) CHECK(REAL(l)[i] == fake_lambda(k, i));
        CHECK(TYPEOF(it) == INTSXP && XLENGTH(it) == nlam);
        for (int i = 0; i < nlam; i++) CHECK(INTEGER(it)[i] == fake_niter(k, i));
        CHECK(TYPEOF(ls) == REALSXP && XLENGTH(ls) == nlam);
        for (int i = 0; i < nlam; i++) CHECK(REAL(ls)[i] == (compute_loss ? 2.0 * fake_cvm(k, i) : 1e99));
    }
}

static void check_common_opts(int npen, const int *pen, int nlambda, int nlu, int accelerate, int compute_loss)
{
    CHECK(fake.o.npen == npen);
    for (int k = 0; k < npen; k++) CHECK(fake.o.penalty[k] == pen[k]);
    CHECK(fake.o.nlambda == nlambda && fake.o.nlambda_user == nlu && fake.o.lambda_min_ratio == 1e-3);
    CHECK((nlu == 0) == (fake.o.lambda_user == NULL));
    for (int k = 0; k < npen && nlu; k++)
        for (int i = 0; i < nlu; i++) CHECK(fake.o.lambda_user[k * nlu + i] == 1.0 + k - 0.1 * i);
    CHECK(fake.o.alpha == 0.75 && fake.o.gamma == 3.5 && fake.o.tau == 0.25);
    CHECK(fake.o.tol == 1e-9 && fake.o.maxit == 321 && fake.o.accelerate == accelerate && fake.o.compute_loss == compute_loss);
    CHECK(fake.o.device == -1 && fake.o.interrupt != NULL);
}

int main(void)
{
    for (int i = 0; i < N * P; i++) X[i] = 0.25 * i;
    for (int i = 0; i < N; i++) Y[i] = 1.0 + i;
    const char *pens[3] = {"lasso", "ols", "grp.lasso"};
    const int pen_codes[3] = {OEMGPU_LASSO, OEMGPU_OLS, OEMGPU_GRP_LASSO};
    const int groups[P] = {1, 1, 2}, ugroups[2] = {1, 2};
    const double gw[2] = {1.5, 1.0};

    /* ---- oem_fit_dense: three penalties, generated grid, groups, compute.loss, accelerate ---- */
    {
        SEXP x = R(stub_real_matrix(X, N, P)), y = R(stub_real(Y, N)), g = R(stub_int(groups, P)), ug = R(stub_int(ugroups, 2));
        SEXP pf = R(stub_real(PF, P)), gwv = R(stub_real(gw, 2));
        if (setjmp(stub_jmp)) { fprintf(stderr, "unexpected R error: %s\n", stub_error_msg); abort(); }
        SEXP res = oem_fit_dense(x, y, str1("gaussian"), R(stub_str(pens, 3)), empty_real(), g, ug, gwv, lambda_list(3, 0), int1(NL),
                                 real1(1e-3), real1(0.75), real1(3.5), real1(0.25), pf, R(stub_lgl(1)), R(stub_lgl(0)), R(stub_lgl(1)),
                                 make_opts(0, NULL, 0));
        CHECK(stub_protect_depth() == 0);
        R(res);
        CHECK(strcmp(fake.entry, "oemgpu_fit_dense") == 0);
        CHECK(fake.x == (const void *)REAL(x) && fake.y == (const void *)REAL(y) && fake.n == N && fake.p == P);    /* no copy */
        CHECK(fake.standardize == 1 && fake.intercept == 0);
        check_common_opts(3, pen_codes, NL, 0, 1, 1);
        CHECK(fake.o.penalty_factor == REAL(pf) && fake.o.groups == INTEGER(g) && fake.o.ngroupvars == P);
        CHECK(fake.o.unique_groups == INTEGER(ug) && fake.o.ngroups == 2 && fake.o.group_weights == REAL(gwv) && fake.o.n_group_weights == 2);
        CHECK(fake.o.ngpus == 0 && fake.o.devices == NULL);
        check_fit_list(res, pen_codes, 3, P + 1, NL, 1);
        stub_end_call();
    }
    /* ---- user lambdas (one row per penalty), no groups, devices = c(2, 5) ---- */
    {
        const double dev[2] = {2, 5};
        SEXP x = R(stub_real_matrix(X, N, P)), y = R(stub_real(Y, N));
        if (setjmp(stub_jmp)) { fprintf(stderr, "unexpected R error: %s\n", stub_error_msg); abort(); }
        SEXP res = oem_fit_dense(x, y, str1("gaussian"), R(stub_str(pens, 2)), empty_real(), empty_int(), empty_int(), empty_real(),
                                 lambda_list(2, 3), int1(NL), real1(1e-3), real1(0.75), real1(3.5), real1(0.25), R(stub_real(PF, P)),
                                 R(stub_lgl(0)), R(stub_lgl(1)), R(stub_lgl(0)), make_opts(0, dev, 2));
        CHECK(stub_protect_depth() == 0);
        R(res);
        check_common_opts(2, pen_codes, NL, 3, 1, 0);
        CHECK(fake.o.groups == NULL && fake.o.unique_groups == NULL && fake.o.group_weights == NULL && fake.o.ngroups == 0);
        CHECK(fake.o.ngpus == 2 && fake.o.devices && fake.o.devices[0] == 2 && fake.o.devices[1] == 5);
        CHECK(fake.standardize == 0 && fake.intercept == 1);
        check_fit_list(res, pen_codes, 2, P + 1, 3, 0);
        stub_end_call();
    }
    /* ---- weights: the weighted entry; a wrong length, an unknown penalty, another family, a failing library call: R errors ---- */
    {
        double w[N] = {1, 2, 1, 0.5, 1, 3};
        for (int variant = 0; variant < 5; variant++) {
            const char *bad = "ridge";
            SEXP x = R(stub_real_matrix(X, N, P)), y = R(stub_real(Y, N));
            SEXP wv = R(stub_real(w, variant == 1 ? N - 1 : N));
            SEXP pen = variant == 2 ? str1(bad) : R(stub_str(pens, 1));
            SEXP fam = str1(variant == 3 ? "binomial" : "gaussian");
            fake_rc = variant == 4 ? OEMGPU_ERR_HIP : 0;
            const int jumped = setjmp(stub_jmp);
            if (!jumped) {
                SEXP res = oem_fit_dense(x, y, fam, pen, wv, empty_int(), empty_int(), empty_real(), lambda_list(1, 0), int1(NL), real1(1e-3),
                                         real1(0.75), real1(3.5), real1(0.25), R(stub_real(PF, P)), R(stub_lgl(1)), R(stub_lgl(1)), R(stub_lgl(0)),
                                         make_opts(3, NULL, 0));
                CHECK(variant == 0);
                R(res);
                CHECK(strcmp(fake.entry, "oemgpu_fit_dense_weighted") == 0 && fake.weights == (const void *)REAL(wv) && fake.o.ngpus == 3);
                check_fit_list(res, pen_codes, 1, P + 1, NL, 0);
            } else {
                CHECK(jumped == 1 && variant != 0);
                if (variant == 1) CHECK(strstr(stub_error_msg, "length of weights"));
                if (variant == 2) CHECK(strstr(stub_error_msg, "unknown penalty 'ridge'"));
                if (variant == 3) CHECK(strstr(stub_error_msg, "binomial not available"));
                if (variant == 4) CHECK(strcmp(stub_error_msg, oemgpu_last_error()) == 0);
            }
            CHECK(stub_protect_depth() == 0);
            fake_rc = 0;
            stub_end_call();
        }
    }
    /* ---- a user interrupt: the library polls opts->interrupt, returns INTERRUPTED, the shim re-raises it ---- */
    for (int pending = 0; pending < 2; pending++) {
        SEXP x = R(stub_real_matrix(X, N, P)), y = R(stub_real(Y, N));
        fake_poll_interrupt = 1; stub_pending_interrupt = pending;
        const int jumped = setjmp(stub_jmp);
        if (!jumped) {
            SEXP res = oem_fit_dense(x, y, str1("gaussian"), R(stub_str(pens, 1)), empty_real(), empty_int(), empty_int(), empty_real(),
                                     lambda_list(1, 0), int1(NL), real1(1e-3), real1(0.75), real1(3.5), real1(0.25), R(stub_real(PF, P)),
                                     R(stub_lgl(1)), R(stub_lgl(1)), R(stub_lgl(0)), make_opts(0, NULL, 0));
            CHECK(!pending && fake.interrupt_answer == 0);
            R(res);
            check_fit_list(res, pen_codes, 1, P + 1, NL, 0);
        } else
            CHECK(pending && jumped == 2 && fake.interrupt_answer != 0);          /* Rf_onintr, not Rf_error */
        CHECK(stub_protect_depth() == 0);
        fake_poll_interrupt = 0; stub_pending_interrupt = 0;
        stub_end_call();
    }
    /* ---- oem_xtx: 16 arguments, p rows, scale.factor or none ---- */
    for (int with_sf = 0; with_sf < 2; with_sf++) {
        double xtx[P * P], xty[P] = {1, 2, 3}, sf[P] = {1, 2, 4};
        for (int i = 0; i < P * P; i++) xtx[i] = i;
        SEXP a = R(stub_real_matrix(xtx, P, P)), b = R(stub_real(xty, P)), s = with_sf ? R(stub_real(sf, P)) : empty_real();
        if (setjmp(stub_jmp)) { fprintf(stderr, "unexpected R error: %s\n", stub_error_msg); abort(); }
        SEXP res = oem_xtx(a, b, str1("gaussian"), R(stub_str(pens, 2)), empty_int(), empty_int(), empty_real(), lambda_list(2, 0), int1(NL),
                           real1(1e-3), real1(0.75), real1(3.5), real1(0.25), s, R(stub_real(PF, P)), make_opts(0, NULL, 0));
        CHECK(stub_protect_depth() == 0);
        R(res);
        CHECK(strcmp(fake.entry, "oemgpu_fit_xtx") == 0 && fake.x == (const void *)REAL(a) && fake.xty == (const void *)REAL(b) && fake.p == P);
        CHECK(with_sf ? fake.scale_factor == (const void *)REAL(s) : fake.scale_factor == NULL);
        check_common_opts(2, pen_codes, NL, 0, 0, 0);                              /* oem.xtx has neither accelerate nor compute.loss */
        check_fit_list(res, pen_codes, 2, P, NL, 0);
        stub_end_call();
    }
    /* ---- oem_xval_dense: 22 arguments; the list gains cvm and cvsd (ref src/oem_xval_dense.cpp:470-478) ---- */
    for (int mae = 0; mae < 2; mae++) {
        int foldid[N] = {1, 2, 3, 1, 2, 3};
        double w[N] = {1, 1, 2, 2, 1, 1};
        SEXP x = R(stub_real_matrix(X, N, P)), y = R(stub_real(Y, N)), f = R(stub_int(foldid, N)), wv = mae ? R(stub_real(w, N)) : empty_real();
        if (setjmp(stub_jmp)) { fprintf(stderr, "unexpected R error: %s\n", stub_error_msg); abort(); }
        SEXP res = oem_xval_dense(x, y, str1("gaussian"), R(stub_str(pens, 2)), wv, empty_int(), empty_int(), empty_real(), lambda_list(2, 0),
                                  int1(NL), real1(1e-3), real1(0.75), real1(3.5), real1(0.25), R(stub_real(PF, P)), R(stub_lgl(1)), R(stub_lgl(1)),
                                  int1(3), f, R(stub_lgl(0)), str1(mae ? "mae" : "mse"), make_opts(0, NULL, 0));
        CHECK(stub_protect_depth() == 0);
        R(res);
        CHECK(strcmp(fake.entry, "oemgpu_xval_dense") == 0 && fake.foldid == INTEGER(f) && fake.nfolds == 3 && fake.type_measure == mae);
        CHECK(mae ? fake.weights == (const void *)REAL(wv) : fake.weights == NULL);
        const char *nm[7] = {"beta", "lambda", "niter", "loss", "cvm", "cvsd", "d"};
        CHECK(TYPEOF(res) == VECSXP && XLENGTH(res) == 7 && has_names(res, nm, 7));
        CHECK(REAL(VECTOR_ELT(res, 6))[0] == FAKE_D);
        for (int k = 0; k < 2; k++) {
            const int nlam = k == 1 ? 1 : NL;                                       /* "ols": single entries */
            SEXP m = VECTOR_ELT(VECTOR_ELT(res, 4), k), s = VECTOR_ELT(VECTOR_ELT(res, 5), k), b = VECTOR_ELT(VECTOR_ELT(res, 0), k);
            CHECK(TYPEOF(m) == REALSXP && XLENGTH(m) == nlam && TYPEOF(s) == REALSXP && XLENGTH(s) == nlam);
            for (int i = 0; i < nlam; i++) CHECK(REAL(m)[i] == fake_cvm(k, i) && REAL(s)[i] == fake_cvm(k, i) / 4.0);
            CHECK(REAL(b)[1] == fake_beta(k, 0, 1));
        }
        stub_end_call();
    }
    /* ---- oem_fit_sparse: a dgCMatrix's slots, column pointers widened to 64 bit ---- */
    {
        int dim[2] = {N, P}, cp[P + 1] = {0, 2, 3, 5}, ri[5] = {0, 4, 2, 1, 5};
        double xv[5] = {1, 2, 3, 4, 5};
        SEXP m = R(stub_s4()), ii = R(stub_int(ri, 5)), vv = R(stub_real(xv, 5)), y = R(stub_real(Y, N));
        Rf_setAttrib(m, Rf_install("Dim"), R(stub_int(dim, 2)));
        Rf_setAttrib(m, Rf_install("p"), R(stub_int(cp, P + 1)));
        Rf_setAttrib(m, Rf_install("i"), ii);
        Rf_setAttrib(m, Rf_install("x"), vv);
        if (setjmp(stub_jmp)) { fprintf(stderr, "unexpected R error: %s\n", stub_error_msg); abort(); }
        SEXP res = oem_fit_sparse(m, y, str1("gaussian"), R(stub_str(pens, 1)), empty_real(), empty_int(), empty_int(), empty_real(),
                                  lambda_list(1, 0), int1(NL), real1(1e-3), real1(0.75), real1(3.5), real1(0.25), R(stub_real(PF, P)),
                                  R(stub_lgl(1)), R(stub_lgl(1)), R(stub_lgl(1)), make_opts(0, NULL, 0));
        CHECK(stub_protect_depth() == 0);
        R(res);
        CHECK(strcmp(fake.entry, "oemgpu_fit_sparse") == 0 && fake.n == N && fake.p == P && fake.rowidx == INTEGER(ii) && fake.values == REAL(vv));
        for (int j = 0; j <= P; j++) CHECK(fake.colptr[j] == cp[j]);
        check_fit_list(res, pen_codes, 1, P + 1, NL, 1);
        stub_end_call();
    }
    /* ---- oem_fit_big / oem_fit_fb_big: the big.matrix external pointer (r/oem_shim_big.cpp), and what it refuses ---- */
    for (int variant = 0; variant < 5; variant++) {
        /* 0: in-memory, 1: file-backed, 2: integer matrix, 3: separated columns, 4: a sub.big.matrix */
        void *bm = driver_big_matrix(X, N, P, variant == 2 ? 4 : 8, variant == 3, variant == 4 ? N + 2 : N, variant == 4 ? 1 : 0);
        SEXP xp = R(stub_extptr(bm)), y = R(stub_real(Y, N));
        const int g0[P + 1] = {0, 1, 1, 2};                                             /* R/big_oem.R:254-257: a leading 0 for the intercept */
        SEXP g = R(stub_int(g0, P + 1));
        const int jumped = setjmp(stub_jmp);
        if (!jumped) {
            SEXP (*entry)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP) =
                variant == 1 ? oem_fit_fb_big : oem_fit_big;
            SEXP res = entry(xp, y, str1("gaussian"), R(stub_str(pens, 3)), empty_real(), g, R(stub_int(ugroups, 2)), empty_real(),
                             lambda_list(3, 0), int1(NL), real1(1e-3), real1(0.75), real1(3.5), real1(0.25), R(stub_real(PF, P)),
                             R(stub_lgl(1)), R(stub_lgl(1)), R(stub_lgl(0)), make_opts(0, NULL, 0));
            CHECK(variant <= 1);
            R(res);
            CHECK(strcmp(fake.entry, "oemgpu_fit_big") == 0 && fake.x == (const void *)X && fake.n == N && fake.p == P && fake.nshards == 1);
            CHECK(fake.o.groups == INTEGER(g) && fake.o.ngroupvars == P + 1 && fake.o.accelerate == 0);
            check_fit_list(res, pen_codes, 3, P + 1, NL, 0);
        } else {
            CHECK(jumped == 1 && variant >= 2);
            if (variant == 2) CHECK(strstr(stub_error_msg, "type for provided big.matrix not available"));
            if (variant == 3) CHECK(strstr(stub_error_msg, "separated columns"));
            if (variant == 4) CHECK(strstr(stub_error_msg, "sub.big.matrix"));
        }
        CHECK(stub_protect_depth() == 0);
        stub_end_call();
        free(bm);
    }
    /* ---- the cache hooks ---- */
    {
        const int before = fake.releases;
        CHECK(oem_gpu_release_cache() == R_NilValue && fake.releases == before + 1);
        R_unload_oem(NULL);
        CHECK(fake.releases == before + 2);
    }
    printf("shim driver: %d checks passed\n", checks);
    return 0;
}
