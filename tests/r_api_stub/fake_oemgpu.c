/* tests/r_api_stub/fake_oemgpu.c -- TESTS ONLY.  A RECORDING FAKE of the liboemgpu entry points that r/oem_shim.c calls: it computes
 * nothing, remembers its arguments and fills the outputs with a pattern (fake_beta ...), so that tests/test_r_shim.py can check what
 * the shim passes down and what it packs up.  Never linked into anything but the shim test. */
#include <stddef.h>
#include <string.h>

#include "fake_oemgpu.h"

struct fake_record fake;
int fake_rc, fake_poll_interrupt;
static const char *fake_msg = "fake liboemgpu: the error text of oemgpu_last_error()";

double fake_beta(int k, int i, int j) { return 100.0 * k + i + j / 1024.0; }
double fake_lambda(int k, int i) { return 10.0 * k + 1.0 / (1 + i); }
int fake_niter(int k, int i) { return 7 + 3 * k + i; }
double fake_cvm(int k, int i) { return 0.5 + k + i / 128.0; }

static int finish(const char *entry, const oemgpu_opts *o, int rows, double *beta, double *lam, int32_t *niter, double *loss, double *d)
{
    fake.entry = entry;
    fake.o = *o;
    if (fake_poll_interrupt && o->interrupt) {
        fake.interrupt_answer = o->interrupt(o->interrupt_arg);
        if (fake.interrupt_answer) return OEMGPU_ERR_INTERRUPTED;
    }
    if (fake_rc) return fake_rc;
    const int nl = o->nlambda_user > 0 ? o->nlambda_user : o->nlambda;
    for (int k = 0; k < o->npen; k++)
        for (int i = 0; i < nl; i++) {
            for (int j = 0; j < rows; j++) beta[((size_t)k * nl + i) * rows + j] = fake_beta(k, i, j);
            lam[k * nl + i] = fake_lambda(k, i);
            niter[k * nl + i] = fake_niter(k, i);
            loss[k * nl + i] = o->compute_loss ? 2.0 * fake_cvm(k, i) : 1e99;
        }
    *d = FAKE_D;
    return 0;
}

int oemgpu_fit_dense(const double *x, int64_t n, int32_t p, const double *y, int32_t standardize, int32_t intercept,
                     const oemgpu_opts *o, double *beta, double *lam, int32_t *niter, double *loss, double *d)
{
    memset(&fake, 0, offsetof(struct fake_record, releases));
    fake.x = x; fake.n = n; fake.p = p; fake.y = y; fake.standardize = standardize; fake.intercept = intercept;
    return finish("oemgpu_fit_dense", o, p + 1, beta, lam, niter, loss, d);
}

int oemgpu_fit_dense_weighted(const double *x, int64_t n, int32_t p, const double *y, const double *w, int32_t standardize,
                              int32_t intercept, const oemgpu_opts *o, double *beta, double *lam, int32_t *niter, double *loss, double *d)
{
    memset(&fake, 0, offsetof(struct fake_record, releases));
    fake.x = x; fake.n = n; fake.p = p; fake.y = y; fake.weights = w; fake.standardize = standardize; fake.intercept = intercept;
    return finish("oemgpu_fit_dense_weighted", o, p + 1, beta, lam, niter, loss, d);
}

int oemgpu_fit_xtx(const double *xtx, const double *xty, int32_t p, const double *sf, const oemgpu_opts *o,
                   double *beta, double *lam, int32_t *niter, double *loss, double *d)
{
    memset(&fake, 0, offsetof(struct fake_record, releases));
    fake.x = xtx; fake.xty = xty; fake.p = p; fake.scale_factor = sf;
    return finish("oemgpu_fit_xtx", o, p, beta, lam, niter, loss, d);
}

int oemgpu_fit_big(const double *const *xs, const int64_t *ns, int32_t nshards, int32_t p, const double *const *ys,
                   int32_t standardize, int32_t intercept, const oemgpu_opts *o, double *beta, double *lam, int32_t *niter,
                   double *loss, double *d)
{
    memset(&fake, 0, offsetof(struct fake_record, releases));
    fake.x = xs[0]; fake.n = ns[0]; fake.nshards = nshards; fake.p = p; fake.y = ys[0]; fake.standardize = standardize; fake.intercept = intercept;
    return finish("oemgpu_fit_big", o, p + 1, beta, lam, niter, loss, d);
}

int oemgpu_fit_sparse(int64_t n, int32_t p, const int64_t *colptr, const int32_t *rowidx, const double *values, const double *y,
                      int32_t standardize, int32_t intercept, const oemgpu_opts *o, double *beta, double *lam, int32_t *niter,
                      double *loss, double *d)
{
    memset(&fake, 0, offsetof(struct fake_record, releases));
    fake.n = n; fake.p = p; fake.colptr = colptr; fake.rowidx = rowidx; fake.values = values; fake.y = y;
    fake.standardize = standardize; fake.intercept = intercept;
    return finish("oemgpu_fit_sparse", o, p + 1, beta, lam, niter, loss, d);
}

int oemgpu_xval_dense(const double *x, int64_t n, int32_t p, const double *y, const double *w, const int32_t *foldid, int32_t nfolds,
                      int32_t standardize, int32_t intercept, int32_t type_measure, const oemgpu_opts *o, double *beta, double *lam,
                      int32_t *niter, double *loss, double *d, double *cvm, double *cvsd)
{
    memset(&fake, 0, offsetof(struct fake_record, releases));
    fake.x = x; fake.n = n; fake.p = p; fake.y = y; fake.weights = w; fake.foldid = foldid; fake.nfolds = nfolds;
    fake.standardize = standardize; fake.intercept = intercept; fake.type_measure = type_measure;
    const int rc = finish("oemgpu_xval_dense", o, p + 1, beta, lam, niter, loss, d);
    if (rc) return rc;
    const int nl = o->nlambda_user > 0 ? o->nlambda_user : o->nlambda;
    for (int k = 0; k < o->npen; k++)
        for (int i = 0; i < nl; i++) { cvm[k * nl + i] = fake_cvm(k, i); cvsd[k * nl + i] = fake_cvm(k, i) / 4.0; }
    return 0;
}

const char *oemgpu_last_error(void) { return fake_msg; }
void oemgpu_release_cache(void) { fake.releases++; }
