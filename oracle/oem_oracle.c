/*
 * oem_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See oem_oracle.h for scope, pinning status and the third-party notes.
 * All "ref:" citations are paths under /root/reference/.
 */
#include "oem_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[256];
const char *orc_last_error(void) { return g_err; }
static int fail(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return -1; }

/* ------------------------------------------------------------------ */
/* penalty name predicates (ref: src/oem_dense.h:424-426, oem_dense.cpp:204,213) */
static int pen_is_grp(int pen) { return pen >= ORC_GRP_LASSO; }           /* name contains "grp"  */
static int pen_is_net(int pen)                                            /* name contains ".net" */
{
    return pen == ORC_ELASTIC_NET || pen == ORC_MCP_NET || pen == ORC_SCAD_NET ||
           pen == ORC_GRP_LASSO_NET || pen == ORC_GRP_MCP_NET || pen == ORC_GRP_SCAD_NET;
}

/* ------------------------------------------------------------------ */
/* ref: src/utils.cpp:537-549 */
int orc_stop_rule(const double *cur, const double *prev, int32_t p, double tol)
{
    for (int i = 0; i < p; i++) {
        double c = fabs(cur[i]), q = fabs(prev[i]);
        if ((c > 1e-13 && q <= 1e-13) || (c <= 1e-13 && q > 1e-13)) return 0;
        if (c > 1e-13 && q > 1e-13 && fabs((cur[i] - prev[i]) / prev[i]) > tol) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ */
/* threshold operators, ref: src/oem_dense.h:76-315 */
static void soft_threshold(double *res, const double *u, int p, double pen, const double *pf, double d)
{
    for (int i = 0; i < p; i++) {
        double tp = pf[i] * pen;
        res[i] = 0.0;
        if (u[i] > tp) res[i] = (u[i] - tp) / d;
        else if (u[i] < -tp) res[i] = (u[i] + tp) / d;
    }
}

static void soft_threshold_mcp(double *res, const double *u, int p, double pen, const double *pf,
                               double d, double gamma)
{
    double gammad = gamma * d, dmg = d - 1.0 / gamma;
    for (int i = 0; i < p; i++) {
        double tp = pf[i] * pen;
        res[i] = 0.0;
        if (fabs(u[i]) > gammad * tp) res[i] = u[i] / d;
        else if (u[i] > tp) res[i] = (u[i] - tp) / dmg;
        else if (u[i] < -tp) res[i] = (u[i] + tp) / dmg;
    }
}

static void soft_threshold_scad(double *res, const double *u, int p, double pen, const double *pf,
                                double d, double gamma)
{
    double gammad = gamma * d, gm1d = (gamma - 1.0) * d;
    for (int i = 0; i < p; i++) {
        double tp = pf[i] * pen;
        res[i] = 0.0;
        if (fabs(u[i]) > gammad * tp) res[i] = u[i] / d;
        else if (fabs(u[i]) > (d + 1.0) * tp) {
            double gp = (gamma - 1.0) * u[i], gq = gamma * tp;
            if (gp > gq) res[i] = (gp - gq) / (gm1d - 1.0);
            else if (gp < -gq) res[i] = (gp + gq) / (gm1d - 1.0);
        }
        else if (u[i] > tp) res[i] = (u[i] - tp) / d;
        else if (u[i] < -tp) res[i] = (u[i] + tp) / d;
    }
}

/* ref: src/oem_dense.h:151-172 */
static double scad_norm(double b, double pen, double d, double gamma)
{
    double r = 0.0, gammad = gamma * d, gm1d = (gamma - 1.0) * d;
    if (fabs(b) > gammad * pen) r = 1;
    else if (fabs(b) > (d + 1.0) * pen) {
        double gp = gamma - 1.0, gq = gamma * pen / b;
        if (gp > gq) r = d * (gp - gq) / (gm1d - 1.0);
        else if (gp < -gq) r = d * (gp + gq) / (gm1d - 1.0);
    }
    else if (b > pen) r = 1.0 - pen / b;
    else if (b < -pen) r = 1.0 + pen / b;
    return r;
}

/* ref: src/oem_dense.h:174-191 */
static double mcp_norm(double b, double pen, double d, double gamma)
{
    double r = 0.0, gammad = gamma * d, dmg = d - 1.0 / gamma;
    if (fabs(b) > gammad * pen) r = 1;
    else if (b > pen) r = d * (1.0 - pen / b) / dmg;
    else if (b < -pen) r = d * (1.0 + pen / b) / dmg;
    return r;
}

typedef struct {
    int ngroups;
    const int32_t *unique_groups;
    int *start;   /* ngroups+1 */
    int *idx;     /* members, by group */
    double *weights; /* ngroups */
} grp_t;

/* kind: 0 lasso, 1 mcp, 2 scad.  ref: src/oem_dense.h:193-315 */
static void block_threshold(double *res, const double *u, int p, double pen, const grp_t *g, double d,
                            int kind, double gamma)
{
    for (int i = 0; i < p; i++) res[i] = 0.0;
    for (int k = 0; k < g->ngroups; k++) {
        double f;
        if (g->unique_groups[k] == 0) f = 1.0;
        else {
            double s = 0.0;
            for (int v = g->start[k]; v < g->start[k + 1]; v++) s += pow(u[g->idx[v]], 2);
            s = sqrt(s);
            double gw = g->weights[k];
            if (kind == 0) {
                /* std::max(0.0, x): returns 0.0 when x is NaN (quirk Q6) */
                double t = 1.0 - pen * gw / s;
                f = (0.0 < t) ? t : 0.0;
            } else if (kind == 1) f = mcp_norm(s, pen * gw, d, gamma);
            else f = scad_norm(s, pen * gw, d, gamma);
        }
        if (f != 0.0)
            for (int v = g->start[k]; v < g->start[k + 1]; v++) res[g->idx[v]] = u[g->idx[v]] * f / d;
    }
}

/* ref: src/oem_dense.h:421-456.  nscan = number of leading entries of groups that are
 * scanned (nvars; oemBig scans only nvars of its nvars+1 entries: quirk Q17) */
static int build_groups(grp_t *g, const orc_opts *o, int nscan)
{
    g->ngroups = o->ngroups;
    g->unique_groups = o->unique_groups;
    g->start = (int *)calloc((size_t)o->ngroups + 1, sizeof(int));
    g->idx = (int *)calloc((size_t)(nscan > 0 ? nscan : 0) + 1, sizeof(int));
    g->weights = (double *)calloc((size_t)o->ngroups + 1, sizeof(double));
    if (!g->start || !g->idx || !g->weights) return fail("oracle: out of memory");
    /* a variable could in principle match several equal unique_groups entries; R passes sort(unique()) */
    int cnt = 0;
    int cap = nscan + 1;
    for (int k = 0; k < o->ngroups; k++) {
        g->start[k] = cnt;
        for (int v = 0; v < nscan && v < o->ngroupvars; v++)
            if (o->groups[v] == o->unique_groups[k]) {
                if (cnt >= cap) { cap *= 2; g->idx = (int *)realloc(g->idx, (size_t)cap * sizeof(int)); }
                g->idx[cnt++] = v;
            }
    }
    g->start[o->ngroups] = cnt;
    for (int k = 0; k < o->ngroups; k++) {
        if (o->n_group_weights < 1) g->weights[k] = sqrt((double)(g->start[k + 1] - g->start[k]));
        else g->weights[k] = (k < o->n_group_weights) ? o->group_weights[k] : 0.0;
    }
    return 0;
}
static void free_groups(grp_t *g) { free(g->start); free(g->idx); free(g->weights); memset(g, 0, sizeof *g); }

/* ref: src/oem_dense.h:527-628 (dispatch only; acceleration is in solve_one) */
static void next_beta(int pen, double *beta, const double *u, int p, double lambda, double d,
                      double alpha, double gamma, double tau, const double *pf, const grp_t *g, double *tmp)
{
    double denom, lam;
    switch (pen) {
    case ORC_LASSO: soft_threshold(beta, u, p, lambda, pf, d); break;
    case ORC_OLS: for (int i = 0; i < p; i++) beta[i] = u[i] / d; break;
    case ORC_ELASTIC_NET:
        denom = d + (1.0 - alpha) * lambda; lam = lambda * alpha;
        soft_threshold(beta, u, p, lam, pf, denom); break;
    case ORC_SCAD: soft_threshold_scad(beta, u, p, lambda, pf, d, gamma); break;
    case ORC_SCAD_NET:
        denom = d + (1.0 - alpha) * lambda; lam = lambda * alpha;
        if (alpha == 0) { lam = 0; denom = d + lambda; }
        soft_threshold_scad(beta, u, p, lam, pf, denom, gamma); break;
    case ORC_MCP: soft_threshold_mcp(beta, u, p, lambda, pf, d, gamma); break;
    case ORC_MCP_NET:
        denom = d + (1.0 - alpha) * lambda; lam = lambda * alpha;
        soft_threshold_mcp(beta, u, p, lam, pf, denom, gamma); break;
    case ORC_GRP_LASSO: block_threshold(beta, u, p, lambda, g, d, 0, gamma); break;
    case ORC_GRP_LASSO_NET:
        denom = d + (1.0 - alpha) * lambda; lam = lambda * alpha;
        block_threshold(beta, u, p, lam, g, denom, 0, gamma); break;
    case ORC_GRP_MCP: block_threshold(beta, u, p, lambda, g, d, 1, gamma); break;
    case ORC_GRP_SCAD: block_threshold(beta, u, p, lambda, g, d, 2, gamma); break;
    case ORC_GRP_MCP_NET:
        denom = d + (1.0 - alpha) * lambda; lam = lambda * alpha;
        block_threshold(beta, u, p, lam, g, denom, 1, gamma); break;
    case ORC_GRP_SCAD_NET:
        denom = d + (1.0 - alpha) * lambda; lam = lambda * alpha;
        block_threshold(beta, u, p, lam, g, denom, 2, gamma); break;
    case ORC_SPARSE_GRP_LASSO: {
        double lam_grp = (1.0 - tau) * lambda, lam_l1 = tau * lambda;
        soft_threshold(tmp, u, p, lam_l1, pf, 1.0);
        block_threshold(beta, tmp, p, lam_grp, g, d, 0, gamma);
        break; }
    default: break;
    }
}

int orc_threshold(int32_t penalty, const double *u, int32_t p, double lambda, double d,
                  double alpha, double gamma, double tau, const double *penalty_factor,
                  const int32_t *groups, const int32_t *unique_groups, int32_t ngroups,
                  const double *group_weights, double *out)
{
    orc_opts o; memset(&o, 0, sizeof o);
    o.groups = groups; o.ngroupvars = groups ? p : 0; o.unique_groups = unique_groups; o.ngroups = ngroups;
    o.group_weights = group_weights; o.n_group_weights = group_weights ? ngroups : 0;
    grp_t g; memset(&g, 0, sizeof g);
    if (pen_is_grp(penalty)) { if (build_groups(&g, &o, p)) return -1; }
    double *tmp = (double *)malloc(sizeof(double) * (size_t)p);
    next_beta(penalty, out, u, p, lambda, d, alpha, gamma, tau, penalty_factor, &g, tmp);
    free(tmp);
    if (pen_is_grp(penalty)) free_groups(&g);
    return 0;
}

/* ------------------------------------------------------------------ */
/* dense symmetric eigenvalues: Householder tridiagonalisation + implicit QL.
 * Stands in for Spectra::SymEigsSolver (ref: src/oem_dense.h:485-498). */
static double pythag(double a, double b)
{
    double aa = fabs(a), ab = fabs(b);
    if (aa > ab) { double r = ab / aa; return aa * sqrt(1.0 + r * r); }
    if (ab == 0.0) return 0.0;
    double r = aa / ab; return ab * sqrt(1.0 + r * r);
}

double orc_eig_max(const double *a_in, int32_t n)
{
    if (n <= 0) return 0.0;
    if (n == 1) return a_in[0];
    double *a = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *dd = (double *)calloc((size_t)n, sizeof(double));
    double *e = (double *)calloc((size_t)n, sizeof(double));
    memcpy(a, a_in, sizeof(double) * (size_t)n * n);
#define A_(i, j) a[(size_t)(i) * n + (j)]
    /* reduction to tridiagonal form, eigenvalues only */
    for (int i = n - 1; i >= 1; i--) {
        int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k = 0; k <= l; k++) scale += fabs(A_(i, k));
            if (scale == 0.0) e[i] = A_(i, l);
            else {
                for (int k = 0; k <= l; k++) { A_(i, k) /= scale; h += A_(i, k) * A_(i, k); }
                double f = A_(i, l);
                double g = (f >= 0.0) ? -sqrt(h) : sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                A_(i, l) = f - g;
                f = 0.0;
                for (int j = 0; j <= l; j++) {
                    g = 0.0;
                    for (int k = 0; k <= j; k++) g += A_(j, k) * A_(i, k);
                    for (int k = j + 1; k <= l; k++) g += A_(k, j) * A_(i, k);
                    e[j] = g / h;
                    f += e[j] * A_(i, j);
                }
                double hh = f / (h + h);
                for (int j = 0; j <= l; j++) {
                    f = A_(i, j);
                    e[j] = g = e[j] - hh * f;
                    for (int k = 0; k <= j; k++) A_(j, k) -= (f * e[k] + g * A_(i, k));
                }
            }
        } else e[i] = A_(i, l);
        dd[i] = h;
    }
    for (int i = 0; i < n; i++) dd[i] = A_(i, i);
#undef A_
    /* implicit QL on (dd, e) */
    for (int i = 1; i < n; i++) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    for (int l = 0; l < n; l++) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; m++) {
                double s = fabs(dd[m]) + fabs(dd[m + 1]);
                if (fabs(e[m]) <= 2.220446049250313e-16 * s) break;
            }
            if (m != l) {
                if (iter++ == 200) break;
                double g = (dd[l + 1] - dd[l]) / (2.0 * e[l]);
                double r = pythag(g, 1.0);
                g = dd[m] - dd[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; i--) {
                    double f = s * e[i], b = c * e[i];
                    e[i + 1] = (r = pythag(f, g));
                    if (r == 0.0) { dd[i + 1] -= p; e[m] = 0.0; break; }
                    s = f / r; c = g / r;
                    g = dd[i + 1] - p;
                    r = (dd[i] - g) * s + 2.0 * c * b;
                    dd[i + 1] = g + (p = s * r);
                    g = c * r - b;
                }
                if (r == 0.0 && i >= l) continue;
                dd[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    double mx = dd[0];
    for (int i = 1; i < n; i++) if (dd[i] > mx) mx = dd[i];
    free(a); free(dd); free(e);
    return mx;
}

/* ------------------------------------------------------------------ */
/* ref: src/DataStd.h:44-58 (non-AVX) */
static double col_mean(const double *v, int64_t n)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += v[i];
    return s / (double)n;
}
static double col_norm(const double *v, int64_t n)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}
static double sd_n(const double *v, int64_t n)
{
    double mean = col_mean(v, n), s = 0.0;
    for (int64_t i = 0; i < n; i++) { double c = v[i] - mean; s += c * c; }
    return sqrt(s) / sqrt((double)n);
}

/* ref: src/DataStd.h:94-267, weights empty, __AVX__ undefined */
void orc_standardize(double *x, int64_t n, int32_t p, double *y, int32_t standardize, int32_t intercept,
                     double *meanx, double *scalex, double *meany, double *scaley)
{
    int flag = (standardize ? 1 : 0) + 2 * (intercept ? 1 : 0);
    double n_invsqrt = 1.0 / sqrt((double)n);
    *meany = 0.0; *scaley = 1.0;
    for (int j = 0; j < p; j++) { meanx[j] = 0.0; scalex[j] = 1.0; }
    switch (flag) {
    case 1:
        *scaley = sd_n(y, n);
        for (int64_t i = 0; i < n; i++) y[i] /= *scaley;
        break;
    case 2: /* falls through into case 3 (quirk Q1) */
    case 3:
        *meany = col_mean(y, n);
        for (int64_t i = 0; i < n; i++) y[i] -= *meany;
        *scaley = col_norm(y, n) * n_invsqrt;
        for (int64_t i = 0; i < n; i++) y[i] /= *scaley;
        break;
    default: break;
    }
    switch (flag) {
    case 1:
        for (int j = 0; j < p; j++) {
            double *c = x + (size_t)j * n;
            scalex[j] = sd_n(c, n);
            if (scalex[j] == 0.0) scalex[j] = 1.0;
            double r = 1.0 / scalex[j];
            for (int64_t i = 0; i < n; i++) c[i] *= r;
        }
        break;
    case 2:
        for (int j = 0; j < p; j++) {
            double *c = x + (size_t)j * n;
            meanx[j] = col_mean(c, n);
            for (int64_t i = 0; i < n; i++) c[i] -= meanx[j];
        }
        break;
    case 3:
        for (int j = 0; j < p; j++) {
            double *c = x + (size_t)j * n;
            meanx[j] = col_mean(c, n);
            for (int64_t i = 0; i < n; i++) c[i] -= meanx[j];
            scalex[j] = col_norm(c, n) * n_invsqrt;
            if (scalex[j] == 0.0) scalex[j] = 1.0;
            for (int64_t i = 0; i < n; i++) c[i] /= scalex[j];
        }
        break;
    default: break;
    }
}

/* ref: src/DataStd.h:269-293 */
static void recover(int flag, int p, double *coef, double *beta0, const double *meanx, const double *scalex,
                    double meany, double scaley)
{
    double s = 0.0;
    switch (flag) {
    case 0: *beta0 = 0; break;
    case 1:
        *beta0 = 0;
        for (int j = 0; j < p; j++) { coef[j] /= scalex[j]; coef[j] *= scaley; }
        break;
    case 2:
        for (int j = 0; j < p; j++) { coef[j] *= scaley; s += coef[j] * meanx[j]; }
        *beta0 = meany - s;
        break;
    case 3:
        for (int j = 0; j < p; j++) { coef[j] /= scalex[j]; coef[j] *= scaley; s += coef[j] * meanx[j]; }
        *beta0 = meany - s;
        break;
    }
}

/* ------------------------------------------------------------------ */
/* lower-triangular rank-k update on rows [r0, r1) accumulated into acc (p x p, lower, col-major).
 * Row-blocked so that a block of each column stays in cache. */
static void syrk_lower_rows(const double *x, int64_t n, int p, int64_t r0, int64_t r1, double *acc)
{
    const int64_t RB = 512;
    for (int64_t b0 = r0; b0 < r1; b0 += RB) {
        int64_t b1 = b0 + RB < r1 ? b0 + RB : r1;
        for (int j = 0; j < p; j++) {
            const double *cj = x + (size_t)j * n;
            int i = j;
            for (; i + 3 < p; i += 4) {
                const double *c0 = x + (size_t)i * n, *c1 = c0 + n, *c2 = c1 + n, *c3 = c2 + n;
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                for (int64_t k = b0; k < b1; k++) {
                    double v = cj[k];
                    s0 += c0[k] * v; s1 += c1[k] * v; s2 += c2[k] * v; s3 += c3[k] * v;
                }
                acc[(size_t)j * p + i] += s0; acc[(size_t)j * p + i + 1] += s1;
                acc[(size_t)j * p + i + 2] += s2; acc[(size_t)j * p + i + 3] += s3;
            }
            for (; i < p; i++) {
                const double *ci = x + (size_t)i * n;
                double s = 0;
                for (int64_t k = b0; k < b1; k++) s += ci[k] * cj[k];
                acc[(size_t)j * p + i] += s;
            }
        }
    }
}

/* XtX (ref: src/oem_dense.h:318-361; slices: src/oem_big.h:319-361).  Full symmetric, NOT divided. */
static void xtx_full(const double *x, int64_t n, int p, int nblocks, int parallel, double *xx)
{
    memset(xx, 0, sizeof(double) * (size_t)p * p);
    if (nblocks <= 1) {
        syrk_lower_rows(x, n, p, 0, n, xx);
    } else {
        int64_t first = (int64_t)floor((double)n / (double)nblocks);
        (void)parallel;
#ifdef _OPENMP
#pragma omp parallel if (parallel) num_threads(parallel ? nblocks : 1)
#endif
        {
            double *priv = (double *)calloc((size_t)p * p, sizeof(double));
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
            for (int ff = 0; ff < nblocks; ff++) {
                int64_t r0 = (int64_t)ff * first;
                int64_t r1 = (ff + 1 == nblocks) ? n : r0 + first;
                syrk_lower_rows(x, n, p, r0, r1, priv);
            }
#ifdef _OPENMP
#pragma omp critical
#endif
            { for (size_t k = 0; k < (size_t)p * p; k++) xx[k] += priv[k]; }
            free(priv);
        }
    }
    for (int j = 0; j < p; j++)
        for (int i = j + 1; i < p; i++) xx[(size_t)i * p + j] = xx[(size_t)j * p + i];
}

void orc_gram(const double *x, int64_t n, int32_t p, const double *y, int32_t ncores, double *xx, double *xy)
{
    /* XY = X'Y / n  (ref: src/oem_dense.h:704-707) */
    for (int j = 0; j < p; j++) {
        const double *c = x + (size_t)j * n;
        double s = 0.0;
        for (int64_t i = 0; i < n; i++) s += c[i] * y[i];
        xy[j] = s / (double)n;
    }
    xtx_full(x, n, p, ncores, 1, xx);
    for (size_t k = 0; k < (size_t)p * p; k++) xx[k] /= (double)n;   /* ref: :483 */
}

/* ------------------------------------------------------------------ */
/* penalty x lambda loops on a given Gram (ref: src/oem_dense.cpp:206-297, oem_base.h:90-110) */
typedef struct {
    int p;
    const double *A;    /* p x p col-major (symmetric) */
    const double *XY;
    double d;
    double *u, *beta, *beta_prev, *tmp, *beta_last;
    /* p >= n branch (ref: src/oem_dense.h:513-521): A is not formed, next_u goes through X twice */
    const double *X, *Y;   /* standardised X (n x p col-major), Y (n); NULL on the n > p branch */
    int64_t n;
    double *resid;         /* n */
} core_t;

/* ORC_THREADS=k: the two products of next_u on k OpenMP threads.  Columns (p >= n) and row slices (n > p) are dealt to the threads whole
 * and every one of them is summed in the order of the single-thread loop, so the results are bit-identical -- it only shortens the GPU
 * suite, whose wall clock is mostly this loop. */
static int orc_threads(void)
{
    static int t = 0;
    if (t == 0) { const char *e = getenv("ORC_THREADS"); t = e ? atoi(e) : 1; if (t < 1) t = 1; if (t > 64) t = 64; }
    return t;
}

static int solve_one(core_t *s, int pen, double lambda, const orc_opts *o, const double *pf, const grp_t *g,
                     double *ak)
{
    int p = s->p, i;
    const int nth = orc_threads();
    for (i = 0; i < o->maxit; ++i) {
        memcpy(s->beta_prev, s->beta, sizeof(double) * (size_t)p);
        if (s->X) {
            /* next_u: u = X'(Y - X beta_prev) / n + d beta_prev  (ref: src/oem_dense.h:520) */
            memcpy(s->resid, s->Y, sizeof(double) * (size_t)s->n);
            for (int c = 0; c < p; c++) {
                double b = s->beta_prev[c];
                if (b == 0.0) continue;
                const double *col = s->X + (size_t)c * s->n;
                for (int64_t k = 0; k < s->n; k++) s->resid[k] -= col[k] * b;
            }
            #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && (double)p * (double)s->n > 2e5)
            for (int c = 0; c < p; c++) {
                const double *col = s->X + (size_t)c * s->n;
                double t = 0.0;
                for (int64_t k = 0; k < s->n; k++) t += col[k] * s->resid[k];
                s->u[c] = t / (double)s->n + s->d * s->beta_prev[c];
            }
        } else {
        /* next_u: u = A * beta_prev + XY  (ref: src/oem_dense.h:512) */
        #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && (double)p * (double)p > 2e5)
        for (int blk = 0; blk < nth; blk++) {
            const int r0 = (int)((long long)p * blk / nth), r1 = (int)((long long)p * (blk + 1) / nth);
            for (int r = r0; r < r1; r++) s->u[r] = 0.0;
            for (int c = 0; c < p; c++) {
                double b = s->beta_prev[c];
                if (b == 0.0) continue;               /* adds exact zeros otherwise */
                const double *col = s->A + (size_t)c * p;
                for (int r = r0; r < r1; r++) s->u[r] += col[r] * b;
            }
            for (int r = r0; r < r1; r++) s->u[r] += s->XY[r];
        }
        }
        if (o->accelerate) memcpy(s->beta_last, s->beta, sizeof(double) * (size_t)p);
        next_beta(pen, s->beta, s->u, p, lambda, s->d, o->alpha, o->gamma, o->tau, pf, g, s->tmp);
        if (o->accelerate) {
            /* ref: src/oem_dense.h:633-651 */
            double ak_prev = *ak;
            *ak = 0.5 * (1 + sqrt(1.0 + 4.0 * pow(*ak, 2)));
            double ratio = (ak_prev - 1.0) / *ak, adaptive = 0.0;
            for (int r = 0; r < p; r++) {
                double upd = s->beta[r], diff = upd - s->beta_last[r];
                s->beta[r] += ratio * diff;
                adaptive += (s->beta[r] - upd) * diff;
            }
            if (adaptive > 0) *ak = 1;
        }
        if (orc_stop_rule(s->beta, s->beta_prev, p, o->tol)) break;
    }
    return i + 1;     /* maxit + 1 when the loop runs out (quirk Q2) */
}

static int core_alloc(core_t *s, int p)
{
    s->p = p;
    s->u = (double *)calloc((size_t)p * 5, sizeof(double));
    if (!s->u) return fail("oracle: out of memory");
    s->beta = s->u + p; s->beta_prev = s->beta + p; s->tmp = s->beta_prev + p; s->beta_last = s->tmp + p;
    return 0;
}

static int nl_of(const orc_opts *o) { return (o->lambda_user && o->nlambda_user > 0) ? o->nlambda_user : o->nlambda; }

/* lambda grid (ref: src/oem_dense.cpp:180-227).  lam: npen * nl */
static void lambda_grid(const orc_opts *o, double lmax, double *lam)
{
    int nl = nl_of(o);
    if (o->lambda_user && o->nlambda_user > 0) {
        memcpy(lam, o->lambda_user, sizeof(double) * (size_t)o->npen * nl);
        return;
    }
    double lo = log(lmax), hi = log(o->lambda_min_ratio * lmax);
    /* Eigen 3.3 setLinSpaced(size, low, high) (linspaced_op_impl<double>): step = (high-low)/(size-1);
     * |high| < |low| ("flip"): i==0 ? low : high - (size-1-i)*step; else i==size-1 ? high : low + i*step;
     * size==1 -> high */
    int size1 = nl == 1 ? 1 : nl - 1;
    double step = nl == 1 ? 0.0 : (hi - lo) / (double)(nl - 1);
    int flip = fabs(hi) < fabs(lo);
    for (int pp = 0; pp < o->npen; pp++)
        for (int k = 0; k < nl; k++) {
            double v;
            if (nl == 1) v = hi;
            else if (flip) v = (k == 0) ? lo : hi - (double)(size1 - k) * step;
            else v = (k == size1) ? hi : lo + (double)k * step;
            v = exp(v);
            if (pen_is_net(o->penalty[pp])) v = v / o->alpha;
            lam[(size_t)pp * nl + k] = v;
        }
}

/* xx != NULL: the n > p branch on the Gram;  xx == NULL: the p >= n branch on the standardised (X, Y) */
static int path_run(const double *xx, const double *xy, int32_t p, double d, const orc_opts *o,
                    const double *lambda_scaled, int32_t nl, const double *scale_factor_inv,
                    double *beta_std, int32_t *niter, const double *X, const double *Y, int64_t n)
{
    core_t s; memset(&s, 0, sizeof s);
    if (core_alloc(&s, p)) return -1;
    double *A = NULL;
    if (xx) {
        A = (double *)malloc(sizeof(double) * (size_t)p * p);
        for (size_t k = 0; k < (size_t)p * p; k++) A[k] = -xx[k];
        for (int j = 0; j < p; j++) A[(size_t)j * p + j] += d;
    } else {
        s.X = X; s.Y = Y; s.n = n;
        s.resid = (double *)malloc(sizeof(double) * (size_t)n);
    }
    s.A = A; s.XY = xy; s.d = d;
    grp_t g; memset(&g, 0, sizeof g);
    int have_g = 0;
    for (int pp = 0; pp < o->npen; pp++) {
        int pen = o->penalty[pp];
        int nlam = (pen == ORC_OLS) ? 1 : nl;
        double ak = 1.0;
        for (int i = 0; i < nlam; i++) {
            double il = lambda_scaled[(size_t)pp * nl + i];
            if (i == 0) {
                memset(s.beta, 0, sizeof(double) * (size_t)p);
                if (!have_g && pen_is_grp(pen)) { if (build_groups(&g, o, p)) return -1; have_g = 1; }
                ak = 1.0;
            }
            niter[(size_t)pp * nl + i] = solve_one(&s, pen, il, o, o->penalty_factor, &g, &ak);
            if (scale_factor_inv) for (int j = 0; j < p; j++) s.beta[j] *= scale_factor_inv[j];  /* Q5 */
            memcpy(beta_std + ((size_t)pp * nl + i) * p, s.beta, sizeof(double) * (size_t)p);
        }
    }
    if (have_g) free_groups(&g);
    free(A); free(s.u); free(s.resid);
    return 0;
}

int orc_path(const double *xx, const double *xy, int32_t p, double d, const orc_opts *o,
             const double *lambda_scaled, int32_t nl, const double *scale_factor_inv,
             double *beta_std, int32_t *niter)
{
    return path_run(xx, xy, p, d, o, lambda_scaled, nl, scale_factor_inv, beta_std, niter, NULL, NULL, 0);
}

/* XXt / n: the n x n row Gram of the p >= n branch (ref: src/oem_dense.h:363-366, 483) */
static void xxt_over_n(const double *x, int64_t n, int32_t p, double *g)
{
    for (int64_t a = 0; a < n; a++)
        for (int64_t b = 0; b <= a; b++) {
            double t = 0.0;
            for (int j = 0; j < p; j++) t += x[(size_t)j * n + a] * x[(size_t)j * n + b];
            g[(size_t)b * n + a] = g[(size_t)a * n + b] = t / (double)n;
        }
}

/* ------------------------------------------------------------------ */
int orc_fit_dense(const double *x_in, int64_t n, int32_t p, const double *y_in,
                  int32_t standardize, int32_t intercept, const orc_opts *o,
                  double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out)
{
    const int wide = n <= p;          /* the XXt branch (ref: src/oem_dense.h:476-482) */
    if (wide && n > 8192) return fail("oracle: the p >= n branch is restated for n <= 8192 only (dense n x n eigen-solve)");
    int flag = (standardize ? 1 : 0) + 2 * (intercept ? 1 : 0);
    int nl = nl_of(o);
    /* copy (ref: src/oem_dense.cpp:61-67) */
    double *X = (double *)malloc(sizeof(double) * (size_t)n * p);
    double *Y = (double *)malloc(sizeof(double) * (size_t)n);
    double *meanx = (double *)malloc(sizeof(double) * (size_t)p * 2), *scalex = meanx + p;
    double *XX = (double *)malloc(sizeof(double) * (size_t)p * p);
    double *XY = (double *)malloc(sizeof(double) * (size_t)p);
    double *lam = (double *)malloc(sizeof(double) * (size_t)o->npen * nl);
    double *bstd = (double *)malloc(sizeof(double) * (size_t)o->npen * nl * p);
    if (!X || !Y || !meanx || !XX || !XY || !lam || !bstd) return fail("oracle: out of memory");
    memcpy(X, x_in, sizeof(double) * (size_t)n * p);
    memcpy(Y, y_in, sizeof(double) * (size_t)n);
    double meany, scaley;
    orc_standardize(X, n, p, Y, standardize, intercept, meanx, scalex, &meany, &scaley);
    int ncores = o->ncores < 1 ? 1 : o->ncores;
    double d;
    if (!wide) {
        orc_gram(X, n, p, Y, ncores, XX, XY);
        d = (o->d_override > 0) ? o->d_override : orc_eig_max(XX, p) * 1.005;
    } else {
        /* XY = X'Y / n (ref: :704-707);  d from XXt / n (ref: :480-498) */
        for (int j = 0; j < p; j++) {
            const double *c = X + (size_t)j * n;
            double t = 0.0;
            for (int64_t i = 0; i < n; i++) t += c[i] * Y[i];
            XY[j] = t / (double)n;
        }
        if (o->d_override > 0) d = o->d_override;      /* (a test that hands d over does not pay the n^2 p row Gram and its eigen-solve) */
        else {
            double *G = (double *)malloc(sizeof(double) * (size_t)n * n);
            if (!G) return fail("oracle: out of memory");
            xxt_over_n(X, n, p, G);
            d = orc_eig_max(G, (int32_t)n) * 1.005;
            free(G);
        }
    }
    *d_out = d;
    double lmax = 0.0;
    for (int j = 0; j < p; j++) if (fabs(XY[j]) > lmax) lmax = fabs(XY[j]);
    lmax *= scaley;
    lambda_grid(o, lmax, lam);
    for (size_t k = 0; k < (size_t)o->npen * nl; k++) { lambda_out[k] = lam[k]; loss[k] = 1e99; niter[k] = 0; }
    /* ilambda = lambda / scaleY (ref: src/oem_dense.cpp:241) */
    double *lams = (double *)malloc(sizeof(double) * (size_t)o->npen * nl);
    for (size_t k = 0; k < (size_t)o->npen * nl; k++) lams[k] = lam[k] / scaley;
    int rc = path_run(wide ? NULL : XX, XY, p, d, o, lams, nl, NULL, bstd, niter, X, Y, n);
    if (rc == 0) {
        for (int pp = 0; pp < o->npen; pp++) {
            int nlam = (o->penalty[pp] == ORC_OLS) ? 1 : nl;
            for (int i = 0; i < nl; i++) {
                double *out = beta + ((size_t)pp * nl + i) * (p + 1);
                if (i >= nlam) { for (int j = 0; j <= p; j++) out[j] = 0.0; continue; }
                const double *b = bstd + ((size_t)pp * nl + i) * p;
                if (o->compute_loss) {
                    /* loss on the standardised data with the un-recovered beta (quirk Q4) */
                    double l = 0.0;
                    double *r = (double *)malloc(sizeof(double) * (size_t)n);
                    memcpy(r, Y, sizeof(double) * (size_t)n);
                    for (int j = 0; j < p; j++) {
                        if (b[j] == 0.0) continue;
                        const double *c = X + (size_t)j * n;
                        for (int64_t k = 0; k < n; k++) r[k] -= c[k] * b[j];
                    }
                    for (int64_t k = 0; k < n; k++) l += r[k] * r[k];
                    free(r);
                    loss[(size_t)pp * nl + i] = l;
                }
                memcpy(out + 1, b, sizeof(double) * (size_t)p);
                recover(flag, p, out + 1, out, meanx, scalex, meany, scaley);
            }
        }
    }
    free(X); free(Y); free(meanx); free(XX); free(XY); free(lam); free(bstd); free(lams);
    return rc;
}

/* ------------------------------------------------------------------ */
/* oemDense with observation weights -- what `.Call("oem_fit_dense", ..., weights_, ...)` computes (the R front end stops with
 * "weights not implemented yet", R/oem.R:244, so nothing the package ships reaches it: no reference-held number exists for this
 * branch; PARITY UNPINNED here, the restatement is held against numpy in tests/test_oracle_independent.py).  Restated as it is,
 * inconsistencies included:
 *   DataStd::standardize(X, Y, wts), ref src/DataStd.h:94-202: Y by sqrt(w)-weighted statistics in every flag (flag 2 falls through
 *     into flag 3); X by sd_n(x sqrt w) (flag 1), mean(x sqrt w) (flag 2) and the UNWEIGHTED mean / norm (flag 3);
 *   XY = X'(Y w)/n (ref src/oem_dense.h:699-707); nobs > nvars: XX = X' diag(w) X / n (:368-414, 466-483);
 *   nobs <= nvars: d from (sqrt(w) X)(sqrt(w) X)'/n (:410-414) but next_u = X'((Y - X beta) w^2)/n + d beta (:513-517: w SQUARED);
 *   get_loss = sum w (Y - X beta)^2 on the standardised data (:759-770). */
int orc_fit_dense_w(const double *x_in, int64_t n, int32_t p, const double *y_in, const double *w,
                    int32_t standardize, int32_t intercept, const orc_opts *o,
                    double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out)
{
    const int wide = n <= p;
    if (wide && n > 8192) return fail("oracle: the p >= n branch is restated for n <= 8192 only (dense n x n eigen-solve)");
    int flag = (standardize ? 1 : 0) + 2 * (intercept ? 1 : 0);
    int nl = nl_of(o);
    double *X = (double *)malloc(sizeof(double) * (size_t)n * p), *Z = (double *)malloc(sizeof(double) * (size_t)n * p);
    double *Y = (double *)malloc(sizeof(double) * (size_t)n * 3), *Yw = Y + n, *tmp = Yw + n;
    double *meanx = (double *)malloc(sizeof(double) * (size_t)p * 2), *scalex = meanx + p;
    double *XX = (double *)malloc(sizeof(double) * (size_t)(wide ? 1 : p) * (wide ? 1 : p));
    double *XY = (double *)malloc(sizeof(double) * (size_t)p);
    double *lam = (double *)malloc(sizeof(double) * (size_t)o->npen * nl), *lams = (double *)malloc(sizeof(double) * (size_t)o->npen * nl);
    double *bstd = (double *)malloc(sizeof(double) * (size_t)o->npen * nl * p);
    if (!X || !Z || !Y || !meanx || !XX || !XY || !lam || !lams || !bstd) return fail("oracle: out of memory");
    memcpy(X, x_in, sizeof(double) * (size_t)n * p);
    memcpy(Y, y_in, sizeof(double) * (size_t)n);
    const double n_invsqrt = 1.0 / sqrt((double)n);
    double meany = 0.0, scaley = 1.0;
    for (int j = 0; j < p; j++) { meanx[j] = 0.0; scalex[j] = 1.0; }
    /* Y (ref src/DataStd.h:100-137) */
    if (flag == 1) {
        for (int64_t i = 0; i < n; i++) tmp[i] = Y[i] * sqrt(w[i]);
        scaley = sd_n(tmp, n);
        for (int64_t i = 0; i < n; i++) Y[i] /= scaley;
    } else if (flag >= 2) {
        for (int64_t i = 0; i < n; i++) tmp[i] = Y[i] * sqrt(w[i]);
        meany = col_mean(tmp, n);
        for (int64_t i = 0; i < n; i++) Y[i] -= meany;
        for (int64_t i = 0; i < n; i++) tmp[i] = Y[i] * sqrt(w[i]);
        scaley = col_norm(tmp, n) * n_invsqrt;
        for (int64_t i = 0; i < n; i++) Y[i] /= scaley;
    }
    /* X (ref src/DataStd.h:140-202) */
    for (int j = 0; j < p && flag; j++) {
        double *c = X + (size_t)j * n;
        if (flag == 1) {
            for (int64_t i = 0; i < n; i++) tmp[i] = c[i] * sqrt(w[i]);
            scalex[j] = sd_n(tmp, n);
            if (scalex[j] == 0.0) scalex[j] = 1.0;
            double r = 1.0 / scalex[j];
            for (int64_t i = 0; i < n; i++) c[i] *= r;
        } else if (flag == 2) {
            for (int64_t i = 0; i < n; i++) tmp[i] = c[i] * sqrt(w[i]);
            meanx[j] = col_mean(tmp, n);
            for (int64_t i = 0; i < n; i++) c[i] -= meanx[j];
        } else {
            meanx[j] = col_mean(c, n);
            for (int64_t i = 0; i < n; i++) c[i] -= meanx[j];
            scalex[j] = col_norm(c, n) * n_invsqrt;
            if (scalex[j] == 0.0) scalex[j] = 1.0;          /* (the __AVX__ build's guard, ref :172-175) */
            for (int64_t i = 0; i < n; i++) c[i] /= scalex[j];
        }
    }
    /* XY = X'(Y w) / n */
    for (int64_t i = 0; i < n; i++) Yw[i] = Y[i] * w[i];
    for (int j = 0; j < p; j++) {
        const double *c = X + (size_t)j * n;
        double t = 0.0;
        for (int64_t i = 0; i < n; i++) t += c[i] * Yw[i];
        XY[j] = t / (double)n;
    }
    /* Z = diag(sqrt w) X: XtWX = Z'Z, XWXt = Z Z' */
    for (int j = 0; j < p; j++)
        for (int64_t i = 0; i < n; i++) Z[(size_t)j * n + i] = X[(size_t)j * n + i] * sqrt(w[i]);
    double d;
    if (!wide) {
        xtx_full(Z, n, p, o->ncores < 1 ? 1 : o->ncores, 1, XX);
        for (size_t k = 0; k < (size_t)p * p; k++) XX[k] /= (double)n;
        d = (o->d_override > 0) ? o->d_override : orc_eig_max(XX, p) * 1.005;
    } else if (o->d_override > 0) d = o->d_override;
    else {
        double *G = (double *)malloc(sizeof(double) * (size_t)n * n);
        if (!G) return fail("oracle: out of memory");
        xxt_over_n(Z, n, p, G);
        d = orc_eig_max(G, (int32_t)n) * 1.005;
        free(G);
    }
    *d_out = d;
    double lmax = 0.0;
    for (int j = 0; j < p; j++) if (fabs(XY[j]) > lmax) lmax = fabs(XY[j]);
    lmax *= scaley;
    lambda_grid(o, lmax, lam);
    for (size_t k = 0; k < (size_t)o->npen * nl; k++) { lambda_out[k] = lam[k]; loss[k] = 1e99; niter[k] = 0; lams[k] = lam[k] / scaley; }
    int rc;
    if (!wide) rc = path_run(XX, XY, p, d, o, lams, nl, NULL, bstd, niter, NULL, NULL, 0);
    else {
        /* next_u = X'((Y - X beta) w^2)/n + d beta = (wX)'(wY - (wX) beta)/n + d beta: the unweighted form on the w-scaled data */
        for (int j = 0; j < p; j++)
            for (int64_t i = 0; i < n; i++) Z[(size_t)j * n + i] = X[(size_t)j * n + i] * w[i];
        rc = path_run(NULL, XY, p, d, o, lams, nl, NULL, bstd, niter, Z, Yw, n);
    }
    if (rc == 0) {
        for (int pp = 0; pp < o->npen; pp++) {
            int nlam = (o->penalty[pp] == ORC_OLS) ? 1 : nl;
            for (int i = 0; i < nl; i++) {
                double *out = beta + ((size_t)pp * nl + i) * (p + 1);
                if (i >= nlam) { for (int j = 0; j <= p; j++) out[j] = 0.0; continue; }
                const double *b = bstd + ((size_t)pp * nl + i) * p;
                if (o->compute_loss) {
                    double l = 0.0;
                    memcpy(tmp, Y, sizeof(double) * (size_t)n);
                    for (int j = 0; j < p; j++) {
                        if (b[j] == 0.0) continue;
                        const double *c = X + (size_t)j * n;
                        for (int64_t k = 0; k < n; k++) tmp[k] -= c[k] * b[j];
                    }
                    for (int64_t k = 0; k < n; k++) l += tmp[k] * tmp[k] * w[k];
                    loss[(size_t)pp * nl + i] = l;
                }
                memcpy(out + 1, b, sizeof(double) * (size_t)p);
                recover(flag, p, out + 1, out, meanx, scalex, meany, scaley);
            }
        }
    }
    free(X); free(Z); free(Y); free(meanx); free(XX); free(XY); free(lam); free(lams); free(bstd);
    return rc;
}

/* ------------------------------------------------------------------ */
int orc_fit_xtx(const double *xtx, const double *xty, int32_t p, const double *scale_factor,
                const orc_opts *o,
                double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out)
{
    int nl = nl_of(o);
    double *XX = (double *)malloc(sizeof(double) * (size_t)p * p);
    double *XY = (double *)malloc(sizeof(double) * (size_t)p * 2), *sinv = XY + p;
    double *lam = (double *)malloc(sizeof(double) * (size_t)o->npen * nl);
    if (!XX || !XY || !lam) return fail("oracle: out of memory");
    if (scale_factor) {
        /* ref: src/oem_xtx.h:527-531, :349-352 */
        for (int j = 0; j < p; j++) { sinv[j] = 1 / scale_factor[j]; XY[j] = xty[j] * sinv[j]; }
        for (int j = 0; j < p; j++)
            for (int i = 0; i < p; i++) XX[(size_t)j * p + i] = sinv[i] * xtx[(size_t)j * p + i] * sinv[j];
    } else {
        memcpy(XY, xty, sizeof(double) * (size_t)p);
        memcpy(XX, xtx, sizeof(double) * (size_t)p * p);
    }
    double d = (o->d_override > 0) ? o->d_override : orc_eig_max(XX, p) * 1.005;
    *d_out = d;
    double lmax = 0.0;
    for (int j = 0; j < p; j++) if (fabs(XY[j]) > lmax) lmax = fabs(XY[j]);
    lambda_grid(o, lmax, lam);
    for (size_t k = 0; k < (size_t)o->npen * nl; k++) { lambda_out[k] = lam[k]; loss[k] = 1e99; niter[k] = 0; }
    memset(beta, 0, sizeof(double) * (size_t)o->npen * nl * p);
    int rc = orc_path(XX, XY, p, d, o, lam, nl, scale_factor ? sinv : NULL, beta, niter);
    free(XX); free(XY); free(lam);
    return rc;
}

/* ------------------------------------------------------------------ */
/* big.oem and oem() on a sparse x with nobs <= nvars + intercept (ref: src/oem_big.h:537-541, 568-584, 743-764, 880-897; src/oem_sparse.h:
 * 607-612, 638-647 -- the same code).  With an intercept next_u multiplies the n x nvars map X by a vector of nvars + 1 entries:
 * nothing well-formed to restate.  Without one the branch reads
 *     XY  = X'y colsq_inv / n            (standardize: colsq = sum x^2 / (n - 1), no centring; used for lambda_zero ONLY)
 *     d   = 1.005 lambda_max(X X' / n)   from the data as they are -- colsq_inv never reaches XXt()
 *     u   = X'(Y - X beta) / n + d beta  on the data as they are
 *     get_beta() = beta colsq_inv
 * i.e. with standardize the iteration is oemDense's flag-0 iteration, only the lambda grid and the returned coefficients carry the
 * column scales (what the reference does, restated as it is). */
static int big_wide(const double *x, int64_t n, int32_t p, const double *y, int32_t standardize, int32_t intercept, const orc_opts *o,
                    double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out)
{
    if (intercept) return fail("oracle: big.oem / sparse x with p >= n and an intercept: the reference's next_u is ill-formed (src/oem_big.h:568-584)");
    if (n > 8192) return fail("oracle: the p >= n branch is restated for n <= 8192 only (dense n x n eigen-solve)");
    int nl = nl_of(o);
    double *XY = (double *)malloc(sizeof(double) * (size_t)p), *colsq_inv = (double *)malloc(sizeof(double) * (size_t)p);
    double *lam = (double *)malloc(sizeof(double) * (size_t)o->npen * nl);
    double *bstd = (double *)malloc(sizeof(double) * (size_t)o->npen * nl * p);
    if (!XY || !colsq_inv || !lam || !bstd) return fail("oracle: out of memory");
    for (int j = 0; j < p; j++) {
        const double *c = x + (size_t)j * n;
        double s2 = 0.0, t = 0.0;
        for (int64_t i = 0; i < n; i++) { s2 += c[i] * c[i]; t += c[i] * y[i]; }
        double cs = s2 / ((double)n - 1.0);
        if (cs == 0.0) cs = 1.0;
        colsq_inv[j] = standardize ? 1.0 / sqrt(cs) : 1.0;
        XY[j] = t;
        if (standardize) XY[j] *= colsq_inv[j];
        XY[j] /= (double)n;
    }
    double d;
    if (o->d_override > 0) d = o->d_override;
    else {
        double *G = (double *)malloc(sizeof(double) * (size_t)n * n);
        if (!G) return fail("oracle: out of memory");
        xxt_over_n(x, n, p, G);
        d = orc_eig_max(G, (int32_t)n) * 1.005;
        free(G);
    }
    *d_out = d;
    double lmax = 0.0;
    for (int j = 0; j < p; j++) if (fabs(XY[j]) > lmax) lmax = fabs(XY[j]);
    lambda_grid(o, lmax, lam);
    for (size_t k = 0; k < (size_t)o->npen * nl; k++) { lambda_out[k] = lam[k]; loss[k] = 1e99; niter[k] = 0; }
    orc_opts o2 = *o; o2.accelerate = 0;                 /* oemBig / oemSparse have no acceleration */
    int rc = path_run(NULL, XY, p, d, &o2, lam, nl, NULL, bstd, niter, x, y, n);
    if (rc == 0)
        for (int pp = 0; pp < o->npen; pp++) {
            int nlam = (o->penalty[pp] == ORC_OLS) ? 1 : nl;
            for (int i = 0; i < nl; i++) {
                double *out = beta + ((size_t)pp * nl + i) * (p + 1);
                for (int j = 0; j <= p; j++) out[j] = 0.0;
                if (i >= nlam) continue;
                const double *b = bstd + ((size_t)pp * nl + i) * p;
                for (int j = 0; j < p; j++) out[j + 1] = b[j] * colsq_inv[j];
            }
        }
    free(XY); free(colsq_inv); free(lam); free(bstd);
    return rc;
}

int orc_fit_big(const double *x, int64_t n, int32_t p, const double *y,
                int32_t standardize, int32_t intercept, const orc_opts *o,
                double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out)
{
    int q = p + (intercept ? 1 : 0);
    if (o->compute_loss) return fail("oracle: big.oem get_loss is ill-formed in the reference (src/oem_big.h:899-921)");
    if (n <= q) return big_wide(x, n, p, y, standardize, intercept, o, beta, lambda_out, niter, loss, d_out);
    int nl = nl_of(o);
    double *XtXm = (double *)malloc(sizeof(double) * (size_t)p * p);
    double *XX = (double *)calloc((size_t)q * q, sizeof(double));
    double *XY = (double *)calloc((size_t)q, sizeof(double));
    double *colsums = (double *)calloc((size_t)p * 3, sizeof(double)), *colsq = colsums + p, *colsq_inv = colsq + p;
    double *pf = (double *)calloc((size_t)q, sizeof(double));
    double *lam = (double *)malloc(sizeof(double) * (size_t)o->npen * nl);
    double *bstd = (double *)malloc(sizeof(double) * (size_t)o->npen * nl * q);
    if (!XtXm || !XX || !XY || !colsums || !pf || !lam || !bstd) return fail("oracle: out of memory");
    /* ref: src/oem_big.cpp:105-113 */
    for (int j = 0; j < p; j++) pf[j + (intercept ? 1 : 0)] = o->penalty_factor[j];
    /* ref: src/oem_big.h:731-842 */
    double xgigs = 8.0 * (double)n * (double)p / pow(10.0, 9);
    int nslices = (int)ceil(xgigs / (o->gigs > 0 ? o->gigs : 4.0));
    for (int j = 0; j < p; j++) colsq_inv[j] = 1.0;
    if (standardize) {
        for (int j = 0; j < p; j++) {
            const double *c = x + (size_t)j * n; double s = 0.0;
            for (int64_t i = 0; i < n; i++) s += c[i] * c[i];
            colsq[j] = s / ((double)n - 1.0);
            if (colsq[j] == 0.0) colsq[j] = 1.0;
            colsq_inv[j] = 1.0 / sqrt(colsq[j]);
        }
    }
    int off = intercept ? 1 : 0;
    for (int j = 0; j < p; j++) {
        const double *c = x + (size_t)j * n; double s = 0.0;
        for (int64_t i = 0; i < n; i++) s += c[i] * y[i];
        XY[j + off] = s;
    }
    if (intercept) { double s = 0.0; for (int64_t i = 0; i < n; i++) s += y[i]; XY[0] = s; }
    if (standardize) for (int j = 0; j < p; j++) XY[j + off] *= colsq_inv[j];
    for (int j = 0; j < q; j++) XY[j] /= (double)n;
    if (intercept)
        for (int j = 0; j < p; j++) {
            const double *c = x + (size_t)j * n; double s = 0.0;
            for (int64_t i = 0; i < n; i++) s += c[i];
            colsums[j] = s;
        }
    /* ref: src/oem_big.h:469-566 */
    xtx_full(x, n, p, nslices, 0, XtXm);
    if (standardize) {
        if (intercept) for (int j = 0; j < p; j++) colsums[j] *= colsq_inv[j];
        for (int j = 0; j < p; j++)
            for (int i = 0; i < p; i++) XtXm[(size_t)j * p + i] = colsq_inv[i] * XtXm[(size_t)j * p + i] * colsq_inv[j];
    }
    for (int j = 0; j < p; j++)
        for (int i = 0; i < p; i++) XX[(size_t)(j + off) * q + (i + off)] = XtXm[(size_t)j * p + i];
    if (intercept) {
        for (int j = 0; j < p; j++) { XX[(size_t)(j + 1) * q + 0] = colsums[j]; XX[(size_t)0 * q + (j + 1)] = colsums[j]; }
        XX[0] = (double)n;
    }
    for (size_t k = 0; k < (size_t)q * q; k++) XX[k] /= (double)n;
    double d = (o->d_override > 0) ? o->d_override : orc_eig_max(XX, q) * 1.005;
    *d_out = d;
    double lmax = 0.0;   /* includes the intercept slot (quirk Q10) */
    for (int j = 0; j < q; j++) if (fabs(XY[j]) > lmax) lmax = fabs(XY[j]);
    lambda_grid(o, lmax, lam);
    for (size_t k = 0; k < (size_t)o->npen * nl; k++) { lambda_out[k] = lam[k]; loss[k] = 1e99; niter[k] = 0; }
    /* groups are scanned over nvars entries only (quirk Q17, src/oem_big.h:445) */
    orc_opts oo = *o; oo.penalty_factor = pf;
    /* path on dimension q; build_groups inside orc_path scans q entries, so clip here */
    int rc;
    {
        core_t s; memset(&s, 0, sizeof s);
        if (core_alloc(&s, q)) return -1;
        double *A = (double *)malloc(sizeof(double) * (size_t)q * q);
        for (size_t k = 0; k < (size_t)q * q; k++) A[k] = -XX[k];
        for (int j = 0; j < q; j++) A[(size_t)j * q + j] += d;
        s.A = A; s.XY = XY; s.d = d;
        grp_t g; memset(&g, 0, sizeof g); int have_g = 0;
        rc = 0;
        for (int pp = 0; pp < o->npen && rc == 0; pp++) {
            int pen = o->penalty[pp];
            int nlam = (pen == ORC_OLS) ? 1 : nl;
            double ak = 1.0;
            for (int i = 0; i < nlam; i++) {
                if (i == 0) {
                    memset(s.beta, 0, sizeof(double) * (size_t)q);
                    if (!have_g && pen_is_grp(pen)) { if (build_groups(&g, &oo, p)) { rc = -1; break; } have_g = 1; }
                }
                orc_opts o2 = oo; o2.accelerate = 0;   /* oemBig has no acceleration */
                niter[(size_t)pp * nl + i] = solve_one(&s, pen, lam[(size_t)pp * nl + i], &o2, pf, &g, &ak);
                memcpy(bstd + ((size_t)pp * nl + i) * q, s.beta, sizeof(double) * (size_t)q);
            }
        }
        if (have_g) free_groups(&g);
        free(A); free(s.u);
    }
    if (rc == 0)
        for (int pp = 0; pp < o->npen; pp++) {
            int nlam = (o->penalty[pp] == ORC_OLS) ? 1 : nl;
            for (int i = 0; i < nl; i++) {
                double *out = beta + ((size_t)pp * nl + i) * (p + 1);
                for (int j = 0; j <= p; j++) out[j] = 0.0;
                if (i >= nlam) continue;
                const double *b = bstd + ((size_t)pp * nl + i) * q;
                /* get_beta (ref: src/oem_big.h:880-897) + placement (src/oem_big.cpp:213-220) */
                if (intercept) out[0] = b[0];
                for (int j = 0; j < p; j++) out[j + 1] = b[j + off] * (standardize ? colsq_inv[j] : 1.0);
            }
        }
    free(XtXm); free(XX); free(XY); free(colsums); free(pf); free(lam); free(bstd);
    return rc;
}

/* ------------------------------------------------------------------ */
/* xval.oem: ref src/oem_xval_dense.cpp:31-482 + src/oem_xval_dense.h.
 * Per-fold Grams once, K+1 fits on sums of them, per-observation CV error with Welford's update. */
static int xval_paths(const double *XX, const double *XY, int q, double d, const orc_opts *oo, const double *pf,
                      const double *lam, int nl, grp_t *g, int *have_g, double *bstd, int32_t *niter)
{
    core_t s; memset(&s, 0, sizeof s);
    if (core_alloc(&s, q)) return -1;
    double *A = (double *)malloc(sizeof(double) * (size_t)q * q);
    if (!A) return fail("oracle: out of memory");
    for (size_t k = 0; k < (size_t)q * q; k++) A[k] = -XX[k];          /* ref: oem_xval_dense.h:786-789, 845-848 */
    for (int j = 0; j < q; j++) A[(size_t)j * q + j] += d;
    s.A = A; s.XY = XY; s.d = d;
    int rc = 0;
    for (int pp = 0; pp < oo->npen && rc == 0; pp++) {
        int pen = oo->penalty[pp];
        int nlam = (pen == ORC_OLS) ? 1 : nl;
        double ak = 1.0;
        for (int i = 0; i < nlam; i++) {
            if (i == 0) {                                                 /* init(): cold start, ref :1039-1058 */
                memset(s.beta, 0, sizeof(double) * (size_t)q);
                /* get_group_indexes scans nvars + intercept entries (ref :636), once per solver (found_grp_idx) */
                if (!*have_g && pen_is_grp(pen)) { if (build_groups(g, oo, q)) { rc = -1; break; } *have_g = 1; }
            }
            orc_opts o2 = *oo; o2.accelerate = 0;
            int it = solve_one(&s, pen, lam[(size_t)pp * nl + i], &o2, pf, g, &ak);
            if (niter) niter[(size_t)pp * nl + i] = it;
            memcpy(bstd + ((size_t)pp * nl + i) * q, s.beta, sizeof(double) * (size_t)q);
        }
    }
    free(A); free(s.u);
    return rc;
}

/* weights: n observation weights or NULL (ref src/oem_xval_dense.h:486-623 XtWX_xval / XtWX_xval_int: the fold Grams, X'y and
 * the intercept border carry the weights; the column scales and the divisor nobs do NOT -- "we do not standardize with respect
 * to weights", :533-535 -- and the CV error of row i is multiplied by w_i, ref src/oem_xval_dense.cpp:389-437). */
int orc_xval_dense_w(const double *x, int64_t n, int32_t p, const double *y, const double *weights, const int32_t *foldid, int32_t nfolds,
                     int32_t standardize, int32_t intercept, int32_t type_measure, const orc_opts *o,
                     double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out,
                     double *cvm, double *cvsd);
int orc_xval_dense(const double *x, int64_t n, int32_t p, const double *y, const int32_t *foldid, int32_t nfolds,
                   int32_t standardize, int32_t intercept, int32_t type_measure, const orc_opts *o,
                   double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out,
                   double *cvm, double *cvsd)
{
    return orc_xval_dense_w(x, n, p, y, NULL, foldid, nfolds, standardize, intercept, type_measure, o, beta, lambda_out, niter, loss,
                            d_out, cvm, cvsd);
}
int orc_xval_dense_w(const double *x, int64_t n, int32_t p, const double *y, const double *weights, const int32_t *foldid, int32_t nfolds,
                     int32_t standardize, int32_t intercept, int32_t type_measure, const orc_opts *o,
                     double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out,
                     double *cvm, double *cvsd)
{
    if (n <= p) return fail("dimension of x larger than number of observations");     /* ref: oem_xval_dense.h:690-731 */
    const int off = intercept ? 1 : 0, q = p + off, K = nfolds;
    const int nl = nl_of(o);
    const size_t qq = (size_t)q * q;
    double *fxx = (double *)calloc(qq * K, sizeof(double)), *fxy = (double *)calloc((size_t)q * K, sizeof(double));
    double *fcs = (double *)calloc((size_t)p * K, sizeof(double));
    int64_t *fn = (int64_t *)calloc((size_t)K, sizeof(int64_t));
    double *XX = (double *)malloc(sizeof(double) * qq), *XY = (double *)malloc(sizeof(double) * (size_t)q);
    double *colsq = (double *)malloc(sizeof(double) * (size_t)p * 2), *colsq_inv = colsq + p;
    double *pf = (double *)calloc((size_t)q, sizeof(double));
    double *lam = (double *)malloc(sizeof(double) * (size_t)o->npen * nl);
    double *bstd = (double *)malloc(sizeof(double) * (size_t)o->npen * nl * q);
    /* coefficients of every fold fit on the original scale: [fold][pen][lambda][p + 1] */
    double *bf = (double *)calloc((size_t)K * o->npen * nl * (p + 1), sizeof(double));
    if (!fxx || !fxy || !fcs || !fn || !XX || !XY || !colsq || !pf || !lam || !bstd || !bf) return fail("oracle: out of memory");
    for (int j = 0; j < p; j++) pf[j + off] = o->penalty_factor[j];     /* ref: oem_xval_dense.cpp:139-146 */
    /* per-fold pieces (ref: oem_xval_dense.h:358-484) */
    for (int64_t i = 0; i < n; i++) {
        int k = foldid[i] - 1;
        if (k < 0 || k >= K) return fail("oracle: foldid out of range");
        double *G = fxx + qq * k, *b = fxy + (size_t)q * k, *cs = fcs + (size_t)p * k;
        fn[k]++;
        const double wi = weights ? weights[i] : 1.0;
        if (intercept) { G[0] += wi; b[0] += wi * y[i]; }
        for (int c = 0; c < p; c++) {
            double xc = x[(size_t)c * n + i];
            b[c + off] += xc * (y[i] * wi);
            cs[c] += xc * xc;                                           /* never weighted (ref :533-535) */
            if (intercept) { G[(size_t)(c + 1) * q] += wi * xc; G[c + 1] += wi * xc; }
            for (int r = c; r < p; r++) G[(size_t)(c + off) * q + (r + off)] += x[(size_t)r * n + i] * (wi * xc);
        }
    }
    for (int k = 0; k < K; k++) {
        double *G = fxx + qq * k;
        for (int c = 0; c < p; c++)
            for (int r = c + 1; r < p; r++) G[(size_t)(r + off) * q + (c + off)] = G[(size_t)(c + off) * q + (r + off)];
    }
    grp_t g; memset(&g, 0, sizeof g); int have_g = 0;
    orc_opts oo = *o; oo.penalty_factor = pf;
    int rc = 0;
    for (int ff = 0; ff <= K && rc == 0; ff++) {
        /* ref: oem_xval_dense.h:733-784 (ff == 0) and :791-853 (update_xtx) */
        memset(XX, 0, sizeof(double) * qq); memset(XY, 0, sizeof(double) * (size_t)q);
        for (int j = 0; j < p; j++) colsq[j] = 0.0;
        int64_t nobs = 0;
        for (int k = 1; k <= K; k++) {
            if (k == ff) continue;
            const double *G = fxx + qq * (k - 1), *b = fxy + (size_t)q * (k - 1), *cs = fcs + (size_t)p * (k - 1);
            for (size_t t = 0; t < qq; t++) XX[t] += G[t];
            for (int j = 0; j < q; j++) XY[j] += b[j];
            for (int j = 0; j < p; j++) colsq[j] += cs[j];
            nobs += fn[k - 1];
        }
        if (nobs <= p) { rc = fail("dimension of x larger than number of observations"); break; }
        for (int j = 0; j < p; j++) {
            colsq[j] /= ((double)nobs - 1.0);
            if (colsq[j] == 0.0) colsq[j] = 1.0;
            colsq_inv[j] = 1.0 / sqrt(colsq[j]);
        }
        if (standardize) {
            for (int c = 0; c < p; c++)
                for (int r = 0; r < p; r++)
                    XX[(size_t)(c + off) * q + (r + off)] = colsq_inv[r] * XX[(size_t)(c + off) * q + (r + off)] * colsq_inv[c];
            if (intercept) for (int j = 0; j < p; j++) { XX[(size_t)(j + 1) * q] *= colsq_inv[j]; XX[j + 1] *= colsq_inv[j]; }
            for (int j = 0; j < p; j++) XY[j + off] *= colsq_inv[j];
        }
        for (size_t t = 0; t < qq; t++) XX[t] /= (double)nobs;
        for (int j = 0; j < q; j++) XY[j] /= (double)nobs;
        double d = (o->d_override > 0) ? o->d_override : orc_eig_max(XX, q) * 1.005;
        if (ff == 0) {
            *d_out = d;
            double lmax = 0.0;                 /* excludes the intercept slot (ref: oem_xval_dense.h:1025-1032) */
            for (int j = off; j < q; j++) if (fabs(XY[j]) > lmax) lmax = fabs(XY[j]);
            lambda_grid(o, lmax, lam);
            for (size_t k = 0; k < (size_t)o->npen * nl; k++) { lambda_out[k] = lam[k]; loss[k] = 1e99; niter[k] = 0; }
        }
        rc = xval_paths(XX, XY, q, d, &oo, pf, lam, nl, &g, &have_g, bstd, ff == 0 ? niter : NULL);
        if (rc) break;
        for (int pp = 0; pp < o->npen; pp++) {
            int nlam = (o->penalty[pp] == ORC_OLS) ? 1 : nl;
            for (int i = 0; i < nl; i++) {
                double *out = (ff == 0) ? beta + ((size_t)pp * nl + i) * (p + 1)
                                        : bf + (((size_t)(ff - 1) * o->npen + pp) * nl + i) * (p + 1);
                for (int j = 0; j <= p; j++) out[j] = 0.0;
                if (i >= nlam) continue;
                const double *b = bstd + ((size_t)pp * nl + i) * q;
                if (intercept) out[0] = b[0];                                   /* get_beta, ref :1069-1086 */
                for (int j = 0; j < p; j++) out[j + 1] = b[j + off] * (standardize ? colsq_inv[j] : 1.0);
                if (ff == 0 && o->compute_loss) {                               /* get_loss, ref :1088-1117 */
                    double l = 0.0;
                    for (int64_t r = 0; r < n; r++) {
                        double t = y[r];
                        for (int j = 0; j < p; j++) t -= x[(size_t)j * n + r] * out[j + 1];
                        t -= out[0];
                        l += t * t;
                    }
                    loss[(size_t)pp * nl + i] = l;
                }
            }
        }
    }
    /* CV error per observation, Welford over i (ref: oem_xval_dense.cpp:343-461) */
    if (rc == 0)
        for (int pp = 0; pp < o->npen; pp++) {
            int nlam = (o->penalty[pp] == ORC_OLS) ? 1 : nl;
            double *mean = cvm + (size_t)pp * nl, *ss = cvsd + (size_t)pp * nl;
            for (int i = 0; i < nl; i++) { mean[i] = 0.0; ss[i] = 0.0; }
            for (int64_t r = 0; r < n; r++) {
                const double *B = bf + ((size_t)(foldid[r] - 1) * o->npen + pp) * nl * (p + 1);
                for (int i = 0; i < nlam; i++) {
                    const double *b = B + (size_t)i * (p + 1);
                    double pred = 0.0;
                    for (int j = 0; j < p; j++) pred += x[(size_t)j * n + r] * b[j + 1];
                    pred += b[0];                                               /* 0 without an intercept */
                    double res = y[r] - pred;
                    double v = (type_measure == 1) ? fabs(res) : res * res;
                    if (weights) v *= weights[r];                               /* ref: oem_xval_dense.cpp:389-437 */
                    double delta = v - mean[i];
                    mean[i] += delta / (double)(r + 1);
                    ss[i] += delta * (v - mean[i]);
                }
            }
            for (int i = 0; i < nlam; i++) ss[i] = sqrt(ss[i] / (double)(n - 1)) / sqrt((double)n);
        }
    if (have_g) free_groups(&g);
    free(fxx); free(fxy); free(fcs); free(fn); free(XX); free(XY); free(colsq); free(pf); free(lam); free(bstd); free(bf);
    return rc;
}

/* ------------------------------------------------------------------ */
/* oem() on a sparse X: ref src/oem_sparse.cpp:30-267 + src/oem_sparse.h (n > p branch, no observation weights).
 * X in compressed sparse column form as a dgCMatrix holds it: colptr[p + 1], rowidx[nnz] (increasing inside a column), val[nnz].
 * What differs from oemBig: the intercept column holds intval = sqrt(mean(diag(XX)) / n) instead of 1 (ref :577-593), the
 * coefficient of that column is rescaled IN PLACE by get_beta after every lambda (ref :897-900) and so enters the next warm
 * start rescaled, lambda_zero leaves the intercept slot out (ref :854-863), and the group vector has q = p + intercept entries
 * (ref :452-470 scans all groups.size() slots; with an intercept R prepends the unpenalised group 0, ref R/oem.R:296-338). */
int orc_fit_sparse(int64_t n, int32_t p, const int64_t *colptr, const int32_t *rowidx, const double *val, const double *y,
                   int32_t standardize, int32_t intercept, const orc_opts *o,
                   double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out)
{
    if (n <= p) {
        /* the XXt branch (ref: src/oem_sparse.h:607-612, 638-647) is oemBig's, line for line: see big_wide */
        double *xd = (double *)calloc((size_t)n * p, sizeof(double));
        if (!xd) return fail("oracle: out of memory");
        for (int j = 0; j < p; j++)
            for (int64_t k = colptr[j]; k < colptr[j + 1]; k++) xd[(size_t)j * n + rowidx[k]] = val[k];
        int rcw = big_wide(xd, n, p, y, standardize, intercept, o, beta, lambda_out, niter, loss, d_out);
        free(xd);
        if (rcw == 0 && o->compute_loss) {
            /* get_loss (ref: src/oem_sparse.h:932-941): the member times colsq_inv with standardize -- the coefficients get_beta returned */
            const int nl = nl_of(o);
            double *res = (double *)malloc(sizeof(double) * (size_t)n);
            if (!res) return fail("oracle: out of memory");
            for (int pp = 0; pp < o->npen; pp++) {
                int nlam = (o->penalty[pp] == ORC_OLS) ? 1 : nl;
                for (int i = 0; i < nlam; i++) {
                    const double *out = beta + ((size_t)pp * nl + i) * (p + 1);
                    memcpy(res, y, sizeof(double) * (size_t)n);
                    for (int j = 0; j < p; j++)
                        for (int64_t k = colptr[j]; k < colptr[j + 1]; k++) res[rowidx[k]] -= val[k] * out[j + 1];
                    double l = 0.0;
                    for (int64_t r = 0; r < n; r++) l += res[r] * res[r];
                    loss[(size_t)pp * nl + i] = l;
                }
            }
            free(res);
        }
        return rcw;
    }
    const int off = intercept ? 1 : 0, q = p + off, nl = nl_of(o);
    const size_t qq = (size_t)q * q;
    double *XX = (double *)calloc(qq, sizeof(double)), *XY = (double *)calloc((size_t)q, sizeof(double));
    double *colsq_inv = (double *)malloc(sizeof(double) * (size_t)p), *colsums = (double *)calloc((size_t)p, sizeof(double));
    double *dense = (double *)calloc((size_t)n, sizeof(double));
    double *pf = (double *)calloc((size_t)q, sizeof(double));
    double *lam = (double *)malloc(sizeof(double) * (size_t)o->npen * nl);
    if (!XX || !XY || !colsq_inv || !colsums || !dense || !pf || !lam) return fail("oracle: out of memory");
    for (int j = 0; j < p; j++) pf[j + off] = o->penalty_factor[j];       /* ref: src/oem_sparse.cpp:131-138 */
    /* colsq (ref: src/oem_sparse.h:493-508) */
    for (int j = 0; j < p; j++) {
        double s = 0.0;
        for (int64_t k = colptr[j]; k < colptr[j + 1]; k++) s += val[k] * val[k];
        s /= ((double)n - 1.0);
        if (s == 0.0) s = 1.0;
        colsq_inv[j] = standardize ? 1.0 / sqrt(s) : 1.0;
    }
    /* X'X, X'Y, column sums */
    double sumy = 0.0;
    for (int64_t i = 0; i < n; i++) sumy += y[i];
    for (int a = 0; a < p; a++) {
        for (int64_t k = colptr[a]; k < colptr[a + 1]; k++) dense[rowidx[k]] = val[k];
        for (int b = a; b < p; b++) {
            double s = 0.0;
            for (int64_t k = colptr[b]; k < colptr[b + 1]; k++) s += val[k] * dense[rowidx[k]];
            s = colsq_inv[a] * s * colsq_inv[b];
            XX[(size_t)(a + off) * q + (b + off)] = s; XX[(size_t)(b + off) * q + (a + off)] = s;
        }
        double sxy = 0.0, sx = 0.0;
        for (int64_t k = colptr[a]; k < colptr[a + 1]; k++) { sxy += val[k] * y[rowidx[k]]; sx += val[k]; dense[rowidx[k]] = 0.0; }
        XY[a + off] = sxy * colsq_inv[a];
        colsums[a] = sx;
    }
    double intval = 1.0;
    if (intercept) {
        double xxdiag = 0.0;                                                 /* ref :577-593 */
        for (int j = 0; j < p; j++) xxdiag += XX[(size_t)(j + 1) * q + (j + 1)];
        xxdiag /= (double)p;
        intval = sqrt(xxdiag / (double)n);
        for (int j = 0; j < p; j++) {
            double c = colsums[j] * intval * colsq_inv[j];
            XX[(size_t)(j + 1) * q] = c; XX[j + 1] = c;
        }
        XX[0] = xxdiag;
        XY[0] = sumy * intval;                                               /* ref :829-836 */
    }
    for (size_t t = 0; t < qq; t++) XX[t] /= (double)n;
    for (int j = 0; j < q; j++) XY[j] /= (double)n;
    double d = (o->d_override > 0) ? o->d_override : orc_eig_max(XX, q) * 1.005;
    *d_out = d;
    double lmax = 0.0;
    for (int j = off; j < q; j++) if (fabs(XY[j]) > lmax) lmax = fabs(XY[j]);
    lambda_grid(o, lmax, lam);
    for (size_t k = 0; k < (size_t)o->npen * nl; k++) { lambda_out[k] = lam[k]; loss[k] = 1e99; niter[k] = 0; }
    core_t s; memset(&s, 0, sizeof s);
    if (core_alloc(&s, q)) return -1;
    double *A = (double *)malloc(sizeof(double) * qq);
    for (size_t t = 0; t < qq; t++) A[t] = -XX[t];
    for (int j = 0; j < q; j++) A[(size_t)j * q + j] += d;
    s.A = A; s.XY = XY; s.d = d;
    grp_t g; memset(&g, 0, sizeof g); int have_g = 0, rc = 0;
    orc_opts oo = *o; oo.penalty_factor = pf; oo.accelerate = 0;
    for (int pp = 0; pp < o->npen && rc == 0; pp++) {
        int pen = o->penalty[pp];
        int nlam = (pen == ORC_OLS) ? 1 : nl;
        double ak = 1.0;
        for (int i = 0; i < nl; i++) {
            double *out = beta + ((size_t)pp * nl + i) * (p + 1);
            for (int j = 0; j <= p; j++) out[j] = 0.0;
            if (i >= nlam) continue;
            if (i == 0) {
                memset(s.beta, 0, sizeof(double) * (size_t)q);
                if (!have_g && pen_is_grp(pen)) { if (build_groups(&g, &oo, q)) { rc = -1; break; } have_g = 1; }   /* groups.size() = q: R prepends group 0 for the intercept (ref R/oem.R:296-338, src/oem_sparse.h:465) */
            }
            niter[(size_t)pp * nl + i] = solve_one(&s, pen, lam[(size_t)pp * nl + i], &oo, pf, &g, &ak);
            if (intercept) s.beta[0] *= intval;                              /* get_beta, in place (ref :897-900) */
            if (intercept) out[0] = s.beta[0];
            for (int j = 0; j < p; j++) out[j + 1] = s.beta[j + off] * colsq_inv[j];
            if (o->compute_loss) {                                           /* get_loss after get_beta (ref :919-944) */
                memcpy(dense, y, sizeof(double) * (size_t)n);
                for (int j = 0; j < p; j++)
                    for (int64_t k = colptr[j]; k < colptr[j + 1]; k++) dense[rowidx[k]] -= val[k] * out[j + 1];
                double l = 0.0;
                for (int64_t r = 0; r < n; r++) { double t = dense[r] - (intercept ? s.beta[0] : 0.0); l += t * t; }
                loss[(size_t)pp * nl + i] = l;
                memset(dense, 0, sizeof(double) * (size_t)n);
            }
        }
    }
    if (have_g) free_groups(&g);
    free(A); free(s.u); free(XX); free(XY); free(colsq_inv); free(colsums); free(dense); free(pf); free(lam);
    return rc;
}
