/* tests/r_api_stub/r_stub_runtime.c -- TESTS ONLY.  The few R C-API functions r/oem_shim.c calls, over a toy object model.
 * Not R: see the header comment of Rinternals.h in this directory for what this does and does not show.
 *
 * The one piece of behaviour worth having beyond "it links": every allocation first COLLECTS -- whatever cannot be reached from
 * the protect stack, the roots of the running `.Call` or a permanent object is marked dead -- and every accessor aborts on a dead
 * object.  That is R under gctorture(TRUE), the regime in which a missing PROTECT shows. */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "R.h"
#include "Rinterface.h"
#include "r_stub_runtime.h"

struct SEXPREC {
    int type, dead, mark, permanent;
    R_xlen_t len;
    void *data;
    struct { SEXP sym, val; } attr[6];
    int nattr;
    struct SEXPREC *next;
};

static struct SEXPREC nil_rec = {NILSXP, 0, 0, 1, 0, NULL, {{NULL, NULL}}, 0, NULL};
SEXP R_NilValue = &nil_rec, R_NamesSymbol, R_DimSymbol;

static SEXP all_objects;
static SEXP pstack[256], roots[64];
static int pdepth, nroots;
static void *transient[256];
static int ntransient;
jmp_buf stub_jmp;
static jmp_buf toplevel_jmp;
static int in_toplevel;
char stub_error_msg[512];
int stub_pending_interrupt;

static void die(const char *what)
{
    fprintf(stderr, "r_stub_runtime: %s\n", what);
    abort();
}

static SEXP live(SEXP x, const char *who)
{
    if (!x) die("NULL SEXP");
    if (x->dead) { fprintf(stderr, "r_stub_runtime: %s on a collected SEXP (a PROTECT is missing)\n", who); abort(); }
    return x;
}

static void mark(SEXP x)
{
    if (!x || x->mark) return;
    x->mark = 1;
    for (int a = 0; a < x->nattr; a++) { mark(x->attr[a].sym); mark(x->attr[a].val); }
    if (x->type == VECSXP || x->type == STRSXP)
        for (R_xlen_t i = 0; i < x->len; i++) mark(((SEXP *)x->data)[i]);
}

static void collect(void)
{
    for (SEXP o = all_objects; o; o = o->next) o->mark = 0;
    for (int i = 0; i < pdepth; i++) mark(pstack[i]);
    for (int i = 0; i < nroots; i++) mark(roots[i]);
    for (SEXP o = all_objects; o; o = o->next)
        if (o->permanent) mark(o);
    for (SEXP o = all_objects; o; o = o->next)
        if (!o->mark && !o->dead) {
            o->dead = 1;
            if (o->data && o->type != EXTPTRSXP) memset(o->data, 0xA5, (size_t)o->len * (o->type == REALSXP ? 8 : o->type == VECSXP || o->type == STRSXP ? sizeof(SEXP) : o->type == CHARSXP ? 1 : 4));
        }
}

static SEXP new_object(int type, R_xlen_t n, size_t elt)
{
    collect();
    SEXP o = (SEXP)calloc(1, sizeof *o);
    o->type = type; o->len = n;
    o->data = calloc((size_t)(n > 0 ? n : 1) + (type == CHARSXP || type == SYMSXP), elt);
    if (type == VECSXP || type == STRSXP) for (R_xlen_t i = 0; i < n; i++) ((SEXP *)o->data)[i] = R_NilValue;
    o->next = all_objects; all_objects = o;
    return o;
}

static void init_once(void)
{
    if (R_NamesSymbol) return;
    R_NamesSymbol = Rf_install("names");
    R_DimSymbol = Rf_install("dim");
}

/* ---- the R names ---- */
int TYPEOF(SEXP x) { return live(x, "TYPEOF")->type; }
R_xlen_t XLENGTH(SEXP x) { return live(x, "XLENGTH")->len; }
double *REAL(SEXP x) { if (live(x, "REAL")->type != REALSXP) die("REAL() on a non-numeric"); return (double *)x->data; }
int *INTEGER(SEXP x) { if (live(x, "INTEGER")->type != INTSXP && x->type != LGLSXP) die("INTEGER() on a non-integer"); return (int *)x->data; }
int *LOGICAL(SEXP x) { if (live(x, "LOGICAL")->type != LGLSXP) die("LOGICAL() on a non-logical"); return (int *)x->data; }
const char *CHAR(SEXP x) { if (live(x, "CHAR")->type != CHARSXP) die("CHAR() on a non-CHARSXP"); return (const char *)x->data; }
SEXP STRING_ELT(SEXP x, R_xlen_t i)
{
    if (live(x, "STRING_ELT")->type != STRSXP || i < 0 || i >= x->len) die("STRING_ELT out of range / wrong type");
    return ((SEXP *)x->data)[i];
}
SEXP VECTOR_ELT(SEXP x, R_xlen_t i)
{
    if (live(x, "VECTOR_ELT")->type != VECSXP || i < 0 || i >= x->len) die("VECTOR_ELT out of range / wrong type");
    return ((SEXP *)x->data)[i];
}
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v)
{
    if (live(x, "SET_STRING_ELT")->type != STRSXP || i < 0 || i >= x->len || live(v, "SET_STRING_ELT value")->type != CHARSXP) die("SET_STRING_ELT misuse");
    ((SEXP *)x->data)[i] = v;
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v)
{
    if (live(x, "SET_VECTOR_ELT")->type != VECSXP || i < 0 || i >= x->len) die("SET_VECTOR_ELT misuse");
    ((SEXP *)x->data)[i] = live(v, "SET_VECTOR_ELT value");
    return v;
}

SEXP Rf_allocVector(unsigned int type, R_xlen_t n)
{
    init_once();
    switch (type) {
    case REALSXP: return new_object(REALSXP, n, 8);
    case INTSXP: case LGLSXP: return new_object((int)type, n, 4);
    case VECSXP: case STRSXP: return new_object((int)type, n, sizeof(SEXP));
    default: die("Rf_allocVector: type outside the stand-in");
    }
    return R_NilValue;
}

SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol)
{
    SEXP m = PROTECT(Rf_allocVector(type, (R_xlen_t)nrow * ncol));
    SEXP dim = Rf_allocVector(INTSXP, 2);
    INTEGER(dim)[0] = nrow; INTEGER(dim)[1] = ncol;
    Rf_setAttrib(m, R_DimSymbol, dim);
    UNPROTECT(1);
    return m;
}

SEXP Rf_mkChar(const char *s)
{
    SEXP c = new_object(CHARSXP, (R_xlen_t)strlen(s), 1);
    memcpy(c->data, s, strlen(s) + 1);
    return c;
}

SEXP Rf_ScalarReal(double v)
{
    SEXP r = Rf_allocVector(REALSXP, 1);
    REAL(r)[0] = v;
    return r;
}

SEXP Rf_install(const char *name)
{
    for (SEXP o = all_objects; o; o = o->next)
        if (o->type == SYMSXP && strcmp((const char *)o->data, name) == 0) return o;
    SEXP s = new_object(SYMSXP, (R_xlen_t)strlen(name), 1);
    memcpy(s->data, name, strlen(name) + 1);
    s->permanent = 1;
    return s;
}

SEXP Rf_getAttrib(SEXP x, SEXP name)
{
    live(x, "Rf_getAttrib");
    for (int a = 0; a < x->nattr; a++)
        if (x->attr[a].sym == name) return x->attr[a].val;
    return R_NilValue;
}

SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP v)
{
    live(x, "Rf_setAttrib"); live(v, "Rf_setAttrib value");
    for (int a = 0; a < x->nattr; a++)
        if (x->attr[a].sym == name) { x->attr[a].val = v; return v; }
    if (x->nattr == 6) die("too many attributes for the stand-in");
    x->attr[x->nattr].sym = name; x->attr[x->nattr].val = v; x->nattr++;
    return v;
}

SEXP R_do_slot(SEXP obj, SEXP name)
{
    SEXP v = Rf_getAttrib(obj, name);
    if (v == R_NilValue) Rf_error("no slot of name \"%s\"", (const char *)name->data);
    return v;
}

SEXP Rf_coerceVector(SEXP x, unsigned int type)
{
    live(x, "Rf_coerceVector");
    if ((unsigned)x->type == type) return x;
    if (type != INTSXP || x->type != REALSXP) die("Rf_coerceVector: only numeric -> integer in the stand-in");
    SEXP r = Rf_allocVector(INTSXP, x->len);
    for (R_xlen_t i = 0; i < x->len; i++) INTEGER(r)[i] = (int)REAL(x)[i];
    return r;
}

static double scalar_of(SEXP x, const char *who)
{
    live(x, who);
    if (x->len < 1) die("as*() of a zero-length vector");
    if (x->type == REALSXP) return ((double *)x->data)[0];
    if (x->type == INTSXP || x->type == LGLSXP) return (double)((int *)x->data)[0];
    die("as*() of a non-numeric");
    return 0.0;
}
int Rf_asInteger(SEXP x) { return (int)scalar_of(x, "Rf_asInteger"); }
double Rf_asReal(SEXP x) { return scalar_of(x, "Rf_asReal"); }
int Rf_asLogical(SEXP x) { return scalar_of(x, "Rf_asLogical") != 0.0; }
void *R_ExternalPtrAddr(SEXP x) { if (live(x, "R_ExternalPtrAddr")->type != EXTPTRSXP) die("not an external pointer"); return x->data; }

SEXP Rf_protect(SEXP x)
{
    if (pdepth == 256) die("protect stack overflow");
    pstack[pdepth++] = live(x, "PROTECT");
    return x;
}
void Rf_unprotect(int n)
{
    if (n > pdepth) die("UNPROTECT of more than is protected");
    pdepth -= n;
}

void Rf_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(stub_error_msg, sizeof stub_error_msg, fmt, ap);
    va_end(ap);
    pdepth = 0;                                   /* R unwinds the protect stack to the context it jumps to */
    longjmp(stub_jmp, 1);
}

void Rf_onintr(void)
{
    pdepth = 0;
    longjmp(stub_jmp, 2);
}

char *R_alloc(size_t n, int size)
{
    if (ntransient == 256) die("too many R_alloc blocks for the stand-in");
    void *b = calloc(n ? n : 1, (size_t)size);
    transient[ntransient++] = b;
    return (char *)b;
}

void R_CheckUserInterrupt(void)
{
    if (!stub_pending_interrupt) return;
    if (in_toplevel) longjmp(toplevel_jmp, 1);
    Rf_onintr();
}

Rboolean R_ToplevelExec(void (*fun)(void *), void *data)
{
    const int saved = pdepth;
    if (in_toplevel) die("nested R_ToplevelExec");
    in_toplevel = 1;
    if (setjmp(toplevel_jmp)) { in_toplevel = 0; pdepth = saved; return FALSE; }
    fun(data);
    in_toplevel = 0;
    return TRUE;
}

/* ---- what the driver needs ---- */
SEXP stub_real(const double *v, R_xlen_t n) { SEXP r = Rf_allocVector(REALSXP, n); if (n) memcpy(r->data, v, sizeof(double) * (size_t)n); return r; }
SEXP stub_int(const int *v, R_xlen_t n) { SEXP r = Rf_allocVector(INTSXP, n); if (n) memcpy(r->data, v, sizeof(int) * (size_t)n); return r; }
SEXP stub_lgl(int v) { SEXP r = Rf_allocVector(LGLSXP, 1); ((int *)r->data)[0] = v; return r; }
SEXP stub_str(const char *const *v, R_xlen_t n)
{
    SEXP r = PROTECT(Rf_allocVector(STRSXP, n));
    for (R_xlen_t i = 0; i < n; i++) SET_STRING_ELT(r, i, Rf_mkChar(v[i]));
    UNPROTECT(1);
    return r;
}
SEXP stub_real_matrix(const double *v, int nrow, int ncol)
{
    SEXP m = Rf_allocMatrix(REALSXP, nrow, ncol);
    memcpy(m->data, v, sizeof(double) * (size_t)nrow * ncol);
    return m;
}
SEXP stub_list(R_xlen_t n) { return Rf_allocVector(VECSXP, n); }
void stub_set_names(SEXP list, const char *const *names)
{
    PROTECT(list);
    Rf_setAttrib(list, R_NamesSymbol, stub_str(names, list->len));
    UNPROTECT(1);
}
SEXP stub_extptr(void *p) { SEXP e = new_object(EXTPTRSXP, 0, 1); free(e->data); e->data = p; return e; }
SEXP stub_s4(void) { init_once(); return new_object(S4SXP, 0, 1); }
void stub_root(SEXP x) { if (nroots == 64) die("too many roots"); roots[nroots++] = live(x, "stub_root"); }
void stub_end_call(void)
{
    for (int i = 0; i < ntransient; i++) free(transient[i]);
    ntransient = 0; nroots = 0;
    collect();
}
int stub_protect_depth(void) { return pdepth; }
