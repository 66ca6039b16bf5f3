/*
 * tests/r_api_stub/Rinternals.h -- TESTS ONLY.  A hand-written stand-in for the handful of R C-API names that r/oem_shim.c and
 * r/oem_shim_big.cpp use, so that a compiler can read the shim in an image without R, and so that its marshalling can be RUN
 * against a recording fake of liboemgpu (tests/r_api_stub/fake_oemgpu.c, tests/test_r_shim.py).
 *
 * What this is: proof that the shim parses under -Wall -Werror, that its calls into include/oemgpu.h type-check, that the
 * arguments of `.Call` arrive in the oemgpu_opts fields they belong to, that the result list has the reference's names, storage
 * modes and dimensions (ref src/oem_dense.cpp:280-307), and that PROTECT / UNPROTECT balance on every exit path.
 * What this is NOT: R.  It pins no numerical parity, it is not an `oracle/_ref` build, it is never installed or shipped, and it says
 * nothing about R's real allocator beyond the collector emulation below (every allocation "collects" whatever is not reachable
 * from the protect stack -- R's gctorture(TRUE) -- so an unprotected SEXP that survives an allocation is caught when it is used).
 */
#ifndef OEM_TEST_R_STUB_RINTERNALS_H
#define OEM_TEST_R_STUB_RINTERNALS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE = 1 } Rboolean;

#define NILSXP     0
#define SYMSXP     1
#define CHARSXP    9
#define LGLSXP    10
#define INTSXP    13
#define REALSXP   14
#define STRSXP    16
#define VECSXP    19
#define EXTPTRSXP 22
#define S4SXP     25

extern SEXP R_NilValue, R_NamesSymbol, R_DimSymbol;

int      TYPEOF(SEXP x);
R_xlen_t XLENGTH(SEXP x);
double  *REAL(SEXP x);
int     *INTEGER(SEXP x);
int     *LOGICAL(SEXP x);
const char *CHAR(SEXP x);
SEXP     STRING_ELT(SEXP x, R_xlen_t i);
SEXP     VECTOR_ELT(SEXP x, R_xlen_t i);
void     SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP     SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);

SEXP Rf_allocVector(unsigned int type, R_xlen_t n);
SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol);
SEXP Rf_mkChar(const char *s);
SEXP Rf_ScalarReal(double v);
SEXP Rf_install(const char *name);
SEXP Rf_getAttrib(SEXP x, SEXP name);
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP v);
SEXP R_do_slot(SEXP obj, SEXP name);
SEXP Rf_coerceVector(SEXP x, unsigned int type);
int    Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
int    Rf_asLogical(SEXP x);
void  *R_ExternalPtrAddr(SEXP x);

SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x)   Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)

#if defined(__GNUC__)
__attribute__((noreturn, format(printf, 1, 2)))
#endif
void Rf_error(const char *fmt, ...);

Rboolean R_ToplevelExec(void (*fun)(void *), void *data);

#ifdef __cplusplus
}
#endif
#endif
