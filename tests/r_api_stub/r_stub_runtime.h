/* tests/r_api_stub/r_stub_runtime.h -- TESTS ONLY: what the driver (shim_driver.c) needs from the stand-in runtime beyond the
 * R names of Rinternals.h: constructors for argument SEXPs, roots, the protect depth, and the non-local exits of Rf_error / Rf_onintr. */
#ifndef OEM_TEST_R_STUB_RUNTIME_H
#define OEM_TEST_R_STUB_RUNTIME_H
#include <setjmp.h>
#include "Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
SEXP stub_real(const double *v, R_xlen_t n);
SEXP stub_int(const int *v, R_xlen_t n);
SEXP stub_lgl(int v);
SEXP stub_str(const char *const *v, R_xlen_t n);
SEXP stub_real_matrix(const double *v, int nrow, int ncol);
SEXP stub_list(R_xlen_t n);                              /* fill with SET_VECTOR_ELT, name with stub_set_names */
void stub_set_names(SEXP list, const char *const *names);
SEXP stub_extptr(void *p);
SEXP stub_s4(void);
void stub_root(SEXP x);                                  /* what `.Call` holds for its arguments */
void stub_end_call(void);                                /* releases R_alloc storage and the roots; collects */
int  stub_protect_depth(void);
extern jmp_buf stub_jmp;                                 /* setjmp value 1: Rf_error, 2: Rf_onintr */
extern char stub_error_msg[512];
extern int stub_pending_interrupt;
#ifdef __cplusplus
}
#endif
#endif
