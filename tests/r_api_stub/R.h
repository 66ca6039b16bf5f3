/* tests/r_api_stub/R.h -- TESTS ONLY (see Rinternals.h in this directory). */
#ifndef OEM_TEST_R_STUB_R_H
#define OEM_TEST_R_STUB_R_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
char *R_alloc(size_t n, int size);          /* transient storage, released at the end of .Call */
void  R_CheckUserInterrupt(void);
#ifdef __cplusplus
}
#endif
#endif
