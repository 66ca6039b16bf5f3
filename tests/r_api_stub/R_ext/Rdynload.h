/* tests/r_api_stub/R_ext/Rdynload.h -- TESTS ONLY (see ../Rinternals.h). */
#ifndef OEM_TEST_R_STUB_RDYNLOAD_H
#define OEM_TEST_R_STUB_RDYNLOAD_H
typedef struct _DllInfo DllInfo;
#endif
