/* tests/r_api_stub/Rinterface.h -- TESTS ONLY (see Rinternals.h in this directory). */
#ifndef OEM_TEST_R_STUB_RINTERFACE_H
#define OEM_TEST_R_STUB_RINTERFACE_H
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
__attribute__((noreturn))
#endif
void Rf_onintr(void);
#ifdef __cplusplus
}
#endif
#endif
