/* tests/r_api_stub/fake_oemgpu.h -- TESTS ONLY: what the recording fake of liboemgpu saw in its last call. */
#ifndef OEM_TEST_FAKE_OEMGPU_H
#define OEM_TEST_FAKE_OEMGPU_H
#include "oemgpu.h"
#ifdef __cplusplus
extern "C" {
#endif
struct fake_record {
    const char *entry;
    const void *x, *y, *weights, *xty, *scale_factor;
    int64_t n; int32_t p, nshards, standardize, intercept, nfolds, type_measure;
    const int32_t *foldid, *rowidx; const int64_t *colptr; const double *values;
    oemgpu_opts o;                     /* shallow: the pointers live until the end of the .Call */
    int interrupt_answer;              /* what o.interrupt returned, when polled */
    int releases;                      /* oemgpu_release_cache calls so far */
};
extern struct fake_record fake;
extern int fake_rc;                    /* what the next call returns (0: fills the outputs) */
extern int fake_poll_interrupt;        /* 1: the next call polls o->interrupt as the library does between row blocks */
double fake_beta(int k, int i, int j);
double fake_lambda(int k, int i);
int    fake_niter(int k, int i);
double fake_cvm(int k, int i);
#define FAKE_D 4.25
#ifdef __cplusplus
}
#endif
#endif
