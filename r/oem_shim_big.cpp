/*
 * oem_shim_big.cpp -- the two big.matrix entry points of jaredhuling/oem bound to liboemgpu.
 *
 * oem_fit_big (ref src/oem_big.cpp:30-258) and oem_fit_fb_big (ref src/oem_fb_big.cpp:30-258) take the `big.matrix@address`
 * external pointer (ref R/big_oem.R:360,447-491).  The reference unwraps it with Rcpp::XPtr<BigMatrix> /
 * XPtr<FileBackedBigMatrix> (the two files are otherwise identical); XPtr<T>'s dereference is R_ExternalPtrAddr() plus a cast,
 * and FileBackedBigMatrix derives from BigMatrix, so ONE unwrap serves both symbols.  This needs bigmemory's own header (the
 * package already has `LinkingTo: bigmemory, BH`, DESCRIPTION:38-42) and therefore a C++ translation unit; everything else is the C
 * shim (r/oem_shim.c: oem_shim_fit_big), R C API only, no Rcpp.
 *
 * NOT compiled in this repository's image: neither R nor bigmemory is installed.
 */
#include <R.h>
#include <Rinternals.h>
#include <stdint.h>

#include <bigmemory/BigMatrix.h>

extern "C" {

SEXP oem_shim_fit_big(const double *x, int64_t n, int p, SEXP y_, SEXP family_, SEXP penalty_, SEXP weights_, SEXP groups_,
                      SEXP unique_groups_, SEXP group_weights_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_,
                      SEXP gamma_, SEXP tau_, SEXP penalty_factor_, SEXP standardize_, SEXP intercept_, SEXP compute_loss_, SEXP opts_);

static SEXP fit_big_matrix(SEXP x_, SEXP y_, SEXP family_, SEXP penalty_, SEXP weights_, SEXP groups_, SEXP unique_groups_,
                           SEXP group_weights_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_, SEXP gamma_, SEXP tau_,
                           SEXP penalty_factor_, SEXP standardize_, SEXP intercept_, SEXP compute_loss_, SEXP opts_)
{
    if (TYPEOF(x_) != EXTPTRSXP) Rf_error("x must be the address of a big.matrix");
    BigMatrix *bm = reinterpret_cast<BigMatrix *>(R_ExternalPtrAddr(x_));
    if (!bm) Rf_error("external pointer is not valid");                         /* what Rcpp::XPtr throws on NULL */
    if (bm->matrix_type() != 8) Rf_error("type for provided big.matrix not available");      /* ref src/oem_big.cpp:57-62: double only */
    /* the reference maps bMPtr->matrix() as ONE nrow x ncol column-major block (ref :64); a separated-column or sub-matrix
     * big.matrix is not that, and the reference would read it wrongly: refuse instead */
    if (bm->separated_columns()) Rf_error("big.matrix with separated columns is not supported");
    if (bm->row_offset() != 0 || bm->col_offset() != 0 || bm->nrow() != bm->total_rows())
        Rf_error("sub.big.matrix views are not supported: pass the whole big.matrix");
    return oem_shim_fit_big(reinterpret_cast<const double *>(bm->matrix()), (int64_t)bm->nrow(), (int)bm->ncol(), y_, family_,
                            penalty_, weights_, groups_, unique_groups_, group_weights_, lambda_, nlambda_, lmin_ratio_, alpha_, gamma_,
                            tau_, penalty_factor_, standardize_, intercept_, compute_loss_, opts_);
}

/* in-memory big.matrix (ref src/oem_big.cpp:30-48: the 19 SEXP arguments, order fixed by R/big_oem.R:470-490) */
SEXP oem_fit_big(SEXP x_, SEXP y_, SEXP family_, SEXP penalty_, SEXP weights_, SEXP groups_, SEXP unique_groups_,
                 SEXP group_weights_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_, SEXP gamma_, SEXP tau_,
                 SEXP penalty_factor_, SEXP standardize_, SEXP intercept_, SEXP compute_loss_, SEXP opts_)
{
    return fit_big_matrix(x_, y_, family_, penalty_, weights_, groups_, unique_groups_, group_weights_, lambda_, nlambda_, lmin_ratio_,
                          alpha_, gamma_, tau_, penalty_factor_, standardize_, intercept_, compute_loss_, opts_);
}

/* file-backed big.matrix (ref src/oem_fb_big.cpp:30-48, R/big_oem.R:447-468): the mapping is read through the same pointer; the
 * library's staging lanes memcpy from it block by block, so the page-ins of the mmap overlap the DMA of the previous block */
SEXP oem_fit_fb_big(SEXP x_, SEXP y_, SEXP family_, SEXP penalty_, SEXP weights_, SEXP groups_, SEXP unique_groups_,
                    SEXP group_weights_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP alpha_, SEXP gamma_, SEXP tau_,
                    SEXP penalty_factor_, SEXP standardize_, SEXP intercept_, SEXP compute_loss_, SEXP opts_)
{
    return fit_big_matrix(x_, y_, family_, penalty_, weights_, groups_, unique_groups_, group_weights_, lambda_, nlambda_, lmin_ratio_,
                          alpha_, gamma_, tau_, penalty_factor_, standardize_, intercept_, compute_loss_, opts_);
}

}  /* extern "C" */
