/* tests/r_api_stub/big_matrix_maker.cpp -- TESTS ONLY: builds the BigMatrix stand-in for shim_driver.c (which is C). */
#include <cstdlib>
#include "bigmemory/BigMatrix.h"
extern "C" void *driver_big_matrix(double *data, long nrow, long ncol, int type, int sepcols, long total_rows, long row_offset)
{
    BigMatrix *b = static_cast<BigMatrix *>(std::malloc(sizeof(BigMatrix)));
    b->type = type; b->sepcols = sepcols != 0; b->nr = nrow; b->nc = ncol; b->tr = total_rows; b->roff = row_offset; b->coff = 0; b->data = data;
    return b;
}
