/* tests/r_api_stub/bigmemory/BigMatrix.h -- TESTS ONLY: the member functions of bigmemory's BigMatrix that r/oem_shim_big.cpp
 * calls, over plain fields, so that a C++ compiler can read that file and the test driver can hand it a matrix.  Not bigmemory. */
#ifndef OEM_TEST_R_STUB_BIGMATRIX_H
#define OEM_TEST_R_STUB_BIGMATRIX_H
typedef long index_type;
class BigMatrix {
public:
    int type; bool sepcols; index_type nr, nc, tr, roff, coff; void *data;
    int matrix_type() const { return type; }
    bool separated_columns() const { return sepcols; }
    index_type nrow() const { return nr; }
    index_type ncol() const { return nc; }
    index_type total_rows() const { return tr; }
    index_type row_offset() const { return roff; }
    index_type col_offset() const { return coff; }
    void *matrix() { return data; }
};
#endif
