/*
 * mmio_utils.h -- Matrix Market ingest, same entry points as
 * /root/reference/examples/mmio_utils.h:17-33.  The parser accepts exactly
 * what the reference's reader accepts (coordinate format; real / pattern /
 * integer; general / symmetric) and rejects the rest with the same messages
 * and return value -1 (examples/mmio_utils.c:22-54, examples/mmio.c:96-179).
 */
#ifndef CRP_MMIO_UTILS_H
#define CRP_MMIO_UTILS_H

#ifdef __cplusplus
extern "C" {
#endif

/* 0-based COO; symmetric files get their off-diagonal entries mirrored after
 * the stored ones; explicit zeros and duplicates are kept. Arrays are
 * malloc'd (caller frees). */
int mm_read_sparse_RPI(const char *fname, const int need_symm, int *nrow_, int *ncol_, int *nnz_,
                       int **row_, int **col_, double **val_);

/* COO -> CSR with column indices ascending inside each row
 * (examples/mmio_utils.c:148-190). Arrays are malloc'd (caller frees). */
void coo2csr(const int nrow, const int ncol, const int nnz, const int *row, const int *col,
             const double *val, int **row_ptr_, int **col_idx_, double **csr_val_);

#ifdef __cplusplus
}
#endif
#endif
