/*
 * spmat_part.h -- host-side partition planner, same entry points and outputs
 * (bit-exact integers) as /root/reference/src/spmat_part.h:19-76.
 */
#ifndef CRP_SPMAT_PART_H
#define CRP_SPMAT_PART_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Contiguous row blocks with ~equal nnz: rblk_ptr[i+1] is the row the
 * reference's bisection on row_ptr reaches for (nnz/nblk)*(i+1)
 * (src/spmat_part.c:12-35). rblk_ptr has nblk + 1 entries. */
void csr_mat_row_partition(const int nrow, const int *row_ptr, const int nblk, int *rblk_ptr);

/* Prime factors of n, ascending; *factors is malloc'd (caller frees);
 * returns their number (src/spmat_part.c:66-81). */
int prime_factorization(int n, int **factors);

/* comm_sizes[b] = number of distinct columns touched by row block b that lie
 * outside x[x_displs[b] : x_displs[b+1]); *total_size = their sum
 * (src/spmat_part.c:38-64). */
void csr_mat_row_part_comm_size(const int nrow, const int ncol, const int *row_ptr, const int *col_idx,
                                const int nblk, const int *rblk_ptr, const int *x_displs,
                                int *comm_sizes, int *total_size);

/* Choose the pm x pn grid minimising
 *   floor(1.5 * nnz * (pn - 1)) + rA * n * sum_b comm_sizes[b]
 * by trying the prime factors of nproc from the largest, and derive the four
 * partition arrays, all malloc'd here and freed by the caller
 * (src/spmat_part.c:85-210; meaning of the arrays: src/spmat_part.h:56-70). */
void calc_spmm_part2d_from_1d(const int nproc, const int m, const int n, const int k, const int *rb_displs0,
                              const int *rowptr, const int *colidx, const int rA, int *pm, int *pn,
                              size_t *comm_cost, int **A0_rowptr, int **B_rowptr, int **AC_rowptr,
                              int **BC_colptr, int dbg_print);

#ifdef __cplusplus
}
#endif
#endif
