/* Harness helpers for the example drivers (our own code, same roles as
 * /root/reference/examples/test_utils.{h,c}: can_check_res :3-19, read_mtx_csr :21-55,
 * scatter_csr_rows :57-119, fill_B :121-154; the single-process reference product
 * mkl_csr_spmm :157-179 is replaced by a plain CSR loop so that no MKL is needed). */
#ifndef CRP_EXAMPLES_TEST_UTILS_H
#define CRP_EXAMPLES_TEST_UTILS_H
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mpi.h>
#include "mmio_utils.h"
#include "utils.h"

int can_check_res(int my_rank, int m, int n, int k);
void read_mtx_csr(const char *fname, const int need_symm, int *glb_m, int *glb_k, int glb_n, int **glb_A_rowptr,
                  int **glb_A_colidx, double **glb_A_csrval);
void scatter_csr_rows(MPI_Comm comm, int nproc, int my_rank, int *A_m_displs, int *A_nnz_displs, int *A_m_scnts,
                      int *A_nnz_scnts, int *glb_A_rowptr, int *glb_A_colidx, double *glb_A_csrval,
                      int **loc_A_rowptr_, int **loc_A_colidx_, double **loc_A_csrval_);
void fill_B(int layout, double *B, int ldB, int srow, int nrow, int scol, int ncol, double factor_i, double factor_j);
/* C = A * B, row-major, independent naive loop (the check the reference lacks: SURVEY.md section 4) */
void naive_csr_spmm(int m, int n, const int *rowptr, const int *colidx, const double *val, const double *B, int ldB,
                    double *C, int ldC);
#endif
