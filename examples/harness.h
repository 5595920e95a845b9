/* harness.h -- what the four example programs share: a sparse matrix held by rank 0, its rows
 * dealt out to the ranks, closed-form dense operands, the timing loop and the check against a
 * naive product.  The programs keep the command lines and the output lines of the reference's
 * examples (examples/test_rp_spmm.c, test_para2d_spmm.c, test_spmm_2dpg.c and
 * deprecated/examples/test_crpspmm.c under /root/reference), so that scripts written around those
 * keep working; everything behind the lines is this repository's own code. */
#ifndef CRP_EXAMPLES_HARNESS_H
#define CRP_EXAMPLES_HARNESS_H
#include <mpi.h>
#include <stddef.h>

typedef struct
{
    int     m, k;          /* rows, columns (known on every rank after hx_load) */
    int    *ptr, *idx;     /* CSR of the whole matrix: rank 0 only */
    double *val;
} hx_matrix;

typedef struct
{
    int     first, count;  /* this rank's row block */
    int    *ptr;           /* count + 1 offsets into the GLOBAL nonzero numbering */
    int    *idx;
    double *val;
    int    *cuts;          /* nproc + 1 row cuts, on every rank */
} hx_rows;

typedef struct { int rank, size; MPI_Comm comm; } hx_world;

hx_world hx_start(int *argc, char ***argv);
/* rank 0 reads the file ("B has ..", "Rank 0 read matrix A ..", "A size = .." lines), everybody learns m, k */
void hx_load(const hx_world *w, const char *path, int n_cols, hx_matrix *A);
/* 0 when a dense m x n or k x n matrix would overflow int indexing (with the reference's two lines) */
int hx_can_verify(const hx_world *w, const hx_matrix *A, int n_cols);
/* rows [cuts[r], cuts[r+1]) go to rank r; cuts valid on rank 0 on entry, on every rank on return */
void hx_deal(const hx_world *w, const hx_matrix *A, int *cuts, hx_rows *mine);
/* B(i, j) = 0.19 i + 0.24 j for a block, row-major with leading dimension ld */
void hx_dense_block(double *blk, int ld, int row0, int nrow, int col0, int ncol);
/* rank 0: naive C = A B for the closed-form B, print "||C_ref - C||_f / ||C_ref||_f = ..", return 0 when <= 1e-12 */
int hx_verify(const hx_matrix *A, int n_cols, const double *C_full);
/* n_test timed calls of fn(ctx) between barriers; rank 0 prints one "%.2f" line per call */
void hx_time_loop(const hx_world *w, int n_test, int with_barriers, void (*fn)(void *), void *ctx);
void hx_release(hx_matrix *A, hx_rows *mine);
#endif
