#include "test_utils.h"

int can_check_res(int my_rank, int m, int n, int k)
{
    if (n > INT_MAX / (m > 0 ? m : 1) || n > INT_MAX / (k > 0 ? k : 1))
    {
        if (my_rank == 0)
        {
            printf("The complete B or C matrix is too large to be stored in a dense matrix\n");
            printf("Result validation check will be skipped\n");
            fflush(stdout);
        }
        return 0;
    }
    return 1;
}

void read_mtx_csr(const char *fname, const int need_symm, int *glb_m, int *glb_k, int glb_n, int **glb_A_rowptr,
                  int **glb_A_colidx, double **glb_A_csrval)
{
    int nnz = 0, m = 0, k = 0, bandwidth = 0, *row = NULL, *col = NULL;
    double *val = NULL;
    printf("B has %d columns\n", glb_n);
    printf("Rank 0 read matrix A from file %s", fname);
    fflush(stdout);
    double st = get_wtime_sec();
    if (mm_read_sparse_RPI(fname, need_symm, &m, &k, &nnz, &row, &col, &val) != 0)
    {
        printf("\nCannot ingest %s\n", fname);
        MPI_Abort(MPI_COMM_WORLD, 2);
    }
    coo2csr(m, k, nnz, row, col, val, glb_A_rowptr, glb_A_colidx, glb_A_csrval);
    double et = get_wtime_sec();
    for (int i = 0; i < nnz; i++)
    {
        int bw = abs(row[i] - col[i]);
        if (bw > bandwidth) bandwidth = bw;
    }
    printf(" used %.2f s\n", et - st);
    printf("A size = %d * %d, nnz = %d, nnz/row = %d, bandwidth = %d\n\n", m, k, nnz, nnz / (m > 0 ? m : 1), bandwidth);
    fflush(stdout);
    *glb_m = m;
    *glb_k = k;
    free(row); free(col); free(val);
}

void scatter_csr_rows(MPI_Comm comm, int nproc, int my_rank, int *A_m_displs, int *A_nnz_displs, int *A_m_scnts,
                      int *A_nnz_scnts, int *glb_A_rowptr, int *glb_A_colidx, double *glb_A_csrval,
                      int **loc_A_rowptr_, int **loc_A_colidx_, double **loc_A_csrval_)
{
    MPI_Bcast(A_m_displs, nproc + 1, MPI_INT, 0, comm);
    MPI_Bcast(A_nnz_displs, nproc + 1, MPI_INT, 0, comm);
    for (int i = 0; i < nproc; i++)
    {
        A_m_scnts[i] = A_m_displs[i + 1] - A_m_displs[i];
        A_nnz_scnts[i] = A_nnz_displs[i + 1] - A_nnz_displs[i];
    }
    const int nrow = A_m_scnts[my_rank], nnz = A_nnz_scnts[my_rank];
    int *rp = (int *) malloc(sizeof(int) * (nrow + 1));
    int *ci = (int *) malloc(sizeof(int) * (nnz > 0 ? nnz : 1));
    double *cv = (double *) malloc(sizeof(double) * (nnz > 0 ? nnz : 1));
    /* the row-pointer slice keeps GLOBAL nnz offsets (examples/test_utils.c:78-91) */
    MPI_Scatterv(glb_A_rowptr, A_m_scnts, A_m_displs, MPI_INT, rp, nrow, MPI_INT, 0, comm);
    MPI_Scatterv(glb_A_colidx, A_nnz_scnts, A_nnz_displs, MPI_INT, ci, nnz, MPI_INT, 0, comm);
    MPI_Scatterv(glb_A_csrval, A_nnz_scnts, A_nnz_displs, MPI_DOUBLE, cv, nnz, MPI_DOUBLE, 0, comm);
    rp[nrow] = A_nnz_displs[my_rank + 1];
    MPI_Barrier(comm);
    *loc_A_rowptr_ = rp;
    *loc_A_colidx_ = ci;
    *loc_A_csrval_ = cv;
}

void fill_B(int layout, double *B, int ldB, int srow, int nrow, int scol, int ncol, double factor_i, double factor_j)
{
    for (int i = 0; i < nrow; i++)
        for (int j = 0; j < ncol; j++)
        {
            const double v = (srow + i) * factor_i + (scol + j) * factor_j;
            if (layout == 0) B[(size_t) i * ldB + j] = v;
            else B[(size_t) j * ldB + i] = v;
        }
}

void naive_csr_spmm(int m, int n, const int *rowptr, const int *colidx, const double *val, const double *B, int ldB,
                    double *C, int ldC)
{
    for (int i = 0; i < m; i++)
    {
        double *Ci = C + (size_t) i * ldC;
        for (int j = 0; j < n; j++) Ci[j] = 0.0;
        for (int p = rowptr[i]; p < rowptr[i + 1]; p++)
        {
            const double a = val[p];
            const double *Bc = B + (size_t) colidx[p] * ldB;
            for (int j = 0; j < n; j++) Ci[j] += a * Bc[j];
        }
    }
}
