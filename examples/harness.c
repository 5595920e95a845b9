#include "harness.h"
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mmio_utils.h"
#include "utils.h"

hx_world hx_start(int *argc, char ***argv)
{
    hx_world w;
    MPI_Init(argc, argv);
    w.comm = MPI_COMM_WORLD;
    MPI_Comm_size(w.comm, &w.size);
    MPI_Comm_rank(w.comm, &w.rank);
    return w;
}

void hx_load(const hx_world *w, const char *path, int n_cols, hx_matrix *A)
{
    memset(A, 0, sizeof(*A));
    if (w->rank == 0)
    {
        int nnz = 0, *r = NULL, *c = NULL;
        double *v = NULL;
        printf("B has %d columns\n", n_cols);
        printf("Rank 0 read matrix A from file %s", path);
        fflush(stdout);
        const double t0 = get_wtime_sec();
        if (mm_read_sparse_RPI(path, 0, &A->m, &A->k, &nnz, &r, &c, &v) != 0)
        {
            printf("\nCannot ingest %s\n", path);
            MPI_Abort(w->comm, 2);
        }
        coo2csr(A->m, A->k, nnz, r, c, v, &A->ptr, &A->idx, &A->val);
        const double t1 = get_wtime_sec();
        int reach = 0;
        for (int e = 0; e < nnz; e++)
        {
            const int d = r[e] > c[e] ? r[e] - c[e] : c[e] - r[e];
            if (d > reach) reach = d;
        }
        printf(" used %.2f s\n", t1 - t0);
        printf("A size = %d * %d, nnz = %d, nnz/row = %d, bandwidth = %d\n\n", A->m, A->k, nnz, A->m ? nnz / A->m : 0, reach);
        fflush(stdout);
        free(r);
        free(c);
        free(v);
    }
    int dims[2] = {A->m, A->k};
    MPI_Bcast(dims, 2, MPI_INT, 0, w->comm);
    A->m = dims[0];
    A->k = dims[1];
}

int hx_can_verify(const hx_world *w, const hx_matrix *A, int n_cols)
{
    const long long lim = INT_MAX;
    if ((long long) A->m * n_cols <= lim && (long long) A->k * n_cols <= lim) return 1;
    if (w->rank == 0)
    {
        printf("The complete B or C matrix is too large to be stored in a dense matrix\n");
        printf("Result validation check will be skipped\n");
        fflush(stdout);
    }
    return 0;
}

void hx_deal(const hx_world *w, const hx_matrix *A, int *cuts, hx_rows *mine)
{
    const int P = w->size;
    MPI_Bcast(cuts, P + 1, MPI_INT, 0, w->comm);
    int *nz_cut = (int *) malloc(sizeof(int) * (P + 1));
    if (w->rank == 0)
        for (int r = 0; r <= P; r++) nz_cut[r] = A->ptr[cuts[r]];
    MPI_Bcast(nz_cut, P + 1, MPI_INT, 0, w->comm);
    int *row_cnt = (int *) malloc(sizeof(int) * P), *nz_cnt = (int *) malloc(sizeof(int) * P);
    for (int r = 0; r < P; r++)
    {
        row_cnt[r] = cuts[r + 1] - cuts[r];
        nz_cnt[r] = nz_cut[r + 1] - nz_cut[r];
    }
    mine->first = cuts[w->rank];
    mine->count = row_cnt[w->rank];
    const int my_nz = nz_cnt[w->rank];
    mine->ptr = (int *) malloc(sizeof(int) * (mine->count + 1));
    mine->idx = (int *) malloc(sizeof(int) * (my_nz + 1));
    mine->val = (double *) malloc(sizeof(double) * (my_nz + 1));
    /* offsets stay global: the engines rebase them themselves (src/rowpara_spmm.c:49-55) */
    MPI_Scatterv(A->ptr, row_cnt, cuts, MPI_INT, mine->ptr, mine->count, MPI_INT, 0, w->comm);
    mine->ptr[mine->count] = nz_cut[w->rank + 1];
    MPI_Scatterv(A->idx, nz_cnt, nz_cut, MPI_INT, mine->idx, my_nz, MPI_INT, 0, w->comm);
    MPI_Scatterv(A->val, nz_cnt, nz_cut, MPI_DOUBLE, mine->val, my_nz, MPI_DOUBLE, 0, w->comm);
    mine->cuts = (int *) malloc(sizeof(int) * (P + 1));
    memcpy(mine->cuts, cuts, sizeof(int) * (P + 1));
    free(nz_cut);
    free(row_cnt);
    free(nz_cnt);
    MPI_Barrier(w->comm);
}

void hx_dense_block(double *blk, int ld, int row0, int nrow, int col0, int ncol)
{
    for (int i = 0; i < nrow; i++)
    {
        double *dst = blk + (size_t) i * ld;
        const double base = 0.19 * (double) (row0 + i);
        for (int j = 0; j < ncol; j++) dst[j] = base + 0.24 * (double) (col0 + j);
    }
}

int hx_verify(const hx_matrix *A, int n_cols, const double *C_full)
{
    /* independent of every library path: one row of C at a time, B from its closed form */
    double *acc = (double *) malloc(sizeof(double) * (n_cols > 0 ? n_cols : 1));
    double num = 0.0, den = 0.0;
    for (int i = 0; i < A->m; i++)
    {
        for (int j = 0; j < n_cols; j++) acc[j] = 0.0;
        for (int e = A->ptr[i]; e < A->ptr[i + 1]; e++)
        {
            const double a = A->val[e], bi = 0.19 * (double) A->idx[e];
            for (int j = 0; j < n_cols; j++) acc[j] += a * (bi + 0.24 * (double) j);
        }
        const double *got = C_full + (size_t) i * n_cols;
        for (int j = 0; j < n_cols; j++)
        {
            const double d = got[j] - acc[j];
            num += d * d;
            den += acc[j] * acc[j];
        }
    }
    free(acc);
    const double rel = sqrt(num) / sqrt(den);
    printf("||C_ref - C||_f / ||C_ref||_f = %e\n", rel);
    fflush(stdout);
    return rel <= 1e-12 ? 0 : 1;
}

void hx_time_loop(const hx_world *w, int n_test, int with_barriers, void (*fn)(void *), void *ctx)
{
    for (int t = 0; t < n_test; t++)
    {
        if (with_barriers) MPI_Barrier(w->comm);
        const double t0 = MPI_Wtime();
        fn(ctx);
        if (with_barriers) MPI_Barrier(w->comm);
        const double t1 = MPI_Wtime();
        if (w->rank == 0)
        {
            printf("%.2f\n", t1 - t0);
            fflush(stdout);
        }
    }
}

void hx_release(hx_matrix *A, hx_rows *mine)
{
    if (A) { free(A->ptr); free(A->idx); free(A->val); memset(A, 0, sizeof(*A)); }
    if (mine) { free(mine->ptr); free(mine->idx); free(mine->val); free(mine->cuts); memset(mine, 0, sizeof(*mine)); }
}
