/* drivers.c -- the four example programs, one per value of CRP_DRIVER (see Makefile):
 *   1  test_rp_spmm.exe      <mtx> <n> <ntest> <part-method> [check]     1D row-parallel engine
 *   2  test_para2d_spmm.exe  <mtx> <n> <ntest> <part-method> [check]     2D engine, grid from the planner
 *   3  test_spmm_2dpg.exe    <mtx> <n> <nproc> <part-method>             planner dump, serial
 *   4  test_crpspmm.exe      <mtx> <n> <ntest> [check] [use-CUDA]        older all-in-one engine
 * Command lines and printed lines follow the reference's programs of the same names; the code is
 * built on harness.[ch].  part-method must be 0 (METIS is not part of this build).  With check = 1
 * rank 0 compares against a naive product and the exit code reports the result. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "harness.h"
#include "crpspmm.h"
#include "mat_redist.h"
#include "para2d_spmm.h"
#include "rowpara_spmm.h"
#include "spmat_part.h"
#include "utils.h"

#ifndef CRP_DRIVER
#error "compile with -DCRP_DRIVER=1..4"
#endif

__attribute__((unused)) static int refuse_metis(const hx_world *w, int method)
{
    if (method == 0) return 0;
    if (w->rank == 0) printf("METIS 1D row partitioning is not available in this build (part-method must be 0)\n");
    return 1;
}

__attribute__((unused)) static double *dense(size_t rows, size_t cols)
{
    return (double *) malloc(sizeof(double) * (rows * cols + 1));
}

/* ---------------------------------------------------------------------------------------------- */
#if CRP_DRIVER == 1
typedef struct { rp_spmm_p eng; const double *B; double *C; int ld; } rp_call;
static void rp_once(void *p)
{
    rp_call *c = (rp_call *) p;
    rp_spmm_exec(c->eng, 0, c->B, c->ld, c->C, c->ld);
}

int main(int argc, char **argv)
{
    if (argc < 5)
    {
        printf("Usage: %s <mtx-file> <num-of-B-col> <num-of-tests> <part-method> <check-correct>\n", argv[0]);
        printf("<part-method>: 0 for native 1D partition (METIS partitioning is not available in this build)\n");
        printf("<check-correct>: 0 or 1, optional, default value is 0\n");
        return 255;
    }
    const int n = atoi(argv[2]), n_test = atoi(argv[3]);
    int verify = argc > 5 ? atoi(argv[5]) : 0, status = 0;
    hx_world w = hx_start(&argc, &argv);
    if (refuse_metis(&w, atoi(argv[4]))) { MPI_Finalize(); return 254; }
    hx_matrix A;
    hx_load(&w, argv[1], n, &A);
    if (verify) verify = hx_can_verify(&w, &A, n);

    /* rows of A by nonzero count; rows of B like A when square, evenly otherwise */
    const double t0 = get_wtime_sec();
    int *cuts = (int *) malloc(sizeof(int) * (w.size + 1)), *b_cuts = (int *) malloc(sizeof(int) * (w.size + 1));
    if (w.rank == 0)
    {
        printf("Using naive 1D row partitioning\n");
        csr_mat_row_partition(A.m, A.ptr, w.size, cuts);
        for (int r = 0, len; r <= w.size; r++)
        {
            if (A.m == A.k) b_cuts[r] = cuts[r];
            else calc_block_spos_size(A.k, w.size, r, &b_cuts[r], &len);
        }
    }
    MPI_Bcast(b_cuts, w.size + 1, MPI_INT, 0, w.comm);
    hx_rows mine;
    hx_deal(&w, &A, cuts, &mine);
    if (w.rank == 0)
    {
        printf("1D partition and distribution of A used %.2f s\n", get_wtime_sec() - t0);
        int total = 0, *per_rank = (int *) malloc(sizeof(int) * w.size);
        csr_mat_row_part_comm_size(A.m, A.k, A.ptr, A.idx, w.size, cuts, b_cuts, per_rank, &total);
        free(per_rank);
        printf("Total SpMV comm size = %d\n", total);
        fflush(stdout);
    }

    const int b0 = b_cuts[w.rank], bn = b_cuts[w.rank + 1] - b0;
    double *B = dense((size_t) bn, (size_t) n), *C = dense((size_t) mine.count, (size_t) n);
    hx_dense_block(B, n, b0, bn, 0, n);
    rp_call call = {NULL, B, C, n};
    rp_spmm_init(mine.first, mine.count, mine.ptr, mine.idx, mine.val, b_cuts, n, w.comm, &call.eng);
    rp_once(&call);                                  /* warm up */
    rp_spmm_clear_stat(call.eng);
    hx_time_loop(&w, n_test, 1, rp_once, &call);
    rp_spmm_print_stat(call.eng);
    rp_spmm_free(&call.eng);

    if (verify)
    {
        int *cnt = (int *) malloc(sizeof(int) * w.size), *dsp = (int *) malloc(sizeof(int) * w.size);
        for (int r = 0; r < w.size; r++)
        {
            cnt[r] = (mine.cuts[r + 1] - mine.cuts[r]) * n;
            dsp[r] = mine.cuts[r] * n;
        }
        double *whole = w.rank == 0 ? dense((size_t) A.m, (size_t) n) : NULL;
        MPI_Gatherv(C, mine.count * n, MPI_DOUBLE, whole, cnt, dsp, MPI_DOUBLE, 0, w.comm);
        if (w.rank == 0) status = hx_verify(&A, n, whole);
        MPI_Bcast(&status, 1, MPI_INT, 0, w.comm);
        free(cnt); free(dsp); free(whole);
    }
    free(B); free(C); free(cuts); free(b_cuts);
    hx_release(&A, &mine);
    MPI_Finalize();
    return status;
}

/* ---------------------------------------------------------------------------------------------- */
#elif CRP_DRIVER == 2
typedef struct { para2d_spmm_p eng; const double *B; double *C; int ld; } p2d_call;
static void p2d_once(void *p)
{
    p2d_call *c = (p2d_call *) p;
    para2d_spmm_exec(c->eng, 0, c->B, c->ld, c->C, c->ld);
}

/* broadcast an int array that only rank 0 holds so far (allocating it elsewhere) */
static int *share(const hx_world *w, int *arr, int count)
{
    if (w->rank != 0) arr = (int *) malloc(sizeof(int) * count);
    MPI_Bcast(arr, count, MPI_INT, 0, w->comm);
    return arr;
}

int main(int argc, char **argv)
{
    if (argc < 5)
    {
        printf("Usage: %s <mtx-file> <num-of-B-col> <num-of-tests> <part-method> <check-correct>\n", argv[0]);
        printf("<part-method>: 0 for native 1D partition (METIS partitioning is not available in this build)\n");
        printf("<check-correct>: 0 or 1, optional, default value is 0\n");
        return 255;
    }
    const int n = atoi(argv[2]), n_test = atoi(argv[3]);
    int verify = argc > 5 ? atoi(argv[5]) : 0, status = 0;
    hx_world w = hx_start(&argc, &argv);
    if (refuse_metis(&w, atoi(argv[4]))) { MPI_Finalize(); return 254; }
    hx_matrix A;
    hx_load(&w, argv[1], n, &A);
    if (verify) verify = hx_can_verify(&w, &A, n);

    /* plan on rank 0, then everybody gets the grid and the four partition arrays */
    int grid[2] = {0, 0}, *src_rows = NULL, *b_rows = NULL, *c_rows = NULL, *cols = NULL;
    if (w.rank == 0)
    {
        const double t0 = get_wtime_sec();
        int *cuts1d = (int *) malloc(sizeof(int) * (w.size + 1));
        size_t cost = 0;
        csr_mat_row_partition(A.m, A.ptr, w.size, cuts1d);
        calc_spmm_part2d_from_1d(w.size, A.m, n, A.k, cuts1d, A.ptr, A.idx, 1, &grid[0], &grid[1], &cost, &src_rows, &b_rows,
                                 &c_rows, &cols, 0);
        free(cuts1d);
        printf("Rank 0 calculate 2D partitioning time = %.2f s\n", get_wtime_sec() - t0);
        printf("2D process grid: pm, pn = %d, %d\n", grid[0], grid[1]);
    }
    MPI_Bcast(grid, 2, MPI_INT, 0, w.comm);
    const int pm = grid[0], pn = grid[1], gi = w.rank / pn, gj = w.rank % pn;
    src_rows = share(&w, src_rows, w.size + 1);
    b_rows = share(&w, b_rows, pm + 1);
    c_rows = share(&w, c_rows, pm + 1);
    cols = share(&w, cols, pn + 1);

    const double t1 = get_wtime_sec();
    hx_rows mine;
    int *cuts = (int *) malloc(sizeof(int) * (w.size + 1));
    memcpy(cuts, src_rows, sizeof(int) * (w.size + 1));
    hx_deal(&w, &A, cuts, &mine);
    if (w.rank == 0) { printf("1D distribution of A used %.2f s\n", get_wtime_sec() - t1); fflush(stdout); }

    const int br0 = b_rows[gi], brn = b_rows[gi + 1] - br0, cr0 = c_rows[gi], crn = c_rows[gi + 1] - cr0;
    const int c0 = cols[gj], cn = cols[gj + 1] - c0, ld = cn > 0 ? cn : 1;
    double *B = dense((size_t) brn, (size_t) ld), *C = dense((size_t) crn, (size_t) ld);
    hx_dense_block(B, ld, br0, brn, c0, cn);
    p2d_call call = {NULL, B, C, ld};
    for (int pass = 0; pass < 2; pass++)             /* the first init only warms the replication of A */
    {
        if (call.eng) para2d_spmm_free(&call.eng);
        para2d_spmm_init(w.comm, pm, pn, src_rows, b_rows, c_rows, cols, mine.ptr, mine.idx, mine.val, &call.eng);
    }
    p2d_once(&call);
    para2d_spmm_clear_stat(call.eng);
    hx_time_loop(&w, n_test, 1, p2d_once, &call);
    para2d_spmm_print_stat(call.eng);
    para2d_spmm_free(&call.eng);

    if (verify)
    {
        /* C back to rank 0 through the redistribution engine (the reference requests glb_k rows here,
         * examples/test_para2d_spmm.c:186; C has m rows) */
        double *whole = w.rank == 0 ? dense((size_t) A.m, (size_t) n) : NULL;
        mat_redist_engine_p rd = NULL;
        mat_redist_engine_init(cr0, c0, crn, cn, 0, 0, w.rank == 0 ? A.m : 0, w.rank == 0 ? n : 0, w.comm, MPI_DOUBLE,
                               sizeof(double), DEV_TYPE_HOST, &rd, NULL);
        mat_redist_engine_exec(rd, C, ld, whole, n);
        mat_redist_engine_free(&rd);
        if (w.rank == 0) status = hx_verify(&A, n, whole);
        MPI_Bcast(&status, 1, MPI_INT, 0, w.comm);
        free(whole);
    }
    free(B); free(C); free(cuts); free(src_rows); free(b_rows); free(c_rows); free(cols);
    hx_release(&A, &mine);
    MPI_Finalize();
    return status;
}

/* ---------------------------------------------------------------------------------------------- */
#elif CRP_DRIVER == 3
static void print_blocks(const char *title, const int *cuts, int count)
{
    printf("\n%s:\n", title);
    for (int b = 0; b < count; b++) printf("Block %d: [%d, %d]\n", b, cuts[b], cuts[b + 1] - 1);
}

int main(int argc, char **argv)
{
    if (argc < 5)
    {
        printf("Usage: %s <mtx-file> <num-of-B-col> <num-of-processes> <part-method>\n", argv[0]);
        printf("<part-method>: 0 for native 1D partition (METIS partitioning is not available in this build)\n");
        return 255;
    }
    const int n = atoi(argv[2]), P = atoi(argv[3]);
    hx_world w = hx_start(&argc, &argv);             /* serial program; MPI only because the loader reports through it */
    if (refuse_metis(&w, atoi(argv[4]))) { MPI_Finalize(); return 254; }
    hx_matrix A;
    hx_load(&w, argv[1], n, &A);
    printf("============================================================\n");
    int *cuts1d = (int *) malloc(sizeof(int) * (P + 1)), pm = 0, pn = 0;
    int *src_rows = NULL, *b_rows = NULL, *c_rows = NULL, *cols = NULL;
    size_t cost = 0;
    double t0 = get_wtime_sec();
    csr_mat_row_partition(A.m, A.ptr, P, cuts1d);
    const double t_1d = get_wtime_sec() - t0;
    printf("Calculate 1D row partitioning time = %.2f s\n", t_1d);
    t0 = get_wtime_sec();
    calc_spmm_part2d_from_1d(P, A.m, n, A.k, cuts1d, A.ptr, A.idx, 1, &pm, &pn, &cost, &src_rows, &b_rows, &c_rows, &cols, 1);
    const double t_2d = get_wtime_sec() - t0;
    printf("Calculate 2D partitioning from 1D partitioning time = %.2f s\n", t_2d);
    printf("Total partitioning time = %.2f s\n", t_1d + t_2d);
    printf("Calculated 2D grid: pm, pn = %d, %d, comm cost = %zu\n\n", pm, pn, cost);
    printf("1D row partitioning of A:\n");
    for (int r = 0; r < P; r++)
    {
        printf("Rank %3d: [%d, %d]\n", r, src_rows[r], src_rows[r + 1] - 1);
        if (r % pn == pn - 1)
            printf("Ranks [%d, %d] all own A rows [%d, %d] after replicating A\n", r - pn + 1, r, src_rows[r - pn + 1],
                   src_rows[r + 1] - 1);
    }
    print_blocks("1D row partitioning of B", b_rows, pm);
    print_blocks("1D row partitioning of C", c_rows, pm);
    print_blocks("1D column partitioning of B and C", cols, pn);
    printf("\n");
    free(cuts1d); free(src_rows); free(b_rows); free(c_rows); free(cols);
    hx_release(&A, NULL);
    MPI_Finalize();
    return 0;
}

/* ---------------------------------------------------------------------------------------------- */
#elif CRP_DRIVER == 4
typedef struct { crpspmm_engine_p eng; const hx_rows *rows; const double *B; int ldB; double *C; int ldC; } ce_call;
static void ce_once(void *p)
{
    ce_call *c = (ce_call *) p;
    crpspmm_engine_exec(c->eng, c->rows->ptr, c->rows->idx, c->rows->val, c->B, c->ldB, c->C, c->ldC);
}

int main(int argc, char **argv)
{
    if (argc < 4)
    {
        printf("Usage: %s <mtx-file> <num-of-B-col> <num-of-tests> <check-correct> <use-CUDA>\n", argv[0]);
        printf("<check-correct> and <use-CUDA>: 0 or 1, optional, default values are 0\n");
        return 255;
    }
    const int n = atoi(argv[2]), n_test = atoi(argv[3]), on_device = argc > 5 ? atoi(argv[5]) : 0;
    int verify = argc > 4 ? atoi(argv[4]) : 0, status = 0;
    hx_world w = hx_start(&argc, &argv);
    hx_matrix A;
    hx_load(&w, argv[1], n, &A);
    if (verify) verify = hx_can_verify(&w, &A, n);

    const double t0 = get_wtime_sec();
    int *cuts = (int *) malloc(sizeof(int) * (w.size + 1));
    if (w.rank == 0) csr_mat_row_partition(A.m, A.ptr, w.size, cuts);
    hx_rows mine;
    hx_deal(&w, &A, cuts, &mine);
    if (w.rank == 0) { printf("1D partition and distribution of A used %.2f s\n", get_wtime_sec() - t0); fflush(stdout); }

    /* the caller's B and C live on a balanced 2D grid of MPI's choosing; for the check C lands whole on rank 0 */
    int dims[2] = {0, 0}, br0, brn, bc0, bcn, cr0, crn, cc0, ccn;
    MPI_Dims_create(w.size, 2, dims);
    const int gi = w.rank / dims[1], gj = w.rank % dims[1];
    calc_block_spos_size(A.k, dims[0], gi, &br0, &brn);
    calc_block_spos_size(n, dims[1], gj, &bc0, &bcn);
    calc_block_spos_size(A.m, dims[0], gi, &cr0, &crn);
    calc_block_spos_size(n, dims[1], gj, &cc0, &ccn);
    if (verify)
    {
        cr0 = cc0 = 0;
        crn = w.rank == 0 ? A.m : 0;
        ccn = w.rank == 0 ? n : 0;
    }
    double *B = dense((size_t) brn, (size_t) bcn), *C = dense((size_t) crn, (size_t) ccn);
    hx_dense_block(B, bcn, br0, brn, bc0, bcn);

    ce_call call = {NULL, &mine, B, bcn, C, ccn};
    crpspmm_engine_init(A.m, n, A.k, mine.first, mine.count, mine.ptr, mine.idx, br0, brn, bc0, bcn, cr0, crn, cc0, ccn, w.comm,
                        on_device, &call.eng, NULL);
    if (w.rank == 0) { printf("CRP-SpMM 2D partition: %d * %d\n", call.eng->np_row, call.eng->np_col); fflush(stdout); }
    ce_once(&call);
    crpspmm_engine_clear_stat(call.eng);
    hx_time_loop(&w, n_test, 0, ce_once, &call);
    crpspmm_engine_print_stat(call.eng);
    crpspmm_engine_free(&call.eng);

    if (verify == 1 && w.rank == 0) status = hx_verify(&A, n, C);
    MPI_Bcast(&status, 1, MPI_INT, 0, w.comm);
    free(B); free(C); free(cuts);
    hx_release(&A, &mine);
    MPI_Finalize();
    return status;
}
#endif
