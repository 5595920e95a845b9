/* Driver for the 2D engine -- same command line, protocol and output lines as
 * /root/reference/examples/test_para2d_spmm.c:7-239 (read -> 1D partition -> 2D plan on rank 0 ->
 * broadcast plan -> scatter A0 -> fill the 2D block of B -> init (twice: the first warms the A
 * replication) -> warm-up -> timed execs -> stats -> mat_redist C to rank 0 -> check). */
#include "test_utils.h"
#include "mat_redist.h"
#include "para2d_spmm.h"
#include "spmat_part.h"

int main(int argc, char **argv)
{
    if (argc < 5)
    {
        printf("Usage: %s <mtx-file> <num-of-B-col> <num-of-tests> <part-method> <check-correct>\n", argv[0]);
        printf("<part-method>: 0 for native 1D partition (METIS partitioning is not available in this build)\n");
        printf("<check-correct>: 0 or 1, optional, default value is 0\n");
        return 255;
    }
    int glb_n = atoi(argv[2]), n_test = atoi(argv[3]), method = atoi(argv[4]);
    int chk_res = (argc >= 6) ? atoi(argv[5]) : 0;
    int nproc, my_rank, rc = 0;
    MPI_Init(&argc, &argv);
    MPI_Comm_size(MPI_COMM_WORLD, &nproc);
    MPI_Comm_rank(MPI_COMM_WORLD, &my_rank);
    if (method != 0)
    {
        if (my_rank == 0) printf("METIS 1D row partitioning is not available in this build (part-method must be 0)\n");
        MPI_Finalize();
        return 254;
    }
    double st, et;
    int glb_m = 0, glb_k = 0, *glb_A_rowptr = NULL, *glb_A_colidx = NULL;
    double *glb_A_csrval = NULL;
    if (my_rank == 0) read_mtx_csr(argv[1], 0, &glb_m, &glb_k, glb_n, &glb_A_rowptr, &glb_A_colidx, &glb_A_csrval);
    int mk[2] = {glb_m, glb_k};
    MPI_Bcast(mk, 2, MPI_INT, 0, MPI_COMM_WORLD);
    glb_m = mk[0];
    glb_k = mk[1];
    if (chk_res) chk_res = can_check_res(my_rank, glb_m, glb_n, glb_k);

    int pm = 0, pn = 0;
    size_t comm_cost = 0;
    int *A_rb_displs = (int *) malloc(sizeof(int) * (nproc + 1));
    int *A0_rowptr = NULL, *B_rowptr = NULL, *AC_rowptr = NULL, *BC_colptr = NULL;
    if (my_rank == 0)
    {
        st = get_wtime_sec();
        csr_mat_row_partition(glb_m, glb_A_rowptr, nproc, A_rb_displs);
        calc_spmm_part2d_from_1d(nproc, glb_m, glb_n, glb_k, A_rb_displs, glb_A_rowptr, glb_A_colidx, 1, &pm, &pn,
                                 &comm_cost, &A0_rowptr, &B_rowptr, &AC_rowptr, &BC_colptr, 0);
        et = get_wtime_sec();
        printf("Rank 0 calculate 2D partitioning time = %.2f s\n", et - st);
        printf("2D process grid: pm, pn = %d, %d\n", pm, pn);
    }
    int pmn[2] = {pm, pn};
    MPI_Bcast(pmn, 2, MPI_INT, 0, MPI_COMM_WORLD);
    pm = pmn[0];
    pn = pmn[1];
    if (my_rank != 0)
    {
        A0_rowptr = (int *) malloc(sizeof(int) * (nproc + 1));
        B_rowptr = (int *) malloc(sizeof(int) * (pm + 1));
        AC_rowptr = (int *) malloc(sizeof(int) * (pm + 1));
        BC_colptr = (int *) malloc(sizeof(int) * (pn + 1));
    }
    MPI_Bcast(A0_rowptr, nproc + 1, MPI_INT, 0, MPI_COMM_WORLD);
    MPI_Bcast(B_rowptr, pm + 1, MPI_INT, 0, MPI_COMM_WORLD);
    MPI_Bcast(AC_rowptr, pm + 1, MPI_INT, 0, MPI_COMM_WORLD);
    MPI_Bcast(BC_colptr, pn + 1, MPI_INT, 0, MPI_COMM_WORLD);
    const int pi = my_rank / pn, pj = my_rank % pn;

    st = get_wtime_sec();
    int *A_m_displs = (int *) malloc(sizeof(int) * (nproc + 1)), *A_nnz_displs = (int *) malloc(sizeof(int) * (nproc + 1));
    int *A_m_scnts = (int *) malloc(sizeof(int) * nproc), *A_nnz_scnts = (int *) malloc(sizeof(int) * nproc);
    memcpy(A_m_displs, A0_rowptr, sizeof(int) * (nproc + 1));
    if (my_rank == 0) for (int i = 0; i <= nproc; i++) A_nnz_displs[i] = glb_A_rowptr[A0_rowptr[i]];
    int *loc_A_rowptr = NULL, *loc_A_colidx = NULL;
    double *loc_A_csrval = NULL;
    scatter_csr_rows(MPI_COMM_WORLD, nproc, my_rank, A_m_displs, A_nnz_displs, A_m_scnts, A_nnz_scnts, glb_A_rowptr,
                     glb_A_colidx, glb_A_csrval, &loc_A_rowptr, &loc_A_colidx, &loc_A_csrval);
    et = get_wtime_sec();
    if (my_rank == 0) { printf("1D distribution of A used %.2f s\n", et - st); fflush(stdout); }

    const int loc_B_srow = B_rowptr[pi], loc_B_nrow = B_rowptr[pi + 1] - loc_B_srow;
    const int loc_C_srow = AC_rowptr[pi], loc_C_nrow = AC_rowptr[pi + 1] - loc_C_srow;
    const int loc_BC_scol = BC_colptr[pj], loc_BC_ncol = BC_colptr[pj + 1] - loc_BC_scol;
    const int ld = loc_BC_ncol > 0 ? loc_BC_ncol : 1;
    double *loc_B = (double *) malloc(sizeof(double) * (size_t) (loc_B_nrow > 0 ? loc_B_nrow : 1) * ld);
    double *loc_C = (double *) malloc(sizeof(double) * (size_t) (loc_C_nrow > 0 ? loc_C_nrow : 1) * ld);
    const int layout = 0;
    const double factor_i = 0.19, factor_j = 0.24;
    fill_B(layout, loc_B, ld, loc_B_srow, loc_B_nrow, loc_BC_scol, loc_BC_ncol, factor_i, factor_j);

    para2d_spmm_p para2d_spmm = NULL;
    para2d_spmm_init(MPI_COMM_WORLD, pm, pn, A0_rowptr, B_rowptr, AC_rowptr, BC_colptr, loc_A_rowptr, loc_A_colidx, loc_A_csrval, &para2d_spmm);
    para2d_spmm_free(&para2d_spmm);     /* the first init only warms up the replication of A */
    para2d_spmm_init(MPI_COMM_WORLD, pm, pn, A0_rowptr, B_rowptr, AC_rowptr, BC_colptr, loc_A_rowptr, loc_A_colidx, loc_A_csrval, &para2d_spmm);
    para2d_spmm_exec(para2d_spmm, layout, loc_B, ld, loc_C, ld);
    para2d_spmm_clear_stat(para2d_spmm);
    for (int i = 0; i < n_test; i++)
    {
        MPI_Barrier(MPI_COMM_WORLD);
        st = get_wtime_sec();
        para2d_spmm_exec(para2d_spmm, layout, loc_B, ld, loc_C, ld);
        MPI_Barrier(MPI_COMM_WORLD);
        et = get_wtime_sec();
        if (my_rank == 0) { printf("%.2f\n", et - st); fflush(stdout); }
    }
    para2d_spmm_print_stat(para2d_spmm);
    para2d_spmm_free(&para2d_spmm);

    if (chk_res)
    {
        double *glb_B = NULL, *ref_C = NULL, *recv_C = NULL;
        int req_nrow = 0, req_ncol = 0;
        if (my_rank == 0)
        {
            req_nrow = glb_m;            /* the reference asks for glb_k rows here (test_para2d_spmm.c:186) */
            req_ncol = glb_n;
            glb_B = (double *) malloc(sizeof(double) * (size_t) glb_k * glb_n);
            ref_C = (double *) malloc(sizeof(double) * (size_t) glb_m * glb_n);
            recv_C = (double *) malloc(sizeof(double) * (size_t) glb_m * glb_n);
            fill_B(0, glb_B, glb_n, 0, glb_k, 0, glb_n, factor_i, factor_j);
        }
        mat_redist_engine_p rd_C = NULL;
        mat_redist_engine_init(loc_C_srow, loc_BC_scol, loc_C_nrow, loc_BC_ncol, 0, 0, req_nrow, req_ncol, MPI_COMM_WORLD,
                               MPI_DOUBLE, sizeof(double), DEV_TYPE_HOST, &rd_C, NULL);
        mat_redist_engine_exec(rd_C, loc_C, ld, recv_C, glb_n);
        mat_redist_engine_free(&rd_C);
        if (my_rank == 0)
        {
            naive_csr_spmm(glb_m, glb_n, glb_A_rowptr, glb_A_colidx, glb_A_csrval, glb_B, glb_n, ref_C, glb_n);
            double C_fnorm, err_fnorm;
            calc_err_2norm(glb_m * glb_n, ref_C, recv_C, &C_fnorm, &err_fnorm);
            printf("||C_ref - C||_f / ||C_ref||_f = %e\n", err_fnorm / C_fnorm);
            fflush(stdout);
            if (!(err_fnorm / C_fnorm <= 1e-12)) rc = 1;
        }
        MPI_Bcast(&rc, 1, MPI_INT, 0, MPI_COMM_WORLD);
        free(glb_B); free(ref_C); free(recv_C);
    }
    free(glb_A_rowptr); free(glb_A_colidx); free(glb_A_csrval); free(A_rb_displs); free(A0_rowptr); free(B_rowptr);
    free(AC_rowptr); free(BC_colptr); free(A_m_displs); free(A_nnz_displs); free(A_m_scnts); free(A_nnz_scnts);
    free(loc_A_rowptr); free(loc_A_colidx); free(loc_A_csrval); free(loc_B); free(loc_C);
    MPI_Finalize();
    return rc;
}
