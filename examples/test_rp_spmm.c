/* Driver for the 1D row-parallel engine -- same command line, protocol and output lines as
 * /root/reference/examples/test_rp_spmm.c:7-219 (read -> 1D partition -> scatter -> fill B ->
 * init -> warm-up -> <ntest> timed execs -> stats -> gather C -> check), linked against
 * libcrpspmm.so.  The check uses an independent naive CSR loop and the exit code reports it. */
#include "test_utils.h"
#include "rowpara_spmm.h"
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

    st = get_wtime_sec();
    int *A_m_displs = (int *) malloc(sizeof(int) * (nproc + 1)), *A_nnz_displs = (int *) malloc(sizeof(int) * (nproc + 1));
    int *x_displs = (int *) malloc(sizeof(int) * (nproc + 1));
    int *A_m_scnts = (int *) malloc(sizeof(int) * nproc), *A_nnz_scnts = (int *) malloc(sizeof(int) * nproc);
    if (my_rank == 0)
    {
        printf("Using naive 1D row partitioning\n");
        csr_mat_row_partition(glb_m, glb_A_rowptr, nproc, A_m_displs);
        for (int i = 0; i <= nproc; i++) A_nnz_displs[i] = glb_A_rowptr[A_m_displs[i]];
        if (glb_m == glb_k) memcpy(x_displs, A_m_displs, sizeof(int) * (nproc + 1));
        else
        {
            int tmp;
            for (int i = 0; i <= nproc; i++) calc_block_spos_size(glb_k, nproc, i, x_displs + i, &tmp);
        }
    }
    int *loc_A_rowptr = NULL, *loc_A_colidx = NULL;
    double *loc_A_csrval = NULL;
    MPI_Bcast(x_displs, nproc + 1, MPI_INT, 0, MPI_COMM_WORLD);
    scatter_csr_rows(MPI_COMM_WORLD, nproc, my_rank, A_m_displs, A_nnz_displs, A_m_scnts, A_nnz_scnts, glb_A_rowptr,
                     glb_A_colidx, glb_A_csrval, &loc_A_rowptr, &loc_A_colidx, &loc_A_csrval);
    int loc_A_srow = A_m_displs[my_rank], loc_A_nrow = A_m_displs[my_rank + 1] - loc_A_srow;
    et = get_wtime_sec();
    if (my_rank == 0)
    {
        printf("1D partition and distribution of A used %.2f s\n", et - st);
        int total_size = 0, *comm_sizes = (int *) malloc(sizeof(int) * nproc);
        csr_mat_row_part_comm_size(glb_m, glb_k, glb_A_rowptr, glb_A_colidx, nproc, A_m_displs, x_displs, comm_sizes, &total_size);
        free(comm_sizes);
        printf("Total SpMV comm size = %d\n", total_size);
        fflush(stdout);
    }

    int loc_B_srow = x_displs[my_rank], loc_B_nrow = x_displs[my_rank + 1] - loc_B_srow, loc_C_nrow = loc_A_nrow;
    double *loc_B = (double *) malloc(sizeof(double) * (size_t) (loc_B_nrow > 0 ? loc_B_nrow : 1) * glb_n);
    double *loc_C = (double *) malloc(sizeof(double) * (size_t) (loc_C_nrow > 0 ? loc_C_nrow : 1) * glb_n);
    const int layout = 0;
    const double factor_i = 0.19, factor_j = 0.24;
    fill_B(layout, loc_B, glb_n, loc_B_srow, loc_B_nrow, 0, glb_n, factor_i, factor_j);

    rp_spmm_p rp_spmm = NULL;
    rp_spmm_init(loc_A_srow, loc_A_nrow, loc_A_rowptr, loc_A_colidx, loc_A_csrval, x_displs, glb_n, MPI_COMM_WORLD, &rp_spmm);
    rp_spmm_exec(rp_spmm, layout, loc_B, glb_n, loc_C, glb_n);   /* warm up */
    rp_spmm_clear_stat(rp_spmm);
    for (int i = 0; i < n_test; i++)
    {
        MPI_Barrier(MPI_COMM_WORLD);
        st = get_wtime_sec();
        rp_spmm_exec(rp_spmm, layout, loc_B, glb_n, loc_C, glb_n);
        MPI_Barrier(MPI_COMM_WORLD);
        et = get_wtime_sec();
        if (my_rank == 0) { printf("%.2f\n", et - st); fflush(stdout); }
    }
    rp_spmm_print_stat(rp_spmm);
    rp_spmm_free(&rp_spmm);

    if (chk_res)
    {
        double *glb_B = NULL, *ref_C = NULL, *recv_C = NULL;
        int *C_rcnts = (int *) malloc(sizeof(int) * nproc), *C_rdispls = (int *) malloc(sizeof(int) * (nproc + 1));
        C_rdispls[0] = 0;
        for (int i = 0; i < nproc; i++)
        {
            C_rcnts[i] = A_m_scnts[i] * glb_n;
            C_rdispls[i + 1] = C_rdispls[i] + C_rcnts[i];
        }
        if (my_rank == 0)
        {
            glb_B = (double *) malloc(sizeof(double) * (size_t) glb_k * glb_n);
            ref_C = (double *) malloc(sizeof(double) * (size_t) glb_m * glb_n);
            recv_C = (double *) malloc(sizeof(double) * (size_t) glb_m * glb_n);
            fill_B(0, glb_B, glb_n, 0, glb_k, 0, glb_n, factor_i, factor_j);
        }
        MPI_Gatherv(loc_C, loc_C_nrow * glb_n, MPI_DOUBLE, recv_C, C_rcnts, C_rdispls, MPI_DOUBLE, 0, MPI_COMM_WORLD);
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
        free(glb_B); free(ref_C); free(recv_C); free(C_rcnts); free(C_rdispls);
    }
    free(glb_A_rowptr); free(glb_A_colidx); free(glb_A_csrval); free(A_m_displs); free(A_m_scnts);
    free(A_nnz_scnts); free(A_nnz_displs); free(x_displs); free(loc_A_rowptr); free(loc_A_colidx);
    free(loc_A_csrval); free(loc_B); free(loc_C);
    MPI_Finalize();
    return rc;
}
