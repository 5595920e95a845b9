/* Driver for the older all-in-one engine -- command line, protocol and output lines of
 * /root/reference/deprecated/examples/test_crpspmm.c:5-178 (read -> 1D partition -> scatter A ->
 * B / C on a balanced 2D grid -> init -> warm-up -> <ntest> timed execs -> stats -> check on rank 0),
 * linked against libcrpspmm.so.  <use-CUDA> is accepted for compatibility; the local SpMM always
 * runs on the GPU.  The check uses an independent naive CSR loop and the exit code reports it. */
#include "test_utils.h"
#include "crpspmm.h"
#include "spmat_part.h"

int main(int argc, char **argv)
{
    if (argc < 4)
    {
        printf("Usage: %s <mtx-file> <num-of-B-col> <num-of-tests> <check-correct> <use-CUDA>\n", argv[0]);
        printf("<check-correct> and <use-CUDA>: 0 or 1, optional, default values are 0\n");
        return 255;
    }
    const int glb_n = atoi(argv[2]), n_test = atoi(argv[3]);
    int chk_res = (argc >= 5) ? atoi(argv[4]) : 0;
    const int use_dev = (argc >= 6) ? atoi(argv[5]) : 0;
    int nproc, my_rank, rc = 0;
    MPI_Init(&argc, &argv);
    MPI_Comm_size(MPI_COMM_WORLD, &nproc);
    MPI_Comm_rank(MPI_COMM_WORLD, &my_rank);

    int glb_m = 0, glb_k = 0, *glb_A_rowptr = NULL, *glb_A_colidx = NULL;
    double *glb_A_csrval = NULL;
    if (my_rank == 0) read_mtx_csr(argv[1], 0, &glb_m, &glb_k, glb_n, &glb_A_rowptr, &glb_A_colidx, &glb_A_csrval);
    int mk[2] = {glb_m, glb_k};
    MPI_Bcast(mk, 2, MPI_INT, 0, MPI_COMM_WORLD);
    glb_m = mk[0];
    glb_k = mk[1];
    if (chk_res) chk_res = can_check_res(my_rank, glb_m, glb_n, glb_k);

    /* A: contiguous row blocks with about equal nonzeros */
    double st = get_wtime_sec();
    int *A_m_displs = (int *) malloc(sizeof(int) * (nproc + 1)), *A_nnz_displs = (int *) malloc(sizeof(int) * (nproc + 1));
    int *A_m_scnts = (int *) malloc(sizeof(int) * nproc), *A_nnz_scnts = (int *) malloc(sizeof(int) * nproc);
    if (my_rank == 0)
    {
        csr_mat_row_partition(glb_m, glb_A_rowptr, nproc, A_m_displs);
        for (int i = 0; i <= nproc; i++) A_nnz_displs[i] = glb_A_rowptr[A_m_displs[i]];
    }
    int *loc_A_rowptr = NULL, *loc_A_colidx = NULL;
    double *loc_A_csrval = NULL;
    scatter_csr_rows(MPI_COMM_WORLD, nproc, my_rank, A_m_displs, A_nnz_displs, A_m_scnts, A_nnz_scnts, glb_A_rowptr,
                     glb_A_colidx, glb_A_csrval, &loc_A_rowptr, &loc_A_colidx, &loc_A_csrval);
    const int loc_A_srow = A_m_displs[my_rank], loc_A_nrow = A_m_displs[my_rank + 1] - loc_A_srow;
    if (my_rank == 0)
    {
        printf("1D partition and distribution of A used %.2f s\n", get_wtime_sec() - st);
        fflush(stdout);
    }

    /* B and C: blocks of a balanced 2D grid chosen by MPI; with the check on, C lands whole on rank 0 */
    int dims[2] = {0, 0};
    MPI_Dims_create(nproc, 2, dims);
    const int grid_r = dims[0], grid_c = dims[1], my_r = my_rank / grid_c, my_c = my_rank % grid_c;
    int B_srow, B_nrow, B_scol, B_ncol, C_srow, C_nrow, C_scol, C_ncol;
    calc_block_spos_size(glb_k, grid_r, my_r, &B_srow, &B_nrow);
    calc_block_spos_size(glb_n, grid_c, my_c, &B_scol, &B_ncol);
    calc_block_spos_size(glb_m, grid_r, my_r, &C_srow, &C_nrow);
    calc_block_spos_size(glb_n, grid_c, my_c, &C_scol, &C_ncol);
    if (chk_res > 0)
    {
        C_srow = C_scol = 0;
        C_nrow = (my_rank == 0) ? glb_m : 0;
        C_ncol = (my_rank == 0) ? glb_n : 0;
    }
    double *loc_B = (double *) malloc(sizeof(double) * ((size_t) B_nrow * B_ncol + 1));
    double *loc_C = (double *) malloc(sizeof(double) * ((size_t) C_nrow * C_ncol + 1));
    const double factor_i = 0.19, factor_j = 0.24;
    fill_B(0, loc_B, B_ncol, B_srow, B_nrow, B_scol, B_ncol, factor_i, factor_j);

    crpspmm_engine_p eng = NULL;
    crpspmm_engine_init(glb_m, glb_n, glb_k, loc_A_srow, loc_A_nrow, loc_A_rowptr, loc_A_colidx, B_srow, B_nrow, B_scol,
                        B_ncol, C_srow, C_nrow, C_scol, C_ncol, MPI_COMM_WORLD, use_dev, &eng, NULL);
    if (my_rank == 0)
    {
        printf("CRP-SpMM 2D partition: %d * %d\n", eng->np_row, eng->np_col);
        fflush(stdout);
    }
    crpspmm_engine_exec(eng, loc_A_rowptr, loc_A_colidx, loc_A_csrval, loc_B, B_ncol, loc_C, C_ncol);   /* warm up */
    crpspmm_engine_clear_stat(eng);
    for (int i = 0; i < n_test; i++)
    {
        const double t0 = MPI_Wtime();
        crpspmm_engine_exec(eng, loc_A_rowptr, loc_A_colidx, loc_A_csrval, loc_B, B_ncol, loc_C, C_ncol);
        const double t1 = MPI_Wtime();
        if (my_rank == 0) { printf("%.2f\n", t1 - t0); fflush(stdout); }
    }
    crpspmm_engine_print_stat(eng);
    crpspmm_engine_free(&eng);

    if (chk_res == 1 && my_rank == 0)
    {
        double *glb_B = (double *) malloc(sizeof(double) * (size_t) glb_k * glb_n);
        double *ref_C = (double *) malloc(sizeof(double) * (size_t) glb_m * glb_n);
        fill_B(0, glb_B, glb_n, 0, glb_k, 0, glb_n, factor_i, factor_j);
        naive_csr_spmm(glb_m, glb_n, glb_A_rowptr, glb_A_colidx, glb_A_csrval, glb_B, glb_n, ref_C, glb_n);
        double C_fnorm, err_fnorm;
        calc_err_2norm(glb_m * glb_n, ref_C, loc_C, &C_fnorm, &err_fnorm);
        printf("||C_ref - C||_f / ||C_ref||_f = %e\n", err_fnorm / C_fnorm);
        fflush(stdout);
        if (!(err_fnorm / C_fnorm <= 1e-12)) rc = 1;
        free(glb_B);
        free(ref_C);
    }
    MPI_Bcast(&rc, 1, MPI_INT, 0, MPI_COMM_WORLD);
    free(glb_A_rowptr); free(glb_A_colidx); free(glb_A_csrval); free(A_m_displs); free(A_m_scnts); free(A_nnz_scnts);
    free(A_nnz_displs); free(loc_A_rowptr); free(loc_A_colidx); free(loc_A_csrval); free(loc_B); free(loc_C);
    MPI_Finalize();
    return rc;
}
