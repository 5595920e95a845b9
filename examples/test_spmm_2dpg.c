/* Serial planner dump -- same command line and output as
 * /root/reference/examples/test_spmm_2dpg.c:5-90 (no MPI launch needed). */
#include "test_utils.h"
#include "spmat_part.h"

int main(int argc, char **argv)
{
    if (argc < 5)
    {
        printf("Usage: %s <mtx-file> <num-of-B-col> <num-of-processes> <part-method>\n", argv[0]);
        printf("<part-method>: 0 for native 1D partition (METIS partitioning is not available in this build)\n");
        return 255;
    }
    int n = atoi(argv[2]), nproc = atoi(argv[3]), method = atoi(argv[4]);
    if (method != 0) { printf("METIS 1D row partitioning is not available in this build\n"); return 254; }
    int m, k, *rowptr = NULL, *colidx = NULL;
    double *val = NULL;
    MPI_Init(&argc, &argv);    /* read_mtx_csr aborts through MPI on a bad file */
    read_mtx_csr(argv[1], 0, &m, &k, n, &rowptr, &colidx, &val);
    int pm = 0, pn = 0;
    size_t comm_cost = 0;
    int *A_rb_displs = (int *) malloc(sizeof(int) * (nproc + 1));
    int *A0_rowptr = NULL, *B_rowptr = NULL, *AC_rowptr = NULL, *BC_colptr = NULL;
    printf("============================================================\n");
    double st = get_wtime_sec();
    csr_mat_row_partition(m, rowptr, nproc, A_rb_displs);
    double t1 = get_wtime_sec() - st;
    printf("Calculate 1D row partitioning time = %.2f s\n", t1);
    st = get_wtime_sec();
    calc_spmm_part2d_from_1d(nproc, m, n, k, A_rb_displs, rowptr, colidx, 1, &pm, &pn, &comm_cost, &A0_rowptr, &B_rowptr,
                             &AC_rowptr, &BC_colptr, 1);
    double t2 = get_wtime_sec() - st;
    printf("Calculate 2D partitioning from 1D partitioning time = %.2f s\n", t2);
    printf("Total partitioning time = %.2f s\n", t1 + t2);
    printf("Calculated 2D grid: pm, pn = %d, %d, comm cost = %zu\n\n", pm, pn, comm_cost);
    printf("1D row partitioning of A:\n");
    for (int i = 0; i < pm; i++)
    {
        for (int j = 0; j < pn; j++)
        {
            int rank = i * pn + j;
            printf("Rank %3d: [%d, %d]\n", rank, A0_rowptr[rank], A0_rowptr[rank + 1] - 1);
        }
        int rs = i * pn, re = (i + 1) * pn - 1;
        printf("Ranks [%d, %d] all own A rows [%d, %d] after replicating A\n", rs, re, A0_rowptr[rs], A0_rowptr[re + 1] - 1);
    }
    printf("\n1D row partitioning of B:\n");
    for (int i = 0; i < pm; i++) printf("Block %d: [%d, %d]\n", i, B_rowptr[i], B_rowptr[i + 1] - 1);
    printf("\n1D row partitioning of C:\n");
    for (int i = 0; i < pm; i++) printf("Block %d: [%d, %d]\n", i, AC_rowptr[i], AC_rowptr[i + 1] - 1);
    printf("\n1D column partitioning of B and C:\n");
    for (int i = 0; i < pn; i++) printf("Block %d: [%d, %d]\n", i, BC_colptr[i], BC_colptr[i + 1] - 1);
    printf("\n");
    free(val); free(rowptr); free(colidx); free(A_rb_displs); free(A0_rowptr); free(B_rowptr); free(AC_rowptr); free(BC_colptr);
    MPI_Finalize();
    return 0;
}
