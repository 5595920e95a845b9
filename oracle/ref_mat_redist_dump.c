/* TEST INFRASTRUCTURE.  Driver (our code) around the REFERENCE's mat_redist engine
 * (/root/reference/src/mat_redist.c, dev_type.c, utils.c compiled unmodified, see
 * oracle/Makefile target ref_mpi).  Reads rectangles from a scenario file, runs
 * mat_redist_engine_init / exec on MPI_COMM_WORLD with DEV_TYPE_HOST and prints the plan
 * fields and the redistributed block of every rank.  Used once, in the build container, by
 * tests/golden/make_golden_redist.py to produce tests/golden/mat_redist_P*.json.
 *
 * scenario file: first line S (number of scenarios), P (ranks), M, N (global matrix);
 * then S * P lines of 8 ints: src_srow src_scol src_nrow src_ncol req_srow req_scol req_nrow req_ncol.
 * global matrix element (i, j) = i * 4096 + j (exact in fp64). */
#include <stdio.h>
#include <stdlib.h>
#include <mpi.h>
#include "mat_redist.h"

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int P, me;
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
    FILE *f = fopen(argv[1], "r");
    int S, Pf, M, N;
    if (fscanf(f, "%d %d %d %d", &S, &Pf, &M, &N) != 4 || Pf != P) { fprintf(stderr, "bad scenario file\n"); MPI_Abort(MPI_COMM_WORLD, 1); }
    int *rect = (int *) malloc(sizeof(int) * 8 * S * P);
    for (int i = 0; i < 8 * S * P; i++) if (fscanf(f, "%d", &rect[i]) != 1) MPI_Abort(MPI_COMM_WORLD, 2);
    fclose(f);
    for (int s = 0; s < S; s++)
    {
        int *r = rect + 8 * (s * P + me);
        mat_redist_engine_p e = NULL;
        mat_redist_engine_init(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], MPI_COMM_WORLD, MPI_DOUBLE,
                               sizeof(double), DEV_TYPE_HOST, &e, NULL);
        int src_ld = r[3] > 0 ? r[3] + 1 : 1, dst_ld = r[7] > 0 ? r[7] + 2 : 1;
        double *src = (double *) malloc(sizeof(double) * (size_t) (r[2] > 0 ? r[2] : 1) * src_ld);
        double *dst = (double *) malloc(sizeof(double) * (size_t) (r[6] > 0 ? r[6] : 1) * dst_ld);
        for (int i = 0; i < r[2]; i++)
            for (int j = 0; j < r[3]; j++) src[i * src_ld + j] = (double) (r[0] + i) * 4096.0 + (r[1] + j);
        for (int i = 0; i < (r[6] > 0 ? r[6] : 1) * dst_ld; i++) dst[i] = -1.0;
        mat_redist_engine_exec(e, src, src_ld, dst, dst_ld);
        for (int q = 0; q < P; q++)
        {
            if (q == me)
            {
                printf("S %d R %d nsend %d nrecv %d send_cnt %d recv_cnt %d\n", s, me, e->n_proc_send, e->n_proc_recv, e->send_cnt, e->recv_cnt);
                printf("send_ranks"); for (int i = 0; i < e->n_proc_send; i++) printf(" %d", e->send_ranks[i]); printf("\n");
                printf("send_sizes"); for (int i = 0; i < e->n_proc_send; i++) printf(" %d", e->send_sizes[i]); printf("\n");
                printf("send_displs"); for (int i = 0; i <= e->n_proc_send; i++) printf(" %d", e->send_displs[i]); printf("\n");
                printf("sblk_sizes"); for (int i = 0; i < 4 * e->n_proc_send; i++) printf(" %d", e->sblk_sizes[i]); printf("\n");
                printf("recv_ranks"); for (int i = 0; i < e->n_proc_recv; i++) printf(" %d", e->recv_ranks[i]); printf("\n");
                printf("recv_sizes"); for (int i = 0; i < e->n_proc_recv; i++) printf(" %d", e->recv_sizes[i]); printf("\n");
                printf("recv_displs"); for (int i = 0; i <= e->n_proc_recv; i++) printf(" %d", e->recv_displs[i]); printf("\n");
                printf("rblk_sizes"); for (int i = 0; i < 4 * e->n_proc_recv; i++) printf(" %d", e->rblk_sizes[i]); printf("\n");
                printf("dst");
                for (int i = 0; i < r[6]; i++) for (int j = 0; j < r[7]; j++) printf(" %.0f", dst[i * dst_ld + j]);
                printf("\n");
                fflush(stdout);
            }
            MPI_Barrier(MPI_COMM_WORLD);
        }
        mat_redist_engine_free(&e);
        free(src);
        free(dst);
    }
    MPI_Finalize();
    return 0;
}
