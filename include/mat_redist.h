/*
 * mat_redist.h -- MPI-typed facade of the generic dense 2D-block redistribution engine: the
 * reference's public API, unchanged (/root/reference/src/mat_redist.h:7-100; used by
 * examples/test_para2d_spmm.c:193-200 to gather C and by the deprecated engine for A/B/C).
 * Implemented in crp-spmm_amd/csrc/mpi_facade.cpp on crp_mat_redist_* (crp_engine.h).
 * The struct keeps the reference's field names; graph_comm is MPI_COMM_NULL (the exchange
 * is an all-to-all with empty non-neighbour slots instead of a dist-graph communicator).
 */
#ifndef CRP_MAT_REDIST_H
#define CRP_MAT_REDIST_H

#include <mpi.h>
#include "dev_type.h"

struct mat_redist_engine
{
    MPI_Comm graph_comm;
    MPI_Datatype dtype;
    size_t  dt_size;
    int     nproc, rank;
    int     src_srow, src_scol, src_nrow, src_ncol;
    int     req_srow, req_scol, req_nrow, req_ncol;
    int     n_proc_send, n_proc_recv;
    int     send_cnt, recv_cnt;
    int     alloc_workbuf;
    int     *send_ranks, *send_sizes, *send_displs, *sblk_sizes;
    int     *recv_ranks, *recv_sizes, *recv_displs, *rblk_sizes;
    int     *send_info0, *recv_info0;
    void    *sendbuf_h, *recvbuf_h, *sendbuf_d, *recvbuf_d;
    void    *workbuf_h, *workbuf_d;
    double  hd_trans_ms;
    dev_type_t dev_type;
    void    *impl;              /* crp_mat_redist_p + communicator glue (not in the reference) */
};
typedef struct mat_redist_engine  mat_redist_engine_s;
typedef struct mat_redist_engine *mat_redist_engine_p;

#ifdef __cplusplus
extern "C" {
#endif

/* Arguments as src/mat_redist.h:53-74.  On an invalid dev_type prints "[ERROR] ... Invalid
 * device type" and returns with *engine_ untouched. */
void mat_redist_engine_init(
    const int src_srow, const int src_scol, const int src_nrow, const int src_ncol,
    const int req_srow, const int req_scol, const int req_nrow, const int req_ncol,
    MPI_Comm comm, MPI_Datatype dtype, const size_t dt_size, dev_type_t dev_type,
    mat_redist_engine_p *engine_, size_t *workbuf_bytes
);
void mat_redist_engine_attach_workbuf(mat_redist_engine_p engine, void *workbuf_h, void *workbuf_d);
void mat_redist_engine_exec(
    mat_redist_engine_p engine, const void *src_blk, const int src_ld,
    void *dst_blk, const int dst_ld
);
void mat_redist_engine_free(mat_redist_engine_p *engine_);

#ifdef __cplusplus
}
#endif

/* src/dev_type.h:63-88 */
#define MALLOC_ATTACH_WORKBUF(attach_func, free_func, engine, dev_type, workbuf_bytes, workbuf_h, workbuf_d) \
    do {                                                                                \
        workbuf_h = NULL;                                                               \
        workbuf_d = NULL;                                                               \
        if ((dev_type == DEV_TYPE_HOST) || (dev_type == DEV_TYPE_HIP))                  \
        {                                                                               \
            workbuf_h = dev_type_malloc(workbuf_bytes, DEV_TYPE_HOST);                  \
            if (workbuf_h == NULL) { ERROR_PRINTF("Allocate host workbuf failed\n"); free_func(&engine); break; } \
        }                                                                               \
        if ((dev_type == DEV_TYPE_HIP) || (dev_type == DEV_TYPE_HIP_RCCL))              \
        {                                                                               \
            workbuf_d = dev_type_malloc(workbuf_bytes, DEV_TYPE_HIP);                   \
            if (workbuf_d == NULL) { ERROR_PRINTF("Allocate device workbuf failed\n"); free_func(&engine); break; } \
        }                                                                               \
        attach_func(engine, workbuf_h, workbuf_d);                                      \
    } while (0)

#endif
