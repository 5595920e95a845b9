/*
 * rowpara_spmm.h -- MPI-typed facade of the 1D row-parallel SpMM engine: the
 * reference's public API, unchanged (/root/reference/src/rowpara_spmm.h:8-87),
 * so that callers written against the reference (examples/test_rp_spmm.c:124-145)
 * compile and link against lib/libcrpspmm.so.  Implemented in
 * crp-spmm_amd/csrc/mpi_facade.cpp on top of crp_engine.h; the struct keeps the
 * reference's field names (rp_spmm->rB_nrow etc. are documented as user-visible,
 * src/rowpara_spmm.h:73-74) and appends one opaque pointer for the device state.
 */
#ifndef CRP_ROWPARA_SPMM_H
#define CRP_ROWPARA_SPMM_H

#include <stddef.h>
#include <stdlib.h>
#include <mpi.h>

struct rowpara_spmm
{
    int    nproc, my_rank;      /* size / rank in comm                               */
    int    glb_n;               /* columns of B and C                                */
    int    A_nrow;              /* rows of local A                                   */
    int    rB_nrow;             /* rows of the redistributed B (compact id space)    */
    int    rB_self_src_offset;  /* first locally served row: offset in local B       */
    int    rB_self_dst_offset;  /* ... and its compact id                            */
    int    rB_self_nrow;        /* rows served from the local B block                */
    int    rB_p2p;              /* RP_SPMM_P2P                                       */
    int    rB_reidx;            /* RP_SPMM_REIDX                                     */
    int    *A_rowptr;           /* A_nrow + 1, rebased                               */
    int    *A_colidx;           /* compact column ids                                */
    int    *rB_self_src_ridxs;  /* rB_self_nrow                                      */
    int    *rB_scnts;           /* nproc, elements                                   */
    int    *rB_sridxs;          /* local B rows to send                              */
    int    *rB_sdispls;         /* nproc + 1, elements                               */
    int    *rB_rcnts;           /* nproc, elements                                   */
    int    *rB_rridxs;          /* compact ids of received rows                      */
    int    *rB_rdispls;         /* nproc + 1, elements                               */
    double *A_val;
    MPI_Comm comm;

    size_t rB_recv_size;        /* rows received per multiply                        */
    int    n_exec;
    double t_init, t_pack, t_a2a, t_unpack, t_spmm, t_exec;

    void   *impl;               /* crp_rp_spmm_p + communicator glue (not in the reference) */
};
typedef struct rowpara_spmm  rp_spmm_s;
typedef struct rowpara_spmm *rp_spmm_p;

#ifdef __cplusplus
extern "C" {
#endif

/* Arguments exactly as src/rowpara_spmm.h:49-64. B and C passed to
 * rp_spmm_exec may be host or device (hipMalloc) pointers. */
void rp_spmm_init(
    const int A_srow, const int A_nrow, const int *A_rowptr, const int *A_colidx,
    const double *A_val, const int *B_row_displs, const int glb_n, MPI_Comm comm,
    rp_spmm_p *rp_spmm
);
void rp_spmm_free(rp_spmm_p *rp_spmm);
void rp_spmm_exec(
    rp_spmm_p rp_spmm, const int BC_layout, const double *B, const int ldB,
    double *C, const int ldC
);
void rp_spmm_print_stat(rp_spmm_p rp_spmm);
void rp_spmm_clear_stat(rp_spmm_p rp_spmm);

#ifdef __cplusplus
}
#endif
#endif
