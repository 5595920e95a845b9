/*
 * crpspmm.h -- MPI-typed facade of the older all-in-one engine (C := A * B with A in any 1D row
 * distribution, B and C in arbitrary 2D blocks, everything on the host, A's values handed over
 * on every call).  Public API of /root/reference/deprecated/src/crpspmm.h:8-130, implemented in
 * crp-spmm_amd/csrc/mpi_facade.cpp on crp_crpspmm_* (crp_engine.h), i.e. on the device engine
 * of the live path.  What differs from the reference, by design:
 *   - use_CUDA is accepted and ignored: the local SpMM always runs on the GPU (there is no CPU
 *     kernel in this library); inputs and outputs stay host arrays as the reference documents.
 *   - the engine owns its buffers.  With workbuf_bytes != NULL init reports 0 bytes and
 *     crpspmm_engine_attach_workbuf() is accepted and ignored.
 *   - the B replication inside a grid column always moves exactly the rows the panel needs
 *     (what the reference calls A2A_B_FINEGRAIN=1); the knob is still read -- a2a_B_finegrain holds
 *     its value and the reference's "[INFO] ... Overriding parameter a2a_B_finegrain" line is printed.
 *   - pointer members describing the reference's internal buffers that have no host counterpart
 *     here (a2a_* arrays, loc_B, workbuf) are NULL; loc_A_*, red_B and loc_C are live.
 */
#ifndef CRP_CRPSPMM_H
#define CRP_CRPSPMM_H

#include <mpi.h>
#include "dev_type.h"
#include "mat_redist.h"

struct crpspmm_engine
{
    /* process grid: rank r sits at (r / np_col, r % np_col) */
    int    np_glb, rank_glb, np_row, np_col, rank_row, rank_col;
    int    glb_m, glb_n, glb_k;
    /* row panel of A held by this grid row: rows [loc_A_srow, loc_A_erow), loc_A_nnz nonzeros
     * starting at global nonzero loc_A_nnz_s */
    int    loc_A_srow, loc_A_erow, loc_A_nrow, loc_A_nnz, loc_A_nnz_s;
    /* B after the first redistribution: rows [rd_B_srow, rd_B_erow) x columns [loc_B_scol, loc_B_ecol) */
    int    rd_B_srow, rd_B_erow;
    /* B rows the panel touches: hull [loc_B_srow, loc_B_erow), loc_B_nrow = rows actually needed */
    int    loc_B_srow, loc_B_erow, loc_B_scol, loc_B_ecol, loc_B_nrow, loc_B_ncol;
    int    a2a_B_finegrain, alloc_workbuf, use_CUDA;
    size_t self_workbuf_bytes, rd_workbuf_bytes;
    int    *agv_A_recvcnts, *agv_A_displs;
    int    *a2a_B_sendcnts, *a2a_B_sdispls, *a2a_B_recvcnts, *a2a_B_rdispls;
    int    *a2a_B_send_ridx, *a2a_B_recv_ridx;
    int    *loc_A_rowptr, *loc_A_colidx;
    double *loc_A_val;
    double *a2a_B_sbuf, *a2a_B_rbuf, *red_B, *loc_B, *loc_C, *workbuf;
    MPI_Comm comm_row, comm_col, comm_glb;
    mat_redist_engine_p rd_Ai, rd_Av, rd_B, rd_C;   /* NULL: the engine holds crp_mat_redist handles */

    int    n_exec;
    double t_init, t_exec, t_rd_A, t_agv_A, t_rd_B, t_a2a_B, t_spmm, t_rd_C, t_exec_nr;
    size_t nelem_A_rd, nelem_A_agv, nelem_B_rd, nelem_B_a2av, nelem_B_a2av_min;

    void   *impl;               /* crp_crpspmm_p + communicator glue (not in the reference) */
};
typedef struct crpspmm_engine  crpspmm_engine_s;
typedef struct crpspmm_engine *crpspmm_engine_p;

#ifdef __cplusplus
extern "C" {
#endif

/* deprecated/src/crpspmm.h:89-98.  src_A_rowptr holds GLOBAL nonzero offsets (rowptr[0] = index
 * of this rank's first nonzero in the whole matrix), src_A_colidx global column indices. */
void crpspmm_engine_init(
    const int m, const int n, const int k,
    const int src_A_srow, const int src_A_nrow, const int *src_A_rowptr, const int *src_A_colidx,
    const int src_B_srow, const int src_B_nrow, const int src_B_scol, const int src_B_ncol,
    const int dst_C_srow, const int dst_C_nrow, const int dst_C_scol, const int dst_C_ncol,
    MPI_Comm comm, int use_CUDA, crpspmm_engine_p *engine_, size_t *workbuf_bytes
);
void crpspmm_engine_attach_workbuf(crpspmm_engine_p engine, double *workbuf);
/* deprecated/src/crpspmm.h:118-122: all arrays on the host, B / C row-major */
void crpspmm_engine_exec(
    crpspmm_engine_p engine, const int *src_A_rowptr, const int *src_A_colidx, const double *src_A_val,
    const double *src_B, const int ldB, double *dst_C, const int ldC
);
void crpspmm_engine_free(crpspmm_engine_p *engine_);
void crpspmm_engine_print_stat(crpspmm_engine_p engine);
void crpspmm_engine_clear_stat(crpspmm_engine_p engine);

#ifdef __cplusplus
}
#endif
#endif
