/*
 * para2d_spmm.h -- MPI-typed facade of the 2D engine: the reference's public
 * API, unchanged (/root/reference/src/para2d_spmm.h:6-75; caller protocol in
 * examples/test_para2d_spmm.c:141-167). Implemented in
 * crp-spmm_amd/csrc/mpi_facade.cpp on top of crp_engine.h.
 */
#ifndef CRP_PARA2D_SPMM_H
#define CRP_PARA2D_SPMM_H

#include "rowpara_spmm.h"

struct para2d_spmm
{
    rp_spmm_p rp_spmm;      /* 1D engine on the grid column                  */
    MPI_Comm  comm_glb;     /* caller's communicator, not owned              */
    MPI_Comm  comm_col;     /* grid-column communicator, owned               */
    size_t    rA_cost;      /* floor(1.5 * nnz(A) * (pn - 1))                */
    double    t_init;
    double    t_ag_A;
    void     *impl;         /* crp_para2d_spmm_p + glue (not in the reference) */
};
typedef struct para2d_spmm  para2d_spmm_s;
typedef struct para2d_spmm *para2d_spmm_p;

#ifdef __cplusplus
extern "C" {
#endif

/* Rank r sits at grid position (r / pn, r % pn); array meanings as
 * src/para2d_spmm.h:22-41. Works at one rank (the reference self-sends and
 * hangs there, src/para2d_spmm.c:102-109). */
void para2d_spmm_init(
    MPI_Comm comm, const int pm, const int pn, const int *A0_rowptr,
    const int *B_rowptr, const int *AC_rowptr, const int *BC_colptr,
    const int *A_rowptr, const int *A_colidx, const double *A_val,
    para2d_spmm_p *para2d_spmm
);
void para2d_spmm_free(para2d_spmm_p *para2d_spmm);
void para2d_spmm_exec(
    para2d_spmm_p para2d_spmm, const int BC_layout, const double *B, const int ldB,
    double *C, const int ldC
);
void para2d_spmm_print_stat(para2d_spmm_p para2d_spmm);
void para2d_spmm_clear_stat(para2d_spmm_p para2d_spmm);

#ifdef __cplusplus
}
#endif
#endif
