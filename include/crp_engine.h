/*
 * crp_engine.h -- communicator-agnostic C ABI of the two CRP-SpMM engines.
 *
 * Same operations, argument meaning and error behaviour as the reference's
 * public API, with the MPI_Comm replaced by a crp_comm_t* (crp_comm.h):
 *
 *   reference (/root/reference)                         this header
 *   --------------------------------------------------  -----------------------
 *   rp_spmm_init        src/rowpara_spmm.h:60-64         crp_rp_spmm_init
 *   rp_spmm_exec        src/rowpara_spmm.h:78-81         crp_rp_spmm_exec
 *   rp_spmm_free        src/rowpara_spmm.h:67            crp_rp_spmm_free
 *   rp_spmm_print_stat  src/rowpara_spmm.h:84            crp_rp_spmm_print_stat
 *   rp_spmm_clear_stat  src/rowpara_spmm.h:87            crp_rp_spmm_clear_stat
 *   para2d_spmm_*       src/para2d_spmm.h:42-75          crp_para2d_spmm_*
 *
 * include/rowpara_spmm.h and include/para2d_spmm.h are the MPI-typed facade
 * (exact reference signatures) over these functions.
 *
 * Differences that are deliberate ("reference defects"):
 *   - A is uploaded once at init and stays device-resident; exec does no
 *     malloc/free and creates no sparse handle (defect 6);
 *   - locally owned B rows are read in place: there is no self-to-self copy
 *     (defect 7), received rows land contiguously so there is no unpack pass;
 *   - all offsets are 64-bit internally (defect 3);
 *   - para2d at one rank does not self-send (defect 1).
 * B and C may be host pointers (staged through device buffers, as the
 * reference API implies) or device pointers (detected automatically; the
 * zero-copy path the benchmark uses).
 */
#ifndef CRP_ENGINE_H
#define CRP_ENGINE_H

#include <stddef.h>
#include <stdint.h>
#include "crp_comm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct crp_rp_spmm     *crp_rp_spmm_p;
typedef struct crp_para2d_spmm *crp_para2d_spmm_p;

/* Host-side view of the exchange plan: the fields of struct rowpara_spmm
 * (src/rowpara_spmm.h:8-40), same names and meaning (counts / displs are in
 * ELEMENTS = rows * glb_n, as the reference stores them, but 64-bit). */
typedef struct crp_rp_plan_view
{
    int nproc, my_rank, glb_n, A_nrow, rB_nrow;
    int rB_self_src_offset, rB_self_dst_offset, rB_self_nrow;
    int rB_p2p, rB_reidx;
    const int       *A_rowptr;           /* A_nrow + 1, rebased to 0          */
    const int       *A_colidx;           /* compact (re-indexed) column ids   */
    const double    *A_val;
    const int       *rB_self_src_ridxs;  /* rB_self_nrow global row ids       */
    const long long *rB_scnts, *rB_sdispls;   /* nproc, nproc + 1             */
    const int       *rB_sridxs;          /* local B row ids to send           */
    const long long *rB_rcnts, *rB_rdispls;
    const int       *rB_rridxs;          /* compact rB row ids of received rows */
    size_t rB_recv_size;                 /* rows received from other ranks    */
    int    n_exec;
    double t_init, t_pack, t_a2a, t_unpack, t_spmm, t_exec;
} crp_rp_plan_view_t;

/* See rp_spmm_init (src/rowpara_spmm.h:49-64).  A_rowptr is a slice of the
 * global row pointer (global nnz offsets; rebased internally), A_srow is
 * accepted and ignored exactly like the reference.  comm is not duplicated
 * and must outlive the engine.  Honours RP_SPMM_P2P / RP_SPMM_REIDX
 * (src/rowpara_spmm.c:42-43).  On failure prints "[FATAL] ..." and aborts,
 * like ASSERT_PRINTF (src/utils.h:58-68). */
void crp_rp_spmm_init(int A_srow, int A_nrow, const int *A_rowptr, const int *A_colidx,
                      const double *A_val, const int *B_row_displs, int glb_n,
                      crp_comm_t *comm, crp_rp_spmm_p *rp_spmm);
void crp_rp_spmm_free(crp_rp_spmm_p *rp_spmm);
/* C := A * B.  BC_layout 0 row-major / 1 column-major; NULL engine is a no-op
 * (src/rowpara_spmm.c:217). */
void crp_rp_spmm_exec(crp_rp_spmm_p rp_spmm, int BC_layout, const double *B, int ldB,
                      double *C, int ldC);
/* Same, 64-bit leading dimensions and an explicit HIP stream, taken literally
 * (NULL = the null stream; crp_rp_spmm_exec uses a stream the engine owns).
 * With device pointers and timing off nothing synchronises. */
void crp_rp_spmm_exec_ex(crp_rp_spmm_p rp_spmm, int BC_layout, const double *B, long long ldB,
                         double *C, long long ldC, void *stream);
void crp_rp_spmm_print_stat(crp_rp_spmm_p rp_spmm);
void crp_rp_spmm_clear_stat(crp_rp_spmm_p rp_spmm);
/* rp_spmm_init for a caller that ALSO holds the values in device memory, in the order of A_val (A_val_dev: the panel a
 * device all-gather produced, csrc/para2d_engine.cpp): the engine's device matrices take them from there and nothing is
 * uploaded a second time.  A_val (host) is still required: the plan and the derived formats are built on the host and
 * struct rowpara_spmm::A_val is a public host field (/root/reference/src/rowpara_spmm.h:8-40).  A_val_dev may be freed
 * when the call returns.  crp_rp_spmm_values_from_device(): 1 when an engine was built that way. */
void crp_rp_spmm_init_dv(int A_srow, int A_nrow, const int *A_rowptr, const int *A_colidx, const double *A_val,
                         const double *A_val_dev, const int *B_row_displs, int glb_n, crp_comm_t *comm, crp_rp_spmm_p *rp_spmm);
int crp_rp_spmm_values_from_device(crp_rp_spmm_p rp_spmm);
void crp_rp_spmm_get_plan(crp_rp_spmm_p rp_spmm, crp_rp_plan_view_t *view);
/* Host seconds spent INSIDE the B exchange call (issuing the grouped sends / receives) since the last clear_stat, summed
 * over the execs -- accumulated in every timing mode; print_stat reports it per exec when there is more than one rank. */
double crp_rp_spmm_exchange_host_seconds(crp_rp_spmm_p rp_spmm);
/* timing = 1 (default): every phase is bracketed by stream synchronisation and
 * billed to t_pack / t_a2a / t_unpack / t_spmm like the reference; 0: fully
 * asynchronous exec, only t_exec (host enqueue time) is accumulated. */
/* Exchange / compute overlap: with more than one rank the rows of A are split at init into
 * "interior" rows (no column owned by a peer) and "boundary" rows; with timing off, exec runs the
 * B exchange on a second stream beside the interior rows' product and the boundary rows' product
 * after the rows have landed (per-row summation order unchanged).  Both counts are 0 when the
 * engine runs one product (one rank, nothing to receive, a part too small to pay for a launch,
 * or CRPSPMM_OVERLAP=0). */
void crp_rp_spmm_overlap_rows(crp_rp_spmm_p rp_spmm, int *n_interior, int *n_boundary);
void crp_rp_spmm_set_timing(crp_rp_spmm_p rp_spmm, int timing);
/* kernel variant for the local SpMM (crpspmm_hip.h: crp_spmm_variant_name). */
void crp_rp_spmm_set_variant(crp_rp_spmm_p rp_spmm, int variant);
/* bytes of HBM the local kernel must touch per exec (SURVEY 8d bytes_alg for
 * this rank): 12*nnz + 4*(A_nrow+1) + 8*n*(distinct B rows) + 8*n*A_nrow. */
long long crp_rp_spmm_alg_bytes(crp_rp_spmm_p rp_spmm);
long long crp_rp_spmm_nnz(crp_rp_spmm_p rp_spmm);
/* What the local SpMM of this engine is (for reports): the kernel variant in use (crp_spmm_variant_name; the auto
 * choice resolved for the engine's glb_n), 1 if its formats hold the rows in the locality order of
 * csrc/locality.cpp, 1 if a stride lattice was found.  Any pointer may be NULL. */
void crp_rp_spmm_kernel_info(crp_rp_spmm_p rp_spmm, int *variant, int *reordered, int *lattice);
/* New values for the same sparsity pattern, in the order of the A_val given to init (the
 * deprecated crpspmm_engine passes A's values on every exec: deprecated/src/crpspmm.h:108-117). */
void crp_rp_spmm_update_values(crp_rp_spmm_p rp_spmm, const double *A_val);

/* See para2d_spmm_init (src/para2d_spmm.h:22-47). Rank r sits at grid position
 * (r / pn, r % pn); A_rowptr/A_colidx/A_val is the rank's A0 slice. */
void crp_para2d_spmm_init(crp_comm_t *comm, int pm, int pn, const int *A0_rowptr,
                          const int *B_rowptr, const int *AC_rowptr, const int *BC_colptr,
                          const int *A_rowptr, const int *A_colidx, const double *A_val,
                          crp_para2d_spmm_p *para2d_spmm);
void crp_para2d_spmm_free(crp_para2d_spmm_p *para2d_spmm);
void crp_para2d_spmm_exec(crp_para2d_spmm_p para2d_spmm, int BC_layout, const double *B, int ldB,
                          double *C, int ldC);
void crp_para2d_spmm_exec_ex(crp_para2d_spmm_p para2d_spmm, int BC_layout, const double *B,
                             long long ldB, double *C, long long ldC, void *stream);
void crp_para2d_spmm_print_stat(crp_para2d_spmm_p para2d_spmm);
void crp_para2d_spmm_clear_stat(crp_para2d_spmm_p para2d_spmm);
crp_rp_spmm_p crp_para2d_spmm_rp(crp_para2d_spmm_p para2d_spmm);
/* 1 when init replicated the panel's column indices and values between device buffers (the communicator's
 * allgatherv_dev: RCCL; CRPSPMM_REPLICATE=host forces the host path), 0 when it went through allgatherv_bytes. */
int crp_para2d_spmm_replicated_on_device(crp_para2d_spmm_p para2d_spmm);
/* How often the values of the replicated panel crossed PCIe towards the device: 0 = never (they were all-gathered between
 * device buffers and the engine's matrices were filled from that copy), 1 = once (host replication, or one grid column). */
int crp_para2d_spmm_value_uploads(crp_para2d_spmm_p para2d_spmm);
size_t crp_para2d_spmm_rA_cost(crp_para2d_spmm_p para2d_spmm);
double crp_para2d_spmm_t_ag_A(crp_para2d_spmm_p para2d_spmm);

/* ---- generic dense 2D-block redistribution (src/mat_redist.h:7-100) over a crp_comm_t --------
 * Every rank owns the rectangle (src_srow, src_scol, src_nrow, src_ncol) of a global row-major
 * matrix (owners must not overlap) and asks for (req_srow, req_scol, req_nrow, req_ncol).  init
 * gathers all rectangles, intersects them (src/mat_redist.c:9-41, 81-153) and records, in rank
 * order, which rectangles go to / come from whom; exec packs the send rectangles contiguously
 * (row-major, ld = ncol), exchanges them, and unpacks into dst.  dev_type (include/dev_type.h):
 * 0 host buffers; 1 device buffers, exchange staged through pinned host memory; 2 device buffers,
 * exchanged device to device (dt_size 8 only; otherwise staged).  Pure byte movement: bit-exact. */
typedef struct crp_mat_redist *crp_mat_redist_p;
typedef struct crp_mat_redist_view
{
    int nproc, rank, src_srow, src_scol, src_nrow, src_ncol, req_srow, req_scol, req_nrow, req_ncol;
    int n_proc_send, n_proc_recv, send_cnt, recv_cnt;      /* counts in elements                      */
    const int *send_ranks, *send_sizes, *send_displs, *sblk_sizes;   /* as src/mat_redist.h:27-30     */
    const int *recv_ranks, *recv_sizes, *recv_displs, *rblk_sizes;   /* as src/mat_redist.h:31-34     */
    size_t dt_size;
    int    dev_type;
    double hd_trans_ms;
} crp_mat_redist_view_t;
/* *engine is left untouched (NULL) on an invalid dev_type, after the reference's "[ERROR] ...
 * Invalid device type" message (src/mat_redist.c:51-55).  workbuf_bytes != NULL: the size of the
 * work buffer is returned and the caller attaches one; NULL: the engine allocates it. */
void crp_mat_redist_init(int src_srow, int src_scol, int src_nrow, int src_ncol, int req_srow, int req_scol,
                         int req_nrow, int req_ncol, crp_comm_t *comm, size_t dt_size, int dev_type,
                         crp_mat_redist_p *engine, size_t *workbuf_bytes);
void crp_mat_redist_attach_workbuf(crp_mat_redist_p engine, void *workbuf_h, void *workbuf_d);
void crp_mat_redist_exec(crp_mat_redist_p engine, const void *src_blk, int src_ld, void *dst_blk, int dst_ld);
void crp_mat_redist_free(crp_mat_redist_p *engine);
void crp_mat_redist_get_view(crp_mat_redist_p engine, crp_mat_redist_view_t *view);

/* ---- compatibility engine: the older all-in-one API (deprecated/src/crpspmm.h:89-130) ----------
 * A arrives in any 1D row distribution, B and C in arbitrary 2D blocks, all on the HOST, and A's
 * values are passed on every exec.  init plans the np_row x np_col grid with the deprecated
 * engine's own rule (per prime factor of P, largest first, split M or N by comparing
 * 1.5*nnz*n_split(*p) + k*n-style upper bounds built from per-row column RANGES,
 * deprecated/src/crpspmm.c:136-195), redistributes A's pattern to row panels, B to an even
 * (k / np_row) x (n / np_col) layout and C back to the caller's layout with crp_mat_redist, and
 * runs the 1D row-parallel device engine inside every grid column.  Same statistics lines as
 * crpspmm_engine_print_stat (deprecated/src/crpspmm.c:715-772). */
typedef struct crp_crpspmm *crp_crpspmm_p;
typedef struct crp_crpspmm_view
{
    int np_glb, rank_glb, np_row, np_col, rank_row, rank_col, glb_m, glb_n, glb_k;
    int loc_A_srow, loc_A_erow, loc_A_nrow, loc_A_nnz, loc_A_nnz_s;
    int rd_B_srow, rd_B_erow, loc_B_scol, loc_B_ecol, loc_B_ncol;
    int loc_B_srow, loc_B_erow, loc_B_nrow;   /* hull and count of the B rows the panel touches */
    int a2a_B_finegrain;                      /* value of the A2A_B_FINEGRAIN knob (reported only) */
    const int *loc_A_rowptr, *loc_A_colidx;   /* panel CSR on the host (rowptr rebased to 0) */
    const double *loc_A_val, *red_B, *loc_C;
    int n_exec;
    double t_init, t_exec, t_rd_A, t_agv_A, t_rd_B, t_a2a_B, t_spmm, t_rd_C, t_exec_nr;
    size_t nelem_A_rd, nelem_A_agv, nelem_B_rd, nelem_B_a2av, nelem_B_a2av_min;
} crp_crpspmm_view_t;
void crp_crpspmm_init(int m, int n, int k, int src_A_srow, int src_A_nrow, const int *src_A_rowptr,
                      const int *src_A_colidx, int src_B_srow, int src_B_nrow, int src_B_scol, int src_B_ncol,
                      int dst_C_srow, int dst_C_nrow, int dst_C_scol, int dst_C_ncol, crp_comm_t *comm,
                      crp_crpspmm_p *engine);
/* host planning and redistribution only (no device state): exec stops after A's values and B
 * have reached the internal layout (inspect them through the view); used by the CPU tests */
void crp_crpspmm_init_plan_only(int m, int n, int k, int src_A_srow, int src_A_nrow, const int *src_A_rowptr,
                                const int *src_A_colidx, int src_B_srow, int src_B_nrow, int src_B_scol,
                                int src_B_ncol, int dst_C_srow, int dst_C_nrow, int dst_C_scol, int dst_C_ncol,
                                crp_comm_t *comm, crp_crpspmm_p *engine);
void crp_crpspmm_exec(crp_crpspmm_p engine, const int *src_A_rowptr, const int *src_A_colidx,
                      const double *src_A_val, const double *src_B, int ldB, double *dst_C, int ldC);
void crp_crpspmm_free(crp_crpspmm_p *engine);
void crp_crpspmm_print_stat(crp_crpspmm_p engine);
void crp_crpspmm_clear_stat(crp_crpspmm_p engine);
void crp_crpspmm_get_view(crp_crpspmm_p engine, crp_crpspmm_view_t *view);
/* the deprecated engine's grid rule alone (host, no communication): A_rowptr_glb has m + 1
 * entries, cidx_se holds (first, last) column of every row (2*m ints; an empty row holds any
 * pair with first > last and is ignored).  m_split_idx receives np_row + 1 row offsets
 * (room for P + 1). */
void crp_crpspmm_plan_grid(int P, int m, int n, int k, const int *A_rowptr_glb, const int *cidx_se,
                           int *np_row, int *np_col, int *m_split_idx);

/* ---- planner extension: grid for an A that is reused rA times ------------------------------------
 * Same arguments and output arrays as calc_spmm_part2d_from_1d (spmat_part.h; src/spmat_part.h:55-76)
 * without dbg_print.  Prices EVERY pm x pn with pn | nproc as floor(1.5 nnz (pn-1)) [replicate A once]
 * + rA * n * (B rows exchanged per multiply) and returns the cheapest; the reference rule leaves rA
 * out of its 1D starting cost and searches greedily, so there rA > 1 favours 1D.  Host only. */
void crp_spmm_part2d_amortized(int nproc, int m, int n, int k, const int *rb_displs0, const int *rowptr,
                               const int *colidx, int rA, int *pm, int *pn, size_t *comm_cost, int **A0_rowptr,
                               int **B_rowptr, int **AC_rowptr, int **BC_colptr);

/* Extension: grid choice by a TIME model of one node of point-to-point links instead of a byte
 * count -- one-time replication of the A panels (fan-out: the pn - 1 pieces arrive over different links), per multiply
 * max(local product at the kernels' measured roofline fraction for n / pn columns, slowest PAIR of the B exchange), priced
 * as t_rep / rA + t_exec; grids that do not fit a GPU's HBM are skipped.  mm = NULL or {link GB/s one way (64), HBM GB/s
 * (8000), HBM bytes per GPU (288e9)}; times (optional) = {t_rep, t_exch, t_comp} of the winner, seconds.  Output arrays as
 * calc_spmm_part2d_from_1d.  Host only. */
void crp_spmm_part2d_timed(int nproc, int m, int n, int k, const int *rb_displs0, const int *rowptr, const int *colidx,
                           int rA, const double *mm, int *pm, int *pn, double *times, int **A0_rowptr, int **B_rowptr,
                           int **AC_rowptr, int **BC_colptr);

/* ---- binary CSR cache (ingest extension) ---------------------------------------------------------
 * A converted matrix kept beside its .mtx so that later runs skip the text parse
 * (examples/mmio_utils.c:11-190 takes 3 s for pwtk, minutes for nlpkkt240).  One little-endian file:
 * "CRPCSR01", int64 nrow / ncol / nnz, rowptr, colidx, val.  Return 0, or -1 (unreadable, wrong
 * magic, inconsistent sizes); read hands back malloc'd arrays (caller frees). */
int crp_csr_cache_write(const char *fname, int nrow, int ncol, const int *rowptr, const int *colidx,
                        const double *val);
int crp_csr_cache_read(const char *fname, int *nrow, int *ncol, int **rowptr, int **colidx, double **val);

/* ---- host-only pieces exposed for tests (no GPU needed) -------------------
 * Build only the exchange plan (everything rp_spmm_init computes on the host,
 * including the alltoall of needed row ids) without touching the device.
 * Release with crp_rp_spmm_free. exec on such an engine aborts. */
void crp_rp_spmm_init_plan_only(int A_srow, int A_nrow, const int *A_rowptr, const int *A_colidx,
                                const double *A_val, const int *B_row_displs, int glb_n,
                                crp_comm_t *comm, crp_rp_spmm_p *rp_spmm);
void crp_para2d_spmm_init_plan_only(crp_comm_t *comm, int pm, int pn, const int *A0_rowptr,
                                    const int *B_rowptr, const int *AC_rowptr, const int *BC_colptr,
                                    const int *A_rowptr, const int *A_colidx, const double *A_val,
                                    crp_para2d_spmm_p *para2d_spmm);
/* The device-side (two-source) column index the kernel consumes, host copy:
 * c >= 0 local B row, c < 0 -> ~c = row of the receive buffer. */
const int *crp_rp_spmm_dev_colidx_host(crp_rp_spmm_p rp_spmm);

#ifdef __cplusplus
}
#endif
#endif
