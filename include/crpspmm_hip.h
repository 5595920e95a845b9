/*
 * crpspmm_hip.h -- device-level C ABI of the MI355X (gfx950) CRP-SpMM hot path.
 *
 * These are the entry points a host binding (cgo / JNI / ctypes / the MPI
 * facade in rowpara_spmm.h) calls for the work the reference does inside
 * rp_spmm_exec():
 *
 *   reference call site (under /root/reference)        replaced by
 *   -------------------------------------------------  -------------------------
 *   src/rowpara_spmm.c:388-408  mkl_sparse_d_create_csr
 *        + mkl_sparse_d_mm + mkl_sparse_destroy         crp_csr_dev_create (once)
 *                                                       + crp_spmm_csr_f64
 *   src/rowpara_spmm.c:232-262  pack B rows (OpenMP)    crp_gather_rows_f64
 *   src/rowpara_spmm.c:313-344  unpack B rows           crp_scatter_rows_f64
 *   src/rowpara_spmm.c:348-384  self-to-self copy       eliminated (two-source
 *                                                       column index, see below)
 *   deprecated/src/cuda_proxy.cu:53-118 mem/copy shims  crp_dev_* helpers
 *
 * Plain pointers and sizes only; no torch / MPI types.  Every function
 * returns 0 on success or a hipError_t value (> 0) / negative argument
 * error; nothing here falls back to the CPU.
 *
 * Two-source column index: a column index c >= 0 addresses row c of the
 * caller's local B block (B0, leading dimension ldB0); c < 0 addresses row
 * (~c) of the compact buffer of rows received from peers (B1, ldB1).  With
 * one rank every index is >= 0 and B1 may be NULL.
 */
#ifndef CRPSPMM_HIP_H
#define CRPSPMM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRP_LAYOUT_ROW_MAJOR 0
#define CRP_LAYOUT_COL_MAJOR 1

/* Opaque device-resident sparse matrix (CSR arrays + launch schedule). */
typedef struct crp_csr_dev *crp_csr_dev_p;

/* ---- device management --------------------------------------------------- */
int crp_hip_device_count(int *count);
int crp_hip_set_device(int dev);
int crp_hip_get_device(int *dev);
/* name must hold >= 256 bytes; cu_count / hbm_bytes may be NULL. */
int crp_hip_device_info(int dev, char *name, int *cu_count, size_t *hbm_bytes);

/* PCI bus id of the current device ("0000:c1:00.0"): tells whether two ranks share a GPU. len >= 16. */
int crp_hip_device_bus_id(char *out, size_t len);

int crp_dev_malloc(void **ptr, size_t bytes);
int crp_dev_free(void *ptr);
int crp_dev_memset(void *ptr, int value, size_t bytes, void *stream);
/* kind: 0 host->device, 1 device->host, 2 device->device.  Asynchronous on
 * `stream` when the host side is pinned; crp_stream_sync() to wait. */
int crp_dev_memcpy(void *dst, const void *src, size_t bytes, int kind, void *stream);
/* strided copy of `height` rows of `width_bytes` bytes (hipMemcpy2DAsync); pitches in bytes. */
int crp_dev_memcpy2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes,
                     size_t height, int kind, void *stream);
/* pinned host memory (hipHostMalloc / hipHostFree) */
int crp_host_malloc(void **ptr, size_t bytes);
int crp_host_free(void *ptr);
/* returns 1 in *is_dev when ptr is device memory, 0 for host memory. */
int crp_dev_ptr_is_device(const void *ptr, int *is_dev);

int crp_stream_create(void **stream);
int crp_stream_destroy(void *stream);
int crp_stream_sync(void *stream);
int crp_event_create(void **event);
int crp_event_destroy(void *event);
int crp_event_record(void *event, void *stream);
int crp_event_sync(void *event);
int crp_stream_wait_event(void *stream, void *event);
int crp_event_elapsed_ms(void *start, void *stop, float *ms);

/* ---- device-resident CSR -------------------------------------------------- */
/* Upload a host CSR (0-based rowptr[0] == 0, int32 indices, fp64 values).
 * colidx may carry the two-source encoding described above; ncol is the row
 * count of B0 (used for argument checking only).  Blocking. */
int crp_csr_dev_create(int nrow, int ncol, const int *rowptr, const int *colidx,
                       const double *val, crp_csr_dev_p *out);
/* The same, for a matrix whose VALUES already sit in device memory (a row panel replicated between GPUs,
 * /root/reference/src/para2d_spmm.c:56-86): val_host is still read (the derived formats are built on the host), but the
 * device CSR takes its values from val_dev instead of a second upload.  src_start = NULL: val_dev holds the nnz values in
 * order; else row t's values start at val_dev[src_start[t]] (host array of nrow entries: a row subset of a larger matrix). */
int crp_csr_dev_create_dv(int nrow, int ncol, const int *rowptr, const int *colidx, const double *val_host,
                          const double *val_dev, const int *src_start, crp_csr_dev_p *out);
int crp_csr_dev_destroy(crp_csr_dev_p *A);
/* New values for the same sparsity pattern (val in the order given at create; host or device
 * pointer): refreshes the CSR copy and every derived format on `stream`. */
int crp_csr_dev_update_values(crp_csr_dev_p A, const double *val, void *stream);
/* Row-subset matrices: row i of A writes row rowmap[i] of C (0 <= rowmap[i] < c_nrow; host array of
 * nrow entries, copied).  Lets a product be split into row subsets that run at different times
 * (rows with no remote column while the B exchange is in flight, the rest after it) without
 * touching the per-row summation order.  rowmap = NULL restores the identity.  Blocking. */
int crp_csr_dev_set_rowmap(crp_csr_dev_p A, const int *rowmap, int c_nrow);
int crp_csr_dev_nrow(crp_csr_dev_p A);
long long crp_csr_dev_nnz(crp_csr_dev_p A);
/* bytes of HBM the kernel must touch for A itself: 12*nnz + 4*(nrow+1). */
long long crp_csr_dev_bytes(crp_csr_dev_p A);
/* csr_mat_row_part_comm_size (/root/reference/src/spmat_part.c:38-64; include/spmat_part.h) evaluated ON THE DEVICE for a
 * matrix that is device-resident already -- re-planning a grid for other widths or rank counts without touching the host CSR:
 * comm_sizes[b] = distinct columns the rows rblk_ptr[b] .. rblk_ptr[b + 1] name outside [x_displs[b], x_displs[b + 1]),
 * *total_size their sum; bit-exact against the host function.  One bitmap of ncol bits per block in HBM (nblk * ncol / 8
 * bytes, temporary).  Only for matrices with plain column indices (no two-source encoding): -2 otherwise. */
int crp_csr_dev_row_part_comm_size(crp_csr_dev_p A, int nblk, const int *rblk_ptr, const int *x_displs, int *comm_sizes,
                                   int *total_size);

/* what variant 0 resolves to for this matrix: 1 csr-rowgroup, 2 rowpanel-R4, 3 rowpanel-R8
 * (chosen at create time from how many columns the rows of a panel share;
 * CRPSPMM_SPMM_VARIANT=1|2|3 overrides). */
/* The fp32 value path (BASELINE configs[3]; the reference itself is fp64-only, src/rowpara_spmm.h:28): the same
 * product with A's values, B and C in fp32 and fp32 FMAs, row-major operands.  A is the matrix created from fp64
 * values; its fp32 copies are derived on first use and follow crp_csr_dev_update_values.  variant 0 = auto (the
 * fp32 instance of the team kernel from 32 columns on (from 65 on mostly-hole panels) where teams share columns and the operands are 16-byte
 * aligned with n, ldB, ldC multiples of 4; else the fp32 CSR row-group kernel), 1 = row-group, 5 = team kernel.
 * Parity is defined against the fp64 product: relative Frobenius error <= 1e-5 (tests/test_gpu_parity.py). */
int crp_spmm_csr_f32(crp_csr_dev_p A, int n, const float *B0, long long ldB0, const float *B1, long long ldB1, float *C,
                     long long ldC, int variant, void *stream);

int crp_csr_dev_auto_variant(crp_csr_dev_p A);
/* 1 when the derived formats (row panels, teams) hold the rows in the locality order of csrc/locality.cpp
 * instead of the caller's order (taken at create time when it lets the rows of a panel share more columns;
 * CRPSPMM_REORDER=0|1 overrides).  Results do not depend on it: C rows are written through a row map and
 * every row's products are still summed in the kernel variant's own order. */
int crp_csr_dev_reordered(crp_csr_dev_p A);
/* the variant `variant = 0` (auto) runs for a row-major fp64 product of n columns with 16-byte aligned operands and even n,
 * ldB, ldC: the create-time choice (crp_csr_dev_auto_variant), replaced by 5 (team2-R8) from 96 columns on (from 33 when the
 * row-panel format would ask for more than 12 B row slices per row of A, from 48 when the matrix has no stride lattice; from 33 when
 * fewer than 35 % of the (row, entry) pairs of the R = 8 panels exist; for 61 .. 64 columns otherwise) where 64 consecutive rows share columns, and by 1
 * below 24 columns.  Operands that are not aligned like that fall back (5 -> 3 -> 1): what a product actually launched is
 * crp_csr_dev_last_variant().  Variants 4 and 6 (the round-1 LDS team kernel, the narrow team kernel of round 3) were
 * measured slower than what auto picks at every width and removed in round 4: asking for them returns -1;
 * variant 7 (team2r-R8: lane groups own rows, csrc/team2r_kernel.hip) replaces the create-time choice at 24 <= n <= 32 when fewer
 * than 35 % of the (row, entry) pairs of the R = 8 panels exist and 64 consecutive rows share columns (CRPSPMM_TEAM2R=0|1 forces; asked
 * for explicitly it runs up to 64 columns). */
int crp_csr_dev_resolved_variant(crp_csr_dev_p A, int n);
/* the variant the last crp_spmm_csr_f64 / _f32 on this matrix launched (after every fallback), or 0 before the first */
int crp_csr_dev_last_variant(crp_csr_dev_p A);
/* 1 when the team formats built so far found the two nested strides of a mesh numbered along its lines (their
 * teams are then blocks of neighbouring mesh lines), 0 otherwise / not built yet. */
int crp_csr_dev_lattice(crp_csr_dev_p A);
/* Host-only: build the row-panel format the rowpanel kernels consume (R = 4 or 8)
 * and return malloc'd copies (caller frees).  Panel p owns entries pptr[p] .. pptr[p+1]
 * (padded to multiples of 8 with mask-0 entries); entry q has column pcol[q] (two-source
 * encoding), row-presence mask byte (pmask4[q/4] >> 8*(q%4)) & 0xFF and values
 * pval[q*R .. q*R+R-1].  The entries of a panel are in column order, except under the team schedule
 * (below) where they are in the order the panel's wave meets them.
 * porder (optional) receives the processing order: *norder positions, position s is the panel the
 * s-th wave takes (4 consecutive positions = one workgroup), -1 = none.  Which order:
 *   - matrices with two nested far strides (3D meshes in natural order), R = 8: the TEAM SCHEDULE --
 *     the four waves of a workgroup take a 2 x 2 block of tooth-mate panels (panels that read the
 *     same B rows through different bands), every panel's entries are re-ordered so that the four
 *     waves reach a shared row after the same number of entries, and the workgroups sweep the
 *     teeth in lockstep per XCD; norder = 4 * teams;
 *   - the same matrices with R = 4: the stride-lattice order of single panels; norder = npanel;
 *   - otherwise groups of consecutive panels visited breadth-first over shared B rows.
 * CRPSPMM_PANEL_ORDER=0|1|2|3 forces natural / breadth-first / lattice / team schedule,
 * CRPSPMM_PANEL_GROUP sets the breadth-first group size.  Used by the CPU tests of the format. */
int crp_panel_format_host(int nrow, const int *rowptr, const int *colidx, const double *val, int R,
                          int *npanel, int **pptr, int **pcol, unsigned **pmask4, double **pval,
                          long long *real_entries, int **porder, int *norder);

/* Host-only: the team format variant 4 ("team-R8") consumes, built on the R = 8 panels: team g owns
 * panels tpanel[4g .. 4g+3] (-1 = none), one per wave of a workgroup, and union entries
 * tptr[g] .. tptr[g+1] (padded to multiples of 8 with mask-0 entries): column tcol[q], and in
 * tmask[q] byte w the row mask of wave w's panel for that column (0: wave w skips the entry).  Every
 * wave meets its own panel's entries in that panel's own order.  torder = processing order of
 * the teams; *lattice = 1 when the teams are 2 x 2 blocks of a stride lattice, 0 for four consecutive
 * panels.  malloc'd copies (caller frees).  Used by the CPU tests of the format. */
int crp_team_format_host(int nrow, const int *rowptr, const int *colidx, const double *val, int *nteam, int *lattice,
                         int **tpanel, int **tptr, int **tcol, unsigned **tmask, int **torder);

/* Host-only: the streams variant 5 ("team2-R8", csrc/team2_kernel.hip) consumes: teams of 8 panels of 8 rows
 * (one per wave of a 512-thread workgroup; tpanel, -1 = none), the union of whose columns is walked in rounds
 * of up to 8 slots.  tinfo[4g] = rounds of team g, [4g+1] = its first record block, [4g+2] = parts of all its
 * waves, [4g+3] = filled slots.  trec = record blocks of 8 rounds x 8 waves x 4 words (csrc/panel_format.h, Team2Host):
 * word 0 = part count | ring slots | flags | value position of part 0; word 1 = ranges (6 bits per part: first * 8 + len
 * - 1) | value position of part 1 | size class of round r+3's value block; word 2 = offset of round r+3's value block in
 * the wave's stream (20 bits, units of 4 values) | value positions of parts 2 and 3; word 3 = column this wave fetches for
 * round r+3.  tpro[((3g + d)*8 + w)*2 ..] = {column, value offset} of round d < 3.  Values are compact:
 * a part of len rows holds len values; a round's parts form one block (padded to 4 values) of the wave's stream, which
 * starts at tval[4 * tvoff[8g + w]]; part i's first value sits (position_i - 7 + first_i) values into the block.
 * vmap[nz] = index in tval of CSR nonzero nz.  *nvalent = values in tval.  malloc'd copies (caller frees).  Used by the
 * CPU tests, which replay the streams in numpy. */
int crp_team2_format_host(int nrow, const int *rowptr, const int *colidx, const double *val, int *nteam, int *lattice,
                          int **tpanel, int **tinfo, int **tpro, unsigned **trec, long long *nrecwords,
                          long long **tvoff, double **tval, long long *nvalent, int **torder, unsigned **vmap);
/* The launch grid of the streams crp_team2_format_host() built last (process-wide, planning / test helper only): 8 runs
 * of *ngrid / 8 entries, run x = the teams XCD x processes, in order (-1 = none); a generation = 64 consecutive
 * entries of a run.  Slots of a round that hold no B row name a row of the team (fetched, not read). */
int crp_team2_format_host_grid(int **tgrid, int *ngrid);
/* 1 when the value blocks of those streams are compact (a part of len rows holds len values), 0 when every part holds 8
 * values, row r of part i at 8 i + r of the round's block (panels filled to 40 % and more; CRPSPMM_TEAM2_COMPACT=0|1 forces) */
int crp_team2_format_host_compact(void);

/* Host-only: the streams variant 7 ("team2r-R8", csrc/team2r_kernel.hip: the row-owner team kernel for operands of 24 .. 64 fp64
 * columns) consumes.  Teams as variant 5.  A round has 8 G d ring slots of 1024 / G bytes (G = 4: n <= 32, 2: n <= 64; d = 2 row DMA
 * instructions per wave and round); wave w fetches slots G d w .. G d (w + 1) - 1.  Round r (from tinfo[2g + 1] on, tinfo[2g] rounds), wave w owns trec[(r * 8 + w) * 16 ..]: [0] Lp =
 * steps (multiple of 2, <= 12); [1] first 16-byte unit of its block inside the wave's stream, which starts at byte 16 * tvoff[8g + w]
 * of tval; [2 .. 2 + G d) the columns of the slots the wave fetches for this round.  A block = [8 rows][Lp] doubles, then
 * [8 rows][Lp] uint16, then a 64-byte header = the record of round r + 2 of the same team and wave (zeros past the last round):
 * step s of row i multiplies the value with the B row slice at that byte offset of the round's ring set (slot * 1024 / G);
 * 8192 d = the slice of zeros (padding, value 0.0).  vmap[nz] = 8-byte word of tval that holds CSR nonzero nz.
 * tent (may be NULL): per entry e of tgrid and wave w 32 words at tent[(e * 8 + w) * 32]: [0] rounds (0: no team), [1] panel, [2], [3]
 * tvoff, [4 .. 14) record of round 0, [14 .. 24) of round 1, [24 .. 32) 0xFFFFFFFF (the C rows, filled on the device).
 * stats (4, may be NULL): rounds, steps (sum of Lp), filled slots, nonzeros.  malloc'd copies (caller frees). */
int crp_team2r_format_host(int nrow, const int *rowptr, const int *colidx, const double *val, int G, int *nteam, int *lattice, int **tpanel,
                           int **tinfo, unsigned **trec, long long *nrecwords, long long **tvoff, double **tval, long long *nwords,
                           int **tgrid, int *ngrid, unsigned **vmap, long long *stats, unsigned **tent);

/* Host-only: the processing order of the rows of a square A that crp_csr_dev_create() applies for B-row
 * locality (csrc/locality.cpp: row groups with identical column lists, `nparts` slabs by breadth-first
 * bisection, reverse Cuthill-McKee inside every slab).  perm[i] = row processed at position i (caller provides
 * nrow ints).  Returns 0, or 1 when the matrix does not qualify (not square, two-source indices, too small):
 * perm is then the identity.  info (4 doubles, may be NULL): row groups, parts, mean |pos(col) - pos(row)| before, after. */
int crp_locality_order_host(int nrow, int ncol, const int *rowptr, const int *colidx, int nparts, int *perm, double *info);

/* ---- the hot kernel --------------------------------------------------------
 * C[nrow x n] := A * B (alpha = 1, beta = 0; C is overwritten, never read),
 * the arithmetic of mkl_sparse_d_mm as called at src/rowpara_spmm.c:403-406:
 * C[i][j] = sum_p val[p] * B[col[p]][j]; every product is formed once (absent pairs are skipped,
 * never multiplied by zero) and a row's products are summed one after the other with FMAs -- in
 * ascending p, except under the team schedule / variant 4 where the order is the team's schedule
 * (fixed at create time, so repeated calls are bit-identical).
 * layout 0: B0/B1/C row-major (ld >= n); layout 1: column-major
 * (ldB0 >= rows of B0, ldB1 >= rows of B1, ldC >= nrow).  All pointers are
 * device pointers; the launch is asynchronous on `stream`.  `variant` picks a
 * kernel (0 = automatic); see crp_spmm_variant_name(). */
int crp_spmm_csr_f64(crp_csr_dev_p A, int layout, int n,
                     const double *B0, long long ldB0,
                     const double *B1, long long ldB1,
                     double *C, long long ldC, int variant, void *stream);
const char *crp_spmm_variant_name(int variant);
int crp_spmm_variant_count(void);

/* ---- row gather / scatter (pack / unpack of the B exchange) ----------------
 * gather : dst[i][0:n] = src[ridx[i]][0:n]   (i < nidx)
 * scatter: dst[ridx[i]][0:n] = src[i][0:n]
 * layout 0: rows are contiguous (row-major, leading dimensions in elements);
 * layout 1: column-major operands (element (r, j) at r + j*ld).  ridx is a
 * device array of int32. */
int crp_gather_rows_f64(int layout, int nidx, int n, const int *ridx,
                        const double *src, long long lds,
                        double *dst, long long ldd, void *stream);
int crp_scatter_rows_f64(int layout, int nidx, int n, const int *ridx,
                         const double *src, long long lds,
                         double *dst, long long ldd, void *stream);
/* out-of-place transpose: dst[c][r] = src[r][c] for an nrow x ncol row-major
 * src (equivalently col-major <-> row-major conversion). */
int crp_transpose_f64(int nrow, int ncol, const double *src, long long lds,
                      double *dst, long long ldd, void *stream);

/* Diagnostics for the exchange / compute overlap (tools/overlap_probe.py): a stand-in for a transport's copy kernel --
 * `blocks` workgroups of 256 threads copy `bytes` (a multiple of 16) between device buffers and record the 100 MHz device
 * wall clock: stamps_dev[0] = earliest workgroup start (initialise to ~0), stamps_dev[1] = latest end (initialise to 0);
 * crp_probe_stamp writes the clock to *out_dev from a one-thread kernel.  crp_stream_create_cu_mask creates a stream whose
 * kernels run only on the compute units whose bit is set in mask32 (words of 32 CUs; hipExtStreamCreateWithCUMask). */
int crp_probe_copy(long long bytes, const void *src_dev, void *dst_dev, int blocks, unsigned long long *stamps_dev, void *stream);
int crp_probe_stamp(unsigned long long *out_dev, void *stream);
int crp_stream_create_cu_mask(void **stream, int nwords, const unsigned *mask32);

/* Library identification: "crpspmm-hip <version> gfx950". */
const char *crp_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
