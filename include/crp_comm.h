/*
 * crp_comm.h -- communicator abstraction the CRP-SpMM engines are written
 * against.  The reference talks to MPI directly (call sites listed per
 * member below, paths under /root/reference); here the same operations are a
 * table of function pointers so that one engine serves
 *   - the MPI facade (rowpara_spmm.h / para2d_spmm.h: MPI for the host
 *     control plane, device payloads staged or RCCL),
 *   - a torch.distributed host binding (gloo on CPU for tests, nccl == RCCL
 *     over xGMI on MI355X), and
 *   - the single-rank case (crp_comm_self()).
 * Plain C, no MPI / torch types.  All operations are collective over the
 * communicator and blocking on the host unless stated otherwise.
 */
#ifndef CRP_COMM_H
#define CRP_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRP_OP_MAX 0
#define CRP_OP_SUM 1

typedef struct crp_comm crp_comm_t;

struct crp_comm
{
    void *ctx;
    int   nproc;   /* MPI_Comm_size  src/rowpara_spmm.c:31 */
    int   rank;    /* MPI_Comm_rank  src/rowpara_spmm.c:32 */

    /* MPI_Alltoall of `count` int32 per peer              src/rowpara_spmm.c:154 */
    void (*alltoall_i32)(void *ctx, const int *send, int *recv, int count);
    /* MPI_Alltoallv of int32 (counts / displs in elements) src/rowpara_spmm.c:159-162 */
    void (*alltoallv_i32)(void *ctx, const int *send, const int *scnts, const int *sdispls,
                          int *recv, const int *rcnts, const int *rdispls);
    /* MPI_Allgather / MPI_Allgatherv of raw bytes          src/para2d_spmm.c:61,69,81-83;
     * src/mat_redist.c:88.  rbytes / rdispls have nproc entries (bytes). */
    void (*allgatherv_bytes)(void *ctx, const void *send, size_t sbytes, void *recv,
                             const size_t *rbytes, const size_t *rdispls);
    /* MPI_Barrier                                          examples/test_rp_spmm.c:133 */
    void (*barrier)(void *ctx);
    /* MPI_Reduce to rank 0, op = CRP_OP_MAX | CRP_OP_SUM   src/rowpara_spmm.c:439-442 */
    void (*reduce_f64)(void *ctx, const double *in, double *out, int count, int op);
    void (*reduce_u64)(void *ctx, const uint64_t *in, uint64_t *out, int count, int op);

    /* The per-multiply B exchange (MPI_Isend/Irecv ring or MPI_Alltoallv,
     * src/rowpara_spmm.c:275-309): sparse all-to-all of fp64 elements between
     * DEVICE buffers; counts / displs in elements, 64-bit; pairs with a zero
     * count are skipped.  Enqueued on / ordered after `stream`; must be
     * complete from the point of view of work enqueued on `stream` afterwards. */
    void (*alltoallv_dev_f64)(void *ctx, const double *send_dev, const long long *scnts,
                              const long long *sdispls, double *recv_dev, const long long *rcnts,
                              const long long *rdispls, void *stream);

    /* Host all-to-all of raw bytes (counts / displs in bytes, nproc entries each): the payload
     * of the generic redistribution, MPI_Neighbor_alltoallv in src/mat_redist.c:357-360. */
    void (*alltoallv_bytes)(void *ctx, const void *send, const size_t *scnts, const size_t *sdispls,
                            void *recv, const size_t *rcnts, const size_t *rdispls);

    /* MPI_Comm_split(color, key)                           src/para2d_spmm.c:41-43.
     * Returns a new communicator owned by the caller (release with ->free). */
    crp_comm_t *(*split)(void *ctx, int color, int key);
    void (*free)(crp_comm_t *self);

    /* (appended member; NULL = not available, the engines then use allgatherv_bytes on host copies.)
     * The one-time replication of an A row panel inside a grid row, between DEVICE buffers: every rank's
     * `sbytes` land at recv_dev + rdispls[q] on all ranks (2 x MPI_Iallgatherv in src/para2d_spmm.c:81-83).
     * Enqueued on `stream`. */
    void (*allgatherv_dev)(void *ctx, const void *send_dev, size_t sbytes, void *recv_dev, const size_t *rbytes,
                           const size_t *rdispls, void *stream);
};

/* Single-rank communicator (nproc = 1); every collective is a local copy.
 * Release with comm->free(comm). */
crp_comm_t *crp_comm_self(void);

#ifdef __cplusplus
}
#endif
#endif
