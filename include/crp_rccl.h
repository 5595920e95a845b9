/*
 * crp_rccl.h -- device-payload collectives of the CRP-SpMM engines on RCCL (xGMI inside a node), in
 * libcrpspmm_hip.so.  One handle = one RCCL communicator, one rank per GPU, created from a unique id that the
 * caller's control plane (MPI_Bcast, a gloo broadcast, ...) carries from rank 0 to everybody: the library itself
 * needs no launcher.  It replaces, for data that lives in HBM,
 *   - the per-multiply B exchange  MPI_Isend/Irecv ring or MPI_Alltoallv   /root/reference/src/rowpara_spmm.c:275-309
 *   - the one-time replication of an A row panel  2 x MPI_Iallgatherv       /root/reference/src/para2d_spmm.c:56-86
 *   - MPI_Neighbor_alltoallv of the generic redistribution (device mode)    /root/reference/src/mat_redist.c:380-386
 * RCCL has no "v" collectives: all three are ONE group of ncclSend / ncclRecv (pairs with nothing to move are
 * skipped; over xGMI every pair has a direct link).  Counts are 64-bit: no 2 GiB limit.
 * Every call returns 0 or a negative error (the RCCL / HIP error is printed to stderr); all ranks must call the
 * collectives in the same order.  The crp_comm_* adaptors at the end have the signatures of the device members of
 * crp_comm_t (crp_comm.h) with ctx = the handle, so a communicator can point at them directly.
 */
#ifndef CRP_RCCL_H
#define CRP_RCCL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRP_RCCL_ID_BYTES 128

typedef struct crp_rccl *crp_rccl_p;

/* rank 0: a fresh unique id (CRP_RCCL_ID_BYTES bytes) to hand to every rank */
int crp_rccl_get_unique_id(void *id);
/* collective: every rank passes the same id, its rank and the HIP device it has current.  The communicator is created
 * non-blocking and polled under a deadline (CRPSPMM_RCCL_TIMEOUT seconds, default 120): when a peer fails before or inside
 * its own call, the others return an error instead of staying inside ncclCommInitRank, and the caller's control plane can
 * agree on a fallback.  Replaces the MPI_Comm_split of /root/reference/src/para2d_spmm.c:41-43 for device payloads. */
int crp_rccl_create(const void *id, int nranks, int rank, crp_rccl_p *out);
/* wall time crp_rccl_create() took on this rank (reported by bench.py) */
double crp_rccl_create_seconds(crp_rccl_p h);
/* 1 when the communicator was created with the classic blocking ncclCommInitRank (CRPSPMM_RCCL_BLOCKING=1: the known
 * fallback; a peer that dies before its own call then leaves the others inside theirs), 0 = non-blocking and polled */
int crp_rccl_is_blocking(crp_rccl_p h);
/* host seconds spent inside the collectives' calls of this handle so far (issuing the groups), and how many calls */
double crp_rccl_issue_seconds(crp_rccl_p h, long long *calls);
int crp_rccl_destroy(crp_rccl_p *h);
int crp_rccl_nranks(crp_rccl_p h);
int crp_rccl_rank(crp_rccl_p h);

/* sparse all-to-all of fp64 elements between device buffers; counts / displs in elements (nranks entries) */
int crp_rccl_alltoallv_f64(crp_rccl_p h, const double *send_dev, const long long *scnts, const long long *sdispls,
                           double *recv_dev, const long long *rcnts, const long long *rdispls, void *stream);
/* all-gather of raw bytes between device buffers: every rank's `sbytes` land at recv_dev + rdispls[q] on all ranks
 * (rbytes[rank] == sbytes); the own piece is a device-to-device copy */
int crp_rccl_allgatherv(crp_rccl_p h, const void *send_dev, size_t sbytes, void *recv_dev, const size_t *rbytes,
                        const size_t *rdispls, void *stream);
/* all-to-all of raw bytes between device buffers (counts / displs in bytes, nranks entries) */
int crp_rccl_alltoallv_bytes(crp_rccl_p h, const void *send_dev, const size_t *scnts, const size_t *sdispls, void *recv_dev,
                             const size_t *rcnts, const size_t *rdispls, void *stream);

/* adaptors for crp_comm_t: ctx = crp_rccl_p; a failing collective aborts (the engines' API returns void) */
void crp_rccl_comm_alltoallv_dev_f64(void *ctx, const double *send_dev, const long long *scnts, const long long *sdispls,
                                     double *recv_dev, const long long *rcnts, const long long *rdispls, void *stream);
void crp_rccl_comm_allgatherv_dev(void *ctx, const void *send_dev, size_t sbytes, void *recv_dev, const size_t *rbytes,
                                  const size_t *rdispls, void *stream);

#ifdef __cplusplus
}
#endif
#endif
