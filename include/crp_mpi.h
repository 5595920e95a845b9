/*
 * crp_mpi.h -- MPI back end of the communicator table (crp_comm.h), for C / C++ callers that hold an
 * MPI_Comm and want the engines of crp_engine.h directly (device operands, caller's stream, value
 * updates ...) instead of the reference-typed facade.  Exported by libcrpspmm.so.
 * Control-plane members call MPI; the device all-to-all is RCCL (grouped ncclSend / ncclRecv over
 * xGMI on the caller's stream) when every rank has a GPU of its own, host-staged MPI otherwise or
 * with CRPSPMM_EXCHANGE=host.  The communicator is not duplicated and must outlive the wrapper;
 * release with comm->free(comm).
 */
#ifndef CRP_MPI_H
#define CRP_MPI_H
#include <mpi.h>
#include "crp_comm.h"

#ifdef __cplusplus
extern "C" {
#endif
crp_comm_t *crp_mpi_comm_wrap(MPI_Comm comm);
/* collective over the communicator: 1 when device payloads travel by RCCL, 0 when they are host-staged */
int crp_mpi_comm_uses_rccl(crp_comm_t *comm);
#ifdef __cplusplus
}
#endif
#endif
