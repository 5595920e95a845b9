// crp_rccl.cpp -- device-payload collectives on RCCL (include/crp_rccl.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "crp_rccl.h"

static_assert(sizeof(ncclUniqueId) <= CRP_RCCL_ID_BYTES, "unique id does not fit CRP_RCCL_ID_BYTES");

struct crp_rccl
{
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
};

#define RCCL_TRY(expr)                                                                              \
    do                                                                                              \
    {                                                                                               \
        ncclResult_t r__ = (expr);                                                                  \
        if (r__ != ncclSuccess)                                                                     \
        {                                                                                           \
            fprintf(stderr, "[crp_rccl] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr, ncclGetErrorString(r__)); \
            return -(int) r__ - 1000;                                                               \
        }                                                                                           \
    } while (0)

extern "C" {

int crp_rccl_get_unique_id(void *id)
{
    if (id == NULL) return -1;
    ncclUniqueId u;
    memset(&u, 0, sizeof(u));
    RCCL_TRY(ncclGetUniqueId(&u));
    memset(id, 0, CRP_RCCL_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return 0;
}

int crp_rccl_create(const void *id, int nranks, int rank, crp_rccl_p *out)
{
    if (id == NULL || out == NULL || nranks < 1 || rank < 0 || rank >= nranks) return -1;
    *out = NULL;
    crp_rccl *h = new (std::nothrow) crp_rccl;
    if (h == NULL) return -3;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = ncclCommInitRank(&h->comm, nranks, u, rank);
    if (r != ncclSuccess)
    {
        fprintf(stderr, "[crp_rccl] ncclCommInitRank(%d ranks, rank %d) -> %s\n", nranks, rank, ncclGetErrorString(r));
        delete h;
        return -(int) r - 1000;
    }
    h->nranks = nranks;
    h->rank = rank;
    *out = h;
    return 0;
}

int crp_rccl_destroy(crp_rccl_p *h)
{
    if (h == NULL || *h == NULL) return 0;
    if ((*h)->comm) (void) ncclCommDestroy((*h)->comm);
    delete *h;
    *h = NULL;
    return 0;
}

int crp_rccl_nranks(crp_rccl_p h) { return h ? h->nranks : -1; }
int crp_rccl_rank(crp_rccl_p h) { return h ? h->rank : -1; }

int crp_rccl_alltoallv_f64(crp_rccl_p h, const double *send, const long long *sc, const long long *sd, double *recv,
                           const long long *rc, const long long *rd, void *stream)
{
    if (h == NULL) return -1;
    const int P = h->nranks, me = h->rank;
    hipStream_t s = (hipStream_t) stream;
    // the own piece never leaves the device
    if (sc[me] > 0)
    {
        if (sc[me] != rc[me]) return -2;
        if (hipMemcpyAsync(recv + rd[me], send + sd[me], sizeof(double) * (size_t) sc[me], hipMemcpyDeviceToDevice, s) != hipSuccess) return -4;
    }
    bool any = false;
    for (int q = 0; q < P; q++)
        if (q != me && (sc[q] > 0 || rc[q] > 0)) any = true;
    if (!any) return 0;
    RCCL_TRY(ncclGroupStart());
    for (int i = 1; i < P; i++)
    {
        const int q = (me + i) % P;         // ring order of the reference's p2p variant (src/rowpara_spmm.c:277-296)
        if (rc[q] > 0) RCCL_TRY(ncclRecv(recv + rd[q], (size_t) rc[q], ncclDouble, q, h->comm, s));
        const int t = (me - i + P) % P;
        if (sc[t] > 0) RCCL_TRY(ncclSend(send + sd[t], (size_t) sc[t], ncclDouble, t, h->comm, s));
    }
    RCCL_TRY(ncclGroupEnd());
    return 0;
}

int crp_rccl_alltoallv_bytes(crp_rccl_p h, const void *send, const size_t *sc, const size_t *sd, void *recv, const size_t *rc,
                             const size_t *rd, void *stream)
{
    if (h == NULL) return -1;
    const int P = h->nranks, me = h->rank;
    hipStream_t s = (hipStream_t) stream;
    const char *sb = (const char *) send;
    char *rb = (char *) recv;
    if (sc[me] > 0)
    {
        if (sc[me] != rc[me]) return -2;
        if (hipMemcpyAsync(rb + rd[me], sb + sd[me], sc[me], hipMemcpyDeviceToDevice, s) != hipSuccess) return -4;
    }
    bool any = false;
    for (int q = 0; q < P; q++)
        if (q != me && (sc[q] > 0 || rc[q] > 0)) any = true;
    if (!any) return 0;
    RCCL_TRY(ncclGroupStart());
    for (int i = 1; i < P; i++)
    {
        const int q = (me + i) % P, t = (me - i + P) % P;
        if (rc[q] > 0) RCCL_TRY(ncclRecv(rb + rd[q], rc[q], ncclChar, q, h->comm, s));
        if (sc[t] > 0) RCCL_TRY(ncclSend(sb + sd[t], sc[t], ncclChar, t, h->comm, s));
    }
    RCCL_TRY(ncclGroupEnd());
    return 0;
}

int crp_rccl_allgatherv(crp_rccl_p h, const void *send, size_t sbytes, void *recv, const size_t *rbytes, const size_t *rdispls,
                        void *stream)
{
    if (h == NULL) return -1;
    const int P = h->nranks, me = h->rank;
    hipStream_t s = (hipStream_t) stream;
    char *rb = (char *) recv;
    if (rbytes[me] != sbytes) return -2;
    if (sbytes > 0 && rb + rdispls[me] != (const char *) send)
        if (hipMemcpyAsync(rb + rdispls[me], send, sbytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return -4;
    if (P == 1) return 0;
    // direct fan-out: every source pushes its piece over P - 1 distinct links at once (a ring all-gather would be
    // bound by one link)
    RCCL_TRY(ncclGroupStart());
    for (int i = 1; i < P; i++)
    {
        const int q = (me + i) % P, t = (me - i + P) % P;
        if (rbytes[q] > 0) RCCL_TRY(ncclRecv(rb + rdispls[q], rbytes[q], ncclChar, q, h->comm, s));
        if (sbytes > 0) RCCL_TRY(ncclSend(send, sbytes, ncclChar, t, h->comm, s));
    }
    RCCL_TRY(ncclGroupEnd());
    return 0;
}

void crp_rccl_comm_alltoallv_dev_f64(void *ctx, const double *send, const long long *sc, const long long *sd, double *recv,
                                     const long long *rc, const long long *rd, void *stream)
{
    const int r = crp_rccl_alltoallv_f64((crp_rccl_p) ctx, send, sc, sd, recv, rc, rd, stream);
    if (r != 0)
    {
        fprintf(stderr, "[FATAL] crp_rccl_alltoallv_f64 failed (%d)\n", r);
        abort();
    }
}

void crp_rccl_comm_allgatherv_dev(void *ctx, const void *send, size_t sbytes, void *recv, const size_t *rbytes,
                                  const size_t *rdispls, void *stream)
{
    const int r = crp_rccl_allgatherv((crp_rccl_p) ctx, send, sbytes, recv, rbytes, rdispls, stream);
    if (r != 0)
    {
        fprintf(stderr, "[FATAL] crp_rccl_allgatherv failed (%d)\n", r);
        abort();
    }
}

}  // extern "C"
