// crp_rccl.cpp -- device-payload collectives on RCCL (include/crp_rccl.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "crp_rccl.h"
#include "knobs.h"

static_assert(sizeof(ncclUniqueId) <= CRP_RCCL_ID_BYTES, "unique id does not fit CRP_RCCL_ID_BYTES");

#include <time.h>
#include <sched.h>

struct crp_rccl
{
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    bool blocking = false;          // created with the classic blocking ncclCommInitRank (CRPSPMM_RCCL_BLOCKING=1)
    double create_s = 0.0;          // wall time of the communicator's creation
    double issue_s = 0.0;           // host time inside the collectives' calls (issuing groups), and how many
    long long issue_calls = 0;
};

static double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

// The communicators are NON-BLOCKING (ncclConfig_t::blocking = 0): ncclCommInitRank is itself a collective, and a blocking
// one leaves every healthy rank inside it for good when one peer fails before or during its own call -- the control plane's
// "did everybody succeed" vote is then never reached.  Non-blocking calls return ncclInProgress and are polled here under a
// deadline (CRPSPMM_RCCL_TIMEOUT seconds, default 120 for the creation; the hot-path collectives poll without one, like a
// blocking call would -- spinning, not sleeping: wait_comm).  CRPSPMM_RCCL_BLOCKING=1 selects the classic blocking communicator.
static double rccl_deadline_s() { return crp::knobs().rccl_timeout; }

// CRPSPMM_RCCL_BLOCKING=1: the classic blocking communicator (ncclCommInitRank; every call returns when it is done).  The known
// fallback for a multi-GPU box on which the non-blocking path misbehaves -- at the price named above: a peer that fails
// before its own ncclCommInitRank leaves the others inside theirs.
static bool rccl_blocking_env() { return crp::knobs().rccl_blocking; }

// Waits until the communicator's pending call has finished; deadline_s <= 0: no deadline.
// hot = true (the per-multiply collectives): the group is on the stream within microseconds, and the host has kernels to
// launch behind it -- spin on the status, giving the core away (sched_yield) between polls once a short burst of polls has
// not sufficed; never sleep.  hot = false (creation, finalize: milliseconds to seconds): nap 0.2 ms between polls.
static ncclResult_t wait_comm(ncclComm_t comm, double deadline_s, bool hot)
{
    const double t0 = now_s();
    for (long it = 0;; it++)
    {
        ncclResult_t st = ncclSuccess;
        const ncclResult_t r = ncclCommGetAsyncError(comm, &st);
        if (r != ncclSuccess) return r;
        if (st != ncclInProgress) return st;
        if (deadline_s > 0.0 && now_s() - t0 > deadline_s) return ncclInProgress;
        if (hot)
        {
            if (it >= 64) sched_yield();
        }
        else
        {
            struct timespec nap = {0, 200000};          // 0.2 ms
            nanosleep(&nap, NULL);
        }
    }
}

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

// inside ncclGroupStart .. ncclGroupEnd: a failing call is remembered (`err`) and the loop left; group_end() then CLOSES the
// group (an open group would swallow every later RCCL call of the process), waits -- bounded -- for what the group had
// already queued on a non-blocking communicator, and returns the error.  In-progress is not a failure.
#define RCCL_IN_GROUP(err, expr)                                                                    \
    do                                                                                              \
    {                                                                                               \
        if ((err) == ncclSuccess)                                                                   \
        {                                                                                           \
            ncclResult_t r__ = (expr);                                                              \
            if (r__ != ncclSuccess && r__ != ncclInProgress)                                        \
            {                                                                                       \
                fprintf(stderr, "[crp_rccl] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr, ncclGetErrorString(r__)); \
                (err) = r__;                                                                        \
            }                                                                                       \
        }                                                                                           \
    } while (0)

// ncclGroupEnd; on a non-blocking communicator: poll until the group has been issued to the stream.  err = the first failure
// inside the group, if any.
static int group_end(crp_rccl *h, ncclResult_t err, double t_call0)
{
    ncclResult_t r = ncclGroupEnd();
    if (err != ncclSuccess)
    {
        // whatever part of the group was accepted is still being issued: do not leave the communicator in progress behind
        // the caller's back (its next call would fail with ncclInvalidUsage) -- wait a bounded time, abort past it
        if (!h->blocking && (r == ncclInProgress || r == ncclSuccess))
        {
            const ncclResult_t w = wait_comm(h->comm, 5.0, false);
            if (w == ncclInProgress) { (void) ncclCommAbort(h->comm); h->comm = nullptr; }
        }
        return -(int) err - 1000;
    }
    if (r == ncclInProgress) r = wait_comm(h->comm, 0.0, true);
    h->issue_s += now_s() - t_call0;
    h->issue_calls++;
    if (r != ncclSuccess)
    {
        fprintf(stderr, "[crp_rccl] ncclGroupEnd -> %s\n", ncclGetErrorString(r));
        return -(int) r - 1000;
    }
    return 0;
}

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
    const double t0 = now_s();
    h->blocking = rccl_blocking_env();
    ncclResult_t r;
    if (h->blocking) r = ncclCommInitRank(&h->comm, nranks, u, rank);
    else
    {
        ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
        cfg.blocking = 0;
        r = ncclCommInitRankConfig(&h->comm, nranks, u, rank, &cfg);
        if ((r == ncclSuccess || r == ncclInProgress) && h->comm != nullptr) r = wait_comm(h->comm, rccl_deadline_s(), false);
    }
    if (r != ncclSuccess)
    {
        fprintf(stderr, "[crp_rccl] %s(%d ranks, rank %d) -> %s%s\n", h->blocking ? "ncclCommInitRank" : "ncclCommInitRankConfig", nranks, rank, ncclGetErrorString(r),
                r == ncclInProgress ? " (deadline passed: a peer never arrived)" : "");
        if (h->comm != nullptr) (void) ncclCommAbort(h->comm);
        delete h;
        return -(int) r - 1000;
    }
    h->nranks = nranks;
    h->rank = rank;
    h->create_s = now_s() - t0;
    *out = h;
    return 0;
}

double crp_rccl_create_seconds(crp_rccl_p h) { return h ? h->create_s : -1.0; }
int crp_rccl_is_blocking(crp_rccl_p h) { return h ? (h->blocking ? 1 : 0) : -1; }
double crp_rccl_issue_seconds(crp_rccl_p h, long long *calls)
{
    if (h == NULL) return -1.0;
    if (calls) *calls = h->issue_calls;
    return h->issue_s;
}

int crp_rccl_destroy(crp_rccl_p *h)
{
    if (h == NULL || *h == NULL) return 0;
    if ((*h)->comm)
    {
        if ((*h)->blocking) (void) ncclCommDestroy((*h)->comm);
        else
        {
            // (non-blocking communicator: finalize, wait for it, then destroy)
            ncclResult_t r = ncclCommFinalize((*h)->comm);
            if (r == ncclSuccess || r == ncclInProgress) r = wait_comm((*h)->comm, rccl_deadline_s(), false);
            if (r == ncclSuccess) (void) ncclCommDestroy((*h)->comm);
            else (void) ncclCommAbort((*h)->comm);
        }
    }
    delete *h;
    *h = NULL;
    return 0;
}

int crp_rccl_nranks(crp_rccl_p h) { return h ? h->nranks : -1; }
int crp_rccl_rank(crp_rccl_p h) { return h ? h->rank : -1; }

int crp_rccl_alltoallv_f64(crp_rccl_p h, const double *send, const long long *sc, const long long *sd, double *recv,
                           const long long *rc, const long long *rd, void *stream)
{
    if (h == NULL || h->comm == nullptr) return -1;
    const double t_call0 = now_s();
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
    ncclResult_t err = ncclSuccess;
    for (int i = 1; i < P && err == ncclSuccess; i++)
    {
        const int q = (me + i) % P;         // ring order of the reference's p2p variant (src/rowpara_spmm.c:277-296)
        if (rc[q] > 0) RCCL_IN_GROUP(err, ncclRecv(recv + rd[q], (size_t) rc[q], ncclDouble, q, h->comm, s));
        const int t = (me - i + P) % P;
        if (sc[t] > 0) RCCL_IN_GROUP(err, ncclSend(send + sd[t], (size_t) sc[t], ncclDouble, t, h->comm, s));
    }
    return group_end(h, err, t_call0);
}

int crp_rccl_alltoallv_bytes(crp_rccl_p h, const void *send, const size_t *sc, const size_t *sd, void *recv, const size_t *rc,
                             const size_t *rd, void *stream)
{
    if (h == NULL || h->comm == nullptr) return -1;
    const double t_call0 = now_s();
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
    ncclResult_t err = ncclSuccess;
    for (int i = 1; i < P && err == ncclSuccess; i++)
    {
        const int q = (me + i) % P, t = (me - i + P) % P;
        if (rc[q] > 0) RCCL_IN_GROUP(err, ncclRecv(rb + rd[q], rc[q], ncclChar, q, h->comm, s));
        if (sc[t] > 0) RCCL_IN_GROUP(err, ncclSend(sb + sd[t], sc[t], ncclChar, t, h->comm, s));
    }
    return group_end(h, err, t_call0);
}

int crp_rccl_allgatherv(crp_rccl_p h, const void *send, size_t sbytes, void *recv, const size_t *rbytes, const size_t *rdispls,
                        void *stream)
{
    if (h == NULL || h->comm == nullptr) return -1;
    const double t_call0 = now_s();
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
    ncclResult_t err = ncclSuccess;
    for (int i = 1; i < P && err == ncclSuccess; i++)
    {
        const int q = (me + i) % P, t = (me - i + P) % P;
        if (rbytes[q] > 0) RCCL_IN_GROUP(err, ncclRecv(rb + rdispls[q], rbytes[q], ncclChar, q, h->comm, s));
        if (sbytes > 0) RCCL_IN_GROUP(err, ncclSend(send, sbytes, ncclChar, t, h->comm, s));
    }
    return group_end(h, err, t_call0);
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
