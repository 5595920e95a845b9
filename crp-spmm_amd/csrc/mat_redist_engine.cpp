// mat_redist_engine.cpp -- generic 2D-block redistribution of a dense row-major matrix
// (include/crp_engine.h: crp_mat_redist_*).  Same plan and wire format as
// /root/reference/src/mat_redist.c:44-419 (rectangles in rank order, packed row-major with
// ld = ncol), written against the communicator table instead of a dist-graph MPI communicator:
// the exchange is an all-to-all whose non-neighbours carry zero bytes.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "crp_engine.h"
#include "crpspmm_hip.h"
#include "dev_type.h"
#include "utils.h"

struct crp_mat_redist
{
    crp_comm_t *comm = nullptr;
    int nproc = 0, rank = 0;
    int src[4] = {0, 0, 0, 0}, req[4] = {0, 0, 0, 0};   // srow, scol, nrow, ncol
    size_t dt_size = 0;
    dev_type_t dev_type = DEV_TYPE_HOST;
    std::vector<int> send_ranks, send_sizes, send_displs, sblk_sizes;
    std::vector<int> recv_ranks, recv_sizes, recv_displs, rblk_sizes;
    int send_cnt = 0, recv_cnt = 0;
    bool own_workbuf = false;
    void *workbuf_h = nullptr, *workbuf_d = nullptr;
    char *sendbuf_h = nullptr, *recvbuf_h = nullptr, *sendbuf_d = nullptr, *recvbuf_d = nullptr;
    double hd_trans_ms = 0.0;
};

namespace {

// inclusive segments [s0, e0] and [s1, e1]; empty segments (e < s) intersect nothing
bool seg_overlap(int s0, int e0, int s1, int e1, int *is, int *ie)
{
    if (s0 > e0 || s1 > e1) return false;
    const int lo = s0 > s1 ? s0 : s1, hi = e0 < e1 ? e0 : e1;
    if (lo > hi) return false;
    *is = lo;
    *ie = hi;
    return true;
}

bool on_device(dev_type_t t) { return t == DEV_TYPE_HIP || t == DEV_TYPE_HIP_RCCL; }

}  // namespace

extern "C" {

void crp_mat_redist_init(int src_srow, int src_scol, int src_nrow, int src_ncol, int req_srow, int req_scol,
                         int req_nrow, int req_ncol, crp_comm_t *comm, size_t dt_size, int dev_type_,
                         crp_mat_redist_p *engine_, size_t *workbuf_bytes)
{
    const dev_type_t dev_type = (dev_type_t) dev_type_;
    if (!is_dev_type_valid(dev_type))
    {
        ERROR_PRINTF("Invalid device type %d\n", dev_type_);
        return;
    }
    crp_mat_redist *e = new crp_mat_redist;
    e->comm = comm;
    e->nproc = comm->nproc;
    e->rank = comm->rank;
    e->dt_size = dt_size;
    e->dev_type = dev_type;
    const int mine[8] = {src_srow, src_scol, src_srow + src_nrow - 1, src_scol + src_ncol - 1,
                         req_srow, req_scol, req_srow + req_nrow - 1, req_scol + req_ncol - 1};
    e->src[0] = src_srow; e->src[1] = src_scol; e->src[2] = src_nrow; e->src[3] = src_ncol;
    e->req[0] = req_srow; e->req[1] = req_scol; e->req[2] = req_nrow; e->req[3] = req_ncol;
    const int P = comm->nproc;
    std::vector<int> all((size_t) 8 * P);
    std::vector<size_t> cnt(P, 8 * sizeof(int)), dsp(P);
    for (int q = 0; q < P; q++) dsp[q] = (size_t) q * 8 * sizeof(int);
    comm->allgatherv_bytes(comm->ctx, mine, 8 * sizeof(int), all.data(), cnt.data(), dsp.data());

    int r0, r1, c0, c1;
    e->send_displs.push_back(0);
    e->recv_displs.push_back(0);
    for (int q = 0; q < P; q++)
    {
        const int *o = &all[(size_t) q * 8];
        // my source rectangle against q's request
        if (seg_overlap(mine[0], mine[2], o[4], o[6], &r0, &r1) && seg_overlap(mine[1], mine[3], o[5], o[7], &c0, &c1))
        {
            e->send_ranks.push_back(q);
            const int blk[4] = {r0, c0, r1 - r0 + 1, c1 - c0 + 1};
            e->sblk_sizes.insert(e->sblk_sizes.end(), blk, blk + 4);
            e->send_sizes.push_back(blk[2] * blk[3]);
            e->send_cnt += blk[2] * blk[3];
            e->send_displs.push_back(e->send_cnt);
        }
        // my request against q's source rectangle
        if (seg_overlap(mine[4], mine[6], o[0], o[2], &r0, &r1) && seg_overlap(mine[5], mine[7], o[1], o[3], &c0, &c1))
        {
            e->recv_ranks.push_back(q);
            const int blk[4] = {r0, c0, r1 - r0 + 1, c1 - c0 + 1};
            e->rblk_sizes.insert(e->rblk_sizes.end(), blk, blk + 4);
            e->recv_sizes.push_back(blk[2] * blk[3]);
            e->recv_cnt += blk[2] * blk[3];
            e->recv_displs.push_back(e->recv_cnt);
        }
    }
    const size_t need = dt_size * ((size_t) e->send_cnt + (size_t) e->recv_cnt);
    if (workbuf_bytes != NULL)
    {
        *workbuf_bytes = need;
        e->own_workbuf = false;
    }
    else
    {
        e->own_workbuf = true;
        void *wh = NULL, *wd = NULL;
        if (dev_type == DEV_TYPE_HOST || dev_type == DEV_TYPE_HIP)
        {
            wh = dev_type_malloc(need, DEV_TYPE_HOST);
            if (wh == NULL && need > 0)
            {
                ERROR_PRINTF("Allocate host workbuf failed\n");
                delete e;
                return;
            }
        }
        if (on_device(dev_type))
        {
            wd = dev_type_malloc(need, DEV_TYPE_HIP);
            if (wd == NULL && need > 0)
            {
                ERROR_PRINTF("Allocate device workbuf failed\n");
                dev_type_free(wh, DEV_TYPE_HOST);
                delete e;
                return;
            }
        }
        crp_mat_redist_attach_workbuf(e, wh, wd);
    }
    *engine_ = e;
    comm->barrier(comm->ctx);
}

void crp_mat_redist_attach_workbuf(crp_mat_redist_p e, void *workbuf_h, void *workbuf_d)
{
    if (e == NULL)
    {
        WARNING_PRINTF("mat_redist_engine not initialized\n");
        return;
    }
    e->workbuf_h = workbuf_h;
    e->workbuf_d = workbuf_d;
    const size_t soff = e->dt_size * (size_t) e->send_cnt;
    if (e->dev_type == DEV_TYPE_HOST || e->dev_type == DEV_TYPE_HIP)
    {
        e->sendbuf_h = (char *) workbuf_h;
        e->recvbuf_h = (char *) workbuf_h + soff;
    }
    if (on_device(e->dev_type))
    {
        e->sendbuf_d = (char *) workbuf_d;
        e->recvbuf_d = (char *) workbuf_d + soff;
    }
}

void crp_mat_redist_exec(crp_mat_redist_p e, const void *src_blk, int src_ld, void *dst_blk, int dst_ld)
{
    if (e == NULL)
    {
        WARNING_PRINTF("mat_redist_engine not initialized\n");
        return;
    }
    const size_t dt = e->dt_size;
    const bool dev = on_device(e->dev_type);
    e->hd_trans_ms = 0.0;
    // pack (src/mat_redist.c:325-348): rectangle i -> contiguous run at send_displs[i]
    char *sbuf = dev ? e->sendbuf_d : e->sendbuf_h;
    for (size_t i = 0; i < e->send_ranks.size(); i++)
    {
        const int *b = &e->sblk_sizes[4 * i];
        const char *from = (const char *) src_blk + dt * ((size_t) (b[0] - e->src[0]) * (size_t) src_ld + (size_t) (b[1] - e->src[1]));
        dev_type_copy_matrix(dt, b[2], b[3], from, src_ld, sbuf + dt * (size_t) e->send_displs[i], b[3], e->dev_type);
    }
    // exchange
    const int P = e->nproc;
    std::vector<size_t> sc(P, 0), sd(P, 0), rc(P, 0), rd(P, 0);
    for (size_t i = 0; i < e->send_ranks.size(); i++)
    {
        sc[e->send_ranks[i]] = dt * (size_t) e->send_sizes[i];
        sd[e->send_ranks[i]] = dt * (size_t) e->send_displs[i];
    }
    for (size_t i = 0; i < e->recv_ranks.size(); i++)
    {
        rc[e->recv_ranks[i]] = dt * (size_t) e->recv_sizes[i];
        rd[e->recv_ranks[i]] = dt * (size_t) e->recv_displs[i];
    }
    crp_comm_t *c = e->comm;
    if (e->dev_type == DEV_TYPE_HOST)
        c->alltoallv_bytes(c->ctx, e->sendbuf_h, sc.data(), sd.data(), e->recvbuf_h, rc.data(), rd.data());
    else if (e->dev_type == DEV_TYPE_HIP_RCCL && dt == 8)
    {
        std::vector<long long> lsc(P), lsd(P + 1), lrc(P), lrd(P + 1);
        for (int q = 0; q < P; q++)
        {
            lsc[q] = (long long) (sc[q] / 8); lsd[q] = (long long) (sd[q] / 8);
            lrc[q] = (long long) (rc[q] / 8); lrd[q] = (long long) (rd[q] / 8);
        }
        lsd[P] = e->send_cnt;
        lrd[P] = e->recv_cnt;
        c->alltoallv_dev_f64(c->ctx, (const double *) e->sendbuf_d, lsc.data(), lsd.data(), (double *) e->recvbuf_d,
                             lrc.data(), lrd.data(), NULL);
        (void) crp_stream_sync(NULL);
    }
    else
    {
        // staged: device -> pinned host -> exchange -> device (src/mat_redist.c:362-378)
        char *hs = e->sendbuf_h, *hr = e->recvbuf_h;
        std::vector<char> tmp;
        if (hs == NULL)   // HIP_RCCL with an element size the device exchange does not carry
        {
            tmp.resize(dt * ((size_t) e->send_cnt + (size_t) e->recv_cnt) + 1);
            hs = tmp.data();
            hr = tmp.data() + dt * (size_t) e->send_cnt;
        }
        double t0 = get_wtime_sec();
        dev_type_memcpy(hs, e->sendbuf_d, dt * (size_t) e->send_cnt, DEV_TYPE_HOST, DEV_TYPE_HIP);
        e->hd_trans_ms += 1000.0 * (get_wtime_sec() - t0);
        c->alltoallv_bytes(c->ctx, hs, sc.data(), sd.data(), hr, rc.data(), rd.data());
        t0 = get_wtime_sec();
        dev_type_memcpy(e->recvbuf_d, hr, dt * (size_t) e->recv_cnt, DEV_TYPE_HIP, DEV_TYPE_HOST);
        e->hd_trans_ms += 1000.0 * (get_wtime_sec() - t0);
    }
    // unpack (src/mat_redist.c:389-416)
    const char *rbuf = dev ? e->recvbuf_d : e->recvbuf_h;
    for (size_t i = 0; i < e->recv_ranks.size(); i++)
    {
        const int *b = &e->rblk_sizes[4 * i];
        char *to = (char *) dst_blk + dt * ((size_t) (b[0] - e->req[0]) * (size_t) dst_ld + (size_t) (b[1] - e->req[1]));
        dev_type_copy_matrix(dt, b[2], b[3], rbuf + dt * (size_t) e->recv_displs[i], b[3], to, dst_ld, e->dev_type);
    }
    c->barrier(c->ctx);
}

void crp_mat_redist_free(crp_mat_redist_p *engine_)
{
    if (engine_ == NULL || *engine_ == NULL) return;
    crp_mat_redist *e = *engine_;
    if (e->own_workbuf)
    {
        dev_type_free(e->workbuf_h, DEV_TYPE_HOST);
        if (e->workbuf_d) dev_type_free(e->workbuf_d, DEV_TYPE_HIP);
    }
    delete e;
    *engine_ = NULL;
}

void crp_mat_redist_get_view(crp_mat_redist_p e, crp_mat_redist_view_t *v)
{
    if (e == NULL || v == NULL) return;
    v->nproc = e->nproc; v->rank = e->rank;
    v->src_srow = e->src[0]; v->src_scol = e->src[1]; v->src_nrow = e->src[2]; v->src_ncol = e->src[3];
    v->req_srow = e->req[0]; v->req_scol = e->req[1]; v->req_nrow = e->req[2]; v->req_ncol = e->req[3];
    v->n_proc_send = (int) e->send_ranks.size(); v->n_proc_recv = (int) e->recv_ranks.size();
    v->send_cnt = e->send_cnt; v->recv_cnt = e->recv_cnt;
    v->send_ranks = e->send_ranks.data(); v->send_sizes = e->send_sizes.data();
    v->send_displs = e->send_displs.data(); v->sblk_sizes = e->sblk_sizes.data();
    v->recv_ranks = e->recv_ranks.data(); v->recv_sizes = e->recv_sizes.data();
    v->recv_displs = e->recv_displs.data(); v->rblk_sizes = e->rblk_sizes.data();
    v->dt_size = e->dt_size; v->dev_type = (int) e->dev_type; v->hd_trans_ms = e->hd_trans_ms;
}

}  // extern "C"
