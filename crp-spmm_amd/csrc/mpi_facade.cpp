// mpi_facade.cpp -- lib/libcrpspmm.so: the reference's MPI-typed API
// (include/rowpara_spmm.h, include/para2d_spmm.h) on top of the communicator-
// agnostic engines (include/crp_engine.h).
//
// crp_comm_t over MPI: the control-plane members call the same MPI routines the
// reference calls (src/rowpara_spmm.c:154-162,439-442; src/para2d_spmm.c:41-83).
// The per-multiply B exchange and the replication of an A row panel move DEVICE buffers.  With one GPU per
// rank they are RCCL (include/crp_rccl.h): the RCCL communicator is bootstrapped over the MPI one (unique id
// broadcast by rank 0) when the MPI communicator is wrapped -- i.e. inside the collective *_init call, never
// inside exec --, and the collectives are groups of ncclSend / ncclRecv on the caller's stream: device to device
// over xGMI, no host copy, asynchronous.  When ranks of one NODE share a GPU (RCCL refuses that; bus ids are only
// compared among the ranks MPI_Comm_split_type(SHARED) puts together, they repeat across nodes), when RCCL
// cannot be set up, or with CRPSPMM_EXCHANGE=host, the payload is staged through host buffers and plain MPI
// point-to-point in the reference's ring order (src/rowpara_spmm.c:275-303).  Rank 0 says once which it is.
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <mpi.h>
#include "crp_engine.h"
#include "crp_rccl.h"
#include "crpspmm_hip.h"
#include "mat_redist.h"
#include "para2d_spmm.h"
#include "crpspmm.h"
#include "rowpara_spmm.h"
#include "utils.h"
#include "knobs.h"

namespace {

struct MpiCtx
{
    MPI_Comm comm;
    bool     owned;
    int      p2p;
    std::vector<double> hsend, hrecv;   // host staging of the device exchange
    crp_rccl_p rccl = nullptr;          // device collectives; NULL = host staging
};

// Collective over x->comm (called from wrap(), i.e. inside the *_init calls): decide how device payloads travel.
void rccl_setup(MpiCtx *x)
{
    int P, me;
    MPI_Comm_size(x->comm, &P);
    MPI_Comm_rank(x->comm, &me);
    int want = !crp::knobs().exchange_host;
    // every rank needs a GPU of its own: compare the PCI bus ids of the ranks of this node
    char mine[64] = {0};
    if (crp_hip_device_bus_id(mine, sizeof(mine)) != 0) want = 0;
    {
        MPI_Comm node;
        MPI_Comm_split_type(x->comm, MPI_COMM_TYPE_SHARED, me, MPI_INFO_NULL, &node);
        int np = 1;
        MPI_Comm_size(node, &np);
        std::vector<char> all((size_t) np * 64, 0);
        MPI_Allgather(mine, 64, MPI_CHAR, all.data(), 64, MPI_CHAR, node);
        for (int a = 0; a < np && want; a++)
            for (int b = a + 1; b < np; b++)
                if (strncmp(&all[(size_t) a * 64], &all[(size_t) b * 64], 64) == 0) want = 0;
        MPI_Comm_free(&node);
    }
    int all_want = 0;
    MPI_Allreduce(&want, &all_want, 1, MPI_INT, MPI_MIN, x->comm);
    if (all_want)
    {
        char id[CRP_RCCL_ID_BYTES];
        memset(id, 0, sizeof(id));
        int ok = 1;
        if (me == 0) ok = (crp_rccl_get_unique_id(id) == 0);
        MPI_Bcast(&ok, 1, MPI_INT, 0, x->comm);
        if (ok)
        {
            MPI_Bcast(id, (int) sizeof(id), MPI_BYTE, 0, x->comm);
            ok = (crp_rccl_create(id, P, me, &x->rccl) == 0);
            int all_ok = 0;
            MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, x->comm);
            if (!all_ok) crp_rccl_destroy(&x->rccl);
        }
    }
    static bool said = false;
    int wrank = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &wrank);
    if (!said && wrank == 0 && P > 1)
    {
        said = true;
        printf("[INFO] crpspmm device payloads: %s (%d ranks)\n", x->rccl ? "RCCL, device to device" : "host staging + MPI", P);
        fflush(stdout);
    }
}

void m_alltoall(void *c, const int *s, int *r, int count)
{
    MPI_Alltoall(s, count, MPI_INT, r, count, MPI_INT, ((MpiCtx *) c)->comm);
}
void m_alltoallv(void *c, const int *s, const int *sc, const int *sd, int *r, const int *rc, const int *rd)
{
    MPI_Alltoallv(s, sc, sd, MPI_INT, r, rc, rd, MPI_INT, ((MpiCtx *) c)->comm);
}
void m_allgatherv(void *c, const void *s, size_t sb, void *r, const size_t *rb, const size_t *rd)
{
    MpiCtx *x = (MpiCtx *) c;
    int P, me;
    MPI_Comm_size(x->comm, &P);
    MPI_Comm_rank(x->comm, &me);
    bool small = true;
    for (int i = 0; i < P; i++) small = small && rb[i] <= (size_t) INT_MAX && rd[i] <= (size_t) INT_MAX;
    if (small)
    {
        std::vector<int> cnt(P), dsp(P);
        for (int i = 0; i < P; i++) { cnt[i] = (int) rb[i]; dsp[i] = (int) rd[i]; }
        MPI_Allgatherv(s, (int) sb, MPI_BYTE, r, cnt.data(), dsp.data(), MPI_BYTE, x->comm);
        return;
    }
    // pieces or displacements beyond what MPI's int counts carry (an nlpkkt240 value slice is 3 GB): one
    // broadcast per source, in chunks
    const size_t chunk = (size_t) 1 << 30;
    if (sb > 0) memcpy((char *) r + rd[me], s, sb);
    for (int q = 0; q < P; q++)
        for (size_t off = 0; off < rb[q]; off += chunk)
        {
            const size_t len = rb[q] - off < chunk ? rb[q] - off : chunk;
            MPI_Bcast((char *) r + rd[q] + off, (int) len, MPI_BYTE, q, x->comm);
        }
}
void m_barrier(void *c) { MPI_Barrier(((MpiCtx *) c)->comm); }
void m_red_f64(void *c, const double *in, double *out, int n, int op)
{
    MPI_Reduce(in, out, n, MPI_DOUBLE, op == CRP_OP_MAX ? MPI_MAX : MPI_SUM, 0, ((MpiCtx *) c)->comm);
}
void m_red_u64(void *c, const uint64_t *in, uint64_t *out, int n, int op)
{
    MPI_Reduce(in, out, n, MPI_UINT64_T, op == CRP_OP_MAX ? MPI_MAX : MPI_SUM, 0, ((MpiCtx *) c)->comm);
}

void m_alltoallv_dev(void *c, const double *send_dev, const long long *sc, const long long *sd, double *recv_dev,
                     const long long *rc, const long long *rd, void *stream)
{
    MpiCtx *x = (MpiCtx *) c;
    int P, me;
    MPI_Comm_size(x->comm, &P);
    MPI_Comm_rank(x->comm, &me);
    if (x->rccl != nullptr)
    {
        const int r = crp_rccl_alltoallv_f64(x->rccl, send_dev, sc, sd, recv_dev, rc, rd, stream);
        ASSERT_PRINTF(r == 0, "RCCL exchange failed (%d)\n", r);
        return;
    }
    const long long ns = sd[P], nr = rd[P];
    if ((long long) x->hsend.size() < ns) x->hsend.resize((size_t) ns);
    if ((long long) x->hrecv.size() < nr) x->hrecv.resize((size_t) nr);
    if (ns > 0)
    {
        int rc_ = crp_dev_memcpy(x->hsend.data(), send_dev, sizeof(double) * (size_t) ns, 1, stream);
        ASSERT_PRINTF(rc_ == 0, "device -> host staging failed (%d)\n", rc_);
    }
    crp_stream_sync(stream);
    std::vector<MPI_Request> reqs;
    reqs.reserve(2 * (size_t) P);
    const long long chunk = INT_MAX / 2;   // counts are int in MPI: split large messages
    for (int i = 1; i < P; i++)
    {
        const int src = (me + i) % P;
        for (long long off = 0; off < rc[src]; off += chunk)
        {
            const int cnt = (int) ((rc[src] - off < chunk) ? rc[src] - off : chunk);
            reqs.emplace_back();
            MPI_Irecv(x->hrecv.data() + rd[src] + off, cnt, MPI_DOUBLE, src, src, x->comm, &reqs.back());
        }
    }
    for (int i = 1; i < P; i++)
    {
        const int dst = (me - i + P) % P;
        for (long long off = 0; off < sc[dst]; off += chunk)
        {
            const int cnt = (int) ((sc[dst] - off < chunk) ? sc[dst] - off : chunk);
            reqs.emplace_back();
            MPI_Isend(x->hsend.data() + sd[dst] + off, cnt, MPI_DOUBLE, dst, me, x->comm, &reqs.back());
        }
    }
    if (sc[me] > 0) memcpy(x->hrecv.data() + rd[me], x->hsend.data() + sd[me], sizeof(double) * (size_t) sc[me]);   // own block
    MPI_Waitall((int) reqs.size(), reqs.data(), MPI_STATUSES_IGNORE);
    if (nr > 0)
    {
        int rc_ = crp_dev_memcpy(recv_dev, x->hrecv.data(), sizeof(double) * (size_t) nr, 0, stream);
        ASSERT_PRINTF(rc_ == 0, "host -> device staging failed (%d)\n", rc_);
    }
}

void m_alltoallv_bytes(void *c, const void *s, const size_t *sc, const size_t *sd, void *r, const size_t *rc,
                       const size_t *rd)
{
    MpiCtx *x = (MpiCtx *) c;
    int P;
    MPI_Comm_size(x->comm, &P);
    std::vector<int> isc(P), isd(P), irc(P), ird(P);
    for (int i = 0; i < P; i++)
    {
        ASSERT_PRINTF(sc[i] <= INT_MAX && sd[i] <= INT_MAX && rc[i] <= INT_MAX && rd[i] <= INT_MAX,
                      "alltoallv piece exceeds 2 GiB\n");
        isc[i] = (int) sc[i]; isd[i] = (int) sd[i]; irc[i] = (int) rc[i]; ird[i] = (int) rd[i];
    }
    MPI_Alltoallv(s, isc.data(), isd.data(), MPI_BYTE, r, irc.data(), ird.data(), MPI_BYTE, x->comm);
}

void m_allgatherv_dev(void *c, const void *send_dev, size_t sb, void *recv_dev, const size_t *rb, const size_t *rd, void *stream)
{
    const int r = crp_rccl_allgatherv(((MpiCtx *) c)->rccl, send_dev, sb, recv_dev, rb, rd, stream);
    ASSERT_PRINTF(r == 0, "RCCL all-gather failed (%d)\n", r);
}

crp_comm_t *wrap(MPI_Comm comm, bool owned);

crp_comm_t *m_split(void *c, int color, int key)
{
    MPI_Comm sub;
    MPI_Comm_split(((MpiCtx *) c)->comm, color, key, &sub);
    return wrap(sub, true);
}

void m_free(crp_comm_t *self)
{
    MpiCtx *x = (MpiCtx *) self->ctx;
    crp_rccl_destroy(&x->rccl);
    if (x->owned) MPI_Comm_free(&x->comm);
    delete x;
    free(self);
}

crp_comm_t *wrap(MPI_Comm comm, bool owned)
{
    crp_comm_t *c = (crp_comm_t *) calloc(1, sizeof(crp_comm_t));
    MpiCtx *x = new MpiCtx;
    x->comm = comm;
    x->owned = owned;
    GET_ENV_INT_VAR(x->p2p, "RP_SPMM_P2P", "rB_p2p", 1, 0, 1, 0);
    c->ctx = x;
    MPI_Comm_size(comm, &c->nproc);
    MPI_Comm_rank(comm, &c->rank);
    c->alltoall_i32 = m_alltoall;
    c->alltoallv_i32 = m_alltoallv;
    c->allgatherv_bytes = m_allgatherv;
    c->barrier = m_barrier;
    c->reduce_f64 = m_red_f64;
    c->reduce_u64 = m_red_u64;
    c->alltoallv_dev_f64 = m_alltoallv_dev;
    c->alltoallv_bytes = m_alltoallv_bytes;
    c->split = m_split;
    c->free = m_free;
    rccl_setup(x);
    if (x->rccl != nullptr) c->allgatherv_dev = m_allgatherv_dev;
    return c;
}

// one GPU per rank, chosen from the launcher's local-rank variable
// (same variables as /root/reference/deprecated/src/cuda_proxy.cu:11-46)
void select_device_once()
{
    static bool done = false;
    if (done) return;
    done = true;
    const char *names[] = {"MPI_LOCALRANKID", "MV2_COMM_WORLD_LOCAL_RANK", "OMPI_COMM_WORLD_NODE_RANK",
                           "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", "PBS_O_VNODENUM", "PMI_RANK", "LOCAL_RANK"};
    int local = 0, ndev = 0;
    for (const char *nm : names)
    {
        const char *v = getenv(nm);
        if (v != NULL) { local = atoi(v); break; }
    }
    if (crp_hip_device_count(&ndev) == 0 && ndev > 0) crp_hip_set_device(local % ndev);
}

struct RpGlue
{
    crp_rp_spmm_p eng = nullptr;
    crp_comm_t   *comm = nullptr;   // wrapper we own (not the MPI_Comm itself)
    bool          own_eng = true;
    std::vector<int> scnts, sdispls, rcnts, rdispls;
};

int sat(long long v) { return v > INT_MAX ? INT_MAX : (int) v; }

void sync_public(rp_spmm_p s)
{
    RpGlue *g = (RpGlue *) s->impl;
    crp_rp_plan_view_t v;
    crp_rp_spmm_get_plan(g->eng, &v);
    s->nproc = v.nproc; s->my_rank = v.my_rank; s->glb_n = v.glb_n; s->A_nrow = v.A_nrow; s->rB_nrow = v.rB_nrow;
    s->rB_self_src_offset = v.rB_self_src_offset; s->rB_self_dst_offset = v.rB_self_dst_offset;
    s->rB_self_nrow = v.rB_self_nrow; s->rB_p2p = v.rB_p2p; s->rB_reidx = v.rB_reidx;
    s->A_rowptr = (int *) v.A_rowptr; s->A_colidx = (int *) v.A_colidx; s->A_val = (double *) v.A_val;
    s->rB_self_src_ridxs = (int *) v.rB_self_src_ridxs;
    s->rB_sridxs = (int *) v.rB_sridxs; s->rB_rridxs = (int *) v.rB_rridxs;
    const int P = v.nproc;
    g->scnts.resize(P); g->rcnts.resize(P); g->sdispls.resize(P + 1); g->rdispls.resize(P + 1);
    for (int q = 0; q < P; q++) { g->scnts[q] = sat(v.rB_scnts[q]); g->rcnts[q] = sat(v.rB_rcnts[q]); }
    for (int q = 0; q <= P; q++) { g->sdispls[q] = sat(v.rB_sdispls[q]); g->rdispls[q] = sat(v.rB_rdispls[q]); }
    s->rB_scnts = g->scnts.data(); s->rB_sdispls = g->sdispls.data();
    s->rB_rcnts = g->rcnts.data(); s->rB_rdispls = g->rdispls.data();
    s->rB_recv_size = v.rB_recv_size; s->n_exec = v.n_exec;
    s->t_init = v.t_init; s->t_pack = v.t_pack; s->t_a2a = v.t_a2a; s->t_unpack = v.t_unpack;
    s->t_spmm = v.t_spmm; s->t_exec = v.t_exec;
}

struct RdGlue
{
    crp_mat_redist_p eng = nullptr;
    crp_comm_t      *comm = nullptr;
};

void rd_sync_public(mat_redist_engine_p s)
{
    RdGlue *g = (RdGlue *) s->impl;
    crp_mat_redist_view_t v;
    crp_mat_redist_get_view(g->eng, &v);
    s->nproc = v.nproc; s->rank = v.rank;
    s->src_srow = v.src_srow; s->src_scol = v.src_scol; s->src_nrow = v.src_nrow; s->src_ncol = v.src_ncol;
    s->req_srow = v.req_srow; s->req_scol = v.req_scol; s->req_nrow = v.req_nrow; s->req_ncol = v.req_ncol;
    s->n_proc_send = v.n_proc_send; s->n_proc_recv = v.n_proc_recv; s->send_cnt = v.send_cnt; s->recv_cnt = v.recv_cnt;
    s->send_ranks = (int *) v.send_ranks; s->send_sizes = (int *) v.send_sizes;
    s->send_displs = (int *) v.send_displs; s->sblk_sizes = (int *) v.sblk_sizes;
    s->recv_ranks = (int *) v.recv_ranks; s->recv_sizes = (int *) v.recv_sizes;
    s->recv_displs = (int *) v.recv_displs; s->rblk_sizes = (int *) v.rblk_sizes;
    s->hd_trans_ms = v.hd_trans_ms;
}

struct P2dGlue
{
    crp_para2d_spmm_p eng = nullptr;
    crp_comm_t       *comm = nullptr;
};

struct CeGlue
{
    crp_crpspmm_p eng = nullptr;
    crp_comm_t   *comm = nullptr;
};

void ce_sync_public(crpspmm_engine_p s)
{
    CeGlue *g = (CeGlue *) s->impl;
    crp_crpspmm_view_t v;
    crp_crpspmm_get_view(g->eng, &v);
    s->np_glb = v.np_glb; s->rank_glb = v.rank_glb; s->np_row = v.np_row; s->np_col = v.np_col;
    s->rank_row = v.rank_row; s->rank_col = v.rank_col; s->glb_m = v.glb_m; s->glb_n = v.glb_n; s->glb_k = v.glb_k;
    s->loc_A_srow = v.loc_A_srow; s->loc_A_erow = v.loc_A_erow; s->loc_A_nrow = v.loc_A_nrow;
    s->loc_A_nnz = v.loc_A_nnz; s->loc_A_nnz_s = v.loc_A_nnz_s;
    s->rd_B_srow = v.rd_B_srow; s->rd_B_erow = v.rd_B_erow;
    s->loc_B_srow = v.loc_B_srow; s->loc_B_erow = v.loc_B_erow; s->loc_B_nrow = v.loc_B_nrow;
    s->a2a_B_finegrain = v.a2a_B_finegrain;
    s->loc_B_scol = v.loc_B_scol; s->loc_B_ecol = v.loc_B_ecol; s->loc_B_ncol = v.loc_B_ncol;
    s->loc_A_rowptr = (int *) v.loc_A_rowptr; s->loc_A_colidx = (int *) v.loc_A_colidx;
    s->loc_A_val = (double *) v.loc_A_val; s->red_B = (double *) v.red_B; s->loc_C = (double *) v.loc_C;
    s->n_exec = v.n_exec; s->t_init = v.t_init; s->t_exec = v.t_exec; s->t_rd_A = v.t_rd_A; s->t_agv_A = v.t_agv_A;
    s->t_rd_B = v.t_rd_B; s->t_a2a_B = v.t_a2a_B; s->t_spmm = v.t_spmm; s->t_rd_C = v.t_rd_C; s->t_exec_nr = v.t_exec_nr;
    s->nelem_A_rd = v.nelem_A_rd; s->nelem_A_agv = v.nelem_A_agv; s->nelem_B_rd = v.nelem_B_rd;
    s->nelem_B_a2av = v.nelem_B_a2av; s->nelem_B_a2av_min = v.nelem_B_a2av_min;
}

}  // namespace

extern "C" {

void crpspmm_engine_init(const int m, const int n, const int k, const int src_A_srow, const int src_A_nrow,
                         const int *src_A_rowptr, const int *src_A_colidx, const int src_B_srow, const int src_B_nrow,
                         const int src_B_scol, const int src_B_ncol, const int dst_C_srow, const int dst_C_nrow,
                         const int dst_C_scol, const int dst_C_ncol, MPI_Comm comm, int use_CUDA,
                         crpspmm_engine_p *engine_, size_t *workbuf_bytes)
{
    select_device_once();
    crpspmm_engine_p s = (crpspmm_engine_p) calloc(1, sizeof(crpspmm_engine_s));
    CeGlue *g = new CeGlue;
    g->comm = wrap(comm, false);
    crp_crpspmm_init(m, n, k, src_A_srow, src_A_nrow, src_A_rowptr, src_A_colidx, src_B_srow, src_B_nrow, src_B_scol,
                     src_B_ncol, dst_C_srow, dst_C_nrow, dst_C_scol, dst_C_ncol, g->comm, &g->eng);
    s->impl = g;
    s->comm_glb = comm;
    s->comm_row = MPI_COMM_NULL;
    s->comm_col = MPI_COMM_NULL;    // owned by the engine's communicator wrapper
    s->use_CUDA = use_CUDA;
    s->alloc_workbuf = 1;
    if (workbuf_bytes != NULL) *workbuf_bytes = 0;
    ce_sync_public(s);
    *engine_ = s;
}

void crpspmm_engine_attach_workbuf(crpspmm_engine_p engine, double *workbuf)
{
    (void) engine;
    (void) workbuf;                 // the engine owns its buffers (see crpspmm.h)
}

void crpspmm_engine_exec(crpspmm_engine_p s, const int *src_A_rowptr, const int *src_A_colidx, const double *src_A_val,
                         const double *src_B, const int ldB, double *dst_C, const int ldC)
{
    if (s == NULL) return;
    crp_crpspmm_exec(((CeGlue *) s->impl)->eng, src_A_rowptr, src_A_colidx, src_A_val, src_B, ldB, dst_C, ldC);
    ce_sync_public(s);
}

void crpspmm_engine_free(crpspmm_engine_p *engine_)
{
    if (engine_ == NULL || *engine_ == NULL) return;
    crpspmm_engine_p s = *engine_;
    CeGlue *g = (CeGlue *) s->impl;
    crp_crpspmm_free(&g->eng);
    if (g->comm) g->comm->free(g->comm);
    delete g;
    free(s);
    *engine_ = NULL;
}

void crpspmm_engine_print_stat(crpspmm_engine_p s)
{
    if (s == NULL) return;
    crp_crpspmm_print_stat(((CeGlue *) s->impl)->eng);
}

void crpspmm_engine_clear_stat(crpspmm_engine_p s)
{
    if (s == NULL) return;
    crp_crpspmm_clear_stat(((CeGlue *) s->impl)->eng);
    ce_sync_public(s);
}

crp_comm_t *crp_mpi_comm_wrap(MPI_Comm comm)
{
    select_device_once();
    return wrap(comm, false);
}

int crp_mpi_comm_uses_rccl(crp_comm_t *c)
{
    return (c != NULL && ((MpiCtx *) c->ctx)->rccl != nullptr) ? 1 : 0;
}

void rp_spmm_init(const int A_srow, const int A_nrow, const int *A_rowptr, const int *A_colidx, const double *A_val,
                  const int *B_row_displs, const int glb_n, MPI_Comm comm, rp_spmm_p *rp_spmm)
{
    select_device_once();
    rp_spmm_p s = (rp_spmm_p) calloc(1, sizeof(rp_spmm_s));
    RpGlue *g = new RpGlue;
    g->comm = wrap(comm, false);
    crp_rp_spmm_init(A_srow, A_nrow, A_rowptr, A_colidx, A_val, B_row_displs, glb_n, g->comm, &g->eng);
    s->impl = g;
    s->comm = comm;
    sync_public(s);
    *rp_spmm = s;
}

void rp_spmm_free(rp_spmm_p *rp_spmm)
{
    if (rp_spmm == NULL || *rp_spmm == NULL) return;
    rp_spmm_p s = *rp_spmm;
    RpGlue *g = (RpGlue *) s->impl;
    if (g->own_eng) crp_rp_spmm_free(&g->eng);
    if (g->comm) g->comm->free(g->comm);
    delete g;
    free(s);
    *rp_spmm = NULL;
}

void rp_spmm_exec(rp_spmm_p s, const int BC_layout, const double *B, const int ldB, double *C, const int ldC)
{
    if (s == NULL) return;
    RpGlue *g = (RpGlue *) s->impl;
    crp_rp_spmm_exec(g->eng, BC_layout, B, ldB, C, ldC);
    sync_public(s);
}

void rp_spmm_print_stat(rp_spmm_p s)
{
    if (s == NULL) return;
    crp_rp_spmm_print_stat(((RpGlue *) s->impl)->eng);
}

void rp_spmm_clear_stat(rp_spmm_p s)
{
    if (s == NULL) return;
    crp_rp_spmm_clear_stat(((RpGlue *) s->impl)->eng);
    sync_public(s);
}

void mat_redist_engine_init(const int src_srow, const int src_scol, const int src_nrow, const int src_ncol,
                            const int req_srow, const int req_scol, const int req_nrow, const int req_ncol,
                            MPI_Comm comm, MPI_Datatype dtype, const size_t dt_size, dev_type_t dev_type,
                            mat_redist_engine_p *engine_, size_t *workbuf_bytes)
{
    if (is_dev_type_valid(dev_type) == 0)
    {
        ERROR_PRINTF("Invalid device type %d\n", dev_type);
        return;
    }
    if (dev_type != DEV_TYPE_HOST) select_device_once();
    mat_redist_engine_p s = (mat_redist_engine_p) calloc(1, sizeof(mat_redist_engine_s));
    RdGlue *g = new RdGlue;
    g->comm = wrap(comm, false);
    crp_mat_redist_init(src_srow, src_scol, src_nrow, src_ncol, req_srow, req_scol, req_nrow, req_ncol, g->comm,
                        dt_size, (int) dev_type, &g->eng, workbuf_bytes);
    if (g->eng == NULL)
    {
        g->comm->free(g->comm);
        delete g;
        free(s);
        return;
    }
    s->impl = g;
    s->graph_comm = MPI_COMM_NULL;
    s->dtype = dtype;
    s->dt_size = dt_size;
    s->dev_type = dev_type;
    s->alloc_workbuf = (workbuf_bytes == NULL) ? 1 : 0;
    rd_sync_public(s);
    *engine_ = s;
}

void mat_redist_engine_attach_workbuf(mat_redist_engine_p s, void *workbuf_h, void *workbuf_d)
{
    if (s == NULL)
    {
        WARNING_PRINTF("mat_redist_engine not initialized\n");
        return;
    }
    crp_mat_redist_attach_workbuf(((RdGlue *) s->impl)->eng, workbuf_h, workbuf_d);
    s->workbuf_h = workbuf_h;
    s->workbuf_d = workbuf_d;
}

void mat_redist_engine_exec(mat_redist_engine_p s, const void *src_blk, const int src_ld, void *dst_blk,
                            const int dst_ld)
{
    if (s == NULL)
    {
        WARNING_PRINTF("mat_redist_engine not initialized\n");
        return;
    }
    crp_mat_redist_exec(((RdGlue *) s->impl)->eng, src_blk, src_ld, dst_blk, dst_ld);
    rd_sync_public(s);
}

void mat_redist_engine_free(mat_redist_engine_p *engine_)
{
    if (engine_ == NULL || *engine_ == NULL) return;
    mat_redist_engine_p s = *engine_;
    RdGlue *g = (RdGlue *) s->impl;
    crp_mat_redist_free(&g->eng);
    if (g->comm) g->comm->free(g->comm);
    delete g;
    free(s);
    *engine_ = NULL;
}

void para2d_spmm_init(MPI_Comm comm, const int pm, const int pn, const int *A0_rowptr, const int *B_rowptr,
                      const int *AC_rowptr, const int *BC_colptr, const int *A_rowptr, const int *A_colidx,
                      const double *A_val, para2d_spmm_p *para2d_spmm)
{
    select_device_once();
    para2d_spmm_p s = (para2d_spmm_p) calloc(1, sizeof(para2d_spmm_s));
    P2dGlue *g = new P2dGlue;
    g->comm = wrap(comm, false);
    crp_para2d_spmm_init(g->comm, pm, pn, A0_rowptr, B_rowptr, AC_rowptr, BC_colptr, A_rowptr, A_colidx, A_val,
                         &g->eng);
    s->impl = g;
    s->comm_glb = comm;
    s->comm_col = MPI_COMM_NULL;   // owned by the engine's communicator wrapper
    s->rA_cost = crp_para2d_spmm_rA_cost(g->eng);
    s->t_ag_A = crp_para2d_spmm_t_ag_A(g->eng);
    // public 1D view over the engine's inner rp engine (not owned by the view)
    rp_spmm_p v = (rp_spmm_p) calloc(1, sizeof(rp_spmm_s));
    RpGlue *rg = new RpGlue;
    rg->eng = crp_para2d_spmm_rp(g->eng);
    rg->own_eng = false;
    v->impl = rg;
    v->comm = MPI_COMM_NULL;
    sync_public(v);
    s->rp_spmm = v;
    *para2d_spmm = s;
}

void para2d_spmm_free(para2d_spmm_p *para2d_spmm)
{
    if (para2d_spmm == NULL || *para2d_spmm == NULL) return;
    para2d_spmm_p s = *para2d_spmm;
    P2dGlue *g = (P2dGlue *) s->impl;
    rp_spmm_free(&s->rp_spmm);
    crp_para2d_spmm_free(&g->eng);
    if (g->comm) g->comm->free(g->comm);
    delete g;
    free(s);
    *para2d_spmm = NULL;
}

void para2d_spmm_exec(para2d_spmm_p s, const int BC_layout, const double *B, const int ldB, double *C, const int ldC)
{
    if (s == NULL) return;
    crp_para2d_spmm_exec(((P2dGlue *) s->impl)->eng, BC_layout, B, ldB, C, ldC);
    sync_public(s->rp_spmm);
}

void para2d_spmm_print_stat(para2d_spmm_p s)
{
    if (s == NULL) return;
    crp_para2d_spmm_print_stat(((P2dGlue *) s->impl)->eng);
}

void para2d_spmm_clear_stat(para2d_spmm_p s)
{
    if (s == NULL) return;
    crp_para2d_spmm_clear_stat(((P2dGlue *) s->impl)->eng);
    sync_public(s->rp_spmm);
}

}  // extern "C"
