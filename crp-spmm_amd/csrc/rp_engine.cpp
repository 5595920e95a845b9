// rp_engine.cpp -- the 1D row-parallel SpMM engine (include/crp_engine.h).
//
// Host side of /root/reference/src/rowpara_spmm.c re-thought for a device-
// resident data path:
//   init  (reference :20-190): one pass builds the needed-row flags, a prefix
//         rank array gives the compact ids, the needs are exchanged through the
//         communicator's alltoall(v); A is uploaded once with a two-source column
//         index (local B row | row of the receive buffer), so exec never copies
//         locally owned B rows and never unpacks.
//   exec  (reference :212-422): gather kernel -> device all-to-all -> SpMM kernel
//         on one stream; no allocation, no sparse-handle creation.
// Public plan fields keep the reference's meaning (crp_rp_plan_view_t).
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "crp_engine.h"
#include "crpspmm_hip.h"
#include "utils.h"
#include "par.h"
#include <algorithm>

struct crp_rp_spmm
{
    // ---- plan, reference field names (src/rowpara_spmm.h:8-40)
    int nproc = 1, my_rank = 0, glb_n = 0, A_nrow = 0, rB_nrow = 0;
    int rB_self_src_offset = 0, rB_self_dst_offset = 0, rB_self_nrow = 0;
    int rB_p2p = 1, rB_reidx = 1;
    std::vector<int>       A_rowptr, A_colidx;
    std::vector<double>    A_val;
    std::vector<int>       rB_self_src_ridxs, rB_sridxs, rB_rridxs;
    std::vector<long long> rB_scnts, rB_sdispls, rB_rcnts, rB_rdispls;
    size_t rB_recv_size = 0;
    int    n_exec = 0;
    double t_init = 0, t_pack = 0, t_a2a = 0, t_unpack = 0, t_spmm = 0, t_exec = 0;
    bool vals_from_device = false;   // the device matrices took their values from a device copy (no second upload)
    double t_a2a_host = 0;           // host time inside the exchange CALL (issuing the sends / receives), whatever the timing mode
    crp_comm_t *comm = nullptr;

    // ---- device side
    bool plan_only = false;
    int  timing = 1, variant = 0;
    int  loc_B_nrow = 0;
    long long n_send_rows = 0, n_recv_rows = 0, n_needed_rows = 0;
    std::vector<int> dev_colidx_host;
    crp_csr_dev_p A_dev = nullptr;
    // exchange / compute overlap (nproc > 1): rows with no remote column ("interior") run while the
    // B rows travel, the rest ("boundary") after they have landed; A_dev then stays unset
    crp_csr_dev_p A_int = nullptr, A_bnd = nullptr;
    std::vector<long long> int_src, bnd_src;     // position in A_val of every nonzero of the two parts
    std::vector<double>    split_vals;
    void *xstream = nullptr, *ev_packed = nullptr, *ev_landed = nullptr;
    int    *sridxs_dev = nullptr;
    double *sendbuf_dev = nullptr, *recvbuf_dev = nullptr;
    void   *stream = nullptr;
    // staging (host-pointer API) and column-major temporaries, grown on demand
    double *B_stage = nullptr, *C_stage = nullptr, *B_rm = nullptr, *C_rm = nullptr;
    size_t  B_stage_sz = 0, C_stage_sz = 0, B_rm_sz = 0, C_rm_sz = 0;
    // the stream the last exec ran on, and an event at its end: a value update on the engine's own stream must
    // not overtake kernels of an exec that is still in flight on the caller's stream
    void *ev_exec = nullptr;
    bool  exec_pending = false;
    // last operands seen and where they live (the pointer-attribute query is not free)
    const void *last_B = nullptr, *last_C = nullptr;
    int last_B_dev = 0, last_C_dev = 0;
};

#define HIP_OK(call)                                                              \
    do {                                                                          \
        int rc__ = (call);                                                        \
        ASSERT_PRINTF(rc__ == 0, "%s failed with code %d\n", #call, rc__);        \
    } while (0)

static void grow(double **buf, size_t *cur, size_t need_elems)
{
    if (need_elems <= *cur) return;
    if (*buf) HIP_OK(crp_dev_free(*buf));
    void *p = NULL;
    HIP_OK(crp_dev_malloc(&p, need_elems * sizeof(double)));
    *buf = (double *) p;
    *cur = need_elems;
}


// ---------------------------------------------------------------------------
static void build_plan(crp_rp_spmm *e, int A_nrow, const int *A_rowptr, const int *A_colidx,
                       const double *A_val, const int *B_row_displs, int glb_n, crp_comm_t *comm)
{
    const int P = comm->nproc, me = comm->rank;
    e->nproc = P;
    e->my_rank = me;
    e->glb_n = glb_n;
    e->A_nrow = A_nrow;
    e->comm = comm;
    GET_ENV_INT_VAR(e->rB_p2p, "RP_SPMM_P2P", "rB_p2p", 1, 0, 1, me == 0);
    GET_ENV_INT_VAR(e->rB_reidx, "RP_SPMM_REIDX", "rB_reidx", 1, 0, 1, me == 0);

    const int    base = A_rowptr[0];
    const size_t nnz  = (size_t) ((long long) A_rowptr[A_nrow] - (long long) base);
    const int    glb_k = B_row_displs[P];
    const int    lo = B_row_displs[me], hi = B_row_displs[me + 1];
    e->loc_B_nrow = hi - lo;

    // needed-row flags and the span of touched columns
    std::vector<unsigned char> flag((size_t) glb_k + 1, 0);
    int cmin = INT_MAX, cmax = -1;
    for (size_t p = 0; p < nnz; p++)
    {
        const int c = A_colidx[p];
        ASSERT_PRINTF(c >= 0 && c < glb_k, "column index %d outside B (%d rows)\n", c, glb_k);
        flag[c] = 1;
        if (c < cmin) cmin = c;
        if (c > cmax) cmax = c;
    }
    if (nnz == 0) { cmin = 0; cmax = -1; }

    // rank_[g] = number of needed rows with global index < g
    std::vector<int> rank_((size_t) glb_k + 1);
    int run = 0;
    for (int g = 0; g < glb_k; g++)
    {
        rank_[g] = run;
        run += flag[g];
    }
    rank_[glb_k] = run;
    const int n_needed = run;
    e->n_needed_rows = n_needed;
    e->rB_nrow = e->rB_reidx ? n_needed : (cmax - cmin + 1);

    // rebased / re-indexed copy of A (public plan fields)
    e->A_rowptr.resize((size_t) A_nrow + 1);
    for (int i = 0; i <= A_nrow; i++) e->A_rowptr[i] = A_rowptr[i] - base;
    e->A_colidx.resize(nnz);
    e->A_val.assign(A_val, A_val + nnz);
    for (size_t p = 0; p < nnz; p++)
        e->A_colidx[p] = e->rB_reidx ? rank_[A_colidx[p]] : (A_colidx[p] - cmin);

    // rows served from this rank's own block of B
    const int self_n = rank_[hi] - rank_[lo];
    e->rB_self_nrow = self_n;
    e->rB_self_src_ridxs.clear();
    e->rB_self_src_ridxs.reserve((size_t) self_n);
    for (int g = lo; g < hi; g++)
        if (flag[g]) e->rB_self_src_ridxs.push_back(g);
    if (self_n > 0)
    {
        const int first = e->rB_self_src_ridxs[0];
        e->rB_self_src_offset = first - lo;
        e->rB_self_dst_offset = e->rB_reidx ? rank_[first] : (first - cmin);
    }

    // rows to fetch, grouped by owner (owners hold ascending contiguous ranges,
    // so ascending global order is already owner order)
    std::vector<int> rcnt(P, 0), rdsp(P + 1, 0), need_glb;
    need_glb.reserve((size_t) (n_needed - self_n));
    for (int q = 0; q < P; q++)
    {
        if (q != me)
            for (int g = B_row_displs[q]; g < B_row_displs[q + 1]; g++)
                if (flag[g]) need_glb.push_back(g);
        rdsp[q + 1] = (int) need_glb.size();
        rcnt[q] = rdsp[q + 1] - rdsp[q];
    }
    e->n_recv_rows = rdsp[P];
    e->rB_recv_size = (size_t) rdsp[P];

    // tell every owner which of its rows are wanted (reference :152-165)
    std::vector<int> scnt(P, 0), sdsp(P + 1, 0);
    comm->alltoall_i32(comm->ctx, rcnt.data(), scnt.data(), 1);
    for (int q = 0; q < P; q++) sdsp[q + 1] = sdsp[q] + scnt[q];
    e->n_send_rows = sdsp[P];
    e->rB_sridxs.assign((size_t) sdsp[P] + 1, 0);
    if (need_glb.empty()) need_glb.push_back(0);   // keep .data() valid for empty lists
    comm->alltoallv_i32(comm->ctx, need_glb.data(), rcnt.data(), rdsp.data(), e->rB_sridxs.data(), scnt.data(),
                        sdsp.data());
    e->rB_sridxs.resize((size_t) sdsp[P]);
    for (auto &r : e->rB_sridxs)
    {
        ASSERT_PRINTF(r >= lo && r < hi, "peer requested row %d outside my block [%d, %d)\n", r, lo, hi);
        r -= lo;
    }

    e->rB_rridxs.resize((size_t) rdsp[P]);
    for (int i = 0; i < rdsp[P]; i++)
        e->rB_rridxs[i] = e->rB_reidx ? rank_[need_glb[i]] : (need_glb[i] - cmin);

    e->rB_rcnts.resize(P);
    e->rB_rdispls.resize((size_t) P + 1);
    e->rB_scnts.resize(P);
    e->rB_sdispls.resize((size_t) P + 1);
    for (int q = 0; q < P; q++)
    {
        e->rB_rcnts[q] = (long long) rcnt[q] * glb_n;
        e->rB_scnts[q] = (long long) scnt[q] * glb_n;
    }
    for (int q = 0; q <= P; q++)
    {
        e->rB_rdispls[q] = (long long) rdsp[q] * glb_n;
        e->rB_sdispls[q] = (long long) sdsp[q] * glb_n;
    }

    // device column index: local B row, or ~(position in the receive buffer).
    // Position = rank among needed rows minus the self rows that precede it.
    e->dev_colidx_host.resize(nnz);
    for (size_t p = 0; p < nnz; p++)
    {
        const int g = A_colidx[p];
        if (g >= lo && g < hi) e->dev_colidx_host[p] = g - lo;
        else e->dev_colidx_host[p] = ~(rank_[g] - (g >= hi ? self_n : 0));
    }
}

// Upload A: whole, or split by rows into interior / boundary parts when an exchange exists and
// both parts are worth a launch (CRPSPMM_OVERLAP=0 keeps the single product).
// A_val_dev (optional): the values of the rank's panel, in the panel's order, already in device memory (para2d_engine.cpp: the
// device all-gather of the panel) -- the device matrices then take their values from there, not from a second upload.
static void build_device_matrices(crp_rp_spmm *e, const double *A_val_dev)
{
    const int m = e->A_nrow;
    const int overlap = crp::knobs().overlap;
    std::vector<int> rows_int, rows_bnd;
    if (overlap && e->nproc > 1 && e->n_recv_rows > 0)
    {
        for (int i = 0; i < m; i++)
        {
            bool remote = false;
            for (int p = e->A_rowptr[i]; p < e->A_rowptr[i + 1] && !remote; p++) remote = e->dev_colidx_host[p] < 0;
            (remote ? rows_bnd : rows_int).push_back(i);
        }
    }
    // a part smaller than 1/16 of the rows does not pay for a second launch
    if (rows_int.size() < (size_t) m / 16 || rows_bnd.empty())
    {
        if (A_val_dev != nullptr && !e->A_val.empty())
            HIP_OK(crp_csr_dev_create_dv(m, e->loc_B_nrow, e->A_rowptr.data(), e->dev_colidx_host.data(), e->A_val.data(), A_val_dev, nullptr, &e->A_dev));
        else
            HIP_OK(crp_csr_dev_create(m, e->loc_B_nrow, e->A_rowptr.data(), e->dev_colidx_host.data(), e->A_val.data(), &e->A_dev));
        return;
    }
    auto make = [&](const std::vector<int> &rows, std::vector<long long> &src, crp_csr_dev_p *out) {
        std::vector<int> rp(rows.size() + 1, 0), ci, start(rows.size(), 0);
        std::vector<double> va;
        src.clear();
        for (size_t t = 0; t < rows.size(); t++)
        {
            const int i = rows[t];
            start[t] = e->A_rowptr[i];
            for (int p = e->A_rowptr[i]; p < e->A_rowptr[i + 1]; p++)
            {
                ci.push_back(e->dev_colidx_host[p]);
                va.push_back(e->A_val[p]);
                src.push_back(p);
            }
            rp[t + 1] = (int) ci.size();
        }
        const bool have = !ci.empty();
        if (ci.empty()) { ci.push_back(0); va.push_back(0.0); }
        if (A_val_dev != nullptr && have)
            HIP_OK(crp_csr_dev_create_dv((int) rows.size(), e->loc_B_nrow, rp.data(), ci.data(), va.data(), A_val_dev, start.data(), out));
        else
            HIP_OK(crp_csr_dev_create((int) rows.size(), e->loc_B_nrow, rp.data(), ci.data(), va.data(), out));
        HIP_OK(crp_csr_dev_set_rowmap(*out, rows.data(), m));
    };
    make(rows_int, e->int_src, &e->A_int);
    make(rows_bnd, e->bnd_src, &e->A_bnd);
    HIP_OK(crp_stream_create(&e->xstream));
    HIP_OK(crp_event_create(&e->ev_packed));
    HIP_OK(crp_event_create(&e->ev_landed));
}

static void rp_init_common(int A_nrow, const int *A_rowptr, const int *A_colidx, const double *A_val,
                           const int *B_row_displs, int glb_n, crp_comm_t *comm, crp_rp_spmm_p *out,
                           bool plan_only, const double *A_val_dev = nullptr)
{
    ASSERT_PRINTF(out != NULL && comm != NULL && A_rowptr != NULL && B_row_displs != NULL && A_nrow >= 0 && glb_n >= 0,
                  "invalid arguments to rp_spmm_init\n");
    const double t0 = get_wtime_sec();
    crp_rp_spmm *e = new crp_rp_spmm;
    e->plan_only = plan_only;
    build_plan(e, A_nrow, A_rowptr, A_colidx, A_val, B_row_displs, glb_n, comm);
    if (!plan_only)
    {
        build_device_matrices(e, A_val_dev);
        e->vals_from_device = (A_val_dev != nullptr);
        HIP_OK(crp_stream_create(&e->stream));
        void *p = NULL;
        if (e->n_send_rows > 0)
        {
            HIP_OK(crp_dev_malloc(&p, sizeof(int) * (size_t) e->n_send_rows));
            e->sridxs_dev = (int *) p;
            HIP_OK(crp_dev_memcpy(e->sridxs_dev, e->rB_sridxs.data(), sizeof(int) * (size_t) e->n_send_rows, 0, NULL));
            HIP_OK(crp_dev_malloc(&p, sizeof(double) * (size_t) e->n_send_rows * (size_t) glb_n));
            e->sendbuf_dev = (double *) p;
        }
        if (e->n_recv_rows > 0)
        {
            HIP_OK(crp_dev_malloc(&p, sizeof(double) * (size_t) e->n_recv_rows * (size_t) glb_n));
            e->recvbuf_dev = (double *) p;
        }
        HIP_OK(crp_stream_sync(NULL));
    }
    e->t_init = get_wtime_sec() - t0;
    *out = e;
}

extern "C" {

void crp_rp_spmm_init(int A_srow, int A_nrow, const int *A_rowptr, const int *A_colidx, const double *A_val,
                      const int *B_row_displs, int glb_n, crp_comm_t *comm, crp_rp_spmm_p *rp_spmm)
{
    (void) A_srow;   // never read by the reference either (src/rowpara_spmm.c:20-24)
    rp_init_common(A_nrow, A_rowptr, A_colidx, A_val, B_row_displs, glb_n, comm, rp_spmm, false);
}

void crp_rp_spmm_init_dv(int A_srow, int A_nrow, const int *A_rowptr, const int *A_colidx, const double *A_val,
                         const double *A_val_dev, const int *B_row_displs, int glb_n, crp_comm_t *comm, crp_rp_spmm_p *rp_spmm)
{
    (void) A_srow;
    rp_init_common(A_nrow, A_rowptr, A_colidx, A_val, B_row_displs, glb_n, comm, rp_spmm, false, A_val_dev);
}

int crp_rp_spmm_values_from_device(crp_rp_spmm_p e) { return (e && e->vals_from_device) ? 1 : 0; }

void crp_rp_spmm_init_plan_only(int A_srow, int A_nrow, const int *A_rowptr, const int *A_colidx,
                                const double *A_val, const int *B_row_displs, int glb_n, crp_comm_t *comm,
                                crp_rp_spmm_p *rp_spmm)
{
    (void) A_srow;
    rp_init_common(A_nrow, A_rowptr, A_colidx, A_val, B_row_displs, glb_n, comm, rp_spmm, true);
}

void crp_rp_spmm_free(crp_rp_spmm_p *rp_spmm)
{
    if (rp_spmm == NULL || *rp_spmm == NULL) return;
    crp_rp_spmm *e = *rp_spmm;
    if (!e->plan_only)
    {
        crp_csr_dev_destroy(&e->A_dev);
        crp_csr_dev_destroy(&e->A_int);
        crp_csr_dev_destroy(&e->A_bnd);
        if (e->xstream) crp_stream_destroy(e->xstream);
        if (e->ev_packed) crp_event_destroy(e->ev_packed);
        if (e->ev_landed) crp_event_destroy(e->ev_landed);
        crp_dev_free(e->sridxs_dev);
        crp_dev_free(e->sendbuf_dev);
        crp_dev_free(e->recvbuf_dev);
        if (e->ev_exec) crp_event_destroy(e->ev_exec);
        crp_dev_free(e->B_stage);
        crp_dev_free(e->C_stage);
        crp_dev_free(e->B_rm);
        crp_dev_free(e->C_rm);
        crp_stream_destroy(e->stream);
    }
    delete e;
    *rp_spmm = NULL;
}

void crp_rp_spmm_exec_ex(crp_rp_spmm_p e, int BC_layout, const double *B, long long ldB, double *C,
                         long long ldC, void *stream_)
{
    if (e == NULL) return;
    ASSERT_PRINTF(!e->plan_only, "rp_spmm_exec on a plan-only engine (no device state)\n");
    ASSERT_PRINTF(BC_layout == 0 || BC_layout == 1, "BC_layout must be 0 or 1\n");
    const double t_begin = get_wtime_sec();
    void *s = stream_;   // taken literally: NULL is the HIP null stream (torch's default stream)
    const int n = e->glb_n, kb = e->loc_B_nrow, m = e->A_nrow;
    const bool timing = e->timing != 0;
    double t0, t1;

    int B_on_dev = 0, C_on_dev = 0;
    if (B == e->last_B && B != NULL) B_on_dev = e->last_B_dev;
    else
    {
        HIP_OK(crp_dev_ptr_is_device(B, &B_on_dev));
        e->last_B = B;
        e->last_B_dev = B_on_dev;
    }
    if (C == e->last_C && C != NULL) C_on_dev = e->last_C_dev;
    else
    {
        HIP_OK(crp_dev_ptr_is_device(C, &C_on_dev));
        e->last_C = C;
        e->last_C_dev = C_on_dev;
    }

    // ---- bring B to a device-resident row-major view (Bd, ldBd)
    const double *Bd = B;
    long long ldBd = ldB;
    if (!B_on_dev && kb > 0 && n > 0)
    {
        // host operand: stage the whole local block (ld preserved)
        const size_t elems = (BC_layout == 0) ? (size_t) kb * (size_t) ldB : (size_t) n * (size_t) ldB;
        grow(&e->B_stage, &e->B_stage_sz, elems);
        const size_t used = (BC_layout == 0) ? ((size_t) (kb - 1) * (size_t) ldB + (size_t) n)
                                             : ((size_t) (n - 1) * (size_t) ldB + (size_t) kb);
        // one copy straight from the caller's pageable memory: the runtime stages it through its own pinned buffers at
        // PCIe rate (measured: 16 ms per exec for B in + C out of the pwtk-size operands; an engine-owned pinned mirror
        // with a memcpy in front of the DMA took 46 ms, pipelined through two pinned chunks with threaded memcpy 35 ms)
        HIP_OK(crp_dev_memcpy(e->B_stage, B, used * sizeof(double), 0, s));
        Bd = e->B_stage;
    }
    if (BC_layout == 1 && kb > 0 && n > 0)
    {
        grow(&e->B_rm, &e->B_rm_sz, (size_t) kb * (size_t) n);
        // column-major kb x n (ld ldB) == row-major n x kb; transpose to row-major kb x n
        HIP_OK(crp_transpose_f64(n, kb, Bd, ldB, e->B_rm, n, s));
        Bd = e->B_rm;
        ldBd = n;
    }
    double *Cd = C;
    long long ldCd = ldC;
    if (BC_layout == 1)
    {
        grow(&e->C_rm, &e->C_rm_sz, (size_t) m * (size_t) n);
        Cd = e->C_rm;
        ldCd = n;
    }
    else if (!C_on_dev && m > 0 && n > 0)
    {
        grow(&e->C_stage, &e->C_stage_sz, (size_t) m * (size_t) ldC);
        Cd = e->C_stage;
    }

    // ---- 1. pack the rows other ranks asked for (reference :232-262)
    if (timing) { HIP_OK(crp_stream_sync(s)); }
    t0 = get_wtime_sec();
    if (e->n_send_rows > 0 && n > 0)
        HIP_OK(crp_gather_rows_f64(0, (int) e->n_send_rows, n, e->sridxs_dev, Bd, ldBd, e->sendbuf_dev, n, s));
    if (timing)
    {
        HIP_OK(crp_stream_sync(s));
        t1 = get_wtime_sec();
        e->t_pack += t1 - t0;
        t0 = t1;
    }

    // ---- 2. exchange (reference :275-309); received rows land in final order
    const bool split = (e->A_int != nullptr);
    if (split && !timing)
    {
        // The exchange runs on its own stream beside the interior rows' product.  The product is ENQUEUED FIRST: it does not
        // depend on the exchange, and issuing a group of sends / receives can hold the host for a while (a non-blocking RCCL
        // communicator is polled until the group is on the stream) -- with the exchange first, the whole interior product
        // (0.06 - 0.2 ms per GPU at pwtk size) could have passed before its launch was even issued.
        HIP_OK(crp_event_record(e->ev_packed, s));
        HIP_OK(crp_spmm_csr_f64(e->A_int, 0, n, Bd, ldBd, e->recvbuf_dev, n, Cd, ldCd, e->variant, s));
        HIP_OK(crp_stream_wait_event(e->xstream, e->ev_packed));
        {
            const double tx0 = get_wtime_sec();
            e->comm->alltoallv_dev_f64(e->comm->ctx, e->sendbuf_dev, e->rB_scnts.data(), e->rB_sdispls.data(),
                                       e->recvbuf_dev, e->rB_rcnts.data(), e->rB_rdispls.data(), e->xstream);
            e->t_a2a_host += get_wtime_sec() - tx0;
        }
        HIP_OK(crp_event_record(e->ev_landed, e->xstream));
        HIP_OK(crp_stream_wait_event(s, e->ev_landed));
        HIP_OK(crp_spmm_csr_f64(e->A_bnd, 0, n, Bd, ldBd, e->recvbuf_dev, n, Cd, ldCd, e->variant, s));
    }
    else
    {
        if (e->nproc > 1)
        {
            const double tx0 = get_wtime_sec();
            e->comm->alltoallv_dev_f64(e->comm->ctx, e->sendbuf_dev, e->rB_scnts.data(), e->rB_sdispls.data(),
                                       e->recvbuf_dev, e->rB_rcnts.data(), e->rB_rdispls.data(), s);
            e->t_a2a_host += get_wtime_sec() - tx0;
        }
        if (timing)
        {
            HIP_OK(crp_stream_sync(s));
            t1 = get_wtime_sec();
            e->t_a2a += t1 - t0;
            t0 = t1;
        }

        // ---- 3. local SpMM (reference :388-408)
        if (split)
        {
            HIP_OK(crp_spmm_csr_f64(e->A_int, 0, n, Bd, ldBd, e->recvbuf_dev, n, Cd, ldCd, e->variant, s));
            HIP_OK(crp_spmm_csr_f64(e->A_bnd, 0, n, Bd, ldBd, e->recvbuf_dev, n, Cd, ldCd, e->variant, s));
        }
        else HIP_OK(crp_spmm_csr_f64(e->A_dev, 0, n, Bd, ldBd, e->recvbuf_dev, n, Cd, ldCd, e->variant, s));
    }
    if (BC_layout == 1 && m > 0 && n > 0)
    {
        double *Ccm = C;
        if (!C_on_dev)
        {
            grow(&e->C_stage, &e->C_stage_sz, (size_t) n * (size_t) ldC);
            Ccm = e->C_stage;
        }
        HIP_OK(crp_transpose_f64(m, n, Cd, n, Ccm, ldC, s));   // row-major n x m (ld ldC) == column-major m x n
        Cd = Ccm;
    }
    if (timing)
    {
        HIP_OK(crp_stream_sync(s));
        t1 = get_wtime_sec();
        e->t_spmm += t1 - t0;
    }

    if (!C_on_dev && m > 0 && n > 0)
    {
        const size_t used = (BC_layout == 0) ? ((size_t) (m - 1) * (size_t) ldC + (size_t) n)
                                             : ((size_t) (n - 1) * (size_t) ldC + (size_t) m);
        // ONE 2D copy straight into the caller's C (round 1 issued one copy per row when ldC != n); the caller's padding
        // between rows (columns) is never written
        const size_t w = (BC_layout == 0) ? (size_t) n : (size_t) m, h = (BC_layout == 0) ? (size_t) m : (size_t) n;
        (void) used;
        HIP_OK(crp_dev_memcpy2d(C, (size_t) ldC * sizeof(double), Cd, (size_t) ldC * sizeof(double), w * sizeof(double), h, 1, s));
        HIP_OK(crp_stream_sync(s));
    }
    else if (!B_on_dev || timing)
    {
        HIP_OK(crp_stream_sync(s));
    }
    else
    {
        // asynchronous return: remember where this exec ends (crp_rp_spmm_update_values waits for it)
        if (e->ev_exec == nullptr) HIP_OK(crp_event_create(&e->ev_exec));
        HIP_OK(crp_event_record(e->ev_exec, s));
        e->exec_pending = true;
    }
    e->t_exec += get_wtime_sec() - t_begin;
    e->n_exec++;
}

void crp_rp_spmm_exec(crp_rp_spmm_p e, int BC_layout, const double *B, int ldB, double *C, int ldC)
{
    // The reference's entry point has no stream argument.  Host operands run on the engine's own (non-blocking)
    // stream.  Device operands were produced, and will be consumed, by work the caller enqueued somewhere the engine
    // cannot know -- by HIP's rules the null stream orders against that (every blocking stream, and the null stream
    // itself), the engine's non-blocking stream would not: device operands run on the null stream.
    void *s = e ? e->stream : NULL;
    if (e != NULL && !e->plan_only)
    {
        int bd = 0, cd = 0;
        if (B == e->last_B && B != NULL) bd = e->last_B_dev; else crp_dev_ptr_is_device(B, &bd);
        if (C == e->last_C && C != NULL) cd = e->last_C_dev; else crp_dev_ptr_is_device(C, &cd);
        if (bd || cd) s = NULL;
    }
    crp_rp_spmm_exec_ex(e, BC_layout, B, (long long) ldB, C, (long long) ldC, s);
}

void crp_rp_spmm_print_stat(crp_rp_spmm_p e)
{
    if (e == NULL) return;
    const int n_exec = e->n_exec;
    if (n_exec == 0) return;
    uint64_t recv = (uint64_t) e->rB_recv_size, recv_max = 0, recv_sum = 0;
    double raw[7] = {e->t_init, e->t_pack, e->t_a2a, e->t_unpack, e->t_spmm, e->t_exec, e->t_a2a_host}, tmax[7], tavg[7];
    crp_comm_t *c = e->comm;
    c->reduce_u64(c->ctx, &recv, &recv_max, 1, CRP_OP_MAX);
    c->reduce_u64(c->ctx, &recv, &recv_sum, 1, CRP_OP_SUM);
    c->reduce_f64(c->ctx, raw, tmax, 7, CRP_OP_MAX);
    c->reduce_f64(c->ctx, raw, tavg, 7, CRP_OP_SUM);
    if (e->my_rank != 0) return;
    for (int i = 1; i <= 6; i++)
    {
        tmax[i] /= n_exec;
        tavg[i] /= ((double) n_exec * e->nproc);
    }
    recv_sum *= (uint64_t) e->glb_n;
    recv_max *= (uint64_t) e->glb_n;
    // same lines as src/rowpara_spmm.c:450-461 (harness scripts grep them)
    printf("rp_spmm_init() time = %.2f s\n", tmax[0]);
    printf("Total / rank-max SpMM comm size = %zu, %zu\n", (size_t) recv_sum, (size_t) recv_max);
    printf("-------------------- Runtime (s) --------------------\n");
    printf("                                     avg         max\n");
    printf("Pack B matrix for redistribution  %6.3f      %6.3f\n", tavg[1], tmax[1]);
    printf("Redistribute B matrix             %6.3f      %6.3f\n", tavg[2], tmax[2]);
    printf("Unpack received B matrix data     %6.3f      %6.3f\n", tavg[3], tmax[3]);
    printf("Local SpMM                        %6.3f      %6.3f\n", tavg[4], tmax[4]);
    printf("Total rp_spmm_exec()              %6.3f      %6.3f\n", tavg[5], tmax[5]);
    // additive to the reference's block: the device kernels finish in well under a millisecond
    printf("Local SpMM (us)                %9.1f   %9.1f\n", tavg[4] * 1e6, tmax[4] * 1e6);
    printf("Total rp_spmm_exec() (us)      %9.1f   %9.1f\n", tavg[5] * 1e6, tmax[5] * 1e6);
    if (e->nproc > 1) printf("Exchange call, host side (us)  %9.1f   %9.1f\n", tavg[6] * 1e6, tmax[6] * 1e6);
    printf("\n");
    fflush(stdout);
}

void crp_rp_spmm_clear_stat(crp_rp_spmm_p e)
{
    if (e == NULL) return;
    e->n_exec = 0;
    e->t_pack = e->t_a2a = e->t_unpack = e->t_spmm = e->t_exec = e->t_a2a_host = 0.0;
}

void crp_rp_spmm_get_plan(crp_rp_spmm_p e, crp_rp_plan_view_t *v)
{
    if (e == NULL || v == NULL) return;
    v->nproc = e->nproc; v->my_rank = e->my_rank; v->glb_n = e->glb_n; v->A_nrow = e->A_nrow;
    v->rB_nrow = e->rB_nrow;
    v->rB_self_src_offset = e->rB_self_src_offset;
    v->rB_self_dst_offset = e->rB_self_dst_offset;
    v->rB_self_nrow = e->rB_self_nrow;
    v->rB_p2p = e->rB_p2p; v->rB_reidx = e->rB_reidx;
    v->A_rowptr = e->A_rowptr.data(); v->A_colidx = e->A_colidx.data(); v->A_val = e->A_val.data();
    v->rB_self_src_ridxs = e->rB_self_src_ridxs.data();
    v->rB_scnts = e->rB_scnts.data(); v->rB_sdispls = e->rB_sdispls.data(); v->rB_sridxs = e->rB_sridxs.data();
    v->rB_rcnts = e->rB_rcnts.data(); v->rB_rdispls = e->rB_rdispls.data(); v->rB_rridxs = e->rB_rridxs.data();
    v->rB_recv_size = e->rB_recv_size;
    v->n_exec = e->n_exec;
    v->t_init = e->t_init; v->t_pack = e->t_pack; v->t_a2a = e->t_a2a; v->t_unpack = e->t_unpack;
    v->t_spmm = e->t_spmm; v->t_exec = e->t_exec;
}

double crp_rp_spmm_exchange_host_seconds(crp_rp_spmm_p e) { return e ? e->t_a2a_host : -1.0; }

void crp_rp_spmm_update_values(crp_rp_spmm_p e, const double *A_val)
{
    if (e == NULL) return;
    const size_t nnz = e->A_val.size();
    if (nnz == 0) return;
    ASSERT_PRINTF(A_val != NULL, "rp_spmm_update_values: NULL values\n");
    memcpy(e->A_val.data(), A_val, sizeof(double) * nnz);
    if (!e->plan_only)
    {
        if (e->exec_pending)        // kernels of an exec that returned asynchronously may still read the old values
        {
            HIP_OK(crp_stream_wait_event(e->stream, e->ev_exec));
            e->exec_pending = false;
        }
        if (e->A_int != nullptr)
        {
            for (int part = 0; part < 2; part++)
            {
                const std::vector<long long> &src = part == 0 ? e->int_src : e->bnd_src;
                if (src.empty()) continue;
                e->split_vals.resize(src.size());
                for (size_t t = 0; t < src.size(); t++) e->split_vals[t] = A_val[src[t]];
                HIP_OK(crp_csr_dev_update_values(part == 0 ? e->A_int : e->A_bnd, e->split_vals.data(), e->stream));
                HIP_OK(crp_stream_sync(e->stream));     // split_vals is reused by the next part
            }
        }
        else HIP_OK(crp_csr_dev_update_values(e->A_dev, A_val, e->stream));
        HIP_OK(crp_stream_sync(e->stream));
    }
}

void crp_rp_spmm_overlap_rows(crp_rp_spmm_p e, int *n_interior, int *n_boundary)
{
    if (n_interior) *n_interior = (e && e->A_int) ? crp_csr_dev_nrow(e->A_int) : 0;
    if (n_boundary) *n_boundary = (e && e->A_bnd) ? crp_csr_dev_nrow(e->A_bnd) : 0;
}

void crp_rp_spmm_set_timing(crp_rp_spmm_p e, int timing) { if (e) e->timing = timing ? 1 : 0; }
void crp_rp_spmm_set_variant(crp_rp_spmm_p e, int variant) { if (e) e->variant = variant; }

long long crp_rp_spmm_nnz(crp_rp_spmm_p e) { return e ? (long long) e->A_val.size() : -1; }

long long crp_rp_spmm_alg_bytes(crp_rp_spmm_p e)
{
    if (e == NULL) return -1;
    const long long nnz = (long long) e->A_val.size();
    return 12LL * nnz + 4LL * ((long long) e->A_nrow + 1) + 8LL * e->glb_n * e->n_needed_rows +
           8LL * e->glb_n * (long long) e->A_nrow;
}

// what the local kernel is for this engine's width: variant the auto choice resolves to (of the main device
// matrix, or of the interior part when the rows are split), whether its formats hold the rows in locality
// order, whether a stride lattice was found
void crp_rp_spmm_kernel_info(crp_rp_spmm_p e, int *variant, int *reordered, int *lattice)
{
    if (variant) *variant = -1;
    if (reordered) *reordered = 0;
    if (lattice) *lattice = 0;
    if (e == NULL) return;
    crp_csr_dev_p A = e->A_dev ? e->A_dev : e->A_int;
    if (A == NULL) return;
    // what the last exec launched (after alignment fallbacks); before the first exec, what auto would pick
    if (variant)
    {
        const int last = crp_csr_dev_last_variant(A);
        *variant = last > 0 ? last : (e->variant != 0 ? e->variant : crp_csr_dev_resolved_variant(A, e->glb_n));
    }
    if (reordered) *reordered = crp_csr_dev_reordered(A);
    if (lattice) *lattice = crp_csr_dev_lattice(A);
}

const int *crp_rp_spmm_dev_colidx_host(crp_rp_spmm_p e) { return e ? e->dev_colidx_host.data() : NULL; }

// ---------------------------------------------------------------------------
// single-rank communicator
static void self_a2a(void *, const int *s, int *r, int count) { memcpy(r, s, sizeof(int) * (size_t) count); }
static void self_a2av(void *, const int *s, const int *sc, const int *sd, int *r, const int *, const int *rd)
{
    memcpy(r + rd[0], s + sd[0], sizeof(int) * (size_t) sc[0]);
}
static void self_agv(void *, const void *s, size_t sb, void *r, const size_t *, const size_t *rd)
{
    memcpy((char *) r + rd[0], s, sb);
}
static void self_barrier(void *) {}
static void self_red_f64(void *, const double *in, double *out, int n, int) { memcpy(out, in, sizeof(double) * (size_t) n); }
static void self_red_u64(void *, const uint64_t *in, uint64_t *out, int n, int) { memcpy(out, in, sizeof(uint64_t) * (size_t) n); }
static void self_a2av_dev(void *, const double *, const long long *, const long long *, double *, const long long *,
                          const long long *, void *) {}
static void self_a2av_bytes(void *, const void *s, const size_t *sc, const size_t *sd, void *r, const size_t *,
                            const size_t *rd)
{
    memcpy((char *) r + rd[0], (const char *) s + sd[0], sc[0]);
}
static void self_free(crp_comm_t *c) { free(c); }
static crp_comm_t *self_split(void *, int, int) { return crp_comm_self(); }

crp_comm_t *crp_comm_self(void)
{
    crp_comm_t *c = (crp_comm_t *) calloc(1, sizeof(crp_comm_t));
    c->nproc = 1;
    c->rank = 0;
    c->alltoall_i32 = self_a2a;
    c->alltoallv_i32 = self_a2av;
    c->allgatherv_bytes = self_agv;
    c->barrier = self_barrier;
    c->reduce_f64 = self_red_f64;
    c->reduce_u64 = self_red_u64;
    c->alltoallv_dev_f64 = self_a2av_dev;
    c->alltoallv_bytes = self_a2av_bytes;
    c->split = self_split;
    c->free = self_free;
    return c;
}

}  // extern "C"
