// crpspmm_engine.cpp -- compatibility engine with the older all-in-one calling convention
// (include/crp_engine.h: crp_crpspmm_*; reference: deprecated/src/crpspmm.{h,c}).
// Built from the pieces of the live path: crp_mat_redist (A pattern / values, B in, C out) and
// the 1D row-parallel device engine inside every grid column.
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "crp_engine.h"
#include "spmat_part.h"
#include "utils.h"
#include "knobs.h"

struct crp_crpspmm
{
    crp_crpspmm_view_t v;
    crp_comm_t *comm = nullptr, *comm_col = nullptr;
    crp_mat_redist_p rd_Ai = nullptr, rd_Av = nullptr, rd_B = nullptr, rd_C = nullptr;
    crp_rp_spmm_p rp = nullptr;
    std::vector<int>    loc_A_rowptr, loc_A_colidx, pub_rowptr;
    std::vector<double> loc_A_val, red_B, loc_C;
    int  src_A_nnz = 0;
    bool a_static = false, a_loaded = false, plan_only = false;
};

extern "C" {

void crp_crpspmm_plan_grid(int P, int m, int n, int k, const int *A_rowptr_glb, const int *cidx_se, int *np_row,
                           int *np_col, int *m_split_idx)
{
    const double nnz_cf = 1.5;           // one nonzero (int32 + fp64) in fp64 elements
    int m_split = 1, n_split = 1, *fac = NULL;
    const int nfac = prime_factorization(P, &fac);
    std::vector<int> cand((size_t) P + 1);
    m_split_idx[0] = 0;
    m_split_idx[1] = m;
    size_t copy_B = (size_t) k * (size_t) n;      // one copy of B to start with
    const int nnz = A_rowptr_glb[m];
    for (int i = 0; i < nfac; i++)
    {
        const int p = fac[nfac - 1 - i];
        // splitting N multiplies the copies of A by p and leaves the B volume unchanged
        const size_t A1 = (size_t) ((double) nnz * (double) n_split * nnz_cf);
        size_t cost_n = A1 * (size_t) p + copy_B;
        if (n_split * p > n) cost_n = SIZE_MAX;
        // splitting M keeps A and re-derives the B rows every panel may need from the rows'
        // (first, last) column ranges -- an upper bound, not the exact set
        const int ms = m_split * p;
        size_t copy_B2 = 0;
        int srow = 0;
        cand[0] = 0;
        for (int j = 0; j < ms; j++)
        {
            const int target = (j == ms - 1) ? nnz : nnz / ms * (j + 1);
            int erow = srow + 1;
            if (erow > m) erow = m;
            int lo = INT_MAX, hi = -1;
            auto widen = [&](int row) {
                if (row >= m) return;
                const int a = cidx_se[2 * row], b = cidx_se[2 * row + 1];
                if (a > b) return;                                  // empty row
                if (a < lo) lo = a;
                if (b > hi) hi = b;
            };
            widen(srow);
            while (erow < m && A_rowptr_glb[erow] < target)
            {
                widen(erow);
                erow++;
            }
            if (hi >= lo) copy_B2 += (size_t) (hi - lo + 1) * (size_t) n;
            cand[j + 1] = erow;
            srow = erow;
        }
        cand[ms] = m;
        const size_t cost_m = A1 + copy_B2;
        if (cost_m < cost_n)
        {
            m_split = ms;
            copy_B = copy_B2;
            memcpy(m_split_idx, cand.data(), sizeof(int) * (size_t) (ms + 1));
        }
        else n_split *= p;
    }
    free(fac);
    *np_row = m_split;
    *np_col = n_split;
}

static void crpspmm_init_impl(int m, int n, int k, int src_A_srow, int src_A_nrow, const int *src_A_rowptr,
                              const int *src_A_colidx, int src_B_srow, int src_B_nrow, int src_B_scol, int src_B_ncol,
                              int dst_C_srow, int dst_C_nrow, int dst_C_scol, int dst_C_ncol, crp_comm_t *comm,
                              crp_crpspmm_p *engine_, bool plan_only)
{
    ASSERT_PRINTF(engine_ != NULL && comm != NULL, "crpspmm_engine_init: NULL argument\n");
    *engine_ = NULL;
    const double t0 = get_wtime_sec();
    crp_crpspmm *e = new crp_crpspmm;
    memset(&e->v, 0, sizeof(e->v));
    e->comm = comm;
    const int P = comm->nproc, me = comm->rank;
    e->v.np_glb = P; e->v.rank_glb = me; e->v.glb_m = m; e->v.glb_n = n; e->v.glb_k = k;
    // the reference's knob (deprecated/src/crpspmm.c:294): read and reported with the reference's
    // message; the exchange here always moves exactly the rows a panel needs, whatever it says
    GET_ENV_INT_VAR(e->v.a2a_B_finegrain, "A2A_B_FINEGRAIN", "a2a_B_finegrain", 0, 0, 1, me == 0);
    e->a_static = crp::knobs().engine_a_static == 1;

    // 1. global row pointer and per-row column ranges (deprecated/src/crpspmm.c:91-126)
    std::vector<int> nrow_of(P), A_rowptr((size_t) m + 1, 0), cse((size_t) 2 * (m > 0 ? m : 1), 0);
    std::vector<size_t> cnt(P), dsp(P);
    for (int q = 0; q < P; q++) { cnt[q] = sizeof(int); dsp[q] = sizeof(int) * (size_t) q; }
    comm->allgatherv_bytes(comm->ctx, &src_A_nrow, sizeof(int), nrow_of.data(), cnt.data(), dsp.data());
    size_t off = 0;
    for (int q = 0; q < P; q++) { cnt[q] = sizeof(int) * (size_t) nrow_of[q]; dsp[q] = off; off += cnt[q]; }
    ASSERT_PRINTF(off == sizeof(int) * (size_t) m, "crpspmm_engine_init: the ranks' A row blocks do not add up to m\n");
    comm->allgatherv_bytes(comm->ctx, src_A_rowptr, sizeof(int) * (size_t) src_A_nrow, A_rowptr.data(), cnt.data(), dsp.data());
    {
        std::vector<int> ends(P);
        for (int q = 0; q < P; q++) { cnt[q] = sizeof(int); dsp[q] = sizeof(int) * (size_t) q; }
        const int my_end = src_A_rowptr[src_A_nrow];
        comm->allgatherv_bytes(comm->ctx, &my_end, sizeof(int), ends.data(), cnt.data(), dsp.data());
        A_rowptr[m] = ends[P - 1];
    }
    std::vector<int> my_se((size_t) 2 * (src_A_nrow > 0 ? src_A_nrow : 1));
    for (int i = 0; i < src_A_nrow; i++)
    {
        int lo = INT_MAX, hi = -1;
        for (int p = src_A_rowptr[i] - src_A_rowptr[0]; p < src_A_rowptr[i + 1] - src_A_rowptr[0]; p++)
        {
            if (src_A_colidx[p] < lo) lo = src_A_colidx[p];
            if (src_A_colidx[p] > hi) hi = src_A_colidx[p];
        }
        my_se[2 * i] = lo;
        my_se[2 * i + 1] = hi;
    }
    off = 0;
    for (int q = 0; q < P; q++) { cnt[q] = 2 * sizeof(int) * (size_t) nrow_of[q]; dsp[q] = off; off += cnt[q]; }
    comm->allgatherv_bytes(comm->ctx, my_se.data(), 2 * sizeof(int) * (size_t) src_A_nrow, cse.data(), cnt.data(), dsp.data());

    // 2. grid (deprecated/src/crpspmm.c:128-206)
    std::vector<int> m_split_idx((size_t) P + 1, 0);
    int np_row = 1, np_col = 1;
    crp_crpspmm_plan_grid(P, m, n, k, A_rowptr.data(), cse.data(), &np_row, &np_col, m_split_idx.data());
    const int rank_row = me / np_col, rank_col = me % np_col;
    e->v.np_row = np_row; e->v.np_col = np_col; e->v.rank_row = rank_row; e->v.rank_col = rank_col;
    int loc_B_scol, loc_B_ncol, tmp;
    calc_block_spos_size(n, np_col, rank_col, &loc_B_scol, &loc_B_ncol);
    const int a_s = m_split_idx[rank_row], a_e = m_split_idx[rank_row + 1];
    e->v.loc_A_srow = a_s; e->v.loc_A_erow = a_e; e->v.loc_A_nrow = a_e - a_s;
    e->v.loc_A_nnz_s = A_rowptr[a_s]; e->v.loc_A_nnz = A_rowptr[a_e] - A_rowptr[a_s];
    e->v.loc_B_scol = loc_B_scol; e->v.loc_B_ncol = loc_B_ncol; e->v.loc_B_ecol = loc_B_scol + loc_B_ncol;
    std::vector<int> B_displs((size_t) np_row + 1);
    for (int i = 0; i <= np_row; i++) calc_block_spos_size(k, np_row, i, &B_displs[i], &tmp);
    e->v.rd_B_srow = B_displs[rank_row]; e->v.rd_B_erow = B_displs[rank_row + 1];

    // 3. A's pattern to the row panel of this grid row: the nonzero arrays are 1 x nnz "matrices"
    //    (deprecated/src/crpspmm.c:228-255; here the panel is requested whole, one step)
    const int src_nnz_s = A_rowptr[src_A_srow];
    e->src_A_nnz = A_rowptr[src_A_srow + src_A_nrow] - src_nnz_s;
    crp_mat_redist_init(0, src_nnz_s, 1, e->src_A_nnz, 0, e->v.loc_A_nnz_s, 1, e->v.loc_A_nnz, comm, sizeof(int), 0, &e->rd_Ai, NULL);
    crp_mat_redist_init(0, src_nnz_s, 1, e->src_A_nnz, 0, e->v.loc_A_nnz_s, 1, e->v.loc_A_nnz, comm, sizeof(double), 0, &e->rd_Av, NULL);
    const int pn = e->v.loc_A_nnz;
    e->loc_A_colidx.assign((size_t) (pn > 0 ? pn : 1), 0);
    e->loc_A_val.assign((size_t) (pn > 0 ? pn : 1), 0.0);
    {
        std::vector<int> dummy(1, 0);
        crp_mat_redist_exec(e->rd_Ai, e->src_A_nnz > 0 ? (const void *) src_A_colidx : (const void *) dummy.data(),
                            e->src_A_nnz > 0 ? e->src_A_nnz : 1, e->loc_A_colidx.data(), pn > 0 ? pn : 1);
    }
    e->loc_A_rowptr.resize((size_t) e->v.loc_A_nrow + 1);
    for (int i = 0; i <= e->v.loc_A_nrow; i++) e->loc_A_rowptr[i] = A_rowptr[a_s + i];   // global offsets, like the live API

    e->pub_rowptr.resize(e->loc_A_rowptr.size());
    for (size_t i = 0; i < e->pub_rowptr.size(); i++) e->pub_rowptr[i] = e->loc_A_rowptr[i] - e->v.loc_A_nnz_s;
    {
        int lo = INT_MAX, hi = -1;
        std::vector<char> seen((size_t) (k > 0 ? k : 1), 0);
        int cntr = 0;
        for (int p = 0; p < pn; p++)
        {
            const int c = e->loc_A_colidx[p];
            if (c < lo) lo = c;
            if (c > hi) hi = c;
            if (!seen[c]) { seen[c] = 1; cntr++; }
        }
        e->v.loc_B_srow = (hi >= lo) ? lo : 0;
        e->v.loc_B_erow = (hi >= lo) ? hi + 1 : 0;
        e->v.loc_B_nrow = cntr;
    }

    // 4. B: caller's blocks -> (even k split) x (even n split); C back to the caller's blocks
    crp_mat_redist_init(src_B_srow, src_B_scol, src_B_nrow, src_B_ncol, e->v.rd_B_srow, loc_B_scol,
                        e->v.rd_B_erow - e->v.rd_B_srow, loc_B_ncol, comm, sizeof(double), 0, &e->rd_B, NULL);
    crp_mat_redist_init(a_s, loc_B_scol, e->v.loc_A_nrow, loc_B_ncol, dst_C_srow, dst_C_scol, dst_C_nrow, dst_C_ncol,
                        comm, sizeof(double), 0, &e->rd_C, NULL);
    const size_t ldl = (size_t) (loc_B_ncol > 0 ? loc_B_ncol : 1);
    e->red_B.assign((size_t) (e->v.rd_B_erow - e->v.rd_B_srow > 0 ? e->v.rd_B_erow - e->v.rd_B_srow : 1) * ldl, 0.0);
    e->loc_C.assign((size_t) (e->v.loc_A_nrow > 0 ? e->v.loc_A_nrow : 1) * ldl, 0.0);

    // 5. the device engine inside the grid column (ranks with equal rank_col, ordered by rank_row)
    e->comm_col = comm->split(comm->ctx, rank_col, me);
    e->plan_only = plan_only;
    (plan_only ? crp_rp_spmm_init_plan_only : crp_rp_spmm_init)(a_s, e->v.loc_A_nrow, e->loc_A_rowptr.data(),
        e->loc_A_colidx.data(), e->loc_A_val.data(), B_displs.data(), loc_B_ncol, e->comm_col, &e->rp);

    // 6. communication volumes as the deprecated engine reports them (deprecated/src/crpspmm.c:446-456)
    int share_s, share_n;
    calc_block_spos_size(pn, np_col, rank_col, &share_s, &share_n);
    crp_rp_plan_view_t pv;
    crp_rp_spmm_get_plan(e->rp, &pv);
    e->v.nelem_A_rd = (size_t) share_n;
    e->v.nelem_A_agv = (np_col == 1) ? 0 : (size_t) pn;
    e->v.nelem_B_rd = (size_t) (e->v.rd_B_erow - e->v.rd_B_srow) * (size_t) loc_B_ncol;
    e->v.nelem_B_a2av = (np_row == 1) ? 0 : (size_t) pv.rB_recv_size * (size_t) loc_B_ncol;
    e->v.nelem_B_a2av_min = e->v.nelem_B_a2av;
    e->v.loc_A_rowptr = e->pub_rowptr.data(); e->v.loc_A_colidx = e->loc_A_colidx.data();
    e->v.loc_A_val = e->loc_A_val.data(); e->v.red_B = e->red_B.data(); e->v.loc_C = e->loc_C.data();
    e->v.t_init = get_wtime_sec() - t0;
    *engine_ = e;
}

void crp_crpspmm_init(int m, int n, int k, int src_A_srow, int src_A_nrow, const int *src_A_rowptr,
                      const int *src_A_colidx, int src_B_srow, int src_B_nrow, int src_B_scol, int src_B_ncol,
                      int dst_C_srow, int dst_C_nrow, int dst_C_scol, int dst_C_ncol, crp_comm_t *comm,
                      crp_crpspmm_p *engine_)
{
    crpspmm_init_impl(m, n, k, src_A_srow, src_A_nrow, src_A_rowptr, src_A_colidx, src_B_srow, src_B_nrow, src_B_scol,
                      src_B_ncol, dst_C_srow, dst_C_nrow, dst_C_scol, dst_C_ncol, comm, engine_, false);
}

void crp_crpspmm_init_plan_only(int m, int n, int k, int src_A_srow, int src_A_nrow, const int *src_A_rowptr,
                                const int *src_A_colidx, int src_B_srow, int src_B_nrow, int src_B_scol, int src_B_ncol,
                                int dst_C_srow, int dst_C_nrow, int dst_C_scol, int dst_C_ncol, crp_comm_t *comm,
                                crp_crpspmm_p *engine_)
{
    crpspmm_init_impl(m, n, k, src_A_srow, src_A_nrow, src_A_rowptr, src_A_colidx, src_B_srow, src_B_nrow, src_B_scol,
                      src_B_ncol, dst_C_srow, dst_C_nrow, dst_C_scol, dst_C_ncol, comm, engine_, true);
}

void crp_crpspmm_exec(crp_crpspmm_p e, const int *src_A_rowptr, const int *src_A_colidx, const double *src_A_val,
                      const double *src_B, int ldB, double *dst_C, int ldC)
{
    if (e == NULL) return;
    (void) src_A_rowptr;
    (void) src_A_colidx;
    const double t_begin = get_wtime_sec();
    double t0 = t_begin, t1;
    const int pn = e->v.loc_A_nnz, ldl = e->v.loc_B_ncol > 0 ? e->v.loc_B_ncol : 1;
    // A's values (they may change between calls; CRPSPMM_ENGINE_A_STATIC=1 declares that they do not)
    if (!(e->a_static && e->a_loaded))
    {
        double dummy = 0.0;
        crp_mat_redist_exec(e->rd_Av, e->src_A_nnz > 0 ? (const void *) src_A_val : (const void *) &dummy,
                            e->src_A_nnz > 0 ? e->src_A_nnz : 1, e->loc_A_val.data(), pn > 0 ? pn : 1);
        t1 = get_wtime_sec();
        e->v.t_rd_A += t1 - t0;
        t0 = t1;
        if (!e->plan_only) crp_rp_spmm_update_values(e->rp, e->loc_A_val.data());
        e->a_loaded = true;
        t1 = get_wtime_sec();
        e->v.t_agv_A += t1 - t0;      // the replication step of the old engine: here the upload of the panel
        t0 = t1;
    }
    crp_mat_redist_exec(e->rd_B, src_B, ldB, e->red_B.data(), ldl);
    t1 = get_wtime_sec();
    e->v.t_rd_B += t1 - t0;
    if (e->plan_only)
    {
        // no device state: the redistributed inputs stay inspectable through the view, C is not produced
        e->v.t_exec += t1 - t_begin;
        e->v.n_exec++;
        return;
    }
    const double t_nr = t1;
    crp_rp_plan_view_t before, after;
    crp_rp_spmm_get_plan(e->rp, &before);
    crp_rp_spmm_exec(e->rp, 0, e->red_B.data(), ldl, e->loc_C.data(), ldl);
    crp_rp_spmm_get_plan(e->rp, &after);
    t0 = get_wtime_sec();
    e->v.t_a2a_B += (after.t_pack - before.t_pack) + (after.t_a2a - before.t_a2a) + (after.t_unpack - before.t_unpack);
    e->v.t_spmm += after.t_spmm - before.t_spmm;
    e->v.t_exec_nr += t0 - t_nr;
    crp_mat_redist_exec(e->rd_C, e->loc_C.data(), ldl, dst_C, ldC);
    t1 = get_wtime_sec();
    e->v.t_rd_C += t1 - t0;
    e->v.t_exec += t1 - t_begin;
    e->v.n_exec++;
}

void crp_crpspmm_free(crp_crpspmm_p *engine_)
{
    if (engine_ == NULL || *engine_ == NULL) return;
    crp_crpspmm *e = *engine_;
    crp_rp_spmm_free(&e->rp);
    crp_mat_redist_free(&e->rd_Ai);
    crp_mat_redist_free(&e->rd_Av);
    crp_mat_redist_free(&e->rd_B);
    crp_mat_redist_free(&e->rd_C);
    if (e->comm_col) e->comm_col->free(e->comm_col);
    delete e;
    *engine_ = NULL;
}

void crp_crpspmm_print_stat(crp_crpspmm_p e)
{
    if (e == NULL) return;
    if (e->v.rank_glb == 0) printf("crpspmm_engine init time: %.3f s\n", e->v.t_init);
    const int n_exec = e->v.n_exec;
    if (n_exec == 0) return;
    double raw[8] = {e->v.t_rd_A, e->v.t_rd_B, e->v.t_agv_A, e->v.t_a2a_B, e->v.t_spmm, e->v.t_exec_nr, e->v.t_rd_C, e->v.t_exec};
    double neg[8], tmin[8], tmax[8], tavg[8];
    uint64_t cs[5] = {e->v.nelem_A_rd, e->v.nelem_A_agv, e->v.nelem_B_rd, e->v.nelem_B_a2av, e->v.nelem_B_a2av_min};
    uint64_t cneg[5], cmin[5], cmax[5], csum[5];
    crp_comm_t *c = e->comm;
    for (int i = 0; i < 8; i++) neg[i] = -raw[i];
    for (int i = 0; i < 5; i++) cneg[i] = ~cs[i];            // min via max of the complement
    c->reduce_f64(c->ctx, neg, tmin, 8, CRP_OP_MAX);
    c->reduce_f64(c->ctx, raw, tmax, 8, CRP_OP_MAX);
    c->reduce_f64(c->ctx, raw, tavg, 8, CRP_OP_SUM);
    c->reduce_u64(c->ctx, cneg, cmin, 5, CRP_OP_MAX);
    c->reduce_u64(c->ctx, cs, cmax, 5, CRP_OP_MAX);
    c->reduce_u64(c->ctx, cs, csum, 5, CRP_OP_SUM);
    if (e->v.rank_glb != 0) return;
    for (int i = 0; i < 8; i++)
    {
        tmin[i] = -tmin[i] / n_exec;
        tmax[i] /= n_exec;
        tavg[i] /= ((double) e->v.np_glb * n_exec);
    }
    for (int i = 0; i < 5; i++) cmin[i] = ~cmin[i];
    const char *names[8] = {"Redist A to internal 1D layout ", "Redist B to internal 2D layout ", "Replicate A with allgatherv    ",
                            "Replicate B with alltoallv     ", "Local SpMM                     ", "SpMM w/o Redist                ",
                            "Redist C to user's 2D layout   "};
    printf("-------------------------- Runtime (s) -------------------------\n");
    printf("                                   min         avg         max\n");
    for (int i = 0; i < 7; i++) printf("%s %6.3f      %6.3f      %6.3f\n", names[i], tmin[i], tavg[i], tmax[i]);
    printf("SpMM total (avg of %3d runs)    %6.3f      %6.3f      %6.3f\n", n_exec, tmin[7], tavg[7], tmax[7]);
    printf("----------------------------------------------------------------\n");
    printf("------------------ Communicated Matrix Elements -----------------\n");
    printf("                               min           max            sum\n");
    const char *cn[5] = {"Redist A              ", "Allgatherv A          ", "Redist B              ", "Alltoallv B           ",
                         "Alltoallv B necessary "};
    for (int i = 0; i < 5; i++)
        printf("%s  %10zu    %10zu    %11zu\n", cn[i], (size_t) cmin[i], (size_t) cmax[i], (size_t) csum[i]);
    printf("----------------------------------------------------------------\n");
    printf("\n");
    fflush(stdout);
}

void crp_crpspmm_clear_stat(crp_crpspmm_p e)
{
    if (e == NULL) return;
    e->v.n_exec = 0;
    e->v.t_exec = e->v.t_rd_A = e->v.t_agv_A = e->v.t_rd_B = e->v.t_a2a_B = e->v.t_spmm = e->v.t_rd_C = e->v.t_exec_nr = 0.0;
}

void crp_crpspmm_get_view(crp_crpspmm_p e, crp_crpspmm_view_t *view)
{
    if (e == NULL || view == NULL) return;
    *view = e->v;
}

}  // extern "C"
