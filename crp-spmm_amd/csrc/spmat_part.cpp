// spmat_part.cpp -- host partition planner behind include/spmat_part.h.
// Integer outputs are bit-exact with /root/reference/src/spmat_part.c:12-210;
// the distinct-column counting uses a stamp array instead of the reference's
// per-thread byte flags (one pass over the nonzeros, no memset per block).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "spmat_part.h"
#include "utils.h"
#include "par.h"

// Row index the reference's bisection reaches for `target` (src/spmat_part.c:22-32):
// a lower bound on row_ptr over [0, nrow) that stops early on an exact hit, so
// with runs of empty rows the answer is the hit the bisection meets first.
static int bisect_row_ptr(const int *row_ptr, int nrow, int target)
{
    int lo = 0, hi = nrow;
    while (lo < hi)
    {
        const int mid = (lo + hi) / 2;
        const int v = row_ptr[mid];
        if (v == target) return mid;
        if (v < target) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

extern "C" {

void csr_mat_row_partition(const int nrow, const int *row_ptr, const int nblk, int *rblk_ptr)
{
    const int nnz = row_ptr[nrow];
    const int share = nnz / nblk;
    rblk_ptr[0] = 0;
    for (int b = 0; b < nblk; b++)
    {
        const int target = (b == nblk - 1) ? nnz : share * (b + 1);
        rblk_ptr[b + 1] = bisect_row_ptr(row_ptr, nrow, target);
    }
}

int prime_factorization(int n, int **factors)
{
    // at most log2(n) factors
    int cap = 2;
    for (int t = n; t > 1; t >>= 1) cap++;
    int *fac = (int *) malloc(sizeof(int) * cap);
    int nfac = 0;
    for (int c = 2; n > 1;)
    {
        if (n % c == 0)
        {
            fac[nfac++] = c;
            n /= c;
        }
        else c++;
    }
    *factors = fac;
    return nfac;
}

void csr_mat_row_part_comm_size(const int nrow, const int ncol, const int *row_ptr, const int *col_idx,
                                const int nblk, const int *rblk_ptr, const int *x_displs,
                                int *comm_sizes, int *total_size)
{
    (void) nrow;
    // one stamp array per worker thread: stamp[c] == b + 1  <=>  column c already seen in block b
    std::vector<std::vector<int>> stamps((size_t) crp::host_threads());
    crp::parallel_chunks(nblk, 1, [&](long long b0, long long b1, int tid) {
        std::vector<int> &stamp = stamps[(size_t) tid];
        if (stamp.empty()) stamp.assign((size_t) (ncol > 0 ? ncol : 1), 0);
        for (int b = (int) b0; b < (int) b1; b++)
        {
            const int xlo = x_displs[b], xhi = x_displs[b + 1], tag = b + 1;
            int cnt = 0;
            for (int p = row_ptr[rblk_ptr[b]]; p < row_ptr[rblk_ptr[b + 1]]; p++)
            {
                const int c = col_idx[p];
                if (stamp[c] != tag)
                {
                    stamp[c] = tag;
                    if (c < xlo || c >= xhi) cnt++;
                }
            }
            comm_sizes[b] = cnt;
        }
    });
    int tot = 0;
    for (int b = 0; b < nblk; b++) tot += comm_sizes[b];
    *total_size = tot;
}

}  // extern "C"

// output arrays of a chosen grid (src/spmat_part.c:162-208): panel rows, B rows, C columns and the
// nnz-balanced re-split of every replicated panel into gn source slices
static void part2d_emit(const int nproc, const int m, const int n, const int k, const int *rowptr, const int gm,
                        const int gn, const int *best_rows_, int **A0_rowptr, int **B_rowptr, int **AC_rowptr,
                        int **BC_colptr)
{
    int tmp;
    std::vector<int> best_rows(best_rows_, best_rows_ + gm + 1);
    auto b_rows_for = [&](const int *rows, int nblk, int *out) {
        if (m == k) memcpy(out, rows, sizeof(int) * (nblk + 1));
        else for (int i = 0; i <= nblk; i++) calc_block_spos_size(k, nblk, i, out + i, &tmp);
    };
    int *ac = (int *) malloc(sizeof(int) * (gm + 1));
    int *br = (int *) malloc(sizeof(int) * (gm + 1));
    int *bc = (int *) malloc(sizeof(int) * (gn + 1));
    int *a0 = (int *) malloc(sizeof(int) * (nproc + 1));
    memcpy(ac, best_rows.data(), sizeof(int) * (gm + 1));
    b_rows_for(ac, gm, br);
    for (int j = 0; j <= gn; j++) calc_block_spos_size(n, gn, j, bc + j, &tmp);

    // nnz-balanced re-split of every replicated panel into gn source slices
    std::vector<int> local_ptr((size_t) m + 1);
    for (int i = 0; i < gm; i++)
    {
        const int r0 = ac[i], r1 = ac[i + 1];
        for (int r = r0; r <= r1; r++) local_ptr[r - r0] = rowptr[r] - rowptr[r0];
        int *slice = a0 + i * gn;
        csr_mat_row_partition(r1 - r0, local_ptr.data(), gn, slice);
        for (int j = 0; j <= gn; j++) slice[j] += r0;
    }
    *A0_rowptr = a0;
    *B_rowptr  = br;
    *AC_rowptr = ac;
    *BC_colptr = bc;
}

extern "C" {

void calc_spmm_part2d_from_1d(const int nproc, const int m, const int n, const int k, const int *rb_displs0,
                              const int *rowptr, const int *colidx, const int rA, int *pm, int *pn,
                              size_t *comm_cost, int **A0_rowptr, int **B_rowptr, int **AC_rowptr,
                              int **BC_colptr, int dbg_print)
{
    const double nnz_cf = 1.5;   // cost of one replicated nonzero in fp64 elements (int32 + fp64)
    std::vector<int> best_rows(rb_displs0, rb_displs0 + nproc + 1), cand_rows(nproc + 1), xd(nproc + 1),
        sizes(nproc);
    int tmp;

    // B's rows follow A's row blocks when A is square, else an even split
    auto b_rows_for = [&](const int *rows, int nblk, int *out) {
        if (m == k) memcpy(out, rows, sizeof(int) * (nblk + 1));
        else for (int i = 0; i <= nblk; i++) calc_block_spos_size(k, nblk, i, out + i, &tmp);
    };

    // pure 1D start point: nproc x 1
    b_rows_for(rb_displs0, nproc, xd.data());
    int vol = 0;
    csr_mat_row_part_comm_size(m, k, rowptr, colidx, nproc, rb_displs0, xd.data(), sizes.data(), &vol);
    size_t best = (size_t) vol * (size_t) n;
    if (dbg_print) printf("Basic 1D row partitioning comm cost: %zu\n", best);

    int gm = nproc, gn = 1, rejected = -1;
    const int nnz = rowptr[m];
    int *fac = NULL;
    const int nfac = prime_factorization(nproc, &fac);
    for (int step = 0; step < nfac; step++)
    {
        const int p = fac[nfac - 1 - step];   // largest factor first
        if (p == rejected) continue;
        const int tn = gn * p, tm = nproc / tn;
        for (int i = 0; i <= tm; i++) cand_rows[i] = rb_displs0[i * tn];
        b_rows_for(cand_rows.data(), tm, xd.data());
        const double t0 = get_wtime_sec();
        csr_mat_row_part_comm_size(m, k, rowptr, colidx, tm, cand_rows.data(), xd.data(), sizes.data(), &vol);
        const double t1 = get_wtime_sec();
        const size_t costA = (size_t) ((double) nnz * (double) (tn - 1) * nnz_cf);
        const size_t costB = (size_t) rA * (size_t) vol * (size_t) n;
        const size_t cost = costA + costB;
        if (dbg_print)
        {
            printf("Step %d, factor %d, time = %.2f\n", step, p, t1 - t0);
            printf("Evaluated: pm = %d, pn = %d, cost = %zu\n", tm, tn, cost);
            if (cost < best) printf("Found better partitioning\n");
        }
        if (cost < best)
        {
            best = cost;
            gm = tm;
            gn = tn;
            memcpy(best_rows.data(), cand_rows.data(), sizeof(int) * (tm + 1));
            rejected = -1;
        }
        else rejected = p;
    }
    free(fac);
    *comm_cost = best;
    *pm = gm;
    *pn = gn;
    if (dbg_print) printf("Final 2D partitioning: pm = %d, pn = %d, cost = %zu\n", gm, gn, best);

    part2d_emit(nproc, m, n, k, rowptr, gm, gn, best_rows.data(), A0_rowptr, B_rowptr, AC_rowptr, BC_colptr);
}

// Grid choice for an A that is multiplied rA times (iterative solvers, the benchmark loop): the
// reference rule above leaves rA out of the pure-1D starting cost and walks the prime factors
// greedily (src/spmat_part.c:113,120-159), so rA > 1 pushes it TOWARDS 1D.  Here every pm x pn with
// pn | nproc is priced the same way -- floor(1.5 nnz (pn-1)) for the one-time replication of A plus
// rA * n * (B rows exchanged per multiply) -- and the cheapest wins.  Output arrays as above.
void crp_spmm_part2d_amortized(const int nproc, const int m, const int n, const int k, const int *rb_displs0,
                               const int *rowptr, const int *colidx, const int rA, int *pm, int *pn,
                               size_t *comm_cost, int **A0_rowptr, int **B_rowptr, int **AC_rowptr, int **BC_colptr)
{
    const double nnz_cf = 1.5;
    const int nnz = rowptr[m];
    std::vector<int> rows((size_t) nproc + 1), xd((size_t) nproc + 1), sizes((size_t) nproc), best_rows;
    size_t best = SIZE_MAX;
    int gm = nproc, gn = 1, tmp;
    for (int tn = 1; tn <= nproc; tn++)
    {
        if (nproc % tn != 0 || (tn > 1 && tn > n)) continue;
        const int tm = nproc / tn;
        for (int i = 0; i <= tm; i++) rows[i] = rb_displs0[i * tn];
        if (m == k) memcpy(xd.data(), rows.data(), sizeof(int) * (tm + 1));
        else for (int i = 0; i <= tm; i++) calc_block_spos_size(k, tm, i, xd.data() + i, &tmp);
        int vol = 0;
        csr_mat_row_part_comm_size(m, k, rowptr, colidx, tm, rows.data(), xd.data(), sizes.data(), &vol);
        const size_t cost = (size_t) ((double) nnz * (double) (tn - 1) * nnz_cf) + (size_t) rA * (size_t) vol * (size_t) n;
        if (cost < best)
        {
            best = cost;
            gm = tm;
            gn = tn;
            best_rows.assign(rows.begin(), rows.begin() + tm + 1);
        }
    }
    *comm_cost = best;
    *pm = gm;
    *pn = gn;
    part2d_emit(nproc, m, n, k, rowptr, gm, gn, best_rows.data(), A0_rowptr, B_rowptr, AC_rowptr, BC_colptr);
}

// Fraction of the HBM roofline (algorithmic bytes) the local kernels reach at a given operand width, measured on
// MI355X (profiles/r04_sweep_n32_256_1024.jsonl, r04_bench_pwtk_n*.json and DESIGN.md section 4.00): the width a grid leaves every GPU with is
// part of its price.
static double kernel_fraction(const int n_local)
{
    if (n_local >= 256) return 0.44;
    if (n_local >= 96) return 0.43;          // (round 4: the one-piece team instances run three workgroups per CU)
    if (n_local > 32) return 0.40;
    if (n_local >= 24) return 0.50;
    return 0.13;
}

// Grid choice by a TIME model of one node of point-to-point links (the reference prices bytes,
// src/spmat_part.c:113-159; xGMI is not a switch, every pair of GPUs has its own link).  For every pm x pn:
//   one-time   t_rep  = largest piece of a replicated panel (12 bytes per nonzero) / link rate: the pn - 1 pieces a rank
//                       receives arrive over different links at once (the engines' device all-gather is a fan-out);
//   per exec   t_exch = the largest number of B rows one rank needs from ONE peer x n_local x 8 / link rate: pairs
//                       exchange in parallel, the slowest pair ends the step (exact per-pair row counts);
//              t_comp = the largest per-rank algorithmic bytes / (HBM rate x the kernels' measured fraction at n_local);
//              the exchange runs beside the interior rows' product: t_exec = max(t_comp, t_exch);
//   price      t_rep / rA + t_exec.
// Grids whose panel replicas or operands do not fit `hbm_bytes` per GPU are skipped.  times[0..2] (optional) = t_rep,
// t_exch, t_comp of the winner in seconds.  mm (optional): {link GB/s per direction, HBM GB/s, HBM bytes per GPU}.
void crp_spmm_part2d_timed(const int nproc, const int m, const int n, const int k, const int *rb_displs0, const int *rowptr,
                           const int *colidx, const int rA, const double *mm, int *pm, int *pn, double *times,
                           int **A0_rowptr, int **B_rowptr, int **AC_rowptr, int **BC_colptr)
{
    const double link = (mm && mm[0] > 0 ? mm[0] : 64.0) * 1e9;            // xGMI: 153 GB/s per link both ways, ~64 one way in practice
    const double hbm = (mm && mm[1] > 0 ? mm[1] : 8000.0) * 1e9;
    const double cap = (mm && mm[2] > 0 ? mm[2] : 288e9) * 0.9;
    std::vector<int> rows((size_t) nproc + 1), xd((size_t) nproc + 1), best_rows;
    double best = 1e300, bt[3] = {0, 0, 0};
    int gm = nproc, gn = 1, tmp;
    const int reuse = rA > 0 ? rA : 1;
    for (int tn = 1; tn <= nproc; tn++)
    {
        if (nproc % tn != 0 || (tn > 1 && tn > n)) continue;
        const int tm = nproc / tn;
        for (int i = 0; i <= tm; i++) rows[(size_t) i] = rb_displs0[i * tn];
        if (m == k) memcpy(xd.data(), rows.data(), sizeof(int) * (size_t) (tm + 1));
        else for (int i = 0; i <= tm; i++) calc_block_spos_size(k, tm, i, xd.data() + i, &tmp);
        const int nl = (n + tn - 1) / tn;
        // per panel: nonzeros, distinct B rows, and B rows needed from every single peer
        double t_rep = 0, t_exch = 0, t_comp = 0;
        bool fits = true;
        std::vector<std::vector<int>> stamps((size_t) crp::host_threads());
        std::vector<double> p_rep((size_t) tm, 0.0), p_exch((size_t) tm, 0.0), p_comp((size_t) tm, 0.0);
        std::vector<char> p_fit((size_t) tm, 1);
        crp::parallel_chunks(tm, 1, [&](long long b0, long long b1, int tid) {
            std::vector<int> &stamp = stamps[(size_t) tid];
            if (stamp.empty()) stamp.assign((size_t) (k > 0 ? k : 1), 0);
            std::vector<long long> from((size_t) tm);
            for (int b = (int) b0; b < (int) b1; b++)
            {
                const int tag = b + 1;
                std::fill(from.begin(), from.end(), 0LL);
                long long distinct = 0;
                for (int p = rowptr[rows[(size_t) b]]; p < rowptr[rows[(size_t) b + 1]]; p++)
                {
                    const int c = colidx[p];
                    if (stamp[(size_t) c] == tag) continue;
                    stamp[(size_t) c] = tag;
                    distinct++;
                    const int q = (int) (std::upper_bound(xd.begin(), xd.begin() + tm + 1, c) - xd.begin()) - 1;
                    if (q != b && q >= 0 && q < tm) from[(size_t) q]++;
                }
                const double pnnz = (double) (rowptr[rows[(size_t) b + 1]] - rowptr[rows[(size_t) b]]);
                const double prow = (double) (rows[(size_t) b + 1] - rows[(size_t) b]);
                long long worst = 0;
                for (long long v : from) worst = std::max(worst, v);
                p_rep[(size_t) b] = tn > 1 ? 12.0 * pnnz / tn / link : 0.0;
                p_exch[(size_t) b] = 8.0 * (double) worst * nl / link;
                const double alg = 12.0 * pnnz + 4.0 * (prow + 1) + 8.0 * nl * (double) distinct + 8.0 * nl * prow;
                p_comp[(size_t) b] = alg / (hbm * kernel_fraction(nl));
                // device memory: CSR + derived formats (~3 x the CSR), local B / C blocks, receive buffer
                const double mem = 4.0 * 12.0 * pnnz + 8.0 * nl * (2.0 * prow + (double) distinct);
                if (mem > cap) p_fit[(size_t) b] = 0;
            }
        });
        for (int b = 0; b < tm; b++)
        {
            t_rep = std::max(t_rep, p_rep[(size_t) b]);
            t_exch = std::max(t_exch, p_exch[(size_t) b]);
            t_comp = std::max(t_comp, p_comp[(size_t) b]);
            fits = fits && p_fit[(size_t) b];
        }
        if (!fits) continue;
        const double price = t_rep / reuse + std::max(t_comp, t_exch);
        if (price < best)
        {
            best = price;
            gm = tm;
            gn = tn;
            bt[0] = t_rep; bt[1] = t_exch; bt[2] = t_comp;
            best_rows.assign(rows.begin(), rows.begin() + tm + 1);
        }
    }
    if (best_rows.empty())              // nothing fits by the model: the pure 1D row partition (least memory per GPU)
    {
        gm = nproc; gn = 1;
        best_rows.assign(rb_displs0, rb_displs0 + nproc + 1);
    }
    *pm = gm;
    *pn = gn;
    if (times) { times[0] = bt[0]; times[1] = bt[1]; times[2] = bt[2]; }
    part2d_emit(nproc, m, n, k, rowptr, gm, gn, best_rows.data(), A0_rowptr, B_rowptr, AC_rowptr, BC_colptr);
}

}  // extern "C"
