/*
 * crp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's algorithms on the CRP-SpMM hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this; the product library (crp-spmm_amd/lib/libcrpspmm_hip.so)
 * never links or calls it.
 *
 * Pinning (see oracle/README.md, DESIGN.md "Oracle"):
 *   - planner / ingest functions are checked against the reference's own
 *     sources compiled unmodified into oracle/_ref/libcrpref.so and against
 *     the committed golden fixtures in tests/golden/;
 *   - the SpMM arithmetic lives in Intel MKL (mkl_sparse_d_mm, un-vendored,
 *     version unpinned by the reference; the image carries 2021.4 as a
 *     runtime only).  orc_spmm_csr_f64 restates its published definition
 *     (C = 1.0 * A * B + 0.0 * C) and is pinned by golden C matrices produced
 *     by that MKL through ctypes with the reference's argument set
 *     (tests/golden/make_golden.py) and by the closed-form answer for the
 *     reference's fill_B operand.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- src/utils.c:26-48 calc_block_spos_size ------------------------------ */
void orc_block_spos_size(int len, int nblk, int iblk, int *spos, int *size)
{
    if (iblk < 0 || iblk > nblk) { *spos = -1; *size = 0; return; }
    int rem = len % nblk, bs0 = len / nblk;
    if (iblk < rem) { *spos = (bs0 + 1) * iblk; *size = bs0 + 1; }
    else            { *spos = bs0 * iblk + rem; *size = bs0; }
}

/* ---- src/utils.c:75-89 calc_err_2norm (naive sums, same order) ------------ */
void orc_err_2norm(long long len, const double *x0, const double *x1, double *x0_2norm, double *err_2norm)
{
    double a = 0.0, e = 0.0;
    for (long long i = 0; i < len; i++)
    {
        double d = x0[i] - x1[i];
        a += x0[i] * x0[i];
        e += d * d;
    }
    *x0_2norm = sqrt(a);
    *err_2norm = sqrt(e);
}

/* ---- examples/test_utils.c:121-154 fill_B -------------------------------- */
void orc_fill_B(int layout, double *B, long long ldB, int srow, int nrow, int scol, int ncol, double fi, double fj)
{
    for (int i = 0; i < nrow; i++)
        for (int j = 0; j < ncol; j++)
        {
            double v = (srow + i) * fi + (scol + j) * fj;
            if (layout == 0) B[(long long) i * ldB + j] = v;
            else             B[(long long) j * ldB + i] = v;
        }
}

/* ---- src/rowpara_spmm.c:388-408 / examples/test_utils.c:157-179 -----------
 * mkl_sparse_d_mm(NON_TRANSPOSE, alpha = 1, A (CSR, base 0, general), layout,
 * B, n, ldB, beta = 0, C, ldC): C[i][j] = sum_p val[p] * B[col[p]][j].
 * Summation in ascending p, one rounding per product and per add
 * (build with -ffp-contract=off).  C is overwritten, never read (beta = 0). */
void orc_spmm_csr_f64(int m, int n, const int *rowptr, const int *colidx, const double *val, int layout,
                      const double *B, long long ldB, double *C, long long ldC)
{
    for (int i = 0; i < m; i++)
    {
        for (int j = 0; j < n; j++)
        {
            double acc = 0.0;
            for (int p = rowptr[i]; p < rowptr[i + 1]; p++)
            {
                long long c = colidx[p];
                double b = (layout == 0) ? B[c * ldB + j] : B[(long long) j * ldB + c];
                acc += val[p] * b;
            }
            if (layout == 0) C[(long long) i * ldC + j] = acc;
            else             C[(long long) j * ldC + i] = acc;
        }
    }
}

/* Row-major fast form of the same sum (j innermost) used as bench.py's CPU
 * baseline ("port"); OpenMP over rows when built with -fopenmp. */
void orc_spmm_csr_f64_rm_fast(int m, int n, const int *rowptr, const int *colidx, const double *val,
                              const double *B, long long ldB, double *C, long long ldC)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < m; i++)
    {
        double *Ci = C + (long long) i * ldC;
        for (int j = 0; j < n; j++) Ci[j] = 0.0;
        for (int p = rowptr[i]; p < rowptr[i + 1]; p++)
        {
            const double a = val[p];
            const double *Bc = B + (long long) colidx[p] * ldB;
            for (int j = 0; j < n; j++) Ci[j] += a * Bc[j];
        }
    }
}

/* ---- examples/mmio_utils.c:127-145 qsort_ascend_int_dbl_pair ---------------
 * Same Hoare partition / middle pivot so that the order of duplicate
 * (row, col) entries matches the reference bit for bit. */
static void orc_qsort_pair(int *key, double *val, int l, int r)
{
    int i = l, j = r, tk;
    int mid = key[(l + r) / 2];
    double tv;
    while (i <= j)
    {
        while (key[i] < mid) i++;
        while (key[j] > mid) j--;
        if (i <= j)
        {
            tk = key[i]; key[i] = key[j]; key[j] = tk;
            tv = val[i]; val[i] = val[j]; val[j] = tv;
            i++; j--;
        }
    }
    if (i < r) orc_qsort_pair(key, val, i, r);
    if (j > l) orc_qsort_pair(key, val, l, j);
}

/* ---- examples/mmio_utils.c:148-190 coo2csr --------------------------------- */
void orc_coo2csr(int nrow, int nnz, const int *row, const int *col, const double *val,
                 int *row_ptr, int *col_idx, double *csr_val)
{
    memset(row_ptr, 0, sizeof(int) * ((size_t) nrow + 1));
    for (int i = 0; i < nnz; i++) row_ptr[row[i] + 1]++;
    for (int i = 2; i <= nrow; i++) row_ptr[i] += row_ptr[i - 1];
    for (int i = 0; i < nnz; i++)
    {
        int idx = row_ptr[row[i]];
        col_idx[idx] = col[i];
        csr_val[idx] = val[i];
        row_ptr[row[i]]++;
    }
    for (int i = nrow; i >= 1; i--) row_ptr[i] = row_ptr[i - 1];
    row_ptr[0] = 0;
    for (int i = 0; i < nrow; i++)
        if (row_ptr[i + 1] - 1 > row_ptr[i])   /* the reference calls it on empty rows too: l > r is a no-op */
            orc_qsort_pair(col_idx, csr_val, row_ptr[i], row_ptr[i + 1] - 1);
}

/* ---- examples/mmio_utils.c:100-117: mirror the off-diagonals of a symmetric
 * coordinate list.  row/col/val have room for 2*nnz entries; returns new nnz. */
int orc_symm_expand(int nnz, int *row, int *col, double *val)
{
    int idx = nnz;
    for (int i = 0; i < nnz; i++)
        if (row[i] != col[i])
        {
            row[idx] = col[i];
            col[idx] = row[i];
            val[idx] = val[i];
            idx++;
        }
    return idx;
}

/* ---- src/spmat_part.c:12-35 csr_mat_row_partition --------------------------- */
void orc_csr_row_partition(int nrow, const int *row_ptr, int nblk, int *rblk_ptr)
{
    int nnz = row_ptr[nrow];
    rblk_ptr[0] = 0;
    for (int i = 0; i < nblk; i++)
    {
        int target = (nnz / nblk) * (i + 1);
        if (i == nblk - 1) target = nnz;
        int st = 0, end = nrow;
        while (st < end)
        {
            int mid = (st + end) / 2;
            if (row_ptr[mid] == target) { st = mid; break; }
            if (row_ptr[mid] < target) st = mid + 1;
            else end = mid;
        }
        rblk_ptr[i + 1] = st;
    }
}

/* ---- src/spmat_part.c:38-64 csr_mat_row_part_comm_size ---------------------- */
void orc_csr_row_part_comm_size(int nrow, int ncol, const int *row_ptr, const int *col_idx, int nblk,
                                const int *rblk_ptr, const int *x_displs, int *comm_sizes, int *total_size)
{
    char *flag = (char *) malloc((size_t) (ncol > 0 ? ncol : 1));
    (void) nrow;
    for (int b = 0; b < nblk; b++)
    {
        int cnt = 0;
        memset(flag, 0, (size_t) ncol);
        for (int j = row_ptr[rblk_ptr[b]]; j < row_ptr[rblk_ptr[b + 1]]; j++) flag[col_idx[j]] = 1;
        for (int i = 0; i < ncol; i++) cnt += flag[i];
        for (int i = x_displs[b]; i < x_displs[b + 1]; i++) cnt -= flag[i];
        comm_sizes[b] = cnt;
    }
    *total_size = 0;
    for (int b = 0; b < nblk; b++) *total_size += comm_sizes[b];
    free(flag);
}

/* ---- src/spmat_part.c:66-81 prime_factorization ------------------------------ */
int orc_prime_factorization(int n, int *fac /* room for 32 */)
{
    int nfac = 0, c = 2;
    while (n > 1)
    {
        if (n % c == 0) { fac[nfac++] = c; n /= c; }
        else c++;
    }
    return nfac;
}

/* ---- src/spmat_part.c:85-210 calc_spmm_part2d_from_1d -------------------------
 * Outputs are caller-allocated: A0_rowptr[nproc+1], B_rowptr / AC_rowptr
 * [nproc+1] (pm+1 used), BC_colptr[nproc+1] (pn+1 used). */
void orc_part2d_from_1d(int nproc, int m, int n, int k, const int *rb_displs0, const int *rowptr,
                        const int *colidx, int rA, int *pm_out, int *pn_out, unsigned long long *comm_cost,
                        int *A0_rowptr, int *B_rowptr, int *AC_rowptr, int *BC_colptr)
{
    const double nnz_cf = 1.5;
    int *m_displs   = (int *) malloc(sizeof(int) * (nproc + 1));
    int *m_displs2  = (int *) malloc(sizeof(int) * (nproc + 1));
    int *k_displs   = (int *) malloc(sizeof(int) * (nproc + 1));
    int *comm_sizes = (int *) malloc(sizeof(int) * nproc);
    int tmp, fac[32];

    if (m == k) memcpy(k_displs, rb_displs0, sizeof(int) * (nproc + 1));
    else for (int i = 0; i <= nproc; i++) orc_block_spos_size(k, nproc, i, k_displs + i, &tmp);
    orc_csr_row_part_comm_size(m, k, rowptr, colidx, nproc, rb_displs0, k_displs, comm_sizes, &tmp);
    size_t best_cost = (size_t) tmp * (size_t) n;
    memcpy(m_displs, rb_displs0, sizeof(int) * (nproc + 1));

    int pm_ = nproc, pn_ = 1, failed_p = -1, A_nnz = rowptr[m];
    int nfac = orc_prime_factorization(nproc, fac);
    for (int ifac = 0; ifac < nfac; ifac++)
    {
        int p_i = fac[nfac - 1 - ifac];
        if (p_i == failed_p) continue;
        int pn2 = pn_ * p_i, pm2 = nproc / pn2;
        for (int i = 0; i <= pm2; i++) m_displs2[i] = rb_displs0[i * pn2];
        if (m == k) memcpy(k_displs, m_displs2, sizeof(int) * (pm2 + 1));
        else for (int i = 0; i <= pm2; i++) orc_block_spos_size(k, pm2, i, k_displs + i, &tmp);
        orc_csr_row_part_comm_size(m, k, rowptr, colidx, pm2, m_displs2, k_displs, comm_sizes, &tmp);
        size_t A_copy = (size_t) ((double) A_nnz * (double) (pn2 - 1) * nnz_cf);
        size_t B_copy = (size_t) rA * (size_t) tmp * (size_t) n;
        size_t cur = A_copy + B_copy;
        if (cur < best_cost)
        {
            best_cost = cur; pn_ = pn2; pm_ = pm2;
            memcpy(m_displs, m_displs2, sizeof(int) * (pm2 + 1));
            failed_p = -1;
        }
        else failed_p = p_i;
    }
    *comm_cost = (unsigned long long) best_cost;
    *pm_out = pm_;
    *pn_out = pn_;
    memcpy(AC_rowptr, m_displs, sizeof(int) * (pm_ + 1));
    if (m == k) memcpy(B_rowptr, AC_rowptr, sizeof(int) * (pm_ + 1));
    else for (int i = 0; i <= pm_; i++) orc_block_spos_size(k, pm_, i, B_rowptr + i, &tmp);
    for (int i = 0; i <= pn_; i++) orc_block_spos_size(n, pn_, i, BC_colptr + i, &tmp);

    int *tmp_rowptr = (int *) malloc(sizeof(int) * ((size_t) m + 1));
    for (int im = 0; im < pm_; im++)
    {
        int srow = m_displs[im], erow = m_displs[im + 1];
        for (int i = srow; i <= erow; i++) tmp_rowptr[i - srow] = rowptr[i] - rowptr[srow];
        int *A0_i = A0_rowptr + im * pn_;
        orc_csr_row_partition(erow - srow, tmp_rowptr, pn_, A0_i);
        for (int j = 0; j <= pn_; j++) A0_i[j] += srow;
    }
    free(m_displs); free(m_displs2); free(k_displs); free(comm_sizes); free(tmp_rowptr);
}

/* ---- src/mat_redist.c:9-29 calc_seg_intersection (inclusive ends) ------------- */
static void orc_seg_intersection(int s0, int e0, int s1, int e1, int *hit, int *is, int *ie)
{
    if (s0 > s1)
    {
        int t;
        t = s0; s0 = s1; s1 = t;
        t = e0; e0 = e1; e1 = t;
    }
    if (s1 > e0 || s1 > e1 || s0 > e0) { *hit = 0; *is = -1; *ie = -1; return; }
    *hit = 1;
    *is = s1;
    *ie = (e0 < e1) ? e0 : e1;
}

/* ---- src/mat_redist.c:31-41 calc_rect_intersection ------------------------------ */
void orc_rect_intersection(int xs0, int xe0, int ys0, int ye0, int xs1, int xe1, int ys1, int ye1,
                           int *hit, int *ixs, int *ixe, int *iys, int *iye)
{
    orc_seg_intersection(xs0, xe0, xs1, xe1, hit, ixs, ixe);
    if (*hit == 0) return;
    orc_seg_intersection(ys0, ye0, ys1, ye1, hit, iys, iye);
}
