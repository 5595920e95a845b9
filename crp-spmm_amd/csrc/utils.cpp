// utils.cpp -- host helpers behind include/utils.h
// (behaviour of /root/reference/src/utils.c:15-162).
#include <math.h>
#include <string.h>
#include <sys/time.h>
#include "utils.h"
#include "par.h"

extern "C" {

double get_wtime_sec(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double) tv.tv_sec + (double) tv.tv_usec * 1e-6;
}

void calc_block_spos_size(const int len, const int nblk, const int iblk, int *blk_spos, int *blk_size)
{
    if (iblk < 0 || iblk > nblk)
    {
        *blk_spos = -1;
        *blk_size = 0;
        return;
    }
    const int q = len / nblk, r = len % nblk;
    // the first r blocks hold q + 1 elements
    *blk_size = (iblk < r) ? q + 1 : q;
    *blk_spos = (iblk < r) ? (q + 1) * iblk : q * iblk + r;
}

void *malloc_aligned(size_t size, size_t alignment)
{
    void *p = NULL;
    if (posix_memalign(&p, alignment, size) != 0) return NULL;
    return p;
}

void free_aligned(void *mem) { free(mem); }

double calc_2norm(const int len, const double *x)
{
    double s = 0.0;
    for (int i = 0; i < len; i++) s += x[i] * x[i];
    return sqrt(s);
}

void calc_err_2norm(const int len, const double *x0, const double *x1, double *x0_2norm_, double *err_2norm_)
{
    double s0 = 0.0, se = 0.0;
    for (int i = 0; i < len; i++)
    {
        const double d = x0[i] - x1[i];
        s0 += x0[i] * x0[i];
        se += d * d;
    }
    *x0_2norm_  = sqrt(s0);
    *err_2norm_ = sqrt(se);
}

void copy_matrix(const size_t dt_size, const int nrow, const int ncol, const void *src, const int lds,
                 void *dst, const int ldd, const int use_omp)
{
    const char *s = (const char *) src;
    char *d = (char *) dst;
    const size_t sp = dt_size * (size_t) lds, dp = dt_size * (size_t) ldd, rb = dt_size * (size_t) ncol;
    auto rows = [&](long long b, long long e, int) {
        for (long long r = b; r < e; r++) memcpy(d + (size_t) r * dp, s + (size_t) r * sp, rb);
    };
    if (use_omp) crp::parallel_chunks(nrow, 1024, rows);
    else rows(0, nrow, 0);
}

void print_matrix(const int dtype, const int stype, const void *mat, const int ldm, const int nrow,
                  const int ncol, const char *fmt, const char *name)
{
    printf("%s:\n", name);
    const size_t rs = (stype == 0) ? (size_t) ldm : 1, cs = (stype == 0) ? 1 : (size_t) ldm;
    for (int i = 0; i < nrow; i++)
    {
        for (int j = 0; j < ncol; j++)
        {
            const size_t off = (size_t) i * rs + (size_t) j * cs;
            if (dtype == 0) printf(fmt, ((const int *) mat)[off]);
            if (dtype == 1) printf(fmt, ((const double *) mat)[off]);
        }
        printf("\n");
    }
}

void dump_binary(const char *fname, void *data, const size_t bytes)
{
    FILE *fp = fopen(fname, "wb");
    if (fp == NULL) return;
    fwrite(data, 1, bytes, fp);
    fclose(fp);
}

}  // extern "C"
