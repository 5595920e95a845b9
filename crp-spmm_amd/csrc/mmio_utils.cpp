// mmio_utils.cpp -- Matrix Market ingest behind include/mmio_utils.h.
//
// Accept / reject set, messages and output ordering follow the reference
// reader (/root/reference/examples/mmio_utils.c:11-125 on top of the NIST
// banner rules in examples/mmio.c:96-179), but the file is slurped once and
// tokenised with strtol/strtod instead of one fscanf per entry, and the COO ->
// CSR conversion sorts rows in parallel with a stable key sort.
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "mmio_utils.h"
#include "utils.h"
#include "par.h"

namespace {

struct Banner
{
    bool sparse = false;
    char dtype = 0;   // 'r' real, 'c' complex, 'p' pattern, 'i' integer
    char sym = 0;     // 'g' general, 's' symmetric, 'h' hermitian, 'k' skew-symmetric
};

std::string lower(std::string s)
{
    for (auto &ch : s) ch = (char) tolower((unsigned char) ch);
    return s;
}

// 0 ok, -1 not a processable banner (examples/mmio.c:96-179 error returns)
int parse_banner(const std::string &line, Banner *b)
{
    char t[5][64];
    if (sscanf(line.c_str(), "%63s %63s %63s %63s %63s", t[0], t[1], t[2], t[3], t[4]) != 5) return -1;
    if (strncmp(t[0], "%%MatrixMarket", 14) != 0) return -1;
    const std::string obj = lower(t[1]), fmt = lower(t[2]), dt = lower(t[3]), sy = lower(t[4]);
    if (obj != "matrix") return -1;
    if (fmt == "coordinate") b->sparse = true;
    else if (fmt == "array") b->sparse = false;
    else return -1;
    if (dt == "real") b->dtype = 'r';
    else if (dt == "complex") b->dtype = 'c';
    else if (dt == "pattern") b->dtype = 'p';
    else if (dt == "integer") b->dtype = 'i';
    else return -1;
    if (sy == "general") b->sym = 'g';
    else if (sy == "symmetric") b->sym = 's';
    else if (sy == "hermitian") b->sym = 'h';
    else if (sy == "skew-symmetric") b->sym = 'k';
    else return -1;
    return 0;
}

std::string typecode_str(const Banner &b)
{
    const char *dt = b.dtype == 'r' ? "real" : b.dtype == 'c' ? "complex" : b.dtype == 'p' ? "pattern" : "integer";
    const char *sy = b.sym == 'g' ? "general" : b.sym == 's' ? "symmetric" : b.sym == 'h' ? "hermitian" : "skew-symmetric";
    return std::string("matrix ") + (b.sparse ? "coordinate" : "array") + " " + dt + " " + sy;
}

}  // namespace

extern "C" {

int mm_read_sparse_RPI(const char *fname, const int need_symm, int *nrow_, int *ncol_, int *nnz_,
                       int **row_, int **col_, double **val_)
{
    FILE *f = fopen(fname, "rb");
    if (f == NULL) return -1;
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t) fsize + 1);
    const size_t got = fread(buf.data(), 1, (size_t) fsize, f);
    fclose(f);
    buf[got] = '\0';

    const char *p = buf.data(), *end = buf.data() + got;
    const char *eol = (const char *) memchr(p, '\n', (size_t) (end - p));
    std::string first(p, eol ? (size_t) (eol - p) : (size_t) (end - p));
    Banner bn;
    if (got == 0 || parse_banner(first, &bn) != 0)
    {
        printf("Could not process Matrix Market banner in file [%s]\n", fname);
        return -1;
    }
    const bool symm = (bn.sym == 's'), general = (bn.sym == 'g');
    if (need_symm && !symm)
    {
        fprintf(stderr, "The matrix is not symmetric.\n");
        return -1;
    }
    const bool ok = (bn.dtype == 'r' || bn.dtype == 'p' || bn.dtype == 'i') && bn.sparse && (general || symm);
    if (!ok)
    {
        fprintf(stderr, "Does not support Market Market type: [%s]\n", typecode_str(bn).c_str());
        return -1;
    }

    // comment lines, then the size line (blank lines before it are tolerated)
    p = eol ? eol + 1 : end;
    while (p < end && *p == '%')
    {
        const char *e = (const char *) memchr(p, '\n', (size_t) (end - p));
        p = e ? e + 1 : end;
    }
    char *q = NULL;
    long dims[3];
    for (int t = 0; t < 3; t++)
    {
        dims[t] = strtol(p, &q, 10);
        if (q == p)
        {
            fprintf(stderr, "Could not parse matrix size.\n");
            return -1;
        }
        p = q;
    }
    const int nrow = (int) dims[0], ncol = (int) dims[1], nnz = (int) dims[2];
    const size_t cap = (size_t) nnz * (symm ? 2 : 1);
    int *row = (int *) malloc(sizeof(int) * (cap ? cap : 1));
    int *col = (int *) malloc(sizeof(int) * (cap ? cap : 1));
    double *val = (double *) malloc(sizeof(double) * (cap ? cap : 1));
    ASSERT_PRINTF(row != NULL && col != NULL && val != NULL, "Failed to allocate COO arrays for %s\n", fname);

    // Large files: one entry per line is the rule, so the data region is cut at line ends into one
    // piece per host thread (count lines, prefix-sum, parse in place).  Anything else -- an entry
    // spread over two lines, extra tokens on a line, fewer lines than nnz -- falls back to the
    // token-by-token reader below, which accepts exactly what the reference's fscanf loop accepts.
    bool parsed = false;
    if (nnz >= 200000)
    {
        const int nt = crp::host_threads();
        std::vector<const char *> cut((size_t) nt + 1);
        cut[0] = p;
        cut[(size_t) nt] = end;
        for (int t = 1; t < nt; t++)
        {
            const char *c = p + (size_t) ((double) (end - p) * t / nt);
            const char *e = (const char *) memchr(c, '\n', (size_t) (end - c));
            cut[(size_t) t] = e ? e + 1 : end;
            if (cut[(size_t) t] < cut[(size_t) t - 1]) cut[(size_t) t] = cut[(size_t) t - 1];
        }
        auto blank = [](const char *a, const char *b) {
            for (; a < b; a++)
                if (*a != ' ' && *a != '\t' && *a != '\r') return false;
            return true;
        };
        std::vector<long long> lines((size_t) nt + 1, 0);
        crp::parallel_chunks(nt, 1, [&](long long b, long long e2, int) {
            for (long long t = b; t < e2; t++)
            {
                long long c = 0;
                for (const char *a = cut[(size_t) t]; a < cut[(size_t) t + 1];)
                {
                    const char *e = (const char *) memchr(a, '\n', (size_t) (cut[(size_t) t + 1] - a));
                    const char *le = e ? e : cut[(size_t) t + 1];
                    if (!blank(a, le)) c++;
                    a = e ? e + 1 : cut[(size_t) t + 1];
                }
                lines[(size_t) t + 1] = c;
            }
        });
        for (int t = 0; t < nt; t++) lines[(size_t) t + 1] += lines[(size_t) t];
        if (lines[(size_t) nt] >= nnz)
        {
            std::vector<int> bad_t((size_t) nt, 0);
            const char dtype = bn.dtype;
            crp::parallel_chunks(nt, 1, [&](long long b, long long e2, int) {
                for (long long t = b; t < e2; t++)
                {
                    long long g = lines[(size_t) t];
                    for (const char *a = cut[(size_t) t]; a < cut[(size_t) t + 1] && g < nnz;)
                    {
                        const char *e = (const char *) memchr(a, '\n', (size_t) (cut[(size_t) t + 1] - a));
                        const char *le = e ? e : cut[(size_t) t + 1];
                        if (!blank(a, le))
                        {
                            char *qq = NULL;
                            const long r = strtol(a, &qq, 10);
                            bool bad = (qq == a) || qq > le;
                            const char *c1 = qq;
                            const long c = strtol(c1, &qq, 10);
                            bad = bad || (qq == c1) || qq > le;
                            double v = 1.0;
                            if (!bad && dtype == 'r')
                            {
                                const char *c2 = qq;
                                v = strtod(c2, &qq);
                                bad = (qq == c2) || qq > le;
                            }
                            else if (!bad && dtype == 'i')
                            {
                                const char *c2 = qq;
                                v = (double) strtol(c2, &qq, 10);
                                bad = (qq == c2) || qq > le;
                            }
                            if (bad || !blank(qq, le)) { bad_t[(size_t) t] = 1; break; }
                            row[g] = (int) r - 1;
                            col[g] = (int) c - 1;
                            val[g] = v;
                            g++;
                        }
                        a = e ? e + 1 : cut[(size_t) t + 1];
                    }
                }
            });
            parsed = true;
            for (int t = 0; t < nt; t++) parsed = parsed && !bad_t[(size_t) t];
        }
    }
    if (!parsed)
    {
    for (int i = 0; i < nnz; i++)
        {
            const long r = strtol(p, &q, 10);
            bool bad = (q == p);
            p = q;
            const long c = strtol(p, &q, 10);
            bad = bad || (q == p);
            p = q;
            double v = 1.0;
            if (bn.dtype == 'r')
            {
                v = strtod(p, &q);
                bad = bad || (q == p);
                p = q;
            }
            else if (bn.dtype == 'i')
            {
                v = (double) strtol(p, &q, 10);
                bad = bad || (q == p);
                p = q;
            }
            if (bad)
            {
                // the reference would return uninitialised entries here; fail instead
                fprintf(stderr, "Premature end of Matrix Market data in file [%s] (entry %d of %d)\n", fname, i, nnz);
                free(row); free(col); free(val);
                return -1;
            }
            row[i] = (int) r - 1;
            col[i] = (int) c - 1;
            val[i] = v;
        }
    }
    int total = nnz;
    if (symm)
    {
        for (int i = 0; i < nnz; i++)
            if (row[i] != col[i])
            {
                row[total] = col[i];
                col[total] = row[i];
                val[total] = val[i];
                total++;
            }
    }
    *nrow_ = nrow;
    *ncol_ = ncol;
    *nnz_  = total;
    *row_  = row;
    *col_  = col;
    *val_  = val;
    return 0;
}

void coo2csr(const int nrow, const int ncol, const int nnz, const int *row, const int *col,
             const double *val, int **row_ptr_, int **col_idx_, double **csr_val_)
{
    (void) ncol;
    int *row_ptr = (int *) malloc(sizeof(int) * ((size_t) nrow + 1));
    int *col_idx = (int *) malloc(sizeof(int) * (size_t) (nnz > 0 ? nnz : 1));
    double *csr_val = (double *) malloc(sizeof(double) * (size_t) (nnz > 0 ? nnz : 1));
    ASSERT_PRINTF(row_ptr != NULL && col_idx != NULL && csr_val != NULL,
                  "Failed to allocate work arrays for %s\n", __FUNCTION__);

    // counting sort by row (keeps input order inside a row) ...
    std::vector<int> fill((size_t) nrow + 1, 0);
    for (int i = 0; i < nnz; i++) fill[row[i] + 1]++;
    row_ptr[0] = 0;
    for (int r = 0; r < nrow; r++) row_ptr[r + 1] = row_ptr[r] + fill[r + 1];
    for (int r = 0; r < nrow; r++) fill[r] = row_ptr[r];
    for (int i = 0; i < nnz; i++)
    {
        const int dst = fill[row[i]]++;
        col_idx[dst] = col[i];
        csr_val[dst] = val[i];
    }

    // ... then order every row by column; rows are independent
    crp::parallel_chunks(nrow, 4096, [&](long long r0, long long r1, int) {
        std::vector<std::pair<int, double>> tmp;
        for (long long r = r0; r < r1; r++)
        {
            const int s = row_ptr[r], e = row_ptr[r + 1];
            if (e - s < 2) continue;
            bool sorted = true;
            for (int p = s + 1; p < e && sorted; p++) sorted = col_idx[p - 1] <= col_idx[p];
            if (sorted) continue;
            tmp.resize((size_t) (e - s));
            for (int p = s; p < e; p++) tmp[p - s] = std::make_pair(col_idx[p], csr_val[p]);
            std::stable_sort(tmp.begin(), tmp.end(),
                             [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.first < b.first; });
            for (int p = s; p < e; p++)
            {
                col_idx[p] = tmp[p - s].first;
                csr_val[p] = tmp[p - s].second;
            }
        }
    });
    *row_ptr_ = row_ptr;
    *col_idx_ = col_idx;
    *csr_val_ = csr_val;
}

// ---- binary CSR cache (extension; include/crp_engine.h) -------------------------------------------
// Text ingest of a 400 M-line file takes minutes even parsed in parallel; a converted matrix is
// kept next to it as one little-endian file: magic "CRPCSR01", int64 nrow, ncol, nnz, then rowptr
// (int32 x (nrow + 1)), colidx (int32 x nnz), val (fp64 x nnz).
int crp_csr_cache_write(const char *fname, int nrow, int ncol, const int *rowptr, const int *colidx, const double *val)
{
    if (fname == NULL || nrow < 0 || rowptr == NULL) return -1;
    FILE *f = fopen(fname, "wb");
    if (f == NULL) return -1;
    const long long hdr[3] = {nrow, ncol, rowptr[nrow]};
    const size_t nnz = (size_t) rowptr[nrow];
    bool ok = fwrite("CRPCSR01", 1, 8, f) == 8 && fwrite(hdr, sizeof(long long), 3, f) == 3;
    ok = ok && fwrite(rowptr, sizeof(int), (size_t) nrow + 1, f) == (size_t) nrow + 1;
    ok = ok && (nnz == 0 || (fwrite(colidx, sizeof(int), nnz, f) == nnz && fwrite(val, sizeof(double), nnz, f) == nnz));
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -1;
}

int crp_csr_cache_read(const char *fname, int *nrow_, int *ncol_, int **rowptr_, int **colidx_, double **val_)
{
    if (fname == NULL || !nrow_ || !ncol_ || !rowptr_ || !colidx_ || !val_) return -1;
    FILE *f = fopen(fname, "rb");
    if (f == NULL) return -1;
    char magic[8];
    long long hdr[3];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "CRPCSR01", 8) != 0 || fread(hdr, sizeof(long long), 3, f) != 3 ||
        hdr[0] < 0 || hdr[1] < 0 || hdr[2] < 0 || hdr[0] > 2147483646LL || hdr[2] > 2147483647LL)
    {
        fclose(f);
        return -1;
    }
    const size_t nrow = (size_t) hdr[0], nnz = (size_t) hdr[2];
    int *rp = (int *) malloc(sizeof(int) * (nrow + 1));
    int *ci = (int *) malloc(sizeof(int) * (nnz ? nnz : 1));
    double *va = (double *) malloc(sizeof(double) * (nnz ? nnz : 1));
    bool ok = rp && ci && va && fread(rp, sizeof(int), nrow + 1, f) == nrow + 1;
    ok = ok && (nnz == 0 || (fread(ci, sizeof(int), nnz, f) == nnz && fread(va, sizeof(double), nnz, f) == nnz));
    fclose(f);
    ok = ok && rp[0] == 0 && (long long) rp[nrow] == hdr[2];
    if (!ok)
    {
        free(rp); free(ci); free(va);
        return -1;
    }
    *nrow_ = (int) hdr[0];
    *ncol_ = (int) hdr[1];
    *rowptr_ = rp;
    *colidx_ = ci;
    *val_ = va;
    return 0;
}

}  // extern "C"
