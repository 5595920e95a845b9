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

}  // extern "C"
