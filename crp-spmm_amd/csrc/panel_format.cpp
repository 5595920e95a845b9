// panel_format.cpp -- host construction of the row-panel format (panel_format.h).
#include <algorithm>
#include <string.h>
#include "panel_format.h"
#include "par.h"

namespace crp {

namespace {

// ordering key of a two-source column index: receive-buffer rows (c < 0, ~c ascending)
// first, then local rows ascending.  With one rank this is plain column order, i.e. the
// CSR order of the reference.
inline uint32_t col_key(int c) { return c < 0 ? (uint32_t) (~c) : ((uint32_t) c | 0x80000000u); }

struct Trip
{
    uint32_t key;
    int      col;
    int      r;
    int      p;    // position in the CSR arrays (keeps duplicates in input order)
};

// Visits the entries of one panel in order; fn(col, rmask, pos[R]) with pos[r] = CSR
// position of row r's value or -1.
template <typename F>
void panel_entries(int nrow, const int *rowptr, const int *colidx, int R, int panel, std::vector<Trip> &tmp, F fn)
{
    const int r0 = panel * R, r1 = std::min(nrow, r0 + R);
    tmp.clear();
    for (int r = r0; r < r1; r++)
        for (int p = rowptr[r]; p < rowptr[r + 1]; p++) tmp.push_back({col_key(colidx[p]), colidx[p], r - r0, p});
    std::sort(tmp.begin(), tmp.end(), [](const Trip &a, const Trip &b) {
        if (a.key != b.key) return a.key < b.key;
        if (a.r != b.r) return a.r < b.r;
        return a.p < b.p;
    });
    int pos[16];
    size_t i = 0;
    while (i < tmp.size())
    {
        // [i, j): every (row, value) of one column, ordered by (row, input position).
        // The k-th occurrence of the column inside a row goes to the k-th entry, so a
        // row that repeats a column (duplicates are legal: examples/mmio_utils.c keeps
        // them) opens further entries and keeps its input order.
        size_t j = i;
        int max_occ = 0, occ = 0;
        const uint32_t k0 = tmp[i].key;
        while (j < tmp.size() && tmp[j].key == k0)
        {
            occ = (j > i && tmp[j - 1].r == tmp[j].r) ? occ + 1 : 0;
            tmp[j].key = (uint32_t) occ;      // key is not needed any more inside the group
            if (occ > max_occ) max_occ = occ;
            j++;
        }
        for (int e = 0; e <= max_occ; e++)
        {
            unsigned mask = 0;
            for (int r = 0; r < R; r++) pos[r] = -1;
            for (size_t t = i; t < j; t++)
                if ((int) tmp[t].key == e)
                {
                    mask |= 1u << tmp[t].r;
                    pos[tmp[t].r] = tmp[t].p;
                }
            fn(tmp[i].col, mask, pos);
        }
        i = j;
    }
}

}  // namespace

double PanelHost::fill() const
{
    return real_entries > 0 ? (double) nnz / ((double) real_entries * R) : 1.0;
}

long long count_panel_entries(int nrow, const int *rowptr, const int *colidx, int R)
{
    const int npanel = (nrow + R - 1) / R;
    std::vector<long long> part((size_t) host_threads(), 0);
    parallel_chunks(npanel, 512, [&](long long b, long long e, int tid) {
        std::vector<Trip> tmp;
        long long cnt = 0;
        for (long long pn = b; pn < e; pn++)
            panel_entries(nrow, rowptr, colidx, R, (int) pn, tmp, [&](int, unsigned, const int *) { cnt++; });
        part[(size_t) tid] += cnt;
    });
    long long tot = 0;
    for (long long v : part) tot += v;
    return tot;
}

void build_panels(int nrow, const int *rowptr, const int *colidx, const double *val, int R, PanelHost *out)
{
    const int npanel = (nrow + R - 1) / R;
    out->R = R;
    out->npanel = npanel;
    out->nnz = rowptr[nrow];
    out->pptr.assign((size_t) npanel + 1, 0);
    // pass 1: entry counts per panel
    std::vector<int> cnt((size_t) npanel, 0);
    parallel_chunks(npanel, 512, [&](long long b, long long e, int) {
        std::vector<Trip> tmp;
        for (long long pn = b; pn < e; pn++)
        {
            int c = 0;
            panel_entries(nrow, rowptr, colidx, R, (int) pn, tmp, [&](int, unsigned, const int *) { c++; });
            cnt[(size_t) pn] = c;
        }
    });
    long long real = 0;
    for (int pn = 0; pn < npanel; pn++)
    {
        real += cnt[pn];
        const int padded = (cnt[pn] + PANEL_PAD - 1) / PANEL_PAD * PANEL_PAD;
        out->pptr[pn + 1] = out->pptr[pn] + padded;
    }
    out->real_entries = real;
    const size_t total = (size_t) out->pptr[npanel];
    out->pcol.assign(total, 0);
    out->pmask4.assign(total / 4 + 2, 0u);
    out->pval.assign(total * (size_t) R, 0.0);
    // pass 2: fill
    parallel_chunks(npanel, 512, [&](long long b, long long e, int) {
        std::vector<Trip> tmp;
        for (long long pn = b; pn < e; pn++)
        {
            size_t q = (size_t) out->pptr[pn];
            const size_t qend = (size_t) out->pptr[pn + 1];
            int last_col = 0;
            panel_entries(nrow, rowptr, colidx, R, (int) pn, tmp, [&](int col, unsigned mask, const int *pos) {
                out->pcol[q] = col;
                // pmask4 words are private to a panel: panel starts are multiples of 4
                out->pmask4[q >> 2] |= (mask & 0xFFu) << (8 * (q & 3));
                for (int r = 0; r < R; r++)
                    if (pos[r] >= 0) out->pval[q * (size_t) R + r] = val[pos[r]];
                last_col = col;
                q++;
            });
            for (; q < qend; q++) out->pcol[q] = last_col;   // padding: valid address, mask 0
        }
    });
}

}  // namespace crp
