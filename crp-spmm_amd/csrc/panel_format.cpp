// panel_format.cpp -- host construction of the row-panel format (panel_format.h).
#include <algorithm>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "locality.h"
#include "panel_format.h"
#include "team_order.h"
#include "knobs.h"
#include "par.h"
#include <time.h>


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

bool build_compact_values(PanelHost *p)
{
    if (p->R != 8) return false;
    const int np = p->npanel;
    auto mask_of = [&](size_t q) { return (p->pmask4[q >> 2] >> (8 * (q & 3))) & 0xFFu; };
    p->cbase.assign((size_t) np + 1, 0);
    std::vector<long long> cnt((size_t) np, 0);
    bool fits = true;
    parallel_chunks(np, 4096, [&](long long b, long long e, int) {
        for (long long pn = b; pn < e; pn++)
        {
            long long c = 0;
            for (int q = p->pptr[(size_t) pn]; q < p->pptr[(size_t) pn + 1]; q++) c += __builtin_popcount(mask_of((size_t) q));
            cnt[(size_t) pn] = c;
        }
    });
    for (int pn = 0; pn < np; pn++)
    {
        if (cnt[(size_t) pn] >= (1LL << 24)) fits = false;
        p->cbase[(size_t) pn + 1] = p->cbase[(size_t) pn] + cnt[(size_t) pn];
    }
    const long long total = p->cbase[(size_t) np];
    if (!fits || total >= (1LL << 32)) { p->cbase.clear(); return false; }
    const size_t nent = p->pcol.size();
    p->cmo.resize(nent);
    parallel_fill(p->cval, (size_t) total + 16, 0.0);
    big_vector<uint32_t> ebase;                       // absolute index of every entry's first value
    ebase.resize(nent);
    parallel_chunks(np, 4096, [&](long long b, long long e, int) {
        for (long long pn = b; pn < e; pn++)
        {
            long long off = 0;
            for (int q = p->pptr[(size_t) pn]; q < p->pptr[(size_t) pn + 1]; q++)
            {
                const unsigned m = mask_of((size_t) q);
                p->cmo[(size_t) q] = m | ((uint32_t) off << 8);
                ebase[(size_t) q] = (uint32_t) (p->cbase[(size_t) pn] + off);
                for (int r = 0; r < 8; r++)
                    if ((m >> r) & 1u) p->cval[(size_t) (p->cbase[(size_t) pn] + off++)] = p->pval[(size_t) q * 8 + (size_t) r];
            }
        }
    });
    p->cmap.resize(p->pmap.size());
    parallel_chunks((long long) p->pmap.size(), 1 << 18, [&](long long b, long long e, int) {
        for (long long nz = b; nz < e; nz++)
        {
            const uint32_t sl = p->pmap[(size_t) nz];
            const size_t q = sl >> 3;
            const unsigned r = sl & 7u;
            p->cmap[(size_t) nz] = ebase[q] + (uint32_t) __builtin_popcount(mask_of(q) & ((1u << r) - 1u));
        }
    });
    return true;
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

long long count_block_union(int nrow, const int *rowptr, const int *colidx, int block)
{
    const int nblk = (nrow + block - 1) / block;
    std::vector<long long> part((size_t) host_threads(), 0);
    parallel_chunks(nblk, 256, [&](long long b, long long e, int tid) {
        std::vector<int> tmp;
        long long cnt = 0;
        for (long long g = b; g < e; g++)
        {
            const int r0 = (int) g * block, r1 = std::min(nrow, r0 + block);
            tmp.assign(colidx + rowptr[r0], colidx + rowptr[r1]);
            std::sort(tmp.begin(), tmp.end());
            cnt += (long long) (std::unique(tmp.begin(), tmp.end()) - tmp.begin());
        }
        part[(size_t) tid] += cnt;
    });
    long long tot = 0;
    for (long long v : part) tot += v;
    return tot;
}

void build_panels(int nrow, const int *rowptr, const int *colidx, const double *val, int R, PanelHost *out,
                  bool team_schedule, bool need_order)
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
    parallel_fill(out->pcol, total, 0);
    parallel_fill(out->pmask4, total / 4 + 2, 0u);
    const bool with_vals = val != nullptr;
    if (with_vals) parallel_fill(out->pval, total * (size_t) R, 0.0);
    parallel_fill(out->pmap, (size_t) rowptr[nrow], 0u);
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
                    if (pos[r] >= 0)
                    {
                        if (with_vals) out->pval[q * (size_t) R + r] = val[pos[r]];
                        out->pmap[(size_t) pos[r]] = (uint32_t) (q * (size_t) R + r);
                    }
                last_col = col;
                q++;
            });
            for (; q < qend; q++) out->pcol[q] = last_col;   // padding: valid address, mask 0
        }
    });
    // processing order.  CRPSPMM_PANEL_ORDER: 0 natural, 1 breadth-first groups, 2 stride lattice of panels,
    // 3 team schedule (2 and 3 only when the matrix has a stride lattice, else natural); unset = team
    // schedule (R = 8) or panel lattice (R = 4) when a lattice is detected, else breadth-first groups.
    out->porder.clear();
    out->psync.clear();
    if (!need_order) return;
    const int group = 16;                                       // panels per breadth-first group
    const int mode = knobs().panel_order;
    const int chunk = ((((npanel + 3) / 4) + 7) / 8) * 4;      // order positions per XCD (the kernels' block -> XCD map)
    bool done = false;
    out->psync.clear();
    if ((mode == 3 || mode == -1) && R == 8 && team_schedule && npanel >= 64)
    {
        double D1, D2;
        int M;
        if (detect_stride_lattice(nrow, rowptr, colidx, R, &D1, &D2, &M))
        {
            // workgroups of four waves = 2 x 2 teeth (six-wave workgroups, 3 x 2 teeth, were measured slower in round 1: 0.44
            // against 0.34 ms on the pwtk stand-in -- at 3 waves per SIMD only one six-wave workgroup fits a CU)
            TeamHost th;
            build_teams(*out, nrow, rowptr, colidx, &th, 4);
            apply_team_schedule(out, th);
            done = true;
        }
    }
    if (!done && (mode == 2 || mode == 3 || mode == -1))
        done = stride_lattice_order(nrow, rowptr, colidx, R, npanel, chunk, &out->porder);
    if (!done && (mode == 0 || mode == 2 || mode == 3))
    {
        out->porder.resize((size_t) npanel);
        for (int i = 0; i < npanel; i++) out->porder[(size_t) i] = i;
        done = true;
    }
    if (!done) locality_order(*out, group, &out->porder);
}

bool detect_stride_lattice(int nrow, const int *rowptr, const int *colidx, int R, double *D1_, double *D2_, int *M_)
{
    if (nrow < 4096) return false;
    // histogram of |col - row| over the locally owned columns, 64-row buckets
    // 8-row buckets up to 2 M rows (the strides of a grid with short lines -- 56 nodes x 3 unknowns = 168 rows -- then
    // separate from the near band: fem3d stand-in, lattice teams of 2 x 2 lines x 2 panels need 5.7 union entries per row
    // against 6.95 for clusters, 0.913 -> 0.838 ms at n = 256), 64-row buckets beyond (one histogram per thread).
    const int SH = nrow <= (1 << 21) ? 3 : 6;                   // log2 of the bucket width
    const size_t nb = ((size_t) nrow >> SH) + 2;
    const int nt = host_threads();
    std::vector<std::vector<long long>> cnt_t((size_t) nt, std::vector<long long>(nb, 0)), sum_t(cnt_t);
    parallel_chunks(nrow, 4096, [&](long long b, long long e, int tid) {
        std::vector<long long> &cnt = cnt_t[(size_t) tid], &sum = sum_t[(size_t) tid];
        for (long long r = b; r < e; r++)
            for (int p = rowptr[r]; p < rowptr[r + 1]; p++)
            {
                const int c = colidx[p];
                if (c < 0) continue;
                long long d = (long long) c - r;
                if (d < 0) d = -d;
                const size_t k = std::min((size_t) (d >> SH), nb - 1);
                cnt[k]++;
                sum[k] += d;
            }
    });
    std::vector<long long> cnt(nb, 0), sum(nb, 0);
    long long total = 0;
    for (int t = 0; t < nt; t++)
        for (size_t k = 0; k < nb; k++) { cnt[k] += cnt_t[(size_t) t][k]; sum[k] += sum_t[(size_t) t][k]; }
    for (size_t k = 0; k < nb; k++) total += cnt[k];
    if (total == 0) return false;
    // runs of non-empty buckets (one empty bucket allowed inside a run)
    struct Run { long long w; double center; };
    std::vector<Run> far;
    for (size_t k = 0; k < nb;)
    {
        if (cnt[k] == 0) { k++; continue; }
        size_t e = k;
        long long w = 0, sm = 0;
        while (e < nb && (cnt[e] > 0 || (e + 1 < nb && cnt[e + 1] > 0)))
        {
            w += cnt[e];
            sm += sum[e];
            e++;
        }
        if (k > 0 && w * 100 >= total * 2) far.push_back({w, (double) sm / (double) w});
        k = e;
    }
    // D1 = the nearest far cluster that carries >= 6 % of the nonzeros.  The outer stride may show up as
    // several clusters (a 27-point stencil has nx*ny - nx, nx*ny, nx*ny + nx): clusters within 1.5 D1 of
    // each other are one group, D2 = centre of mass of the heaviest group beyond D1 (>= 6 % as well).
    // Anything between the two, or beyond the second, means another shape: left alone.
    size_t i1 = far.size();
    for (size_t t = 0; t < far.size(); t++)
        if (far[t].w * 100 >= total * 6) { i1 = t; break; }
    if (i1 > 0 || i1 + 1 >= far.size()) return false;        // (a light cluster in front of D1 would be a third stride)
    const double D1 = far[0].center;
    double D2 = 0.0;
    {
        long long gw = 0;
        double gs = 0.0, first = far[1].center, last = far[1].center;
        for (size_t t = 1; t < far.size(); t++)
        {
            if (far[t].center - last > 1.5 * D1) return false;          // a second group further out
            gw += far[t].w;
            gs += far[t].center * (double) far[t].w;
            last = far[t].center;
        }
        if (gw * 100 < total * 6 || last - first > 3.0 * D1) return false;
        D2 = gs / (double) gw;
    }
    const double ratio = D2 / D1;
    const int M = (int) (ratio + 0.5);
    const double d1min = 8.0;                                   // teeth of >= 8 panels
    if (D1 < d1min * R || M < 2 || std::abs(ratio - M) > 0.02 * M || D2 * 2 > nrow) return false;
    *D1_ = D1;
    *D2_ = D2;
    *M_ = M;
    return true;
}

void lattice_coords(int panel, int R, double D1, double D2, int M, int *i_, int *j_, int *t_)
{
    const double r = (double) panel * R;
    const int j = (int) (r / D2);
    const double rem = r - j * D2;
    int i = (int) (rem / D1);
    if (i > M) i = M;
    *i_ = i;
    *j_ = j;
    *t_ = (int) ((rem - i * D1) / R);
}

bool stride_lattice_order(int nrow, const int *rowptr, const int *colidx, int R, int npanel, int chunk,
                          std::vector<int> *order)
{
    if (npanel < 64 || chunk < 1) return false;
    double D1, D2;
    int M;
    if (!detect_stride_lattice(nrow, rowptr, colidx, R, &D1, &D2, &M)) return false;

    // tooth coordinates of every panel
    struct Key { int i, j, t, p; };
    std::vector<Key> keys((size_t) npanel);
    for (int p = 0; p < npanel; p++)
    {
        int i, j, t;
        lattice_coords(p, R, D1, D2, M, &i, &j, &t);
        keys[(size_t) p] = {i, j, t, p};
    }
    // XCD blocks: consecutive teeth in (i, j) order, cut every `chunk` panels
    std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
        if (a.i != b.i) return a.i < b.i;
        if (a.j != b.j) return a.j < b.j;
        return a.p < b.p;
    });
    // lockstep sweep along t inside every block
    for (size_t s0 = 0; s0 < keys.size(); s0 += (size_t) chunk)
    {
        const size_t s1 = std::min(keys.size(), s0 + (size_t) chunk);
        std::sort(keys.begin() + (long) s0, keys.begin() + (long) s1, [](const Key &a, const Key &b) {
            if (a.t != b.t) return a.t < b.t;
            return a.p < b.p;
        });
    }
    order->resize((size_t) npanel);
    for (int q = 0; q < npanel; q++) (*order)[(size_t) q] = keys[(size_t) q].p;
    return true;
}

void locality_order(const PanelHost &p, int group, std::vector<int> *order)
{
    const int np = p.npanel;
    order->resize((size_t) np);
    if (group < 1) group = 1;
    const int ng = (np + group - 1) / group;
    if (ng <= 2)
    {
        for (int i = 0; i < np; i++) (*order)[i] = i;
        return;
    }
    // distinct B rows per group (column codes folded to a dense id space)
    int max_loc = -1, max_rem = -1;
    for (int c : p.pcol)
    {
        if (c >= 0) { if (c > max_loc) max_loc = c; }
        else if (~c > max_rem) max_rem = ~c;
    }
    const long long nb = (long long) max_loc + 1 + (long long) max_rem + 1;
    auto bid = [&](int c) -> long long { return c >= 0 ? c : (long long) max_loc + 1 + (~c); };
    std::vector<std::vector<int>> rows_of((size_t) ng);
    parallel_chunks(ng, 64, [&](long long b, long long e, int) {
        for (long long g = b; g < e; g++)
        {
            const int pa = (int) g * group, pb = std::min(np, pa + group);
            std::vector<int> &v = rows_of[(size_t) g];
            for (int q = p.pptr[pa]; q < p.pptr[pb]; q++) v.push_back((int) bid(p.pcol[(size_t) q]));
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        }
    });
    // inverted index: B row -> groups touching it
    std::vector<int> deg((size_t) nb + 1, 0);
    for (int g = 0; g < ng; g++)
        for (int c : rows_of[(size_t) g]) deg[(size_t) c + 1]++;
    for (long long c = 0; c < nb; c++) deg[(size_t) c + 1] += deg[(size_t) c];
    std::vector<int> inv((size_t) deg[(size_t) nb]), fillp(deg.begin(), deg.end() - 1);
    for (int g = 0; g < ng; g++)
        for (int c : rows_of[(size_t) g]) inv[(size_t) fillp[(size_t) c]++] = g;
    // breadth-first over groups; B rows shared by very many groups (dense columns) say nothing
    // about locality and are skipped
    const int hub = 64;
    std::vector<char> seen((size_t) ng, 0);
    std::vector<int> gorder, nbrs;
    gorder.reserve((size_t) ng);
    size_t head = 0;
    for (int start = 0; start < ng; start++)
    {
        if (seen[(size_t) start]) continue;
        seen[(size_t) start] = 1;
        gorder.push_back(start);
        while (head < gorder.size())
        {
            const int u = gorder[head++];
            nbrs.clear();
            for (int c : rows_of[(size_t) u])
            {
                const int d0 = deg[(size_t) c], d1 = deg[(size_t) c + 1];
                if (d1 - d0 > hub) continue;
                for (int t = d0; t < d1; t++)
                    if (!seen[(size_t) inv[(size_t) t]])
                    {
                        seen[(size_t) inv[(size_t) t]] = 1;
                        nbrs.push_back(inv[(size_t) t]);
                    }
            }
            std::sort(nbrs.begin(), nbrs.end());
            gorder.insert(gorder.end(), nbrs.begin(), nbrs.end());
        }
    }
    size_t w = 0;
    for (int g : gorder)
        for (int pn = g * group; pn < std::min(np, (g + 1) * group); pn++) (*order)[w++] = pn;
}

// ---- teams: four panels whose B rows one workgroup loads once (panel_format.h) --------------------

// ---- greedy clustering (teams of panels, super-teams of teams) -----------------------------------------
// Items carry sorted lists of distinct keys (CSR iptr / ikey).  Groups of up to G items are grown from the lowest
// unassigned item by repeatedly adding the unassigned item that shares most keys with the group's union (ties: the
// nearest index).  Work per group: the union's keys times the items per key.  Items are handled in independent
// ranges of `span` items (threads), a group never crosses a range.  -> group of every item, in creation order;
// slot = its position inside the group.
// ratio = true: the item that has the largest FRACTION of its own keys in the union already (ties: more shared keys) --
// an item whose keys are a subset of the group's costs the group nothing, however short its list (the 7-point dual
// rows of a KKT system next to the 27-point primal rows of the same nodes).
static void greedy_cluster(int n, const std::vector<long long> &iptr, const big_vector<uint32_t> &ikey, int G, int span,
                           std::vector<int> *group_of, std::vector<int> *slot_of, int *ngroups, bool ratio = false)
{
    group_of->assign((size_t) n, -1);
    slot_of->assign((size_t) n, 0);
    const int nrange = (n + span - 1) / span;
    std::vector<int> range_groups((size_t) nrange, 0);
    parallel_chunks(nrange, 1, [&](long long rb, long long re, int) {
        for (long long rg = rb; rg < re; rg++)
        {
            const int i0 = (int) rg * span, i1 = std::min(n, i0 + span), cnt = i1 - i0;
            // inverted index of the range: (key, item) pairs sorted by key
            big_vector<std::pair<uint32_t, int>> pairs;          // (big_vector: huge pages for the range's tens of megabytes, par.h)
            pairs.reserve((size_t) (iptr[(size_t) i1] - iptr[(size_t) i0]));
            for (int i = i0; i < i1; i++)
                for (long long q = iptr[(size_t) i]; q < iptr[(size_t) i + 1]; q++) pairs.push_back({ikey[(size_t) q], i - i0});
            std::sort(pairs.begin(), pairs.end());
            // dense local key ids
            std::vector<long long> kptr;
            big_vector<int> kitem(pairs.size());
            big_vector<int> lkey(pairs.size());                 // per pair (in item order below): local key id
            for (size_t t = 0; t < pairs.size(); t++)
            {
                if (t == 0 || pairs[t].first != pairs[t - 1].first) kptr.push_back((long long) t);
                kitem[t] = pairs[t].second;
            }
            kptr.push_back((long long) pairs.size());
            // item -> local key ids (same order as ikey)
            std::vector<long long> lptr((size_t) cnt + 1, 0);
            for (int i = 0; i < cnt; i++) lptr[(size_t) i + 1] = lptr[(size_t) i] + (iptr[(size_t) (i0 + i) + 1] - iptr[(size_t) (i0 + i)]);
            {
                std::vector<long long> fill(lptr.begin(), lptr.end() - 1);
                const int nk = (int) kptr.size() - 1;
                for (int kk = 0; kk < nk; kk++)
                    for (long long t = kptr[(size_t) kk]; t < kptr[(size_t) kk + 1]; t++) lkey[(size_t) fill[(size_t) kitem[(size_t) t]]++] = kk;
            }
            const int nk = (int) kptr.size() - 1;
            std::vector<char> assigned((size_t) cnt, 0), inkey((size_t) nk, 0);
            std::vector<int> cc((size_t) cnt, 0), touched, ukeys;
            int seed = 0, groups = 0;
            auto add = [&](int it) {
                for (long long q = lptr[(size_t) it]; q < lptr[(size_t) it + 1]; q++)
                {
                    const int kk = lkey[(size_t) q];
                    if (inkey[(size_t) kk]) continue;
                    inkey[(size_t) kk] = 1;
                    ukeys.push_back(kk);
                    for (long long t = kptr[(size_t) kk]; t < kptr[(size_t) kk + 1]; t++)
                    {
                        const int r = kitem[(size_t) t];
                        if (assigned[(size_t) r]) continue;
                        if (cc[(size_t) r]++ == 0) touched.push_back(r);
                    }
                }
            };
            for (;;)
            {
                while (seed < cnt && assigned[(size_t) seed]) seed++;
                if (seed >= cnt) break;
                const int gid = groups++;
                int members = 0;
                auto take = [&](int it) {
                    assigned[(size_t) it] = 1;
                    (*group_of)[(size_t) (i0 + it)] = gid;          // range-local id, made global below
                    (*slot_of)[(size_t) (i0 + it)] = members++;
                    add(it);
                };
                take(seed);
                while (members < G)
                {
                    int best = -1, bo = 0;
                    long long bsz = 1;
                    for (int r : touched)
                    {
                        if (assigned[(size_t) r]) continue;
                        const int o = cc[(size_t) r];
                        if (!ratio)
                        {
                            if (o > bo || (o == bo && best >= 0 && std::abs(r - seed) < std::abs(best - seed))) { best = r; bo = o; }
                            continue;
                        }
                        const long long sz = std::max<long long>(1, lptr[(size_t) r + 1] - lptr[(size_t) r]);
                        // o / sz against bo / bsz
                        const long long lhs = (long long) o * bsz, rhs = (long long) bo * sz;
                        if (best < 0 || lhs > rhs || (lhs == rhs && (o > bo || (o == bo && std::abs(r - seed) < std::abs(best - seed))))) { best = r; bo = o; bsz = sz; }
                    }
                    if (best < 0)
                    {
                        // nothing shares a key with the group (isolated rows, empty panels): the next unassigned item
                        int nx = seed;
                        while (nx < cnt && assigned[(size_t) nx]) nx++;
                        if (nx >= cnt) break;
                        best = nx;
                    }
                    take(best);
                }
                for (int r : touched) cc[(size_t) r] = 0;
                touched.clear();
                for (int kk : ukeys) inkey[(size_t) kk] = 0;
                ukeys.clear();
            }
            range_groups[(size_t) rg] = groups;
        }
    });
    std::vector<int> base((size_t) nrange + 1, 0);
    for (int rg = 0; rg < nrange; rg++) base[(size_t) rg + 1] = base[(size_t) rg] + range_groups[(size_t) rg];
    parallel_chunks(n, 1 << 16, [&](long long b, long long e, int) {
        for (long long i = b; i < e; i++) (*group_of)[(size_t) i] += base[(size_t) (i / span)];
    });
    *ngroups = base[(size_t) nrange];
}


// CSR of sorted distinct keys per item, built in parallel: raw(i, buf) appends item i's keys to buf.
template <typename F>
static void build_key_csr(int n, F raw, std::vector<long long> *iptr, big_vector<uint32_t> *ikey)
{
    constexpr int CH = 2048;
    const int nch = (n + CH - 1) / CH;
    iptr->assign((size_t) n + 1, 0);
    std::vector<big_vector<uint32_t>> cbuf((size_t) nch);
    parallel_chunks(nch, 1, [&](long long cb, long long ce, int) {
        for (long long c = cb; c < ce; c++)
        {
            big_vector<uint32_t> &buf = cbuf[(size_t) c];
            const int i0 = (int) c * CH, i1 = std::min(n, i0 + CH);
            for (int i = i0; i < i1; i++)
            {
                const size_t at = buf.size();
                raw(i, buf);
                std::sort(buf.begin() + (long) at, buf.end());
                buf.erase(std::unique(buf.begin() + (long) at, buf.end()), buf.end());
                (*iptr)[(size_t) i + 1] = (long long) (buf.size() - at);
            }
        }
    });
    for (int i = 0; i < n; i++) (*iptr)[(size_t) i + 1] += (*iptr)[(size_t) i];
    ikey->resize((size_t) (*iptr)[(size_t) n]);
    parallel_chunks(nch, 1, [&](long long cb, long long ce, int) {
        for (long long c = cb; c < ce; c++)
            if (!cbuf[(size_t) c].empty())
                memcpy(ikey->data() + (*iptr)[(size_t) c * CH], cbuf[(size_t) c].data(), sizeof(uint32_t) * cbuf[(size_t) c].size());
    });
}

void build_teams(const PanelHost &p, int nrow, const int *rowptr, const int *colidx, TeamHost *out, int T, const int *colpos, bool balanced, int mix_mode,
                 TeamSeed *seed)
{
    constexpr int TMAX = 16;
    if (T != 4 && T != 6 && T != 8 && T != 16) T = 4;
    const int TI = T / 2;                 // lattice teams: TI teeth along i times 2 along j
    out->T = T;
    const int np = p.npanel, R = p.R;
    double D1 = 0, D2 = 0;
    int M = 0;
    PhaseClock clk;
    bool lattice = (np >= 64) && detect_stride_lattice(nrow, rowptr, colidx, R, &D1, &D2, &M);
    clk.lap("build_teams: lattice detection");
    // teams of eight (team2): shape of a team in tooth coordinates, si x sj teeth x st consecutive panels along
    // the teeth
    int si = TI, sj = 2, st = 1;
    if (T >= 8)
    {
        // 2 x 2 teeth x 2 consecutive panels: 4.98 union entries per row on the pwtk stand-in, against 6.5 for
        // 4 x 2 x 1 and 5.4 for eight consecutive panels (0.351 / 0.418 / 0.424 ms with the first team2 kernel);
        // teams of sixteen: 2 x 2 x 4
        si = 2; sj = 2; st = T / 4;
    }
    out->st = st;
    out->lattice = lattice;
    // entries of every panel before its padding (counted once, by all threads: the builders ask several times per panel)
    std::vector<int> rcount((size_t) np, 0);
    parallel_chunks(np, 4096, [&](long long b, long long e, int) {
        for (long long panel = b; panel < e; panel++)
        {
            int c = 0;
            for (int q = p.pptr[(size_t) panel]; q < p.pptr[(size_t) panel + 1]; q++)
            {
                const unsigned m = (p.pmask4[(size_t) q >> 2] >> (8 * (q & 3))) & 0xFFu;
                if (m == 0) break;          // padding starts here: real entries always carry a row
                c++;
            }
            rcount[(size_t) panel] = c;
        }
    });
    auto real_count = [&](int panel) { return rcount[(size_t) panel]; };
    // Off a lattice, teams of eight are CLUSTERED: the eight panels of a team are picked for the columns they share
    // (greedy_cluster), not for being consecutive -- on a 3-D stencil in natural order eight consecutive panels are
    // a thin strip of one grid line (9.9 union entries per row on the 27-point fem3d stand-in), a cluster is a
    // compact block (6.9); nlpkkt stand-in 7.1 (its lattice teams) -> 4.4.
    bool clustered = T >= 8 && np >= 2 * T;
    std::vector<int> team_of, slot_of;
    const bool seeded = seed != nullptr && seed->valid && seed->T == T && seed->np == np;
    if (seeded)
    {
        // the teams of an earlier format of the same panels
        clustered = seed->clustered;
        lattice = out->lattice = seed->lattice;
        team_of = seed->team_of;
        slot_of = seed->slot_of;
    }
    else if (clustered)
    {
        std::vector<long long> iptr;
        big_vector<uint32_t> ikey;
        // The clustering works on ranges of consecutive items: the panels are taken in the order of their MEDIAN column,
        // so that panels far apart in the row numbering that read the same B rows (the dual rows of a KKT system and the
        // primal rows of the same nodes) fall into one range; for a mesh numbered along its own lines that is the row order.
        // Taken when the panels are of two kinds -- at least 15 % of them hold under half the mean number of entries --;
        // with panels of one size the rule changes nothing but the ties, and the row order is the better seed order
        // (shell stand-in 0.264 -> 0.275 ms with it, nlpkkt stand-in 2.39 -> 2.12).
        bool mix = false;
        {
            long long tot = 0;
            std::vector<int> rc((size_t) np);
            for (int q = 0; q < np; q++) { rc[(size_t) q] = real_count(q); tot += rc[(size_t) q]; }
            long long small = 0;
            for (int q = 0; q < np; q++) small += (2LL * rc[(size_t) q] * np < tot);
            mix = small * 100 >= 15LL * np;
            if (mix_mode >= 0) mix = mix_mode != 0;
        }
        std::vector<int> pord((size_t) np);
        for (int q = 0; q < np; q++) pord[(size_t) q] = q;
        if (mix)
        {
            std::vector<uint32_t> med((size_t) np, 0);
            parallel_chunks(np, 4096, [&](long long b, long long e, int) {
                for (long long q = b; q < e; q++)
                {
                    const int e0 = p.pptr[(size_t) q], cnt = real_count((int) q);
                    med[(size_t) q] = cnt > 0 ? col_key(p.pcol[(size_t) (e0 + cnt / 2)]) : 0xFFFFFFFFu;      // (entries are in column order)
                }
            });
            std::stable_sort(pord.begin(), pord.end(), [&](int x, int y) { return med[(size_t) x] < med[(size_t) y]; });
        }
        build_key_csr(np, [&](int i, big_vector<uint32_t> &buf) {
            const int q = pord[(size_t) i];
            const int e0 = p.pptr[q], e1 = e0 + real_count(q);
            for (int e = e0; e < e1; e++) buf.push_back(col_key(p.pcol[(size_t) e]));
        }, &iptr, &ikey);
        int ng = 0;
        {
            std::vector<int> tof, sof;
            greedy_cluster(np, iptr, ikey, T, 1 << 15, &tof, &sof, &ng, mix);
            team_of.assign((size_t) np, 0);
            slot_of.assign((size_t) np, 0);
            for (int i = 0; i < np; i++) { team_of[(size_t) pord[(size_t) i]] = tof[(size_t) i]; slot_of[(size_t) pord[(size_t) i]] = sof[(size_t) i]; }
        }
        std::vector<int> pos_of((size_t) np);                 // panel -> its item in the key CSR
        for (int i = 0; i < np; i++) pos_of[(size_t) pord[(size_t) i]] = i;
        if (lattice)
        {
            // A lattice has both: its tooth-shaped teams sweep in lockstep along the teeth and re-fetch less (pwtk
            // stand-in: 1.8 x B against 2.1 x B for clusters with 5.0 / 4.9 union entries per row), so they stay
            // unless the clusters need clearly fewer B rows (nlpkkt stand-in: 7.1 -> 4.5 entries per row).
            // Union entries of a grouping = distinct (group, column) pairs.
            auto union_total = [&](auto group_of_panel) {
                std::vector<std::pair<long long, int>> ord((size_t) np);
                for (int q = 0; q < np; q++) ord[(size_t) q] = {group_of_panel(q), q};
                std::sort(ord.begin(), ord.end());
                std::vector<size_t> gs;
                for (size_t t = 0; t < ord.size(); t++)
                    if (t == 0 || ord[t].first != ord[t - 1].first) gs.push_back(t);
                gs.push_back(ord.size());
                const int ngr = (int) gs.size() - 1;
                std::vector<long long> part((size_t) ngr, 0);
                parallel_chunks(ngr, 256, [&](long long b, long long e, int) {
                    std::vector<uint32_t> keys;
                    for (long long g = b; g < e; g++)
                    {
                        keys.clear();
                        for (size_t t = gs[(size_t) g]; t < gs[(size_t) g + 1]; t++)
                        {
                            const int q = pos_of[(size_t) ord[t].second];
                            keys.insert(keys.end(), ikey.begin() + (long) iptr[(size_t) q], ikey.begin() + (long) iptr[(size_t) q + 1]);
                        }
                        std::sort(keys.begin(), keys.end());
                        part[(size_t) g] = (long long) (std::unique(keys.begin(), keys.end()) - keys.begin());
                    }
                });
                long long tot = 0;
                for (long long v : part) tot += v;
                return tot;
            };
            const long long u_cl = union_total([&](int q) { return (long long) team_of[(size_t) q]; });
            const long long u_la = union_total([&](int q) {
                int i, j, t;
                lattice_coords(q, R, D1, D2, M, &i, &j, &t);
                return ((long long) (i / si) << 40) | ((long long) (j / sj) << 24) | (long long) (t / st);
            });
            if ((double) u_la <= 1.15 * (double) u_cl) clustered = false;
            else lattice = out->lattice = false;
        }
    }
    out->clustered = clustered;
    if (clustered) out->plocal = slot_of;
    if (seed != nullptr && !seeded)
    {
        seed->T = T;
        seed->np = np;
        seed->clustered = clustered;
        seed->lattice = lattice;
        if (clustered) { seed->team_of = team_of; seed->slot_of = slot_of; }
    }
    clk.lap(seeded ? "build_teams: panel clustering (from the seed)" : "build_teams: panel clustering (+ lattice choice)");
    // membership: (team key, slot)
    struct Mem { long long key; int slot, panel, a, b, t; };
    std::vector<Mem> mem((size_t) np);
    for (int q = 0; q < np; q++)
    {
        if (clustered) { mem[(size_t) q] = {(long long) team_of[(size_t) q], slot_of[(size_t) q], q, 0, 0, team_of[(size_t) q]}; continue; }
        if (lattice)
        {
            int i, j, t;
            lattice_coords(q, R, D1, D2, M, &i, &j, &t);
            const int a = i / si, b = j / sj, tt = t / st;
            mem[(size_t) q] = {((long long) a << 40) | ((long long) b << 24) | (long long) tt, (i % si) + si * ((j % sj) + sj * (t % st)), q, a, b, tt};
        }
        else mem[(size_t) q] = {(long long) (q / T), q % T, q, 0, 0, q / T};
    }
    std::sort(mem.begin(), mem.end(), [](const Mem &x, const Mem &y) {
        if (x.key != y.key) return x.key < y.key;
        if (x.slot != y.slot) return x.slot < y.slot;
        return x.panel < y.panel;
    });
    // teams in key order; a slot that is taken twice (irregular tooth ends) opens a new team
    struct TeamKey { int a, b, t; };
    std::vector<TeamKey> tk;
    out->tpanel.clear();
    for (size_t s0 = 0; s0 < mem.size();)
    {
        size_t s1 = s0;
        int slots[TMAX];
        for (int w = 0; w < TMAX; w++) slots[w] = -1;
        while (s1 < mem.size() && mem[s1].key == mem[s0].key && slots[mem[s1].slot] < 0)
        {
            slots[mem[s1].slot] = mem[s1].panel;
            s1++;
        }
        for (int w = 0; w < T; w++) out->tpanel.push_back(slots[w]);
        tk.push_back({mem[s0].a, mem[s0].b, mem[s0].t});
        s0 = s1;
    }
    const int nteam = (int) tk.size();
    out->nteam = nteam;
    out->lat_key.clear();
    if (lattice)
    {
        out->lat_key.resize((size_t) nteam * 3);
        for (int g = 0; g < nteam; g++) { out->lat_key[(size_t) g * 3] = tk[(size_t) g].a; out->lat_key[(size_t) g * 3 + 1] = tk[(size_t) g].b; out->lat_key[(size_t) g * 3 + 2] = tk[(size_t) g].t; }
    }

    // union entry lists: 4-way merge by (column key, occurrence inside the panel)
    std::vector<int> cnt((size_t) nteam, 0);
    // The unions of `upool` consecutive teams share three arrays (sized by the teams' panel entries, an upper bound on their union
    // entries): one vector of each kind per TEAM was 1.3 M small allocations on the nlpkkt240-size matrix, whose fresh 4 KiB pages
    // were faulted in no faster by 16 threads than by 4.
    struct UnionPool { big_vector<int> col; big_vector<uint32_t> mask; big_vector<int> src; };
    const int upool = (int) std::min<long long>(2048, std::max<long long>(64, nteam / (4LL * host_threads())));
    const int npool = (nteam + upool - 1) / upool;
    std::vector<UnionPool> pools((size_t) npool);
    std::vector<const int *> ucol((size_t) nteam, nullptr), usrc((size_t) nteam, nullptr);      // team g: cnt[g] union entries at ucol[g], T * cnt[g] at usrc[g]
    std::vector<const uint32_t *> umask((size_t) nteam, nullptr);
    // Nodes: the union of the panels' entry lists, equal (column, occurrence) keys merged.
    struct Node { int col; uint32_t mask; int src[TMAX]; int users; bool done; };
    parallel_chunks(npool, 1, [&](long long pb, long long pe, int) {
        std::vector<Node> nodes;                               // (scratch of the builder thread, not of the team)
        std::vector<int> list[TMAX];                           // node ids of every wave, in column order
        for (long long pl = pb; pl < pe; pl++)
        {
        const long long b = pl * upool, e = std::min<long long>(nteam, b + upool);
        UnionPool &pool = pools[(size_t) pl];
        {
            size_t cap = 0;
            for (long long g = b; g < e; g++)
                for (int w = 0; w < T; w++)
                {
                    const int panel = out->tpanel[(size_t) g * T + w];
                    if (panel >= 0) cap += (size_t) (p.pptr[panel + 1] - p.pptr[panel]);
                }
            pool.col.resize(cap);
            pool.mask.resize(cap);
            pool.src.resize(cap * (size_t) T);
        }
        size_t pat = 0;                                        // union entries of the pool so far
        for (long long g = b; g < e; g++)
        {
            int head[TMAX], end[TMAX], occ[TMAX];
            for (int w = 0; w < T; w++)
            {
                const int panel = out->tpanel[(size_t) g * T + w];
                head[w] = panel >= 0 ? p.pptr[panel] : 0;
                end[w] = panel >= 0 ? head[w] + real_count(panel) : 0;
                occ[w] = 0;
            }
            int *const uc = pool.col.data() + pat;
            uint32_t *const um = pool.mask.data() + pat;
            int *const us = pool.src.data() + pat * (size_t) T;    // per union entry: panel entry of wave 0 .. T - 1 (or -1)
            size_t un = 0;                                     // union entries of the team so far
            nodes.clear();
            for (int w = 0; w < T; w++) list[w].clear();
            for (;;)
            {
                bool any = false;
                uint64_t best = 0;
                for (int w = 0; w < T; w++)
                    if (head[w] < end[w])
                    {
                        const uint64_t k = ((uint64_t) col_key(p.pcol[(size_t) head[w]]) << 8) | (uint64_t) occ[w];
                        if (!any || k < best) best = k;
                        any = true;
                    }
                if (!any) break;
                Node nd;
                nd.col = 0; nd.mask = 0; nd.users = 0; nd.done = false;
                for (int w = 0; w < T; w++) nd.src[w] = -1;
                for (int w = 0; w < T; w++)
                    if (head[w] < end[w])
                    {
                        const int c = p.pcol[(size_t) head[w]];
                        const uint64_t k = ((uint64_t) col_key(c) << 8) | (uint64_t) occ[w];
                        if (k != best) continue;
                        const int q = head[w];
                        if (w < 4) nd.mask |= ((p.pmask4[(size_t) q >> 2] >> (8 * (q & 3))) & 0xFFu) << (8 * w);
                        nd.col = c;
                        nd.src[w] = q;
                        nd.users++;
                        list[w].push_back((int) nodes.size());
                        head[w]++;
                        occ[w] = (head[w] < end[w] && p.pcol[(size_t) head[w]] == c) ? occ[w] + 1 : 0;
                    }
                nodes.push_back(nd);
            }
            // Rounds of 8 union entries end at a barrier, so a round costs what its busiest wave
            // costs; a wave's columns are clustered, and in column order the four waves would work
            // one after the other.  The order inside a team is free (a row's products are summed in
            // the order its wave meets them), so the entries are dealt out in balanced passes.
            // Passes: every pass hands each wave that still has entries exactly ONE of them -- a set of open
            // nodes whose user sets are disjoint and cover the waves (a node shared by A and B plus one
            // shared by C and D; or four private nodes; ...).  All waves then meet a shared node after
            // exactly the same number of own entries, i.e. in the same ring slot of the same round.
            int cursor[TMAX];
            for (int w = 0; w < TMAX; w++) cursor[w] = 0;
            size_t left = nodes.size();
            // (Clustered teams too: their phase key has 128 values for some 440 nodes, the ties the passes would order are few --
            //  nlpkkt / fem3d / shell stand-ins at n = 128 .. 1024 within 0.1 % either way, profiles/r04_build_time.txt -- and the
            //  passes were 1.3 s of the nlpkkt240-size build.)
            if (!balanced || clustered)
            {
                // (the caller orders the union itself -- build_team2 by the phase key --: column order will do, and the
                //  passes below were a fifth of the nlpkkt240-size format's build time)
                for (size_t id = 0; id < nodes.size(); id++)
                {
                    uc[un] = nodes[id].col;
                    um[un] = nodes[id].mask;
                    for (int u = 0; u < T; u++) us[un * (size_t) T + (size_t) u] = nodes[id].src[u];
                    un++;
                }
                left = 0;
            }
            auto emit = [&](int id) {
                Node &nd = nodes[(size_t) id];
                nd.done = true;
                left--;
                uc[un] = nd.col;
                um[un] = nd.mask;
                for (int u = 0; u < T; u++) us[un * (size_t) T + (size_t) u] = nd.src[u];
                un++;
            };
            while (left > 0)
            {
                bool covered[TMAX];
                for (int w = 0; w < TMAX; w++) covered[w] = false;
                for (int w = 0; w < T; w++)
                {
                    if (covered[w]) continue;
                    while (cursor[w] < (int) list[w].size() && nodes[(size_t) list[w][(size_t) cursor[w]]].done) cursor[w]++;
                    // among the wave's next open nodes: the one with the most users, all of them uncovered
                    int pick = -1, pick_users = 0;
                    for (int t = cursor[w], seen = 0; t < (int) list[w].size() && seen < 96; t++)
                    {
                        const int id = list[w][(size_t) t];
                        const Node &nd = nodes[(size_t) id];
                        if (nd.done) continue;
                        seen++;
                        bool ok = true;
                        for (int u = 0; u < T; u++)
                            if (nd.src[u] >= 0 && covered[u]) ok = false;
                        if (ok && nd.users > pick_users) { pick = id; pick_users = nd.users; if (nd.users >= 3) break; }
                    }
                    if (pick < 0) continue;              // everything this wave has left is shared with a covered wave
                    for (int u = 0; u < T; u++)
                        if (nodes[(size_t) pick].src[u] >= 0) covered[u] = true;
                    emit(pick);
                }
                // a pass that could place nothing would loop forever: take any open node (cannot happen while
                // a wave has an open node at all, its first open node is always eligible when it comes first)
                bool any_cov = false;
                for (int w = 0; w < T; w++) any_cov = any_cov || covered[w];
                if (!any_cov)
                    for (size_t id = 0; id < nodes.size(); id++)
                        if (!nodes[id].done) { emit((int) id); break; }
            }
            cnt[(size_t) g] = (int) un;
            ucol[(size_t) g] = uc;
            umask[(size_t) g] = um;
            usrc[(size_t) g] = us;
            pat += un;
        }
        }
    });
    clk.lap("build_teams: union lists + balanced passes");
    out->tptr.assign((size_t) nteam + 1, 0);
    long long real = 0;
    for (int g = 0; g < nteam; g++)
    {
        real += cnt[(size_t) g];
        out->tptr[(size_t) g + 1] = out->tptr[(size_t) g] + (cnt[(size_t) g] + PANEL_PAD - 1) / PANEL_PAD * PANEL_PAD;
    }
    out->real_entries = real;
    const size_t total = (size_t) out->tptr[(size_t) nteam];
    // (filled by all threads: these arrays hold gigabytes on the nlpkkt240-size matrix, and the serial version of this stage
    //  was the longest single piece of its format build)
    out->tcol.resize(total);
    out->tmask.resize(total);
    out->tsrc.resize(total * (size_t) T);
    parallel_chunks(nteam, 256, [&](long long b, long long e, int) {
        for (long long g = b; g < e; g++)
        {
            size_t q = (size_t) out->tptr[(size_t) g];
            int last = 0;
            for (size_t t = 0; t < (size_t) cnt[(size_t) g]; t++, q++)
            {
                out->tcol[q] = ucol[(size_t) g][t];
                out->tmask[q] = umask[(size_t) g][t];
                last = out->tcol[q];
            }
            int *ts = &out->tsrc[(size_t) out->tptr[(size_t) g] * T];
            if (cnt[(size_t) g] > 0) memcpy(ts, usrc[(size_t) g], sizeof(int) * (size_t) cnt[(size_t) g] * (size_t) T);
            for (; q < (size_t) out->tptr[(size_t) g + 1]; q++)
            {
                out->tcol[q] = last;      // padding: valid row, no reader
                out->tmask[q] = 0u;
                for (int w = 0; w < T; w++) out->tsrc[q * (size_t) T + (size_t) w] = -1;
            }
        }
    });

    // value streams: wave w of team g reads 8 values per own entry from tvoff[4g + w] on, in the
    // order it meets its entries; tq = where every entry of the panel format went (teams of 4 / 6 only: the team2 streams
    // of build_team2 have their own)
    out->tvoff.assign((size_t) nteam * T + 1, 0);
    if (T < 8)
    {
        out->tq.assign(p.pcol.size(), -1);
        long long run = 0;
        for (int g = 0; g < nteam; g++)
            for (int w = 0; w < T; w++)
            {
                out->tvoff[(size_t) g * T + w] = run;
                const int *us = usrc[(size_t) g];
                for (size_t t = 0; t < (size_t) cnt[(size_t) g]; t++)
                {
                    const int q = us[t * (size_t) T + (size_t) w];
                    if (q >= 0) out->tq[(size_t) q] = run++;
                }
            }
        out->tvoff[(size_t) nteam * T] = run;
    }

    clk.lap("build_teams: layout (tcol, tsrc, value streams)");
    // processing order: XCD blocks of neighbouring team columns swept in lockstep along t (lattice),
    // else the natural order
    out->torder.resize((size_t) nteam);
    for (int g = 0; g < nteam; g++) out->torder[(size_t) g] = g;
    // (A recursive bisection of the team graph with generation-wide absolute rounds and a generation start barrier in the kernel
    //  was built and measured in round 3 -- profiles/r03_schedule_matrix.txt: the bytes fetched beyond L2 fall as the L2 model
    //  predicts, nlpkkt stand-in 10.3 -> 7.9 GB, but the slots that wait for their generation cost more time than the bytes
    //  save, +13 % / +33 %; without the barrier the alignment is gone within a few generations -- and removed in round 4.)
    if (seeded && seed->torder.size() == (size_t) nteam) out->torder = seed->torder;
    else if (clustered && nteam >= 128)
    {
        // Clustered teams: the workgroups resident on an XCD at one time (64: 32 CUs x 2) start together and walk
        // their unions by the same phase key, so rows shared INSIDE such a generation are requested together and
        // served by the XCD's L2 once.  Generations = super-teams of 64 teams clustered by shared columns, again
        // greedily; the kernel deals the order to the XCDs in eight contiguous runs.
        std::vector<long long> iptr;
        big_vector<uint32_t> ikey;
        build_key_csr(nteam, [&](int g, big_vector<uint32_t> &buf) {
            for (int t = 0; t < cnt[(size_t) g]; t++) buf.push_back(col_key(ucol[(size_t) g][t]));
        }, &iptr, &ikey);
        std::vector<int> super_of, sslot;
        int ns = 0;
        greedy_cluster(nteam, iptr, ikey, T == 16 ? 32 : 64, 1 << 13, &super_of, &sslot, &ns);   // the workgroups resident on an XCD
        // order of the super-teams: the slab order of locality.cpp on their graph (two super-teams are adjacent when
        // they share a B row; weight = union entries) -- eight slabs, one per XCD, each swept along its long axis,
        // so that an XCD's L2 sees one compact region and consecutive generations are neighbours
        std::vector<int> srank((size_t) ns);
        {
            // (key, super-team) pairs are sorted per range of teams, in parallel -- the ranges greedy_cluster() worked on, so a
            // super-team lies inside one; edges between super-teams of different ranges are left out except for a link
            // between the last of a range and the first of the next, which keeps the slabs in range order
            const int span = 1 << 13;
            const int nrange = (nteam + span - 1) / span;
            std::vector<int> weight((size_t) ns, 0);
            std::vector<std::vector<std::pair<int, int>>> redges((size_t) nrange);
            parallel_chunks(nrange, 1, [&](long long rb, long long re, int) {
                for (long long rg = rb; rg < re; rg++)
                {
                    const int g0 = (int) rg * span, g1 = std::min(nteam, g0 + span);
                    std::vector<std::pair<uint32_t, int>> ks;
                    ks.reserve((size_t) (iptr[(size_t) g1] - iptr[(size_t) g0]));
                    for (int g = g0; g < g1; g++)
                        for (long long q = iptr[(size_t) g]; q < iptr[(size_t) g + 1]; q++) ks.push_back({ikey[(size_t) q], super_of[(size_t) g]});
                    std::sort(ks.begin(), ks.end());
                    ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
                    std::vector<std::pair<int, int>> &edges = redges[(size_t) rg];
                    for (size_t a = 0; a < ks.size();)
                    {
                        size_t b = a;
                        while (b < ks.size() && ks[b].first == ks[a].first) b++;
                        for (size_t x = a; x < b; x++)
                        {
                            weight[(size_t) ks[x].second]++;            // (a super-team belongs to one range: no race)
                            for (size_t y = a; y < b; y++)
                                if (x != y) edges.push_back({ks[x].second, ks[y].second});
                        }
                        a = b;
                        if (edges.size() > (size_t) 1 << 22) { std::sort(edges.begin(), edges.end()); edges.erase(std::unique(edges.begin(), edges.end()), edges.end()); }
                    }
                    std::sort(edges.begin(), edges.end());
                    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
                }
            });
            std::vector<std::pair<int, int>> edges;
            for (int rg = 0; rg < nrange; rg++)
            {
                edges.insert(edges.end(), redges[(size_t) rg].begin(), redges[(size_t) rg].end());
                if (rg + 1 < nrange)
                {
                    const int last = super_of[(size_t) std::min(nteam, (rg + 1) * span) - 1], first = super_of[(size_t) (rg + 1) * span];
                    if (last != first) { edges.push_back({last, first}); edges.push_back({first, last}); }
                }
                redges[(size_t) rg].clear();
                redges[(size_t) rg].shrink_to_fit();
            }
            std::sort(edges.begin(), edges.end());
            edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
            std::vector<int> gp((size_t) ns + 1, 0), ga(edges.size());
            for (size_t e = 0; e < edges.size(); e++) { gp[(size_t) edges[e].first + 1]++; ga[e] = edges[e].second; }
            for (int q = 0; q < ns; q++) gp[(size_t) q + 1] += gp[(size_t) q];
            std::vector<int> so;
            if (ns < 16 || !graph_slab_order(ns, gp, ga, weight, 8, &so))
            {
                so.resize((size_t) ns);
                for (int q = 0; q < ns; q++) so[(size_t) q] = q;
            }
            for (int q = 0; q < ns; q++) srank[(size_t) so[(size_t) q]] = q;
        }
        std::sort(out->torder.begin(), out->torder.end(), [&](int x, int y) {
            if (super_of[(size_t) x] != super_of[(size_t) y]) return srank[(size_t) super_of[(size_t) x]] < srank[(size_t) super_of[(size_t) y]];
            return sslot[(size_t) x] < sslot[(size_t) y];
        });
    }
    if (lattice && !(seeded && seed->torder.size() == (size_t) nteam))
    {
        const int chunk = (nteam + 7) / 8;
        std::sort(out->torder.begin(), out->torder.end(), [&](int x, int y) {
            if (tk[(size_t) x].a != tk[(size_t) y].a) return tk[(size_t) x].a < tk[(size_t) y].a;
            if (tk[(size_t) x].b != tk[(size_t) y].b) return tk[(size_t) x].b < tk[(size_t) y].b;
            return x < y;
        });
        for (size_t s0 = 0; s0 < out->torder.size(); s0 += (size_t) chunk)
        {
            const size_t s1 = std::min(out->torder.size(), s0 + (size_t) chunk);
            std::sort(out->torder.begin() + (long) s0, out->torder.begin() + (long) s1, [&](int x, int y) {
                if (tk[(size_t) x].t != tk[(size_t) y].t) return tk[(size_t) x].t < tk[(size_t) y].t;
                return x < y;
            });
        }
    }
    if (seed != nullptr && !seeded) { seed->torder = out->torder; seed->valid = true; }
    clk.lap(seeded ? "build_teams: processing order (from the seed)" : "build_teams: processing order (super-teams)");
    parallel_chunks(npool, 1, [&](long long b, long long e, int) {
        for (long long pl = b; pl < e; pl++)
        {
            big_vector<int>().swap(pools[(size_t) pl].col);
            big_vector<uint32_t>().swap(pools[(size_t) pl].mask);
            big_vector<int>().swap(pools[(size_t) pl].src);
        }
    });
}

// ---- team2 streams (panel_format.h) ------------------------------------------------------------------
void build_team2(const PanelHost &p, int nrow, const int *rowptr, const int *colidx, Team2Host *out, const int *colpos, TeamSeed *seed)
{
    constexpr int D = TEAM2_D, CAP = TEAM2_CAP, T = TEAM2_T, W = TEAM2_T;   // panels of a team = waves = slots of a round
    const bool compact = out->compact;
    constexpr int sbits = 3, fbase = 16;                                    // slot bits and first flag bit of record word 0
    const size_t blkw = (size_t) 32 * W;                                    // words of a record block (8 rounds x W waves x 4)
    PhaseClock clk;
    released_async<TeamHost> th_owner;                                      // (freed by a background thread)
    TeamHost &th = *th_owner;
    // The balanced passes of build_teams break the ties of the phase key (a lattice team has twenty nodes per key value): in
    // plain column order the nodes of one wave come in runs, the rounds then hold four parts of one wave and none of another,
    // and a round lasts as long as its busiest wave -- pwtk stand-in 0.304 -> 0.315 ms at n = 256, 0.199 -> 0.210 at n = 128.
    build_teams(p, nrow, rowptr, colidx, &th, T, colpos, true, -1, seed);
    const bool with_vals = !p.pval.empty() || p.pcol.empty();             // (structure-only panels: the caller scatters the values through vmap)
    clk.lap("build_team2: build_teams total");
    // Phase key of a union entry: (position of its B row in the processing order) mod S, S = rows a team advances
    // along its sweep (8 x the consecutive panels of a lattice team, 64 for eight consecutive panels).  Teams are
    // dealt to the workgroups of an XCD in order and start a fraction of a microsecond apart; a B row shared by
    // neighbouring teams sits S positions further in the next one.  Walking every team's union by this key makes
    // all its readers ask for it at the same point of their lives, i.e. within the few microseconds a line
    // survives in the XCD's L2 -- instead of at unrelated moments of 35-microsecond lives.
    const int S = th.lattice ? 8 * th.st : 8 * T;
    const int nteam = th.nteam;
    out->nteam = nteam;
    out->lattice = th.lattice;
    out->tpanel = th.tpanel;
    out->torder = th.torder;
    auto mask_of = [&](size_t q) { return (p.pmask4[q >> 2] >> (8 * (q & 3))) & 0xFFu; };

    struct Part { int src; unsigned char slot, first, len; };               // (8 bytes: the parts of the nlpkkt240-size format are 24 M rounds x 32)
    struct TeamOut
    {
        int nr = 0, filled = 0, nparts = 0;
        int anycol = 0;                             // a column of the team (a valid row for the prologue's empty slots)
        size_t r0 = 0;                              // rounds of the pool in front of this team's
        int *col = nullptr;                         // nr * W slot columns (TEAM2_NOCOL = empty slot)
        // parts of wave w in round r: ownp[(r * W + w) * CAP .. + ownc[r * W + w])  (flat: one small vector per
        // (round, wave) was 126 M heap allocations on the nlpkkt240-size matrix)
        Part *ownp = nullptr;
        unsigned char *ownc = nullptr;
    };
    // (The rounds of `rpool` consecutive teams share three arrays, like the unions of build_teams: three vectors per team were 1.3 M
    //  allocations whose pages no number of threads faulted in faster; col / ownp / ownc of a team point into its pool.)
    struct RoundPool { big_vector<int> col; big_vector<Part> ownp; big_vector<unsigned char> ownc; };
    const int rpool = (int) std::min<long long>(2048, std::max<long long>(32, nteam / (4LL * host_threads())));
    const int nrpool = (nteam + rpool - 1) / rpool;
    std::vector<RoundPool> rpools((size_t) nrpool);
    std::vector<TeamOut> res((size_t) nteam);
    // real union entries of team g
    auto team_nodes = [&](int g, std::vector<int> &nodes) {
        nodes.clear();
        for (int q = th.tptr[(size_t) g]; q < th.tptr[(size_t) g + 1]; q++)
        {
            bool used = false;
            for (int w = 0; w < T; w++) used = used || th.tsrc[(size_t) q * T + (size_t) w] >= 0;
            if (used) nodes.push_back(q);
        }
    };
    auto key = [&](int q) {
        const int c = th.tcol[(size_t) q];
        const long long ps = c >= 0 ? (colpos ? colpos[c] : c) : (long long) (~c);
        // clustered teams (square part): where the row of A with this number sits inside ITS team
        if (th.clustered && c >= 0 && ps / 8 < (long long) th.plocal.size()) return (int) (ps % 8) * 16 + th.plocal[(size_t) (ps / 8)];      // (row of the panel, slot): neighbours in the order belong to different waves
        return (int) (ps % S);
    };
    // contiguous row ranges of every (node, wave)
    auto ranges = [&](unsigned m, Part *dst) {
        int n = 0;
        for (int r = 0; r < 8;)
        {
            if (!((m >> r) & 1u)) { r++; continue; }
            int l = 1;
            while (r + l < 8 && ((m >> (r + l)) & 1u)) l++;
            dst[n].first = (unsigned char) r; dst[n].len = (unsigned char) l; n++;
            r += l;
        }
        return n;
    };
    // List scheduler of one team: `nodes` in the order they are to be met; <= W slots per round, <= CAP parts per wave and
    // round, look-ahead 4 W nodes.  Empty slots are marked TEAM2_NOCOL here and written to the records as a row of the team
    // (a fetch nobody reads).  (Measured and removed: a scheduler that picks by the busiest wave's load, a cap on a wave's
    // load per round -- round 3)
    const bool swap_on = nteam <= 120000;                                   // (the balance pass costs 2 s per 100 k teams on 16 CPUs)
    struct SchedScratch { std::vector<unsigned char> rk, rr; std::vector<char> taken; };    // (one per builder thread, not one per team)
    auto schedule_team = [&](int g, const std::vector<int> &nodes, SchedScratch &scr, RoundPool &pool) {
        TeamOut &to = res[(size_t) g];
        to.r0 = pool.ownc.size() / (size_t) W;
        to.anycol = nodes.empty() ? 0 : th.tcol[(size_t) nodes[0]];
        // the row ranges of every (node, wave), once: byte = first << 4 | len, up to 4 per wave (the look-ahead visits a
        // node several times before it fits)
        const size_t nn = nodes.size();
        std::vector<unsigned char> &rk = scr.rk, &rr = scr.rr;
        rk.assign(nn * (size_t) T, 0);
        rr.resize(nn * (size_t) T * 4);                                    // (read only where rk says an entry exists)
        for (size_t t = 0; t < nn; t++)
            for (int w = 0; w < T; w++)
            {
                const int src = th.tsrc[(size_t) nodes[t] * T + (size_t) w];
                if (src < 0) continue;
                Part tmp[4];
                const int kk = ranges(mask_of((size_t) src), tmp);
                rk[t * (size_t) T + (size_t) w] = (unsigned char) kk;
                for (int i = 0; i < kk; i++) rr[(t * (size_t) T + (size_t) w) * 4 + (size_t) i] = (unsigned char) (tmp[i].first << 4 | tmp[i].len);
            }
        std::vector<char> &taken = scr.taken;
        taken.assign(nn, 0);
        size_t head = 0, left = nn;
        while (left > 0)
        {
            int cnt[W];
            for (int w = 0; w < W; w++) cnt[w] = 0;
            int nslot = 0;
            const size_t base_col = (to.r0 + (size_t) to.nr) * (size_t) W;
            pool.col.resize(base_col + (size_t) W, TEAM2_NOCOL);
            pool.ownp.resize((base_col + (size_t) W) * CAP);
            pool.ownc.resize(base_col + (size_t) W, 0);
            while (head < nn && taken[head]) head++;
            int seen = 0;
            for (size_t t = head; t < nn && nslot < W && seen < 4 * W; t++)
            {
                if (taken[t]) continue;
                seen++;
                const unsigned char *kk = &rk[t * (size_t) T];
                bool fits = true;
                for (int w = 0; w < W; w++)
                    if (cnt[w] + kk[w] > CAP) fits = false;
                if (!fits) continue;
                const int q = nodes[t];
                for (int w = 0; w < W; w++)
                    for (int i = 0; i < kk[w]; i++)
                    {
                        const unsigned char b = rr[(t * (size_t) T + (size_t) w) * 4 + (size_t) i];
                        Part pt;
                        pt.first = (unsigned char) (b >> 4);
                        pt.len = (unsigned char) (b & 15);
                        pt.slot = (unsigned char) nslot;
                        pt.src = th.tsrc[(size_t) q * T + (size_t) w];
                        pool.ownp[(base_col + (size_t) w) * CAP + (size_t) cnt[w]] = pt;
                        pool.ownc[base_col + (size_t) w]++;
                        cnt[w]++;
                        to.nparts++;
                    }
                pool.col[base_col + (size_t) nslot] = th.tcol[(size_t) q];
                nslot++;
                taken[t] = 1;
                left--;
            }
            if (nslot == 0)
            {
                // (cannot happen: the first open node of an empty round always fits -- a panel has at most 4 row ranges)
                fprintf(stderr, "[FATAL] team2 scheduler: a round placed nothing\n");
                abort();
            }
            to.filled += nslot;
            to.nr++;
        }
    };
    // Balance pass over a team's finished rounds: a round lasts as long as its busiest wave (one barrier per round), so for
    // every pair of consecutive rounds the exchange of one slot of each that lowers (busiest wave of r) + (busiest wave of
    // r + 1) most is made -- rounds stay full (an empty slot is a fetch), a node moves by one round at most per pass (the phase
    // order it was placed by has that much slack).  Work of a part = 3 + its rows.
    auto balance_rounds = [&](TeamOut &to) {
        const int nr = to.nr;
        if (nr < 2) return;
        // work[(r * W + slot) * W + w], count likewise: what slot `slot` of round r gives wave w
        std::vector<unsigned char> swork((size_t) nr * W * W, 0), scnt((size_t) nr * W * W, 0);
        std::vector<int> load((size_t) nr * W, 0), cnt((size_t) nr * W, 0);
        for (int r = 0; r < nr; r++)
            for (int w = 0; w < W; w++)
            {
                const Part *ow = &to.ownp[((size_t) r * W + (size_t) w) * CAP];
                const int c = to.ownc[(size_t) r * W + (size_t) w];
                cnt[(size_t) r * W + (size_t) w] = c;
                for (int i = 0; i < c; i++)
                {
                    swork[((size_t) r * W + (size_t) ow[i].slot) * W + (size_t) w] += (unsigned char) (3 + ow[i].len);
                    scnt[((size_t) r * W + (size_t) ow[i].slot) * W + (size_t) w]++;
                    load[(size_t) r * W + (size_t) w] += 3 + ow[i].len;
                }
            }
        auto maxload = [&](int r) { int m = 0; for (int w = 0; w < W; w++) m = std::max(m, load[(size_t) r * W + (size_t) w]); return m; };
        for (int pass = 0; pass < 2; pass++)
            for (int r = 0; r + 1 < nr; r++)
            {
                const int cur = maxload(r) + maxload(r + 1);
                int best = cur, bi = -1, bj = -1;
                for (int i = 0; i < W; i++)
                {
                    if (to.col[(size_t) r * W + (size_t) i] == TEAM2_NOCOL) continue;
                    const unsigned char *wa = &swork[((size_t) r * W + (size_t) i) * W], *ca = &scnt[((size_t) r * W + (size_t) i) * W];
                    for (int j = 0; j < W; j++)
                    {
                        if (to.col[(size_t) (r + 1) * W + (size_t) j] == TEAM2_NOCOL) continue;
                        const unsigned char *wb = &swork[((size_t) (r + 1) * W + (size_t) j) * W], *cb = &scnt[((size_t) (r + 1) * W + (size_t) j) * W];
                        int m0 = 0, m1 = 0;
                        bool ok = true;
                        for (int w = 0; w < W; w++)
                        {
                            if (cnt[(size_t) r * W + (size_t) w] - ca[w] + cb[w] > CAP || cnt[(size_t) (r + 1) * W + (size_t) w] - cb[w] + ca[w] > CAP) { ok = false; break; }
                            m0 = std::max(m0, load[(size_t) r * W + (size_t) w] - wa[w] + wb[w]);
                            m1 = std::max(m1, load[(size_t) (r + 1) * W + (size_t) w] - wb[w] + wa[w]);
                        }
                        if (ok && m0 + m1 < best) { best = m0 + m1; bi = i; bj = j; }
                    }
                }
                if (bi < 0) continue;
                // exchange slot bi of round r with slot bj of round r + 1
                std::swap(to.col[(size_t) r * W + (size_t) bi], to.col[(size_t) (r + 1) * W + (size_t) bj]);
                for (int w = 0; w < W; w++)
                {
                    Part *p0 = &to.ownp[((size_t) r * W + (size_t) w) * CAP], *p1 = &to.ownp[((size_t) (r + 1) * W + (size_t) w) * CAP];
                    Part keep0[4], keep1[4], mv0[4], mv1[4];
                    int k0 = 0, k1 = 0, n0 = 0, n1 = 0;
                    for (int i = 0; i < (int) to.ownc[(size_t) r * W + (size_t) w]; i++) { if (p0[i].slot == bi) mv0[n0++] = p0[i]; else keep0[k0++] = p0[i]; }
                    for (int i = 0; i < (int) to.ownc[(size_t) (r + 1) * W + (size_t) w]; i++) { if (p1[i].slot == bj) mv1[n1++] = p1[i]; else keep1[k1++] = p1[i]; }
                    for (int i = 0; i < n1; i++) { mv1[i].slot = (unsigned char) bi; keep0[k0++] = mv1[i]; }
                    for (int i = 0; i < n0; i++) { mv0[i].slot = (unsigned char) bj; keep1[k1++] = mv0[i]; }
                    for (int i = 0; i < k0; i++) p0[i] = keep0[i];
                    for (int i = 0; i < k1; i++) p1[i] = keep1[i];
                    to.ownc[(size_t) r * W + (size_t) w] = (unsigned char) k0;
                    to.ownc[(size_t) (r + 1) * W + (size_t) w] = (unsigned char) k1;
                    const int wa = swork[((size_t) r * W + (size_t) bi) * W + (size_t) w], wb = swork[((size_t) (r + 1) * W + (size_t) bj) * W + (size_t) w];
                    const int ca = scnt[((size_t) r * W + (size_t) bi) * W + (size_t) w], cb = scnt[((size_t) (r + 1) * W + (size_t) bj) * W + (size_t) w];
                    load[(size_t) r * W + (size_t) w] += wb - wa;
                    load[(size_t) (r + 1) * W + (size_t) w] += wa - wb;
                    cnt[(size_t) r * W + (size_t) w] += cb - ca;
                    cnt[(size_t) (r + 1) * W + (size_t) w] += ca - cb;
                    std::swap(swork[((size_t) r * W + (size_t) bi) * W + (size_t) w], swork[((size_t) (r + 1) * W + (size_t) bj) * W + (size_t) w]);
                    std::swap(scnt[((size_t) r * W + (size_t) bi) * W + (size_t) w], scnt[((size_t) (r + 1) * W + (size_t) bj) * W + (size_t) w]);
                }
            }
    };
    // ---- the launch grid: the order cut into 8 contiguous pieces of equal work (union entries / W + a fixed cost per
    // team), one per XCD -- pieces of equal team COUNT leave XCDs idle when the teams differ (KKT systems: 27-point primal
    // rows, short dual rows).
    constexpr int WGS = 64;                                             // workgroups resident on an XCD = a generation
    std::vector<int> cut(9, nteam);
    std::vector<int> nn((size_t) nteam, 0);
    parallel_chunks(nteam, 256, [&](long long b, long long e, int) {
        std::vector<int> nodes;
        for (long long g = b; g < e; g++) { team_nodes((int) g, nodes); nn[(size_t) g] = (int) nodes.size(); }
    });
    auto compute_cut = [&]() {
        for (int q = 0; q <= 8; q++) cut[(size_t) q] = nteam;
        long long total = 0;
        for (int g = 0; g < nteam; g++) total += (nn[(size_t) g] + W - 1) / W + 4;
        cut[0] = 0;
        long long acc = 0;
        int x = 1;
        for (int i = 0; i < nteam && x < 8; i++)
        {
            acc += (nn[(size_t) out->torder[(size_t) i]] + W - 1) / W + 4;
            while (x < 8 && acc * 8 >= total * x) cut[(size_t) x++] = i + 1;
        }
    };
    compute_cut();
    parallel_chunks(nrpool, 1, [&](long long pb, long long pe, int) {
        std::vector<int> nodes;
        std::vector<std::pair<int, int>> keyed;                 // (key, node): the key is looked up once per node, not per comparison
        SchedScratch scr;
        for (long long pl = pb; pl < pe; pl++)
        {
            RoundPool &pool = rpools[(size_t) pl];
            const long long b = pl * rpool, e = std::min<long long>(nteam, b + rpool);
            {
                size_t est = 0;                                 // rounds: an eighth of the nodes, a quarter more for rounds left partly empty
                for (long long g = b; g < e; g++) est += (size_t) nn[(size_t) g] / (size_t) W + (size_t) nn[(size_t) g] / (size_t) (4 * W) + 2;
                pool.col.reserve(est * (size_t) W);
                pool.ownp.reserve(est * (size_t) W * CAP);
                pool.ownc.reserve(est * (size_t) W);
            }
            auto bind = [&](TeamOut &to) {
                to.col = pool.col.data() + to.r0 * (size_t) W;
                to.ownp = pool.ownp.data() + to.r0 * (size_t) W * CAP;
                to.ownc = pool.ownc.data() + to.r0 * (size_t) W;
            };
            for (long long g = b; g < e; g++)
            {
                team_nodes((int) g, nodes);
                keyed.resize(nodes.size());
                for (size_t i = 0; i < nodes.size(); i++) keyed[i] = {key(nodes[i]), nodes[i]};
                std::sort(keyed.begin(), keyed.end());          // (nodes come in ascending order: ties keep it, as a stable sort by key would)
                for (size_t i = 0; i < nodes.size(); i++) nodes[i] = keyed[i].second;
                schedule_team((int) g, nodes, scr, pool);
                if (swap_on) { bind(res[(size_t) g]); balance_rounds(res[(size_t) g]); }
            }
            for (long long g = b; g < e; g++) bind(res[(size_t) g]);          // (the pool's arrays have stopped growing)
        }
    });
    clk.lap("build_team2: rounds (phase sort, list scheduler)");
    // ---- lattice teams: the processing order by search over block orders against an L2 model (team_order.h).  CRPSPMM_T2_LATORDER=0
    // keeps the round-2 order (strips of team columns swept along the teeth).
    if (th.lattice && th.lat_key.size() == (size_t) nteam * 3 && knobs().t2_latorder)
    {
        std::vector<const int *> cols((size_t) nteam);
        std::vector<int> nrs((size_t) nteam);
        for (int g = 0; g < nteam; g++) { cols[(size_t) g] = res[(size_t) g].col; nrs[(size_t) g] = res[(size_t) g].nr; }
        LatticeOrderInfo li;
        // an XCD's 4 MiB of L2 in row slices of the widest tile (2 KiB); a generation = the workgroups resident on an XCD
        const bool changed = lattice_block_order(nteam, th.lat_key.data(), W, WGS, 2048, TEAM2_NOCOL, cols.data(), nrs.data(), &out->torder, &li);
        if (clk.on)
            fprintf(stderr, "[crpspmm timing] lattice order: %d candidates, model misses %.0f (given) -> %.0f (boxes %d x %d, blocks %d x %d x %d, flags %d)%s\n",
                    li.candidates, li.miss_given, li.miss_best, li.pa, li.pb, li.bt, li.ba, li.bb, li.flags, changed ? "" : " -- kept the given order");
        if (changed) compute_cut();
        clk.lap("build_team2: lattice order search");
    }
    const int nunit = nteam;
    std::vector<TeamOut> &ures = res;
    // ---- layout: record blocks, value streams
    out->tinfo.assign((size_t) nunit * 4, 0);
    out->tpro.assign((size_t) nunit * D * W * 2, 0);
    out->tvoff.assign((size_t) nunit * W + 1, 0);
    std::vector<int> blk0((size_t) nunit + 1, 0);
    long long run = 0;
    out->real_entries = out->slots = out->parts = 0;
    // value units (TEAM2_VUNIT values) of a round of a wave: its parts' rows, padded
    auto round_units = [&](const TeamOut &to, int r, int w) {
        int nv = 0;
        const Part *ow = &to.ownp[((size_t) r * W + (size_t) w) * CAP];
        for (int i = 0; i < (int) to.ownc[(size_t) r * W + (size_t) w]; i++) nv += compact ? ow[i].len : 8;
        return (nv + TEAM2_VUNIT - 1) / TEAM2_VUNIT;
    };
    {
        std::vector<long long> wunits((size_t) nunit * W, 0);
        parallel_chunks(nunit, 256, [&](long long b, long long e, int) {
            for (long long g = b; g < e; g++)
            {
                const TeamOut &to = ures[(size_t) g];
                for (int w = 0; w < W; w++)
                {
                    long long u = 0;
                    for (int r = 0; r < to.nr; r++) u += round_units(to, r, w);
                    wunits[(size_t) g * W + (size_t) w] = u;
                }
            }
        });
        for (int g = 0; g < nunit; g++)
        {
            const TeamOut &to = ures[(size_t) g];
            blk0[(size_t) g + 1] = blk0[(size_t) g] + (to.nr + 7) / 8;
            out->tinfo[(size_t) g * 4] = to.nr;
            out->tinfo[(size_t) g * 4 + 1] = blk0[(size_t) g];
            out->tinfo[(size_t) g * 4 + 2] = to.nparts;
            out->tinfo[(size_t) g * 4 + 3] = to.filled;
            out->real_entries += to.filled;
            out->slots += (long long) to.nr * W;
            out->parts += to.nparts;
            for (int w = 0; w < W; w++)
            {
                out->tvoff[(size_t) g * W + (size_t) w] = run;
                run += wunits[(size_t) g * W + (size_t) w];
            }
        }
    }
    out->tvoff[(size_t) nunit * W] = run;
    out->nvalues = run * TEAM2_VUNIT;
    // launch grid: run x of tgrid = what XCD x processes, in order (the cuts computed above)
    {
        int cpx = 1;
        for (int q = 0; q < 8; q++) cpx = std::max(cpx, cut[(size_t) q + 1] - cut[(size_t) q]);
        out->tgrid.assign((size_t) cpx * 8, -1);
        for (int q = 0; q < 8; q++)
            for (int i = cut[(size_t) q]; i < cut[(size_t) q + 1]; i++) out->tgrid[(size_t) q * cpx + (size_t) (i - cut[(size_t) q])] = out->torder[(size_t) i];
    }
    parallel_fill(out->trec, (size_t) blk0[(size_t) nunit] * blkw + blkw, 0u);
    if (with_vals) parallel_fill(out->tval, (size_t) run * TEAM2_VUNIT, 0.0);
    else big_vector<double>().swap(out->tval);
    // vmap through the panel format's slot map: pmap[nz] = q * 8 + row of the panel format
    big_vector<uint32_t> slot_of;                                          // panel-format value slot -> tval slot
    slot_of.resize(p.pcol.size() * 8);          // (only the (entry, row) pairs that exist are written below and read through pmap)
    parallel_chunks(nunit, 32, [&](long long b, long long e, int) {
        for (long long g = b; g < e; g++)
        {
            const TeamOut &to = ures[(size_t) g];
            for (int w = 0; w < W; w++)
            {
                // value units of every round of this wave (prefix), then the records
                std::vector<long long> voff((size_t) to.nr + 1, 0);
                std::vector<int> nvals((size_t) to.nr + 1, 0);
                for (int r = 0; r < to.nr; r++)
                {
                    const Part *ow = &to.ownp[((size_t) r * W + (size_t) w) * CAP];
                    int nv = 0;
                    for (int i = 0; i < (int) to.ownc[(size_t) r * W + (size_t) w]; i++) nv += compact ? ow[i].len : 8;
                    nvals[(size_t) r] = nv;
                    voff[(size_t) r + 1] = voff[(size_t) r] + (nv + TEAM2_VUNIT - 1) / TEAM2_VUNIT;
                }
                if (voff[(size_t) to.nr] >= (1LL << 20)) { fprintf(stderr, "[FATAL] team2 format: a wave's value stream exceeds 2^20 units\n"); abort(); }
                const long long e0 = out->tvoff[(size_t) g * W + (size_t) w] * TEAM2_VUNIT;     // first value of the wave's stream
                for (int r = 0; r < to.nr; r++)
                {
                    const Part *ow = &to.ownp[((size_t) r * W + (size_t) w) * CAP];
                    const size_t nown = to.ownc[(size_t) r * W + (size_t) w];
                    uint32_t x = (uint32_t) nown, y = 0, z = 0;
                    long long e = e0 + voff[(size_t) r] * TEAM2_VUNIT;       // where the round's block starts
                    int prefix = 0;
                    for (size_t i = 0; i < nown; i++)
                    {
                        const Part &pt = ow[i];
                        x |= (uint32_t) pt.slot << (4 + sbits * (int) i);
                        y |= (uint32_t) (pt.first * 8 + pt.len - 1) << (6 * i);
                        // value position of the part: prefix + 7 - first (tools/gen_team2_asm.py); full groups: the part's 8
                        // values start at 8 i, row r at 8 i + r, i.e. "prefix" = 8 i + first
                        if (!compact) prefix = 8 * (int) i + pt.first;
                        const uint32_t pos = (uint32_t) (prefix + 7 - pt.first);
                        if (i == 0) x |= pos << (fbase + 5);
                        else if (i == 1) y |= pos << 24;
                        else if (i == 2) z |= pos << 20;
                        else z |= pos << 26;
                        for (int rr = pt.first; rr < pt.first + pt.len; rr++)
                        {
                            const size_t at = (size_t) (e + prefix + (rr - pt.first));
                            if (with_vals) out->tval[at] = p.pval[(size_t) pt.src * 8 + (size_t) rr];
                            slot_of[(size_t) pt.src * 8 + (size_t) rr] = (uint32_t) at;
                        }
                        prefix += pt.len;
                    }
                    uint32_t *rec = &out->trec[((size_t) blk0[(size_t) g] + (size_t) (r >> 3)) * blkw + (size_t) (r & 7) * 4 * W + (size_t) w * 4];
                    rec[0] = x;
                    rec[1] = y;
                    rec[2] = z;
                }
                // what is fetched D rounds ahead: value block (offset, size class), column
                for (int r = 0; r < to.nr; r++)
                {
                    const int rd = r + D;
                    uint32_t *rec = &out->trec[((size_t) blk0[(size_t) g] + (size_t) (r >> 3)) * blkw + (size_t) (r & 7) * 4 * W + (size_t) w * 4];
                    rec[2] |= (uint32_t) (rd < to.nr ? voff[(size_t) rd] : voff[(size_t) to.nr]);
                    if (rd < to.nr && nvals[(size_t) rd] > 0) rec[1] |= (uint32_t) ((nvals[(size_t) rd] + 7) / 8 - 1) << 30;
                    // (an empty slot fetches a row of the team that nobody reads: testing for it in the kernel's issue block, behind
                    //  the barrier and on the CU's one scalar unit, cost more than the few fetches of the default schedules)
                    rec[3] = (uint32_t) ((rd < to.nr && to.col[(size_t) rd * W + (size_t) w] != TEAM2_NOCOL) ? to.col[(size_t) rd * W + (size_t) w] : to.anycol);
                    // flags that steer the kernel's round (tools/gen_team2_asm.py)
                    if (rd < to.nr) rec[0] |= 1u << fbase;                                   // ISSUE: fetch for round r + D
                    if (r + D - 1 >= to.nr) rec[0] |= 1u << (fbase + 1);                           // TAIL: fewer than D-1 younger rounds in flight
                    if (r == to.nr - 1) rec[0] |= 1u << (fbase + 2);                               // LAST
                    if (w == 0 && (r & 7) == 0 && (r >> 3) + 1 < (to.nr + 7) / 8) rec[0] |= 1u << (fbase + 3);   // RECS: fetch the next record block
                }
                for (int d = 0; d < D; d++)
                {
                    int *pr = &out->tpro[(((size_t) g * D + (size_t) d) * W + (size_t) w) * 2];
                    // (the prologue's fetches are compiled code with a fixed DMA count: an empty slot fetches a valid row)
                    pr[0] = (d < to.nr && to.col[(size_t) d * W + (size_t) w] != TEAM2_NOCOL) ? to.col[(size_t) d * W + (size_t) w] : to.anycol;
                    pr[1] = (int) ((d < to.nr) ? voff[(size_t) d] : voff[(size_t) to.nr]);
                }
            }
        }
    });
    clk.lap("build_team2: records, value streams");
    out->vmap.resize(p.pmap.size());
    parallel_chunks((long long) p.pmap.size(), 1 << 18, [&](long long b, long long e, int) {
        for (long long nz = b; nz < e; nz++) out->vmap[(size_t) nz] = slot_of[(size_t) p.pmap[(size_t) nz]];
    });
    clk.lap("build_team2: value-update map");
    // (hundreds of thousands of small vectors: released by all threads, not by the one that leaves the function)
    parallel_chunks(nrpool, 1, [&](long long b, long long e, int) {
        for (long long pl = b; pl < e; pl++)
        {
            big_vector<int>().swap(rpools[(size_t) pl].col);
            big_vector<Part>().swap(rpools[(size_t) pl].ownp);
            big_vector<unsigned char>().swap(rpools[(size_t) pl].ownc);
        }
    });
    clk.lap("build_team2: release");
}

// ---- team2r streams (panel_format.h) ----------------------------------------------------------------------
bool build_team2r(const PanelHost &p, int nrow, const int *rowptr, const int *colidx, Team2RHost *out, const int *colpos, TeamSeed *seed)
{
    constexpr int W = 8, T = 8;
    const int G = out->G == 2 ? 2 : 4;
    out->G = G;
    constexpr int RD = TEAM2R_ROWDMA;
    const int S = 8 * G * RD, SLOTB = 1024 / G, PERW = G * RD;              // slots of a round, bytes of a slot, slots a wave fetches
    const int ZERO = team2r_zero(RD);
    PhaseClock clk;
    released_async<TeamHost> th_owner;
    TeamHost &th = *th_owner;
    // (teams as team2 builds them, KKT systems' primal + dual mixes included: the B rows both kinds share are fetched once --
    //  teams of one kind of panel give the waves of a round more equal steps (useful / issued row slots 0.53 against 0.32) and are
    //  slower all the same: nlpkkt stand-in n = 32 0.533 against 0.500 ms, at nlpkkt240 size 8.96 against 7.83, where the kernel is
    //  bound by what it fetches from beyond L2)
    build_teams(p, nrow, rowptr, colidx, &th, T, colpos, false, -1, seed);
    const bool with_vals = !p.pval.empty() || p.pcol.empty();
    clk.lap("build_team2r: build_teams total");
    const int nteam = th.nteam;
    out->nteam = nteam;
    out->lattice = th.lattice;
    out->tpanel = th.tpanel;
    out->torder = th.torder;
    auto mask_of = [&](size_t q) { return (p.pmask4[q >> 2] >> (8 * (q & 3))) & 0xFFu; };
    // Order of a team's union entries over its rounds.  team2 sorts by a PHASE (position mod 8 first) so that its parts are
    // contiguous row ranges; here a round should give every row of a panel about the same number of nonzeros (a wave's steps are
    // the maximum over its 8 rows): the natural order of the columns does that -- a run of consecutive columns is one mesh line,
    // which the 8 consecutive rows of a panel touch alike -- where the phase order gives a round the columns that only one or two
    // of the 8 rows have (nlpkkt stand-in: 2.7 padded steps per nonzero against 1.3 with this order).
    auto key = [&](int q) -> long long {
        const int c = th.tcol[(size_t) q];
        return c >= 0 ? (long long) (colpos ? colpos[c] : c) : (1LL << 40) + (long long) (~c);
    };
    struct ItemR { unsigned char slot; int src; };                            // a panel entry of a wave placed on a slot of the round
    struct TeamOutR
    {
        int nr = 0, anycol = 0;
        size_t ocol = 0, oiptr = 0, oitem = 0, olp = 0;     // where the team's arrays start in its pool
        long long wunits[8] = {0, 0, 0, 0, 0, 0, 0, 0}, steps = 0, filled = 0;   // 16-byte units of every wave's stream, padded steps, filled slots
        const int *col = nullptr;             // nr * S (TEAM2_NOCOL = empty)
        const int *iptr = nullptr;            // (nr * W) + 1: items of (round, wave)
        const ItemR *items = nullptr;
        const unsigned char *lp = nullptr;    // nr * W: padded steps
    };
    // (pools of consecutive teams instead of four vectors per team: see build_team2)
    struct RoundPoolR { big_vector<int> col, iptr; big_vector<ItemR> items; big_vector<unsigned char> lp; };
    const int rpool = (int) std::min<long long>(2048, std::max<long long>(32, nteam / (4LL * host_threads())));
    const int nrpool = (nteam + rpool - 1) / rpool;
    std::vector<RoundPoolR> rpools((size_t) nrpool);
    std::vector<TeamOutR> res((size_t) nteam);
    parallel_chunks(nrpool, 1, [&](long long pb, long long pe, int) {
        std::vector<int> nodes;
        std::vector<std::pair<long long, int>> keyed;
        std::vector<std::vector<ItemR>> wl((size_t) W);
        struct { std::vector<int> col, iptr; std::vector<ItemR> items; std::vector<unsigned char> lp; } to_v;     // the team being built
        for (long long pl = pb; pl < pe; pl++)
        {
        RoundPoolR &pool = rpools[(size_t) pl];
        const long long b = pl * rpool, e = std::min<long long>(nteam, b + rpool);
        for (long long g = b; g < e; g++)
        {
            TeamOutR &to = res[(size_t) g];
            to_v.col.clear(); to_v.iptr.clear(); to_v.items.clear(); to_v.lp.clear();
            nodes.clear();
            for (int q = th.tptr[(size_t) g]; q < th.tptr[(size_t) g + 1]; q++)
            {
                bool used = false;
                for (int w = 0; w < T; w++) used = used || th.tsrc[(size_t) q * T + (size_t) w] >= 0;
                if (used) nodes.push_back(q);
            }
            keyed.resize(nodes.size());
            for (size_t i = 0; i < nodes.size(); i++) keyed[i] = {key(nodes[i]), nodes[i]};
            std::sort(keyed.begin(), keyed.end());              // (ties keep the ascending order of the nodes)
            for (size_t i = 0; i < nodes.size(); i++) nodes[i] = keyed[i].second;
            // (dealing the ordered entries out to the rounds like cards, so that the waves of a round have equal steps -- mean / max
            //  0.59 -> 0.86 -- was 15 % slower: consecutive columns in a round are consecutive B rows in time for every team of the XCD)
            to.anycol = nodes.empty() ? 0 : th.tcol[(size_t) nodes[0]];
            to_v.iptr.push_back(0);
            const size_t nn = nodes.size();
            std::vector<char> taken(nn, 0);
            size_t head = 0, left = nn;
            while (left > 0)
            {
                int cnt[W][8];
                for (int w = 0; w < W; w++)
                    for (int r = 0; r < 8; r++) cnt[w][r] = 0;
                for (int w = 0; w < W; w++) wl[(size_t) w].clear();
                int nslot = 0;
                const size_t base_col = to_v.col.size();
                to_v.col.resize(base_col + (size_t) S, TEAM2_NOCOL);
                while (head < nn && taken[head]) head++;
                int seen = 0;
                for (size_t t = head; t < nn && nslot < S && seen < 3 * S; t++)
                {
                    if (taken[t]) continue;
                    seen++;
                    const int q = nodes[t];
                    bool fits = true;
                    for (int w = 0; w < W && fits; w++)
                    {
                        const int src = th.tsrc[(size_t) q * T + (size_t) w];
                        if (src < 0) continue;
                        const unsigned mk = mask_of((size_t) src);
                        for (int r = 0; r < 8; r++)
                            if (((mk >> r) & 1) && cnt[w][r] + 1 > TEAM2R_LCAP) fits = false;
                    }
                    if (!fits) continue;
                    for (int w = 0; w < W; w++)
                    {
                        const int src = th.tsrc[(size_t) q * T + (size_t) w];
                        if (src < 0) continue;
                        const unsigned mk = mask_of((size_t) src);
                        for (int r = 0; r < 8; r++) cnt[w][r] += (mk >> r) & 1;
                        ItemR it;
                        it.slot = (unsigned char) nslot;
                        it.src = src;
                        wl[(size_t) w].push_back(it);
                    }
                    to_v.col[base_col + (size_t) nslot] = th.tcol[(size_t) q];
                    nslot++;
                    taken[t] = 1;
                    left--;
                }
                if (nslot == 0) { fprintf(stderr, "[FATAL] team2r scheduler: a round placed nothing\n"); abort(); }
                for (int w = 0; w < W; w++)
                {
                    int mx = 0;
                    for (int r = 0; r < 8; r++) mx = std::max(mx, cnt[w][r]);
                    to_v.lp.push_back((unsigned char) ((mx + 1) / 2 * 2));           // steps come in pairs (the kernel's half chunk)
                    to_v.items.insert(to_v.items.end(), wl[(size_t) w].begin(), wl[(size_t) w].end());
                    to_v.iptr.push_back((int) to_v.items.size());
                }
                to.nr++;
            }
            if (to.nr == 0)                                                   // no nonzero in 64 rows: one empty round (the kernel's pipeline wants one)
            {
                to_v.col.assign((size_t) S, TEAM2_NOCOL);
                for (int w = 0; w < W; w++)
                {
                    to_v.lp.push_back(0);
                    to_v.iptr.push_back(0);
                }
                to.nr = 1;
            }
            to.ocol = pool.col.size(); to.oiptr = pool.iptr.size(); to.oitem = pool.items.size(); to.olp = pool.lp.size();
            pool.col.insert(pool.col.end(), to_v.col.begin(), to_v.col.end());
            pool.iptr.insert(pool.iptr.end(), to_v.iptr.begin(), to_v.iptr.end());
            pool.items.insert(pool.items.end(), to_v.items.begin(), to_v.items.end());
            pool.lp.insert(pool.lp.end(), to_v.lp.begin(), to_v.lp.end());
        }
        for (long long g = b; g < e; g++)                                     // (the pool's arrays have stopped growing)
        {
            TeamOutR &to = res[(size_t) g];
            to.col = pool.col.data() + to.ocol; to.iptr = pool.iptr.data() + to.oiptr; to.items = pool.items.data() + to.oitem; to.lp = pool.lp.data() + to.olp;
            for (int w = 0; w < W; w++)
                for (int r = 0; r < to.nr; r++)
                {
                    to.wunits[w] += 5LL * to.lp[(size_t) r * W + (size_t) w] + 4;   // 80 Lp bytes of values and offsets + the 64-byte header
                    to.steps += to.lp[(size_t) r * W + (size_t) w];
                }
            for (size_t i = 0; i < (size_t) to.nr * (size_t) S; i++) to.filled += to.col[i] != TEAM2_NOCOL;
        }
        }
    });
    clk.lap("build_team2r: rounds");
    out->tinfo.assign((size_t) nteam * 2, 0);
    out->tvoff.assign((size_t) nteam * W + 1, 0);
    long long rec0 = 0, run = 0;                                              // run: units of 16 bytes
    out->rounds = out->steps = out->nnz = out->slots_filled = 0;
    for (int g = 0; g < nteam; g++)
    {
        const TeamOutR &to = res[(size_t) g];
        out->tinfo[(size_t) g * 2] = to.nr;
        out->tinfo[(size_t) g * 2 + 1] = (int) rec0;
        rec0 += to.nr;
        out->rounds += to.nr;
        for (int w = 0; w < W; w++)
        {
            out->tvoff[(size_t) g * W + (size_t) w] = run;
            run += to.wunits[w];
        }
        out->steps += to.steps;
        out->slots_filled += to.filled;
    }
    out->tvoff[(size_t) nteam * W] = run;
    out->nwords = run * 2;
    if (rec0 >= (1LL << 31) / 128 || run >= (1LL << 31)) return false;     // (the caller falls back to the row-panel kernels)
    {
        std::vector<int> cut(9, nteam);
        long long total = 0;
        for (int g = 0; g < nteam; g++) total += res[(size_t) g].nr + 1;
        cut[0] = 0;
        long long acc = 0;
        int x = 1;
        for (int i = 0; i < nteam && x < 8; i++)
        {
            acc += res[(size_t) out->torder[(size_t) i]].nr + 1;
            while (x < 8 && acc * 8 >= total * x) cut[(size_t) x++] = i + 1;
        }
        int cpx = 1;
        for (int q = 0; q < 8; q++) cpx = std::max(cpx, cut[(size_t) q + 1] - cut[(size_t) q]);
        out->tgrid.assign((size_t) cpx * 8, -1);
        for (int q = 0; q < 8; q++)
            for (int i = cut[(size_t) q]; i < cut[(size_t) q + 1]; i++) out->tgrid[(size_t) q * cpx + (size_t) (i - cut[(size_t) q])] = out->torder[(size_t) i];
    }
    parallel_fill(out->trec, (size_t) (rec0 + 1) * 128, 0u);
    parallel_fill(out->tval, (size_t) run * 2 + 512, 0.0);
    big_vector<uint32_t> slot_of;
    slot_of.resize(p.pcol.size() * 8);
    parallel_chunks(nteam, 32, [&](long long b, long long e, int) {
        for (long long g = b; g < e; g++)
        {
            const TeamOutR &to = res[(size_t) g];
            for (int w = 0; w < W; w++)
            {
                long long at16 = 0;                                           // units of 16 bytes inside the wave's stream
                const long long w0 = out->tvoff[(size_t) g * W + (size_t) w] * 2;   // first 8-byte word of the stream
                for (int r = 0; r < to.nr; r++)
                {
                    uint32_t *rec = &out->trec[((size_t) out->tinfo[(size_t) g * 2 + 1] + (size_t) r) * 128 + (size_t) w * 16];
                    const int Lp = to.lp[(size_t) r * W + (size_t) w];
                    rec[0] = (uint32_t) Lp;
                    rec[1] = (uint32_t) at16;
                    for (int j = 0; j < PERW; j++)
                    {
                        const int c = to.col[(size_t) r * S + (size_t) (w * PERW + j)];
                        rec[2 + j] = (uint32_t) (c != TEAM2_NOCOL ? c : to.anycol);
                    }
                    double *vals = &out->tval[(size_t) (w0 + at16 * 2)];                         // [8][Lp]
                    uint16_t *offs = reinterpret_cast<uint16_t *>(vals + (size_t) 8 * Lp);       // [8][Lp]
                    for (int i = 0; i < 8 * Lp; i++) offs[i] = (uint16_t) ZERO;
                    int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int i = to.iptr[(size_t) r * W + (size_t) w]; i < to.iptr[(size_t) r * W + (size_t) w + 1]; i++)
                    {
                        const ItemR &it = to.items[(size_t) i];
                        const unsigned mk = mask_of((size_t) it.src);
                        for (int rr = 0; rr < 8; rr++)
                            if ((mk >> rr) & 1)
                            {
                                const int st = fill[rr]++;
                                if (with_vals) vals[(size_t) rr * Lp + (size_t) st] = p.pval[(size_t) it.src * 8 + (size_t) rr];
                                offs[(size_t) rr * Lp + (size_t) st] = (uint16_t) (it.slot * SLOTB);
                                slot_of[(size_t) it.src * 8 + (size_t) rr] = (uint32_t) (w0 + at16 * 2 + (long long) rr * Lp + st);
                            }
                    }
                    at16 += 5LL * Lp + 4;
                }
                // headers: the record of round r + 2 behind the block of round r
                for (int r = 0; r + 2 < to.nr; r++)
                {
                    const uint32_t *rec = &out->trec[((size_t) out->tinfo[(size_t) g * 2 + 1] + (size_t) r) * 128 + (size_t) w * 16];
                    const uint32_t *rec2 = rec + 2 * 128;
                    uint32_t *hdr = reinterpret_cast<uint32_t *>(&out->tval[(size_t) (w0 + (long long) rec[1] * 2 + 10LL * rec[0])]);
                    for (int i = 0; i < 16; i++) hdr[i] = rec2[i];
                }
            }
        }
    });
    // what a workgroup needs when it turns to an entry of the launch grid
    parallel_fill(out->tent, out->tgrid.size() * 256, 0u);
    parallel_chunks((long long) out->tgrid.size(), 256, [&](long long b, long long e, int) {
        for (long long en = b; en < e; en++)
        {
            const int g = out->tgrid[(size_t) en];
            if (g < 0) continue;
            const int nr = out->tinfo[(size_t) g * 2];
            for (int w = 0; w < W; w++)
            {
                uint32_t *t = &out->tent[((size_t) en * 8 + (size_t) w) * 32];
                const uint32_t *rec = &out->trec[(size_t) out->tinfo[(size_t) g * 2 + 1] * 128 + (size_t) w * 16];
                const long long vo = out->tvoff[(size_t) g * W + (size_t) w];
                t[0] = (uint32_t) nr;
                t[1] = (uint32_t) out->tpanel[(size_t) g * W + (size_t) w];
                t[2] = (uint32_t) (vo & 0xFFFFFFFFLL);
                t[3] = (uint32_t) (vo >> 32);
                for (int i = 0; i < 10; i++) t[4 + i] = rec[i];
                if (nr > 1)
                    for (int i = 0; i < 10; i++) t[14 + i] = rec[128 + i];
                for (int i = 0; i < 8; i++) t[24 + i] = 0xFFFFFFFFu;
            }
        }
    });
    clk.lap("build_team2r: records, streams, entry table");
    out->vmap.resize(p.pmap.size());
    out->nnz = (long long) p.pmap.size();
    parallel_chunks((long long) p.pmap.size(), 1 << 18, [&](long long b, long long e, int) {
        for (long long nz = b; nz < e; nz++) out->vmap[(size_t) nz] = slot_of[(size_t) p.pmap[(size_t) nz]];
    });
    clk.lap("build_team2r: value-update map");
    // (hundreds of thousands of small vectors: released by all threads, not by the one that leaves the function)
    parallel_chunks(nrpool, 1, [&](long long b, long long e, int) {
        for (long long pl = b; pl < e; pl++)
        {
            big_vector<int>().swap(rpools[(size_t) pl].col);
            big_vector<int>().swap(rpools[(size_t) pl].iptr);
            big_vector<ItemR>().swap(rpools[(size_t) pl].items);
            big_vector<unsigned char>().swap(rpools[(size_t) pl].lp);
        }
    });
    clk.lap("build_team2r: release");
    return true;
}

void apply_team_schedule(PanelHost *p, const TeamHost &t)
{
    const int R = p->R, T = t.T;
    p->team_waves = T;
    big_vector<int> ncol(p->pcol.size(), 0);
    big_vector<uint32_t> nmask4(p->pmask4.size(), 0u);
    big_vector<double> nval(p->pval.size(), 0.0);
    std::vector<long long> moved(p->pcol.size(), -1);        // old entry -> new entry
    auto mask_of = [&](size_t q) { return (p->pmask4[q >> 2] >> (8 * (q & 3))) & 0xFFu; };
    for (int g = 0; g < t.nteam; g++)
        for (int w = 0; w < T; w++)
        {
            const int panel = t.tpanel[(size_t) g * T + w];
            if (panel < 0) continue;
            size_t dst = (size_t) p->pptr[panel];
            int last = 0;
            for (int q = t.tptr[(size_t) g]; q < t.tptr[(size_t) g + 1]; q++)
            {
                const int src = t.tsrc[(size_t) q * T + w];
                if (src < 0) continue;
                ncol[dst] = p->pcol[(size_t) src];
                nmask4[dst >> 2] |= mask_of((size_t) src) << (8 * (dst & 3));
                memcpy(&nval[dst * R], &p->pval[(size_t) src * R], sizeof(double) * R);
                moved[(size_t) src] = (long long) dst;
                last = ncol[dst];
                dst++;
            }
            for (; dst < (size_t) p->pptr[panel + 1]; dst++) ncol[dst] = last;       // padding: valid row, mask 0
        }
    for (uint32_t &slot : p->pmap) slot = (uint32_t) (moved[slot / R] * R + slot % R);
    p->pcol.swap(ncol);
    p->pmask4.swap(nmask4);
    p->pval.swap(nval);
    p->porder.assign((size_t) t.nteam * T, -1);
    p->psync.assign((size_t) t.nteam, 0);
    for (int pos = 0; pos < t.nteam; pos++)
    {
        int minnr = 1 << 30;
        for (int w = 0; w < T; w++)
        {
            const int panel = t.tpanel[(size_t) t.torder[(size_t) pos] * T + w];
            p->porder[(size_t) pos * T + w] = panel;
            if (panel >= 0) minnr = std::min(minnr, (p->pptr[panel + 1] - p->pptr[panel]) / PANEL_PAD);
        }
        p->psync[(size_t) pos] = (minnr == (1 << 30) || minnr < 2) ? 0 : minnr - 1;
    }
}

}  // namespace crp
