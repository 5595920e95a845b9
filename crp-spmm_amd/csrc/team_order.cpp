// team_order.cpp -- recursive graph bisection order (team_order.h).
//
// One bisection of a vertex set: scalar fields over the set, each the difference of the hop distances to the two ends
// of a far pair (a pseudo-diameter; a second pair found far from both ends of the first, a third far from all four) and
// the sums and differences of those three -- on a mesh the far pairs are diagonals, their sums and differences the
// axes --; every field is split at its work-weighted median and the split that cuts the least edge weight (B rows shared
// across it) wins, after a few Jacobi sweeps that straighten its boundary.  Disconnected sets are split between
// components first.  The halves are bisected again, level by level, until a part holds one generation.
#include "team_order.h"
#include <math.h>
#include <algorithm>
#include <numeric>
#include "par.h"

namespace crp {

namespace {

struct Graph
{
    int n;
    const long long *ptr;
    const int *adj, *wgt, *work;
    int leaf;
    std::vector<int> sub, loc;      // task of every vertex at the current level, its index inside the task
};

struct Task
{
    std::vector<int> nodes;         // vertices of the part (global ids), in the order the parent left them
    long long off = 0;              // where the part starts in the final order
};

// hop distances from `src` (local index) inside task `tid`; dist must be all -1 on entry (unreached stays -1), queue = the
// vertices reached, in order.  Returns the vertex reached last.  unvisit() restores dist for the next search at the cost
// of the vertices reached (a set of many small components must not pay its whole size per search).
void unvisit(std::vector<int> &dist, const std::vector<int> &queue)
{
    for (int v : queue) dist[(size_t) v] = -1;
}

int bfs(const Graph &g, const Task &t, int tid, int src, std::vector<int> &dist, std::vector<int> &queue)
{
    queue.clear();
    queue.push_back(src);
    dist[(size_t) src] = 0;
    for (size_t h = 0; h < queue.size(); h++)
    {
        const int v = queue[h], gv = t.nodes[(size_t) v];
        for (long long e = g.ptr[gv]; e < g.ptr[gv + 1]; e++)
        {
            const int gw = g.adj[e];
            if (g.sub[(size_t) gw] != tid) continue;
            const int w = g.loc[(size_t) gw];
            if (dist[(size_t) w] >= 0) continue;
            dist[(size_t) w] = dist[(size_t) v] + 1;
            queue.push_back(w);
        }
    }
    return queue.back();
}

// where a field is cut: the work-weighted median of the sorted order, moved to a multiple of `leaf`
int split_point(const Graph &g, const Task &t, const std::vector<int> &ord)
{
    const int n = (int) ord.size();
    long long total = 0;
    for (int v : ord) total += g.work[t.nodes[(size_t) v]];
    long long acc = 0;
    int k = 0;
    while (k < n && 2 * acc < total) acc += g.work[t.nodes[(size_t) ord[(size_t) k++]]];
    const int leaf = g.leaf;
    if (n > 2 * leaf) k = std::max(leaf, std::min(n - leaf, (k + leaf / 2) / leaf * leaf));
    else k = leaf;                                    // (n > leaf here): one full generation, then the rest
    return std::max(1, std::min(n - 1, k));
}

long long cut_weight(const Graph &g, const Task &t, int tid, const std::vector<char> &left)
{
    long long cut = 0;
    const int n = (int) t.nodes.size();
    for (int v = 0; v < n; v++)
    {
        if (!left[(size_t) v]) continue;
        const int gv = t.nodes[(size_t) v];
        for (long long e = g.ptr[gv]; e < g.ptr[gv + 1]; e++)
        {
            const int gw = g.adj[e];
            if (g.sub[(size_t) gw] == tid && !left[(size_t) g.loc[(size_t) gw]]) cut += g.wgt[e];
        }
    }
    return cut;
}

void sort_by_field(const std::vector<float> &f, std::vector<int> &ord)
{
    ord.resize(f.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return f[(size_t) a] != f[(size_t) b] ? f[(size_t) a] < f[(size_t) b] : a < b; });
}

// Splits task `tid` (more than `leaf` vertices) into two; the vertex lists come out in field order.
void bisect(const Graph &g, const Task &t, int tid, bool inner_parallel, Task *lo, Task *hi)
{
    const int n = (int) t.nodes.size();
    std::vector<int> dist((size_t) n, -1), queue;
    queue.reserve((size_t) n);
    std::vector<std::vector<float>> fields;
    // components (in order of their first vertex); the far pair of each
    std::vector<int> comp((size_t) n, -1);
    int ncomp = 0;
    std::vector<float> f1((size_t) n, 0.0f);
    std::vector<int> du((size_t) n), dv((size_t) n);
    for (int s = 0; s < n; s++)
    {
        if (comp[(size_t) s] >= 0) continue;
        const int u = bfs(g, t, tid, s, dist, queue);
        for (int v : queue) comp[(size_t) v] = ncomp;
        unvisit(dist, queue);
        const int v_end = bfs(g, t, tid, u, dist, queue);
        for (int v : queue) du[(size_t) v] = dist[(size_t) v];
        unvisit(dist, queue);
        bfs(g, t, tid, v_end, dist, queue);
        for (int v : queue) dv[(size_t) v] = dist[(size_t) v];
        for (int v : queue) f1[(size_t) v] = (float) (du[(size_t) v] - dv[(size_t) v]);
        unvisit(dist, queue);
        ncomp++;
    }
    if (ncomp > 1)
    {
        // between components first: (component, field inside it); 1e6 exceeds any hop difference
        for (int v = 0; v < n; v++) f1[(size_t) v] += 2.0e6f * (float) comp[(size_t) v];
        fields.push_back(f1);
    }
    else
    {
        auto far_from = [&](const std::vector<const std::vector<int> *> &ds) {
            int best = 0, bd = -1;
            for (int v = 0; v < n; v++)
            {
                int m = 1 << 30;
                for (const std::vector<int> *d : ds) m = std::min(m, (*d)[(size_t) v]);
                if (m > bd) { bd = m; best = v; }
            }
            return best;
        };
        std::vector<int> d3((size_t) n), d4((size_t) n), d5((size_t) n), d6((size_t) n);
        const int u2 = far_from({&du, &dv});
        const int v2 = bfs(g, t, tid, u2, dist, queue);
        d3 = dist;
        unvisit(dist, queue);
        bfs(g, t, tid, v2, dist, queue);
        d4 = dist;
        unvisit(dist, queue);
        const int u3 = far_from({&du, &dv, &d3, &d4});
        const int v3 = bfs(g, t, tid, u3, dist, queue);
        d5 = dist;
        unvisit(dist, queue);
        bfs(g, t, tid, v3, dist, queue);
        d6 = dist;
        unvisit(dist, queue);
        std::vector<float> f2((size_t) n), f3((size_t) n);
        for (int v = 0; v < n; v++)
        {
            f2[(size_t) v] = (float) (d3[(size_t) v] - d4[(size_t) v]);
            f3[(size_t) v] = (float) (d5[(size_t) v] - d6[(size_t) v]);
        }
        static const int comb[13][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1},
                                        {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {-1, 1, 1}};
        fields.resize(13);
        for (int c = 0; c < 13; c++)
        {
            fields[(size_t) c].resize((size_t) n);
            for (int v = 0; v < n; v++)
                fields[(size_t) c][(size_t) v] = (float) comb[c][0] * f1[(size_t) v] + (float) comb[c][1] * f2[(size_t) v] + (float) comb[c][2] * f3[(size_t) v];
        }
    }
    // the field whose median split cuts least
    const int nf = (int) fields.size();
    std::vector<long long> cuts((size_t) nf, 0);
    auto evaluate = [&](int c) {
        std::vector<int> ord;
        sort_by_field(fields[(size_t) c], ord);
        const int k = split_point(g, t, ord);
        std::vector<char> left((size_t) n, 0);
        for (int i = 0; i < k; i++) left[(size_t) ord[(size_t) i]] = 1;
        cuts[(size_t) c] = cut_weight(g, t, tid, left);
    };
    if (inner_parallel && nf > 1) parallel_chunks(nf, 1, [&](long long b, long long e, int) { for (long long c = b; c < e; c++) evaluate((int) c); });
    else for (int c = 0; c < nf; c++) evaluate(c);
    int best = 0;
    for (int c = 1; c < nf; c++)
        if (cuts[(size_t) c] < cuts[(size_t) best]) best = c;
    std::vector<float> f = fields[(size_t) best];
    fields.clear();
    if (ncomp == 1)
    {
        // Jacobi sweeps: a vertex moves towards the weighted mean of its neighbours (the boundary of a hop-count field
        // is a staircase); kept only if the cut does not grow
        std::vector<float> s = f, nx((size_t) n);
        for (int it = 0; it < 6; it++)
        {
            for (int v = 0; v < n; v++)
            {
                const int gv = t.nodes[(size_t) v];
                double sw = 0.0, sf = 0.0;
                for (long long e = g.ptr[gv]; e < g.ptr[gv + 1]; e++)
                {
                    const int gw = g.adj[e];
                    if (g.sub[(size_t) gw] != tid) continue;
                    sw += (double) g.wgt[e];
                    sf += (double) g.wgt[e] * (double) s[(size_t) g.loc[(size_t) gw]];
                }
                nx[(size_t) v] = sw > 0.0 ? (float) (0.5 * (double) s[(size_t) v] + 0.5 * sf / sw) : s[(size_t) v];
            }
            s.swap(nx);
        }
        std::vector<int> ord;
        sort_by_field(s, ord);
        const int k = split_point(g, t, ord);
        std::vector<char> left((size_t) n, 0);
        for (int i = 0; i < k; i++) left[(size_t) ord[(size_t) i]] = 1;
        if (cut_weight(g, t, tid, left) <= cuts[(size_t) best]) f.swap(s);
    }
    std::vector<int> ord;
    sort_by_field(f, ord);
    const int k = split_point(g, t, ord);
    lo->nodes.resize((size_t) k);
    hi->nodes.resize((size_t) (n - k));
    for (int i = 0; i < k; i++) lo->nodes[(size_t) i] = t.nodes[(size_t) ord[(size_t) i]];
    for (int i = k; i < n; i++) hi->nodes[(size_t) (i - k)] = t.nodes[(size_t) ord[(size_t) i]];
    lo->off = t.off;
    hi->off = t.off + k;
}

}  // namespace

void bisection_order(int n, const std::vector<long long> &ptr, const std::vector<int> &adj, const std::vector<int> &wgt,
                     const std::vector<int> &work, int leaf, std::vector<int> *order)
{
    order->assign((size_t) std::max(n, 0), 0);
    if (n <= 0) return;
    if (leaf < 1) leaf = 1;
    Graph g;
    g.n = n; g.ptr = ptr.data(); g.adj = adj.data(); g.wgt = wgt.data(); g.work = work.data(); g.leaf = leaf;
    g.sub.assign((size_t) n, 0);
    g.loc.resize((size_t) n);
    std::vector<Task> level(1);
    level[0].nodes.resize((size_t) n);
    std::iota(level[0].nodes.begin(), level[0].nodes.end(), 0);
    const int nthr = host_threads();
    while (!level.empty())
    {
        const int nt = (int) level.size();
        // every vertex is re-labelled at every level (the task it is in now, or -1 once its part is final)
        std::vector<Task> next((size_t) nt * 2);
        std::vector<char> live((size_t) nt, 0);
        for (int i = 0; i < nt; i++)
        {
            Task &t = level[(size_t) i];
            if ((int) t.nodes.size() <= leaf)
            {
                std::copy(t.nodes.begin(), t.nodes.end(), order->begin() + t.off);
                for (int v : t.nodes) g.sub[(size_t) v] = -1;
                continue;
            }
            live[(size_t) i] = 1;
            for (size_t k = 0; k < t.nodes.size(); k++)
            {
                g.sub[(size_t) t.nodes[k]] = i;
                g.loc[(size_t) t.nodes[k]] = (int) k;
            }
        }
        const bool inner = nt * 4 < nthr;
        if (inner)
        {
            for (int i = 0; i < nt; i++)
                if (live[(size_t) i]) bisect(g, level[(size_t) i], i, true, &next[(size_t) 2 * i], &next[(size_t) 2 * i + 1]);
        }
        else
            parallel_chunks(nt, 1, [&](long long b, long long e, int) {
                for (long long i = b; i < e; i++)
                    if (live[(size_t) i]) bisect(g, level[(size_t) i], (int) i, false, &next[(size_t) 2 * i], &next[(size_t) 2 * i + 1]);
            });
        std::vector<Task> compact;
        for (int i = 0; i < nt; i++)
            if (live[(size_t) i])
            {
                compact.push_back(std::move(next[(size_t) 2 * i]));
                compact.push_back(std::move(next[(size_t) 2 * i + 1]));
            }
        level.swap(compact);
    }
}

}  // namespace crp

// ---- lattice_block_order (team_order.h) -------------------------------------------------------------------------------
#include "par.h"

namespace crp {

namespace {

// misses of the replay of one run (teams run[0 .. n)) against an exact LRU of cap rows.  last = scratch of one entry per row
// id (all zero on entry and on return); a row id = column c >= 0, or nb0 + ~c for a row of the second source.
long long replay_run(const int *run, int n, int W, int slots, int cap, int nocol, int nb0, const int *const *cols, const int *nrounds,
                     std::vector<long long> &last, std::vector<int> &touched, std::vector<std::pair<int, long long>> &events)
{
    long long now = 0, miss = 0;
    size_t head = 0;
    int cached = 0;
    events.clear();
    touched.clear();
    std::vector<std::pair<int, int>> active;            // (team, next round)
    int qi = 0;
    while (qi < n || !active.empty())
    {
        while ((int) active.size() < slots && qi < n)
        {
            if (nrounds[run[qi]] > 0) active.push_back({run[qi], 0});
            qi++;
        }
        size_t keep = 0;
        for (size_t i = 0; i < active.size(); i++)
        {
            const int g = active[i].first, r = active[i].second;
            const int *c = cols[g] + (size_t) r * W;
            for (int w = 0; w < W; w++)
            {
                if (c[w] == nocol) continue;
                const int id = c[w] >= 0 ? c[w] : nb0 + ~c[w];
                now++;
                if (last[(size_t) id] == 0)
                {
                    miss++;
                    cached++;
                    touched.push_back(id);
                    while (cached > cap)
                    {
                        // the oldest use that is still the last use of its row: that row leaves
                        const std::pair<int, long long> ev = events[head++];
                        if (last[(size_t) ev.first] == ev.second) { last[(size_t) ev.first] = 0; cached--; }
                    }
                }
                last[(size_t) id] = now;
                events.push_back({id, now});
            }
            if (r + 1 < nrounds[g]) active[keep++] = {g, r + 1};
        }
        active.resize(keep);
    }
    for (int id : touched) last[(size_t) id] = 0;
    return miss;
}

}  // namespace

bool lattice_block_order(int nteam, const int *lat, int W, int slots, int lru_rows, int nocol,
                         const int *const *cols, const int *nrounds, std::vector<int> *order, LatticeOrderInfo *info)
{
    if (nteam < 16 * slots || (int) order->size() != nteam) return false;
    int A = 0, B = 0, Tn = 0;
    for (int g = 0; g < nteam; g++)
    {
        A = std::max(A, lat[(size_t) g * 3] + 1);
        B = std::max(B, lat[(size_t) g * 3 + 1] + 1);
        Tn = std::max(Tn, lat[(size_t) g * 3 + 2] + 1);
    }
    if (A < 1 || B < 1 || Tn < 1) return false;
    int nb0 = 0, nb1 = 0;                               // row ids: B0 rows, then B1 rows
    for (int g = 0; g < nteam; g++)
        for (long long q = 0; q < (long long) nrounds[g] * W; q++)
        {
            const int c = cols[g][q];
            if (c == nocol) continue;
            if (c >= 0) nb0 = std::max(nb0, c + 1); else nb1 = std::max(nb1, ~c + 1);
        }
    struct Cand { int pa, pb, bt, ba, bb, flags; };     // flags: 1 = blocks with t slowest, 2 = teams inside a block with t slowest
    std::vector<Cand> cands;
    cands.push_back({0, 0, 0, 0, 0, 0});                // the given order
    const int sizes[] = {1, 2, 3, 4, 6, 8, 16, 32};
    for (int pa : {1, 2, 4, 8})
    {
        const int pb = 8 / pa;
        if (pa > A || pb > B) continue;
        for (int bt : sizes)
            for (int ba : sizes)
                for (int bb : sizes)
                {
                    if (bt > Tn || ba > (A + pa - 1) / pa || bb > (B + pb - 1) / pb) continue;
                    const int vol = bt * ba * bb;
                    if (vol < slots * 3 / 4 || vol > slots * 3 / 2) continue;
                    for (int fl = 0; fl < 4; fl++) cands.push_back({pa, pb, bt, ba, bb, fl});
                }
    }
    std::vector<long long> work((size_t) nteam);
    long long total = 0;
    for (int g = 0; g < nteam; g++) { work[(size_t) g] = nrounds[g] + 4; total += work[(size_t) g]; }
    const std::vector<int> given = *order;
    auto make = [&](const Cand &c, std::vector<int> *o) {
        *o = given;
        if (c.pa == 0) return;
        std::vector<uint64_t> key((size_t) nteam);
        for (int g = 0; g < nteam; g++)
        {
            const int a = lat[(size_t) g * 3], b = lat[(size_t) g * 3 + 1], t = lat[(size_t) g * 3 + 2];
            const int ia = (int) ((long long) a * c.pa / A), ib = (int) ((long long) b * c.pb / B);
            const uint64_t box = (uint64_t) (ia * c.pb + ib);
            const uint64_t Bt = (uint64_t) (t / c.bt), Ba = (uint64_t) (a / c.ba), Bb = (uint64_t) (b / c.bb);
            const uint64_t blk = (c.flags & 1) ? ((Bt << 24) | (Ba << 12) | Bb) : ((Ba << 24) | (Bb << 12) | Bt);
            const uint64_t it = (uint64_t) (t % c.bt), ja = (uint64_t) (a % c.ba), jb = (uint64_t) (b % c.bb);
            const uint64_t in = (c.flags & 2) ? ((it << 12) | (ja << 6) | jb) : ((ja << 12) | (jb << 6) | it);
            key[(size_t) g] = (box << 56) | (blk << 18) | in;
        }
        std::stable_sort(o->begin(), o->end(), [&](int x, int y) { return key[(size_t) x] < key[(size_t) y]; });
    };
    struct Scratch { std::vector<long long> last; std::vector<int> touched; std::vector<std::pair<int, long long>> events; };
    auto cost = [&](const std::vector<int> &o, int sample, std::initializer_list<int> runs, Scratch &sc) {
        if (sc.last.empty()) sc.last.assign((size_t) nb0 + (size_t) nb1 + 1, 0);
        // the launch grid's cut (build_team2): 8 contiguous runs of equal work
        int cut[9];
        cut[0] = 0;
        for (int x = 1; x <= 8; x++) cut[x] = nteam;
        long long acc = 0;
        int x = 1;
        for (int i = 0; i < nteam && x < 8; i++)
        {
            acc += work[(size_t) o[(size_t) i]];
            while (x < 8 && acc * 8 >= total * x) cut[x++] = i + 1;
        }
        long long miss = 0;
        for (int q : runs)
            miss += replay_run(o.data() + cut[q], std::min(sample, cut[q + 1] - cut[q]), W, slots, lru_rows, nocol, nb0, cols, nrounds, sc.last, sc.touched, sc.events);
        return miss;
    };
    // stage 1: every candidate on a short prefix of one run; stage 2: the best dozen (and the given order) on longer prefixes of two
    std::vector<long long> res(cands.size(), 0);
    parallel_chunks((long long) cands.size(), 4, [&](long long b, long long e, int) {
        std::vector<int> o;
        Scratch sc;
        for (long long i = b; i < e; i++)
        {
            make(cands[(size_t) i], &o);
            res[(size_t) i] = cost(o, 6 * slots, {3}, sc);
        }
    });
    std::vector<size_t> top(cands.size());
    for (size_t i = 0; i < cands.size(); i++) top[i] = i;
    std::sort(top.begin() + 1, top.end(), [&](size_t x, size_t y) { return res[x] < res[y]; });
    top.resize(std::min<size_t>(top.size(), 13));
    std::vector<long long> res2(top.size(), 0);
    parallel_chunks((long long) top.size(), 1, [&](long long b, long long e, int) {
        std::vector<int> o;
        Scratch sc;
        for (long long i = b; i < e; i++)
        {
            make(cands[top[(size_t) i]], &o);
            res2[(size_t) i] = cost(o, 24 * slots, {2, 5}, sc);
        }
    });
    size_t bi = 0;
    for (size_t i = 1; i < top.size(); i++)
        if (res2[i] < res2[bi]) bi = i;
    const size_t best = top[bi];
    for (size_t i = 0; i < top.size(); i++) res[top[i]] = res2[i];
    if (info)
    {
        info->miss_given = (double) res[0];
        info->miss_best = (double) res[best];
        info->candidates = (int) cands.size();
        info->pa = cands[best].pa; info->pb = cands[best].pb; info->bt = cands[best].bt; info->ba = cands[best].ba; info->bb = cands[best].bb;
        info->flags = cands[best].flags;
    }
    if (best == 0 || (double) res[best] > 0.97 * (double) res[0]) return false;
    make(cands[best], order);
    return true;
}

}  // namespace crp
