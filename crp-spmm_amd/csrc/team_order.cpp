// team_order.cpp -- processing order of lattice teams by search against an L2 model (team_order.h).
// (The recursive bisection of the team graph of round 3 lived here; it was removed in round 4 with the generation barrier it
//  served -- panel_format.cpp, build_teams.)
#include "team_order.h"
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <numeric>
#include <initializer_list>
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
    if (A < 1 || B < 1 || Tn < 1 || A >= (1 << 21) || B >= (1 << 21) || Tn >= (1 << 21)) return false;
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
        // sort key = (box, block, position inside the block); 21 bits per block coordinate
        std::vector<uint64_t> kblk((size_t) nteam);
        std::vector<uint32_t> kin((size_t) nteam);
        for (int g = 0; g < nteam; g++)
        {
            const int a = lat[(size_t) g * 3], b = lat[(size_t) g * 3 + 1], t = lat[(size_t) g * 3 + 2];
            const int ia = (int) ((long long) a * c.pa / A), ib = (int) ((long long) b * c.pb / B);
            const uint64_t box = (uint64_t) (ia * c.pb + ib);
            const uint64_t Bt = (uint64_t) (t / c.bt), Ba = (uint64_t) (a / c.ba), Bb = (uint64_t) (b / c.bb);
            const uint64_t blk = (c.flags & 1) ? ((Bt << 42) | (Ba << 21) | Bb) : ((Ba << 42) | (Bb << 21) | Bt);
            const uint32_t it = (uint32_t) (t % c.bt), ja = (uint32_t) (a % c.ba), jb = (uint32_t) (b % c.bb);
            kin[(size_t) g] = (c.flags & 2) ? ((it << 12) | (ja << 6) | jb) : ((ja << 12) | (jb << 6) | it);
            kblk[(size_t) g] = blk;
            kin[(size_t) g] |= (uint32_t) box << 18;                  // box in the bits above the in-block position ...
        }
        // ... but compared FIRST: (box, block, in-block)
        std::stable_sort(o->begin(), o->end(), [&](int x, int y) {
            const uint32_t bx = kin[(size_t) x] >> 18, by = kin[(size_t) y] >> 18;
            if (bx != by) return bx < by;
            if (kblk[(size_t) x] != kblk[(size_t) y]) return kblk[(size_t) x] < kblk[(size_t) y];
            return (kin[(size_t) x] & 0x3FFFFu) < (kin[(size_t) y] & 0x3FFFFu);
        });
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
