// locality.cpp -- processing order of the rows of A for B-row locality (locality.h).
//
// The reference leaves the inspector step to MKL (mkl_sparse_d_mm, /root/reference/src/rowpara_spmm.c:398-408)
// and offers a METIS re-partitioning only in its examples (/root/reference/examples/metis_mat_part.c:31-113,
// third-party, absent here).  What the GPU kernels need is narrower than a partition of the matrix: rows that
// are processed together (one row panel, one team, the waves in flight on one XCD) should read the same B
// rows.  The ROWS of A are only re-ordered for processing -- B is read in place by column index and C is
// written through a row map -- so the order is free.
//
//   1. row groups: consecutive rows with identical column lists (the unknowns of one node of a finite-element
//      mesh) stay together; the graph below has one vertex per group;
//   2. the vertices are cut into `nparts` parts of equal row count by recursive bisection along the
//      breadth-first level structure rooted at a pseudo-peripheral vertex (George & Liu): a part is a slab
//      between two level fronts, and the kernels hand every XCD one contiguous range of the order, so
//      every XCD's L2 sees one slab;
//   3. inside a part: reverse Cuthill-McKee from a pseudo-peripheral vertex of the part, which sweeps the slab
//      along its LONG axis -- the front it moves is the slab's narrow cross-section, i.e. the set of B rows
//      that has to stay in L2 is small.
// Cost: O(nnz) per bisection level plus a handful of breadth-first searches.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <string.h>
#include "locality.h"

namespace crp {

namespace {

struct Graph
{
    int n = 0;
    std::vector<int> ptr, adj, weight;       // symmetric adjacency without self loops; weight = rows of the vertex
};

// Breadth-first levels of the vertices with part[v] == pid reachable from root; returns the visit order and
// fills level[] for the visited vertices.  `mark` must hold a value != stamp for unvisited vertices.
void bfs(const Graph &g, const std::vector<int> &part, int pid, int root, int stamp, std::vector<int> &mark,
         std::vector<int> &level, std::vector<int> &order)
{
    order.clear();
    order.push_back(root);
    mark[(size_t) root] = stamp;
    level[(size_t) root] = 0;
    for (size_t head = 0; head < order.size(); head++)
    {
        const int u = order[head];
        for (int t = g.ptr[(size_t) u]; t < g.ptr[(size_t) u + 1]; t++)
        {
            const int v = g.adj[(size_t) t];
            if (part[(size_t) v] != pid || mark[(size_t) v] == stamp) continue;
            mark[(size_t) v] = stamp;
            level[(size_t) v] = level[(size_t) u] + 1;
            order.push_back(v);
        }
    }
}

// pseudo-peripheral vertex of the component of `start` inside part pid
int pseudo_peripheral(const Graph &g, const std::vector<int> &part, int pid, int start, int &stamp, std::vector<int> &mark,
                      std::vector<int> &level, std::vector<int> &order)
{
    int root = start, ecc = -1;
    for (int iter = 0; iter < 8; iter++)
    {
        bfs(g, part, pid, root, ++stamp, mark, level, order);
        const int e = level[(size_t) order.back()];
        if (e <= ecc) break;
        ecc = e;
        // a vertex of smallest degree in the last level
        int best = order.back(), bestdeg = g.ptr[(size_t) best + 1] - g.ptr[(size_t) best];
        for (size_t t = order.size(); t-- > 0;)
        {
            const int v = order[t];
            if (level[(size_t) v] != e) break;
            const int d = g.ptr[(size_t) v + 1] - g.ptr[(size_t) v];
            if (d < bestdeg) { best = v; bestdeg = d; }
        }
        if (best == root) break;
        root = best;
    }
    return root;
}

// All vertices of part pid in breadth-first order from pseudo-peripheral roots (one component after the other).
// rcm = true: neighbours are visited by ascending degree and every component's order is reversed.
// `placed` (one int per vertex, any content) marks with a fresh stamp what is already in `out`.
void part_order(const Graph &g, const std::vector<int> &part, int pid, const std::vector<int> &members, bool rcm, int &stamp,
                std::vector<int> &mark, std::vector<int> &level, std::vector<int> &placed, std::vector<int> &scratch,
                std::vector<int> &out)
{
    out.clear();
    std::vector<int> nb;
    const int pstamp = ++stamp;
    for (int m0 : members)
    {
        if (placed[(size_t) m0] == pstamp) continue;
        const int root = pseudo_peripheral(g, part, pid, m0, stamp, mark, level, scratch);
        const int st = ++stamp;
        const size_t first = out.size();
        out.push_back(root);
        mark[(size_t) root] = st;
        for (size_t head = first; head < out.size(); head++)
        {
            const int u = out[head];
            nb.clear();
            for (int t = g.ptr[(size_t) u]; t < g.ptr[(size_t) u + 1]; t++)
            {
                const int v = g.adj[(size_t) t];
                if (part[(size_t) v] != pid || mark[(size_t) v] == st) continue;
                mark[(size_t) v] = st;
                nb.push_back(v);
            }
            if (rcm)
                std::sort(nb.begin(), nb.end(), [&](int a, int b) {
                    const int da = g.ptr[(size_t) a + 1] - g.ptr[(size_t) a], db = g.ptr[(size_t) b + 1] - g.ptr[(size_t) b];
                    return da != db ? da < db : a < b;
                });
            out.insert(out.end(), nb.begin(), nb.end());
        }
        for (size_t t = first; t < out.size(); t++) placed[(size_t) out[t]] = pstamp;
        if (rcm) std::reverse(out.begin() + (long) first, out.end());
    }
}


// Steps 2 and 3 of the header comment on any weighted graph: `nparts` (a power of two) parts of equal weight by
// recursive bisection along breadth-first level structures, reverse Cuthill-McKee inside every part.
// -> the vertices part after part; first[q] = where part q starts in that order (nparts + 1 entries).
void slab_order(const Graph &g, int nparts, std::vector<int> *gorder_out, std::vector<int> *first)
{
    const int ng = g.n;
    int depth = 0;
    while ((1 << depth) < nparts) depth++;
    std::vector<int> part((size_t) ng, 0), mark((size_t) ng, 0), level((size_t) ng, 0), placed((size_t) ng, 0), scratch, order;
    int stamp = 0;
    std::vector<std::vector<int>> members(1);
    members[0].resize((size_t) ng);
    std::iota(members[0].begin(), members[0].end(), 0);
    for (int d = 0; d < depth; d++)
    {
        std::vector<std::vector<int>> next(members.size() * 2);
        for (size_t pid = 0; pid < members.size(); pid++)
        {
            // (part ids are re-assigned after every level: vertex v of part pid goes to 2 pid or 2 pid + 1)
            part_order(g, part, (int) pid, members[pid], false, stamp, mark, level, placed, scratch, order);
            long long total = 0, run = 0;
            for (int v : order) total += g.weight[(size_t) v];
            size_t cut = 0;
            while (cut < order.size() && (run + g.weight[(size_t) order[cut]] / 2) * 2 < total) run += g.weight[(size_t) order[cut++]];
            next[2 * pid].assign(order.begin(), order.begin() + (long) cut);
            next[2 * pid + 1].assign(order.begin() + (long) cut, order.end());
        }
        members.swap(next);
        for (size_t pid = 0; pid < members.size(); pid++)
            for (int v : members[pid]) part[(size_t) v] = (int) pid;
    }
    // ---- 3. reverse Cuthill-McKee inside every part
    std::vector<int> &gorder = *gorder_out;
    gorder.clear();
    if (first) first->assign(1, 0);
    gorder.reserve((size_t) ng);
    for (size_t pid = 0; pid < members.size(); pid++)
    {
        part_order(g, part, (int) pid, members[pid], true, stamp, mark, level, placed, scratch, order);
        gorder.insert(gorder.end(), order.begin(), order.end());
        if (first) first->push_back((int) gorder.size());
    }
}

}  // namespace

bool locality_reorder(int nrow, int ncol, const int *rowptr, const int *colidx, int nparts, std::vector<int> *perm,
                      LocalityInfo *info)
{
    perm->clear();
    if (nrow != ncol || nrow < 64 || rowptr[nrow] <= 0) return false;
    const long long nnz = rowptr[nrow];
    for (long long p = 0; p < nnz; p++)
        if (colidx[p] < 0 || colidx[p] >= nrow) return false;        // two-source or rectangular column space
    // ---- 1. row groups
    std::vector<int> grp((size_t) nrow);
    int ng = 0;
    for (int r = 0; r < nrow; r++)
    {
        bool same = false;
        if (r > 0)
        {
            const int la = rowptr[r] - rowptr[r - 1], lb = rowptr[r + 1] - rowptr[r];
            same = (la == lb) && lb > 0 && memcmp(colidx + rowptr[r - 1], colidx + rowptr[r], sizeof(int) * (size_t) lb) == 0;
        }
        if (!same) ng++;
        grp[(size_t) r] = ng - 1;
    }
    Graph g;
    g.n = ng;
    g.weight.assign((size_t) ng, 0);
    std::vector<int> rep((size_t) ng, 0);
    for (int r = nrow - 1; r >= 0; r--) { g.weight[(size_t) grp[(size_t) r]]++; rep[(size_t) grp[(size_t) r]] = r; }
    // ---- quotient graph, symmetrised
    std::vector<std::vector<int>> nbr((size_t) ng);
    {
        std::vector<int> last((size_t) ng, -1);
        for (int s = 0; s < ng; s++)
        {
            const int r = rep[(size_t) s];
            for (int p = rowptr[r]; p < rowptr[r + 1]; p++)
            {
                const int t = grp[(size_t) colidx[p]];
                if (t == s || last[(size_t) t] == s) continue;
                last[(size_t) t] = s;
                nbr[(size_t) s].push_back(t);
            }
        }
        std::vector<std::vector<int>> rev((size_t) ng);
        for (int s = 0; s < ng; s++)
            for (int t : nbr[(size_t) s]) rev[(size_t) t].push_back(s);
        for (int s = 0; s < ng; s++)
        {
            std::vector<int> &v = nbr[(size_t) s];
            v.insert(v.end(), rev[(size_t) s].begin(), rev[(size_t) s].end());
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        }
    }
    g.ptr.assign((size_t) ng + 1, 0);
    for (int s = 0; s < ng; s++) g.ptr[(size_t) s + 1] = g.ptr[(size_t) s] + (int) nbr[(size_t) s].size();
    g.adj.resize((size_t) g.ptr[(size_t) ng]);
    for (int s = 0; s < ng; s++) std::copy(nbr[(size_t) s].begin(), nbr[(size_t) s].end(), g.adj.begin() + g.ptr[(size_t) s]);
    nbr.clear();
    nbr.shrink_to_fit();

    // ---- 2., 3.
    std::vector<int> gorder;
    std::vector<int> first;
    slab_order(g, nparts, &gorder, &first);
    if ((int) gorder.size() != ng) return false;
    // ---- rows of the groups in that order
    std::vector<int> gstart((size_t) ng + 1, 0);
    for (int s = 0; s < ng; s++) gstart[(size_t) s + 1] = gstart[(size_t) s] + g.weight[(size_t) s];
    perm->reserve((size_t) nrow);
    for (int s : gorder)
        for (int r = rep[(size_t) s]; r < rep[(size_t) s] + g.weight[(size_t) s]; r++) perm->push_back(r);
    if (info != nullptr)
    {
        info->groups = ng;
        info->parts = (int) first.size() - 1;
        // mean distance, in the new order, between a row and the rows its columns name: what the sweep has to keep
        std::vector<int> pos((size_t) nrow);
        for (int i = 0; i < nrow; i++) pos[(size_t) (*perm)[(size_t) i]] = i;
        double before = 0.0, after = 0.0;
        for (int r = 0; r < nrow; r++)
            for (int p = rowptr[r]; p < rowptr[r + 1]; p++)
            {
                before += std::abs((double) colidx[p] - r);
                after += std::abs((double) pos[(size_t) colidx[p]] - pos[(size_t) r]);
            }
        info->mean_dist_before = before / (double) nnz;
        info->mean_dist_after = after / (double) nnz;
    }
    return true;
}

bool graph_slab_order(int n, const std::vector<int> &ptr, const std::vector<int> &adj, const std::vector<int> &weight, int nparts,
                      std::vector<int> *order, std::vector<int> *first)
{
    if (n <= 0 || (int) ptr.size() != n + 1 || (int) weight.size() != n) return false;
    Graph g;
    g.n = n;
    g.ptr = ptr;
    g.adj = adj;
    g.weight = weight;
    slab_order(g, nparts, order, first);
    return (int) order->size() == n;
}

}  // namespace crp
