// team_order.h -- processing order of the teams of the LDS-sharing kernel by recursive graph bisection (internal).
//
// Why: a B row is fetched from beyond an XCD's L2 once per GENERATION (the workgroups resident on the XCD at one
// time, 64 teams of 64 rows) that touches it, and once per XCD.  What a launch fetches is therefore the sum, over
// the generations, of the B rows each of them names -- the surface of the generations.  Recursive bisection makes
// every aligned range of the order (a generation, an XCD's eighth) a compact piece of the matrix graph; the greedy
// super-teams it replaces left the pieces between the first-grown balls ragged (nlpkkt stand-in: 1.99 x B summed
// over the generations against 1.70 for hand-made bricks; bisection: 1.76).
#pragma once
#include <vector>

namespace crp {

// Weighted undirected graph in CSR form (both directions stored, no self loops): ptr[n + 1], adj, wgt (shared B rows).
// work[v] > 0 = what a split balances.  leaf = vertices of a generation: ranges are bisected until they hold at most
// `leaf` vertices, every left part a multiple of `leaf` (so that generations = consecutive `leaf` vertices).
// order = the vertices, leaf after leaf.  Deterministic for a given graph.
void bisection_order(int n, const std::vector<long long> &ptr, const std::vector<int> &adj, const std::vector<int> &wgt,
                     const std::vector<int> &work, int leaf, std::vector<int> *order);

}  // namespace crp
