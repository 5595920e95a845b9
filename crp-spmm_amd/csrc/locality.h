// locality.h -- processing order of the rows of A for B-row locality (internal; see locality.cpp).
#pragma once
#include <vector>

namespace crp {

struct LocalityInfo
{
    int groups = 0;                 // row groups (rows with identical column lists)
    int parts = 0;
    double mean_dist_before = 0.0;  // mean |position(col) - position(row)| over the nonzeros, natural order
    double mean_dist_after = 0.0;   // ... in the new order
};

// perm[i] = original row processed at position i.  Only for square matrices whose column indices name rows of
// the same index space (nrow == ncol, no two-source encoding); returns false (perm empty) otherwise.
// nparts = number of contiguous ranges the kernels deal to the XCDs (8).
bool locality_reorder(int nrow, int ncol, const int *rowptr, const int *colidx, int nparts, std::vector<int> *perm,
                      LocalityInfo *info = nullptr);

// The slab order of locality.cpp (steps 2 and 3) on any weighted graph (symmetric adjacency, no self loops):
// `nparts` (a power of two) parts of equal weight, reverse Cuthill-McKee inside each.  order = the vertices part
// after part; first (optional) = where every part starts, nparts + 1 entries.
bool graph_slab_order(int n, const std::vector<int> &ptr, const std::vector<int> &adj, const std::vector<int> &weight, int nparts,
                      std::vector<int> *order, std::vector<int> *first = nullptr);

}  // namespace crp
