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

}  // namespace crp
