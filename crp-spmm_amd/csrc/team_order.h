// team_order.h -- processing order of the teams of the LDS-sharing kernel (internal).
//
// Why: a B row is fetched from beyond an XCD's L2 once per GENERATION (the workgroups resident on the XCD at one
// time, 64 teams of 64 rows) that touches it, and once per XCD.  What a launch fetches is therefore, to first order,
// the sum over the generations of the B rows each of them names -- the surface of the generations.
#pragma once
#include <vector>

namespace crp {

// Processing order of LATTICE teams by search over block orders, judged by a replay against a model of an XCD's L2.
//
// Why: the lattice order of round 2 gives every XCD a strip of team columns (a, b) and sweeps it along the teeth (t) -- fine for
// the pwtk stand-in (45 team columns), thin PLATES for a 3-D stencil in natural order (fem3d stand-in: 784 columns, 98 per
// XCD; the 64 teams resident on an XCD are 64 columns at one t: 5.6 x 6 x 40 nodes), whose rows an XCD's 4 MiB cannot keep
// from one plate to the next: B arrives 2.8 times from beyond L2 in the model, 3.2 times on the hardware.  Candidates: the
// given order, and for every split of the team columns into pa x pb = 8 boxes (one per XCD) and every block shape
// (bt, ba, bb) of about a generation the order "box, block, position inside the block" (blocks with t fastest or slowest,
// teams inside a block with t fastest or slowest).  Every candidate is cut into 8 runs of equal work like the launch grid
// and two of the runs (a prefix of `sample` teams each) are replayed: `slots` teams resident, admitted in order, every
// resident team issuing one round (its row slices cols[r * W .. r * W + W), nocol = nothing) per step, against an exact LRU of
// `lru_rows` rows.  The order with the fewest misses wins; the given order is kept unless a candidate is 3 % better.
// lat = 3 ints per team (a, b, t); rounds(g, &nr) = the team's slot columns, nr * W of them.  Returns true when *order changed.
struct LatticeOrderInfo { double miss_given = 0, miss_best = 0; int pa = 0, pb = 0, bt = 0, ba = 0, bb = 0, flags = 0, candidates = 0; };
bool lattice_block_order(int nteam, const int *lat, int W, int slots, int lru_rows, int nocol,
                         const int *const *cols, const int *nrounds, std::vector<int> *order, LatticeOrderInfo *info);

}  // namespace crp
