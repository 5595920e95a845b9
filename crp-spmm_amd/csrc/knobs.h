// knobs.h -- every CRPSPMM_* environment variable the library honours, read ONCE (first use) into one struct.  No launch path
// calls getenv(); INTEGRATION.md section 5 documents the same list.
#pragma once

namespace crp {

struct Knobs
{
    int t2_chain;          // CRPSPMM_T2_CHAIN: teams per chain of the persistent team kernel (0, the default = one workgroup per team: measured faster, profiles/r04_chains_ab.txt)
    bool t2_latorder;      // CRPSPMM_T2_LATORDER: lattice teams: search the processing order against the L2 model (default 1; 0 = strips along the teeth)
};

const Knobs &knobs();

}  // namespace crp
