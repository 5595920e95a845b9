// knobs.h -- every CRPSPMM_* environment variable the library honours, read ONCE (first use) into one struct: no format builder
// and no launch path calls getenv().  INTEGRATION.md section 5 documents the same list.  (The reference's own variables --
// RP_SPMM_P2P, RP_SPMM_REIDX, A2A_B_FINEGRAIN -- are read where the reference reads them, at engine init, with its message.)
// CRPSPMM_KNOBS_LIVE=1 (the test suite sets it): the environment is re-read on every use, so that a test can change a knob
// between two calls of one process.
#pragma once

namespace crp {

struct Knobs
{
    bool   timing;            // CRPSPMM_TIMING=1: phase times of the format builders on stderr
    int    num_threads;       // CRPSPMM_NUM_THREADS (else OMP_NUM_THREADS): host threads of the builders; 0 = the CPUs the process may use
    bool   sync_release;      // CRPSPMM_SYNC_RELEASE=1: builders free their temporaries before returning (leak checkers), not in the background
    int    spmm_variant;      // CRPSPMM_SPMM_VARIANT=1|2|3: what variant 0 resolves to below the team kernel's widths (0 = by the matrix)
    int    reorder;           // CRPSPMM_REORDER=0|1: locality order of the derived formats never / whenever the matrix qualifies (-1 = by gain)
    int    panel_order;       // CRPSPMM_PANEL_ORDER=0..3: row-panel processing order natural / breadth-first / lattice / team schedule (-1 = auto)
    int    narrow_max;        // CRPSPMM_NARROW_MAX=32|64: widest operand of the narrow row-panel kernel (0 = 32, or 64 on compact values)
    int    team2_compact;     // CRPSPMM_TEAM2_COMPACT=0|1: value blocks of the team kernel with 8 values per part / compact (-1 = by panel fill)
    int    team2r;            // CRPSPMM_TEAM2R=0|1: the row-owner team kernel at 24..64 columns never / whenever applicable (-1 = by panel fill)
    bool   t2_latorder;       // CRPSPMM_T2_LATORDER=0: lattice teams keep the strips-along-the-teeth order (default 1: search against the L2 model)
    int    overlap;           // CRPSPMM_OVERLAP=0: no interior / boundary split of the rows (exchange not overlapped)
    bool   exchange_host;     // CRPSPMM_EXCHANGE=host: device payloads staged through the host (ranks that share a GPU; rehearsal)
    bool   replicate_host;    // CRPSPMM_REPLICATE=host: the A panel is replicated with the host all-gather
    double rccl_timeout;      // CRPSPMM_RCCL_TIMEOUT: seconds a non-blocking communicator may take to come up (120)
    bool   rccl_blocking;     // CRPSPMM_RCCL_BLOCKING=1: classic blocking ncclCommInitRank
    int    engine_a_static;   // CRPSPMM_ENGINE_A_STATIC=0|1: crpspmm_engine re-sends A's values on every exec / never (-1 = unset)
};

const Knobs &knobs();

}  // namespace crp
