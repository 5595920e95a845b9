// panel_format.h -- row-panel storage of A used by the register-blocked SpMM
// kernels (internal; built once at crp_csr_dev_create time, the inspector step
// the reference leaves to MKL inside mkl_sparse_d_mm,
// /root/reference/src/rowpara_spmm.c:398-408).
//
// Rows are grouped into panels of R consecutive rows.  A panel stores the
// union of its rows' column indices once ("entries"); every entry carries R
// values and an R-bit presence mask, so one load of a B row slice feeds up to
// R rows of C.  Absent (row, column) pairs are skipped through the mask -- they
// are never multiplied by zero, which keeps 0 * Inf out of the result -- and an
// entry is split when a row holds the same column twice.  Entry counts are
// padded to a multiple of PANEL_PAD with mask-0 entries (column = the panel's last
// real column, a valid address) so that the kernels' register ring needs no guards
// and fetches index / mask groups with aligned scalar loads.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#include "par.h"
#include "knobs.h"

namespace crp {

// CRPSPMM_TIMING=1: phase times of the format builders on stderr
struct PhaseClock
{
    bool on = knobs().timing;
    double t0 = now();
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec; }
    void lap(const char *what)
    {
        if (!on) return;
        const double t = now();
        fprintf(stderr, "[crpspmm timing] %-44s %8.3f s\n", what, t - t0);
        t0 = t;
    }
};


constexpr int PANEL_PAD = 8;   // = PANEL_RING of the kernels (spmm_kernels.hip)

struct PanelHost
{
    int R = 0;
    int npanel = 0;
    std::vector<int>      pptr;    // npanel + 1, entry offsets (multiples of PANEL_PAD)
    big_vector<int>       pcol;    // entries: two-source column index
    big_vector<uint32_t>  pmask4;  // entries / 4 words: byte u of word g = mask of entry 4g + u
    big_vector<double>    pval;    // entries * R, value of row r of entry q at q*R + r
    big_vector<uint32_t>  pmap;    // nnz: slot (q*R + r) in pval of CSR nonzero p (for value updates)
    std::vector<int>      porder;  // processing order of the panels (npanel positions; team schedule: 4 per team, -1 = none)
    std::vector<int>      psync;   // team schedule only: per workgroup, the rounds its waves start together
    int team_waves = 4;            // waves per workgroup the processing order is laid out for (4, or 6 under the team schedule)
    long long real_entries = 0;    // entries before padding
    double fill() const;           // nnz / (real_entries * R)
    long long nnz = 0;
    // Compact values (build_compact_values; the narrow-operand kernel on panels that are mostly holes): only the (entry, row)
    // pairs that exist, entry after entry, rows ascending.  cmo[q] = row mask of entry q | (index of its first value,
    // relative to the panel's first) << 8; cbase[panel] = the panel's first value in cval; cmap[nz] = where CSR nonzero nz sits.
    big_vector<uint32_t>  cmo;
    std::vector<long long> cbase;  // npanel + 1
    big_vector<double>    cval;
    big_vector<uint32_t>  cmap;
};
// R = 8 only.  Returns false (nothing built) when a panel holds 2^24 values or more, or the matrix 2^32.
bool build_compact_values(PanelHost *p);

// Processing order of the panels for temporal locality of B: panels are taken in groups of
// `group` consecutive panels (neighbouring rows share most of their columns, so a group keeps its
// B rows in L1 / L2 while it runs) and the groups are visited breadth-first over the "shares a B
// row" relation, starting from group 0.  For a matrix from a 3D mesh in natural order (bands at
// +-1, +-nx, +-nx*ny) this turns five temporally distant touches of every B row into touches a few
// hundred groups apart, i.e. inside the 256 MiB Infinity Cache instead of HBM; for a narrow banded
// matrix it reproduces the natural order.  Pure scheduling: which wave computes which panel.
void locality_order(const PanelHost &p, int group, std::vector<int> *order);

// Stride-lattice order for matrices whose far nonzeros sit on two nested strides D1 < D2 (a 3D mesh in
// natural order: bands at +-nx and +-nx*ny; the pwtk stand-in: 1200 and 36000).  Rows split into "teeth"
// of D1 rows; tooth (i, j) starts at row j*D2 + i*D1.  Panels at the same offset t inside their teeth
// touch the same B rows (one through its near band, its neighbours through a far band), so every XCD
// gets a block of neighbouring teeth -- consecutive i, all j -- and sweeps them in lockstep along t:
// the five temporally distant touches of a B row become touches that are in flight together on
// one XCD.  Strides are detected from the histogram of |col - row|: D1 = the nearest far cluster, D2 =
// the centre of the group of clusters beyond it (a 27-point stencil spreads the outer stride over
// nx*ny - nx, nx*ny, nx*ny + nx), each >= 6 % of the nonzeros, D2 an integer multiple of D1 within 2 %;
// returns false when the matrix does not look like that (the caller then uses locality_order()).  `chunk` = order positions per XCD.
bool stride_lattice_order(int nrow, const int *rowptr, const int *colidx, int R, int npanel, int chunk,
                          std::vector<int> *order);

// Two nested far strides (see stride_lattice_order): detection alone, and the tooth coordinates
// (i, j, offset t in panels) of a panel.
bool detect_stride_lattice(int nrow, const int *rowptr, const int *colidx, int R, double *D1, double *D2, int *M);
void lattice_coords(int panel, int R, double D1, double D2, int M, int *i, int *j, int *t);

// Teams: four panels (one per wave of a workgroup) whose B rows the workgroup loads ONCE into an LDS
// ring shared by its waves.  For a stride-lattice matrix a team is a 2 x 2 block of teeth at the
// same offset t -- the panels that read a B row through their near band and through their far
// bands sit in one team, so that row crosses L2 once instead of three times; otherwise four
// consecutive panels.  The union of the four panels' columns is stored once (tcol), with one mask
// word per union entry: byte w = row mask of wave w (0 = wave w does not use the column).  The
// order of the union entries is a balanced schedule, not column order (see build_teams); wave w
// reads the 8 values of its k-th own entry from the value stream starting at entry tvoff[4g + w]
// (tq maps every entry of the panel format to its place in the streams).
struct TeamHost
{
    int nteam = 0;
    int T = 4;                     // panels (= waves) per team: 4 (2 x 2 teeth) or 6 (3 x 2 teeth)
    bool lattice = false;
    int st = 1;                    // lattice teams of 8 / 16: consecutive panels along the teeth (the team's advance = 8 st rows)
    bool clustered = false;        // T = 8 off a lattice: panels grouped by shared columns (plocal = slot of every panel in its team)
    std::vector<int>      plocal;
    std::vector<int>      tpanel;  // T * nteam: panel of wave w, or -1
    std::vector<int>      tptr;    // nteam + 1: union entry offsets (multiples of PANEL_PAD)
    big_vector<int>       tcol;    // union entries: column index
    big_vector<uint32_t>  tmask;   // union entries: row masks of waves 0..3 (complete for T = 4 only)
    std::vector<int>      torder;  // processing order of the teams
    std::vector<long long> tvoff;  // T * nteam + 1: first entry of wave w's value stream
    std::vector<long long> tq;     // per panel-format entry: its entry index in the value streams, or -1
    big_vector<int>       tsrc;    // T per union entry: the panel-format entry of wave w behind it, or -1
    long long real_entries = 0;    // union entries before padding
    std::vector<int>      lat_key; // lattice teams: 3 per team -- team column (a, b) and position t along the teeth (team_order.h, lattice_block_order)
};
// What build_teams decides before it lays the entries out -- which panels form a team and in what order the teams are processed.
// A matrix builds up to three formats on the same teams (team2, team2r for 32 and for 64 columns): the first build fills the seed,
// the others take the teams from it (panel clustering and super-team order are 40 % of build_teams on the nlpkkt240-size matrix).
struct TeamSeed
{
    bool valid = false;
    int T = 0, np = 0;
    bool lattice = false, clustered = false;
    std::vector<int> team_of, slot_of;     // clustered teams: team and slot of every panel
    std::vector<int> torder;
};
// colpos (optional, matrices in a locality order): position of row c of A in the order the panels were built on.
// balanced = false: the union entries of a team stay in column order (the caller orders them itself).
void build_teams(const PanelHost &p, int nrow, const int *rowptr, const int *colidx, TeamHost *out, int T = 4, const int *colpos = nullptr,
                 bool balanced = true, int mix_mode = -1,       // mix_mode: panels of two kinds in one team (median-column order): -1 = by rule, 0 / 1
                 TeamSeed *seed = nullptr);

// Team schedule for the row-panel kernel itself (no LDS sharing): the entries of every panel are
// re-ordered to the order in which its wave meets them in the team's balanced schedule, and the
// processing order becomes team after team, four positions per team (-1 where a team has no
// panel for a wave).  The four waves of a workgroup then reach an entry they share after the same
// number of own entries, i.e. at nearly the same time, and the later ones find the row in L2.
void apply_team_schedule(PanelHost *p, const TeamHost &t);

// ---- team2: the streams of the LDS-sharing kernel of csrc/team2_kernel.hip ---------------------------
// A team is 8 panels on the 8 waves of one 512-thread workgroup.  (Teams of 16 panels on 16 waves, and on 8 waves with two
// accumulator banks each, were built in rounds 2 - 3, measured 2 - 9 % slower / no faster, and removed in round 4.)  The union of
// their columns is walked in rounds of up to 8 union entries ("slots"); wave w fetches slot w of a round (one B row slice) by
// LDS-DMA into a ring shared by the workgroup, TEAM2_D rounds ahead.  What a wave owns of a round is a list of at most TEAM2_CAP
// PARTS: a part is one slot together with a CONTIGUOUS range of the rows of its panel that have the column (an entry whose rows
// are not contiguous is split into several parts), so that the kernel can jump to straight-line code for the range instead of
// testing a mask bit per row.  Per (round, wave) one 16-byte record:
//   word 0 : bits 0-2 = number of parts c (0..4); bits 4+3i .. = ring slot of part i; flags from bit 16: ISSUE (r + D < rounds),
//            TAIL (r + D - 1 >= rounds), LAST round, RECS (wave 0, r % 8 == 0 and a further record block exists: fetch it now),
//            one spare; bits 21-26 = value position of part 0
//   word 1 : bits 6i .. 6i+5 = range of part i as first * 8 + len - 1; bits 24-29 = value position of part 1;
//            bits 30-31 = size class q of the value block of round r + TEAM2_D (at most 8 (q + 1) values)
//   word 2 : bits 0-19 = offset, inside the wave's value stream and in units of TEAM2_VUNIT values, of the block of
//            round r + TEAM2_D; bits 20-25, 26-31 = value positions of parts 2 and 3
//   word 3 : column (two-source encoding) of the union entry this wave fetches for round r + TEAM2_D (an empty slot: a row of
//            the team, fetched and not read)
// Records are stored in blocks of 8 rounds x 8 waves (one KiB); for the first TEAM2_D rounds the (column, value offset) pairs come
// from tpro.  A wave's values are COMPACT: a part of len rows holds len values; the parts of a round form one block of the wave's
// stream (padded to TEAM2_VUNIT values), in the order the wave meets them; part i's value position = (values of the parts before
// it in the block) + 7 - first_i, what the kernel adds to a lane's row to find its value -- or, for well-filled panels
// (Team2Host::compact = false), 8 values per part: part i's row r at 8 i + r, no position to decode.
// The stream of wave w of team g starts at value TEAM2_VUNIT * tvoff[8 g + w].
constexpr int TEAM2_T = 8;
constexpr int TEAM2_D = 3;
constexpr int TEAM2_CAP = 4;
constexpr int TEAM2_VUNIT = 4;                    // value-stream offsets count units of 4 values (16 bytes of fp32, 32 of fp64)
constexpr int TEAM2_NOCOL = (int) 0x80000000;   // builder-internal mark of an empty slot (the records name a row of the team instead)
struct Team2Host
{
    bool compact = true;             // value blocks hold only the values that exist; false: 8 values per part, part i's row r at
                                     // 8 i + r of the round's block -- the kernel then decodes no value position (set before build_team2)
    int nteam = 0;
    bool lattice = false;
    std::vector<int>       tpanel;   // 8 * nteam: panel of wave w, or -1
    std::vector<int>       torder;   // processing order of the teams
    std::vector<int>       tgrid;    // the launch grid: 8 runs of tgrid.size() / 8 entries, run x = what XCD x processes, in
                                     // order (a contiguous piece of torder; -1 = no team); the pieces carry equal ROUNDS
    std::vector<int>       tinfo;    // 4 * nteam: rounds, first record block, parts of all waves, filled slots
    std::vector<int>       tpro;     // nteam * TEAM2_D * 8 * 2: {column, value offset} wave w fetches for round d < TEAM2_D
    big_vector<uint32_t>   trec;     // record blocks: 256 words each (8 rounds x 8 waves x 4)
    std::vector<long long> tvoff;    // 8 * nteam + 1: first value unit of wave w's stream
    big_vector<double>     tval;     // the value streams (empty when the panels came without values: the caller scatters them through vmap)
    std::vector<uint32_t>  vmap;     // per CSR nonzero (panel format's own order of pmap): its slot in tval
    long long nvalues = 0;           // values in tval (blocks padded to TEAM2_VUNIT)
    long long real_entries = 0;      // union entries (filled slots)
    long long slots = 0;             // slots including the empty ones of partly filled rounds
    long long parts = 0;
};
// p must hold its entries in column order (build_panels(..., team_schedule = false)); p.pmap is consumed to
// build vmap (indexed like p.pmap: by the CSR nonzero position the panels were built from).
// colpos (optional, square matrices in a locality order): position of row c of B in the processing order of the
// rows of A, for the phase key of the union order (see build_team2); NULL = the column index itself.
// Panels built without values (build_panels(val = NULL)): tval stays empty, nvalues says how many zeros to allocate, and the values go in
// through vmap (on the device: scatter_vals_f64 from the CSR values already in HBM -- no 8-values-per-entry copy on the host at all).
void build_team2(const PanelHost &p, int nrow, const int *rowptr, const int *colidx, Team2Host *out, const int *colpos = nullptr, TeamSeed *seed = nullptr);

// ---- team2r: the streams of the row-owner team kernel (csrc/team2r_kernel.hip) -- narrow operands, panels that are mostly holes --
// Operands of at most 64 fp64 columns: a B row slice fills a quarter (n <= 32) or half (n <= 64) of a wave.  Same teams as team2 (8 panels on
// 8 waves), a ring of row slices shared by the workgroup, and the lane groups OWN rows: with G = 4 (n <= 32) lane group q of a wave accumulates rows q
// and q + 4 of the wave's panel (G = 2, n <= 64: group q owns rows q, q + 2, q + 4, q + 6), and a STEP gives every row its next
// nonzero of the round: an LDS byte offset of the B row slice inside the ring set + the value.  No row masks, no EXEC games: a row
// that has run out of nonzeros in this round gets the value 0.0 and the offset of a slice of zeros (TEAM2R_ZERO) -- so an absent
// (row, column) pair still is never multiplied with a B entry.  ~1.1 instructions per nonzero where the masked-row step of the
// narrow kernels spends 57 per four panel entries = 7.7 per nonzero when an entry holds 1.84 of 8 rows (nlpkkt).
// A round has 16 KiB of slices: S = 16 G slots of 1024 / G bytes; wave w fetches slots 2 G w .. 2 G (w + 1) - 1 with
// TEAM2R_ROWDMA = 2 DMA instructions.  What a wave owns of a round: Lp steps (a multiple of 2, at most TEAM2R_LCAP: the scheduler closes a round before a
// row would pass it), stored as a BLOCK of the wave's stream: [8 rows][Lp] values (doubles), [8 rows][Lp] offsets (uint16), and a
// 64-byte HEADER = the wave's record of round r + 2 of the same team (zeros past the team's last round): the kernel issues the DMAs
// of round r + 2 while it consumes round r, and finds what to fetch in the block that has just landed -- no load on its path.
// Record of (round, wave), 16 words: [0] Lp; [1] first 16-byte unit of the block inside the wave's stream; [2 .. 2 + 2 G) columns of
// the slots the wave fetches for THIS round (two-source encoding; an empty slot names a row of the team).  Records of a team:
// trec[(tinfo[2 g + 1] + r) * 128 + w * 16]; wave w's stream starts at byte 16 * tvoff[8 g + w] of tval.
// tent: what a (persistent) workgroup needs when it turns to the team at entry e of the launch grid, per wave, 32 words at
// tent[(e * 8 + w) * 32]: [0] rounds (0 = no team: the run ends), [1] panel, [2], [3] tvoff (low, high), [4 .. 14) record of round 0,
// [14 .. 24) record of round 1 (its first 10 words), [24 .. 32) the C rows of the panel's 8 rows (filled on the device from the row map).
constexpr int TEAM2R_LCAP = 12;
// (Half rounds -- 8 KiB ring sets, one row DMA per wave and round, three workgroups per CU -- were measured 33 % slower in round 3
//  and removed: what a round costs beside its FMAs does not halve with its slots.)  The zero slice sits behind the slots of a set.
constexpr int TEAM2R_ROWDMA = 2;
inline int team2r_zero(int rowdma) { return 8192 * rowdma; }
struct Team2RHost
{
    int G = 4;
    int nteam = 0;
    bool lattice = false;
    std::vector<int>       tpanel, torder, tgrid, tinfo;
    big_vector<uint32_t>   trec;     // 128 words per round
    big_vector<uint32_t>   tent;     // 256 words per entry of tgrid
    std::vector<long long> tvoff;    // 8 * nteam + 1, units of 16 bytes
    big_vector<double>     tval;     // the streams, as 8-byte words (panels without values: the value words stay 0.0, offsets and headers are written)
    std::vector<uint32_t>  vmap;     // per CSR nonzero: its 8-byte word in tval
    long long nwords = 0, rounds = 0, steps = 0, nnz = 0, slots_filled = 0;   // steps = sum of Lp over (round, wave)
};
// false: the streams would pass what their 32-bit offsets address (34 GB); nothing usable in *out then
bool build_team2r(const PanelHost &p, int nrow, const int *rowptr, const int *colidx, Team2RHost *out, const int *colpos = nullptr, TeamSeed *seed = nullptr);

// Number of panel entries (before padding) a given R would need: cheap pass used
// to pick R.  colidx may carry the two-source encoding.
long long count_panel_entries(int nrow, const int *rowptr, const int *colidx, int R);

// Distinct columns of every block of `block` consecutive rows, summed over the blocks: what a team of block / 8
// consecutive panels would fetch (cheap pass used to decide whether the LDS-sharing kernel pays).
long long count_block_union(int nrow, const int *rowptr, const int *colidx, int block);

// team_schedule = false keeps the entries of every panel in column order whatever CRPSPMM_PANEL_ORDER
// says (the team format is built on that order).  need_order = false skips the processing order (porder stays empty):
// the team formats bring their own.  val = NULL: structure only (pval stays empty; pmap says where every nonzero's value belongs).
void build_panels(int nrow, const int *rowptr, const int *colidx, const double *val, int R, PanelHost *out,
                  bool team_schedule = true, bool need_order = true);

}  // namespace crp
