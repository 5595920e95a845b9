// hip_api.hip -- implementation of include/crpspmm_hip.h (device-level C ABI).
// Replaces the host/CUDA shims of /root/reference/deprecated/src/cuda_proxy.cu:53-182
// with HIP-only code for gfx950; there is no CPU fallback in this file.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <memory>
#include <algorithm>
#include <vector>
#include "crpspmm_hip.h"
#include "kernels.h"
#include "panel_format.h"
#include "locality.h"
#include "knobs.h"

// device copy of one row-panel format (panel_format.h)
struct PanelDev
{
    bool      built = false;
    int       R = 0, npanel = 0;
    int      *pptr = nullptr, *pcol = nullptr, *porder = nullptr;
    int       norder = 0, team_waves = 4;
    int      *psync = nullptr;
    uint32_t *pmask4 = nullptr, *pmap = nullptr;
    double   *pval = nullptr;
    double    fill = 0.0;
    long long entries = 0;
    // compact values for the narrow-operand kernel (R = 8 panels that are mostly holes; PanelHost::cmo / cbase / cval / cmap)
    uint32_t *cmo = nullptr, *cmap = nullptr;
    long long *cbase = nullptr;
    double   *cval = nullptr;
    long long cvalues = 0;
};

// fp64 columns from which auto picks variant 5.  Round 4: the one-piece instances (n <= 128) run THREE workgroups per CU (80
// VGPRs, 43 KiB of LDS) -- pwtk stand-in n = 128 0.181 -> 0.163 ms, nlpkkt stand-in n = 96 1.18 -> 0.98 -- and the crossover against
// the row-panel kernel moved down (profiles/r04_team2_min_n.txt, variant 3 / 5: pwtk stand-in n = 80 0.138 / 0.146, n = 96 0.155 /
// 0.151, n = 112 0.167 / 0.154; shell n = 64 0.120 / 0.119, n = 96 0.166 / 0.137; Queen stand-in n = 64 0.447 / 0.403, n = 96 0.531 / 0.444).
constexpr int TEAM2_MIN_N = 96;
// ... and for a FULL half-piece tile (61 .. 64 columns) since the value blocks are compact (round 4, item 7): the half-piece instance takes the
// same time from 48 to 64 columns, the row-panel kernel's grows with them -- pwtk stand-in, variant 3 / 5: n = 48 0.0929 / 0.1093 ms, n = 64
// 0.1134 / 0.1096, n = 80 0.1389 / 0.1378, n = 96 0.1546 / 0.1547 (profiles/r04_compact_ab.txt).
constexpr int TEAM2_HALF_FULL_LO = 61, TEAM2_HALF_FULL_HI = 64;
// ... where the row-panel format asks for more than 12 B row slices per row of A (Queen stand-in: 17.3; pwtk: 10.4), or the matrix has
// no stride lattice (the row-panel kernel then runs without its team schedule).
// From 48 columns since the HALF-piece instances (operands of at most 64 fp64 / 128 fp32 columns: 8 bytes per lane, one FMA per row
// and part, four workgroups per CU; profiles/r04_half_piece_instances.txt, variant 3 / 5: Queen stand-in n = 48 0.424 / 0.339 ms,
// n = 64 0.451 / 0.361; shell n = 64 0.124 / 0.108; pwtk stand-in n = 64 0.119 / 0.126: stays with the row-panel kernel).
constexpr int TEAM2_MIN_N_NOLATTICE = 48;
// ... dense row-panel formats from 33 columns: the half-piece instance takes the same time from 34 to 64 columns (Queen stand-in, variant 3 / 5,
// compact values: n = 34 0.409 / 0.323 ms, n = 40 0.404 / 0.326; the shell stand-in, no lattice and 9 slices per row: 0.097 / 0.099 at both -- it
// keeps 48); at 32 columns the narrow kernel is ahead (0.235 / 0.314).
constexpr int TEAM2_MIN_N_DENSE = 33;
// ... when fewer than 35 % of the (row, entry) pairs of the R = 8 panels are present (KKT systems): the row-panel format then stores mostly
// zeros (8 values per entry) while the team kernel's value streams are compact.  From 33 columns since the half-piece instances and three
// workgroups per CU (round 4): up to 32 columns the row-owner team kernel (variant 7, four rows' slices per wave instruction) is 2 x ahead;
// above, variant 3 / 5 / 7 on the nlpkkt stand-in (profiles/r04_team2r_probes.txt): n = 34 0.971 / 0.784 / 0.833 ms, n = 48 1.016 / 0.801 /
// 0.834, n = 64 1.096 / 0.849 / 0.839, n = 72 1.148 / 0.930 / -; at nlpkkt240 size variant 5 / 7: n = 40 12.81 / 13.53, n = 48 13.27 / 13.35,
// n = 56 13.83 / 13.49, n = 64 13.72 / 13.91 -- the two-rows-per-lane-group format of variant 7 (n <= 64) buys nothing that the team format
// the matrix has anyway does not, and costs 4 s of build and 4 GB of HBM at that size: variant 0 no longer takes it (it was 80 in round 3).
constexpr int TEAM2_MIN_N_SPARSE = 33;
// fp32: the only other fp32 kernel is the CSR row-group one.  Since the half-piece instances (at most 128 fp32 columns: one time from 32 to
// 128) the team kernel is level or ahead from 32 columns -- row-group / team, ms (profiles/r04_compact_ab.txt, fp32 block): Queen stand-in n = 24
// 0.377 / 0.311, 32 0.385 / 0.310, 48 0.643 / 0.312, 64 0.662 / 0.316; shell 32 0.116 / 0.097, 48 0.177 / 0.098; pwtk stand-in 24 0.102 / 0.107, 32 0.107 /
// 0.108, 48 0.167 / 0.108 -- except on mostly-hole panels, where a part is 1.8 rows: nlpkkt stand-in 32 0.385 / 0.720, 48 0.664 / 0.734, 64 0.706 /
// 0.742, 96 1.301 / 0.789, 128 1.365 / 0.842 (it was 64 for every matrix).
constexpr int TEAM2_MIN_N_F32 = 32, TEAM2_MIN_N_F32_SPARSE = 65;
struct Team2Dev
{
    bool built = false;
    int  nteam = 0;
    int ngrid = 0;                 // entries of torder (= the launch grid, 8 equal runs, -1 = no team)
    bool compact = true;           // Team2Host::compact
    int *torder = nullptr, *tpanel = nullptr, *tinfo = nullptr, *tpro = nullptr;
    uint32_t *trec = nullptr;
    long long *tvoff = nullptr;
    double   *tval = nullptr;
    uint32_t *tmap = nullptr;      // per CSR nonzero: its slot in tval (value updates)
    float    *tval32 = nullptr;    // fp32 copy of the value groups (fp32 path), built on first use
    long long entries = 0, value_entries = 0;
    bool lattice = false;
};

struct Team2RDev              // panel_format.h, Team2RHost: the row-owner team kernel's streams (variant 7)
{
    bool built = false;
    int G = 4, nteam = 0, ngrid = 0;
    int *tgrid = nullptr, *tpanel = nullptr, *tinfo = nullptr;
    uint32_t *trec = nullptr;
    long long *tvoff = nullptr;
    double *tval = nullptr;
    uint32_t *tmap = nullptr;
    uint32_t *tent = nullptr;      // the entry table (Team2RHost::tent); its C rows are filled for rows_epoch
    long long rows_epoch = -1;
    const int *rows_map = nullptr;
    bool refused = false;          // the streams of this matrix would pass their 32-bit offsets: remembered, not rebuilt per product
    long long value_entries = 0;
    bool lattice = false;
};

struct crp_csr_dev
{
    int       nrow = 0;
    int       ncol = 0;
    long long nnz = 0;
    int      *rowptr = nullptr;
    int      *colidx = nullptr;
    double   *val = nullptr;
    // host copy kept for building further formats on demand
    std::vector<int>    h_rowptr, h_colidx;
    std::vector<double> h_val;
    PanelDev pan[2];          // [0]: R = 4, [1]: R = 8
    Team2Dev team2;           // teams of eight R = 8 panels, LDS-shared B rows (variant 5)
    Team2RDev team2r[2];      // the row-owner team kernel's streams (variant 7): [0] n <= 32 (G = 4), [1] n <= 64 (G = 2); tvoff in units of 16 bytes
    int      auto_variant = 1; // what variant 0 resolves to below 96 columns (1 rowgroup, 2 panel R4, 3 panel R8)
    long long rowmap_epoch = 0;    // bumped by crp_csr_dev_set_rowmap: the team2r entry tables hold C rows
    bool     team2r_pays = false;  // narrow operands (24 .. 64 columns): the row-owner team kernel beats the row-panel kernels (panels mostly holes)
    bool     team2_pays = false;   // 64 consecutive rows (in format order) share columns: variant 0 takes team2 from team2_min_n columns on
    int      last_variant = 0;     // what the last product launched (crp_csr_dev_last_variant)
    int      team2_min_n = TEAM2_MIN_N;    // or TEAM2_MIN_N_SPARSE when the R = 8 panels are mostly holes
    long long b0_rows = 0, b1_rows = 0;   // 1 + largest local / receive-buffer row a column index addresses
    float    *val32 = nullptr;            // fp32 copy of val (fp32 path), built on first use
    int      *rowmap = nullptr;           // row-subset matrices: C row of every row (device), else nullptr
    // locality order (locality.h): the derived formats (panels, teams) are built on the rows in processing order
    // perm[i] = original row at position i; f_* = that CSR, f_nz[p'] = original position of its nonzero p';
    // rowmap_fmt = C row of every position (the caller's row map composed with perm).  Empty perm = natural order.
    std::vector<int>      perm, f_rowptr, f_colidx;
    std::vector<double>   f_val;
    std::vector<uint32_t> f_nz;
    std::vector<int>      h_rowmap;       // host copy of the caller's row map (empty: none)
    int      *rowmap_fmt = nullptr;
    int       c_nrow = 0;                 // rows of C the product writes into (nrow without a rowmap)
    // crp_csr_dev_update_values() with a DEVICE pointer leaves the host copies (h_val / f_val) behind: formats built
    // afterwards take their values from the device CSR through their fresh slot maps (refresh_values_after_build)
    bool      host_vals_stale = false;
    // The R = 8 panels in column order WITHOUT values (pcol, masks, slot map: 7 bytes per nonzero at fill 0.23) and the teams built on
    // them: shared by the team formats of this matrix (team2, team2r for <= 32 and <= 64 columns) -- a further operand width costs the
    // streams of its format, not the panels and the clustering again.  Dropped once the two that variant 0 uses exist.
    std::unique_ptr<crp::PanelHost> skel8;
    crp::TeamSeed seed8;
};

static const int *fmt_rowptr(const crp_csr_dev *A) { return A->perm.empty() ? A->h_rowptr.data() : A->f_rowptr.data(); }
static const int *fmt_colidx(const crp_csr_dev *A) { return A->perm.empty() ? A->h_colidx.data() : A->f_colidx.data(); }
static const double *fmt_val(const crp_csr_dev *A) { return A->perm.empty() ? A->h_val.data() : A->f_val.data(); }
// slot map of a derived format, built on the processing order, re-indexed by the caller's nonzero positions
static void fmt_slotmap_to_caller(const crp_csr_dev *A, crp::big_vector<uint32_t> *pmap)
{
    if (A->perm.empty()) return;
    crp::big_vector<uint32_t> out(pmap->size());
    for (size_t pz = 0; pz < pmap->size(); pz++) out[(size_t) A->f_nz[pz]] = (*pmap)[pz];
    pmap->swap(out);
}

// the shared structure-only panels (see crp_csr_dev::skel8)
static const crp::PanelHost &panel_skeleton(crp_csr_dev *A)
{
    if (!A->skel8)
    {
        A->skel8.reset(new crp::PanelHost);
        crp::build_panels(A->nrow, fmt_rowptr(A), fmt_colidx(A), nullptr, 8, A->skel8.get(), false, false);
        fmt_slotmap_to_caller(A, &A->skel8->pmap);
    }
    return *A->skel8;
}
static void drop_skeleton_when_done(crp_csr_dev *A)
{
    // (the formats variant 0 can ask for: the team format and the row-owner format of n <= 32; an explicit variant 7 at 33 .. 64 columns
    //  afterwards builds its panels and teams again)
    const bool r0 = A->team2r[0].built || A->team2r[0].refused;
    if (A->team2.built && r0)
    {
        A->skel8.reset();
        A->seed8 = crp::TeamSeed();
    }
}

#define CRP_TRY(expr)                                 \
    do                                                \
    {                                                 \
        hipError_t e__ = (expr);                      \
        if (e__ != hipSuccess) return (int) e__;      \
    } while (0)

// Build (once) and upload the row-panel format with R = 4 (idx 0) or 8 (idx 1). Blocking.
static int ensure_panel(crp_csr_dev *A, int idx, hipStream_t stream)
{
    PanelDev &d = A->pan[idx];
    if (d.built) return 0;
    crp::PanelHost h;
    crp::build_panels(A->nrow, fmt_rowptr(A), fmt_colidx(A), fmt_val(A), idx == 0 ? 4 : 8, &h);
    fmt_slotmap_to_caller(A, &h.pmap);
    d.R = h.R;
    d.npanel = h.npanel;
    d.norder = (int) h.porder.size();
    d.team_waves = h.team_waves;
    d.fill = h.fill();
    d.entries = (long long) h.pcol.size();
    hipError_t e = hipMalloc((void **) &d.pptr, sizeof(int) * h.pptr.size());
    if (e == hipSuccess) e = hipMalloc((void **) &d.pcol, sizeof(int) * (h.pcol.size() + 64));
    if (e == hipSuccess) e = hipMalloc((void **) &d.pmask4, sizeof(uint32_t) * (h.pmask4.size() + 16));
    if (e == hipSuccess) e = hipMalloc((void **) &d.pval, sizeof(double) * (h.pval.size() + 512));
    if (e == hipSuccess) e = hipMalloc((void **) &d.pmap, sizeof(uint32_t) * (h.pmap.size() + 1));
    if (e == hipSuccess && !h.pmap.empty())
        e = hipMemcpy(d.pmap, h.pmap.data(), sizeof(uint32_t) * h.pmap.size(), hipMemcpyHostToDevice);
    // processing order as 16-byte records {panel (-1: none), first entry, rounds, 0}, one per position
    {
        std::vector<int> rec(4 * (h.porder.size() + 4), 0);
        for (size_t i = 0; i < h.porder.size(); i++)
        {
            const int pnl = h.porder[i];
            rec[4 * i] = pnl;
            if (pnl >= 0)
            {
                rec[4 * i + 1] = h.pptr[(size_t) pnl];
                rec[4 * i + 2] = (h.pptr[(size_t) pnl + 1] - h.pptr[(size_t) pnl]) / crp::PANEL_PAD;
            }
        }
        for (size_t i = h.porder.size(); i < h.porder.size() + 4; i++) rec[4 * i] = -1;
        if (e == hipSuccess) e = hipMalloc((void **) &d.porder, sizeof(int) * rec.size());
        if (e == hipSuccess) e = hipMemcpy(d.porder, rec.data(), sizeof(int) * rec.size(), hipMemcpyHostToDevice);
    }
    // team schedule: the waves of a workgroup start their rounds together
    if (e == hipSuccess && !h.psync.empty())
    {
        e = hipMalloc((void **) &d.psync, sizeof(int) * (h.psync.size() + 8));
        if (e == hipSuccess) e = hipMemset(d.psync, 0, sizeof(int) * (h.psync.size() + 8));
        if (e == hipSuccess) e = hipMemcpy(d.psync, h.psync.data(), sizeof(int) * h.psync.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(d.pptr, h.pptr.data(), sizeof(int) * h.pptr.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && !h.pcol.empty())
    {
        // the kernels read column indices up to three rounds past a panel: the tail of the array
        // repeats the last real column (an addressable row), never an arbitrary value
        std::vector<int> padded(h.pcol.begin(), h.pcol.end());
        padded.resize(h.pcol.size() + 64, h.pcol.back());
        e = hipMemcpy(d.pcol, padded.data(), sizeof(int) * padded.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(d.pmask4, h.pmask4.data(), sizeof(uint32_t) * h.pmask4.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && !h.pval.empty())
        e = hipMemcpy(d.pval, h.pval.data(), sizeof(double) * h.pval.size(), hipMemcpyHostToDevice);
    // Compact values for the narrow-operand kernel when under 60 % of the panels' (row, entry) pairs exist (n = 32, compact
    // against full values: nlpkkt stand-in, fill 0.23: 0.527 against 0.599 ms; Queen stand-in, 0.57: 0.243 against 0.260; pwtk
    // stand-in, 0.61: 0.068 against 0.067)
    {
        const bool want = idx == 1 && d.fill < 0.6;
        if (e == hipSuccess && want && A->nnz > 0 && crp::build_compact_values(&h))
        {
            d.cvalues = h.cbase.back();          // (cmap is derived from pmap, which is indexed by the caller's nonzeros already)
            e = hipMalloc((void **) &d.cmo, sizeof(uint32_t) * (h.cmo.size() + 64));
            if (e == hipSuccess) e = hipMemset(d.cmo, 0, sizeof(uint32_t) * (h.cmo.size() + 64));
            if (e == hipSuccess) e = hipMemcpy(d.cmo, h.cmo.data(), sizeof(uint32_t) * h.cmo.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void **) &d.cbase, sizeof(long long) * h.cbase.size());
            if (e == hipSuccess) e = hipMemcpy(d.cbase, h.cbase.data(), sizeof(long long) * h.cbase.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void **) &d.cval, sizeof(double) * h.cval.size());
            if (e == hipSuccess) e = hipMemcpy(d.cval, h.cval.data(), sizeof(double) * h.cval.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void **) &d.cmap, sizeof(uint32_t) * (h.cmap.size() + 1));
            if (e == hipSuccess && !h.cmap.empty()) e = hipMemcpy(d.cmap, h.cmap.data(), sizeof(uint32_t) * h.cmap.size(), hipMemcpyHostToDevice);
        }
    }
    if (e != hipSuccess) return (int) e;
    if (A->host_vals_stale && A->nnz > 0)
    {
        CRP_TRY(crp::scatter_vals_f64(A->nnz, d.pmap, A->val, d.pval, stream));
        if (d.cmap) CRP_TRY(crp::scatter_vals_f64(A->nnz, d.cmap, A->val, d.cval, stream));
    }
    d.built = true;
    return 0;
}

// Build (once) and upload the team2 streams on top of the R = 8 panels (column-ordered entries). Blocking.
static int ensure_team2(crp_csr_dev *A, hipStream_t stream, bool for_f32 = false)
{
    Team2Dev &t = A->team2;
    if (t.built) return 0;
    crp::PhaseClock clk;
    const crp::PanelHost &h = panel_skeleton(A);
    clk.lap("ensure_team2: build_panels (R = 8, structure only)");
    crp::released_async<crp::Team2Host> th_owner;
    crp::Team2Host &th = *th_owner;
    // Value blocks: compact (only the values that exist), or 8 per part -- the kernel instance for full groups decodes no value position
    // (two instructions per part and three per round fewer) and streams up to 64 % more value bytes.  Mostly-hole panels (under 40 % of the
    // (row, entry) pairs exist: KKT systems) are always compact: nlpkkt stand-in 1.91 against 2.05 ms, 13 GB smaller at nlpkkt240 size.  On
    // filled panels it used to be a wash that full groups won by 1 %; since the round-4 loop (fewer scalar instructions per round) the
    // kernels run at the speed of their memory schedule and the bytes decide -- fp64, compact against full, same box
    // (profiles/r04_compact_ab.txt): pwtk stand-in n = 256 0.2724 / 0.2776 ms, n = 1024 1.077 / 1.104, n = 128 0.1687 / 0.1697, shell n = 128
    // 0.1432 / 0.1481, n = 64 0.0986 / 0.1024, Queen stand-in n = 256 0.7917 / 0.8066, n = 64 0.3472 / 0.3581, n = 1024 3.170 / 3.161.  In
    // fp32 a value is 4 bytes and the decoding costs the same: full groups stay 0.3 - 0.8 % ahead (Queen stand-in n = 128 / 256 / 1024), so a
    // format that is first built for the fp32 path keeps them.  CRPSPMM_TEAM2_COMPACT=0|1 forces.
    th.compact = crp::knobs().team2_compact >= 0 ? crp::knobs().team2_compact != 0 : (h.fill() < 0.4 || !for_f32);
    std::vector<int> colpos;                    // position of every row in the processing order (square, re-ordered matrices)
    if (!A->perm.empty())
    {
        colpos.resize(A->perm.size());
        for (size_t i = 0; i < A->perm.size(); i++) colpos[(size_t) A->perm[i]] = (int) i;
    }
    crp::build_team2(h, A->nrow, fmt_rowptr(A), fmt_colidx(A), &th, colpos.empty() ? nullptr : colpos.data(), &A->seed8);
    clk.lap("ensure_team2: build_team2");
    t.nteam = th.nteam;
    t.entries = th.real_entries;
    t.lattice = th.lattice;
    auto up = [](void **dst, const void *src, size_t bytes, size_t pad) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes + pad);
        if (e == hipSuccess && pad) e = hipMemset((char *) *dst + bytes, 0, pad);
        if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    t.ngrid = (int) th.tgrid.size();
    hipError_t e = up((void **) &t.torder, th.tgrid.data(), sizeof(int) * th.tgrid.size(), 4);      // the launch grid (panel_format.h)
    if (e == hipSuccess) e = up((void **) &t.tpanel, th.tpanel.data(), sizeof(int) * th.tpanel.size(), 4);
    if (e == hipSuccess) e = up((void **) &t.tinfo, th.tinfo.data(), sizeof(int) * th.tinfo.size(), 16);
    if (e == hipSuccess) e = up((void **) &t.tpro, th.tpro.data(), sizeof(int) * th.tpro.size(), 8);
    if (e == hipSuccess) e = up((void **) &t.trec, th.trec.data(), sizeof(uint32_t) * th.trec.size(), 1024);
    t.value_entries = th.nvalues;          // values of the streams
    t.compact = th.compact;
    if (e == hipSuccess) e = up((void **) &t.tvoff, th.tvoff.data(), sizeof(long long) * th.tvoff.size(), 8);
    // the kernel requests 256 bytes per wave and round: up to four groups past a wave's last part.  The streams start as zeros in
    // HBM and take their values from the device CSR through the slot map: no copy of them is ever made on the host.
    if (e == hipSuccess) e = hipMalloc((void **) &t.tval, sizeof(double) * (size_t) th.nvalues + 4096);
    if (e == hipSuccess) e = hipMemsetAsync(t.tval, 0, sizeof(double) * (size_t) th.nvalues + 4096, stream);
    if (e == hipSuccess) e = up((void **) &t.tmap, th.vmap.data(), sizeof(uint32_t) * th.vmap.size(), 4);
    if (e != hipSuccess) return (int) e;
    if (A->nnz > 0) CRP_TRY(crp::scatter_vals_f64(A->nnz, t.tmap, A->val, t.tval, stream));
    CRP_TRY(hipStreamSynchronize(stream));
    clk.lap("ensure_team2: upload");
    t.built = true;
    drop_skeleton_when_done(A);
    return 0;
}

static void team2_args(const Team2Dev &d, crp::Team2Args *t)
{
    t->nteam = d.nteam; t->ngrid = d.ngrid; t->compact = d.compact; t->torder = d.torder; t->tpanel = d.tpanel;
    t->tinfo = d.tinfo; t->tpro = d.tpro; t->trec = d.trec; t->tvoff = d.tvoff; t->tval = d.tval; t->tval32 = d.tval32;
}

static int ensure_team2r(crp_csr_dev *A, hipStream_t stream, int G)
{
    Team2RDev &t = A->team2r[G == 2 ? 1 : 0];
    if (t.built) return 0;
    if (t.refused) return -6;
    // a cheap bound first: a wave's block takes at least 10 bytes per nonzero (value + offset), and the streams address 34 GB
    if ((double) A->nnz * 10.0 > 34.0e9) { t.refused = true; return -6; }
    crp::PhaseClock clk;
    const crp::PanelHost &h = panel_skeleton(A);
    clk.lap("ensure_team2r: build_panels (R = 8, structure only)");
    crp::released_async<crp::Team2RHost> th_owner;
    crp::Team2RHost &th = *th_owner;
    th.G = G == 2 ? 2 : 4;
    std::vector<int> colpos;
    if (!A->perm.empty())
    {
        colpos.resize(A->perm.size());
        for (size_t i = 0; i < A->perm.size(); i++) colpos[(size_t) A->perm[i]] = (int) i;
    }
    if (!crp::build_team2r(h, A->nrow, fmt_rowptr(A), fmt_colidx(A), &th, colpos.empty() ? nullptr : colpos.data(), &A->seed8))
    {
        t.refused = true;                   // too large for this format
        drop_skeleton_when_done(A);
        return -6;
    }
    clk.lap("ensure_team2r: build_team2r");
    t.G = th.G;
    t.nteam = th.nteam;
    t.lattice = th.lattice;
    t.ngrid = (int) th.tgrid.size();
    t.value_entries = th.nwords;
    auto up = [](void **dst, const void *src, size_t bytes, size_t pad) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes + pad);
        if (e == hipSuccess && pad) e = hipMemset((char *) *dst + bytes, 0, pad);
        if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up((void **) &t.tgrid, th.tgrid.data(), sizeof(int) * th.tgrid.size(), 4);
    if (e == hipSuccess) e = up((void **) &t.tpanel, th.tpanel.data(), sizeof(int) * th.tpanel.size(), 4);
    if (e == hipSuccess) e = up((void **) &t.tinfo, th.tinfo.data(), sizeof(int) * th.tinfo.size(), 16);
    if (e == hipSuccess) e = up((void **) &t.trec, th.trec.data(), sizeof(uint32_t) * th.trec.size(), 1024);
    if (e == hipSuccess) e = up((void **) &t.tvoff, th.tvoff.data(), sizeof(long long) * th.tvoff.size(), 8);
    // (the streams: values and uint16 offsets; a wave's DMAs take whole 16-byte lanes of its block)
    if (e == hipSuccess) e = up((void **) &t.tval, th.tval.data(), sizeof(double) * th.tval.size(), 4096);
    if (e == hipSuccess) e = up((void **) &t.tmap, th.vmap.data(), sizeof(uint32_t) * th.vmap.size(), 4);
    if (e == hipSuccess) e = up((void **) &t.tent, th.tent.data(), sizeof(uint32_t) * th.tent.size(), 1024);
    t.rows_epoch = -1;
    if (e != hipSuccess) return (int) e;
    // (the streams were uploaded with their offsets and headers and 0.0 for every value: the values come from the device CSR)
    if (A->nnz > 0) CRP_TRY(crp::scatter_vals_f64(A->nnz, t.tmap, A->val, t.tval, stream));
    CRP_TRY(hipStreamSynchronize(stream));
    clk.lap("ensure_team2r: upload");
    t.built = true;
    drop_skeleton_when_done(A);
    return 0;
}

// the widths at which variant 0 takes the team kernel on this matrix (fp64): from its class's threshold on, and a full half-piece tile
static bool team2_width(const crp_csr_dev *A, int n)
{
    return n >= A->team2_min_n || (A->team2_min_n == TEAM2_MIN_N && n >= TEAM2_HALF_FULL_LO && n <= TEAM2_HALF_FULL_HI);
}

// variant 0 on narrow operands: the row-owner team kernel where the R = 8 panels are mostly holes (CRPSPMM_TEAM2R=0|1 forces)
static bool team2r_auto(const crp_csr_dev *A)
{
    if (crp::knobs().team2r >= 0) return crp::knobs().team2r != 0;
    return A->team2r_pays;          // panels that are mostly holes (set at create)
}

extern "C" {

const char *crp_hip_version(void) { return "crpspmm-hip 0.1 gfx950"; }

int crp_hip_device_count(int *count)
{
    if (count == NULL) return -1;
    *count = 0;
    CRP_TRY(hipGetDeviceCount(count));
    return 0;
}

int crp_hip_set_device(int dev) { CRP_TRY(hipSetDevice(dev)); return 0; }
int crp_hip_get_device(int *dev) { if (!dev) return -1; CRP_TRY(hipGetDevice(dev)); return 0; }

int crp_hip_device_info(int dev, char *name, int *cu_count, size_t *hbm_bytes)
{
    hipDeviceProp_t prop;
    CRP_TRY(hipGetDeviceProperties(&prop, dev));
    if (name)
    {
        snprintf(name, 256, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return 0;
}

int crp_hip_device_bus_id(char *out, size_t len)
{
    if (out == NULL || len < 16) return -1;
    int dev = 0;
    CRP_TRY(hipGetDevice(&dev));
    CRP_TRY(hipDeviceGetPCIBusId(out, (int) len, dev));
    return 0;
}

int crp_dev_malloc(void **ptr, size_t bytes)
{
    if (ptr == NULL) return -1;
    *ptr = NULL;
    if (bytes == 0) return 0;
    CRP_TRY(hipMalloc(ptr, bytes));
    return 0;
}

int crp_dev_free(void *ptr)
{
    if (ptr == NULL) return 0;
    CRP_TRY(hipFree(ptr));
    return 0;
}

int crp_dev_memset(void *ptr, int value, size_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    CRP_TRY(hipMemsetAsync(ptr, value, bytes, (hipStream_t) stream));
    return 0;
}

int crp_dev_memcpy(void *dst, const void *src, size_t bytes, int kind, void *stream)
{
    if (bytes == 0) return 0;
    hipMemcpyKind k;
    if (kind == 0) k = hipMemcpyHostToDevice;
    else if (kind == 1) k = hipMemcpyDeviceToHost;
    else if (kind == 2) k = hipMemcpyDeviceToDevice;
    else return -1;
    CRP_TRY(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t) stream));
    return 0;
}

int crp_dev_memcpy2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t height,
                     int kind, void *stream)
{
    if (width_bytes == 0 || height == 0) return 0;
    hipMemcpyKind k;
    if (kind == 0) k = hipMemcpyHostToDevice;
    else if (kind == 1) k = hipMemcpyDeviceToHost;
    else if (kind == 2) k = hipMemcpyDeviceToDevice;
    else return -1;
    CRP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, k, (hipStream_t) stream));
    return 0;
}

int crp_host_malloc(void **ptr, size_t bytes)
{
    if (ptr == NULL) return -1;
    *ptr = NULL;
    if (bytes == 0) return 0;
    CRP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return 0;
}

int crp_host_free(void *ptr)
{
    if (ptr == NULL) return 0;
    CRP_TRY(hipHostFree(ptr));
    return 0;
}

int crp_dev_ptr_is_device(const void *ptr, int *is_dev)
{
    if (is_dev == NULL) return -1;
    *is_dev = 0;
    if (ptr == NULL) return 0;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess)
    {
        (void) hipGetLastError();   // plain malloc'd host memory is "invalid value": not an error for us
        return 0;
    }
    *is_dev = (attr.type == hipMemoryTypeDevice) ? 1 : 0;
    return 0;
}

int crp_stream_create(void **stream)
{
    if (stream == NULL) return -1;
    hipStream_t s;
    CRP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *) s;
    return 0;
}
int crp_stream_create_cu_mask(void **stream, int nwords, const unsigned *mask32)
{
    if (stream == NULL || nwords <= 0 || mask32 == NULL) return -1;
    hipStream_t s;
    CRP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t) nwords, mask32));
    *stream = (void *) s;
    return 0;
}
int crp_probe_copy(long long bytes, const void *src, void *dst, int blocks, unsigned long long *stamps, void *stream)
{
    if (bytes <= 0 || (bytes & 15) || src == NULL || dst == NULL || stamps == NULL) return -1;
    CRP_TRY(crp::probe_copy(bytes, src, dst, blocks, stamps, (hipStream_t) stream));
    return 0;
}
int crp_probe_stamp(unsigned long long *out, void *stream)
{
    if (out == NULL) return -1;
    CRP_TRY(crp::probe_stamp(out, (hipStream_t) stream));
    return 0;
}
int crp_stream_destroy(void *stream) { if (stream) CRP_TRY(hipStreamDestroy((hipStream_t) stream)); return 0; }
int crp_stream_sync(void *stream) { CRP_TRY(hipStreamSynchronize((hipStream_t) stream)); return 0; }

int crp_event_create(void **event)
{
    if (event == NULL) return -1;
    hipEvent_t e;
    CRP_TRY(hipEventCreate(&e));
    *event = (void *) e;
    return 0;
}
int crp_event_destroy(void *event) { if (event) CRP_TRY(hipEventDestroy((hipEvent_t) event)); return 0; }
int crp_event_record(void *event, void *stream) { CRP_TRY(hipEventRecord((hipEvent_t) event, (hipStream_t) stream)); return 0; }
int crp_event_sync(void *event) { CRP_TRY(hipEventSynchronize((hipEvent_t) event)); return 0; }
int crp_stream_wait_event(void *stream, void *event)
{
    CRP_TRY(hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) event, 0));
    return 0;
}
int crp_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (ms == NULL) return -1;
    CRP_TRY(hipEventElapsedTime(ms, (hipEvent_t) start, (hipEvent_t) stop));
    return 0;
}

// ---------------------------------------------------------------------------
// dst_val[dst_rowptr[t] + k] = src_val[src_start[t] + k]: the values of selected rows of a matrix whose values already sit in
// HBM (crp_csr_dev_create_dv); one wave per row
__global__ void gather_row_vals_kernel(const int nrow, const int *__restrict__ dst_rowptr, const int *__restrict__ src_start,
                                       const double *__restrict__ src_val, double *__restrict__ dst_val)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long) blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((long long) gridDim.x * blockDim.x) >> 6;
    for (long long t = wave0; t < nrow; t += nwave)
    {
        const int d0 = dst_rowptr[t], len = dst_rowptr[t + 1] - d0;
        const long long s0 = src_start[t];
        for (int k = lane; k < len; k += 64) dst_val[(long long) d0 + k] = src_val[s0 + k];
    }
}

static int csr_dev_create_impl(int nrow, int ncol, const int *rowptr, const int *colidx, const double *val, const double *val_dev,
                               const int *src_start, crp_csr_dev_p *out);

int crp_csr_dev_create(int nrow, int ncol, const int *rowptr, const int *colidx, const double *val,
                       crp_csr_dev_p *out)
{
    return csr_dev_create_impl(nrow, ncol, rowptr, colidx, val, nullptr, nullptr, out);
}

int crp_csr_dev_create_dv(int nrow, int ncol, const int *rowptr, const int *colidx, const double *val_host, const double *val_dev,
                          const int *src_start, crp_csr_dev_p *out)
{
    if (val_dev == NULL) return -1;
    return csr_dev_create_impl(nrow, ncol, rowptr, colidx, val_host, val_dev, src_start, out);
}

static int csr_dev_create_impl(int nrow, int ncol, const int *rowptr, const int *colidx, const double *val, const double *val_dev,
                               const int *src_start, crp_csr_dev_p *out)
{
    if (out == NULL) return -1;
    *out = NULL;
    if (nrow < 0 || ncol < 0 || rowptr == NULL) return -1;
    if (rowptr[0] != 0) return -2;
    const long long nnz = rowptr[nrow];
    if (nnz < 0) return -2;
    if (nnz > 0 && (colidx == NULL || val == NULL)) return -1;
    crp_csr_dev *A = new (std::nothrow) crp_csr_dev;
    if (A == NULL) return -3;
    A->nrow = nrow;
    A->ncol = ncol;
    A->nnz  = nnz;
    hipError_t e;
    e = hipMalloc((void **) &A->rowptr, sizeof(int) * ((size_t) nrow + 1));
    if (e == hipSuccess) e = hipMalloc((void **) &A->colidx, sizeof(int) * (size_t) (nnz > 0 ? nnz : 1));
    if (e == hipSuccess) e = hipMalloc((void **) &A->val, sizeof(double) * (size_t) (nnz > 0 ? nnz : 1));
    if (e == hipSuccess) e = hipMemcpy(A->rowptr, rowptr, sizeof(int) * ((size_t) nrow + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0) e = hipMemcpy(A->colidx, colidx, sizeof(int) * (size_t) nnz, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0 && val_dev == nullptr) e = hipMemcpy(A->val, val, sizeof(double) * (size_t) nnz, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0 && val_dev != nullptr)
    {
        // the values are in HBM already (a panel replicated between devices): no second trip over PCIe
        if (src_start == nullptr) e = hipMemcpy(A->val, val_dev, sizeof(double) * (size_t) nnz, hipMemcpyDeviceToDevice);
        else
        {
            int *d_start = nullptr;
            e = hipMalloc((void **) &d_start, sizeof(int) * (size_t) (nrow > 0 ? nrow : 1));
            if (e == hipSuccess && nrow > 0) e = hipMemcpy(d_start, src_start, sizeof(int) * (size_t) nrow, hipMemcpyHostToDevice);
            if (e == hipSuccess && nrow > 0)
            {
                const int blocks = (int) std::min<long long>(((long long) nrow + 3) / 4, 65536);
                hipLaunchKernelGGL(gather_row_vals_kernel, dim3(blocks), dim3(256), 0, 0, nrow, A->rowptr, d_start, val_dev, A->val);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipDeviceSynchronize();
            }
            if (d_start) (void) hipFree(d_start);
        }
    }
    if (e != hipSuccess)
    {
        crp_csr_dev_p tmp = A;
        crp_csr_dev_destroy(&tmp);
        return (int) e;
    }
    for (long long p = 0; p < nnz; p++)
    {
        const int c = colidx[p];
        if (c >= 0) { if (c + 1LL > A->b0_rows) A->b0_rows = c + 1LL; }
        else if ((long long) (~c) + 1 > A->b1_rows) A->b1_rows = (long long) (~c) + 1;
    }
    if (A->b0_rows > ncol && ncol > 0)
    {
        crp_csr_dev_p tmp = A;
        crp_csr_dev_destroy(&tmp);
        return -2;                         // a column index addresses a row past B0
    }
    A->h_rowptr.assign(rowptr, rowptr + nrow + 1);
    if (nnz > 0)
    {
        A->h_colidx.assign(colidx, colidx + nnz);
        A->h_val.assign(val, val + nnz);
    }
    // Locality order of the rows (locality.h) for the derived formats: taken when it lets rows of a panel share
    // more columns than the caller's order does (fewer R = 8 panel entries).  A mesh numbered along its own lines
    // (the stride-lattice matrices) keeps the caller's order -- consecutive rows there are neighbours already and
    // the lattice schedules build on that.  CRPSPMM_REORDER=0 never, =1 whenever the matrix qualifies.
    if (nnz > 0 && nrow >= 2048 && nrow == ncol && A->b1_rows == 0 && nnz <= 200000000LL)    // (the graph of a larger matrix costs tens of GB)
    {
        const int mode = crp::knobs().reorder;
        std::vector<int> perm;
        if (mode != 0 && crp::locality_reorder(nrow, ncol, rowptr, colidx, 8, &perm))
        {
            A->f_rowptr.assign((size_t) nrow + 1, 0);
            for (int i = 0; i < nrow; i++) A->f_rowptr[(size_t) i + 1] = A->f_rowptr[(size_t) i] + (rowptr[perm[(size_t) i] + 1] - rowptr[perm[(size_t) i]]);
            A->f_colidx.resize((size_t) nnz);
            A->f_nz.resize((size_t) nnz);
            for (int i = 0; i < nrow; i++)
            {
                const int r = perm[(size_t) i];
                int q = A->f_rowptr[(size_t) i];
                for (int pz = rowptr[r]; pz < rowptr[r + 1]; pz++, q++)
                {
                    A->f_colidx[(size_t) q] = colidx[pz];
                    A->f_nz[(size_t) q] = (uint32_t) pz;
                }
            }
            const long long e_nat = crp::count_panel_entries(nrow, rowptr, colidx, 8);
            const long long e_loc = crp::count_panel_entries(nrow, A->f_rowptr.data(), A->f_colidx.data(), 8);
            if (mode == 1 || (double) e_loc < 0.9 * (double) e_nat)
            {
                A->perm.swap(perm);
                A->f_val.resize((size_t) nnz);
                for (long long q = 0; q < nnz; q++) A->f_val[(size_t) q] = val[A->f_nz[(size_t) q]];
                hipError_t e2 = hipMalloc((void **) &A->rowmap_fmt, sizeof(int) * (size_t) nrow);
                if (e2 == hipSuccess) e2 = hipMemcpy(A->rowmap_fmt, A->perm.data(), sizeof(int) * (size_t) nrow, hipMemcpyHostToDevice);
                if (e2 != hipSuccess)
                {
                    crp_csr_dev_p tmp = A;
                    crp_csr_dev_destroy(&tmp);
                    return (int) e2;
                }
            }
            else
            {
                A->f_rowptr.clear(); A->f_colidx.clear(); A->f_nz.clear();
                A->f_rowptr.shrink_to_fit(); A->f_colidx.shrink_to_fit(); A->f_nz.shrink_to_fit();
            }
        }
    }
    // Pick the kernel family variant 0 resolves to.  The panel kernels pay off when rows of a
    // panel share columns (banded / FEM / block structure); with no sharing (fill -> 1/R) the
    // plain CSR kernel moves fewer bytes.  CRPSPMM_SPMM_VARIANT overrides (1, 2 or 3).
    A->auto_variant = 1;
    if (nnz > 0 && nrow >= 8)
    {
        const long long e4 = crp::count_panel_entries(nrow, fmt_rowptr(A), fmt_colidx(A), 4);
        const long long e8 = crp::count_panel_entries(nrow, fmt_rowptr(A), fmt_colidx(A), 8);
        const double fill4 = (double) nnz / (4.0 * (double) e4);
        if (fill4 >= 0.45) A->auto_variant = ((double) e8 <= 0.72 * (double) e4) ? 3 : 2;
        // R = 8 panels also when an entry serves 1.7 rows or more on average, whatever R = 4 would do: the nlpkkt stand-in
        // (fill4 0.42, e8 / e4 0.89, e8 = 0.53 nnz) runs 0.63 / 1.04 / 1.32 ms at n = 32 / 64 / 96 on R = 8 panels against
        // 0.73 / 1.34 / 1.95 through CSR and 1.00 / 1.10 / 1.45 on R = 4; the shell stand-in 0.106 / 0.129 against 0.135 / 0.144
        // on R = 4.  Erdos-Renyi (e8 = nnz) stays with CSR.
        if ((double) e8 <= 0.6 * (double) nnz) A->auto_variant = 3;
        // ... or has no lattice team schedule to time its panels' shared rows in L2 (shell stand-in, locality order: n = 64 0.122 ms on the
        // row-panel kernel, 0.103 on the team kernel; the pwtk stand-in, a lattice with 10.4 slices per row: 0.115 / 0.120)
        {
            double D1 = 0, D2 = 0;
            int M = 0;
            const bool lat = nrow >= 4096 && crp::detect_stride_lattice(nrow, fmt_rowptr(A), fmt_colidx(A), 8, &D1, &D2, &M);
            if (!lat) A->team2_min_n = TEAM2_MIN_N_NOLATTICE;
            if ((double) e8 > 12.0 * (double) nrow) A->team2_min_n = TEAM2_MIN_N_DENSE;
        }
        if ((double) nnz < 0.35 * 8.0 * (double) e8)
        {
            A->team2_min_n = TEAM2_MIN_N_SPARSE;
            // ... and at 24 .. 32 columns (24 .. 64 until the team kernel's half-piece instances: TEAM2_MIN_N_SPARSE) such panels go to the row-owner team kernel (variant 7, csrc/team2r_kernel.hip): nlpkkt
            // stand-in 0.388 / 0.839 ms at n = 32 / 64 against 0.546 / 1.04 of the narrow and row-panel kernels, at nlpkkt240 size
            // 6.58 / 14.0 against 8.85 / 18.2 (pwtk stand-in, fill 0.61: 0.075 against 0.062 -- stays).  CRPSPMM_TEAM2R=0|1 forces.
            A->team2r_pays = true;
        }
    }
    // The LDS-sharing team kernel fetches a B row once per team of 64 rows: it pays when those rows name far fewer
    // distinct columns than they have nonzeros (pwtk stand-in 0.10, shell 0.09, kkt 0.27, fem3d 0.13 of the nonzeros;
    // Erdos-Renyi 0.99, where the CSR kernel stays).
    if (nnz > 0 && nrow >= 64)
        A->team2_pays = (double) crp::count_block_union(nrow, fmt_rowptr(A), fmt_colidx(A), 64) <= 0.6 * (double) nnz;
    if (crp::knobs().spmm_variant >= 1 && crp::knobs().spmm_variant <= 3) A->auto_variant = crp::knobs().spmm_variant;
    // (the derived formats are built by the first product that uses them: a matrix multiplied by wide operands only never
    //  needs its row-panel format -- 20 GB for the nlpkkt240-size stand-in)
    *out = A;
    return 0;
}

int crp_csr_dev_destroy(crp_csr_dev_p *A_)
{
    if (A_ == NULL || *A_ == NULL) return 0;
    crp_csr_dev *A = *A_;
    for (int i = 0; i < 2; i++)
    {
        if (A->pan[i].pptr) (void) hipFree(A->pan[i].pptr);
        if (A->pan[i].porder) (void) hipFree(A->pan[i].porder);
        if (A->pan[i].psync) (void) hipFree(A->pan[i].psync);
        if (A->pan[i].pcol) (void) hipFree(A->pan[i].pcol);
        if (A->pan[i].pmask4) (void) hipFree(A->pan[i].pmask4);
        if (A->pan[i].pmap) (void) hipFree(A->pan[i].pmap);
        if (A->pan[i].pval) (void) hipFree(A->pan[i].pval);
        if (A->pan[i].cmo) (void) hipFree(A->pan[i].cmo);
        if (A->pan[i].cbase) (void) hipFree(A->pan[i].cbase);
        if (A->pan[i].cval) (void) hipFree(A->pan[i].cval);
        if (A->pan[i].cmap) (void) hipFree(A->pan[i].cmap);
    }
    for (Team2Dev *t2 : {&A->team2})
    {
        if (t2->torder) (void) hipFree(t2->torder);
        if (t2->tpanel) (void) hipFree(t2->tpanel);
        if (t2->tinfo) (void) hipFree(t2->tinfo);
        if (t2->tpro) (void) hipFree(t2->tpro);
        if (t2->trec) (void) hipFree(t2->trec);
        if (t2->tvoff) (void) hipFree(t2->tvoff);
        if (t2->tval) (void) hipFree(t2->tval);
        if (t2->tmap) (void) hipFree(t2->tmap);
        if (t2->tval32) (void) hipFree(t2->tval32);
    }
    for (Team2RDev *tnp : {&A->team2r[0], &A->team2r[1]})
    {
        Team2RDev &tn = *tnp;
        if (tn.tgrid) (void) hipFree(tn.tgrid);
        if (tn.tpanel) (void) hipFree(tn.tpanel);
        if (tn.tinfo) (void) hipFree(tn.tinfo);
        if (tn.trec) (void) hipFree(tn.trec);
        if (tn.tvoff) (void) hipFree(tn.tvoff);
        if (tn.tval) (void) hipFree(tn.tval);
        if (tn.tmap) (void) hipFree(tn.tmap);
        if (tn.tent) (void) hipFree(tn.tent);
    }
    if (A->val32) (void) hipFree(A->val32);
    if (A->rowptr) (void) hipFree(A->rowptr);
    if (A->colidx) (void) hipFree(A->colidx);
    if (A->val) (void) hipFree(A->val);
    if (A->rowmap) (void) hipFree(A->rowmap);
    if (A->rowmap_fmt) (void) hipFree(A->rowmap_fmt);
    delete A;
    *A_ = NULL;
    return 0;
}

int crp_csr_dev_update_values(crp_csr_dev_p A, const double *val, void *stream)
{
    if (A == NULL || (val == NULL && A->nnz > 0)) return -1;
    if (A->nnz == 0) return 0;
    // the slot maps of the derived formats are 32-bit
    for (int i = 0; i < 2; i++)
        if (A->pan[i].built && A->pan[i].entries * (long long) A->pan[i].R >= (1LL << 32)) return -5;
    if (A->team2.built && A->team2.value_entries >= (1LL << 32)) return -5;
    for (Team2RDev *tnp : {&A->team2r[0], &A->team2r[1]})
        if (tnp->built && tnp->value_entries >= (1LL << 32)) return -5;
    int is_dev = 0;
    crp_dev_ptr_is_device(val, &is_dev);
    CRP_TRY(hipMemcpyAsync(A->val, val, sizeof(double) * (size_t) A->nnz, is_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           (hipStream_t) stream));
    if (!is_dev)
    {
        memcpy(A->h_val.data(), val, sizeof(double) * (size_t) A->nnz);   // formats built later see the new values
        for (size_t q = 0; q < A->f_val.size(); q++) A->f_val[q] = val[A->f_nz[q]];
        A->host_vals_stale = false;
    }
    else A->host_vals_stale = true;      // formats built later are refreshed from the device CSR (ensure_*)
    for (int i = 0; i < 2; i++)
        if (A->pan[i].built)
        {
            CRP_TRY(crp::scatter_vals_f64(A->nnz, A->pan[i].pmap, A->val, A->pan[i].pval, (hipStream_t) stream));
            if (A->pan[i].cmap) CRP_TRY(crp::scatter_vals_f64(A->nnz, A->pan[i].cmap, A->val, A->pan[i].cval, (hipStream_t) stream));
        }
    for (Team2Dev *t2 : {&A->team2})
        if (t2->built) CRP_TRY(crp::scatter_vals_f64(A->nnz, t2->tmap, A->val, t2->tval, (hipStream_t) stream));
    for (Team2RDev *tnp : {&A->team2r[0], &A->team2r[1]})
        if (tnp->built) CRP_TRY(crp::scatter_vals_f64(A->nnz, tnp->tmap, A->val, tnp->tval, (hipStream_t) stream));
    // fp32 copies follow
    if (A->val32) CRP_TRY(crp::convert_f64_f32(A->nnz, A->val, A->val32, (hipStream_t) stream));
    for (Team2Dev *t2 : {&A->team2})
        if (t2->tval32) CRP_TRY(crp::convert_f64_f32(t2->value_entries, t2->tval, t2->tval32, (hipStream_t) stream));
    return 0;
}

int crp_csr_dev_set_rowmap(crp_csr_dev_p A, const int *rowmap, int c_nrow)
{
    if (A != NULL) A->rowmap_epoch++;
    if (A == NULL || (rowmap != NULL && c_nrow < 0)) return -1;
    if (A->rowmap) { CRP_TRY(hipFree(A->rowmap)); A->rowmap = nullptr; }
    A->c_nrow = A->nrow;
    A->h_rowmap.clear();
    if (!A->perm.empty())        // formats in processing order: C row of position i = map[perm[i]]
    {
        std::vector<int> comp(A->perm);
        if (rowmap != NULL)
            for (int i = 0; i < A->nrow; i++)
            {
                const int r = rowmap[A->perm[(size_t) i]];
                if (r < 0 || r >= c_nrow) return -2;
                comp[(size_t) i] = r;
            }
        CRP_TRY(hipMemcpy(A->rowmap_fmt, comp.data(), sizeof(int) * (size_t) A->nrow, hipMemcpyHostToDevice));
    }
    if (rowmap == NULL || A->nrow == 0) return 0;
    for (int i = 0; i < A->nrow; i++)
        if (rowmap[i] < 0 || rowmap[i] >= c_nrow) return -2;
    CRP_TRY(hipMalloc((void **) &A->rowmap, sizeof(int) * (size_t) A->nrow));
    CRP_TRY(hipMemcpy(A->rowmap, rowmap, sizeof(int) * (size_t) A->nrow, hipMemcpyHostToDevice));
    A->h_rowmap.assign(rowmap, rowmap + A->nrow);
    A->c_nrow = c_nrow;
    return 0;
}

int crp_csr_dev_nrow(crp_csr_dev_p A) { return A ? A->nrow : -1; }
long long crp_csr_dev_nnz(crp_csr_dev_p A) { return A ? A->nnz : -1; }
long long crp_csr_dev_bytes(crp_csr_dev_p A) { return A ? 12LL * A->nnz + 4LL * ((long long) A->nrow + 1) : -1; }

// csr_mat_row_part_comm_size (/root/reference/src/spmat_part.c:38-64) on the device-resident CSR of A: the re-plan case -- the
// matrix is in HBM already, so nothing but the two partition arrays goes up and nblk + 1 ints come down.
int crp_csr_dev_row_part_comm_size(crp_csr_dev_p A, int nblk, const int *rblk_ptr, const int *x_displs, int *comm_sizes, int *total_size)
{
    if (A == NULL || nblk < 1 || rblk_ptr == NULL || x_displs == NULL || comm_sizes == NULL || total_size == NULL) return -1;
    if (rblk_ptr[0] != 0 || rblk_ptr[nblk] != A->nrow) return -2;
    for (int b = 0; b < nblk; b++)
        if (rblk_ptr[b + 1] < rblk_ptr[b] || x_displs[b + 1] < x_displs[b] || x_displs[b] < 0 || x_displs[b + 1] > A->ncol) return -2;
    const long long words = ((long long) A->ncol + 31) / 32;
    unsigned *bits = nullptr;
    int *dv = nullptr;                      // rblk (nblk + 1), xd (nblk + 1), comm (nblk), bad (1)
    const size_t nint = (size_t) 3 * (size_t) nblk + 3;
    hipError_t e = hipMalloc((void **) &bits, sizeof(unsigned) * (size_t) std::max<long long>(words * nblk, 1));
    if (e == hipSuccess) e = hipMalloc((void **) &dv, sizeof(int) * nint);
    if (e == hipSuccess) e = hipMemset(bits, 0, sizeof(unsigned) * (size_t) std::max<long long>(words * nblk, 1));
    if (e == hipSuccess) e = hipMemset(dv, 0, sizeof(int) * nint);
    if (e == hipSuccess) e = hipMemcpy(dv, rblk_ptr, sizeof(int) * ((size_t) nblk + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dv + nblk + 1, x_displs, sizeof(int) * ((size_t) nblk + 1), hipMemcpyHostToDevice);
    int *comm_dev = dv + 2 * (nblk + 1), *bad_dev = comm_dev + nblk;
    if (e == hipSuccess) e = crp::row_part_comm_size(A->nrow, A->ncol, A->rowptr, A->colidx, nblk, dv, dv + nblk + 1, bits, comm_dev, bad_dev, 0);
    std::vector<int> out((size_t) nblk + 1, 0);
    if (e == hipSuccess) e = hipMemcpy(out.data(), comm_dev, sizeof(int) * ((size_t) nblk + 1), hipMemcpyDeviceToHost);
    if (bits) (void) hipFree(bits);
    if (dv) (void) hipFree(dv);
    if (e != hipSuccess) return (int) e;
    if (out[(size_t) nblk] != 0) return -2;              // two-source column indices: not a matrix the planner partitions
    *total_size = 0;
    for (int b = 0; b < nblk; b++) { comm_sizes[b] = out[(size_t) b]; *total_size += out[(size_t) b]; }
    return 0;
}

// (4 and 6 were the round-1 LDS team kernel and the narrow team kernel of round 3: measured slower than what variant 0 picks,
//  removed in round 4; the numbers stay so that 5 and 7 keep their meaning)
static const char *k_variant_names[] = {"auto", "csr-rowgroup", "rowpanel-R4", "rowpanel-R8", "(removed)", "team2-R8", "(removed)", "team2r-R8"};
int crp_spmm_variant_count(void) { return (int) (sizeof(k_variant_names) / sizeof(k_variant_names[0])); }
const char *crp_spmm_variant_name(int variant)
{
    if (variant < 0 || variant >= crp_spmm_variant_count()) return NULL;
    return k_variant_names[variant];
}

int crp_spmm_csr_f64(crp_csr_dev_p A, int layout, int n, const double *B0, long long ldB0, const double *B1,
                     long long ldB1, double *C, long long ldC, int variant, void *stream)
{
    if (A == NULL || n < 0) return -1;
    if (layout != CRP_LAYOUT_ROW_MAJOR && layout != CRP_LAYOUT_COL_MAJOR) return -1;
    if (variant < 0 || variant >= crp_spmm_variant_count() || variant == 4 || variant == 6) return -1;
    if (A->nrow == 0 || n == 0) return 0;
    if (C == NULL || (B0 == NULL && B1 == NULL && A->nnz > 0)) return -1;
    if (layout == CRP_LAYOUT_ROW_MAJOR && (ldC < n || (B0 && ldB0 < n) || (B1 && ldB1 < n))) return -4;
    if (layout == CRP_LAYOUT_COL_MAJOR && ldC < (A->rowmap ? A->c_nrow : A->nrow)) return -4;
    crp::SpmmArgs a;
    a.nrow = A->nrow; a.n = n;
    a.rowptr = A->rowptr; a.colidx = A->colidx; a.val = A->val;
    a.B0 = B0; a.ldB0 = ldB0; a.B1 = B1; a.ldB1 = ldB1; a.C = C; a.ldC = ldC;
    a.rowmap = A->rowmap;
    hipError_t e;
    if (layout == CRP_LAYOUT_COL_MAJOR) return (int) crp::spmm_cm_f64(a, (hipStream_t) stream);
    int v = (variant == 0) ? A->auto_variant : variant;
    // auto: from TEAM2_MIN_N columns on the LDS-sharing team kernel wherever teams share columns (against the best
    // other variant on the pwtk / shell / fem3d stand-ins: n = 128: 1.00 / 0.81 / 0.73 of its time, n = 256: 0.85 /
    // 0.65 / 0.63; at n = 96 -- a tile of 128 columns three quarters used -- 1.17 / 1.00 / 0.94, at n = 32 1.6 x)
    if (variant == 0 && A->team2_pays && team2_width(A, n) && crp::spmm_team2_applicable(a)) v = 5;
    if (v == 5 && (!crp::spmm_team2_applicable(a) || A->nnz == 0 || A->nrow < 8)) v = 3;
    // narrow operands (24 <= n <= 64) whose R = 8 panels are mostly holes: the team kernel whose lane groups own rows (variant 7)
    {
        crp::Team2NArgs tn;
        tn.G = n <= 32 ? 4 : 2;
        if (variant == 0 && v != 5 && A->team2_pays && team2r_auto(A) && n <= 64 && A->nnz > 0 && A->nrow >= 8 && crp::spmm_team2r_applicable(tn, a)) v = 7;
        if (v == 7 && (!crp::spmm_team2r_applicable(tn, a) || A->nnz == 0 || A->nrow < 8)) v = 3;
    }
    // the derived formats hold the rows in processing order: their C row map is chosen per launch, AFTER every fallback has
    // resolved (a re-ordered matrix that falls back to the CSR kernel writes through the caller's map)
    int *const fmt_map = A->rowmap_fmt != nullptr ? A->rowmap_fmt : A->rowmap;
    if (v == 5)
    {
        a.rowmap = fmt_map;
        A->last_variant = 5;
        const int rc = ensure_team2(A, (hipStream_t) stream);
        if (rc != 0) return rc;
        crp::Team2Args t;
        team2_args(A->team2, &t);
        return (int) crp::spmm_rm_f64_team2(t, a, (hipStream_t) stream);
    }
    if (v == 7)
    {
        a.rowmap = fmt_map;
        A->last_variant = 7;
        const int G = n <= 32 ? 4 : 2;
        const int rc = ensure_team2r(A, (hipStream_t) stream, G);
        if (rc == -6 && variant == 0)
        {
            // the streams of this matrix would pass their 32-bit offsets: variant 0 goes on with the row-panel kernels, for good
            A->team2r_pays = false;
            return crp_spmm_csr_f64(A, layout, n, B0, ldB0, B1, ldB1, C, ldC, crp::knobs().team2r >= 0 ? 3 : 0, stream);
        }
        if (rc != 0) return rc;
        Team2RDev &d = A->team2r[G == 2 ? 1 : 0];
        crp::Team2NArgs t;
        t.G = d.G; t.nteam = d.nteam; t.ngrid = d.ngrid; t.tgrid = d.tgrid; t.tpanel = d.tpanel; t.tinfo = d.tinfo; t.trec = d.trec; t.tvoff = d.tvoff; t.tval = d.tval;
        t.tent = d.tent;
        if (d.rows_epoch != A->rowmap_epoch || d.rows_map != a.rowmap)      // the C rows of the panels, once per row map
        {
            CRP_TRY(crp::team2r_fill_rows(t, a, (hipStream_t) stream));
            d.rows_epoch = A->rowmap_epoch;
            d.rows_map = a.rowmap;
        }
        return (int) crp::spmm_rm_f64_team2r(t, a, (hipStream_t) stream);
    }
    if (v >= 2 && (!crp::spmm_panel_applicable(a) || A->nnz == 0)) v = 1;   // narrow / unaligned operands
    A->last_variant = v;
    if (v >= 2)
    {
        a.rowmap = fmt_map;
        const int rc = ensure_panel(A, v - 2, (hipStream_t) stream);       // no-op unless an explicit variant asks for a new format
        if (rc != 0) return rc;
        const PanelDev &d = A->pan[v - 2];
        crp::PanelArgs p;
        memset(&p, 0, sizeof(p));
        p.R = d.R; p.npanel = d.npanel; p.pptr = d.pptr; p.porder = d.porder; p.norder = d.norder; p.team_waves = d.team_waves; p.psync = d.psync; p.pcol = d.pcol; p.pmask4 = d.pmask4; p.pval = d.pval;
        p.b0_rows = A->b0_rows; p.b1_rows = A->b1_rows;
        p.cmo = d.cmo; p.cbase = d.cbase; p.cval = d.cval;
        p.narrow64 = false;                    // (the two-piece instance loses to the row-panel kernel even on compact values: nlpkkt
                                               //  stand-in n = 64 1.36 against 1.13 ms, pwtk stand-in 0.135 against 0.119; CRPSPMM_NARROW_MAX=64 forces it)
        e = crp::spmm_rm_f64_panel(p, a, (hipStream_t) stream);
    }
    else e = crp::spmm_rm_f64_rowgroup(a, (hipStream_t) stream);
    return (int) e;
}

// C[nrow x n] := A * B with values, B and C in fp32 (row-major only): the fp32 instance of the team kernel where it
// applies and pays (variant 0 / 5), the fp32 CSR row-group kernel otherwise (variant 1, any width and alignment)
int crp_spmm_csr_f32(crp_csr_dev_p A, int n, const float *B0, long long ldB0, const float *B1, long long ldB1, float *C,
                     long long ldC, int variant, void *stream)
{
    if (A == NULL || n < 0) return -1;
    if (variant != 0 && variant != 1 && variant != 5) return -1;
    if (A->nrow == 0 || n == 0) return 0;
    if (C == NULL || (B0 == NULL && B1 == NULL && A->nnz > 0)) return -1;
    if (ldC < n || (B0 && ldB0 < n) || (B1 && ldB1 < n)) return -4;
    if (A->val32 == nullptr)
    {
        CRP_TRY(hipMalloc((void **) &A->val32, sizeof(float) * (size_t) (A->nnz > 0 ? A->nnz : 1)));
        CRP_TRY(crp::convert_f64_f32(A->nnz, A->val, A->val32, (hipStream_t) stream));
    }
    crp::SpmmArgsF32 a;
    a.nrow = A->nrow; a.n = n; a.rowptr = A->rowptr; a.colidx = A->colidx; a.val = A->val32;
    a.B0 = B0; a.ldB0 = ldB0; a.B1 = B1; a.ldB1 = ldB1; a.C = C; a.ldC = ldC; a.rowmap = A->rowmap;
    const bool team = (variant == 5 || (variant == 0 && A->team2_pays && n >= (A->team2r_pays ? TEAM2_MIN_N_F32_SPARSE : TEAM2_MIN_N_F32))) && A->nnz > 0 && A->nrow >= 8 &&
                      crp::spmm_team2_applicable_f32(a);
    A->last_variant = team ? 5 : 1;
    if (!team) return (int) crp::spmm_rm_f32_rowgroup(a, (hipStream_t) stream);
    const int rc = ensure_team2(A, (hipStream_t) stream, true);
    if (rc != 0) return rc;
    Team2Dev &d = A->team2;
    if (d.tval32 == nullptr)
    {
        CRP_TRY(hipMalloc((void **) &d.tval32, sizeof(float) * ((size_t) d.value_entries + 1024)));
        CRP_TRY(hipMemsetAsync(d.tval32, 0, sizeof(float) * ((size_t) d.value_entries + 1024), (hipStream_t) stream));
        CRP_TRY(crp::convert_f64_f32(d.value_entries, d.tval, d.tval32, (hipStream_t) stream));
    }
    if (A->rowmap_fmt != nullptr) a.rowmap = A->rowmap_fmt;
    crp::Team2Args t;
    team2_args(d, &t);
    return (int) crp::spmm_rm_f32_team2(t, a, (hipStream_t) stream);
}

int crp_csr_dev_auto_variant(crp_csr_dev_p A) { return A ? A->auto_variant : -1; }
int crp_csr_dev_reordered(crp_csr_dev_p A) { return A ? (A->perm.empty() ? 0 : 1) : -1; }
int crp_csr_dev_resolved_variant(crp_csr_dev_p A, int n)
{
    if (A == NULL) return -1;
    int v = A->auto_variant;
    if (v >= 2 && n < 24) v = 1;
    if (A->team2_pays && team2_width(A, n) && (n % 2 == 0)) v = 5;
    else if (A->team2_pays && team2r_auto(A) && n >= 24 && n <= 64 && (n % 2 == 0) && A->nnz > 0 && A->nrow >= 8) v = 7;
    return v;
}
int crp_csr_dev_last_variant(crp_csr_dev_p A) { return A ? A->last_variant : -1; }
int crp_csr_dev_lattice(crp_csr_dev_p A) { return A ? ((A->team2.built && A->team2.lattice) ? 1 : 0) : -1; }

int crp_panel_format_host(int nrow, const int *rowptr, const int *colidx, const double *val, int R, int *npanel,
                          int **pptr, int **pcol, unsigned **pmask4, double **pval, long long *real_entries,
                          int **porder, int *norder)
{
    if (nrow < 0 || rowptr == NULL || (R != 4 && R != 8) || !npanel || !pptr || !pcol || !pmask4 || !pval) return -1;
    crp::PanelHost h;
    crp::build_panels(nrow, rowptr, colidx, val, R, &h);
    *npanel = h.npanel;
    *pptr = (int *) malloc(sizeof(int) * h.pptr.size());
    *pcol = (int *) malloc(sizeof(int) * (h.pcol.size() + 1));
    *pmask4 = (unsigned *) malloc(sizeof(unsigned) * h.pmask4.size());
    *pval = (double *) malloc(sizeof(double) * (h.pval.size() + 1));
    memcpy(*pptr, h.pptr.data(), sizeof(int) * h.pptr.size());
    if (!h.pcol.empty()) memcpy(*pcol, h.pcol.data(), sizeof(int) * h.pcol.size());
    memcpy(*pmask4, h.pmask4.data(), sizeof(unsigned) * h.pmask4.size());
    if (!h.pval.empty()) memcpy(*pval, h.pval.data(), sizeof(double) * h.pval.size());
    if (real_entries) *real_entries = h.real_entries;
    if (norder) *norder = (int) h.porder.size();
    if (porder)
    {
        *porder = (int *) malloc(sizeof(int) * (h.porder.size() + 1));
        if (!h.porder.empty()) memcpy(*porder, h.porder.data(), sizeof(int) * h.porder.size());
    }
    return 0;
}

int crp_team_format_host(int nrow, const int *rowptr, const int *colidx, const double *val, int *nteam, int *lattice,
                         int **tpanel, int **tptr, int **tcol, unsigned **tmask, int **torder)
{
    if (nrow < 0 || rowptr == NULL || !nteam || !tpanel || !tptr || !tcol || !tmask || !torder) return -1;
    crp::PanelHost h;
    crp::build_panels(nrow, rowptr, colidx, val, 8, &h, false);
    crp::TeamHost th;
    crp::build_teams(h, nrow, rowptr, colidx, &th);
    *nteam = th.nteam;
    if (lattice) *lattice = th.lattice ? 1 : 0;
    auto dup_i = [](const auto &v) {
        int *p = (int *) malloc(sizeof(int) * (v.size() + 1));
        if (!v.empty()) memcpy(p, v.data(), sizeof(int) * v.size());
        return p;
    };
    *tpanel = dup_i(th.tpanel);
    *tptr = dup_i(th.tptr);
    *tcol = dup_i(th.tcol);
    *torder = dup_i(th.torder);
    *tmask = (unsigned *) malloc(sizeof(unsigned) * (th.tmask.size() + 1));
    if (!th.tmask.empty()) memcpy(*tmask, th.tmask.data(), sizeof(unsigned) * th.tmask.size());
    return 0;
}

static std::vector<int> g_last_tgrid;       // launch grid of the last crp_team2_format_host() (planning / test helper)
static int g_last_compact = 1;              // ... and whether its value blocks are compact
int crp_team2_format_host_compact(void) { return g_last_compact; }
int crp_team2_format_host_grid(int **tgrid, int *ngrid)
{
    if (tgrid == NULL || ngrid == NULL) return -1;
    *ngrid = (int) g_last_tgrid.size();
    *tgrid = (int *) malloc(sizeof(int) * (g_last_tgrid.size() + 1));
    if (!g_last_tgrid.empty()) memcpy(*tgrid, g_last_tgrid.data(), sizeof(int) * g_last_tgrid.size());
    return 0;
}

int crp_team2r_format_host(int nrow, const int *rowptr, const int *colidx, const double *val, int G, int *nteam, int *lattice, int **tpanel,
                           int **tinfo, unsigned **trec, long long *nrecwords, long long **tvoff, double **tval, long long *nwords,
                           int **tgrid, int *ngrid, unsigned **vmap, long long *stats, unsigned **tent)
{
    if (nrow < 0 || rowptr == NULL || (G != 2 && G != 4) || !nteam || !tpanel || !tinfo || !trec || !nrecwords || !tvoff || !tval || !nwords || !tgrid || !ngrid)
        return -1;
    crp::PanelHost h;
    crp::build_panels(nrow, rowptr, colidx, val, 8, &h, false, false);
    crp::Team2RHost th;
    th.G = G;
    if (!crp::build_team2r(h, nrow, rowptr, colidx, &th)) return -6;
    *nteam = th.nteam;
    if (lattice) *lattice = th.lattice ? 1 : 0;
    auto dup = [](const void *src, size_t bytes) {
        void *p = malloc(bytes + 8);
        if (bytes) memcpy(p, src, bytes);
        return p;
    };
    *tpanel = (int *) dup(th.tpanel.data(), sizeof(int) * th.tpanel.size());
    *tinfo = (int *) dup(th.tinfo.data(), sizeof(int) * th.tinfo.size());
    *tgrid = (int *) dup(th.tgrid.data(), sizeof(int) * th.tgrid.size());
    *ngrid = (int) th.tgrid.size();
    *trec = (unsigned *) dup(th.trec.data(), sizeof(unsigned) * th.trec.size());
    *nrecwords = (long long) th.trec.size();
    *tvoff = (long long *) dup(th.tvoff.data(), sizeof(long long) * th.tvoff.size());
    *tval = (double *) dup(th.tval.data(), sizeof(double) * (size_t) th.nwords);
    *nwords = th.nwords;
    if (vmap) *vmap = (unsigned *) dup(th.vmap.data(), sizeof(unsigned) * th.vmap.size());
    if (stats) { stats[0] = th.rounds; stats[1] = th.steps; stats[2] = th.slots_filled; stats[3] = th.nnz; }
    if (tent) *tent = (unsigned *) dup(th.tent.data(), sizeof(unsigned) * th.tent.size());
    return 0;
}

int crp_team2_format_host(int nrow, const int *rowptr, const int *colidx, const double *val, int *nteam, int *lattice,
                          int **tpanel, int **tinfo, int **tpro, unsigned **trec, long long *nrecwords,
                          long long **tvoff, double **tval, long long *nvalent, int **torder, unsigned **vmap)
{
    if (nrow < 0 || rowptr == NULL || !nteam || !tpanel || !tinfo || !tpro || !trec || !nrecwords || !tvoff || !tval ||
        !nvalent || !torder)
        return -1;
    crp::PhaseClock clk;
    crp::PanelHost h;
    crp::build_panels(nrow, rowptr, colidx, val, 8, &h, false, false);
    clk.lap("crp_team2_format_host: build_panels (R = 8)");
    crp::Team2Host th;
    // (the test helper reads CRPSPMM_TEAM2_COMPACT per call: the product reads it once, crp::knobs())
    th.compact = getenv("CRPSPMM_TEAM2_COMPACT") ? atoi(getenv("CRPSPMM_TEAM2_COMPACT")) != 0 : h.fill() < 0.4;
    crp::build_team2(h, nrow, rowptr, colidx, &th);
    g_last_compact = th.compact ? 1 : 0;
    clk.lap("crp_team2_format_host: build_team2");
    g_last_tgrid = th.tgrid;
    *nteam = th.nteam;
    if (lattice) *lattice = th.lattice ? 1 : 0;
    auto dup_i = [](const std::vector<int> &v) {
        int *p = (int *) malloc(sizeof(int) * (v.size() + 1));
        if (!v.empty()) memcpy(p, v.data(), sizeof(int) * v.size());
        return p;
    };
    *tpanel = dup_i(th.tpanel);
    *tinfo = dup_i(th.tinfo);
    *tpro = dup_i(th.tpro);
    *torder = dup_i(th.torder);
    *trec = (unsigned *) malloc(sizeof(unsigned) * (th.trec.size() + 1));
    if (!th.trec.empty()) memcpy(*trec, th.trec.data(), sizeof(unsigned) * th.trec.size());
    *nrecwords = (long long) th.trec.size();
    *tvoff = (long long *) malloc(sizeof(long long) * (th.tvoff.size() + 1));
    memcpy(*tvoff, th.tvoff.data(), sizeof(long long) * th.tvoff.size());
    *nvalent = th.nvalues;
    *tval = (double *) calloc(th.tval.size() + 1, sizeof(double));
    if (!th.tval.empty()) memcpy(*tval, th.tval.data(), sizeof(double) * th.tval.size());
    if (vmap)
    {
        *vmap = (unsigned *) malloc(sizeof(unsigned) * (th.vmap.size() + 1));
        if (!th.vmap.empty()) memcpy(*vmap, th.vmap.data(), sizeof(unsigned) * th.vmap.size());
    }
    return 0;
}

int crp_locality_order_host(int nrow, int ncol, const int *rowptr, const int *colidx, int nparts, int *perm, double *info)
{
    if (nrow < 0 || rowptr == NULL || perm == NULL || nparts < 1) return -1;
    std::vector<int> pv;
    crp::LocalityInfo li;
    const bool ok = crp::locality_reorder(nrow, ncol, rowptr, colidx, nparts, &pv, &li);
    if (!ok)
    {
        for (int i = 0; i < nrow; i++) perm[i] = i;
        return 1;
    }
    memcpy(perm, pv.data(), sizeof(int) * (size_t) nrow);
    if (info) { info[0] = li.groups; info[1] = li.parts; info[2] = li.mean_dist_before; info[3] = li.mean_dist_after; }
    return 0;
}

int crp_gather_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, long long lds,
                        double *dst, long long ldd, void *stream)
{
    if (nidx < 0 || n < 0 || (layout != 0 && layout != 1)) return -1;
    return (int) crp::gather_rows_f64(layout, nidx, n, ridx, src, lds, dst, ldd, (hipStream_t) stream);
}

int crp_scatter_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, long long lds,
                         double *dst, long long ldd, void *stream)
{
    if (nidx < 0 || n < 0 || (layout != 0 && layout != 1)) return -1;
    return (int) crp::scatter_rows_f64(layout, nidx, n, ridx, src, lds, dst, ldd, (hipStream_t) stream);
}

int crp_transpose_f64(int nrow, int ncol, const double *src, long long lds, double *dst, long long ldd,
                      void *stream)
{
    if (nrow < 0 || ncol < 0) return -1;
    return (int) crp::transpose_f64(nrow, ncol, src, lds, dst, ldd, (hipStream_t) stream);
}

}  // extern "C"
