// kernels.h -- internal launch interfaces between the C ABI (hip_api.hip) and
// the gfx950 kernels.  Not installed; the public surface is include/*.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crp {

struct SpmmArgs
{
    int nrow;
    int n;
    const int    *rowptr;
    const int    *colidx;
    const double *val;
    const double *B0;
    int64_t       ldB0;
    const double *B1;
    int64_t       ldB1;
    double       *C;
    int64_t       ldC;
    const int    *rowmap;   // nullptr, or the C row of every row (row-subset matrices)
};

struct SpmmArgsF32          // the fp32 path (values, B and C in fp32; BASELINE configs[3])
{
    int nrow;
    int n;
    const int   *rowptr;
    const int   *colidx;
    const float *val;
    const float *B0;
    int64_t      ldB0;
    const float *B1;
    int64_t      ldB1;
    float       *C;
    int64_t      ldC;
    const int   *rowmap;
};

struct PanelArgs
{
    int R;
    int npanel;
    const int      *pptr;
    const int      *porder;    // processing order: norder records {panel or -1, first entry, rounds, 0}
    int             norder;
    int             team_waves; // waves per workgroup the order is laid out for (4 or 6)
    const int      *psync;     // per workgroup (4 positions): rounds that start at a barrier, or nullptr
    const int      *pcol;
    const uint32_t *pmask4;
    const double   *pval;
    long long       b0_rows;   // rows of B0 / B1 the column indices can address (for the 4 GiB check)
    long long       b1_rows;
    // compact values (PanelHost::cmo / cbase / cval), or nullptr: the narrow-operand kernel reads them instead of pmask4 / pval
    const uint32_t *cmo;
    const long long *cbase;
    const double   *cval;
    bool            narrow64;  // take the narrow-operand kernel up to 64 columns (panels that are mostly holes)
};

struct TeamArgs
{
    int nteam;
    const int      *torder;
    const int      *tpanel;
    const int      *tptr;
    const int      *tcol;
    const uint32_t *tmask;
    const long long *tvoff;    // 4 * nteam: first entry of every wave's value stream
    const double   *tval;      // value streams, 8 values per own entry
};

struct Team2Args          // panel_format.h, Team2Host
{
    int nteam;
    int ngrid;                 // entries of torder: the launch grid (Team2Host::tgrid), a multiple of 8
    int tw;                    // waves per team: 8 or 16
    bool compact;              // value blocks hold only the values that exist (Team2Host::compact); false: 8 per part
    int pw;                    // panels per wave: 1, or 2 (Team2Host::P: teams of 16 panels on 8 waves, operands of one 16-byte piece)
    const int      *torder;
    const int      *tpanel;    // tw * pw * nteam
    const int      *tinfo;     // 4 * nteam: rounds, first record block, union entries, 0
    const int      *tpro;      // nteam * TEAM2_D * tw * 2: {column, value offset}
    const uint32_t *trec;      // record blocks (1 KiB each)
    const long long *tvoff;    // 8 * nteam
    const double   *tval;
    const float    *tval32;    // the same value groups in fp32 (fp32 path), or nullptr
    // generation start barrier (team2_kernel.hip): counters [column tile][XCD run][generation], or nullptr = none
    unsigned       *gsync;
    int             gsync_tiles;   // column tiles the counter array covers
    int             gsync_ngen;    // generations per run it covers
    int             wgs;           // teams of a generation (workgroups resident on an XCD)
    int             nreal[8];      // real teams of every run (the -1 entries sit at its end)
    // chains (Team2Host::chain > 0; team2p_kernel.hip): torder / tinfo / tpro / tvoff are per chain, nteam = chains
    int             chain = 0;
    int             nmember = 0;   // entries of cteam
    const int      *cptr = nullptr;    // chains + 1
    const int      *cteam = nullptr;   // the teams of the chains
    int            *trows = nullptr;   // nmember * tw * 8: C rows (team2p_fill_rows)
};

struct Team2NArgs         // panel_format.h, Team2NHost
{
    int G;                     // entries per instruction: 4 (n <= 32) or 2 (n <= 64)
    int nteam;
    int ngrid;                 // entries of tgrid, a multiple of 8
    const int      *tgrid;
    const int      *tpanel;    // 8 * nteam
    const int      *tinfo;     // 2 * nteam: rounds, first record (in rounds)
    const uint32_t *trec;      // 128 words per round
    const long long *tvoff;    // 8 * nteam
    const double   *tval;
    uint32_t       *tent;      // team2r: the entry table (Team2RHost::tent), or nullptr
    int             rowdma = 2; // team2r: row DMAs of a wave per round (Team2RHost::rowdma)
};

// narrow_kernel.hip: row-panel format, n <= 64 (several entries of a panel per instruction)
bool spmm_narrow_applicable(const PanelArgs &p, const SpmmArgs &a);
hipError_t spmm_rm_f64_narrow(const PanelArgs &p, const SpmmArgs &a, hipStream_t s);

// spmm_kernels.hip
hipError_t spmm_rm_f64_rowgroup(const SpmmArgs &a, hipStream_t s);
hipError_t spmm_cm_f64(const SpmmArgs &a, hipStream_t s);
bool spmm_panel_applicable(const SpmmArgs &a);
hipError_t spmm_rm_f64_panel(const PanelArgs &p, const SpmmArgs &a, hipStream_t s);
bool spmm_team_applicable(const SpmmArgs &a);
hipError_t spmm_rm_f64_team(const TeamArgs &t, const SpmmArgs &a, hipStream_t s);

// team2n_kernel.hip
bool spmm_team2n_applicable(const Team2NArgs &t, const SpmmArgs &a);
hipError_t spmm_rm_f64_team2n(const Team2NArgs &t, const SpmmArgs &a, hipStream_t s);

// team2r_kernel.hip (Team2RHost streams; the argument block is Team2NArgs: same arrays, tvoff in units of 16 bytes)
bool spmm_team2r_applicable(const Team2NArgs &t, const SpmmArgs &a);
hipError_t spmm_rm_f64_team2r(const Team2NArgs &t, const SpmmArgs &a, hipStream_t s);
hipError_t team2r_fill_rows(const Team2NArgs &t, const SpmmArgs &a, hipStream_t s);      // the C rows into the entry table: once per row map

// team2_kernel.hip
bool spmm_team2_applicable(const SpmmArgs &a);
hipError_t spmm_rm_f64_team2(const Team2Args &t, const SpmmArgs &a, hipStream_t s);
bool spmm_team2_applicable_f32(const SpmmArgsF32 &a);
hipError_t spmm_rm_f32_team2(const Team2Args &t, const SpmmArgsF32 &a, hipStream_t s);

// team2p_kernel.hip: the same streams laid out in chains, persistent workgroups
hipError_t spmm_rm_f64_team2p(const Team2Args &t, const SpmmArgs &a, hipStream_t s);
hipError_t spmm_rm_f32_team2p(const Team2Args &t, const SpmmArgsF32 &a, hipStream_t s);
hipError_t team2p_fill_rows(const Team2Args &t, int nrow, const int *rowmap, hipStream_t s);   // the C rows of the chains' panels: once per row map

// spmm_f32.hip: CSR row-group kernel of the fp32 path (any width, both sources)
hipError_t spmm_rm_f32_rowgroup(const SpmmArgsF32 &a, hipStream_t s);

// row_kernels.hip
hipError_t gather_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, int64_t lds,
                           double *dst, int64_t ldd, hipStream_t s);
hipError_t scatter_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, int64_t lds,
                            double *dst, int64_t ldd, hipStream_t s);
hipError_t scatter_vals_f64(int64_t n, const uint32_t *map, const double *src, double *dst, hipStream_t s);
hipError_t convert_f64_f32(int64_t n, const double *src, float *dst, hipStream_t s);
hipError_t transpose_f64(int nrow, int ncol, const double *src, int64_t lds, double *dst, int64_t ldd,
                         hipStream_t s);
hipError_t probe_copy(int64_t bytes, const void *src, void *dst, int blocks, unsigned long long *stamps, hipStream_t s);
hipError_t probe_stamp(unsigned long long *out, hipStream_t s);

}  // namespace crp
