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

struct Team2Args          // panel_format.h, Team2Host
{
    int nteam;
    int ngrid;                 // entries of torder: the launch grid (Team2Host::tgrid), a multiple of 8
    bool compact;              // value blocks hold only the values that exist (Team2Host::compact); false: 8 per part
    const int      *torder;
    const int      *tpanel;    // 8 * nteam
    const int      *tinfo;     // 4 * nteam: rounds, first record block, union entries, 0
    const int      *tpro;      // nteam * TEAM2_D * 8 * 2: {column, value offset}
    const uint32_t *trec;      // record blocks (1 KiB each)
    const long long *tvoff;    // 8 * nteam
    const double   *tval;
    const float    *tval32;    // the same value groups in fp32 (fp32 path), or nullptr
};

struct Team2NArgs         // panel_format.h, Team2RHost (the row-owner team kernel, variant 7)
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
    uint32_t       *tent;      // the entry table (Team2RHost::tent)
};

// narrow_kernel.hip: row-panel format, n <= 64 (several entries of a panel per instruction)
bool spmm_narrow_applicable(const PanelArgs &p, const SpmmArgs &a);
hipError_t spmm_rm_f64_narrow(const PanelArgs &p, const SpmmArgs &a, hipStream_t s);

// spmm_kernels.hip
hipError_t spmm_rm_f64_rowgroup(const SpmmArgs &a, hipStream_t s);
hipError_t spmm_cm_f64(const SpmmArgs &a, hipStream_t s);
bool spmm_panel_applicable(const SpmmArgs &a);
hipError_t spmm_rm_f64_panel(const PanelArgs &p, const SpmmArgs &a, hipStream_t s);

// team2r_kernel.hip (Team2RHost streams; the argument block is Team2NArgs: same arrays, tvoff in units of 16 bytes)
bool spmm_team2r_applicable(const Team2NArgs &t, const SpmmArgs &a);
hipError_t spmm_rm_f64_team2r(const Team2NArgs &t, const SpmmArgs &a, hipStream_t s);
hipError_t team2r_fill_rows(const Team2NArgs &t, const SpmmArgs &a, hipStream_t s);      // the C rows into the entry table: once per row map

// team2_kernel.hip
bool spmm_team2_applicable(const SpmmArgs &a);
hipError_t spmm_rm_f64_team2(const Team2Args &t, const SpmmArgs &a, hipStream_t s);
bool spmm_team2_applicable_f32(const SpmmArgsF32 &a);
hipError_t spmm_rm_f32_team2(const Team2Args &t, const SpmmArgsF32 &a, hipStream_t s);

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
// csr_mat_row_part_comm_size on a device-resident CSR (bits: nblk * ceil(ncol / 32) words, zeroed; comm_dev: nblk ints, zeroed)
hipError_t row_part_comm_size(int nrow, int ncol, const int *rowptr, const int *colidx, int nblk, const int *rblk_dev, const int *xd_dev,
                              unsigned *bits, int *comm_dev, int *bad_dev, hipStream_t s);
hipError_t probe_copy(int64_t bytes, const void *src, void *dst, int blocks, unsigned long long *stamps, hipStream_t s);
hipError_t probe_stamp(unsigned long long *out, hipStream_t s);

}  // namespace crp
