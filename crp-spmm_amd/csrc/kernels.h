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
    const int      *torder;
    const int      *tpanel;    // 8 * nteam
    const int      *tinfo;     // 4 * nteam: rounds, first record block, union entries, 0
    const int      *tpro;      // nteam * TEAM2_D * 8 * 2: {column, value offset}
    const uint32_t *trec;      // record blocks (1 KiB each)
    const long long *tvoff;    // 8 * nteam
    const double   *tval;
};

// spmm_kernels.hip
hipError_t spmm_rm_f64_rowgroup(const SpmmArgs &a, hipStream_t s);
hipError_t spmm_cm_f64(const SpmmArgs &a, hipStream_t s);
bool spmm_panel_applicable(const SpmmArgs &a);
hipError_t spmm_rm_f64_panel(const PanelArgs &p, const SpmmArgs &a, hipStream_t s);
bool spmm_team_applicable(const SpmmArgs &a);
hipError_t spmm_rm_f64_team(const TeamArgs &t, const SpmmArgs &a, hipStream_t s);

// team2_kernel.hip
bool spmm_team2_applicable(const SpmmArgs &a);
hipError_t spmm_rm_f64_team2(const Team2Args &t, const SpmmArgs &a, hipStream_t s);

// row_kernels.hip
hipError_t gather_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, int64_t lds,
                           double *dst, int64_t ldd, hipStream_t s);
hipError_t scatter_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, int64_t lds,
                            double *dst, int64_t ldd, hipStream_t s);
hipError_t scatter_vals_f64(int64_t n, const uint32_t *map, const double *src, double *dst, hipStream_t s);
hipError_t transpose_f64(int nrow, int ncol, const double *src, int64_t lds, double *dst, int64_t ldd,
                         hipStream_t s);

}  // namespace crp
