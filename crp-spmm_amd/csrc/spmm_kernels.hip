// spmm_kernels.hip -- CSR x dense SpMM kernels for gfx950 (MI355X, wave64).
//
// Computes what mkl_sparse_d_mm computes at the reference call site
// /root/reference/src/rowpara_spmm.c:388-408 (alpha = 1, beta = 0):
//     C[i][0:n] = sum_p val[p] * B[col[p]][0:n],  p = rowptr[i] .. rowptr[i+1]-1
// The operation is HBM/L2-bound gather work: no MFMA here.  Each group of LPR
// lanes owns one row of A and a TW = LPR*VW*NV wide slice of C; the (col, val)
// pairs of the row are loaded coalesced LPR at a time and broadcast inside the
// group, every lane then streams 16-byte pieces of the addressed B rows.
//
// Two-source column index (see include/crpspmm_hip.h): c >= 0 -> B0 row c,
// c < 0 -> B1 row ~c.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace crp {

typedef double d2 __attribute__((ext_vector_type(2)));

template <int LPR>
__device__ __forceinline__ int bcast_i(int v, int j)
{
    if constexpr (LPR == 64) return __builtin_amdgcn_readlane(v, j);
    else return __shfl(v, j, LPR);
}

template <int LPR>
__device__ __forceinline__ double bcast_d(double v, int j)
{
    if constexpr (LPR == 64)
    {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
        return __hiloint2double(hi, lo);
    }
    else return __shfl(v, j, LPR);
}

// ---------------------------------------------------------------------------
// Row-major kernel.  LPR lanes per row, VW doubles per vector access (1 or 2),
// NV vector accesses per lane => tile width TW = LPR*VW*NV columns.
// grid.x covers rows (RPB = 256/LPR rows per block), grid.y covers column tiles.
// ---------------------------------------------------------------------------
template <int LPR, int VW, int NV>
__global__ __launch_bounds__(256) void spmm_rm_f64_kernel(
    const int nrow, const int n,
    const int *__restrict__ rowptr, const int *__restrict__ colidx, const double *__restrict__ val,
    const double *__restrict__ B0, const int64_t ldB0,
    const double *__restrict__ B1, const int64_t ldB1,
    double *__restrict__ C, const int64_t ldC)
{
    constexpr int RPB = 256 / LPR;
    constexpr int TW  = LPR * VW * NV;
    const int lir  = threadIdx.x % LPR;                 // lane in row group
    const int row  = blockIdx.x * RPB + threadIdx.x / LPR;
    const int col0 = blockIdx.y * TW + lir * VW;        // first column of this lane
    if (row >= nrow) return;

    double acc[NV][VW];
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int w = 0; w < VW; w++) acc[v][w] = 0.0;

    // Column slots of this lane.  Slots past n are clamped to column 0 for the LOADS
    // (always a valid address, so loads stay unconditional and can be batched) and are
    // simply not stored.
    bool ok[NV];
    int  coff[NV];
#pragma unroll
    for (int v = 0; v < NV; v++)
    {
        const int c = col0 + v * LPR * VW;
        ok[v]   = (c + VW - 1) < n;
        coff[v] = ok[v] ? c : 0;
    }

    int p0 = rowptr[row];
    const int pe = rowptr[row + 1];
    if constexpr (LPR == 64)
    {
        p0 = __builtin_amdgcn_readfirstlane(p0);
    }
    constexpr int UNR = 8;
    for (; p0 < pe; p0 += LPR)
    {
        const int my = p0 + lir;
        int    c = 0;
        double a = 0.0;
        if (my < pe)
        {
            c = colidx[my];
            a = val[my];
        }
        const int cnt = min(LPR, pe - p0);
        for (int j = 0; j < cnt; j += UNR)
        {
            // phase 1: issue every B-row load of this group (indices past the row end are
            // clamped to the row's last entry: a valid address whose data is not used)
            double aj[UNR];
            double bv[UNR][NV][VW];
#pragma unroll
            for (int u = 0; u < UNR; u++)
            {
                const int ju = min(j + u, cnt - 1);
                const int cj = bcast_i<LPR>(c, ju);
                aj[u] = bcast_d<LPR>(a, ju);
                const double *brow = (cj >= 0) ? (B0 + (int64_t) cj * ldB0)
                                               : (B1 + (int64_t) (~cj) * ldB1);
#pragma unroll
                for (int v = 0; v < NV; v++)
                {
                    if constexpr (VW == 2)
                    {
                        const d2 t = *reinterpret_cast<const d2 *>(brow + coff[v]);
                        bv[u][v][0] = t.x;
                        bv[u][v][1] = t.y;
                    }
                    else bv[u][v][0] = brow[coff[v]];
                }
            }
            // phase 2: accumulate in nonzero order; only real entries contribute
#pragma unroll
            for (int u = 0; u < UNR; u++)
            {
                if (j + u < cnt)
                {
#pragma unroll
                    for (int v = 0; v < NV; v++)
#pragma unroll
                        for (int w = 0; w < VW; w++) acc[v][w] = fma(aj[u], bv[u][v][w], acc[v][w]);
                }
            }
        }
    }

    double *crow = C + (int64_t) row * ldC;
#pragma unroll
    for (int v = 0; v < NV; v++)
    {
        if (ok[v])
        {
            if constexpr (VW == 2)
            {
                d2 r;
                r.x = acc[v][0];
                r.y = acc[v][1];
                __builtin_nontemporal_store(r, reinterpret_cast<d2 *>(crow + coff[v]));
            }
            else
            {
                __builtin_nontemporal_store(acc[v][0], crow + coff[v]);
            }
        }
    }
}

template <int LPR, int VW, int NV>
static hipError_t launch_rm(const SpmmArgs &a, hipStream_t s)
{
    constexpr int RPB = 256 / LPR;
    constexpr int TW  = LPR * VW * NV;
    dim3 grid((a.nrow + RPB - 1) / RPB, (a.n + TW - 1) / TW);
    hipLaunchKernelGGL((spmm_rm_f64_kernel<LPR, VW, NV>), grid, dim3(256), 0, s,
                       a.nrow, a.n, a.rowptr, a.colidx, a.val, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC);
    return hipGetLastError();
}

// Generic row-major dispatcher: picks the lane group / vector shape from n and
// from the 16-byte alignment of the operands.
hipError_t spmm_rm_f64_rowgroup(const SpmmArgs &a, hipStream_t s)
{
    const bool vec2 =
        (a.n % 2 == 0) && (a.ldB0 % 2 == 0) && (a.ldC % 2 == 0) && (a.B1 == nullptr || a.ldB1 % 2 == 0) &&
        (((uintptr_t) a.B0 | (uintptr_t) a.B1 | (uintptr_t) a.C) % 16 == 0);
    if (vec2)
    {
        if (a.n <= 8)   return launch_rm<4, 2, 1>(a, s);
        if (a.n <= 16)  return launch_rm<8, 2, 1>(a, s);
        if (a.n <= 32)  return launch_rm<16, 2, 1>(a, s);
        if (a.n <= 64)  return launch_rm<32, 2, 1>(a, s);
        if (a.n <= 128) return launch_rm<64, 2, 1>(a, s);
        return launch_rm<64, 2, 2>(a, s);
    }
    if (a.n <= 4)   return launch_rm<4, 1, 1>(a, s);
    if (a.n <= 8)   return launch_rm<8, 1, 1>(a, s);
    if (a.n <= 16)  return launch_rm<16, 1, 1>(a, s);
    if (a.n <= 32)  return launch_rm<32, 1, 1>(a, s);
    if (a.n <= 64)  return launch_rm<64, 1, 1>(a, s);
    return launch_rm<64, 1, 2>(a, s);
}

// ---------------------------------------------------------------------------
// Row-panel kernel (panel_format.h): one wavefront per panel of R rows and a
// TW = 128*NV wide slice of C.  For every panel entry (one column index shared by
// up to R rows) the wave loads the B row slice ONCE -- 16 bytes per lane per NV --
// and feeds it to the accumulators of the rows whose mask bit is set.  Entry
// indices, masks and values are wave-uniform, so they travel through scalar loads
// and SGPRs; the vector memory path carries only B and C.  Entries are consumed in
// groups of four with the next group's B loads issued before the current group's
// FMAs (two register sets, counted vmcnt), so ~8 KiB per wave stay in flight.
// ---------------------------------------------------------------------------
template <int R, int NV>
struct PanelGroup
{
    d2 b[4][NV];
};

template <int R, int NV>
__device__ __forceinline__ void panel_issue(PanelGroup<R, NV> &g, const int q, const int *__restrict__ pcol,
                                            const double *__restrict__ B0, const int64_t ldB0,
                                            const double *__restrict__ B1, const int64_t ldB1, const int (&coff)[NV])
{
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
        const int cj = pcol[q + u];     // uniform address -> scalar load
        const double *brow = (cj >= 0) ? (B0 + (int64_t) cj * ldB0) : (B1 + (int64_t) (~cj) * ldB1);
#pragma unroll
        for (int v = 0; v < NV; v++) g.b[u][v] = *reinterpret_cast<const d2 *>(brow + coff[v]);
    }
}

template <int R, int NV>
__device__ __forceinline__ void panel_consume(const PanelGroup<R, NV> &g, const int q,
                                              const uint32_t *__restrict__ pmask4, const double *__restrict__ pval,
                                              double (&acc)[R][NV][2])
{
    const uint32_t m4 = pmask4[q >> 2];
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
        const double *pv = pval + (int64_t) (q + u) * R;
        double a[R];
#pragma unroll
        for (int r = 0; r < R; r++) a[r] = pv[r];      // uniform -> SGPRs
#pragma unroll
        for (int r = 0; r < R; r++)
        {
            if (m4 & (1u << (8 * u + r)))              // wave-uniform branch: absent pairs cost no FMA
            {
#pragma unroll
                for (int v = 0; v < NV; v++)
                {
                    acc[r][v][0] = fma(a[r], g.b[u][v].x, acc[r][v][0]);
                    acc[r][v][1] = fma(a[r], g.b[u][v].y, acc[r][v][1]);
                }
            }
        }
    }
}

template <int R, int NV>
__global__ __launch_bounds__(256) void spmm_panel_f64_kernel(
    const int npanel, const int nrow, const int n,
    const int *__restrict__ pptr, const int *__restrict__ pcol, const uint32_t *__restrict__ pmask4,
    const double *__restrict__ pval,
    const double *__restrict__ B0, const int64_t ldB0, const double *__restrict__ B1, const int64_t ldB1,
    double *__restrict__ C, const int64_t ldC)
{
    constexpr int TW = 128 * NV;
    const int lane  = threadIdx.x & 63;
    const int panel = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (panel >= npanel) return;
    const int col0 = blockIdx.y * TW + lane * 2;
    bool ok[NV];
    int  coff[NV];
#pragma unroll
    for (int v = 0; v < NV; v++)
    {
        const int c = col0 + v * 128;
        ok[v]   = (c + 1) < n;
        coff[v] = ok[v] ? c : 0;
    }
    double acc[R][NV][2];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int v = 0; v < NV; v++) acc[r][v][0] = acc[r][v][1] = 0.0;

    int q = pptr[panel];
    const int qe = pptr[panel + 1];
    PanelGroup<R, NV> ga, gb;
    if (q < qe)
    {
        panel_issue<R, NV>(ga, q, pcol, B0, ldB0, B1, ldB1, coff);
        for (;;)
        {
            if (q + 4 < qe) panel_issue<R, NV>(gb, q + 4, pcol, B0, ldB0, B1, ldB1, coff);
            panel_consume<R, NV>(ga, q, pmask4, pval, acc);
            q += 4;
            if (q >= qe) break;
            if (q + 4 < qe) panel_issue<R, NV>(ga, q + 4, pcol, B0, ldB0, B1, ldB1, coff);
            panel_consume<R, NV>(gb, q, pmask4, pval, acc);
            q += 4;
            if (q >= qe) break;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
    {
        const int row = panel * R + r;
        if (row < nrow)
        {
            double *crow = C + (int64_t) row * ldC;
#pragma unroll
            for (int v = 0; v < NV; v++)
                if (ok[v])
                {
                    d2 t;
                    t.x = acc[r][v][0];
                    t.y = acc[r][v][1];
                    __builtin_nontemporal_store(t, reinterpret_cast<d2 *>(crow + coff[v]));
                }
        }
    }
}

template <int R, int NV>
static hipError_t launch_panel(const PanelArgs &p, const SpmmArgs &a, hipStream_t s)
{
    constexpr int TW = 128 * NV;
    dim3 grid((p.npanel + 3) / 4, (a.n + TW - 1) / TW);
    hipLaunchKernelGGL((spmm_panel_f64_kernel<R, NV>), grid, dim3(256), 0, s, p.npanel, a.nrow, a.n, p.pptr, p.pcol,
                       p.pmask4, p.pval, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC);
    return hipGetLastError();
}

bool spmm_panel_applicable(const SpmmArgs &a)
{
    return (a.n > 64) && (a.n % 2 == 0) && (a.ldB0 % 2 == 0) && (a.ldC % 2 == 0) &&
           (a.B1 == nullptr || a.ldB1 % 2 == 0) &&
           (((uintptr_t) a.B0 | (uintptr_t) a.B1 | (uintptr_t) a.C) % 16 == 0);
}

hipError_t spmm_rm_f64_panel(const PanelArgs &p, const SpmmArgs &a, hipStream_t s)
{
    const bool wide = a.n > 128;
    if (p.R == 4) return wide ? launch_panel<4, 2>(p, a, s) : launch_panel<4, 1>(p, a, s);
    if (p.R == 8) return wide ? launch_panel<8, 2>(p, a, s) : launch_panel<8, 1>(p, a, s);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// Column-major kernel (BC_layout = 1, /root/reference/src/rowpara_spmm.c:403
// with SPARSE_LAYOUT_COLUMN_MAJOR).  Element (r, j) of B/C sits at r + j*ld.
// One lane per (row, column) pair: 64 consecutive rows per wave so that the C
// store and the val/col loads are coalesced; B accesses are a true gather.
// The reference's own drivers never use this layout (examples/test_rp_spmm.c:109).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spmm_cm_f64_kernel(
    const int nrow, const int n,
    const int *__restrict__ rowptr, const int *__restrict__ colidx, const double *__restrict__ val,
    const double *__restrict__ B0, const int64_t ldB0,
    const double *__restrict__ B1, const int64_t ldB1,
    double *__restrict__ C, const int64_t ldC)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int j   = blockIdx.y;
    if (row >= nrow) return;
    const double *b0 = B0 + (int64_t) j * ldB0;
    const double *b1 = B1 + (int64_t) j * ldB1;
    double acc = 0.0;
    const int pe = rowptr[row + 1];
    for (int p = rowptr[row]; p < pe; p++)
    {
        const int c = colidx[p];
        const double b = (c >= 0) ? b0[c] : b1[~c];
        acc = fma(val[p], b, acc);
    }
    C[(int64_t) j * ldC + row] = acc;
}

hipError_t spmm_cm_f64(const SpmmArgs &a, hipStream_t s)
{
    dim3 grid((a.nrow + 255) / 256, a.n);
    hipLaunchKernelGGL(spmm_cm_f64_kernel, grid, dim3(256), 0, s,
                       a.nrow, a.n, a.rowptr, a.colidx, a.val, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC);
    return hipGetLastError();
}

}  // namespace crp
