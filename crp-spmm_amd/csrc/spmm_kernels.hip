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

    bool ok[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) ok[v] = (col0 + v * LPR * VW + VW - 1) < n;

    int p0 = rowptr[row];
    const int pe = rowptr[row + 1];
    if constexpr (LPR == 64)
    {
        p0 = __builtin_amdgcn_readfirstlane(p0);
    }
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
        // one (col, val) pair: broadcast it inside the row group, stream the B row slice
        auto step = [&](const int j) {
            const int    cj = bcast_i<LPR>(c, j);
            const double aj = bcast_d<LPR>(a, j);
            const double *brow = (cj >= 0) ? (B0 + (int64_t) cj * ldB0)
                                           : (B1 + (int64_t) (~cj) * ldB1);
            brow += col0;
#pragma unroll
            for (int v = 0; v < NV; v++)
            {
                if (ok[v])
                {
                    if constexpr (VW == 2)
                    {
                        const d2 b = *reinterpret_cast<const d2 *>(brow + v * LPR * VW);
                        acc[v][0] = fma(aj, b.x, acc[v][0]);
                        acc[v][1] = fma(aj, b.y, acc[v][1]);
                    }
                    else
                    {
                        acc[v][0] = fma(aj, brow[v * LPR * VW], acc[v][0]);
                    }
                }
            }
        };
        // hand-unrolled by UNR (the cross-lane broadcasts are convergent ops, which
        // stops hipcc from unrolling a runtime-trip-count loop by itself)
        constexpr int UNR = 8;
        int j = 0;
        for (; j + UNR <= cnt; j += UNR)
        {
#pragma unroll
            for (int u = 0; u < UNR; u++) step(j + u);
        }
        for (; j < cnt; j++) step(j);
    }

    double *crow = C + (int64_t) row * ldC + col0;
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
                __builtin_nontemporal_store(r, reinterpret_cast<d2 *>(crow + v * LPR * VW));
            }
            else
            {
                __builtin_nontemporal_store(acc[v][0], crow + v * LPR * VW);
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
