// spmm_f32.hip -- CSR row-group kernel of the fp32 path (values, B and C in fp32; BASELINE configs[3] "Queen_4147
// n = 1024, fp32").  The reference has no fp32 arithmetic (double *A_val, /root/reference/src/rowpara_spmm.h:28);
// this is the same product, C[i][:] = sum_p val[p] * B[col[p]][:] in ascending p with fp32 FMAs, for every width
// and alignment -- the fallback of the fp32 path where the LDS-sharing team kernel (team2_kernel.hip, fp32
// instance) does not apply.  LPR lanes own one row and LPR * VW columns of a tile; (col, val) pairs are loaded
// LPR at a time and broadcast inside the group; absent pairs do not exist in CSR, so nothing is ever multiplied by
// a padding zero.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace crp {

typedef float f4v __attribute__((ext_vector_type(4)));

template <int LPR, int VW>
__global__ __launch_bounds__(256) void spmm_rm_f32_kernel(const int nrow, const int n, const int *__restrict__ rowptr,
                                                          const int *__restrict__ colidx, const float *__restrict__ val,
                                                          const float *__restrict__ B0, const int64_t ldB0,
                                                          const float *__restrict__ B1, const int64_t ldB1,
                                                          float *__restrict__ C, const int64_t ldC, const int *__restrict__ rowmap)
{
    constexpr int RPB = 256 / LPR;
    constexpr int TW = LPR * VW;
    const int lir = threadIdx.x % LPR;
    const int row = blockIdx.x * RPB + threadIdx.x / LPR;
    const int col0 = blockIdx.y * TW + lir * VW;
    if (row >= nrow) return;
    const bool ok = (col0 + VW - 1) < n;
    const int coff = ok ? col0 : 0;                     // lanes past n load column 0 (valid) and store nothing
    float acc[VW];
#pragma unroll
    for (int w = 0; w < VW; w++) acc[w] = 0.0f;
    const int pe = rowptr[row + 1];
    for (int p0 = rowptr[row]; p0 < pe; p0 += LPR)
    {
        const int my = p0 + lir;
        int c = 0;
        float a = 0.0f;
        if (my < pe) { c = colidx[my]; a = val[my]; }
        const int cnt = min(LPR, pe - p0);
        for (int j = 0; j < cnt; j++)
        {
            const int cj = __shfl(c, j, LPR);
            const float aj = __shfl(a, j, LPR);
            const float *brow = (cj >= 0) ? (B0 + (int64_t) cj * ldB0) : (B1 + (int64_t) (~cj) * ldB1);
            if constexpr (VW == 4)
            {
                const f4v t = *reinterpret_cast<const f4v *>(brow + coff);
#pragma unroll
                for (int w = 0; w < 4; w++) acc[w] = fmaf(aj, t[w], acc[w]);
            }
            else acc[0] = fmaf(aj, brow[coff], acc[0]);
        }
    }
    if (!ok) return;
    float *crow = C + (int64_t) (rowmap ? rowmap[row] : row) * ldC;
    if constexpr (VW == 4)
    {
        f4v t;
#pragma unroll
        for (int w = 0; w < 4; w++) t[w] = acc[w];
        __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(crow + coff));
    }
    else __builtin_nontemporal_store(acc[0], crow + coff);
}

template <int LPR, int VW>
static hipError_t launch_f32(const SpmmArgsF32 &a, hipStream_t s)
{
    constexpr int RPB = 256 / LPR, TW = LPR * VW;
    dim3 grid((a.nrow + RPB - 1) / RPB, (a.n + TW - 1) / TW);
    hipLaunchKernelGGL((spmm_rm_f32_kernel<LPR, VW>), grid, dim3(256), 0, s, a.nrow, a.n, a.rowptr, a.colidx, a.val, a.B0, a.ldB0,
                       a.B1, a.ldB1, a.C, a.ldC, a.rowmap);
    return hipGetLastError();
}

hipError_t spmm_rm_f32_rowgroup(const SpmmArgsF32 &a, hipStream_t s)
{
    const bool vec4 = (a.n % 4 == 0) && (a.ldB0 % 4 == 0) && (a.ldC % 4 == 0) && (a.B1 == nullptr || a.ldB1 % 4 == 0) &&
                      (((uintptr_t) a.B0 | (uintptr_t) a.B1 | (uintptr_t) a.C) % 16 == 0);
    if (vec4)
    {
        if (a.n <= 16) return launch_f32<4, 4>(a, s);
        if (a.n <= 32) return launch_f32<8, 4>(a, s);
        if (a.n <= 64) return launch_f32<16, 4>(a, s);
        if (a.n <= 128) return launch_f32<32, 4>(a, s);
        return launch_f32<64, 4>(a, s);
    }
    if (a.n <= 4) return launch_f32<4, 1>(a, s);
    if (a.n <= 8) return launch_f32<8, 1>(a, s);
    if (a.n <= 16) return launch_f32<16, 1>(a, s);
    if (a.n <= 32) return launch_f32<32, 1>(a, s);
    return launch_f32<64, 1>(a, s);
}

}  // namespace crp
