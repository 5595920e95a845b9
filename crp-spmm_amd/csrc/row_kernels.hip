// row_kernels.hip -- index-list row gather / scatter and transpose for gfx950.
//
// These replace the OpenMP pack / unpack loops of the reference's B exchange
// (/root/reference/src/rowpara_spmm.c:232-262 pack, :313-344 unpack) and the
// strided rectangle copy of /root/reference/src/utils.c:92-119.  Pure byte
// movement: 16-byte accesses per lane, consecutive lanes on consecutive
// addresses, grid-stride so that a launch is capped at ~8 blocks per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace crp {

typedef double d2 __attribute__((ext_vector_type(2)));

static inline int grid_for(int64_t work)
{
    int64_t blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int) blocks;
}

// Row-major: rows are contiguous runs of n doubles.  VW = 2 uses 16-byte accesses.
// GATHER: dst[i] = src[ridx[i]];  !GATHER: dst[ridx[i]] = src[i].
template <int VW, bool GATHER>
__global__ __launch_bounds__(256) void move_rows_rm_kernel(
    const int64_t nidx, const int cpr /* chunks per row */, const int *__restrict__ ridx,
    const double *__restrict__ src, const int64_t lds, double *__restrict__ dst, const int64_t ldd)
{
    const int64_t total = nidx * (int64_t) cpr;
    for (int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t) gridDim.x * 256)
    {
        const int64_t i = t / cpr;
        const int     c = (int) (t - i * cpr) * VW;
        const int64_t r = ridx[i];
        const int64_t srow = GATHER ? r : i;
        const int64_t drow = GATHER ? i : r;
        if constexpr (VW == 2)
        {
            const d2 v = *reinterpret_cast<const d2 *>(src + srow * lds + c);
            *reinterpret_cast<d2 *>(dst + drow * ldd + c) = v;
        }
        else
        {
            dst[drow * ldd + c] = src[srow * lds + c];
        }
    }
}

// Column-major: element (r, j) at r + j*ld; consecutive lanes walk the index list.
template <bool GATHER>
__global__ __launch_bounds__(256) void move_rows_cm_kernel(
    const int64_t nidx, const int n, const int *__restrict__ ridx,
    const double *__restrict__ src, const int64_t lds, double *__restrict__ dst, const int64_t ldd)
{
    const int64_t total = nidx * (int64_t) n;
    for (int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t) gridDim.x * 256)
    {
        const int64_t j = t / nidx;
        const int64_t i = t - j * nidx;
        const int64_t r = ridx[i];
        if (GATHER) dst[i + j * ldd] = src[r + j * lds];
        else        dst[r + j * ldd] = src[i + j * lds];
    }
}

template <bool GATHER>
static hipError_t move_rows(int layout, int nidx, int n, const int *ridx, const double *src, int64_t lds,
                            double *dst, int64_t ldd, hipStream_t s)
{
    if (nidx <= 0 || n <= 0) return hipSuccess;
    if (layout == 0)
    {
        const bool vec2 = (n % 2 == 0) && (lds % 2 == 0) && (ldd % 2 == 0) &&
                          (((uintptr_t) src | (uintptr_t) dst) % 16 == 0);
        if (vec2)
        {
            const int cpr = n / 2;
            hipLaunchKernelGGL((move_rows_rm_kernel<2, GATHER>), dim3(grid_for((int64_t) nidx * cpr)), dim3(256), 0, s,
                               (int64_t) nidx, cpr, ridx, src, lds, dst, ldd);
        }
        else
        {
            hipLaunchKernelGGL((move_rows_rm_kernel<1, GATHER>), dim3(grid_for((int64_t) nidx * n)), dim3(256), 0, s,
                               (int64_t) nidx, n, ridx, src, lds, dst, ldd);
        }
    }
    else
    {
        hipLaunchKernelGGL((move_rows_cm_kernel<GATHER>), dim3(grid_for((int64_t) nidx * n)), dim3(256), 0, s,
                           (int64_t) nidx, n, ridx, src, lds, dst, ldd);
    }
    return hipGetLastError();
}

hipError_t gather_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, int64_t lds,
                           double *dst, int64_t ldd, hipStream_t s)
{
    return move_rows<true>(layout, nidx, n, ridx, src, lds, dst, ldd, s);
}

hipError_t scatter_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, int64_t lds,
                            double *dst, int64_t ldd, hipStream_t s)
{
    return move_rows<false>(layout, nidx, n, ridx, src, lds, dst, ldd, s);
}

// 32x32 tile transpose through LDS; 33-double row pitch keeps the column reads
// off a single bank.  dst[c][r] = src[r][c].
__global__ __launch_bounds__(256) void transpose_f64_kernel(
    const int nrow, const int ncol, const double *__restrict__ src, const int64_t lds,
    double *__restrict__ dst, const int64_t ldd)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;   // 32 x 8
    const int64_t r0 = (int64_t) blockIdx.x * 32, c0 = (int64_t) blockIdx.y * 32;   // rows on grid.x (2^31 limit)
#pragma unroll
    for (int k = 0; k < 32; k += 8)
    {
        const int64_t r = r0 + ty + k, c = c0 + tx;
        if (r < nrow && c < ncol) tile[ty + k][tx] = src[r * lds + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8)
    {
        const int64_t c = c0 + ty + k, r = r0 + tx;
        if (r < nrow && c < ncol) dst[c * ldd + r] = tile[tx][ty + k];
    }
}

hipError_t transpose_f64(int nrow, int ncol, const double *src, int64_t lds, double *dst, int64_t ldd,
                         hipStream_t s)
{
    if (nrow <= 0 || ncol <= 0) return hipSuccess;
    dim3 grid((nrow + 31) / 32, (ncol + 31) / 32);
    hipLaunchKernelGGL(transpose_f64_kernel, grid, dim3(256), 0, s, nrow, ncol, src, lds, dst, ldd);
    return hipGetLastError();
}

// dst[map[i]] = src[i]: refresh the values of a derived sparse format from new CSR values
__global__ __launch_bounds__(256) void scatter_vals_kernel(const int64_t n, const uint32_t *__restrict__ map,
                                                           const double *__restrict__ src, double *__restrict__ dst)
{
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) dst[map[i]] = src[i];
}

hipError_t scatter_vals_f64(int64_t n, const uint32_t *map, const double *src, double *dst, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_vals_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, map, src, dst);
    return hipGetLastError();
}

// fp32 copy of fp64 values (the fp32 path keeps A's values in fp64 as the caller gave them and derives its own copy)
__global__ void convert_f64_f32_kernel(const int64_t n, const double *__restrict__ src, float *__restrict__ dst)
{
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) dst[i] = (float) src[i];
}

hipError_t convert_f64_f32(int64_t n, const double *src, float *dst, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int blocks = (int) ((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(convert_f64_f32_kernel, dim3(blocks), dim3(256), 0, s, n, src, dst);
    return hipGetLastError();
}

// ---- exchange-overlap probe (tools/overlap_probe.py; SURVEY 8(e): does a transfer kernel make progress beside the product?)
// A stand-in for the transport's copy kernels: `blocks` workgroups of 256 threads copy n16 16-byte words and stamp the
// 100 MHz wall clock: stamps[0] = the earliest start of a workgroup, stamps[1] = the latest end.
__global__ __launch_bounds__(256) void probe_copy_kernel(const int64_t n16, const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                          unsigned long long *__restrict__ stamps)
{
    if (threadIdx.x == 0) atomicMin(&stamps[0], (unsigned long long) wall_clock64());
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t) gridDim.x * 256) dst[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&stamps[1], (unsigned long long) wall_clock64());
}

__global__ void probe_stamp_kernel(unsigned long long *__restrict__ out) { *out = (unsigned long long) wall_clock64(); }

hipError_t probe_copy(int64_t bytes, const void *src, void *dst, int blocks, unsigned long long *stamps, hipStream_t s)
{
    if (bytes <= 0 || blocks <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(probe_copy_kernel, dim3(blocks), dim3(256), 0, s, bytes / 16, (const uint4 *) src, (uint4 *) dst, stamps);
    return hipGetLastError();
}

hipError_t probe_stamp(unsigned long long *out, hipStream_t s)
{
    hipLaunchKernelGGL(probe_stamp_kernel, dim3(1), dim3(1), 0, s, out);
    return hipGetLastError();
}

}  // namespace crp
