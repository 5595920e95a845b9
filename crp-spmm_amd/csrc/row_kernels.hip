// row_kernels.hip -- index-list row gather / scatter and transpose for gfx950.
//
// These replace the OpenMP pack / unpack loops of the reference's B exchange
// (/root/reference/src/rowpara_spmm.c:232-262 pack, :313-344 unpack) and the
// strided rectangle copy of /root/reference/src/utils.c:92-119.  Pure byte
// movement: 16-byte accesses per lane, consecutive lanes on consecutive
// addresses, grid-stride so that a launch is capped at ~8 blocks per CU.
#include <hip/hip_runtime.h>
#include <algorithm>
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

// ---- csr_mat_row_part_comm_size on the device (planner, /root/reference/src/spmat_part.c:38-64) -------------------------
// For a row partition rblk (nblk + 1) and a partition xd of the columns (nblk + 1): comm[b] = distinct columns the rows of
// block b name that lie outside [xd[b], xd[b + 1]).  One bitmap of ncol bits per block: pass 1 sets the bit of every nonzero
// (one wave per row; the row's block by binary search), pass 2 counts the set bits outside the block's own range.
// Integer work: bit-exact against the reference.
__global__ void comm_mark_kernel(const int nrow, const int *__restrict__ rowptr, const int *__restrict__ colidx, const int nblk,
                                 const int *__restrict__ rblk, const long long words, unsigned *__restrict__ bits, int *__restrict__ bad)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long) blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((long long) gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < nrow; r += nwave)
    {
        int lo = 0, hi = nblk;                       // rblk[lo] <= r < rblk[hi]: the last block that starts at or before r
        while (hi - lo > 1)
        {
            const int mid = (lo + hi) >> 1;
            if (rblk[mid] <= (int) r) lo = mid; else hi = mid;
        }
        unsigned *my = bits + (long long) lo * words;
        for (int p = rowptr[r] + lane; p < rowptr[r + 1]; p += 64)
        {
            const int c = colidx[p];
            if (c < 0) { *bad = 1; continue; }       // (a two-source index: not a column of the global matrix)
            atomicOr(my + (c >> 5), 1u << (c & 31));
        }
    }
}

__global__ void comm_count_kernel(const int ncol, const int nblk, const int *__restrict__ xd, const long long words,
                                  const unsigned *__restrict__ bits, int *__restrict__ comm)
{
    const int b = blockIdx.y;
    const int own0 = xd[b], own1 = xd[b + 1];
    int cnt = 0;
    for (long long w = (long long) blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (long long) gridDim.x * blockDim.x)
    {
        unsigned v = bits[(long long) b * words + w];
        if (v == 0) continue;
        const long long c0 = w << 5;
        // clear the bits of the block's own columns [own0, own1)
        if (c0 + 32 > own0 && c0 < own1)
        {
            const int lo = (int) std::max<long long>(own0 - c0, 0), hi = (int) std::min<long long>(own1 - c0, 32);
            const unsigned m = (hi - lo >= 32) ? 0xFFFFFFFFu : (((1u << (hi - lo)) - 1u) << lo);
            v &= ~m;
        }
        cnt += __popc(v);
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(comm + b, cnt);
    (void) ncol;
}

hipError_t row_part_comm_size(int nrow, int ncol, const int *rowptr, const int *colidx, int nblk, const int *rblk_dev, const int *xd_dev,
                              unsigned *bits, int *comm_dev, int *bad_dev, hipStream_t s)
{
    const long long words = ((long long) ncol + 31) / 32;
    if (nrow > 0)
    {
        const int blocks = (int) std::min<long long>(((long long) nrow + 3) / 4, 65536);
        hipLaunchKernelGGL(comm_mark_kernel, dim3(blocks), dim3(256), 0, s, nrow, rowptr, colidx, nblk, rblk_dev, words, bits, bad_dev);
    }
    if (words > 0)
    {
        const int bx = (int) std::min<long long>((words + 255) / 256, 1024);
        hipLaunchKernelGGL(comm_count_kernel, dim3(bx, nblk), dim3(256), 0, s, ncol, nblk, xd_dev, words, bits, comm_dev);
    }
    return hipGetLastError();
}

}  // namespace crp
