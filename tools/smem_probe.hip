// smem_probe.hip -- how many 128-byte lines per second can the SCALAR path pull into L2?  (Idea: scalar loads as L2
// prefetches for rows the vector path will fetch later: the scalar cache has its own miss queue, so the prefetches would
// not hold the vector L1's request slots for an HBM latency.)  Every wave touches one dword of `lines` distinct 128-byte
// lines of a buffer far larger than the caches, 16 s_load in flight.
// build: hipcc --offload-arch=gfx950 -O3 -o smem_probe tools/smem_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void touch(const char *base, const long long lines_per_wave, const long long stride, unsigned *sink)
{
    const long long wave = (long long) blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const char *p = base + wave * lines_per_wave * stride;
    unsigned acc = 0;
    for (long long i = 0; i < lines_per_wave; i += 16)
    {
        unsigned t[16];
#pragma unroll
        for (int j = 0; j < 16; j++)
        {
            const unsigned long long a = (unsigned long long) (p + (i + j) * stride);
            const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) a);
            const unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (a >> 32));
            const unsigned long long q = ((unsigned long long) hi << 32) | lo;
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(t[j]) : "s"(q));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; j++) acc += t[j];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void vtouch(const char *base, const long long lines_per_wave, const long long stride, unsigned *sink)
{
    // the same lines by the vector path: lane l reads dword l of line (i + l / 32)... one 128-byte line per 32 lanes
    const long long wave = ((long long) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const char *p = base + wave * lines_per_wave * stride;
    unsigned acc = 0;
    for (long long i = 0; i < lines_per_wave; i += 2)
        acc += *reinterpret_cast<const unsigned *>(p + (i + (lane >> 5)) * stride + (lane & 31) * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const long long stride = 128, nwaves = 256LL * 8 * 4, lines_per_wave = 4096;      // 8 workgroups of 4 waves per CU
    const long long bytes = nwaves * lines_per_wave * stride;                            // 4 GiB
    char *buf;
    unsigned *sink;
    if (hipMalloc(&buf, (size_t) bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, (size_t) bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pass = 0; pass < 2; pass++)
        for (int kind = 0; kind < 2; kind++)
        {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(touch, dim3((unsigned) (nwaves / 4)), dim3(256), 0, 0, buf, lines_per_wave, stride, sink);
            else hipLaunchKernelGGL(vtouch, dim3((unsigned) (nwaves / 4)), dim3(256), 0, 0, buf, lines_per_wave, stride, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double lines = (double) nwaves * lines_per_wave;
            printf("%s pass %d: %.3f ms, %.1f G lines/s, %.2f TB/s of 128-byte lines, %.3f lines per cycle and CU (2.4 GHz)\n",
                   kind == 0 ? "scalar s_load_dword" : "vector dword      ", pass, ms, lines / ms / 1e6, lines * 128 / ms / 1e9,
                   lines / (ms * 1e-3) / 2.4e9 / 256);
        }
    return 0;
}
