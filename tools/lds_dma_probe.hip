// Probe: global_load_lds_dwordx4 (LDS-DMA) semantics on gfx950 -- LDS destination = M0 base + lane*16,
// source address per lane; data visible to other waves after vmcnt + s_barrier.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_dma_probe tools/lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define GPTR(p) ((const __attribute__((address_space(1))) void *) (p))
#define LPTR(p) ((__attribute__((address_space(3))) void *) (p))

__global__ __launch_bounds__(256) void probe(const double *__restrict__ src, double *__restrict__ dst, int ld, int nrow)
{
    __shared__ __attribute__((aligned(16))) double ring[4][256];      // one 2 KiB row per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = (blockIdx.x * 4 + wave) % nrow;
    const char *g = reinterpret_cast<const char *>(src + (size_t) row * ld) + lane * 16;
    __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(&ring[wave][0]), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GPTR(g + 1024), LPTR(&ring[wave][128]), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // every wave reads the row loaded by the NEXT wave
    const int other = (wave + 1) & 3;
    const double2 v = *reinterpret_cast<const double2 *>(&ring[other][lane * 2]);
    const double2 w = *reinterpret_cast<const double2 *>(&ring[other][128 + lane * 2]);
    double *o = dst + (size_t) (blockIdx.x * 4 + wave) * 256;
    o[lane * 2] = v.x; o[lane * 2 + 1] = v.y; o[128 + lane * 2] = w.x; o[128 + lane * 2 + 1] = w.y;
}

int main()
{
    const int nrow = 5000, ld = 300, nblk = 2000;
    std::vector<double> h((size_t) nrow * ld);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double) i * 0.5 + 1.0;
    double *src, *dst;
    hipMalloc(&src, h.size() * 8);
    hipMalloc(&dst, (size_t) nblk * 4 * 256 * 8);
    hipMemcpy(src, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, 0, src, dst, ld, nrow);
    std::vector<double> out((size_t) nblk * 4 * 256);
    if (hipMemcpy(out.data(), dst, out.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    long bad = 0;
    for (int b = 0; b < nblk; b++)
        for (int w = 0; w < 4; w++)
        {
            const int row = (b * 4 + ((w + 1) & 3)) % nrow;
            for (int c = 0; c < 256; c++)
                if (out[((size_t) b * 4 + w) * 256 + c] != h[(size_t) row * ld + c]) bad++;
        }
    printf("lds_dma_probe: %ld mismatches of %zu\n", bad, out.size());
    return bad ? 1 : 0;
}
