// Micro-probe: v_fma_f64 throughput per SIMD on gfx950 at 1/2/4 waves per SIMD (diagnostic, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(double *out, int iters, double a, double b)
{
    double acc[16];
    for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3 + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = fma(acc[i], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 16; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (double) (t1 - t0) * 1e-300;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long *) out)[100000] = t1 - t0;
}
int main()
{
    double *d;
    hipMalloc(&d, 8 * 2000000);
    for (int wps = 1; wps <= 4; wps *= 2)
    {
        int threads = 256 * wps, blocks = 256;          // wps waves per SIMD on every CU
        int iters = 20000;
        k<<<blocks, threads>>>(d, 10, 1.000001, 1e-9);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<blocks, threads>>>(d, iters, 1.000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc;
        hipMemcpy(&cyc, ((unsigned long long *) d) + 100000, 8, hipMemcpyDeviceToHost);
        double fmas_per_simd = (double) iters * 16 * wps;    // wave-instructions per SIMD
        printf("waves/SIMD %d: %.3f ms, s_memtime ticks %llu (100MHz?), %.2f ns per wave-FMA per SIMD, TFLOP/s %.1f\n", wps, ms, cyc,
               ms * 1e6 / fmas_per_simd, 2.0 * 64 * iters * 16 * wps * 4 * 256 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
