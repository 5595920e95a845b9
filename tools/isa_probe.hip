// ISA probe for gfx950 (diagnostic, not product): the three hardware behaviours the LDS-sharing SpMM kernel
// (csrc/team2_kernel.hip) is built on.
//   1. v_fmac_f64_dpp ... row_newbcast:N  -- every lane reads lane N of its own row of 16 as the scalar factor:
//      semantics and issue rate against a plain v_fmac_f64;
//   2. global_load_lds_dwordx4 under a partial EXEC mask (lanes 0..31): only the active lanes' 16-byte pieces
//      are written, at M0 base + lane * 16;
//   3. the values -> LDS -> ds_read_b64 (address (lane & 7) * 8) -> DPP chain.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/isa_probe tools/isa_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define GPTR(p) ((const __attribute__((address_space(1))) void *) (p))
#define LPTR(p) ((__attribute__((address_space(3))) void *) (p))

__global__ void dpp_sem(const double *vals, const double *b, double *out)
{
    const int lane = threadIdx.x;
    double vv = vals[lane];                     // lane l holds vals[l]
    double bb = b[lane];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    asm volatile("s_nop 4\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %4, %5 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                 : "v"(vv), "v"(bb));
    for (int i = 0; i < 4; i++) out[i * 64 + lane] = acc[i];
}

template <bool DPP>
__global__ void rate(double *out, int iters, double a, double b)
{
    double acc[16];
    for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3 + i;
    double va = a + (threadIdx.x & 15) * 1e-9, vb = b;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            if constexpr (DPP)
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(va), "v"(vb));
            else
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(va), "v"(vb));
        }
    }
    double s = 0;
    for (int i = 0; i < 16; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(64) void masked_dma(const double *src, double *out)
{
    __shared__ __attribute__((aligned(16))) double buf[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) buf[i] = -1.0;
    __syncthreads();
    if (lane < 32) __builtin_amdgcn_global_load_lds(GPTR(reinterpret_cast<const char *>(src) + lane * 16), LPTR(&buf[0]), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = buf[i];
    // values chain: entry e = 8 doubles at buf[8 e ..]; lane reads (lane & 7), DPP row r multiplies by value r
    const double vv = buf[8 * 3 + (lane & 7)];           // entry 3
    double acc = 0.0, one = 1.0 + lane;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(vv), "v"(one));
    out[256 + lane] = acc;                              // expect buf[8*3+6] * (1 + lane)
}

// 4. global_load_lds_dwordx4 in the SADDR form (SGPR base + 32-bit VGPR offset + immediate): where does the immediate go?
//    expect: the immediate offset is added to BOTH the global address and the LDS address (M0 + offset + lane * 16).
__global__ __launch_bounds__(64) void saddr_dma(const double *src, double *out)
{
    __shared__ __attribute__((aligned(16))) double buf[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) buf[i] = -1.0;
    __syncthreads();
    const unsigned voff = lane * 16;
    const unsigned ldsb = (unsigned) (uintptr_t) &buf[0];
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\t"
                 "global_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                 "s_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(ldsb), "s"(src) : "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = buf[i];
}

// 5. fp32 inner block two ways (BASELINE configs[3] names an "MFMA B-tile path"): a block of 4 entries x 16 rows x 256
//    columns per wave, (a) as 16 x v_mfma_f32_16x16x4_f32 -- every (row, entry) pair multiplied, present or not --,
//    (b) as v_fmac_f32_dpp only for the present pairs (NPAIR of the 64: fill = NPAIR / 64), 4 FMAs per pair.
typedef float f4x __attribute__((ext_vector_type(4)));
__global__ void blk_mfma(float *out, int iters)
{
    f4x acc[16];
    for (int i = 0; i < 16; i++) acc[i] = (f4x) (threadIdx.x * 1e-3f + i);
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f4x s = acc[0];
    for (int i = 1; i < 16; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int NPAIR>
__global__ void blk_fma(float *out, int iters)
{
    float acc[64];
    for (int i = 0; i < 64; i++) acc[i] = threadIdx.x * 1e-3f + i;
    float va = 1.0f + (threadIdx.x & 15) * 1e-6f, vb = 0.5f;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int p = 0; p < NPAIR; p++)
#pragma unroll
            for (int w = 0; w < 4; w++)
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[(p * 4 + w) & 63]) : "v"(va), "v"(vb));
    }
    float s = 0;
    for (int i = 0; i < 64; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    int rc = 0;
    // ---- 1. semantics
    {
        std::vector<double> hv(64), hb(64), ho(256);
        for (int i = 0; i < 64; i++) { hv[i] = 100.0 + i; hb[i] = 1.0 + 0.5 * i; }
        double *dv, *db, *dout;
        hipMalloc(&dv, 512); hipMalloc(&db, 512); hipMalloc(&dout, 2048);
        hipMemcpy(dv, hv.data(), 512, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
        dpp_sem<<<1, 64>>>(dv, db, dout);
        hipMemcpy(ho.data(), dout, 2048, hipMemcpyDeviceToHost);
        const int sel[4] = {0, 3, 7, 15};
        long bad = 0;
        for (int i = 0; i < 4; i++)
            for (int l = 0; l < 64; l++)
            {
                const double expect = hv[(l & ~15) + sel[i]] * hb[l];
                if (ho[i * 64 + l] != expect) { if (bad < 4) printf("  dpp lane %d sel %d: got %g expect %g\n", l, sel[i], ho[i * 64 + l], expect); bad++; }
            }
        printf("dpp row_newbcast semantics: %ld mismatches of 256\n", bad);
        if (bad) rc = 1;
    }
    // ---- 2. rate
    {
        double *d;
        hipMalloc(&d, 8 * 2000000);
        for (int dpp = 0; dpp < 2; dpp++)
            for (int wps = 1; wps <= 4; wps *= 2)
            {
                const int threads = 256 * wps, blocks = 256, iters = 20000;
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; rep++)
                {
                    hipEventRecord(e0);
                    if (dpp) rate<true><<<blocks, threads>>>(d, iters, 1.000001, 1e-9);
                    else rate<false><<<blocks, threads>>>(d, iters, 1.000001, 1e-9);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%s waves/SIMD %d: %.3f ms, %.2f ns per wave-FMA per SIMD, %.1f TFLOP/s\n", dpp ? "fmac_f64_dpp" : "fmac_f64    ", wps, ms,
                       ms * 1e6 / ((double) iters * 16 * wps), 2.0 * 64 * iters * 16 * wps * 4 * 256 / (ms * 1e-3) / 1e12);
            }
    }
    // ---- 3. masked LDS-DMA + value chain
    {
        std::vector<double> hs(256), ho(320);
        for (int i = 0; i < 256; i++) hs[i] = 7.0 + i;
        double *ds, *dout;
        hipMalloc(&ds, 2048); hipMalloc(&dout, 320 * 8);
        hipMemcpy(ds, hs.data(), 2048, hipMemcpyHostToDevice);
        masked_dma<<<1, 64>>>(ds, dout);
        hipMemcpy(ho.data(), dout, 320 * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < 256; i++)
        {
            const double expect = i < 64 ? hs[i] : -1.0;       // 32 lanes x 16 B = 64 doubles
            if (ho[i] != expect) { if (bad < 4) printf("  masked dma [%d]: got %g expect %g\n", i, ho[i], expect); bad++; }
        }
        for (int l = 0; l < 64; l++)
            if (ho[256 + l] != hs[8 * 3 + 6] * (1.0 + l)) { if (bad < 8) printf("  value chain lane %d: got %g\n", l, ho[256 + l]); bad++; }
        printf("masked LDS-DMA (32 lanes) + value chain: %ld mismatches\n", bad);
        if (bad) rc = 1;
    }
    // ---- 4. saddr form + immediate offset
    {
        std::vector<double> hs(512), ho(512);
        for (int i = 0; i < 512; i++) hs[i] = 3.0 + i;
        double *ds, *dout;
        hipMalloc(&ds, 4096); hipMalloc(&dout, 4096);
        hipMemcpy(ds, hs.data(), 4096, hipMemcpyHostToDevice);
        saddr_dma<<<1, 64>>>(ds, dout);
        hipMemcpy(ho.data(), dout, 4096, hipMemcpyDeviceToHost);
        long both = 0, globonly = 0;
        for (int i = 0; i < 128; i++) if (ho[i] != hs[i]) both++, globonly++;
        for (int i = 128; i < 256; i++) { if (ho[i] != hs[i]) both++; }
        // alternative: offset applied to the global address only -> second load overwrites LDS [0,1024) with src[128..255]
        long alt = 0;
        for (int i = 0; i < 128; i++) if (ho[i] != hs[128 + i]) alt++;
        printf("saddr LDS-DMA with offset:1024: mismatches if offset moves both addresses %ld; if only the global address %ld (ho[0]=%g ho[128]=%g)\n",
               both, alt, ho[0], ho[128]);
        if (both != 0) rc = rc ? rc : 0;        // informational: the kernel picks the form that matches
    }
    // ---- 5. fp32 block: MFMA against masked FMAs
    {
        float *d;
        hipMalloc(&d, 4 * 2000000);
        const int threads = 1024, blocks = 256, iters = 4000;       // 4 waves per SIMD
        auto timeit = [&](auto launch) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; rep++)
            {
                hipEventRecord(e0);
                launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            return ms;
        };
        const double blocks_per_simd = (double) iters * 4;          // blocks (4 entries x 16 rows x 256 columns) per SIMD
        const float t_mfma = timeit([&] { blk_mfma<<<blocks, threads>>>(d, iters); });
        const float t_f64 = timeit([&] { blk_fma<64><<<blocks, threads>>>(d, iters); });
        const float t_f38 = timeit([&] { blk_fma<38><<<blocks, threads>>>(d, iters); });
        const float t_f32 = timeit([&] { blk_fma<32><<<blocks, threads>>>(d, iters); });
        printf("fp32 block of 4 entries x 16 rows x 256 columns, ns per block per SIMD (4 waves per SIMD):\n"
               "  16 x v_mfma_f32_16x16x4_f32 (all 64 pairs)      %.1f\n  v_fmac_f32_dpp, fill 1.00 (64 pairs, 256 FMAs) %.1f\n"
               "  v_fmac_f32_dpp, fill 0.60 (38 pairs)            %.1f\n  v_fmac_f32_dpp, fill 0.50 (32 pairs)            %.1f\n",
               t_mfma * 1e6 / blocks_per_simd, t_f64 * 1e6 / blocks_per_simd, t_f38 * 1e6 / blocks_per_simd, t_f32 * 1e6 / blocks_per_simd);
    }
    return rc;
}
