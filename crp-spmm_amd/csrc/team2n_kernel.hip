// team2n_kernel.hip -- team SpMM for narrow operands (24 <= n <= 64 columns, fp64) on gfx950.
//
// Same product as every kernel of this library (what mkl_sparse_d_mm computes at /root/reference/src/rowpara_spmm.c:388-408
// with alpha = 1, beta = 0).  Narrow operands are what the planner's 1 x P grids hand every GPU (n / P columns, all rows), so
// this is the kernel a multi-GPU run spends its time in.
//
// Two ideas, each from a kernel that has one of them:
//   * team2_kernel.hip: a workgroup of 8 waves owns 8 row panels (64 rows); every B row slice the team needs is fetched ONCE,
//     by LDS-DMA into a ring, and read from LDS by every panel that uses it.  The row-panel kernels request a B row slice
//     once per PANEL entry instead: 11 against 5 requests per B row on the pwtk stand-in, 10.9 against 4.4 on the nlpkkt
//     stand-in -- and at n <= 64 those kernels are bound by exactly that (the vector memory pipe's request rate: 608 M
//     64-byte... lines for nlpkkt240 at n = 32 is 11 ms of the per-CU request slots, measured 8.9 ms).
//   * narrow_kernel.hip: a slice of 256 (n <= 32) or 512 bytes (n <= 64) fills a quarter or half of a wave, so G = 4 or 2
//     entries are taken PER INSTRUCTION: lane group q holds entry q -- its slice (16 bytes per lane), its row mask, its values
//     in the lanes' low three bits -- a row's FMA is one v_fmac_f64_dpp per column of the lane for all G entries, the scalar
//     factor through DPP row_newbcast, absent rows switched off through EXEC (so an absent pair is never multiplied).
// Format: panel_format.h, Team2NHost.  A round has 8 G slots; wave w fetches slots w G .. w G + G - 1 with one DMA instruction
// (1 KiB) and its own compact value block with another; three rounds are in flight (ring of four sets: 32 KiB of slices,
// 32 KiB of values -> two workgroups per CU).  One barrier per round = per 8 G union entries (team2: per 8).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"
#include "narrow_rows.inc"

namespace crp {

namespace {
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int T2N_NSET = 4, T2N_D = 3, T2N_SETB = 8192, T2N_VSLOT = 1024, T2N_VRING = T2N_NSET * T2N_VSLOT;
constexpr int T2N_LDS = T2N_NSET * T2N_SETB + 8 * T2N_VRING + 256;
#define T2N_GPTR(p) ((const __attribute__((address_space(1))) void *) (p))
#define T2N_LPTR(p) ((__attribute__((address_space(3))) void *) (p))
}  // namespace

template <int G, bool HAS_B1>
__global__ __launch_bounds__(512, 4) void spmm_team2n_kernel(const int ngrid, const int *__restrict__ tgrid, const int *__restrict__ tpanel,
                                                             const int *__restrict__ tinfo, const uint32_t *__restrict__ trec,
                                                             const long long *__restrict__ tvoff, const double *__restrict__ tval, const int nrow,
                                                             const int n, const double *__restrict__ B0, const int64_t ldB0, const double *__restrict__ B1, const int64_t ldB1,
                                                             double *__restrict__ C, const int64_t ldC, const int *__restrict__ rowmap)
{
    constexpr int LPG = 64 / G, SLOTB = 1024 / G;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *const ring = lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    char *const vr = lds + T2N_NSET * T2N_SETB + wave * T2N_VRING;
    const int cpx = (gridDim.x + 7) >> 3;
    const int gi = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);              // XCD x owns entries [x cpx, (x + 1) cpx) of the grid
    if (gi >= ngrid) return;
    const int g = __builtin_amdgcn_readfirstlane(tgrid[gi]);
    if (g < 0) return;
    const int nr = __builtin_amdgcn_readfirstlane(tinfo[2 * g]);
    const uint32_t *const recw = trec + ((size_t) __builtin_amdgcn_readfirstlane(tinfo[2 * g + 1]) * 128 + (size_t) wave * 16);
    const char *const vbase = reinterpret_cast<const char *>(tval + tvoff[(size_t) g * 8 + (size_t) wave] * 4);
    const int q = lane / LPG, l = lane % LPG, myrow = lane & 7;
    const int lo = (2 * l + 1 < n) ? l * 16 : 0;                           // lanes past n fetch the row's first bytes: valid, never stored
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = 0.0;

    auto issue = [&](const int r) {
        const uint32_t *R = recw + (size_t) r * 128;
        const uint4 h0 = *reinterpret_cast<const uint4 *>(R);              // parts | values, value offset, column 0, column 1
        // (readfirstlane: or the optimiser turns the selection into ONE per-lane load of R[2 + q] -- a vector load whose wait
        //  drains every DMA in flight)
        int col = __builtin_amdgcn_readfirstlane((int) h0.z);
        const int c1 = __builtin_amdgcn_readfirstlane((int) h0.w);
        if (q == 1) col = c1;
        if constexpr (G == 4)
        {
            const uint2 h1 = *reinterpret_cast<const uint2 *>(R + 4);
            const int c2 = __builtin_amdgcn_readfirstlane((int) h1.x), c3 = __builtin_amdgcn_readfirstlane((int) h1.y);
            if (q == 2) col = c2;
            if (q == 3) col = c3;
        }
        const char *src = (!HAS_B1 || col >= 0) ? reinterpret_cast<const char *>(B0 + (int64_t) col * ldB0) : reinterpret_cast<const char *>(B1 + (int64_t) (~col) * ldB1);
        const int set = r % T2N_NSET;
        __builtin_amdgcn_global_load_lds(T2N_GPTR(src + lo), T2N_LPTR(ring + set * T2N_SETB + wave * 1024), 16, 0, 0);
        const int nv = (int) ((h0.x >> 8) & 0x1FFu);
        const int nl = max(1, (nv + 1) >> 1);                              // 16-byte lanes of the value block (one at least: the count of DMAs is fixed)
        if (lane < nl) __builtin_amdgcn_global_load_lds(T2N_GPTR(vbase + (size_t) h0.y * 32 + lane * 16), T2N_LPTR(vr + set * T2N_VSLOT), 16, 0, 0);
    };
    auto rows = [&](const double v, const d2 b, const int mk) {
        int tt;
        uint64_t x0, x1, x2, x3, x4, x5, x6, x7;
        asm volatile(CRP_NARROW_STEP_NP1
                     : CRP_NARROW_ACC_NP1(a), [t] "=&v"(tt), [x0] "=&s"(x0), [x1] "=&s"(x1), [x2] "=&s"(x2), [x3] "=&s"(x3), [x4] "=&s"(x4),
                       [x5] "=&s"(x5), [x6] "=&s"(x6), [x7] "=&s"(x7)
                     : [v] "v"(v), [b0x] "v"(b.x), [b0y] "v"(b.y), [mk] "v"(mk));
    };

    for (int d = 0; d < T2N_D; d++)
        if (d < nr) issue(d);
    for (int r = 0; r < nr; r++)
    {
        // this wave's DMAs of round r have landed (two per round are younger for every round in flight behind it) ...
        const int later = min(T2N_D - 1, nr - 1 - r);
        if (later == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and everybody's: the barrier also says that every wave is done reading round r - 1, whose set round r + 3 takes
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (r + T2N_D < nr) issue(r + T2N_D);
        const uint32_t *R = recw + (size_t) r * 128;
        const int np = (int) (R[0] & 31u);
        const uint4 s01 = *reinterpret_cast<const uint4 *>(R + 6);         // steps 0, 1: masks, slots
        const uint4 s23 = *reinterpret_cast<const uint4 *>(R + 10);
        const char *const rs = ring + (r % T2N_NSET) * T2N_SETB + l * 16;
        const char *const vs = vr + (r % T2N_NSET) * T2N_VSLOT;
        int running = 0;
        auto step = [&](const uint32_t wm, const uint32_t ws, double &v, d2 &b, int &mk) {
            mk = (int) ((wm >> (8 * q)) & 0xFFu);
            const int slot = (int) ((ws >> (5 * q)) & 31u);
            b = *reinterpret_cast<const d2 *>(rs + slot * SLOTB);
            const int pre = __builtin_popcount(wm & ((1u << (8 * q)) - 1u));
            const int rank = __builtin_popcount((uint32_t) mk & ((1u << myrow) - 1u));
            v = *reinterpret_cast<const double *>(vs + (running + pre + rank) * 8);
            running += __builtin_popcount(wm);
        };
        double v0, v1;
        d2 b0, b1;
        int m0, m1;
        if (np > 0)
        {
            step(s01.x, s01.y, v0, b0, m0);
            if (np > G) step(s01.z, s01.w, v1, b1, m1);
            rows(v0, b0, m0);
            if (np > G) rows(v1, b1, m1);
        }
        if (np > 2 * G)
        {
            step(s23.x, s23.y, v0, b0, m0);
            if (np > 3 * G) step(s23.z, s23.w, v1, b1, m1);
            rows(v0, b0, m0);
            if (np > 3 * G) rows(v1, b1, m1);
        }
    }
    // the G partial sums of every row
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        if constexpr (G == 4) a[i] += __shfl_xor(a[i], 16);
        a[i] += __shfl_xor(a[i], 32);
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    const int panel = tpanel[(size_t) g * 8 + (size_t) wave];
    if (panel >= 0 && q == 0 && 2 * l + 1 < n)
    {
#pragma unroll
        for (int rr = 0; rr < 8; rr++)
        {
            const int row = panel * 8 + rr;
            if (row < nrow)
            {
                double *crow = C + (int64_t) (rowmap ? rowmap[row] : row) * ldC;
                d2 t2 = {a[rr * 2], a[rr * 2 + 1]};
                __builtin_nontemporal_store(t2, reinterpret_cast<d2 *>(crow + 2 * l));
            }
        }
    }
}

// 24 <= n <= 128 / G (even), 16-byte aligned operands
bool spmm_team2n_applicable(const Team2NArgs &t, const SpmmArgs &a)
{
    return a.n >= 24 && a.n <= 128 / t.G && (a.n % 2 == 0) && (a.ldB0 % 2 == 0) && (a.ldC % 2 == 0) && (a.B1 == nullptr || a.ldB1 % 2 == 0) &&
           (((uintptr_t) a.B0 | (uintptr_t) a.B1 | (uintptr_t) a.C) % 16 == 0);
}

hipError_t spmm_rm_f64_team2n(const Team2NArgs &t, const SpmmArgs &a, hipStream_t s)
{
    const bool has_b1 = a.B1 != nullptr;
    dim3 grid((t.ngrid + 7) / 8 * 8);
#define CRP_T2N_GO(G_, HB1_)                                                                                                                        \
    do                                                                                                                                              \
    {                                                                                                                                               \
        static bool once = false;                                                                                                                   \
        if (!once)                                                                                                                                  \
        {                                                                                                                                           \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&spmm_team2n_kernel<G_, HB1_>), hipFuncAttributeMaxDynamicSharedMemorySize, T2N_LDS); \
            if (e != hipSuccess) return e;                                                                                                          \
            once = true;                                                                                                                            \
        }                                                                                                                                           \
        hipLaunchKernelGGL((spmm_team2n_kernel<G_, HB1_>), grid, dim3(512), T2N_LDS, s, t.ngrid, t.tgrid, t.tpanel, t.tinfo, t.trec, t.tvoff, t.tval, a.nrow, a.n, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC, a.rowmap); \
    } while (0)
    if (t.G == 4) { if (has_b1) CRP_T2N_GO(4, true); else CRP_T2N_GO(4, false); }
    else { if (has_b1) CRP_T2N_GO(2, true); else CRP_T2N_GO(2, false); }
#undef CRP_T2N_GO
    return hipGetLastError();
}

}  // namespace crp
