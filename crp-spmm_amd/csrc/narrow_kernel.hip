// narrow_kernel.hip -- row-panel SpMM for narrow operands (n <= 64 columns, fp64) on gfx950.
//
// Same product as every kernel of this library (what mkl_sparse_d_mm computes at
// /root/reference/src/rowpara_spmm.c:388-408 with alpha = 1, beta = 0), on the R = 8 row-panel format of
// panel_format.h (one wave per panel, entries = union of the panel's columns with an 8-bit row mask and 8 values).
//
// Why another body: under the 1 x P grids of the planner every GPU multiplies by n / P columns, i.e. narrow
// operands are what multi-GPU runs see.  There the row-panel kernel of spmm_kernels.hip is bound by the SCALAR unit
// (38 M scalar instructions per launch on the pwtk stand-in at n = 32 and 64 alike: one EXEC write per row and
// entry; the vector memory path is 36 % busy), because a B row slice of 256 bytes fills only 16 lanes and the
// per-entry bookkeeping is paid for a quarter of a wave.  Here a wave takes G = 64 / LPG entries of its panel PER
// INSTRUCTION: lanes [q * LPG, (q + 1) * LPG) hold entry e + q -- its B row slice (16 bytes per lane), its row mask,
// and its 8 values in the lanes' low three bits -- and every row's FMA is ONE v_fmac_f64_dpp per column of the lane
// for all G entries at once: the scalar factor comes through DPP row_newbcast:row from the lane's own 16-lane row,
// and rows that an entry does not have are switched off through EXEC, set by a VECTOR compare of the lanes' masks
// (v_cmpx), so that no scalar instruction is spent per row and an absent (row, column) pair is never multiplied
// (no 0 * Inf).  The G partial sums of a row are added across the lane groups once per panel.
// Column indices and masks of 64 entries are fetched by one coalesced load each and handed to the lane groups by
// ds_bpermute.  Latency is hidden by occupancy (about 60 VGPRs: eight waves per SIMD), not by a register ring.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

namespace crp {

namespace {
typedef double d2 __attribute__((ext_vector_type(2)));

}  // namespace

// OFF32: B0 alone and smaller than 4 GiB -- every lane multiplies ITS column by the row stride once per 64 entries and
// the 32-bit byte offsets travel through ds_bpermute like the masks (the 64-bit multiply per step and lane group of the
// general path is 84 of its 230 vector cycles per step)
template <int LPG, bool HAS_B1, bool OFF32>
__global__ __launch_bounds__(256) void spmm_narrow_f64_kernel(
    const int norder, const int nrow, const int n, const int *__restrict__ porder, const int *__restrict__ pcol,
    const uint32_t *__restrict__ pmask4, const double *__restrict__ pval,
    const double *__restrict__ B0, const int64_t ldB0, const double *__restrict__ B1, const int64_t ldB1,
    double *__restrict__ C, const int64_t ldC, const int *__restrict__ rowmap)
{
    constexpr int G = 64 / LPG;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // XCD-aware placement, as in the row-panel kernel (the processing order is laid out for four-wave workgroups)
    const int cpx = (gridDim.x + 7) >> 3;
    const int wg = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    const int slot_id = __builtin_amdgcn_readfirstlane((int) (wg * 4 + wave));
    if (slot_id >= norder) return;
    const int4 rec = reinterpret_cast<const int4 *>(porder)[slot_id];      // {panel or -1, first entry, rounds of 8 entries}
    const int panel = __builtin_amdgcn_readfirstlane(rec.x);
    if (panel < 0) return;
    const int e0 = __builtin_amdgcn_readfirstlane(rec.y);
    const int nent = __builtin_amdgcn_readfirstlane(rec.z) * 8;            // padded to a multiple of 8: mask-0 entries, valid column

    const int q = lane / LPG, l = lane % LPG;
    const bool ok = (2 * l + 1) < n;
    const int bo = ok ? 2 * l : 0;                 // lanes past n read the row's first bytes: valid, never stored
    const uint8_t *pmask = reinterpret_cast<const uint8_t *>(pmask4);
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = 0.0;

    const uint32_t ld32 = (uint32_t) (ldB0 * 8);
    const char *const B0b = reinterpret_cast<const char *>(B0) + bo * 8;
    const int sh0 = q * 4;                                                  // ds_bpermute address of lane q
    auto fetch = [&](const int mycol, const int mymask, const int ebase, const int s, double &v, d2 &b, int &mk) {
        const int src = s * G + q;
        v = pval[(size_t) (ebase + src) * 8 + (size_t) (lane & 7)];
        if constexpr (OFF32)
        {
            const int idx = sh0 + s * (G * 4);
            const uint32_t off = (uint32_t) __builtin_amdgcn_ds_bpermute(idx, mycol);      // mycol holds the byte offset of the row
            mk = __builtin_amdgcn_ds_bpermute(idx, mymask);
            b = *reinterpret_cast<const d2 *>(B0b + off);
        }
        else
        {
            const int col = __shfl(mycol, src);
            mk = __shfl(mymask, src);
            const double *brow = (!HAS_B1 || col >= 0) ? (B0 + (int64_t) col * ldB0) : (B1 + (int64_t) (~col) * ldB1);
            b = *reinterpret_cast<const d2 *>(brow + bo);
        }
    };
    auto rows = [&](const double v, const d2 b, const int mk) {
        int t;
        asm volatile(
                "v_and_b32 %[t], 1, %[mk]\n\tv_cmpx_ne_u32 vcc, 0, %[t]\n\ts_nop 4\n\t"
                "v_fmac_f64_dpp %[a0], %[v], %[bx] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %[a1], %[v], %[by] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_and_b32 %[t], 2, %[mk]\n\tv_cmpx_ne_u32 vcc, 0, %[t]\n\ts_nop 4\n\t"
                "v_fmac_f64_dpp %[a2], %[v], %[bx] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %[a3], %[v], %[by] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_and_b32 %[t], 4, %[mk]\n\tv_cmpx_ne_u32 vcc, 0, %[t]\n\ts_nop 4\n\t"
                "v_fmac_f64_dpp %[a4], %[v], %[bx] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %[a5], %[v], %[by] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_and_b32 %[t], 8, %[mk]\n\tv_cmpx_ne_u32 vcc, 0, %[t]\n\ts_nop 4\n\t"
                "v_fmac_f64_dpp %[a6], %[v], %[bx] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %[a7], %[v], %[by] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_and_b32 %[t], 16, %[mk]\n\tv_cmpx_ne_u32 vcc, 0, %[t]\n\ts_nop 4\n\t"
                "v_fmac_f64_dpp %[a8], %[v], %[bx] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %[a9], %[v], %[by] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_and_b32 %[t], 32, %[mk]\n\tv_cmpx_ne_u32 vcc, 0, %[t]\n\ts_nop 4\n\t"
                "v_fmac_f64_dpp %[a10], %[v], %[bx] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %[a11], %[v], %[by] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_and_b32 %[t], 64, %[mk]\n\tv_cmpx_ne_u32 vcc, 0, %[t]\n\ts_nop 4\n\t"
                "v_fmac_f64_dpp %[a12], %[v], %[bx] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %[a13], %[v], %[by] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_and_b32 %[t], 0x80, %[mk]\n\tv_cmpx_ne_u32 vcc, 0, %[t]\n\ts_nop 4\n\t"
                "v_fmac_f64_dpp %[a14], %[v], %[bx] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %[a15], %[v], %[by] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1"
                : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]),
                  [a6] "+v"(a[6]), [a7] "+v"(a[7]), [a8] "+v"(a[8]), [a9] "+v"(a[9]), [a10] "+v"(a[10]), [a11] "+v"(a[11]),
                  [a12] "+v"(a[12]), [a13] "+v"(a[13]), [a14] "+v"(a[14]), [a15] "+v"(a[15]), [t] "=&v"(t)
                : [v] "v"(v), [bx] "v"(b.x), [by] "v"(b.y), [mk] "v"(mk)
                : "vcc");
    };
    for (int base = 0; base < nent; base += 64)
    {
        const int ce = min(64, nent - base);                               // uniform, a multiple of 8
        int mycol = (lane < ce) ? pcol[e0 + base + lane] : 0;
        if constexpr (OFF32) mycol = (int) ((uint32_t) mycol * ld32);
        const int mymask = (lane < ce) ? (int) pmask[e0 + base + lane] : 0;
        const int nstep = ce / G;                                          // even
        // two steps per iteration: their loads are in flight together (the compiler's wait before an asm statement
        // is vmcnt(0), so a deeper software pipeline across iterations would not overlap anything)
        for (int s = 0; s < nstep; s += 2)
        {
            double v0, v1;
            d2 b0, b1;
            int m0, m1;
            fetch(mycol, mymask, e0 + base, s, v0, b0, m0);
            fetch(mycol, mymask, e0 + base, s + 1, v1, b1, m1);
            rows(v0, b0, m0);
            rows(v1, b1, m1);
        }
    }
    // the G partial sums of every row: lanes l, l + LPG, ... -> all of them hold the total
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        if constexpr (G == 4) a[i] += __shfl_xor(a[i], 16);
        a[i] += __shfl_xor(a[i], 32);
    }
    if (q == 0 && ok)
    {
#pragma unroll
        for (int r = 0; r < 8; r++)
        {
            const int row = panel * 8 + r;
            if (row < nrow)
            {
                d2 t2 = {a[2 * r], a[2 * r + 1]};
                double *crow = C + (int64_t) (rowmap ? rowmap[row] : row) * ldC;
                __builtin_nontemporal_store(t2, reinterpret_cast<d2 *>(crow + 2 * l));
            }
        }
    }
}

// n <= 64, even, 16-byte aligned operands, an order laid out for four-wave workgroups (team_waves == 4), R = 8
bool spmm_narrow_applicable(const PanelArgs &p, const SpmmArgs &a)
{
    static const bool on = getenv("CRPSPMM_NARROW") == NULL || atoi(getenv("CRPSPMM_NARROW")) != 0;
    // (LPG = 32, i.e. 32 < n <= 64, is built but not chosen: two entries per instruction pay the row overhead for
    //  half the lanes' worth of work -- pwtk stand-in n = 64: 0.159 ms against 0.117 for the row-panel kernel)
    static const int nmax = getenv("CRPSPMM_NARROW_MAX") ? atoi(getenv("CRPSPMM_NARROW_MAX")) : 32;
    return on && p.R == 8 && p.team_waves == 4 && a.n >= 24 && a.n <= nmax && a.n <= 64 && (a.n % 2 == 0) && (a.ldB0 % 2 == 0) && (a.ldC % 2 == 0) &&
           (a.B1 == nullptr || a.ldB1 % 2 == 0) && (((uintptr_t) a.B0 | (uintptr_t) a.B1 | (uintptr_t) a.C) % 16 == 0);
}

hipError_t spmm_rm_f64_narrow(const PanelArgs &p, const SpmmArgs &a, hipStream_t s)
{
    const int nwg = (p.norder + 3) / 4;
    dim3 grid((nwg + 7) / 8 * 8);
    const bool has_b1 = a.B1 != nullptr && p.b1_rows > 0;
    // 32-bit byte offsets: B0 alone, every addressed byte below 4 GiB
    const bool off32 = !has_b1 && (uint64_t) p.b0_rows * (uint64_t) a.ldB0 * 8ull < (1ull << 32);
#define CRP_NARROW_GO(LPG_, HB1_, O32_)                                                                                          \
    hipLaunchKernelGGL((spmm_narrow_f64_kernel<LPG_, HB1_, O32_>), grid, dim3(256), 0, s, p.norder, a.nrow, a.n, p.porder, p.pcol, \
                       p.pmask4, p.pval, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC, a.rowmap)
    if (a.n <= 32) { if (has_b1) CRP_NARROW_GO(16, true, false); else if (off32) CRP_NARROW_GO(16, false, true); else CRP_NARROW_GO(16, false, false); }
    else           { if (has_b1) CRP_NARROW_GO(32, true, false); else if (off32) CRP_NARROW_GO(32, false, true); else CRP_NARROW_GO(32, false, false); }
#undef CRP_NARROW_GO
    return hipGetLastError();
}

}  // namespace crp
