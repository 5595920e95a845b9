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
// per-entry bookkeeping is paid for a quarter of a wave.  Here a wave takes G = 4 entries of its panel PER
// INSTRUCTION: lanes [16 q, 16 q + 16) hold entry e + q -- its B row slice (16 bytes per lane), its row mask,
// and its 8 values in the lanes' low three bits -- and every row's FMA is ONE v_fmac_f64_dpp per column of the lane
// for all G entries at once: the scalar factor comes through DPP row_newbcast:row from the lane's own 16-lane row,
// and rows that an entry does not have are switched off through EXEC, from a VECTOR compare of the lanes' masks
// (one scalar move per row and FOUR entries instead of one per row and entry), so that an absent (row, column) pair is
// never multiplied (no 0 * Inf).  The G partial sums of a row are added across the lane groups once per panel.
// Column indices and masks of 64 entries are fetched by one coalesced load each and handed to the lane groups by
// ds_bpermute.  Latency is hidden by occupancy (70 VGPRs: seven waves per SIMD, two steps in flight each), not by a
// register ring.  pwtk stand-in n = 32: 0.090 ms (row-panel kernel) -> 0.062 ms, 8.6 M scalar instructions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"
#include "knobs.h"
#include "narrow_rows.inc"

namespace crp {

namespace {
typedef double d2 __attribute__((ext_vector_type(2)));

}  // namespace

// NP = 16-byte pieces per lane: 1 (n <= 32: columns 2 l, 2 l + 1 of lane l of a 16-lane group) or 2 (n <= 64: also columns
// 32 + 2 l, 33 + 2 l -- the masks and the addressing of a step are then paid once for twice the FMAs).
// OFF32: B0 alone and smaller than 4 GiB -- every lane multiplies ITS column by the row stride once per 64 entries and
// the 32-bit byte offsets travel through ds_bpermute like the masks (the 64-bit multiply per step and lane group of the
// general path was 84 of its 230 vector cycles per step).
// COMPACT: the panels' values are stored without the holes (PanelHost::cmo / cbase / cval): pmask4 then points at one word per
// entry -- row mask | (index of the entry's first value inside the panel) << 8 -- and pval at the compact values; a lane
// finds its row's value at the entry's offset + the number of mask bits below its row.  For panels that are mostly holes (the
// nlpkkt stand-in: 23 % of the (row, entry) pairs exist) A shrinks from 69 to ≈ 21 bytes per entry -- at n = 32 A was 60 % of
// everything the kernel reads.
template <int NP, bool HAS_B1, bool OFF32, bool COMPACT>
__global__ __launch_bounds__(256, NP == 2 ? 4 : 7) void spmm_narrow_f64_kernel(
    const int norder, const int nrow, const int n, const int *__restrict__ porder, const int *__restrict__ pcol,
    const uint32_t *__restrict__ pmask4, const double *__restrict__ pval, const long long *__restrict__ cbase,
    const double *__restrict__ B0, const int64_t ldB0, const double *__restrict__ B1, const int64_t ldB1,
    double *__restrict__ C, const int64_t ldC, const int *__restrict__ rowmap)
{
    constexpr int LPG = 16, G = 4;                                          // lanes per entry, entries per step
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
    bool ok[NP];
    int bo[NP];                                                             // lanes past n read the row's first bytes: valid, never stored
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        ok[p] = (32 * p + 2 * l + 1) < n;
        bo[p] = ok[p] ? 32 * p + 2 * l : 0;
    }
    const uint8_t *pmask = reinterpret_cast<const uint8_t *>(pmask4);
    const double *const vpanel = COMPACT ? pval + cbase[panel] : pval;       // compact: the panel's first value
    const int myrow = lane & 7;
    double a[16 * NP];
#pragma unroll
    for (int i = 0; i < 16 * NP; i++) a[i] = 0.0;

    const uint32_t ld32 = (uint32_t) (ldB0 * 8);
    const char *const B0b = reinterpret_cast<const char *>(B0);
    const int sh0 = q * 4;                                                  // ds_bpermute address of lane q
    auto fetch = [&](const int mycol, const int mymask, const int ebase, const int s, double &v, d2 (&b)[NP], int &mk) {
        const int src = s * G + q;
        if constexpr (!COMPACT) v = pval[(size_t) (ebase + src) * 8 + (size_t) (lane & 7)];
        if constexpr (OFF32)
        {
            const int idx = sh0 + s * (G * 4);
            const uint32_t off = (uint32_t) __builtin_amdgcn_ds_bpermute(idx, mycol);      // mycol holds the byte offset of the row
            mk = __builtin_amdgcn_ds_bpermute(idx, mymask);
#pragma unroll
            for (int p = 0; p < NP; p++) b[p] = *reinterpret_cast<const d2 *>(B0b + off + bo[p] * 8);
        }
        else
        {
            const int col = __shfl(mycol, src);
            mk = __shfl(mymask, src);
            const double *brow = (!HAS_B1 || col >= 0) ? (B0 + (int64_t) col * ldB0) : (B1 + (int64_t) (~col) * ldB1);
#pragma unroll
            for (int p = 0; p < NP; p++) b[p] = *reinterpret_cast<const d2 *>(brow + bo[p]);
        }
        if constexpr (COMPACT)
        {
            // mk = row mask | value offset << 8: this lane's row is the (bits below it)-th value of the entry.  A row the
            // entry lacks reads a neighbouring value (inside the panel's values, or the 16 pad values after the last): never used
            const uint32_t word = (uint32_t) mk;
            mk = (int) (word & 0xFFu);
            const uint32_t rank = (uint32_t) __builtin_popcount(word & ((1u << myrow) - 1u));
            v = vpanel[(word >> 8) + rank];
        }
    };
    // One step (narrow_rows.inc): the eight row masks of the lanes' entries first (vector compares into SGPR pairs, EXEC
    // still full), then per row EXEC := its mask and the row's FMAs.  (EXEC written by the scalar unit needs no wait
    // states before a DPP instruction; written by v_cmpx it needs five, which cost 40 idle cycles per step and wave.)
    auto rows = [&](const double v, const d2 (&b)[NP], const int mk) {
        int t;
        uint64_t x0, x1, x2, x3, x4, x5, x6, x7;
#define CRP_NARROW_TMP [t] "=&v"(t), [x0] "=&s"(x0), [x1] "=&s"(x1), [x2] "=&s"(x2), [x3] "=&s"(x3), [x4] "=&s"(x4), [x5] "=&s"(x5), \
                       [x6] "=&s"(x6), [x7] "=&s"(x7)
        if constexpr (NP == 1)
            asm volatile(CRP_NARROW_STEP_NP1 : CRP_NARROW_ACC_NP1(a), CRP_NARROW_TMP : [v] "v"(v), [b0x] "v"(b[0].x), [b0y] "v"(b[0].y), [mk] "v"(mk));
        else
            asm volatile(CRP_NARROW_STEP_NP2 : CRP_NARROW_ACC_NP2(a), CRP_NARROW_TMP
                         : [v] "v"(v), [b0x] "v"(b[0].x), [b0y] "v"(b[0].y), [b1x] "v"(b[NP - 1].x), [b1y] "v"(b[NP - 1].y), [mk] "v"(mk));
#undef CRP_NARROW_TMP
    };
    for (int base = 0; base < nent; base += 64)
    {
        const int ce = min(64, nent - base);                               // uniform, a multiple of 8
        int mycol = (lane < ce) ? pcol[e0 + base + lane] : 0;
        if constexpr (OFF32) mycol = (int) ((uint32_t) mycol * ld32);
        const int mymask = (lane < ce) ? (COMPACT ? (int) pmask4[e0 + base + lane] : (int) pmask[e0 + base + lane]) : 0;
        const int nstep = ce / G;                                          // even
        // NP = 1: two steps per iteration, their loads in flight together (the compiler's wait before an asm statement
        // cannot be counted across the loop's back edge, so a deeper software pipeline would not overlap anything).
        // (Tried: the block's 16 steps spelled out with a three-step prefetch -- it needs the operand pointers without
        //  __restrict__ and a "memory" clobber on the step, or the optimiser sinks every prefetch to its use; then the
        //  waits are vmcnt(6) as intended, at 80-88 VGPRs: 0.063 ms against 0.062 -- the kernel is bound by its vector
        //  instructions, not by exposed latency.)
        // NP = 2: one step (twice the bytes per step already, and two would spill at four waves per SIMD)
        for (int s = 0; s < nstep; s += (NP == 1 ? 2 : 1))
        {
            double v0, v1;
            d2 b0[NP], b1[NP];
            int m0, m1;
            fetch(mycol, mymask, e0 + base, s, v0, b0, m0);
            if constexpr (NP == 1) fetch(mycol, mymask, e0 + base, s + 1, v1, b1, m1);
            rows(v0, b0, m0);
            if constexpr (NP == 1) rows(v1, b1, m1);
        }
    }
    // the G partial sums of every row: lanes l, l + 16, l + 32, l + 48 -> all of them hold the total
#pragma unroll
    for (int i = 0; i < 16 * NP; i++)
    {
        a[i] += __shfl_xor(a[i], 16);
        a[i] += __shfl_xor(a[i], 32);
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);                // a few sums at a time: scheduled in bulk, their temporaries spill
    }
    if (q == 0)
    {
#pragma unroll
        for (int r = 0; r < 8; r++)
        {
            const int row = panel * 8 + r;
            if (row < nrow)
            {
                double *crow = C + (int64_t) (rowmap ? rowmap[row] : row) * ldC;
#pragma unroll
                for (int p = 0; p < NP; p++)
                    if (ok[p])
                    {
                        d2 t2 = {a[(r * NP + p) * 2], a[(r * NP + p) * 2 + 1]};
                        __builtin_nontemporal_store(t2, reinterpret_cast<d2 *>(crow + 32 * p + 2 * l));
                    }
            }
        }
    }
}

// n <= 64, even, 16-byte aligned operands, an order laid out for four-wave workgroups (team_waves == 4), R = 8
bool spmm_narrow_applicable(const PanelArgs &p, const SpmmArgs &a)
{
    // (NP = 2, i.e. 32 < n <= 64, is built but not chosen on the full-value format: 128 VGPRs leave four waves per SIMD with
    //  one step in flight each -- pwtk stand-in n = 64: 0.168 ms against 0.117 for the row-panel kernel; with 32 lanes per
    //  entry and two entries per instruction it was 0.115: no gain either; at five waves per SIMD (96 VGPRs, the epilogue's
    //  sums spilled) 0.209.  CRPSPMM_NARROW_MAX=64 selects it; p.narrow64 = panels that are mostly holes, on compact values.)
    const int nmax_env = knobs().narrow_max;
    const int nmax = nmax_env > 0 ? nmax_env : (p.narrow64 && p.cmo != nullptr ? 64 : 32);
    return p.R == 8 && p.team_waves == 4 && a.n >= 24 && a.n <= nmax && a.n <= 64 && (a.n % 2 == 0) && (a.ldB0 % 2 == 0) && (a.ldC % 2 == 0) &&
           (a.B1 == nullptr || a.ldB1 % 2 == 0) && (((uintptr_t) a.B0 | (uintptr_t) a.B1 | (uintptr_t) a.C) % 16 == 0);
}

hipError_t spmm_rm_f64_narrow(const PanelArgs &p, const SpmmArgs &a, hipStream_t s)
{
    const int nwg = (p.norder + 3) / 4;
    dim3 grid((nwg + 7) / 8 * 8);
    const bool has_b1 = a.B1 != nullptr && p.b1_rows > 0;
    // 32-bit byte offsets: B0 alone, every addressed byte below 4 GiB
    const bool off32 = !has_b1 && (uint64_t) p.b0_rows * (uint64_t) a.ldB0 * 8ull < (1ull << 32);
    const bool compact = p.cmo != nullptr && p.cbase != nullptr && p.cval != nullptr;
#define CRP_NARROW_GO(NP_, HB1_, O32_, CP_)                                                                                            \
    hipLaunchKernelGGL((spmm_narrow_f64_kernel<NP_, HB1_, O32_, CP_>), grid, dim3(256), 0, s, p.norder, a.nrow, a.n, p.porder, p.pcol, \
                       CP_ ? p.cmo : p.pmask4, CP_ ? p.cval : p.pval, p.cbase, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC, a.rowmap)
#define CRP_NARROW_PICK(NP_)                                                                                                     \
    do                                                                                                                           \
    {                                                                                                                            \
        if (compact) { if (has_b1) CRP_NARROW_GO(NP_, true, false, true); else if (off32) CRP_NARROW_GO(NP_, false, true, true); else CRP_NARROW_GO(NP_, false, false, true); } \
        else { if (has_b1) CRP_NARROW_GO(NP_, true, false, false); else if (off32) CRP_NARROW_GO(NP_, false, true, false); else CRP_NARROW_GO(NP_, false, false, false); }     \
    } while (0)
    if (a.n <= 32) CRP_NARROW_PICK(1);
    else CRP_NARROW_PICK(2);
#undef CRP_NARROW_PICK
#undef CRP_NARROW_GO
    return hipGetLastError();
}

}  // namespace crp
