// spmm_kernels.hip -- CSR x dense SpMM kernels for gfx950 (MI355X, wave64).
//
// Computes what mkl_sparse_d_mm computes at the reference call site
// /root/reference/src/rowpara_spmm.c:388-408 (alpha = 1, beta = 0):
//     C[i][0:n] = sum_p val[p] * B[col[p]][0:n],  p = rowptr[i] .. rowptr[i+1]-1
// The operation is HBM/L2-bound gather work: no MFMA here.  Each group of LPR
// lanes owns one row of A and a TW = LPR*VW*NV wide slice of C; the (col, val)
// pairs of the row are loaded coalesced LPR at a time and broadcast inside the
// group, every lane then streams 16-byte pieces of the addressed B rows.
//
// Two-source column index (see include/crpspmm_hip.h): c >= 0 -> B0 row c,
// c < 0 -> B1 row ~c.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"

#ifdef CRP_ABL_PLAINSTORE     // timing experiment: C through the default (write-back) store path
#define CRP_STORE(val, ptr) (*(ptr) = (val))
#else
#define CRP_STORE(val, ptr) __builtin_nontemporal_store((val), (ptr))
#endif

namespace crp {

typedef double d2 __attribute__((ext_vector_type(2)));

template <int LPR>
__device__ __forceinline__ int bcast_i(int v, int j)
{
    if constexpr (LPR == 64) return __builtin_amdgcn_readlane(v, j);
    else return __shfl(v, j, LPR);
}

template <int LPR>
__device__ __forceinline__ double bcast_d(double v, int j)
{
    if constexpr (LPR == 64)
    {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
        return __hiloint2double(hi, lo);
    }
    else return __shfl(v, j, LPR);
}

// ---------------------------------------------------------------------------
// Row-major kernel.  LPR lanes per row, VW doubles per vector access (1 or 2),
// NV vector accesses per lane => tile width TW = LPR*VW*NV columns.
// grid.x covers rows (RPB = 256/LPR rows per block), grid.y covers column tiles.
// ---------------------------------------------------------------------------
template <int LPR, int VW, int NV>
__global__ __launch_bounds__(256) void spmm_rm_f64_kernel(
    const int nrow, const int n,
    const int *__restrict__ rowptr, const int *__restrict__ colidx, const double *__restrict__ val,
    const double *__restrict__ B0, const int64_t ldB0,
    const double *__restrict__ B1, const int64_t ldB1,
    double *__restrict__ C, const int64_t ldC, const int *__restrict__ rowmap)
{
    constexpr int RPB = 256 / LPR;
    constexpr int TW  = LPR * VW * NV;
    const int lir  = threadIdx.x % LPR;                 // lane in row group
    const int row  = blockIdx.x * RPB + threadIdx.x / LPR;
    const int col0 = blockIdx.y * TW + lir * VW;        // first column of this lane
    if (row >= nrow) return;

    double acc[NV][VW];
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int w = 0; w < VW; w++) acc[v][w] = 0.0;

    // Column slots of this lane.  Slots past n are clamped to column 0 for the LOADS
    // (always a valid address, so loads stay unconditional and can be batched) and are
    // simply not stored.
    bool ok[NV];
    int  coff[NV];
#pragma unroll
    for (int v = 0; v < NV; v++)
    {
        const int c = col0 + v * LPR * VW;
        ok[v]   = (c + VW - 1) < n;
        coff[v] = ok[v] ? c : 0;
    }

    int p0 = rowptr[row];
    const int pe = rowptr[row + 1];
    if constexpr (LPR == 64)
    {
        p0 = __builtin_amdgcn_readfirstlane(p0);
    }
    constexpr int UNR = 8;
    for (; p0 < pe; p0 += LPR)
    {
        const int my = p0 + lir;
        int    c = 0;
        double a = 0.0;
        if (my < pe)
        {
            c = colidx[my];
            a = val[my];
        }
        const int cnt = min(LPR, pe - p0);
        for (int j = 0; j < cnt; j += UNR)
        {
            // phase 1: issue every B-row load of this group (indices past the row end are
            // clamped to the row's last entry: a valid address whose data is not used)
            double aj[UNR];
            double bv[UNR][NV][VW];
#pragma unroll
            for (int u = 0; u < UNR; u++)
            {
                const int ju = min(j + u, cnt - 1);
                const int cj = bcast_i<LPR>(c, ju);
                aj[u] = bcast_d<LPR>(a, ju);
                const double *brow = (cj >= 0) ? (B0 + (int64_t) cj * ldB0)
                                               : (B1 + (int64_t) (~cj) * ldB1);
#pragma unroll
                for (int v = 0; v < NV; v++)
                {
                    if constexpr (VW == 2)
                    {
                        const d2 t = *reinterpret_cast<const d2 *>(brow + coff[v]);
                        bv[u][v][0] = t.x;
                        bv[u][v][1] = t.y;
                    }
                    else bv[u][v][0] = brow[coff[v]];
                }
            }
            // phase 2: accumulate in nonzero order; only real entries contribute
#pragma unroll
            for (int u = 0; u < UNR; u++)
            {
                if (j + u < cnt)
                {
#pragma unroll
                    for (int v = 0; v < NV; v++)
#pragma unroll
                        for (int w = 0; w < VW; w++) acc[v][w] = fma(aj[u], bv[u][v][w], acc[v][w]);
                }
            }
        }
    }

    double *crow = C + (int64_t) (rowmap ? rowmap[row] : row) * ldC;     // rowmap: C row of every row of a row-subset matrix
#pragma unroll
    for (int v = 0; v < NV; v++)
    {
        if (ok[v])
        {
            if constexpr (VW == 2)
            {
                d2 r;
                r.x = acc[v][0];
                r.y = acc[v][1];
                __builtin_nontemporal_store(r, reinterpret_cast<d2 *>(crow + coff[v]));
            }
            else
            {
                __builtin_nontemporal_store(acc[v][0], crow + coff[v]);
            }
        }
    }
}

template <int LPR, int VW, int NV>
static hipError_t launch_rm(const SpmmArgs &a, hipStream_t s)
{
    constexpr int RPB = 256 / LPR;
    constexpr int TW  = LPR * VW * NV;
    dim3 grid((a.nrow + RPB - 1) / RPB, (a.n + TW - 1) / TW);
    hipLaunchKernelGGL((spmm_rm_f64_kernel<LPR, VW, NV>), grid, dim3(256), 0, s,
                       a.nrow, a.n, a.rowptr, a.colidx, a.val, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC, a.rowmap);
    return hipGetLastError();
}

// Generic row-major dispatcher: picks the lane group / vector shape from n and
// from the 16-byte alignment of the operands.
hipError_t spmm_rm_f64_rowgroup(const SpmmArgs &a, hipStream_t s)
{
    const bool vec2 =
        (a.n % 2 == 0) && (a.ldB0 % 2 == 0) && (a.ldC % 2 == 0) && (a.B1 == nullptr || a.ldB1 % 2 == 0) &&
        (((uintptr_t) a.B0 | (uintptr_t) a.B1 | (uintptr_t) a.C) % 16 == 0);
    if (vec2)
    {
        if (a.n <= 8)   return launch_rm<4, 2, 1>(a, s);
        if (a.n <= 16)  return launch_rm<8, 2, 1>(a, s);
        if (a.n <= 32)  return launch_rm<16, 2, 1>(a, s);
        if (a.n <= 64)  return launch_rm<32, 2, 1>(a, s);
        if (a.n <= 128) return launch_rm<64, 2, 1>(a, s);
        return launch_rm<64, 2, 2>(a, s);
    }
    if (a.n <= 4)   return launch_rm<4, 1, 1>(a, s);
    if (a.n <= 8)   return launch_rm<8, 1, 1>(a, s);
    if (a.n <= 16)  return launch_rm<16, 1, 1>(a, s);
    if (a.n <= 32)  return launch_rm<32, 1, 1>(a, s);
    if (a.n <= 64)  return launch_rm<64, 1, 1>(a, s);
    return launch_rm<64, 1, 2>(a, s);
}

// ---------------------------------------------------------------------------
// Row-panel kernel (panel_format.h): one wavefront per panel of R rows and a
// TW = 128*NV wide slice of C.  For every panel entry (one column index shared by
// up to R rows) the wave loads the B row slice ONCE -- 16 bytes per lane per NV --
// and feeds it to the accumulators of the rows whose mask bit is set.
//
// Register ring: RING entries' B slices live in VGPRs; slot k is consumed (FMAs) and
// immediately refilled with the entry RING positions ahead, so RING-1 entries
// (~2 KiB each at NV = 2) stay in flight per wave behind counted vmcnt waits.  The
// VGPR file (512 KiB per CU) is the largest buffer on the CU to keep bytes in flight,
// three times the LDS.  Panel entry counts are padded to a multiple of RING (mask-0
// entries), and the last round is peeled, so the rounds are straight-line code.
//
// Operand delivery per entry:
//   column index, mask : wave-uniform scalar loads, fetched one round (RING entries) ahead;
//   R values           : staged per wave through LDS in chunks of PANEL_CHUNK entries
//                        (coalesced vector loads, one entry per lane), then read back with
//                        uniform-address ds_read (broadcast).  A scalar load per entry would
//                        expose its full latency every entry (SMEM completes out of order,
//                        so every wait is lgkmcnt(0));
//   B row slice        : buffer_load_dwordx4 with the row offset in the SGPR soffset operand
//                        (one s_mul per entry, no per-lane address arithmetic); ADDR64 falls
//                        back to 64-bit global addresses when a B block exceeds 4 GiB.
// ---------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N)
    {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int PANEL_RING  = 8;
constexpr int PANEL_CHUNK = 32;

typedef int i4v __attribute__((ext_vector_type(4)));

template <bool ADDR64>
struct BSource
{
    i4v rsrc;              // raw buffer descriptor words (base, stride 0, num_records = 4 GiB - 1, flags)
    const double *base;
    uint32_t ldbytes;
    int64_t  ld;
};

// 128-bit buffer descriptor for a linear byte buffer at `p` (same words as
// __builtin_amdgcn_make_buffer_rsrc(p, 0, 0xFFFFFFFF, 0x00020000)); every word is forced wave-uniform.
__device__ __forceinline__ i4v make_rsrc(const void *p)
{
    const uint64_t a = (uint64_t) p;
    i4v r;
    r.x = __builtin_amdgcn_readfirstlane((int) (uint32_t) a);
    r.y = __builtin_amdgcn_readfirstlane((int) ((uint32_t) (a >> 32) & 0xFFFFu));
    r.z = (int) 0xFFFFFFFFu;
    r.w = 0x00020000;
    return r;
}

typedef unsigned int u2v __attribute__((ext_vector_type(2)));

// one slot element: VW = 2 -> 16-byte load (two columns per lane), VW = 1 -> 8-byte load
template <int VW> struct SlotT;
template <> struct SlotT<2> { typedef d2 type; };
template <> struct SlotT<1> { typedef double type; };

// B-slice loads are issued from inline asm so that the compiler does NOT know they are pending:
// its waitcnt pass is conservative for loads that cross a loop back-edge (it drains the ring with
// vmcnt(0) once per round), whereas the ring is consumed strictly in issue order and the number of
// younger loads at every consume point is known exactly -- the consume blocks carry their own counted
// s_waitcnt vmcnt(N).  Compiler-issued VMEM ops interleaved with these only make its own waits
// stricter (in-order counter), never weaker.
template <int VW>
__device__ __forceinline__ void buf_load_asm(typename SlotT<VW>::type &dst, const i4v rsrc, const int voff, const uint32_t soff)
{
    if constexpr (VW == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff));
    else asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff));
}

template <int VW>
__device__ __forceinline__ void glb_load_asm(typename SlotT<VW>::type &dst, const void *addr)
{
    if constexpr (VW == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(addr));
    else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(addr));
}

template <bool ADDR64, bool HAS_B1, int NV, int VW>
__device__ __forceinline__ void panel_issue1(typename SlotT<VW>::type (&slot)[NV], const int cj, const BSource<ADDR64> &s0,
                                             const BSource<ADDR64> &s1, const int (&voff)[NV])
{
    const bool remote = HAS_B1 && (cj < 0);
    const uint32_t row = remote ? (uint32_t) (~cj) : (uint32_t) cj;
    if constexpr (ADDR64)
    {
        const double *brow = remote ? (s1.base + (int64_t) row * s1.ld) : (s0.base + (int64_t) row * s0.ld);
#pragma unroll
        for (int v = 0; v < NV; v++) glb_load_asm<VW>(slot[v], reinterpret_cast<const char *>(brow) + voff[v]);
    }
    else
    {
        if (remote)
        {
            const uint32_t soff = row * s1.ldbytes;
#pragma unroll
            for (int v = 0; v < NV; v++) buf_load_asm<VW>(slot[v], s1.rsrc, voff[v], soff);
        }
        else
        {
            const uint32_t soff = row * s0.ldbytes;
#pragma unroll
            for (int v = 0; v < NV; v++) buf_load_asm<VW>(slot[v], s0.rsrc, voff[v], soff);
        }
    }
}

template <int VW>
__device__ __forceinline__ double slot_elem(const typename SlotT<VW>::type &s, const int w)
{
    if constexpr (VW == 2) return w == 0 ? s.x : s.y;
    else return s;
}

// wait until at most N vector-memory operations of this wave are outstanding
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Masked in-place FMA of one panel row: if bit BIT of the (wave-uniform, SGPR) mask is set,
// acc[v][w] += a * slot[v][w].  One inline-asm block holding the scalar test, the branch and the
// v_fmac_f64 group: written in C (`if (mask & bit) acc = fma(..)`) hipcc keeps TWO copies of every
// accumulator and emits one v_mov_b64 per accumulator in front of every branch (as many moves as
// FMAs, twice the accumulator registers); the asm block is straight-line code to the compiler and
// updates the accumulators in place.  Absent (row, column) pairs execute no FMA at all (so 0 * Inf
// never appears), they cost one s_bitcmp + one taken s_cbranch.
template <int NV, int VW, int BIT>
__device__ __forceinline__ void fmac_row_masked(double (&acc)[NV][VW], const double a,
                                                const typename SlotT<VW>::type (&slot)[NV], const uint32_t mask)
{
    if constexpr (NV == 2 && VW == 2)
        asm volatile("s_bitcmp0_b32 %9, %10\n\ts_cbranch_scc1 1f\n\tv_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %6\n\t"
                     "v_fmac_f64 %2, %4, %7\n\tv_fmac_f64 %3, %4, %8\n1:"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1])
                     : "v"(a), "v"(slot[0].x), "v"(slot[0].y), "v"(slot[1].x), "v"(slot[1].y), "s"(mask), "n"(BIT)
                     : "scc");
    else if constexpr (NV == 1 && VW == 2)
        asm volatile("s_bitcmp0_b32 %5, %6\n\ts_cbranch_scc1 1f\n\tv_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %1, %2, %4\n1:"
                     : "+v"(acc[0][0]), "+v"(acc[0][1])
                     : "v"(a), "v"(slot[0].x), "v"(slot[0].y), "s"(mask), "n"(BIT)
                     : "scc");
    else if constexpr (NV == 2 && VW == 1)
        asm volatile("s_bitcmp0_b32 %5, %6\n\ts_cbranch_scc1 1f\n\tv_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %1, %2, %4\n1:"
                     : "+v"(acc[0][0]), "+v"(acc[1][0])
                     : "v"(a), "v"(slot[0]), "v"(slot[1]), "s"(mask), "n"(BIT)
                     : "scc");
    else
    {
        static_assert(NV == 1 && VW == 1, "unsupported tile shape");
        asm volatile("s_bitcmp0_b32 %3, %4\n\ts_cbranch_scc1 1f\n\tv_fmac_f64 %0, %1, %2\n1:"
                     : "+v"(acc[0][0])
                     : "v"(a), "v"(slot[0]), "s"(mask), "n"(BIT)
                     : "scc");
    }
}

template <int R, int NV, int VW, int BIT = 0>
__device__ __forceinline__ void fmac_rows(double (&acc)[R][NV][VW], const double (&a)[R],
                                          const typename SlotT<VW>::type (&slot)[NV], const uint32_t mask)
{
    fmac_row_masked<NV, VW, BIT>(acc[BIT], a[BIT], slot, mask);
    if constexpr (BIT + 1 < R) fmac_rows<R, NV, VW, BIT + 1>(acc, a, slot, mask);
}

// DPP delivery of the values: a VGPR pair `vv` holds 16 consecutive values of the panel's value
// array (lane l has value l & 15: 16 / R entries), read with ONE ds_read_b64 per 16 values; the FMA takes
// its scalar factor through DPP row_newbcast:LANE (every lane reads lane LANE of its own row of 16), the
// only DPP control 64-bit operations have on gfx90a+.  This replaces R uniform-address LDS reads per
// entry.  The DPP source was written by an LDS read (no VALU-write -> DPP-read hazard); s_bitcmp +
// s_cbranch in front of the first FMA are two wait states in any case.
template <int NV, int VW, int BIT, int LANE>
__device__ __forceinline__ void fmac_row_masked_dpp(double (&acc)[NV][VW], const double vv,
                                                    const typename SlotT<VW>::type (&slot)[NV], const uint32_t mask)
{
#define CRP_DPPF(A, B) "v_fmac_f64_dpp " A ", %[vv], " B " row_newbcast:%[ln] row_mask:0xf bank_mask:0xf\n\t"
    if constexpr (NV == 2 && VW == 2)
        asm volatile("s_bitcmp0_b32 %[m], %[bit]\n\ts_cbranch_scc1 1f\n\t"
                     CRP_DPPF("%0", "%4") CRP_DPPF("%1", "%5") CRP_DPPF("%2", "%6") CRP_DPPF("%3", "%7") "1:"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1])
                     : "v"(slot[0].x), "v"(slot[0].y), "v"(slot[1].x), "v"(slot[1].y), [vv] "v"(vv), [m] "s"(mask),
                       [bit] "n"(BIT), [ln] "n"(LANE)
                     : "scc");
    else if constexpr (NV == 1 && VW == 2)
        asm volatile("s_bitcmp0_b32 %[m], %[bit]\n\ts_cbranch_scc1 1f\n\t" CRP_DPPF("%0", "%2") CRP_DPPF("%1", "%3") "1:"
                     : "+v"(acc[0][0]), "+v"(acc[0][1])
                     : "v"(slot[0].x), "v"(slot[0].y), [vv] "v"(vv), [m] "s"(mask), [bit] "n"(BIT), [ln] "n"(LANE)
                     : "scc");
    else if constexpr (NV == 2 && VW == 1)
        asm volatile("s_bitcmp0_b32 %[m], %[bit]\n\ts_cbranch_scc1 1f\n\t" CRP_DPPF("%0", "%2") CRP_DPPF("%1", "%3") "1:"
                     : "+v"(acc[0][0]), "+v"(acc[1][0])
                     : "v"(slot[0]), "v"(slot[1]), [vv] "v"(vv), [m] "s"(mask), [bit] "n"(BIT), [ln] "n"(LANE)
                     : "scc");
    else
    {
        static_assert(NV == 1 && VW == 1, "unsupported tile shape");
        asm volatile("s_bitcmp0_b32 %[m], %[bit]\n\ts_cbranch_scc1 1f\n\t" CRP_DPPF("%0", "%1") "1:"
                     : "+v"(acc[0][0])
                     : "v"(slot[0]), [vv] "v"(vv), [m] "s"(mask), [bit] "n"(BIT), [ln] "n"(LANE)
                     : "scc");
    }
#undef CRP_DPPF
}

template <int R, int NV, int VW, int LANE0, int BIT = 0>
__device__ __forceinline__ void fmac_rows_dpp(double (&acc)[R][NV][VW], const double vv,
                                              const typename SlotT<VW>::type (&slot)[NV], const uint32_t mask)
{
    fmac_row_masked_dpp<NV, VW, BIT, LANE0 + BIT>(acc[BIT], vv, slot, mask);
    if constexpr (BIT + 1 < R) fmac_rows_dpp<R, NV, VW, LANE0, BIT + 1>(acc, vv, slot, mask);
}

// Narrow tiles (one vector access per lane): the scalar unit, shared by the four SIMDs of a CU, is
// what bounds them -- s_bitcmp + s_cbranch per row is 16 scalar instructions per entry against 8 FMAs.
// Here the row mask goes into EXEC instead: one s_bfe_i64 per row writes all-ones or zero, the FMA
// that follows runs for all lanes or for none (so an absent pair still multiplies nothing), EXEC
// is restored at the end.  All eight rows sit in ONE asm statement so that no compiler-scheduled
// vector instruction can land between the EXEC writes.
template <int VW>
__device__ __forceinline__ void fmac_rows_exec8(double (&acc)[8][1][VW], const double (&a)[8],
                                                const typename SlotT<VW>::type (&slot)[1], const uint32_t mask)
{
    const uint64_t m64 = mask;
    if constexpr (VW == 1)
        asm volatile("s_cmp_eq_u32 %[m32], 0xff\n\ts_cbranch_scc0 .Lpm1%=\n\t"
                     "v_fmac_f64 %0, %8, %16\n\tv_fmac_f64 %1, %9, %16\n\tv_fmac_f64 %2, %10, %16\n\tv_fmac_f64 %3, %11, %16\n\t"
                     "v_fmac_f64 %4, %12, %16\n\tv_fmac_f64 %5, %13, %16\n\tv_fmac_f64 %6, %14, %16\n\tv_fmac_f64 %7, %15, %16\n\t"
                     "s_branch .Lpd1%=\n.Lpm1%=:\n\t"
                     "s_bfe_i64 exec, %[m], 0x10000\n\tv_fmac_f64 %0, %8, %16\n\t"
                     "s_bfe_i64 exec, %[m], 0x10001\n\tv_fmac_f64 %1, %9, %16\n\t"
                     "s_bfe_i64 exec, %[m], 0x10002\n\tv_fmac_f64 %2, %10, %16\n\t"
                     "s_bfe_i64 exec, %[m], 0x10003\n\tv_fmac_f64 %3, %11, %16\n\t"
                     "s_bfe_i64 exec, %[m], 0x10004\n\tv_fmac_f64 %4, %12, %16\n\t"
                     "s_bfe_i64 exec, %[m], 0x10005\n\tv_fmac_f64 %5, %13, %16\n\t"
                     "s_bfe_i64 exec, %[m], 0x10006\n\tv_fmac_f64 %6, %14, %16\n\t"
                     "s_bfe_i64 exec, %[m], 0x10007\n\tv_fmac_f64 %7, %15, %16\n\t"
                     "s_mov_b64 exec, -1\n.Lpd1%=:"
                     : "+v"(acc[0][0][0]), "+v"(acc[1][0][0]), "+v"(acc[2][0][0]), "+v"(acc[3][0][0]),
                       "+v"(acc[4][0][0]), "+v"(acc[5][0][0]), "+v"(acc[6][0][0]), "+v"(acc[7][0][0])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
                       "v"(slot[0]), [m] "s"(m64), [m32] "s"(mask)
                     : "scc");
    else
        asm volatile("s_cmp_eq_u32 %[m32], 0xff\n\ts_cbranch_scc0 .Lpm2%=\n\t"
                     "v_fmac_f64 %0, %16, %24\n\tv_fmac_f64 %1, %16, %25\n\tv_fmac_f64 %2, %17, %24\n\tv_fmac_f64 %3, %17, %25\n\t"
                     "v_fmac_f64 %4, %18, %24\n\tv_fmac_f64 %5, %18, %25\n\tv_fmac_f64 %6, %19, %24\n\tv_fmac_f64 %7, %19, %25\n\t"
                     "v_fmac_f64 %8, %20, %24\n\tv_fmac_f64 %9, %20, %25\n\tv_fmac_f64 %10, %21, %24\n\tv_fmac_f64 %11, %21, %25\n\t"
                     "v_fmac_f64 %12, %22, %24\n\tv_fmac_f64 %13, %22, %25\n\tv_fmac_f64 %14, %23, %24\n\tv_fmac_f64 %15, %23, %25\n\t"
                     "s_branch .Lpd2%=\n.Lpm2%=:\n\t"
                     "s_bfe_i64 exec, %[m], 0x10000\n\tv_fmac_f64 %0, %16, %24\n\tv_fmac_f64 %1, %16, %25\n\t"
                     "s_bfe_i64 exec, %[m], 0x10001\n\tv_fmac_f64 %2, %17, %24\n\tv_fmac_f64 %3, %17, %25\n\t"
                     "s_bfe_i64 exec, %[m], 0x10002\n\tv_fmac_f64 %4, %18, %24\n\tv_fmac_f64 %5, %18, %25\n\t"
                     "s_bfe_i64 exec, %[m], 0x10003\n\tv_fmac_f64 %6, %19, %24\n\tv_fmac_f64 %7, %19, %25\n\t"
                     "s_bfe_i64 exec, %[m], 0x10004\n\tv_fmac_f64 %8, %20, %24\n\tv_fmac_f64 %9, %20, %25\n\t"
                     "s_bfe_i64 exec, %[m], 0x10005\n\tv_fmac_f64 %10, %21, %24\n\tv_fmac_f64 %11, %21, %25\n\t"
                     "s_bfe_i64 exec, %[m], 0x10006\n\tv_fmac_f64 %12, %22, %24\n\tv_fmac_f64 %13, %22, %25\n\t"
                     "s_bfe_i64 exec, %[m], 0x10007\n\tv_fmac_f64 %14, %23, %24\n\tv_fmac_f64 %15, %23, %25\n\t"
                     "s_mov_b64 exec, -1\n.Lpd2%=:"
                     : "+v"(acc[0][0][0]), "+v"(acc[0][0][1]), "+v"(acc[1][0][0]), "+v"(acc[1][0][1]),
                       "+v"(acc[2][0][0]), "+v"(acc[2][0][1]), "+v"(acc[3][0][0]), "+v"(acc[3][0][1]),
                       "+v"(acc[4][0][0]), "+v"(acc[4][0][1]), "+v"(acc[5][0][0]), "+v"(acc[5][0][1]),
                       "+v"(acc[6][0][0]), "+v"(acc[6][0][1]), "+v"(acc[7][0][0]), "+v"(acc[7][0][1])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
                       "v"(slot[0].x), "v"(slot[0].y), [m] "s"(m64), [m32] "s"(mask)
                     : "scc");
}

template <int R, int NV, int VW>
__device__ __forceinline__ void panel_consume1(const typename SlotT<VW>::type (&slot)[NV], const uint32_t mask,
                                               const double (&a)[R], double (&acc)[R][NV][VW])
{
#if defined(CRP_ABL_FULLMASK)   // timing experiment only: every row takes the FMA path (wrong results)
    fmac_rows<R, NV, VW>(acc, a, slot, (uint32_t) __builtin_amdgcn_readfirstlane((int) (mask | 0xFFu)));
#elif defined(CRP_ABL_NOFMA)    // timing experiment only: no row takes the FMA path
    fmac_rows<R, NV, VW>(acc, a, slot, (uint32_t) __builtin_amdgcn_readfirstlane((int) (mask & 0u)));
#else
    if constexpr (R == 8 && NV == 1)
        fmac_rows_exec8<VW>(acc, a, slot, (uint32_t) __builtin_amdgcn_readfirstlane((int) mask));
    else
        fmac_rows<R, NV, VW>(acc, a, slot, (uint32_t) __builtin_amdgcn_readfirstlane((int) mask));
#endif
}

// DEPTH = number of 8-slot ring sets: round r lives in set r % DEPTH and is refilled, slot by
// slot, with round r + DEPTH while it is consumed, so 8*DEPTH - 1 entries stay in flight.
// WPW = waves per workgroup: 4, or 6 when the processing order was laid out for teams of six panels
template <int R, int NV, int VW, int DEPTH, bool ADDR64, bool HAS_B1, int WPW, bool DPP>
__global__ __launch_bounds__(64 * WPW) void spmm_panel_f64_kernel(
    const int norder, const int nrow, const int n, const int *__restrict__ porder,
    const int *__restrict__ pptr, const int *__restrict__ pcol, const uint32_t *__restrict__ pmask4,
    const double *__restrict__ pval,
    const double *__restrict__ B0, const int64_t ldB0, const double *__restrict__ B1, const int64_t ldB1,
    double *__restrict__ C, const int64_t ldC, const int *__restrict__ rowmap, const int *__restrict__ psync)
{
    constexpr int TW = 64 * VW * NV;
    constexpr int RING = PANEL_RING;             // entries per round = slots per ring set
    constexpr int CHUNK = PANEL_CHUNK;           // entries whose values are staged in LDS at a time
    constexpr int RPC = CHUNK / RING;            // rounds per chunk
    constexpr int NS = (CHUNK * R) / 128;        // staging loads per lane and chunk (16 B each, 64 lanes)
    static_assert(NS >= 1 && NS * 128 == CHUNK * R, "chunk must be a whole number of wave-wide 16-byte loads");
    typedef typename SlotT<VW>::type ST;
    __shared__ __attribute__((aligned(16))) double lds_vals[WPW][2][CHUNK * R];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share
    // one), so give XCD x the contiguous block range [x*cpx, (x+1)*cpx): neighbouring panels
    // read neighbouring B rows and then meet in the same 4 MiB L2 instead of pulling every B
    // row into all eight.  Placement only affects speed, never the result.
    const int cpx   = (gridDim.x + 7) >> 3;
    const int wg    = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    const int slot_id = __builtin_amdgcn_readfirstlane((int) (wg * WPW + wave));
    if (slot_id >= norder) return;
    // one 16-byte record per position: panel (-1: none), first entry, rounds -- a single scalar load
    // instead of the chain order -> panel -> entry range
    const int4 rec = reinterpret_cast<const int4 *>(porder)[slot_id];
    const int panel = rec.x;
    // team schedule: the four waves of the workgroup meet at a barrier before each of the first nb
    // rounds (nb < the rounds of every panel of the team), so that they reach shared B rows together
#ifndef CRP_TEAM_SYNC_K
#define CRP_TEAM_SYNC_K 1
#endif
    const int nb = psync ? psync[wg] : 0;
    if (panel < 0)
    {
        for (int r = 0; r < nb; r += CRP_TEAM_SYNC_K) asm volatile("s_barrier" ::: "memory");
        return;
    }
    double *myvals = &lds_vals[__builtin_amdgcn_readfirstlane(wave)][0][0];

    const int col0 = blockIdx.y * TW + lane * VW;
    bool ok[NV];
    int  coff[NV], voff[NV];
#pragma unroll
    for (int v = 0; v < NV; v++)
    {
        const int c = col0 + v * 64 * VW;
        ok[v]   = (c + VW - 1) < n;
        coff[v] = ok[v] ? c : 0;
        voff[v] = coff[v] * 8;
    }
    BSource<ADDR64> s0, s1;
    s0.base = B0; s0.ld = ldB0; s0.ldbytes = (uint32_t) (ldB0 * 8);
    s1.base = B1; s1.ld = ldB1; s1.ldbytes = (uint32_t) (ldB1 * 8);
    if constexpr (!ADDR64)
    {
        s0.rsrc = make_rsrc(B0);
        s1.rsrc = make_rsrc(HAS_B1 ? B1 : B0);
    }

    double acc[R][NV][VW];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int v = 0; v < NV; v++)
#pragma unroll
            for (int w = 0; w < VW; w++) acc[r][v][w] = 0.0;

    const int q0 = rec.y;
    const int nr = rec.z;                              // rounds of this panel (entry counts are padded to whole rounds)
    if (nr > 0)
    {
        ST ring[DEPTH][RING][NV];
        d2 stage[NS];                                   // next chunk's values of entry `lane`, in flight
        int cA[RING];                                   // column indices of the round refilled next
        int cP[RING];                                   // ... and of the round after it (scalar loads run two rounds ahead)

        // ---- prologue: the values of chunk 0 are requested first (asm-issued like every load of the
        // loop), the B rows of rounds 0..DEPTH-1 right behind them, and only then are the values
        // waited for and parked in LDS: one memory latency instead of two in a row per wave
        {
            // the chunk's CHUNK*R values are contiguous: lane l moves doubles [2*NS*l, 2*NS*(l+1))
            const char *src = reinterpret_cast<const char *>(pval + (int64_t) q0 * R + lane * (2 * NS));
#pragma unroll
            for (int t = 0; t < NS; t++) glb_load_asm<2>(stage[t], src + 16 * t);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
        {
            if (d < nr)
            {
                int c0[RING];
#pragma unroll
                for (int k = 0; k < RING; k++) c0[k] = pcol[q0 + d * RING + k];
#pragma unroll
                for (int k = 0; k < RING; k++) panel_issue1<ADDR64, HAS_B1, NV, VW>(ring[d][k], c0[k], s0, s1, voff);
            }
        }
        {
            // younger than the staging loads: the ring loads just issued (at least one set; waiting for
            // "at most RING*NV outstanding" only over-waits when a second set was issued)
            wait_vmcnt<RING * NV>();
            double *dst = myvals + lane * (2 * NS);
#pragma unroll
            for (int t = 0; t < NS; t++) *reinterpret_cast<d2 *>(dst + 2 * t) = stage[t];
        }
        if constexpr (DPP)
        {
            // hipcc allocates the loop's ring registers apart from the prologue's in this body and moves
            // them with v_mov before the loop: the moves must not read registers whose loads are still in
            // flight, so the prologue's loads are drained here and the ring is pinned behind the wait
            // (costs one exposed latency per panel; the loop itself never copies a ring register).
            wait_vmcnt<0>();
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
#pragma unroll
                for (int k = 0; k < RING; k++)
#pragma unroll
                    for (int v = 0; v < NV; v++) asm volatile("" : "+v"(ring[d][k][v]));
        }
#pragma unroll
        for (int k = 0; k < RING; k++) cA[k] = pcol[q0 + DEPTH * RING + k];     // (arrays are padded)
#pragma unroll
        for (int k = 0; k < RING; k++) cP[k] = pcol[q0 + (DEPTH + 1) * RING + k];

        // One round: consume the 8 slots of ring set `set` (entries of round r) and, when `refill`,
        // re-issue each slot for round r + DEPTH right after its FMAs.  All VMEM of the loop is issued
        // from asm, so every wait below is hand-counted: N = loads YOUNGER than the one waited for.
        auto round = [&](const int r, auto set_tag, auto refill_tag) {
            constexpr int  set = decltype(set_tag)::value;
            constexpr bool refill = decltype(refill_tag)::value;
            const int q = q0 + r * RING;
            int cB[RING];
#pragma unroll
            for (int k = 0; k < RING; k++) cB[k] = pcol[q + (DEPTH + 2) * RING + k];   // scalar loads, two rounds ahead
            const uint32_t m_lo = pmask4[(q >> 2)], m_hi = pmask4[(q >> 2) + 1];
            // values: chunk c = r / RPC sits in LDS buffer c & 1; the last round of a chunk fetches the
            // next chunk (every lane one entry; reads past the panel are padded) before its refills ...
            const bool last_of_chunk = ((r % RPC) == RPC - 1) && (r + 1 < nr);
            if (last_of_chunk)
            {
                const char *src = reinterpret_cast<const char *>(pval + (int64_t) (q + RING) * R + lane * (2 * NS));
#pragma unroll
                for (int t = 0; t < NS; t++) glb_load_asm<2>(stage[t], src + 16 * t);
            }
            const double *lv = myvals + ((r / RPC) & 1) * (CHUNK * R) + (r % RPC) * (RING * R);
            // DPP: the round's RING * R values in RING * R / 16 reads, lane l holding value l & 15 of its 16
            constexpr int NVV = (RING * R) / 16;
            double vv[NVV];
            if constexpr (DPP)
            {
#pragma unroll
                for (int j = 0; j < NVV; j++) vv[j] = lv[j * 16 + (lane & 15)];
            }
            static_for<0, RING>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                double a_cur[R];
                if constexpr (!DPP)
                {
#pragma unroll
#ifdef CRP_ABL_NOVALS      // timing experiment: no LDS broadcast of the values
                    for (int rr = 0; rr < R; rr++) a_cur[rr] = 1.0 + rr;
#else
                    for (int rr = 0; rr < R; rr++) a_cur[rr] = lv[k * R + rr];      // uniform-address LDS broadcast
#endif
                }
                const uint32_t mask = ((k < 4 ? m_lo : m_hi) >> (8 * (k & 3))) & 0xFFu;
                // younger than slot k of this round: slots k+1..7 of the round, the DEPTH-1 rounds issued
                // after it, and (refill) the k slots re-issued so far; the staging loads of a chunk's
                // last round are ignored (N too small only over-waits).  Rounds without refill use the
                // bound that is exact for the panel's last round.
                if constexpr (refill) wait_vmcnt<(RING * DEPTH - 1) * NV>();
                else
                {
                    switch (k)
                    {
                        case 0: wait_vmcnt<7 * NV>(); break;
                        case 1: wait_vmcnt<6 * NV>(); break;
                        case 2: wait_vmcnt<5 * NV>(); break;
                        case 3: wait_vmcnt<4 * NV>(); break;
                        case 4: wait_vmcnt<3 * NV>(); break;
                        case 5: wait_vmcnt<2 * NV>(); break;
                        case 6: wait_vmcnt<1 * NV>(); break;
                        default: wait_vmcnt<0>(); break;
                    }
                }
                if constexpr (DPP)
                    fmac_rows_dpp<R, NV, VW, ((k * R) % 16)>(acc, vv[(k * R) / 16], ring[set][k],
                                                             (uint32_t) __builtin_amdgcn_readfirstlane((int) mask));
                else panel_consume1<R, NV, VW>(ring[set][k], mask, a_cur, acc);
                if constexpr (refill) panel_issue1<ADDR64, HAS_B1, NV, VW>(ring[set][k], cA[k], s0, s1, voff);
            });
            if (last_of_chunk)
            {
                // ... and parks it in the other LDS buffer once it has landed: younger than the last
                // staging load are exactly this round's refills
                if constexpr (refill) wait_vmcnt<RING * NV>();
                else wait_vmcnt<0>();
                double *dst = myvals + (((r / RPC) + 1) & 1) * (CHUNK * R) + lane * (2 * NS);
#pragma unroll
                for (int t = 0; t < NS; t++) *reinterpret_cast<d2 *>(dst + 2 * t) = stage[t];
            }
#pragma unroll
            for (int k = 0; k < RING; k++)
            {
                cA[k] = cP[k];
                cP[k] = cB[k];
            }
        };

        int r = 0;
        if constexpr (DEPTH == 1)
        {
            for (; r + 1 < nr; r++)
            {
                if (r < nb && (r % CRP_TEAM_SYNC_K) == 0) asm volatile("s_barrier" ::: "memory");
                round(r, std::integral_constant<int, 0>{}, std::true_type{});
            }
            round(r, std::integral_constant<int, 0>{}, std::false_type{});
        }
        else
        {
            static_assert(DEPTH == 2, "ring depth is 1 or 2 sets");
            for (; r + 3 < nr; r += 2)
            {
                round(r, std::integral_constant<int, 0>{}, std::true_type{});
                round(r + 1, std::integral_constant<int, 1>{}, std::true_type{});
            }
            // tail: 1 to 3 rounds left; only a round that still has a successor DEPTH ahead refills
            if (r + 2 < nr)
            {
                round(r, std::integral_constant<int, 0>{}, std::true_type{});
                round(r + 1, std::integral_constant<int, 1>{}, std::false_type{});
                round(r + 2, std::integral_constant<int, 0>{}, std::false_type{});
            }
            else if (r + 1 < nr)
            {
                round(r, std::integral_constant<int, 0>{}, std::false_type{});
                round(r + 1, std::integral_constant<int, 1>{}, std::false_type{});
            }
            else round(r, std::integral_constant<int, 0>{}, std::false_type{});
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
    {
        const int row = panel * R + r;
        if (row < nrow)
        {
            double *crow = C + (int64_t) (rowmap ? rowmap[row] : row) * ldC;
#pragma unroll
            for (int v = 0; v < NV; v++)
                if (ok[v])
                {
                    if constexpr (VW == 2)
                    {
                        d2 t;
                        t.x = acc[r][v][0];
                        t.y = acc[r][v][1];
                        CRP_STORE(t, reinterpret_cast<d2 *>(crow + coff[v]));
                    }
                    else CRP_STORE(acc[r][v][0], crow + coff[v]);
                }
        }
    }
}

template <int R, int NV, int VW, int DEPTH, bool ADDR64, bool HAS_B1, int WPW>
static hipError_t launch_panel_w(const PanelArgs &p, const SpmmArgs &a, hipStream_t s)
{
    constexpr int TW = 64 * VW * NV;
    const int nwg = (p.norder + WPW - 1) / WPW;
    dim3 grid((nwg + 7) / 8 * 8, (a.n + TW - 1) / TW);      // multiple of 8 for the XCD remap
    // value delivery: uniform-address LDS broadcast (the DPP row_newbcast body needs 216 VGPRs against 160: not instantiated)
    hipLaunchKernelGGL((spmm_panel_f64_kernel<R, NV, VW, DEPTH, ADDR64, HAS_B1, WPW, false>), grid, dim3(64 * WPW), 0, s, p.norder, a.nrow,
                       a.n, p.porder, p.pptr, p.pcol, p.pmask4, p.pval, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC, a.rowmap, p.psync);
    return hipGetLastError();
}

template <int R, int NV, int VW, int DEPTH, bool ADDR64, bool HAS_B1>
static hipError_t launch_panel(const PanelArgs &p, const SpmmArgs &a, hipStream_t s)
{
    if (p.team_waves != 4) return hipErrorInvalidValue;      // (the order is laid out for four-wave workgroups)
    return launch_panel_w<R, NV, VW, DEPTH, ADDR64, HAS_B1, 4>(p, a, s);
}

template <int R, int NV, int VW>
static hipError_t launch_panel_addr(const PanelArgs &p, const SpmmArgs &a, hipStream_t s)
{
    // 32-bit buffer offsets need every addressed byte of B0 / B1 below 4 GiB
    const bool has_b1 = (a.B1 != nullptr) && (p.b1_rows > 0);
    const bool small = ((uint64_t) p.b0_rows * (uint64_t) a.ldB0 * 8ull < (1ull << 32)) &&
                       (!has_b1 || (uint64_t) p.b1_rows * (uint64_t) a.ldB1 * 8ull < (1ull << 32));
    // ring depth: one set of 8 slots (two sets, 15 entries in flight, need 326 VGPRs and ran 40 % slower: not instantiated)
    if (small)
        return has_b1 ? launch_panel<R, NV, VW, 1, false, true>(p, a, s) : launch_panel<R, NV, VW, 1, false, false>(p, a, s);
    return has_b1 ? launch_panel<R, NV, VW, 1, true, true>(p, a, s) : launch_panel<R, NV, VW, 1, true, false>(p, a, s);
}

// The row-panel kernels need one lane per column (pair): below ~24 columns most lanes of the
// wave would idle and the CSR row-group kernel (several rows per wave) is the better shape.
bool spmm_panel_applicable(const SpmmArgs &a) { return a.n >= 24; }

template <int R>
static hipError_t launch_panel_shape(const PanelArgs &p, const SpmmArgs &a, hipStream_t s)
{
    const bool vec2 = (a.n % 2 == 0) && (a.ldB0 % 2 == 0) && (a.ldC % 2 == 0) && (a.B1 == nullptr || a.ldB1 % 2 == 0) &&
                      (((uintptr_t) a.B0 | (uintptr_t) a.B1 | (uintptr_t) a.C) % 16 == 0);
    if (vec2 && a.n > 128) return launch_panel_addr<R, 2, 2>(p, a, s);   // 256-column tiles, 16 B per lane
    if (vec2 && a.n > 64)  return launch_panel_addr<R, 1, 2>(p, a, s);   // 128-column tile
    if (a.n > 64) return launch_panel_addr<R, 2, 1>(p, a, s);            // odd / unaligned operands: 8 B per lane
    return launch_panel_addr<R, 1, 1>(p, a, s);                          // n <= 64: one column per lane
}

hipError_t spmm_rm_f64_panel(const PanelArgs &p_, const SpmmArgs &a, hipStream_t s)
{
    PanelArgs p = p_;
    // team schedule: the per-round workgroup barrier pays from 128 columns on (n = 128: 0.204 -> 0.184 ms,
    // n = 1024: 1.40 -> 1.27 ms on the pwtk stand-in); at n <= 64 the rounds are too short (0.096 -> 0.099 ms)
    if (a.n <= 64) p.psync = nullptr;
    if (spmm_narrow_applicable(p, a)) return spmm_rm_f64_narrow(p, a, s);      // narrow_kernel.hip
    if (p.R == 4) return launch_panel_shape<4>(p, a, s);
    if (p.R == 8) return launch_panel_shape<8>(p, a, s);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// Column-major kernel (BC_layout = 1, /root/reference/src/rowpara_spmm.c:403
// with SPARSE_LAYOUT_COLUMN_MAJOR).  Element (r, j) of B/C sits at r + j*ld.
// One lane per (row, column) pair: 64 consecutive rows per wave so that the C
// store and the val/col loads are coalesced; B accesses are a true gather.
// The reference's own drivers never use this layout (examples/test_rp_spmm.c:109).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spmm_cm_f64_kernel(
    const int nrow, const int n,
    const int *__restrict__ rowptr, const int *__restrict__ colidx, const double *__restrict__ val,
    const double *__restrict__ B0, const int64_t ldB0,
    const double *__restrict__ B1, const int64_t ldB1,
    double *__restrict__ C, const int64_t ldC, const int *__restrict__ rowmap)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int j   = blockIdx.y;
    if (row >= nrow) return;
    const double *b0 = B0 + (int64_t) j * ldB0;
    const double *b1 = B1 + (int64_t) j * ldB1;
    double acc = 0.0;
    const int pe = rowptr[row + 1];
    for (int p = rowptr[row]; p < pe; p++)
    {
        const int c = colidx[p];
        const double b = (c >= 0) ? b0[c] : b1[~c];
        acc = fma(val[p], b, acc);
    }
    C[(int64_t) j * ldC + (rowmap ? rowmap[row] : row)] = acc;
}

hipError_t spmm_cm_f64(const SpmmArgs &a, hipStream_t s)
{
    dim3 grid((a.nrow + 255) / 256, a.n);
    hipLaunchKernelGGL(spmm_cm_f64_kernel, grid, dim3(256), 0, s,
                       a.nrow, a.n, a.rowptr, a.colidx, a.val, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC, a.rowmap);
    return hipGetLastError();
}

}  // namespace crp
