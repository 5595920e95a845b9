// team2p_kernel.hip -- the LDS-sharing CSR x dense SpMM of team2_kernel.hip as PERSISTENT workgroups ("team2-R8", variant 5).
//
// Same product as every kernel of this library: what mkl_sparse_d_mm computes at
// /root/reference/src/rowpara_spmm.c:388-408 (alpha = 1, beta = 0), C[i][:] = sum_p val[p] * B[col[p]][:].
//
// team2_kernel.hip gives every team of 8 row panels a workgroup of its own.  Its `s_memtime` stamps (round 3) put 11 % of a
// workgroup's life into start-up -- the launch-grid entry, the team's descriptors, the first DMAs' trip to HBM and back, all of it
// with an empty ring -- and 4 % into the stores of its C tile at the end.  Here a workgroup works through a CHAIN of teams
// (panel_format.h, Team2Host::chain): the rounds of the chain's teams form ONE sequence in the record and value streams, the
// ring pipeline (D = 3 rounds of B row slices, values and records in flight) runs across the team boundaries, and a team's C
// rows leave from inside the round loop (record flag FLUSH) while the next team's first rounds are already on their way.
// Start-up is paid once per chain; nothing drains at a team's end.
//
// The whole life of a workgroup after the prologue is one asm statement (team2p_consume.inc, tools/gen_team2_asm.py --persist),
// which owns the accumulators (fixed registers v32..v95) and the C stores.  Ring, records, values, parts, the computed calls into
// straight-line FMA code: as in team2_kernel.hip.
//
// Tile: NV 16-byte pieces per lane and row (fp64 128 NV columns per workgroup, fp32 256 NV); wider operands: grid.y tiles.
// Operands 16-byte aligned, n and the leading dimensions multiples of the elements in 16 bytes; row strides of B and C below
// 4 GiB.  Column indices carry the two-source encoding (c >= 0: B0 row c, c < 0: B1 row ~c).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"
#include "panel_format.h"
#include "team2p_consume.inc"

namespace crp {

namespace {

constexpr int T2_D = TEAM2_D;
constexpr int T2_NSET = T2_D + 1;

template <typename T> constexpr int t2_vgrp() { return 8 * (int) sizeof(T); }           // bytes of 8 values
template <typename T> constexpr int t2_vslot() { return TEAM2_CAP * t2_vgrp<T>(); }      // ... of one round's value block at its largest (one LDS-DMA)
constexpr int T2_VHEAD = 64;                                                             // bytes in front of a wave's value slots
template <typename T> constexpr int t2_vring() { return T2_VHEAD + T2_NSET * t2_vslot<T>(); }   // ... of one wave's value ring
constexpr int t2_ring_bytes(int NV, int TW) { return T2_NSET * TW * NV * 1024; }
constexpr int t2_rec_bytes(int TW) { return 2 * 8 * TW * 16; }            // two record blocks of 8 rounds x TW waves x 16 bytes
template <typename T> constexpr int t2_lds_bytes(int NV, int TW) { return t2_ring_bytes(NV, TW) + TW * t2_vring<T>() + t2_rec_bytes(TW); }

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define T2_GPTR(p) ((const __attribute__((address_space(1))) void *) (p))
#define T2_LPTR(p) ((__attribute__((address_space(3))) void *) (p))

template <int N>
__device__ __forceinline__ void t2_wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// torder = the launch grid of CHAINS (8 runs, -1 = none); tinfo / tpro / tvoff per chain; trows = the C rows of the panels,
// [chain member (in cptr order)][wave][8] (-1 = no such row), filled on the device from the panels and the row map.
template <typename T, int NV, bool HAS_B1, int TW, bool COMPACT>
__global__ __launch_bounds__(64 * TW, 4) void spmm_team2p_kernel(
    const int n, const int *__restrict__ torder, const int *__restrict__ cptr,
    const int *__restrict__ tinfo, const int *__restrict__ tpro, const uint32_t *__restrict__ trec,
    const long long *__restrict__ tvoff, const T *__restrict__ tval, const int *__restrict__ trows,
    const T *__restrict__ B0, const int64_t ldB0, const T *__restrict__ B1, const int64_t ldB1,
    T *__restrict__ C, const int64_t ldC)
{
    constexpr int VW = 16 / (int) sizeof(T);           // elements per 16-byte piece
    constexpr int PCOLS = 64 * VW;                     // columns of one piece across the wave
    constexpr int SLOTB = NV * 1024;                   // bytes of one ring slot
    constexpr int SETB = TW * SLOTB;
    constexpr int OPR = NV + 1;                        // DMAs a wave issues per round
    constexpr int VSLOT = t2_vslot<T>(), VRING = t2_vring<T>();
    constexpr bool F32 = sizeof(T) == 4;
    extern __shared__ __attribute__((aligned(16))) char t2_lds[];
    char *const ring = t2_lds;
    char *const vring_all = t2_lds + t2_ring_bytes(NV, TW);
    char *const recs = vring_all + TW * VRING;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int tix  = (int) blockIdx.y;                                  // column tile of this workgroup
    const int cpx  = (int) (gridDim.x >> 3);                            // grid.x is a multiple of 8: XCD x takes a contiguous range
    const int wg   = (int) (blockIdx.x & 7) * cpx + (int) (blockIdx.x >> 3);
    const int chain = torder[wg];
    if (chain < 0) return;                             // whole workgroup leaves: no barrier is skipped
    const int nr = tinfo[chain * 4];                   // rounds of the chain: >= 1 (every team has at least one)
    const int recblk0 = tinfo[chain * 4 + 1];
    // the C rows of this wave's panel of the chain's first team (lanes 0..7), and where the next teams' rows are
    const int *const rowp = trows + ((int64_t) cptr[chain] * TW + wave) * 8;
    const int rows0 = rowp[lane & 7];
    asm volatile("" ::"v"(rows0));                     // (waited for HERE, in front of the DMAs -- not with vmcnt(0) behind them)

    // this lane's pieces of a row.  Lanes past n read valid bytes of the same row that are never stored: the tile's
    // first piece (piece 0) or the piece before the tile's second one (piece 1, addressed with offset:1024).
    const int colbase = tix * (NV * PCOLS);
    const bool ok0 = (colbase + lane * VW + VW - 1) < n;
    const bool ok1 = (NV == 2) && (colbase + PCOLS + lane * VW + VW - 1) < n;
    const uint32_t cbb = (uint32_t) colbase * (uint32_t) sizeof(T);
    const uint32_t voffa = ok0 ? cbb + (uint32_t) lane * 16u : cbb;
    const uint32_t voffb = ok1 ? cbb + (uint32_t) lane * 16u : (colbase > 0 ? cbb - 1024u : 0u);
    const uint32_t coff = cbb + (uint32_t) lane * 16u;                  // byte offset of this lane's piece 0 inside a C row
    const uint64_t okm0 = __ballot(ok0), okm1 = __ballot(ok1);
    const char *const B0b = reinterpret_cast<const char *>(B0);
    const char *const B1b = reinterpret_cast<const char *>(B1);
    const uint32_t ld0 = (uint32_t) (ldB0 * (int64_t) sizeof(T)), ld1 = (uint32_t) (ldB1 * (int64_t) sizeof(T));   // (the launcher checks they fit)
    const uint32_t ldc = (uint32_t) (ldC * (int64_t) sizeof(T));

    char *const vring = vring_all + wave * VRING;
    constexpr int VUNITB = TEAM2_VUNIT * (int) sizeof(T);
    const char *const vbase = reinterpret_cast<const char *>(tval + tvoff[chain * TW + wave] * TEAM2_VUNIT);
    const uint32_t lane16 = (uint32_t) lane * 16u;
    auto issue_round = [&](const int col, const uint32_t voff, const int round) {
        if (lane < VSLOT / 16)
            __builtin_amdgcn_global_load_lds(T2_GPTR(vbase + (int64_t) voff * VUNITB + lane16), T2_LPTR(vring + T2_VHEAD + (round % T2_NSET) * VSLOT), 16, 0, 0);
        const char *rb = (!HAS_B1 || col >= 0) ? (B0b + (uint64_t) (uint32_t) col * ld0) : (B1b + (uint64_t) (uint32_t) (~col) * ld1);
        char *dst = ring + (round % T2_NSET) * SETB + wave * SLOTB;
        __builtin_amdgcn_global_load_lds(T2_GPTR(rb + voffa), T2_LPTR(dst), 16, 0, 0);
        if constexpr (NV == 2) __builtin_amdgcn_global_load_lds(T2_GPTR(rb + voffb + 1024), T2_LPTR(dst + 1024), 16, 0, 0);
    };

    // ---- prologue: the first record block, then rounds 0 .. D-1 (values + row each)
    const char *recsrc = reinterpret_cast<const char *>(trec) + (int64_t) recblk0 * (8 * TW * 16);
    if (wave == 0)
    {
#pragma unroll
        for (int piece = 0; piece < (8 * TW * 16) / 1024; piece++)
            __builtin_amdgcn_global_load_lds(T2_GPTR(recsrc + piece * 1024 + lane16), T2_LPTR(recs + piece * 1024), 16, 0, 0);
    }
    {
        const int *p0 = tpro + ((chain * T2_D) * TW + wave) * 2;
#pragma unroll
        for (int d = 0; d < T2_D; d++)
            if (d < nr)
                issue_round(__builtin_amdgcn_readfirstlane(p0[d * 2 * TW]), (uint32_t) __builtin_amdgcn_readfirstlane(p0[d * 2 * TW + 1]), d);
    }
    // ---- top of round 0: own DMAs of round 0 (and the record block, which is older) have landed
    if (nr >= T2_D) t2_wait_vmcnt<(T2_D - 1) * OPR>();
    else t2_wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    const u4 rv = *reinterpret_cast<const u4 *>(recs + (wave << 4));
    uint32_t w0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) rv.x);
    uint32_t w1 = (uint32_t) __builtin_amdgcn_readfirstlane((int) rv.y);
    uint32_t w2 = (uint32_t) __builtin_amdgcn_readfirstlane((int) rv.z);
    uint32_t w3 = (uint32_t) __builtin_amdgcn_readfirstlane((int) rv.w);
    // ---- the rounds of the chain (team2p_consume.inc).  Every read-write operand is early-clobber ("+&"): the statement is a
    // loop that writes them long before it has read its inputs for the last time.
    const uint32_t seta = (uint32_t) (uintptr_t) ring + lane16;                      // LDS address of this lane's piece in slot 0 of set 0
    const uint32_t seta2 = seta + 2u * SETB;                                         // ... of set 2 (TW = 16: DS offsets have 16 bits)
    const uint32_t vsl = (uint32_t) (uintptr_t) vring + (uint32_t) T2_VHEAD - 7u * (uint32_t) sizeof(T) + (uint32_t) (lane & 7) * (uint32_t) sizeof(T);
    const uint32_t recbase = (uint32_t) (uintptr_t) recs + ((uint32_t) wave << 4);
    uint32_t recoff = 0, recaddr = recbase;
    uint32_t recdst = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (uintptr_t) recs);
    const uint32_t wslot = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((uint32_t) (uintptr_t) ring + (uint32_t) wave * SLOTB));
    const uint32_t vringw = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((uint32_t) (uintptr_t) vring + (uint32_t) T2_VHEAD));
    const uint64_t vb = (uint64_t) vbase, b0 = (uint64_t) B0b, b1 = (uint64_t) B1b, rs = (uint64_t) recsrc, cb = (uint64_t) C, rt = (uint64_t) rowp;
    const uint32_t b0lo = (uint32_t) b0, b0hi = (uint32_t) (b0 >> 32), b1lo = (uint32_t) b1, b1hi = (uint32_t) (b1 >> 32);
    const uint32_t rslo = (uint32_t) rs, rshi = (uint32_t) (rs >> 32);
    const uint32_t clo = (uint32_t) cb, chi = (uint32_t) (cb >> 32);
    const uint32_t rtlo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) rt), rthi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (rt >> 32));
    const uint32_t rowoff = (uint32_t) (lane & 7) * 4u;
#define T2_STATE [w0] "+&s"(w0), [w1] "+&s"(w1), [w2] "+&s"(w2), [w3] "+&s"(w3), [recdst] "+&s"(recdst), [recoff] "+&v"(recoff), \
                 [recaddr] "+&v"(recaddr)
#define T2_IN [seta] "v"(seta), [vsl] "v"(vsl), [recbase] "v"(recbase), [lane16] "v"(lane16), [voffa] "v"(voffa), [voffb] "v"(voffb), \
              [coff] "v"(coff), [rowoff] "v"(rowoff), [rows0] "v"(rows0), \
              [wslot] "s"(wslot), [vringw] "s"(vringw), [vbase] "s"(vb), [b0lo] "s"(b0lo), [b0hi] "s"(b0hi), [ld0] "s"(ld0), \
              [rslo] "s"(rslo), [rshi] "s"(rshi), [clo] "s"(clo), [chi] "s"(chi), [ldc] "s"(ldc), [rtlo] "s"(rtlo), [rthi] "s"(rthi), \
              [ok0] "s"(okm0), [ok1] "s"(okm1)
#define T2_IN_B1 T2_IN, [b1lo] "s"(b1lo), [b1hi] "s"(b1hi), [ld1] "s"(ld1)
#define T2_IN_W T2_IN, [seta2] "v"(seta2)
#define T2_IN_B1_W T2_IN_B1, [seta2] "v"(seta2)
#define T2_CLOB "scc", "vcc", "memory", CRP_TEAM2P_CLOBBERS
    if constexpr (TW == 8 && !COMPACT)
    {
        if constexpr (!F32 && NV == 2 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV2_B0_F : T2_STATE : T2_IN : T2_CLOB);
        else if constexpr (!F32 && NV == 2 && HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV2_B1_F : T2_STATE : T2_IN_B1 : T2_CLOB);
        else if constexpr (!F32 && NV == 1 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV1_B0_F : T2_STATE : T2_IN : T2_CLOB);
        else if constexpr (!F32 && NV == 1 && HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV1_B1_F : T2_STATE : T2_IN_B1 : T2_CLOB);
        else if constexpr (F32 && NV == 2 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F32_NV2_B0_F : T2_STATE : T2_IN : T2_CLOB);
        else if constexpr (F32 && NV == 2 && HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F32_NV2_B1_F : T2_STATE : T2_IN_B1 : T2_CLOB);
        else if constexpr (F32 && NV == 1 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F32_NV1_B0_F : T2_STATE : T2_IN : T2_CLOB);
        else asm volatile(CRP_TEAM2P_LOOP_F32_NV1_B1_F : T2_STATE : T2_IN_B1 : T2_CLOB);
    }
    else if constexpr (TW == 8)
    {
        if constexpr (!F32 && NV == 2 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV2_B0 : T2_STATE : T2_IN : T2_CLOB);
        else if constexpr (!F32 && NV == 2 && HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV2_B1 : T2_STATE : T2_IN_B1 : T2_CLOB);
        else if constexpr (!F32 && NV == 1 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV1_B0 : T2_STATE : T2_IN : T2_CLOB);
        else if constexpr (!F32 && NV == 1 && HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV1_B1 : T2_STATE : T2_IN_B1 : T2_CLOB);
        else if constexpr (F32 && NV == 2 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F32_NV2_B0 : T2_STATE : T2_IN : T2_CLOB);
        else if constexpr (F32 && NV == 2 && HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F32_NV2_B1 : T2_STATE : T2_IN_B1 : T2_CLOB);
        else if constexpr (F32 && NV == 1 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F32_NV1_B0 : T2_STATE : T2_IN : T2_CLOB);
        else asm volatile(CRP_TEAM2P_LOOP_F32_NV1_B1 : T2_STATE : T2_IN_B1 : T2_CLOB);
    }
    else
    {
        if constexpr (!F32 && NV == 2 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV2_B0_W16 : T2_STATE : T2_IN_W : T2_CLOB);
        else if constexpr (!F32 && NV == 2 && HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV2_B1_W16 : T2_STATE : T2_IN_B1_W : T2_CLOB);
        else if constexpr (!F32 && NV == 1 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV1_B0_W16 : T2_STATE : T2_IN_W : T2_CLOB);
        else if constexpr (!F32 && NV == 1 && HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F64_NV1_B1_W16 : T2_STATE : T2_IN_B1_W : T2_CLOB);
        else if constexpr (F32 && NV == 2 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F32_NV2_B0_W16 : T2_STATE : T2_IN_W : T2_CLOB);
        else if constexpr (F32 && NV == 2 && HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F32_NV2_B1_W16 : T2_STATE : T2_IN_B1_W : T2_CLOB);
        else if constexpr (F32 && NV == 1 && !HAS_B1) asm volatile(CRP_TEAM2P_LOOP_F32_NV1_B0_W16 : T2_STATE : T2_IN_W : T2_CLOB);
        else asm volatile(CRP_TEAM2P_LOOP_F32_NV1_B1_W16 : T2_STATE : T2_IN_B1_W : T2_CLOB);
    }
#undef T2_STATE
#undef T2_IN
#undef T2_IN_B1
#undef T2_IN_W
#undef T2_IN_B1_W
#undef T2_CLOB
    (void) b1lo; (void) b1hi; (void) ld1; (void) seta2;
}

// The C rows of every (chain member, wave): trows[(k * tw + w) * 8 + r] = C row of row r of the panel wave w owns in the team at
// position k of cteam (through the row map when there is one), -1 = no such row.  Once per format and row map.
__global__ void team2p_fill_rows_kernel(const int nmember, const int tw, const int nrow, const int *__restrict__ cteam, const int *__restrict__ tpanel,
                                        const int *__restrict__ rowmap, int *__restrict__ trows)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long) nmember * tw * 8) return;
    const int r = (int) (i & 7);
    const long long kw = i >> 3;
    const int w = (int) (kw % tw);
    const int k = (int) (kw / tw);
    const int panel = tpanel[(long long) cteam[k] * tw + w];
    int row = -1;
    if (panel >= 0)
    {
        const long long rr = (long long) panel * 8 + r;
        if (rr < nrow) row = rowmap ? rowmap[rr] : (int) rr;
    }
    trows[i] = row;
}

hipError_t team2p_fill_rows(const Team2Args &t, int nrow, const int *rowmap, hipStream_t s)
{
    const long long total = (long long) t.nmember * t.tw * 8;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(team2p_fill_rows_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s, t.nmember, t.tw, nrow, t.cteam, t.tpanel, rowmap, t.trows);
    return hipGetLastError();
}

template <typename T, int NV, bool HAS_B1, int TW, bool COMPACT, typename ARGS>
static hipError_t launch_team2p(const Team2Args &t, const T *tval, const ARGS &a, hipStream_t s)
{
    constexpr int VW = 16 / (int) sizeof(T);
    const int lds = t2_lds_bytes<T>(NV, TW);
    // (per launch: the attribute belongs to the device the stream runs on, and a process may drive several)
    hipError_t e = hipFuncSetAttribute((const void *) spmm_team2p_kernel<T, NV, HAS_B1, TW, COMPACT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    const int tile = NV * 64 * VW;
    if (t.ngrid <= 0 || (t.ngrid & 7)) return hipErrorInvalidValue;
    const int ntile = (a.n + tile - 1) / tile;
    if (ntile > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL((spmm_team2p_kernel<T, NV, HAS_B1, TW, COMPACT>), dim3(t.ngrid, ntile), dim3(64 * TW), lds, s, a.n, t.torder, t.cptr, t.tinfo,
                       t.tpro, t.trec, t.tvoff, tval, t.trows, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC);
    return hipGetLastError();
}

template <typename T, typename ARGS>
static hipError_t dispatch_team2p(const Team2Args &t, const T *tval, const ARGS &a, bool nv2, hipStream_t s)
{
    const bool has_b1 = a.B1 != nullptr;
    if ((t.tw != 8 && t.tw != 16) || t.pw != 1 || t.chain <= 0 || t.trows == nullptr) return hipErrorInvalidValue;
    if (a.ldC * (long long) sizeof(T) >= (1ll << 32)) return hipErrorInvalidValue;
#define T2_GO(NV, B1, TW, CP) return launch_team2p<T, NV, B1, TW, CP>(t, tval, a, s)
    if (t.tw == 8 && !t.compact)
    {
        if (nv2) { if (has_b1) T2_GO(2, true, 8, false); else T2_GO(2, false, 8, false); }
        if (has_b1) T2_GO(1, true, 8, false); else T2_GO(1, false, 8, false);
    }
    if (!t.compact) return hipErrorInvalidValue;
    if (t.tw == 8)
    {
        if (nv2) { if (has_b1) T2_GO(2, true, 8, true); else T2_GO(2, false, 8, true); }
        if (has_b1) T2_GO(1, true, 8, true); else T2_GO(1, false, 8, true);
    }
    if (nv2) { if (has_b1) T2_GO(2, true, 16, true); else T2_GO(2, false, 16, true); }
    if (has_b1) T2_GO(1, true, 16, true); else T2_GO(1, false, 16, true);
#undef T2_GO
}

hipError_t spmm_rm_f64_team2p(const Team2Args &t, const SpmmArgs &a, hipStream_t s) { return dispatch_team2p<double>(t, t.tval, a, a.n > 128, s); }

hipError_t spmm_rm_f32_team2p(const Team2Args &t, const SpmmArgsF32 &a, hipStream_t s) { return dispatch_team2p<float>(t, t.tval32, a, a.n > 256, s); }

}  // namespace crp
