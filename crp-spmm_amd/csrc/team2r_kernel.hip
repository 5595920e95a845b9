// team2r_kernel.hip -- row-owner team SpMM for narrow operands (24 <= n <= 64 columns, fp64) on gfx950.  Variant 7: what variant 0
// takes at these widths for panels that are mostly holes (KKT systems).
//
// Same product as every kernel of this library (what mkl_sparse_d_mm computes at /root/reference/src/rowpara_spmm.c:388-408
// with alpha = 1, beta = 0).  Narrow operands are what the planner's 1 x P grids hand every GPU (n / P columns, all rows) -- and
// for KKT-shaped matrices 1 x P is the only grid whose exchange is not larger than the product (profiles/r03_kkt240_grid_proxy*):
// T(n = 256) / T(n = 32) on one GPU is the speed-up bound of 8 GPUs there.
//
// What bounds the row-panel kernels at these widths (profiles/r03_narrow_n32_counters.txt): one L2 request per PANEL entry and
// 128-byte line -- 64 requests in flight per CU at 360-480 cycles each -- and, for panels that are mostly holes (nlpkkt: 1.84 of 8
// rows per entry), a masked-row step that issues 16 FMAs for 7 useful ones.  This kernel
//   * fetches a B row slice once per TEAM (8 panels = 64 rows: 2.1 x fewer L2 requests on the nlpkkt stand-in, measured), by
//     LDS-DMA into a ring of three sets of 16 KiB, as team2_kernel.hip does, and
//   * lets the lane groups OWN rows: with G = 4 (n <= 32) lane group q of a wave accumulates rows q and q + 4 of the wave's panel
//     (G = 2, n <= 64: rows q, q + 2, q + 4, q + 6); a step gives every row its next nonzero of the round -- the LDS byte offset
//     of the B row slice in the ring set and the value -- and is one v_fmac_f64_dpp per column of the lane and row: no row
//     masks, no EXEC writes, no sum across lane groups.  A row without a further nonzero in the round gets 0.0 x (a slice of
//     zeros): an absent (row, column) pair still never meets a B entry (no 0 * Inf).  The values of a round's steps sit in the
//     lanes of a 16-lane DPP row (row_newbcast:s hands step s to all of them); offsets travel as uint16, four steps per read.
// Format: panel_format.h, Team2RHost.  One barrier per round = per 16 G union entries; two rounds in flight; persistent
// workgroups with one pipeline across their teams (below).
//
// MEASURED (round 3, same box as the kernels it replaces): nlpkkt stand-in kkt3d(96) n = 32 / 64: 0.500 / 1.017 ms against 0.586 /
// 1.110 of the narrow and row-panel kernels (0.388 / 0.839 in bench.py's harness with the launch rule of spmm_rm_f64_team2r); at
// nlpkkt240 size 6.58 / 14.0 against 8.85 / 18.2; pwtk stand-in (filled panels) 0.075
// against 0.062: not taken there.  The teams are team2's (primal and dual panels of a KKT system together: with teams of one kind
// the B rows both kinds share come from beyond L2 twice and the gain is gone at nlpkkt240 size), the union entries of a team go
// to its rounds in natural order (dealt like cards they balance the waves of a round and cost 15 %).  A round takes ~3000 cycles of
// which (s_memtime stamps, -DT2R_DBG) ~1400 are the issue of the next round's four DMAs per wave, ~900 the FMAs with their two
// dependent LDS reads per four steps, ~400 the barrier: latency-bound at four waves per SIMD.  The first version read a record per
// round through the scalar cache (2100 + 2300 of 5800 cycles per round were those misses): records now ride in the blocks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include "kernels.h"
#include "panel_format.h"

namespace crp {

namespace {
typedef double d2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
constexpr int T2R_NSET = 3, T2R_BLKB = 80 * TEAM2R_LCAP + 64, T2R_WBLK = T2R_NSET * T2R_BLKB;
constexpr int t2r_setb(int rd) { return 8192 * rd + 512; }                  // slots + the slice of zeros (512 bytes: a slice of the G = 2 instance)
constexpr int t2r_lds(int rd) { return T2R_NSET * t2r_setb(rd) + 8 * T2R_WBLK; }
static_assert(T2R_BLKB == 1024 && 5 * TEAM2R_LCAP + 4 <= 64 && 2 * t2r_lds(2) <= 160 * 1024 && 3 * t2r_lds(1) <= 160 * 1024,
              "block layout; two (full rounds) or three (half rounds) workgroups per CU");
#define T2R_GPTR(p) ((const __attribute__((address_space(1))) void *) (p))
#define T2R_LPTR(p) ((__attribute__((address_space(3))) void *) (p))
// One chunk = four steps of TWO rows' accumulators (lane group q's rows h0 G + q and (h0 + 1) G + q), all in one asm statement
// with its own waits: the compiler cannot tell the ring's DMA writes from these LDS reads and would put s_waitcnt vmcnt(0) --
// every DMA in flight -- before the first of them.  v[88:127] are scratch (named: a 128-bit asm operand has no way to name its
// halves): 2 x 2 offset words, 2 x 4 addresses, 8 slices.
#define T2R_SD(SEL) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" #SEL "\n\t"
#define T2R_FMA(ACC, V, BREG, K) "v_fmac_f64_dpp %[" #ACC "], %[" #V "], " BREG " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define T2R_CHUNK(OFF, K0, K1, K2, K3, AX0, AY0, AX1, AY1, V0, V1, OA0, OA1, RS)                                                  \
    asm volatile("ds_read2_b32 v[88:89], %[oa0] offset0:" #OFF " offset1:" #OFF "+1\n\t"                                            \
                 "ds_read2_b32 v[90:91], %[oa1] offset0:" #OFF " offset1:" #OFF "+1\n\t"                                            \
                 "s_waitcnt lgkmcnt(1)\n\t"                                                                                        \
                 "v_add_u32_sdwa v92, %[rs], v88" T2R_SD(WORD_0) "v_add_u32_sdwa v93, %[rs], v88" T2R_SD(WORD_1)                    \
                 "v_add_u32_sdwa v94, %[rs], v89" T2R_SD(WORD_0) "v_add_u32_sdwa v95, %[rs], v89" T2R_SD(WORD_1)                    \
                 "ds_read_b128 v[96:99], v92\n\t"                                                                                  \
                 "ds_read_b128 v[100:103], v93\n\t"                                                                                \
                 "ds_read_b128 v[104:107], v94\n\t"                                                                                \
                 "ds_read_b128 v[108:111], v95\n\t"                                                                                \
                 "s_waitcnt lgkmcnt(4)\n\t"                                                                                        \
                 "v_add_u32_sdwa v92, %[rs], v90" T2R_SD(WORD_0) "v_add_u32_sdwa v93, %[rs], v90" T2R_SD(WORD_1)                    \
                 "v_add_u32_sdwa v94, %[rs], v91" T2R_SD(WORD_0) "v_add_u32_sdwa v95, %[rs], v91" T2R_SD(WORD_1)                    \
                 "ds_read_b128 v[112:115], v92\n\t"                                                                                \
                 "ds_read_b128 v[116:119], v93\n\t"                                                                                \
                 "ds_read_b128 v[120:123], v94\n\t"                                                                                \
                 "ds_read_b128 v[124:127], v95\n\t"                                                                                \
                 "s_waitcnt lgkmcnt(7)\n\t" T2R_FMA(ax0, v0, "v[96:97]", K0) T2R_FMA(ay0, v0, "v[98:99]", K0)                      \
                 "s_waitcnt lgkmcnt(6)\n\t" T2R_FMA(ax0, v0, "v[100:101]", K1) T2R_FMA(ay0, v0, "v[102:103]", K1)                  \
                 "s_waitcnt lgkmcnt(5)\n\t" T2R_FMA(ax0, v0, "v[104:105]", K2) T2R_FMA(ay0, v0, "v[106:107]", K2)                  \
                 "s_waitcnt lgkmcnt(4)\n\t" T2R_FMA(ax0, v0, "v[108:109]", K3) T2R_FMA(ay0, v0, "v[110:111]", K3)                  \
                 "s_waitcnt lgkmcnt(3)\n\t" T2R_FMA(ax1, v1, "v[112:113]", K0) T2R_FMA(ay1, v1, "v[114:115]", K0)                  \
                 "s_waitcnt lgkmcnt(2)\n\t" T2R_FMA(ax1, v1, "v[116:117]", K1) T2R_FMA(ay1, v1, "v[118:119]", K1)                  \
                 "s_waitcnt lgkmcnt(1)\n\t" T2R_FMA(ax1, v1, "v[120:121]", K2) T2R_FMA(ay1, v1, "v[122:123]", K2)                  \
                 "s_waitcnt lgkmcnt(0)\n\t" T2R_FMA(ax1, v1, "v[124:125]", K3) T2R_FMA(ay1, v1, "v[126:127]", K3)                  \
                 : [ax0] "+v"(AX0), [ay0] "+v"(AY0), [ax1] "+v"(AX1), [ay1] "+v"(AY1)                                               \
                 : [v0] "v"(V0), [v1] "v"(V1), [oa0] "v"(OA0), [oa1] "v"(OA1), [rs] "v"(RS)                                         \
                 : "memory", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", \
                   "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", \
                   "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127")
// ... and its half: two steps (Lp is a multiple of 2; OFF counts dwords of the offset rows, as in T2R_CHUNK)
#define T2R_CHUNK2(OFF, K0, K1, AX0, AY0, AX1, AY1, V0, V1, OA0, OA1, RS)                                                          \
    asm volatile("ds_read_b32 v88, %[oa0] offset:4*" #OFF "\n\t"                                                                    \
                 "ds_read_b32 v90, %[oa1] offset:4*" #OFF "\n\t"                                                                    \
                 "s_waitcnt lgkmcnt(1)\n\t"                                                                                        \
                 "v_add_u32_sdwa v92, %[rs], v88" T2R_SD(WORD_0) "v_add_u32_sdwa v93, %[rs], v88" T2R_SD(WORD_1)                    \
                 "ds_read_b128 v[96:99], v92\n\t"                                                                                  \
                 "ds_read_b128 v[100:103], v93\n\t"                                                                                \
                 "s_waitcnt lgkmcnt(2)\n\t"                                                                                        \
                 "v_add_u32_sdwa v92, %[rs], v90" T2R_SD(WORD_0) "v_add_u32_sdwa v93, %[rs], v90" T2R_SD(WORD_1)                    \
                 "ds_read_b128 v[112:115], v92\n\t"                                                                                \
                 "ds_read_b128 v[116:119], v93\n\t"                                                                                \
                 "s_waitcnt lgkmcnt(3)\n\t" T2R_FMA(ax0, v0, "v[96:97]", K0) T2R_FMA(ay0, v0, "v[98:99]", K0)                      \
                 "s_waitcnt lgkmcnt(2)\n\t" T2R_FMA(ax0, v0, "v[100:101]", K1) T2R_FMA(ay0, v0, "v[102:103]", K1)                  \
                 "s_waitcnt lgkmcnt(1)\n\t" T2R_FMA(ax1, v1, "v[112:113]", K0) T2R_FMA(ay1, v1, "v[114:115]", K0)                  \
                 "s_waitcnt lgkmcnt(0)\n\t" T2R_FMA(ax1, v1, "v[116:117]", K1) T2R_FMA(ay1, v1, "v[118:119]", K1)                  \
                 : [ax0] "+v"(AX0), [ay0] "+v"(AY0), [ax1] "+v"(AX1), [ay1] "+v"(AY1)                                               \
                 : [v0] "v"(V0), [v1] "v"(V1), [oa0] "v"(OA0), [oa1] "v"(OA1), [rs] "v"(RS)                                         \
                 : "memory", "v88", "v90", "v92", "v93", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v112", "v113",  \
                   "v114", "v115", "v116", "v117", "v118", "v119")

}  // namespace

// Persistent workgroups: XCD x (= blockIdx & 7) owns entries [x cpx, (x + 1) cpx) of the launch grid (its teams, in processing
// order, none at the end); its workgroups take them round-robin and run ONE pipeline of rounds across all their teams: the first
// rounds of the next team are in flight while the last rounds of the current one are consumed, and a team's C rows are written one
// round later, after the next round's DMAs have been issued.  Nothing on the path of a round is a load that has to be waited for:
//   * what to fetch for round r + 2 of a team rides behind the block of round r (panel_format.h: the 64-byte header), which has
//     landed in LDS when round r is consumed;
//   * what a workgroup needs when it turns to a team (rounds, stream offset, the records of its rounds 0 and 1, its C rows) is one
//     128-byte row of the entry table, loaded into scalar registers ONE TEAM AHEAD.
// (The first version read a record per round through the scalar cache: measured with s_memtime, 2100 of a round's 5800 cycles were
//  that load's miss, another 2300 the load of the round's step count, against ~300 for the barrier and ~400 for the FMAs.)
struct T2RRec { uint32_t w[10]; };                                          // Lp, block offset, up to 8 columns
struct T2RTeam { int nr; long long vb; T2RRec r0, r1; };

template <int G, bool HAS_B1, int RD>
__global__ __launch_bounds__(512, 4) void spmm_team2r_kernel(const int ngrid, const uint32_t *__restrict__ tent, const double *__restrict__ tval,
                                                             const int n, const double *__restrict__ B0, const int64_t ldB0,
                                                             const double *__restrict__ B1, const int64_t ldB1, double *__restrict__ C,
                                                             const int64_t ldC, const int stagger, const int chain_k, unsigned long long *dbg)
{
    constexpr int LPG = 64 / G, SLOTB = 1024 / G, PERW = RD * G, NH = 8 / G;
    constexpr int T2R_SETB = t2r_setb(RD), ZERO = 8192 * RD;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int cpx = ngrid >> 3;                                             // entries per XCD run (ngrid is a multiple of 8)
    const int wx0 = (int) (gridDim.x >> 3);                                 // workgroups per XCD
    // a workgroup's chain of entries: every wx0-th of its XCD's run (chain_k = 0), or chain_k consecutive ones
    const int wx = chain_k > 0 ? 1 : wx0;
    const int e_first = (int) (blockIdx.x & 7) * cpx + (chain_k > 0 ? (int) (blockIdx.x >> 3) * chain_k : (int) (blockIdx.x >> 3));
    const int e_end = chain_k > 0 ? min((int) ((blockIdx.x & 7) + 1) * cpx, e_first + chain_k) : (int) ((blockIdx.x & 7) + 1) * cpx;
    const int q = lane / LPG, l = lane % LPG, l16 = lane & 15;
    const int lo = (2 * l + 1 < n) ? l * 16 : 0;                           // lanes past n fetch the row's first bytes: valid, never stored
    // LDS byte addresses for the asm reads: the dynamic LDS's own offset (0 while this kernel has no static LDS) + ...
    const uint32_t lds0 = (uint32_t) (uintptr_t) T2R_LPTR(lds);
    const uint32_t blk0 = (uint32_t) (T2R_NSET * T2R_SETB + wave * T2R_WBLK);            // relative to lds (DMA destinations), + lds0 for reads
    // the slices of zeros (one per ring set: offsets are relative to the set; 512 bytes: a slice of the G = 2 instance)
    if (threadIdx.x < T2R_NSET * 32)
        *reinterpret_cast<d2 *>(lds + (threadIdx.x >> 5) * T2R_SETB + ZERO + (threadIdx.x & 31) * 16) = d2{0.0, 0.0};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // written before this wave reaches the first round's barrier
    double a[NH][2], pa[NH][2];
#pragma unroll
    for (int h = 0; h < NH; h++) a[h][0] = a[h][1] = pa[h][0] = pa[h][1] = 0.0;

    auto load_team = [&](const int e) {                                     // scalar loads through a kernel argument: s_load_dwordx8 / x16
        T2RTeam t;
        t.nr = 0;
        t.vb = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) t.r0.w[i] = t.r1.w[i] = 0;
        if (e < e_end)
        {
            const uint32_t *T = tent + ((size_t) e * 8 + (size_t) wave) * 32;
            t.nr = (int) T[0];
            t.vb = ((long long) T[2] | ((long long) T[3] << 32)) * 16;
#pragma unroll
            for (int i = 0; i < 10; i++)
            {
                t.r0.w[i] = T[4 + i];
                t.r1.w[i] = T[14 + i];
            }
        }
        return t;
    };
    auto fetch = [&](const T2RRec &R, const int set, const long long vb) {  // the four DMAs of a round
        const int Lp = (int) R.w[0];
#pragma unroll
        for (int j = 0; j < RD; j++)
        {
            // (readfirstlane: or the optimiser turns the selection into a per-lane load of R.w[2 + j G + q] from a stack copy)
            int col = __builtin_amdgcn_readfirstlane((int) R.w[2 + j * G]);
            const int c1 = __builtin_amdgcn_readfirstlane((int) R.w[3 + j * G]);
            if (q == 1) col = c1;
            if constexpr (G == 4)
            {
                const int c2 = __builtin_amdgcn_readfirstlane((int) R.w[4 + j * G]), c3 = __builtin_amdgcn_readfirstlane((int) R.w[5 + j * G]);
                if (q == 2) col = c2;
                if (q == 3) col = c3;
            }
            const char *src = (!HAS_B1 || col >= 0) ? reinterpret_cast<const char *>(B0 + (int64_t) col * ldB0) : reinterpret_cast<const char *>(B1 + (int64_t) (~col) * ldB1);
            __builtin_amdgcn_global_load_lds(T2R_GPTR(src + lo), T2R_LPTR(lds + set * T2R_SETB + (wave * PERW + j * G) * SLOTB), 16, 0, 0);
        }
        const char *bsrc = reinterpret_cast<const char *>(tval) + vb + (size_t) R.w[1] * 16;
        char *bdst = lds + blk0 + set * T2R_BLKB;
        // the block as it lies in the stream: values (64 Lp bytes), offsets (16 Lp), header (64) = 5 Lp + 4 lanes of ONE instruction
        // (Lp <= 12: 64 lanes; every DMA instruction costs a wave ~170 cycles of issue here)
        if (lane < 5 * Lp + 4) __builtin_amdgcn_global_load_lds(T2R_GPTR(bsrc + lane * 16), T2R_LPTR(bdst), 16, 0, 0);
    };
    auto header = [&](const int set, const int Lp) {                        // the record behind the block in ring set `set`
        const uint32_t ad = lds0 + blk0 + (uint32_t) (set * T2R_BLKB + 80 * Lp);
        u4 h0, h1;
        u2 h2;
        asm volatile("ds_read_b128 %[h0], %[ad]\n\tds_read_b128 %[h1], %[ad] offset:16\n\tds_read_b64 %[h2], %[ad] offset:32\n\ts_waitcnt lgkmcnt(0)"
                     : [h0] "=&v"(h0), [h1] "=&v"(h1), [h2] "=&v"(h2)
                     : [ad] "v"(ad)
                     : "memory");
        T2RRec R;
        R.w[0] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h0.x);
        R.w[1] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h0.y);
        R.w[2] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h0.z);
        R.w[3] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h0.w);
        R.w[4] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h1.x);
        R.w[5] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h1.y);
        R.w[6] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h1.z);
        R.w[7] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h1.w);
        R.w[8] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h2.x);
        R.w[9] = (uint32_t) __builtin_amdgcn_readfirstlane((int) h2.y);
        return R;
    };
    auto flush = [&](const int e_) {                                       // pa -> the C rows of the team at entry e (this wave's panel)
        const int e = __builtin_amdgcn_readfirstlane(e_);
        const uint32_t *T = tent + ((size_t) e * 8 + (size_t) wave) * 32 + 24;   // (a line the team's load_team has brought in)
        int rm[8];
#pragma unroll
        for (int i = 0; i < 8; i++) rm[i] = __builtin_amdgcn_readfirstlane((int) T[i]);
        if (2 * l + 1 < n)
        {
#pragma unroll
            for (int h = 0; h < NH; h++)
            {
                int crow_i = rm[h * G];
#pragma unroll
                for (int qq = 1; qq < G; qq++)
                    if (q == qq) crow_i = rm[h * G + qq];
                if (crow_i >= 0)
                {
                    d2 t2 = {pa[h][0], pa[h][1]};
                    __builtin_nontemporal_store(t2, reinterpret_cast<d2 *>(C + (int64_t) crow_i * ldC + 2 * l));
                }
            }
        }
    };

    // The round to issue next: entry ie, round ir of its team (cnr rounds, stream at byte cvb of tval).  tm = the table row loaded
    // last: the team at ie until its rounds 0 and 1 have been issued, then (reloaded at once) the team at ie + wx, one team ahead.
    int ie = e_first, ir = 0;
    T2RTeam tm = load_team(ie);
    int cnr = tm.nr;
    long long cvb = tm.vb;
    bool ahead_loaded = false;                                              // tm holds the team at ie + wx
    // rounds issued and not yet consumed, oldest first: steps | last round of its team << 8, entry
    int f0 = 0, f1 = 0, f2 = 0, g0 = 0, g1 = 0, g2 = 0, ahead = 0;
    int si = 0, sc = 0;                                                     // ring set of the next round to issue / to consume
#ifdef T2R_DBG
    long long th = 0, tfe = 0;
#endif
    auto issue_next = [&]() {
        T2RRec R;
#ifdef T2R_DBG
        const long long q0 = clock64();
#endif
        if (ir == 0) R = tm.r0;
        else if (ir == 1) R = tm.r1;
        else R = header(sc, f0 & 0xFF);                                     // rounds ir - 2 (being consumed now) and ir - 1 are the ones in flight
#ifdef T2R_DBG
        const long long q1 = clock64();
#endif
        fetch(R, si, cvb);
#ifdef T2R_DBG
        const long long q2 = clock64();
        th += q1 - q0;
        tfe += q2 - q1;
#endif
        const int desc = (int) R.w[0] | ((ir == cnr - 1) ? 256 : 0);
        // (selects, not branches: the optimiser turns a three-way branch into an indexed array on the stack)
        f0 = ahead == 0 ? desc : f0; g0 = ahead == 0 ? ie : g0;
        f1 = ahead == 1 ? desc : f1; g1 = ahead == 1 ? ie : g1;
        f2 = ahead == 2 ? desc : f2; g2 = ahead == 2 ? ie : g2;
        ahead++;
        si = si == T2R_NSET - 1 ? 0 : si + 1;
        ir++;
        if (!ahead_loaded && (ir >= 2 || ir == cnr))                        // the records of rounds 0 and 1 are used up: the next team's row
        {
            tm = load_team(ie + wx);
            ahead_loaded = true;
        }
        if (ir == cnr)
        {
            ie += wx;
            ir = 0;
            cnr = tm.nr;
            cvb = tm.vb;
            ahead_loaded = false;
        }
    };
    for (int d = 0; d < 2; d++)
        if (cnr > 0) issue_next();
    int pe = 0;
    bool pending = false;
    const bool early = stagger == 0 || (wave & 1) == 0;
#ifdef T2R_DBG
    long long tw = 0, tb = 0, ti = 0, tf = 0, tc = 0, nrd = 0;
#define T2R_CLK(x) const long long x = clock64()
#else
#define T2R_CLK(x)
#endif
    while (ahead > 0)
    {
        T2R_CLK(c0);
        // this wave's DMAs of the round to consume (RD rows + the block) have landed (those of the round behind it may still fly; stores of the
        // last flush count too, which can only make this wait longer: loads complete in order) ...
        if (ahead > 1)
        {
            if constexpr (RD == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and everybody's; every wave is also done reading the round before, whose set the next issue takes
        T2R_CLK(c1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        T2R_CLK(c2);
        // Half of the waves issue the next round's DMAs now, the other half after their FMAs: all 32 DMA instructions of a round
        // at once queue up behind each other in the CU's one address unit (measured: ~1400 of a round's ~3000 cycles in this
        // phase, the unit itself 36 % busy), while nothing else of the workgroup runs.  CRPSPMM_T2R_STAGGER=0: all at once.
        if (early && cnr > 0) issue_next();
        T2R_CLK(c3);
        if (pending)
        {
            flush(pe);
            pending = false;
        }
        T2R_CLK(c4);
        const int Lp = f0 & 0xFF;
        const uint32_t rs = lds0 + (uint32_t) (sc * T2R_SETB + l * 16);
        const uint32_t vs = lds0 + blk0 + (uint32_t) (sc * T2R_BLKB);
        const uint32_t os = vs + 64u * (uint32_t) Lp;
        if (Lp > 0)
        {
#pragma unroll
            for (int h0 = 0; h0 < NH; h0 += 2)
            {
                const uint32_t row0 = (uint32_t) (h0 * G + q), row1 = (uint32_t) ((h0 + 1) * G + q);
                double v0, v1;
                asm volatile("ds_read_b64 %[v0], %[a0]\n\tds_read_b64 %[v1], %[a1]\n\ts_waitcnt lgkmcnt(0)"
                             : [v0] "=&v"(v0), [v1] "=&v"(v1)
                             : [a0] "v"(vs + ((row0 * (uint32_t) Lp + (uint32_t) l16) << 3)), [a1] "v"(vs + ((row1 * (uint32_t) Lp + (uint32_t) l16) << 3))
                             : "memory");
                const uint32_t oa0 = os + ((row0 * (uint32_t) Lp) << 1), oa1 = os + ((row1 * (uint32_t) Lp) << 1);
                // steps four at a time, then a pair (Lp is even): OFF = dword of the offset row
                if (Lp >= 4) T2R_CHUNK(0, 0, 1, 2, 3, a[h0][0], a[h0][1], a[h0 + 1][0], a[h0 + 1][1], v0, v1, oa0, oa1, rs);
                if (Lp >= 8) T2R_CHUNK(2, 4, 5, 6, 7, a[h0][0], a[h0][1], a[h0 + 1][0], a[h0 + 1][1], v0, v1, oa0, oa1, rs);
                if (Lp >= 12) T2R_CHUNK(4, 8, 9, 10, 11, a[h0][0], a[h0][1], a[h0 + 1][0], a[h0 + 1][1], v0, v1, oa0, oa1, rs);
                if (Lp == 2) T2R_CHUNK2(0, 0, 1, a[h0][0], a[h0][1], a[h0 + 1][0], a[h0 + 1][1], v0, v1, oa0, oa1, rs);
                else if (Lp == 6) T2R_CHUNK2(2, 4, 5, a[h0][0], a[h0][1], a[h0 + 1][0], a[h0 + 1][1], v0, v1, oa0, oa1, rs);
                else if (Lp == 10) T2R_CHUNK2(4, 8, 9, a[h0][0], a[h0][1], a[h0 + 1][0], a[h0 + 1][1], v0, v1, oa0, oa1, rs);
            }
        }
        if (!early && cnr > 0) issue_next();
        if (f0 & 256)                                                      // the team is complete: its rows go out after the next round's issue
        {
            pe = g0;
            pending = true;
#pragma unroll
            for (int h = 0; h < NH; h++)
            {
                pa[h][0] = a[h][0];
                pa[h][1] = a[h][1];
                a[h][0] = a[h][1] = 0.0;
            }
        }
        f0 = f1; g0 = g1;
        f1 = f2; g1 = g2;
        ahead--;
        sc = sc == T2R_NSET - 1 ? 0 : sc + 1;
#ifdef T2R_DBG
        T2R_CLK(c5);
        tw += c1 - c0; tb += c2 - c1; ti += c3 - c2; tf += c4 - c3; tc += c5 - c4; nrd++;
#endif
    }
    if (pending) flush(pe);
#ifdef T2R_DBG
    if (dbg != nullptr && lane == 0)
    {
        atomicAdd(dbg + 0, (unsigned long long) tw); atomicAdd(dbg + 1, (unsigned long long) tb); atomicAdd(dbg + 2, (unsigned long long) ti);
        atomicAdd(dbg + 3, (unsigned long long) tf); atomicAdd(dbg + 4, (unsigned long long) tc); atomicAdd(dbg + 5, (unsigned long long) nrd);
        atomicAdd(dbg + 6, (unsigned long long) th); atomicAdd(dbg + 7, (unsigned long long) tfe);
    }
#endif
}

// the C rows of every (entry, wave)'s panel into the entry table (words 24 .. 31; -1 = no such row)
__global__ void team2r_fill_rows_kernel(const int nent8, uint32_t *tent, const int nrow, const int *__restrict__ rowmap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nent8) return;
    uint32_t *T = tent + (size_t) i * 32;
    const int nr = (int) T[0], panel = (int) T[1];
    for (int k = 0; k < 8; k++)
    {
        const int row = panel * 8 + k;
        T[24 + k] = (nr > 0 && panel >= 0 && row < nrow) ? (uint32_t) (rowmap ? rowmap[row] : row) : 0xFFFFFFFFu;
    }
}

hipError_t team2r_fill_rows(const Team2NArgs &t, const SpmmArgs &a, hipStream_t s)
{
    const int nent8 = t.ngrid * 8;
    if (nent8 <= 0) return hipSuccess;
    hipLaunchKernelGGL(team2r_fill_rows_kernel, dim3((nent8 + 255) / 256), dim3(256), 0, s, nent8, t.tent, a.nrow, a.rowmap);
    return hipGetLastError();
}

// 24 <= n <= 128 / G (even), 16-byte aligned operands
bool spmm_team2r_applicable(const Team2NArgs &t, const SpmmArgs &a)
{
    return a.n >= 24 && a.n <= 128 / t.G && (a.n % 2 == 0) && (a.ldB0 % 2 == 0) && (a.ldC % 2 == 0) && (a.B1 == nullptr || a.ldB1 % 2 == 0) &&
           (((uintptr_t) a.B0 | (uintptr_t) a.B1 | (uintptr_t) a.C) % 16 == 0);
}

hipError_t spmm_rm_f64_team2r(const Team2NArgs &t, const SpmmArgs &a, hipStream_t s)
{
    const bool has_b1 = a.B1 != nullptr;
    // Persistent workgroups: two fit a CU (the LDS of one is 73.5 KiB), but enough are launched that each takes a chain of about
    // seven teams (and never fewer than are resident): the resident workgroups then work on neighbouring teams -- chains that run
    // for the whole launch drift apart, and with them the B rows their teams share (nlpkkt240 size, n = 32: 7.64 ms with 512
    // workgroups, 7.36 with 2048, 6.90 with 16384, 6.63 with 65536 = chains of 6.7 teams, 7.06 with one team per workgroup).
    // Eight XCD runs.  (Per launch: the device behind the stream may differ from call to call.)
    int ncu = 256;
    {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    const int run = t.ngrid / 8;                                            // entries of an XCD's run
    const int per_xcd = std::max(1, std::min(run, std::max(2 * ncu / 8, (run + 6) / 7)));
    dim3 grid(per_xcd * 8);
    // a workgroup's chain = every per_xcd-th entry of the run (consecutive entries instead: no better, 6.59 against 6.48 ms)
    const int chain_k = 0;
    unsigned long long *dbg = nullptr;
    const int stagger = 1;
#define CRP_T2R_GO(G_, HB1_, RD_)                                                                                                                   \
    do                                                                                                                                              \
    {                                                                                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&spmm_team2r_kernel<G_, HB1_, RD_>), hipFuncAttributeMaxDynamicSharedMemorySize, t2r_lds(RD_)); \
        if (e != hipSuccess) return e;                                                                                                              \
        hipLaunchKernelGGL((spmm_team2r_kernel<G_, HB1_, RD_>), grid, dim3(512), t2r_lds(RD_), s, t.ngrid, t.tent, t.tval, a.n, a.B0, a.ldB0, a.B1, a.ldB1, a.C, a.ldC, stagger, chain_k, dbg); \
    } while (0)
#define CRP_T2R_PICK(RD_)                                                                                       \
    do                                                                                                          \
    {                                                                                                           \
        if (t.G == 4) { if (has_b1) CRP_T2R_GO(4, true, RD_); else CRP_T2R_GO(4, false, RD_); }                  \
        else { if (has_b1) CRP_T2R_GO(2, true, RD_); else CRP_T2R_GO(2, false, RD_); }                           \
    } while (0)
    CRP_T2R_PICK(TEAM2R_ROWDMA);
#undef CRP_T2R_PICK
#undef CRP_T2R_GO
    return hipGetLastError();
}

}  // namespace crp
