// fetch_replay.hip -- a FETCH-ONLY replay of a B-row schedule on the pipeline of the team kernel (csrc/team2_kernel.hip): persistent
// 512-thread workgroups take units (teams, or strips of the sliding-window scheme of DESIGN.md section 8) from one queue per XCD; a unit is
// a list of rounds of S B-row slices of 2 KiB; wave w fetches slot w of round r + 3 by LDS-DMA while the workgroup reads round r from the
// ring (every wave reads `reads` slots), one barrier per round; C rows are written at the unit's end (teams) or one per round (strips).
// No FMAs, no records: what the schedule alone costs on the memory pipeline.  Planning tool for the next kernel, not part of the library.
//   fetch_replay PLAN.bin [launches]        (plans: tools/probe/make_plans.py)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Plan
{
    int nunit, qlen, S, nrowB, nrowC, wgs, wgs_per_cu, write_at_end, alanes, reads, maxnr, store_policy;
    const int *nr; const long long *off; const int *rows; const int *queue; const int *cbase; const int *cn;
};

constexpr int RINGR = 4, D = 3;
// cache policy of the DMAs (-DVPOL=.. value blocks, -DBPOL=.. row slices): 0 default, 1 sc0, 2 nt, 16 sc1 (and sums)
#ifndef VPOL
#define VPOL 0
#endif
#ifndef BPOL
#define BPOL 0
#endif
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef const __attribute__((address_space(1))) void *glb_ptr;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// The round loop keeps its memory pipeline out of the compiler's sight where the compiler would serialise it: the ring is read with
// inline ds_read (a read the compiler sees would get a vmcnt(0) in front: it may alias the DMAs in flight), the barrier is the bare
// instruction, the waits are written by hand, and the unit's row numbers are copied to LDS before its pipeline starts (a vector load of
// them inside the loop would wait for every DMA issued before it: vmcnt counts in order).
constexpr int VBLK = 128;                           // bytes of LDS behind one value DMA (at most 8 lanes of 16 bytes)
template <int S, int ROWB>                          // slots of a round; bytes of a row slice: 2048 (n = 256 fp64) or 1024 (n = 128)
__global__ __launch_bounds__(512) void replay_kernel(Plan p, const char *__restrict__ B, char *__restrict__ C, const char *__restrict__ avals, int *ctr, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) char ring[];             // RINGR * S slots of 2 KiB, 8 x 256 bytes for the value blocks, the unit's rows
    __shared__ int s_unit;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7;
    char *vblk = ring + RINGR * S * ROWB;
    int *urows = (int *) (vblk + RINGR * 8 * 3 * VBLK);
    const unsigned ring_lds = (unsigned) (uintptr_t) (lds_ptr) ring, urows_lds = (unsigned) (uintptr_t) (lds_ptr) urows;
    unsigned acc = 0;
    for (;;)
    {
        __syncthreads();
        if (threadIdx.x == 0)
        {
            const int pos = atomicAdd(&ctr[xcd], 1);
            s_unit = pos < p.qlen ? p.queue[(long long) xcd * p.qlen + pos] : -1;
        }
        __syncthreads();
        const int u = __builtin_amdgcn_readfirstlane(s_unit);
        if (u < 0) break;
        const int nr = __builtin_amdgcn_readfirstlane(p.nr[u]);
        const long long off = p.off[u];
        const int cn = __builtin_amdgcn_readfirstlane(p.cn[u]), cbase = __builtin_amdgcn_readfirstlane(p.cbase[u]);
        for (int i = threadIdx.x; i < nr * S; i += 512) urows[i] = p.rows[off * S + i];
        __syncthreads();                                                    // (waits for everything: the pipeline is empty here)
        auto issue = [&](int r) {
            if (wave < S)
            {
                int row;
                asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(row) : "v"(urows_lds + (unsigned) (r * S + wave) * 4u));
                row = __builtin_amdgcn_readfirstlane(row);
                const char *src = B + (long long) row * ROWB + lane * 16;
                char *dst = ring + ((r & (RINGR - 1)) * S + wave) * ROWB;
                __builtin_amdgcn_global_load_lds((glb_ptr) src, (lds_ptr) dst, 16, 0, BPOL);
                if (ROWB == 2048) __builtin_amdgcn_global_load_lds((glb_ptr) (src + 1024), (lds_ptr) (dst + 1024), 16, 0, BPOL);
            }
            // the wave's value block of the round (streamed once); waves without a slot fetch three times as much of it, so that every
            // wave counts three DMAs per round.  Every DMA has its own place in LDS (the compiler orders DMAs to one address itself).
            const char *vsrc = avals + ((off + r) * 8 + wave) * 256 + lane * 16;
            char *vdst = vblk + (((r & (RINGR - 1)) * 8 + wave) * 3) * VBLK;
            if (lane < p.alanes)
            {
                __builtin_amdgcn_global_load_lds((glb_ptr) vsrc, (lds_ptr) vdst, 16, 0, VPOL);
                if (wave >= S)
                {
                    __builtin_amdgcn_global_load_lds((glb_ptr) vsrc, (lds_ptr) (vdst + VBLK), 16, 0, 0);
                    if (ROWB == 2048) __builtin_amdgcn_global_load_lds((glb_ptr) vsrc, (lds_ptr) (vdst + 2 * VBLK), 16, 0, 0);
                }
            }
        };
        // (the first D rounds are issued from the same place in the code as all others: in a separate prologue the compiler put a
        //  vmcnt(0) between the DMAs of a round)
        for (int r = -D; r < nr; r++)
        {
            if (r >= 0)
            {
                // my fetches of round r have landed when at most those of the younger rounds in flight are outstanding (3 per round)
                const int younger = min(D - 1, nr - 1 - r);
                if (ROWB == 2048)
                {
                    if (younger >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else if (younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                else                                                        // two DMAs per round: the row slice and the value block
                {
                    if (younger >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_barrier" ::: "memory");
            }
            if (r + D < nr) issue(r + D);
            if (r < 0) continue;
            for (int k = 0; k < p.reads; k++)
            {
                const int slot = (wave + k) % S;
                v4u a, b = {0u, 0u, 0u, 0u};
                if (ROWB == 2048)
                    asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)"
                                 : "=&v"(a), "=&v"(b) : "v"(ring_lds + (unsigned) (((r & (RINGR - 1)) * S + slot) * ROWB + lane * 32)));
                else
                    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"(ring_lds + (unsigned) (((r & (RINGR - 1)) * S + slot) * ROWB + lane * 16)));
                acc ^= a.x ^ a.w ^ b.y ^ b.z;
            }
            // strips: one finished row of C per round, written by a wave that fetches no B row (its vmcnt is nobody's business)
            if (!p.write_at_end && r < cn && wave == (S < 8 ? S + (r % (8 - S > 0 ? 8 - S : 1)) : (r & 7)))
            {
                v4u v = {acc, 0u, 0u, 0u};
                v4u *c = (v4u *) (C + (long long) (cbase + r) * ROWB) + lane * (ROWB / 1024);
                __builtin_nontemporal_store(v, c);
                if (ROWB == 2048) __builtin_nontemporal_store(v, c + 1);
            }
        }
        if (p.write_at_end)
            for (int i = wave; i < cn; i += 8)
            {
                v4u v = {acc, 0u, 0u, 0u};
                v4u *c = (v4u *) (C + (long long) (cbase + i) * ROWB) + lane * (ROWB / 1024);
                // cache policy of the C stores (FETCH_REPLAY_STORE): 0 nt (what the kernels use), 1 plain, 2 sc0 sc1, 3 sc0 sc1 nt, 4 sc1, 5 sc1 nt
                switch (p.store_policy)
                {
                case 1: asm volatile("global_store_dwordx4 %0, %1, off\n global_store_dwordx4 %0, %1, off offset:16" :: "v"(c), "v"(v) : "memory"); break;
                case 2: asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n global_store_dwordx4 %0, %1, off offset:16 sc0 sc1" :: "v"(c), "v"(v) : "memory"); break;
                case 3: asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n global_store_dwordx4 %0, %1, off offset:16 sc0 sc1 nt" :: "v"(c), "v"(v) : "memory"); break;
                case 4: asm volatile("global_store_dwordx4 %0, %1, off sc1\n global_store_dwordx4 %0, %1, off offset:16 sc1" :: "v"(c), "v"(v) : "memory"); break;
                case 5: asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n global_store_dwordx4 %0, %1, off offset:16 sc1 nt" :: "v"(c), "v"(v) : "memory"); break;
                default:
                    __builtin_nontemporal_store(v, c);
                    if (ROWB == 2048) __builtin_nontemporal_store(v, c + 1);
                }
            }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// The same replay with the B rows fetched into REGISTERS (global_load_dwordx4, three rounds ahead) and written to the ring by the wave
// when they have landed -- the path the team kernel left for LDS-DMA.  Teams only (S = 8); the waits are the compiler's.
__global__ __launch_bounds__(512) void replay_vgpr_kernel(Plan p, const char *__restrict__ B, char *__restrict__ C, const char *__restrict__ avals, int *ctr, unsigned *sink)
{
    constexpr int S = 8, ROWB = 2048;
    extern __shared__ __attribute__((aligned(16))) char ring[];             // 2 rounds x 8 slots of 2 KiB, the unit's rows
    __shared__ int s_unit;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7;
    int *urows = (int *) (ring + 2 * S * ROWB);
    const unsigned ring_lds = (unsigned) (uintptr_t) (lds_ptr) ring, urows_lds = (unsigned) (uintptr_t) (lds_ptr) urows;
    unsigned acc = 0;
    for (;;)
    {
        __syncthreads();
        if (threadIdx.x == 0)
        {
            const int pos = atomicAdd(&ctr[xcd], 1);
            s_unit = pos < p.qlen ? p.queue[(long long) xcd * p.qlen + pos] : -1;
        }
        __syncthreads();
        const int u = __builtin_amdgcn_readfirstlane(s_unit);
        if (u < 0) break;
        const int nr = __builtin_amdgcn_readfirstlane(p.nr[u]);
        const long long off = p.off[u];
        const int cn = __builtin_amdgcn_readfirstlane(p.cn[u]), cbase = __builtin_amdgcn_readfirstlane(p.cbase[u]);
        for (int i = threadIdx.x; i < nr * S; i += 512) urows[i] = p.rows[off * S + i];
        __syncthreads();
        v4u b0x, b0y, b1x, b1y, b2x, b2y, vv;
        auto load = [&](int r, v4u &x, v4u &y) {
            int row;
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(row) : "v"(urows_lds + (unsigned) (r * S + wave) * 4u));
            row = __builtin_amdgcn_readfirstlane(row);
            const char *src = B + (long long) row * ROWB + lane * 16;
            // (inline: the compiler would wait for every load in flight at the loop's head; the wait is written by hand below)
            // + the wave's value block of the round (every lane asks for its 16 bytes of the wave's 1 KiB window: alanes of them are new)
            const char *vsrc = avals + ((off + r) * 8 + wave) * 256 + (lane < p.alanes ? lane : 0) * 16;
            asm volatile("global_load_dwordx4 %0, %3, off\n global_load_dwordx4 %1, %3, off offset:1024\n global_load_dwordx4 %2, %4, off"
                         : "=&v"(x), "=&v"(y), "=&v"(vv) : "v"(src), "v"(vsrc) : "memory");
        };
        auto round = [&](int r, v4u &x, v4u &y) {
            // my slice of round r goes to the ring, the workgroup meets, the slice of round r + 3 is asked for, round r is read
            const int younger = min(2, nr - 1 - r);                        // three loads per round in flight behind this round's
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %2 offset:1024\n s_waitcnt lgkmcnt(0)\n s_barrier"
                         :: "v"(ring_lds + (unsigned) (((r & 1) * S + wave) * ROWB + lane * 16)), "v"(x), "v"(y) : "memory");
            if (r + 3 < nr) load(r + 3, x, y);
            for (int k = 0; k < p.reads; k++)
            {
                const int slot = (wave + k) % S;
                v4u a, b;
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(a), "=&v"(b) : "v"(ring_lds + (unsigned) (((r & 1) * S + slot) * ROWB + lane * 32)));
                acc ^= a.x ^ a.w ^ b.y ^ b.z;
            }
        };
        load(0, b0x, b0y);
        if (nr > 1) load(1, b1x, b1y);
        if (nr > 2) load(2, b2x, b2y);
        for (int r = 0; r < nr; r += 3)
        {
            round(r, b0x, b0y);
            if (r + 1 < nr) round(r + 1, b1x, b1y);
            if (r + 2 < nr) round(r + 2, b2x, b2y);
        }
        for (int i = wave; i < cn; i += 8)
        {
            v4u v = {acc, 0u, 0u, 0u};
            v4u *c = (v4u *) (C + (long long) (cbase + i) * ROWB) + lane * 2;
            __builtin_nontemporal_store(v, c);
            __builtin_nontemporal_store(v, c + 1);
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: fetch_replay PLAN.bin [launches]\n"); return 2; }
    const int launches = argc > 2 ? atoi(argv[2]) : 50;
    const bool vgpr = argc > 3 && !strcmp(argv[3], "vgpr");
    const int ROWB = argc > 3 && !strcmp(argv[3], "n128") ? 1024 : 2048;      // "n128": row slices of 1 KiB (S = 8 plans)
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int hdr[13];
    if (fread(hdr, sizeof(int), 13, f) != 13 || hdr[0] != 0x46524550) { fprintf(stderr, "bad plan file\n"); return 1; }
    const int nq = hdr[12];                              // alternative launch grids (queues) over the same units, replayed one after the other
    if (nq < 1 || nq > 4096) { fprintf(stderr, "bad number of queues\n"); return 1; }
    Plan p;
    p.nunit = hdr[1]; p.qlen = hdr[2]; p.S = hdr[3]; p.nrowB = hdr[4]; p.nrowC = hdr[5]; p.wgs_per_cu = hdr[6]; p.write_at_end = hdr[7]; p.alanes = hdr[8]; p.reads = hdr[9];
    const long long nrounds = ((long long) hdr[11] << 31) | (unsigned) hdr[10];
    std::vector<int> nr(p.nunit), cbase(p.nunit), cn(p.nunit), queue((size_t) nq * 8 * p.qlen), rows((size_t) nrounds * p.S);
    std::vector<long long> off(p.nunit);
    bool ok = fread(nr.data(), 4, nr.size(), f) == nr.size() && fread(off.data(), 8, off.size(), f) == off.size() && fread(cbase.data(), 4, cbase.size(), f) == cbase.size()
              && fread(cn.data(), 4, cn.size(), f) == cn.size() && fread(queue.data(), 4, queue.size(), f) == queue.size() && fread(rows.data(), 4, rows.size(), f) == rows.size();
    fclose(f);
    if (!ok) { fprintf(stderr, "short plan file\n"); return 1; }
    // everything the kernel indexes by, checked here: it has no bounds tests of its own
    if (p.S != 5 && p.S != 8) { fprintf(stderr, "S must be 5 or 8\n"); return 1; }
    if (p.alanes < 1 || p.alanes > VBLK / 16 || p.reads < 0 || p.reads > 8 || p.wgs_per_cu < 1 || p.wgs_per_cu > 4) { fprintf(stderr, "bad parameters\n"); return 1; }
    for (int u = 0; u < p.nunit; u++)
        if (nr[u] < 1 || nr[u] > 4096 || off[u] < 0 || off[u] + nr[u] > nrounds || cbase[u] < 0 || cn[u] < 0 || (long long) cbase[u] + cn[u] > p.nrowC || (!p.write_at_end && cn[u] > nr[u])) { fprintf(stderr, "bad unit %d\n", u); return 1; }
    p.maxnr = *std::max_element(nr.begin(), nr.end());
    for (int r : rows) if (r < 0 || r >= p.nrowB) { fprintf(stderr, "bad row index\n"); return 1; }
    for (int q : queue) if (q < -1 || q >= p.nunit) { fprintf(stderr, "bad queue entry\n"); return 1; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    p.store_policy = getenv("FETCH_REPLAY_STORE") ? atoi(getenv("FETCH_REPLAY_STORE")) : 0;       // (2 KiB row slices, teams only)
    if (getenv("FETCH_REPLAY_NOWRITE")) { for (int &c : cn) c = 0; }                        // no C rows written: what the stores cost
    if (const char *e = getenv("FETCH_REPLAY_WGS")) p.wgs_per_cu = std::max(1, std::min(4, atoi(e)));       // workgroups per CU, over the plan's
    p.wgs = prop.multiProcessorCount * p.wgs_per_cu;
    p.wgs -= p.wgs % 8;
    char *B, *C, *av;
    int *d_nr, *d_rows, *d_queue, *d_cbase, *d_cn, *ctr;
    long long *d_off;
    unsigned *sink;
    CHECK(hipMalloc(&B, (size_t) p.nrowB * 2048));
    CHECK(hipMalloc(&C, (size_t) std::max(p.nrowC, 1) * 2048));
    CHECK(hipMalloc(&av, (size_t) nrounds * 8 * 256 + 4096));
    CHECK(hipMemset(B, 1, (size_t) p.nrowB * 2048));
    CHECK(hipMemset(av, 1, (size_t) nrounds * 8 * 256 + 4096));
    CHECK(hipMalloc(&d_nr, 4 * nr.size())); CHECK(hipMemcpy(d_nr, nr.data(), 4 * nr.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_off, 8 * off.size())); CHECK(hipMemcpy(d_off, off.data(), 8 * off.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_cbase, 4 * cbase.size())); CHECK(hipMemcpy(d_cbase, cbase.data(), 4 * cbase.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_cn, 4 * cn.size())); CHECK(hipMemcpy(d_cn, cn.data(), 4 * cn.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_queue, 4 * queue.size())); CHECK(hipMemcpy(d_queue, queue.data(), 4 * queue.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_rows, 4 * rows.size())); CHECK(hipMemcpy(d_rows, rows.data(), 4 * rows.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&ctr, 64)); CHECK(hipMalloc(&sink, 64));
    p.nr = d_nr; p.off = d_off; p.rows = d_rows; p.queue = d_queue; p.cbase = d_cbase; p.cn = d_cn;
    if (ROWB == 1024 && p.S != 8) { fprintf(stderr, "n128 replays teams (S = 8)\n"); return 1; }
    if (vgpr && p.S != 8) { fprintf(stderr, "the register variant replays teams (S = 8)\n"); return 1; }
    const size_t lds = vgpr ? (size_t) 2 * 8 * 2048 + (size_t) p.maxnr * 8 * 4 : (size_t) RINGR * p.S * ROWB + RINGR * 8 * 3 * VBLK + (size_t) p.maxnr * p.S * 4;
    if (vgpr) CHECK(hipFuncSetAttribute((const void *) replay_vgpr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    if (p.S == 5) CHECK(hipFuncSetAttribute((const void *) replay_kernel<5, 2048>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    else if (ROWB == 1024) CHECK(hipFuncSetAttribute((const void *) replay_kernel<8, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    else CHECK(hipFuncSetAttribute((const void *) replay_kernel<8, 2048>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int q = 0; q < nq; q++)
    {
        p.queue = d_queue + (size_t) q * 8 * p.qlen;
        std::vector<float> ms;
        for (int it = 0; it < launches + 3; it++)
        {
            CHECK(hipMemsetAsync(ctr, 0, 64, 0));
            CHECK(hipEventRecord(e0, 0));
            if (vgpr) hipLaunchKernelGGL(replay_vgpr_kernel, dim3(p.wgs), dim3(512), lds, 0, p, B, C, av, ctr, sink);
            else if (p.S == 5) hipLaunchKernelGGL((replay_kernel<5, 2048>), dim3(p.wgs), dim3(512), lds, 0, p, B, C, av, ctr, sink);
            else if (ROWB == 1024) hipLaunchKernelGGL((replay_kernel<8, 1024>), dim3(p.wgs), dim3(512), lds, 0, p, B, C, av, ctr, sink);
            else hipLaunchKernelGGL((replay_kernel<8, 2048>), dim3(p.wgs), dim3(512), lds, 0, p, B, C, av, ctr, sink);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float t;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            if (it >= 3) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        double tot = 0;
        for (float t : ms) tot += t;
        printf("%s%s%s [queue %d]: %d units, %lld rounds of %d slots, %d workgroups (%d per CU), LDS %zu B: mean %.4f ms, median %.4f, min %.4f\n", argv[1], vgpr ? " (register fetch)" : "", ROWB == 1024 ? " (1 KiB row slices)" : "", q, p.nunit, nrounds, p.S, p.wgs,
               p.wgs_per_cu, lds, tot / ms.size(), ms[ms.size() / 2], ms[0]);
    }
    return 0;
}
