#!/usr/bin/env python3
"""Generates crp-spmm_amd/csrc/team2_consume.inc: the hand-scheduled gfx950 instruction stream of the round loop of
the LDS-sharing SpMM kernel (csrc/team2_kernel.hip).

Why a generated asm text, and why the whole loop
  * up to two parts' LDS reads are in flight under counted s_waitcnt lgkmcnt(N) while the FMAs of the part before
    issue, and a register whose load is in flight must stay out of the compiler's hands (hipcc treats an asm
    load's destination as written when the statement ends and may copy it before the data has landed); the
    straight-line FMA sequences (below) are shared by all rounds, so the accumulators must sit in the same physical
    registers whenever one is called.  Both hold only inside ONE asm statement: the loop over the rounds of a team is
    that statement, its staging registers are fixed VGPRs / SGPRs named in the clobber list, the accumulators and
    everything the compiler prepared are operands;
  * the scalar unit is the scarce resource of this kernel (one SALU issue per cycle and CU, shared by 16 waves).
    The first version tested one mask bit per row and let the compiler write the round's bookkeeping: 119 M
    scalar instructions per launch on the pwtk stand-in, and the kernel was bound by them.  Now
      - a part names a CONTIGUOUS row range; its FMAs are straight-line code reached by one computed call: one
        sequence per (first, length) and staging buffer, each aligned to 2^SEQ bytes, entered with s_swappc_b64, left with s_setpc_b64
        (per part: 2 x s_bfe, s_lshl, s_add, s_addc + call / return; nothing per row);
      - the loop is unrolled over the NSET = 4 ring sets, so every LDS offset of a round (ring set, value slot,
        record) is an immediate; what a round does is steered by flag bits of its record.

Record of (round r, wave w), 4 words (panel_format.h, Team2Host):
    w0 : bits 0-2 part count c (0..4) | bits 4+3i.. ring slot of part i | bit 16 ISSUE (fetch for round r + D)
         | bit 17 TAIL (fewer than D-1 younger rounds in flight at the top of round r: wait vmcnt(0))
         | bit 18 LAST round of the team | bit 19 RECS (this wave fetches the next record block now)
         | bits 21-26 value position of part 0 (below)
    w1 : bits 6i.. range of part i as first * 8 + len - 1 | bits 24-29 value position of part 1
         | bits 30-31 size class q of the value block of round r + D: it holds at most 8 (q + 1) values
    w2 : bits 0-19 value-stream offset of round r + D, in units of 4 values | bits 20-25, 26-31 value positions of parts 2, 3
    w3 : column of round r + D (an empty slot names a row of the team: a fetch nobody reads)
Values are COMPACT: a part of len rows stores len values (round 2 stored 8 per part, zeros for the rows outside the
range: 146 MB instead of 89 MB on the pwtk stand-in, 4.2 x the nonzeros on the nlpkkt stand-in).  A round's values
are one block of the wave's stream (padded to 4 values), DMA'd to the wave's value slot; part i's first value sits
prefix_i values into the block, and its "value position" is prefix_i + 7 - first_i: lane l reads the value at
(position - 7 + (l & 7)) -- the value of row l & 7 when that row is in the range, anything readable otherwise (never used).
Per round: [wait own DMAs of the round; s_barrier] -> issue (values: 16-lane LDS-DMA of 256 bytes; row: NV DMAs of 1
KiB) -> read the next record -> parts: B row slice by ds_read_b128 per 16-byte piece, the part's 8 values by ONE
ds_read_b64 (lane l holds value l & 7), call of sequence code_i: rows first..first+len-1, per row
    v_fmac_f64_dpp acc, values, slice row_newbcast:row   (NV * 2 of them)
The reads of part i + 1 are issued before the wait for part i.
fp32 instances (gen(..., f32=True)): the same loop with 128-byte value slots (8 floats per part: ds_read_b32, lane l
holds value l & 7); per row one v_mov_b32_dpp broadcasts the value and NV * 2 x v_pk_fma_f32 take the four floats of the
lane's 16-byte piece two at a time (accumulators are 64-bit pairs, as in fp64).
T2_BPOL (environment, generation time) appends a cache policy to the B-row DMAs; measured on the pwtk stand-in:
nt +35 % time, sc0 / sc1 / sc0 sc1 within +-2 %; the committed file uses none.

usage: tools/gen_team2_asm.py > crp-spmm_amd/csrc/team2_consume.inc
"""
import os
import sys

# cache policy of the B-row DMA (experiment knob at generation time): '', ' nt', ' sc0', ' sc1', ' sc0 sc1'
BPOL = os.environ.get('T2_BPOL', '')
# ... of the value and record DMAs (A's streams are read once)
APOL = os.environ.get('T2_APOL', '')

D = 3
NSET = 4
VSLOT = 256
VBASE = 100          # fixed VGPRs: A0(4) A1(4) AV(2) B0(4) B1(4) BV(2) TA TV REC(4) TP(2: fp32 broadcast pair)
# ... of the one-piece instances (nv = 1: 16 accumulator pairs instead of 32): low enough that the kernel fits 80 VGPRs, i.e. six
# waves per SIMD = THREE workgroups per CU (their LDS, 43 KiB each, allows it) -- below 256 columns the round is a chain of
# latencies (DMA wait, barrier, LDS reads of at most four parts), not a queue of requests, and a third workgroup fills it
VBASE1 = 52
# ... of the HALF-piece instances (below): 8 accumulator pairs; 64 VGPRs = eight waves per SIMD = FOUR workgroups per CU (27 KiB of LDS each)
VBASEH = 36
SBASEH = 56          # ... and their fixed SGPRs: eight waves per SIMD leave a wave 80 SGPRs, the top eight of them reserved (VCC, FLAT_SCRATCH, XNACK_MASK)
SBASE = 84           # fixed SGPRs: pc(2) ret(2) tblA(2) tblB(2) t cnt rowbase(2) recsrc(2)
NCODE = 64          # sequences per staging buffer: index first * 8 + len - 1 (28 of them unused)
NVREG = 28
NSREG = 14


def gen(nv, has_b1, f32=False, compact=True, half=False):
    """nv = 16-byte pieces per lane and row; f32: 4 floats per piece (values 4 bytes, 8 per part = 32 bytes, value
    slot 128 bytes), else 2 doubles per piece (values 8 bytes, 64 bytes per part, value slot 256 bytes).
    half (nv = 1 only): the HALF-piece instance for operands of at most 64 fp64 / 128 fp32 columns -- a lane holds 8 bytes of a row
    (one double / two floats), a ring slot is 512 bytes, fetched by the lower 32 lanes of the wave's 16-byte-per-lane DMA (EXEC masked),
    read back with ds_read_b64; ONE FMA per row and part instead of two half-empty ones, 8 accumulator pairs, and a kernel small enough
    for four workgroups per CU.
    compact = False: the value blocks hold 8 values per part (part i's row r at 8 i + r): no value position to decode per part and
    a value DMA of fixed size -- two instructions per part and three per round fewer; the instance for panels that are well filled
    (pwtk stand-in, fill 0.61: 3 % faster at n = 256 and 10 % at n = 128 than on compact values; nlpkkt stand-in, 0.23: 2-5 % slower)."""
    tw = 8                                          # waves (= slots of a round) per team
    sbits = 3                                       # bits of a slot number in w0
    fbase = 16                                      # first flag bit of w0: ISSUE, TAIL, LAST, RECS
    recrow = 16 * tw                                # bytes of the records of one round
    recblk = 8 * recrow                             # ... of a record block
    vw = 4 if f32 else 2                            # elements per 16-byte piece
    vgrp = 32 if f32 else 64                        # bytes of one part's 8 values
    vslot = 4 * vgrp
    assert not half or nv == 1
    b = VBASE if nv == 2 else (VBASEH if half else VBASE1)
    A = {"s0": b, "s1": b + 4, "v": b + 8}
    B = {"s0": b + 10, "s1": b + 14, "v": b + 18}
    TA, TV, REC = b + 20, b + 21, b + 22
    TP = b + 26                                     # fp32: pair whose low register holds the row's broadcast value
    sb = SBASEH if half else SBASE
    PC, RET, TBA, TBB, T, CNT, RB, RS = sb, sb + 2, sb + 4, sb + 6, sb + 8, sb + 9, sb + 10, sb + 12
    slotb = 512 if half else 1024 * nv
    setb = tw * slotb
    slot_shift = 11 if nv == 2 else (9 if half else 10)
    # 2^x >= the longest sequence: fp64 8 rows x nv * 2 FMAs x 8 bytes + return; fp32 8 rows x (1 + nv * 2) x 8 bytes + return
    seq_align = {(1, False): 8, (2, False): 9, (1, True): 8, (2, True): 9}[(nv, f32)]
    if half:
        seq_align = 8 if f32 else 7                  # fp64: 8 rows x 1 FMA x 8 bytes + return; fp32: 8 rows x (1 + 1) x 8 bytes + return
    opr = nv + 1                                    # DMAs a wave issues per round
    tag = "%s%s%d%s_%%=" % ("s" if f32 else "d", "h" if half else str(nv), 1 if has_b1 else 0, "" if compact else "f")
    L = []
    emit = L.append

    def rd(k, i, X):
        emit("s_bfe_u32 s%d, %%[w0], 0x%x" % (T, (sbits << 16) | (4 + sbits * i)))
        koff = k * setb
        emit("v_lshl_add_u32 v%d, s%d, %d, %%[seta]" % (TA, T, slot_shift))
        if half:
            emit("ds_read_b64 v[%d:%d], v%d offset:%d" % (X["s0"], X["s0"] + 1, TA, koff))
        else:
            emit("ds_read_b128 v[%d:%d], v%d offset:%d" % (X["s0"], X["s0"] + 3, TA, koff))
        if nv == 2:
            emit("ds_read_b128 v[%d:%d], v%d offset:%d" % (X["s1"], X["s1"] + 3, TA, koff + 1024))
        if not compact:
            # 8 values per part: the lane's row of part i at a fixed offset (vsl = slot base - 7 values + this lane's row)
            if f32:
                emit("ds_read_b32 v%d, %%[vsl] offset:%d" % (X["v"], k * vslot + vgrp * i + 28))
            else:
                emit("ds_read_b64 v[%d:%d], %%[vsl] offset:%d" % (X["v"], X["v"] + 1, k * vslot + vgrp * i + 56))
            return
        # the part's values: position field -> address (vsl = slot base - 7 values + this lane's row)
        word, bit = (("w0", fbase + 5), ("w1", 24), ("w2", 20), ("w2", 26))[i]
        emit("s_bfe_u32 s%d, %%[%s], 0x%x" % (T, word, (6 << 16) | bit))
        emit("v_lshl_add_u32 v%d, s%d, %d, %%[vsl]" % (TV, T, 2 if f32 else 3))
        if f32:
            emit("ds_read_b32 v%d, v%d offset:%d" % (X["v"], TV, k * vslot))
        else:
            emit("ds_read_b64 v[%d:%d], v%d offset:%d" % (X["v"], X["v"] + 1, TV, k * vslot))

    def call(i, tb):
        emit("s_bfe_u32 s%d, %%[w1], 0x%x" % (T, (6 << 16) | (6 * i)))
        emit("s_lshl_b32 s%d, s%d, %d" % (T, T, seq_align))
        emit("s_add_u32 s%d, s%d, s%d" % (PC, tb, T))
        emit("s_addc_u32 s%d, s%d, 0" % (PC + 1, tb + 1))
        emit("s_swappc_b64 s[%d:%d], s[%d:%d]" % (RET, RET + 1, PC, PC + 1))

    # ---- once: table bases, fixed copies of what is advanced in the loop
    emit("s_getpc_b64 s[%d:%d]" % (TBA, TBA + 1))
    emit(".Lt2pc%s:" % tag)
    emit("s_add_u32 s%d, s%d, .Lt2tabA%s-.Lt2pc%s" % (TBA, TBA, tag, tag))
    emit("s_addc_u32 s%d, s%d, 0" % (TBA + 1, TBA + 1))
    emit("s_add_u32 s%d, s%d, %d" % (TBB, TBA, NCODE << seq_align))
    emit("s_addc_u32 s%d, s%d, 0" % (TBB + 1, TBA + 1))
    emit("s_mov_b32 s%d, %%[rslo]" % RS)
    emit("s_mov_b32 s%d, %%[rshi]" % (RS + 1))
    emit("s_branch .Lt2body0%s" % tag)               # round 0: the compiler's code has waited, synchronised and read the record

    for k in range(NSET):
        kd = (k + D) % NSET                          # set of round r + D
        emit(".Lt2round%d%s:" % (k, tag))
        emit("s_bitcmp1_b32 %%[w0], %d" % (fbase + 1))
        emit("s_cbranch_scc1 .Lt2tw%d%s" % (k, tag))     # TAIL: out of line (behind the loop), back at .Lt2bar
        emit("s_waitcnt vmcnt(%d)" % ((D - 1) * opr))
        emit(".Lt2bar%d%s:" % (k, tag))
        emit("s_barrier")
        emit(".Lt2body%d%s:" % (k, tag))
        # -- issue for round r + D.  The B row first: every wave of the workgroup runs this block right behind the barrier, all
        # of them on the CU's one scalar unit, and what the CU is short of is requests in flight -- the row DMAs go out
        # before the bookkeeping of the value DMA (round 3: with the value block's size and the empty-slot tests in front of
        # them the pwtk stand-in ran 6 % slower).
        emit("s_bitcmp1_b32 %%[w0], %d" % fbase)
        emit("s_cbranch_scc0 .Lt2ni%d%s" % (k, tag))
        # (M0 first: the scalar instructions of the address stand between its write and the DMA that reads it -- no s_nop)
        emit("s_add_u32 m0, %%[wslot], %d" % (kd * setb))
        if has_b1:
            # c < 0: row ~c of B1
            emit("s_cmp_lt_i32 %[w3], 0")
            emit("s_cbranch_scc1 .Lt2b1%d%s" % (k, tag))
        emit("s_mul_hi_u32 s%d, %%[w3], %%[ld0]" % (RB + 1))
        emit("s_mul_i32 s%d, %%[w3], %%[ld0]" % RB)
        emit("s_add_u32 s%d, s%d, %%[b0lo]" % (RB, RB))
        emit("s_addc_u32 s%d, s%d, %%[b0hi]" % (RB + 1, RB + 1))
        if has_b1:
            emit("s_branch .Lt2bj%d%s" % (k, tag))
            emit(".Lt2b1%d%s:" % (k, tag))
            emit("s_not_b32 s%d, %%[w3]" % T)
            emit("s_mul_hi_u32 s%d, s%d, %%[ld1]" % (RB + 1, T))
            emit("s_mul_i32 s%d, s%d, %%[ld1]" % (RB, T))
            emit("s_add_u32 s%d, s%d, %%[b1lo]" % (RB, RB))
            emit("s_addc_u32 s%d, s%d, %%[b1hi]" % (RB + 1, RB + 1))
            emit(".Lt2bj%d%s:" % (k, tag))
        if half:
            emit("s_mov_b64 exec, 0xffffffff")       # 32 lanes x 16 bytes = the 512-byte slice (the value DMA below sets EXEC again)
        emit("global_load_lds_dwordx4 %%[voffa], s[%d:%d]%s" % (RB, RB + 1, BPOL))
        if nv == 2:
            emit("global_load_lds_dwordx4 %%[voffb], s[%d:%d] offset:1024%s" % (RB, RB + 1, BPOL))
        # the wave's value block of round r + D: offset in units of 4 values, 4 (q + 1) lanes of 16 bytes (fp32: 2 (q + 1))
        emit("s_and_b32 s%d, %%[w2], 0xfffff" % T)
        emit("s_add_u32 m0, %%[vringw], %d" % (kd * vslot))      # (early: no s_nop in front of the DMA that reads it)
        emit("v_lshl_add_u32 v%d, s%d, %d, %%[lane16]" % (TV, T, 4 if f32 else 5))
        if compact:
            emit("s_bfe_u32 s%d, %%[w1], 0x2001e" % T)
            if f32:
                emit("s_lshl1_add_u32 s%d, s%d, 2" % (T, T))
            else:
                emit("s_lshl2_add_u32 s%d, s%d, 4" % (T, T))
            emit("s_bfm_b64 exec, s%d, 0" % T)
        else:
            emit("s_mov_b64 exec, 0x%x" % ((1 << (vslot // 16)) - 1))
        emit("global_load_lds_dwordx4 v%d, %%[vbase]%s" % (TV, APOL))
        emit("s_mov_b64 exec, -1")
        emit(".Lt2ni%d%s:" % (k, tag))
        # -- next record block (one wave, every 8 rounds)
        emit("s_bitcmp1_b32 %%[w0], %d" % (fbase + 3))
        emit("s_cbranch_scc0 .Lt2nr%d%s" % (k, tag))
        emit("s_add_u32 s%d, s%d, %d" % (RS, RS, recblk))
        emit("s_addc_u32 s%d, s%d, 0" % (RS + 1, RS + 1))
        emit("s_xor_b32 %%[recdst], %%[recdst], %d" % recblk)
        emit("s_mov_b32 m0, %[recdst]")
        emit("s_nop 0")
        emit("global_load_lds_dwordx4 %%[lane16], s[%d:%d]%s" % (RS, RS + 1, APOL))
        for piece in range(1, recblk // 1024):      # (an immediate offset moves the global AND the LDS address)
            emit("global_load_lds_dwordx4 %%[lane16], s[%d:%d] offset:%d%s" % (RS, RS + 1, 1024 * piece, APOL))
        emit(".Lt2nr%d%s:" % (k, tag))
        # -- record of the next round (LDS reads return in order: it is there when the parts are done)
        if k < NSET - 1:
            emit("ds_read_b128 v[%d:%d], %%[recaddr] offset:%d" % (REC, REC + 3, (k + 1) * recrow))
        else:
            emit("v_add_u32 %%[recoff], %d, %%[recoff]" % (NSET * recrow))
            emit("v_and_b32 %%[recoff], 0x%x, %%[recoff]" % (2 * recblk - 1))
            emit("v_add_u32 %[recaddr], %[recbase], %[recoff]")
            emit("ds_read_b128 v[%d:%d], %%[recaddr]" % (REC, REC + 3))
        # -- parts
        # (two compares on the way to the common counts 3 and 4, three to the others)
        emit("s_and_b32 s%d, %%[w0], 7" % CNT)
        emit("s_cmp_lt_u32 s%d, 3" % CNT)
        emit("s_cbranch_scc1 .Lt2lo%d%s" % (k, tag))
        emit("s_cmp_eq_u32 s%d, 3" % CNT)
        emit("s_cbranch_scc1 .Lt2v%d_3%s" % (k, tag))
        nrd = nv + 1
        for c in (4, 3, 2, 1):
            if c == 2:
                emit(".Lt2lo%d%s:" % (k, tag))
                emit("s_cmp_eq_u32 s%d, 0" % CNT)
                emit("s_cbranch_scc1 .Lt2pe%d%s" % (k, tag))
                emit("s_cmp_eq_u32 s%d, 1" % CNT)
                emit("s_cbranch_scc1 .Lt2v%d_1%s" % (k, tag))
            if c in (3, 1):
                emit(".Lt2v%d_%d%s:" % (k, c, tag))
            bufs = [(A, TBA), (B, TBB)]
            rd(k, 0, A)
            for i in range(c):
                cur, tb = bufs[i % 2]
                if i + 1 < c:
                    rd(k, i + 1, bufs[(i + 1) % 2][0])
                    emit("s_waitcnt lgkmcnt(%d)" % nrd)
                else:
                    emit("s_waitcnt lgkmcnt(0)")
                call(i, tb)
            if c != 1:
                emit("s_branch .Lt2pe%d%s" % (k, tag))
        emit(".Lt2pe%d%s:" % (k, tag))
        emit("s_bitcmp1_b32 %%[w0], %d" % (fbase + 2))
        emit("s_cbranch_scc1 .Lt2done%s" % tag)
        emit("s_waitcnt lgkmcnt(0)")
        emit("v_readfirstlane_b32 %%[w0], v%d" % REC)
        emit("v_readfirstlane_b32 %%[w1], v%d" % (REC + 1))
        emit("v_readfirstlane_b32 %%[w2], v%d" % (REC + 2))
        emit("v_readfirstlane_b32 %%[w3], v%d" % (REC + 3))
        if k == NSET - 1:
            emit("s_branch .Lt2round0%s" % tag)
    # ---- out of line: the TAIL wait of every unrolled round
    for k in range(NSET):
        emit(".Lt2tw%d%s:" % (k, tag))
        emit("s_waitcnt vmcnt(0)")
        emit("s_branch .Lt2bar%d%s" % (k, tag))
    # ---- the sequences
    def aidx(r, v, w):
        """accumulator pair of row r, piece v, half w of the piece (the half-piece instance has one pair per row)"""
        return r if half else (r * nv + v) * 2 + w

    for bank, (name, X) in [(0, nx) for nx in (("A", A), ("B", B))]:
        code = 0
        for first in range(8):
            for ln in range(1, 9):
                emit(".p2align %d" % seq_align)
                if code == 0 and bank == 0:
                    emit(".Lt2tab%s%s:" % (name, tag))
                code += 1
                if first + ln > 8:
                    emit("s_setpc_b64 s[%d:%d]" % (RET, RET + 1))      # (no such range; never called)
                    continue
                for r in range(first, first + ln):
                    if f32:
                        # the row's value into every lane once, then PACKED FMAs (two floats per instruction: the fp32
                        # instance is bound by its vector instructions, 56 % of the time with one v_fmac_f32_dpp per float)
                        emit("v_mov_b32_dpp v%d, v%d row_newbcast:%d row_mask:0xf bank_mask:0xf" % (TP, X["v"], r))
                    for v in range(nv):
                        base = X["s0"] if v == 0 else X["s1"]
                        for w in range(1 if half else 2):
                            if f32:
                                emit("v_pk_fma_f32 %%[a%d], v[%d:%d], v[%d:%d], %%[a%d] op_sel_hi:[1,0,1]"
                                     % (aidx(r, v, w), base + 2 * w, base + 2 * w + 1, TP, TP + 1, aidx(r, v, w)))
                            else:
                                emit("v_fmac_f64_dpp %%[a%d], v[%d:%d], v[%d:%d] row_newbcast:%d row_mask:0xf bank_mask:0xf"
                                     % (aidx(r, v, w), X["v"], X["v"] + 1, base + 2 * w, base + 2 * w + 1, r))
                emit("s_setpc_b64 s[%d:%d]" % (RET, RET + 1))
        assert code == NCODE
    emit(".Lt2done%s:" % tag)
    emit("s_waitcnt lgkmcnt(0)")                     # the record read of the round after the last one
    return L


def main():
    out = sys.stdout
    out.write("// GENERATED by tools/gen_team2_asm.py -- do not edit; see that script for the design.\n")
    out.write("// Fixed registers v%d..v%d (one-piece instances: v%d..v%d) and s%d..s%d (and m0) must be in the clobber list of the statement.\n"
              % (VBASE, VBASE + NVREG - 1, VBASE1, VBASE1 + NVREG - 1, SBASE, SBASE + NSREG - 1))
    for name, vb, sb in (("CRP_TEAM2_CLOBBERS", VBASE, SBASE), ("CRP_TEAM2_CLOBBERS_NV1", VBASE1, SBASE), ("CRP_TEAM2_CLOBBERS_H", VBASEH, SBASEH)):
        out.write("#define %s %s, %s\n" % (name, ", ".join('"v%d"' % r for r in range(vb, vb + NVREG)),
                                           ", ".join('"s%d"' % r for r in range(sb, sb + NSREG))))
    for compact in (True, False):
     for f32 in (False, True):
      for nv in (0, 1, 2):                          # 0 = the half-piece instance
        for hb in (0, 1):
            out.write("#define CRP_TEAM2_LOOP_%s_NV%s_B%d%s \\\n" % ("F32" if f32 else "F64", nv if nv else "H", hb, "" if compact else "_F"))
            lines = gen(max(nv, 1), bool(hb), f32, compact, half=(nv == 0))
            for k, l in enumerate(lines):
                sep = "\\n\\t" if not l.endswith(":") else "\\n"
                last = k == len(lines) - 1
                out.write('    "%s%s"%s\n' % (l, "" if last else sep, "" if last else " \\"))
            out.write("\n")


if __name__ == "__main__":
    main()
