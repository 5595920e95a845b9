#!/usr/bin/env python3
"""Model (CPU) of a SLIDING-WINDOW schedule for band-lattice matrices (not built): a workgroup owns a
strip = one tooth segment of the pwtk stand-in (rows j * 36000 + i * 1200 + [o0, o0 + L)), walks the B rows its strip names in
the order of their offset along the tooth -- the near band once, the four far bands (teeth i +- 1, j +- 1) as four more streams --
five rows per round, and every strip of an XCD advances one round per step with probability p (p = 1: lockstep).  One LRU of 2048
row slices per XCD, as tools/l2sim.py.  Prints requests and misses per row of B, next to the team schedule's 5.0 / 1.95 (measured).

usage: sliding_window_model.py [--L 400] [--p 1.0 0.95 0.9 0.8] [--seed 1]"""
import argparse
from collections import OrderedDict

import numpy as np

D1, D2, NI, NJ = 1200, 36000, 30, 6
BY_OFFSET = True
NEAR, FAR = 14, (0, 5)          # near band +-14, far bands at +D .. +D + 5 and -D - 5 .. -D (symmetric)


def strip_stream(i, j, o0, L):
    """B rows (global numbers) a strip needs, sorted by offset along the tooth; one entry per (stream, row)."""
    out = []
    base = j * D2 + i * D1
    lo, hi = max(0, o0 - NEAR), min(D1, o0 + L + NEAR)
    for o in range(lo, hi):
        out.append((o, base + o))                                   # near stream (own tooth)
    for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        ii, jj = i + di, j + dj
        if not (0 <= ii < NI and 0 <= jj < NJ):
            continue
        nb = jj * D2 + ii * D1
        # rows o0 .. o0 + L - 1 of this tooth name rows o + 0..5 (above) or o - 5..0 (below) of the neighbour tooth
        a, b = (o0, o0 + L + 5) if (di > 0 or dj > 0) else (o0 - 5, o0 + L)
        for o in range(max(0, a), min(D1, b)):
            out.append((o, nb + o))
    out.sort()
    if BY_OFFSET:                                                   # one round = every stream's row at one offset (<= 5 rows)
        rounds, cur, co = [], [], None
        for o, r in out:
            if o != co and cur:
                rounds.append(cur)
                cur = []
            co = o
            cur.append(r)
        rounds.append(cur)
        return rounds
    flat = [r for _, r in out]
    return [flat[q:q + 5] for q in range(0, len(flat), 5)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=400)
    ap.add_argument("--p", type=float, nargs="+", default=[1.0, 0.95, 0.9, 0.8])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--by-count", action="store_true", help="rounds of five stream entries instead of one offset per round")
    a = ap.parse_args()
    global BY_OFFSET
    BY_OFFSET = not a.by_count
    nseg = D1 // a.L
    strips = [(s, i, j) for s in range(nseg) for j in range(NJ) for i in range(NI)]
    # XCD of a strip: segments side by side, inside a segment contiguous ranges of teeth i (all j): neighbours in j always share an XCD
    per_seg = 8 / nseg
    xcd_of = {}
    for (s, i, j) in strips:
        xcd_of[(s, i, j)] = min(7, int(s * per_seg + i * per_seg / NI))
    streams = {k: strip_stream(k[1], k[2], k[0] * a.L, a.L) for k in strips}
    nB = NI * NJ * D1
    total_req = sum(len(r) for v in streams.values() for r in v)
    print("strips %d of %d rows, %.2f requests per row of B" % (len(strips), a.L, total_req / nB))
    for p in a.p:
        rng = np.random.default_rng(a.seed)
        miss = 0
        for x in range(8):
            mine = [k for k in strips if xcd_of[k] == x]
            pos = {k: 0 for k in mine}
            lru = OrderedDict()
            active = list(mine)
            while active:
                nxt = []
                for k in active:
                    if p < 1.0 and rng.random() > p:
                        nxt.append(k)
                        continue
                    st = streams[k]
                    for r in st[pos[k]]:
                        if r in lru:
                            lru.move_to_end(r)
                        else:
                            miss += 1
                            lru[r] = None
                            if len(lru) > a.rows:
                                lru.popitem(last=False)
                    pos[k] += 1
                    if pos[k] < len(st):
                        nxt.append(k)
                active = nxt
        print("p = %.2f: misses %.2f x B (hit rate %.3f)" % (p, miss / nB, 1 - miss / total_req))


if __name__ == "__main__":
    main()
