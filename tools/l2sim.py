#!/usr/bin/env python3
"""Host-side model of the B-row traffic of the team2 kernel (variant 5): replays the dispatch of a format's teams
over the 8 XCDs (workgroup i runs on XCD i % 8, `--wgs` workgroups resident per XCD, every resident team issues one
round of 8 rows per time step) against one LRU of `--rows` B rows per XCD, and prints the rows that miss.
A planning tool for the team order / union order (csrc/panel_format.cpp); measured counterparts: profiles/r02_traffic.json.

usage: l2sim.py [--matrix pwtk|pwtk_shell|...] [--rows 1400] [--wgs 64] [--n 256]
Environment knobs of the packer (CRPSPMM_TEAM2_PHASE, CRPSPMM_TEAM2_SHAPE, ...) apply."""
import argparse
import os
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


NOCOL = -(1 << 31)


def team_rounds(t):
    """cols[g] = int array [rounds, 8] of the B rows team g fetches, in issue order."""
    W = t.get("waves", 8)
    rec = t["trec"].reshape(-1, 8, W, 4)
    out = []
    for g in range(t["nteam"]):
        nr, blk0 = int(t["tinfo"][g, 0]), int(t["tinfo"][g, 1])
        c = np.empty((nr, W), dtype=np.int64)
        k = min(nr, 3)
        c[:k] = t["tpro"][g, :k, :, 0]
        if nr > 3:
            nb = (nr - 3 + 7) >> 3
            c[3:] = rec[blk0:blk0 + nb, :, :, 3].reshape(-1, W)[:nr - 3].astype(np.int32)
        out.append(c)
    return out


def xcd_queues(rounds, order, nxcd=8):
    """The launch grid of build_team2 (csrc/panel_format.cpp): the order cut into pieces of equal rounds + 4 per team."""
    w = [len(rounds[g]) + 4 for g in order]
    total, acc, cuts, x = sum(w), 0, [0], 1
    for i, wi in enumerate(w):
        acc += wi
        while x < nxcd and acc * nxcd >= total * x:
            cuts.append(i + 1)
            x += 1
    cuts += [len(order)] * (nxcd + 1 - len(cuts))
    return [list(order[cuts[q]:cuts[q + 1]]) for q in range(nxcd)]


def simulate(rounds, order, rows, wgs, nxcd=8, grid=None):
    """grid (the format's tgrid: one run per XCD, -1 = none) when given, else the order cut by xcd_queues()."""
    miss = req = 0
    queues = [[int(g) for g in run if g >= 0] for run in grid] if grid is not None else xcd_queues(rounds, list(order), nxcd)
    for queue in queues:
        lru = OrderedDict()
        active = []                                       # [team, next round]
        qi = 0
        while qi < len(queue) or active:
            while len(active) < wgs and qi < len(queue):
                active.append([queue[qi], 0])
                qi += 1
            nxt = []
            for a in active:
                cols = rounds[a[0]]
                for c in cols[a[1]]:
                    c = int(c)
                    if c == NOCOL:                        # empty slot (TEAM2_NOCOL): nothing is fetched
                        continue
                    req += 1
                    if c in lru:
                        lru.move_to_end(c)
                    else:
                        miss += 1
                        lru[c] = None
                        if len(lru) > rows:
                            lru.popitem(last=False)
                a[1] += 1
                if a[1] < len(cols):
                    nxt.append(a)
            active = nxt
    return req, miss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="pwtk")
    ap.add_argument("--rows", type=int, nargs="+", default=[1000, 1400, 2048])
    ap.add_argument("--wgs", type=int, default=64)
    ap.add_argument("--n", type=int, default=256)
    a = ap.parse_args()
    import bench
    from crp_spmm_amd import hip
    _, _, m, k, rp, ci, va = bench.build_matrix(a.matrix, None)
    perm = None
    if os.environ.get("L2SIM_REORDER", "auto") != "0":
        perm, info = hip.locality_order_host(rp, ci, k)
        if perm is not None and info is not None and info.get("applied", True) and os.environ.get("L2SIM_REORDER") == "1":
            import scipy.sparse as sp
            A = sp.csr_matrix((va, ci, rp), shape=(m, k))[perm]
            rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
        else:
            perm = None
    t = hip.team2_format_host(rp, ci, va)
    rounds = team_rounds(t)
    tot = sum(len(r) for r in rounds)
    uniq = len(np.unique(np.concatenate([r.reshape(-1) for r in rounds])))
    W = t.get("waves", 8)
    if "--wgs" not in sys.argv and W == 16:
        a.wgs = 32                                     # one 1024-thread workgroup per CU
    print("%s: %d rows, %d teams of %d, %d rounds, %.2f slots/row, distinct B rows %d" % (a.matrix, m, t["nteam"], W, tot, float(W) * tot / m, uniq))
    for rows in a.rows:
        req, miss = simulate(rounds, list(t["torder"]), rows, a.wgs, grid=t.get("tgrid"))
        print("  L2 rows %5d wgs/xcd %d: requests %.3f GB, misses %.3f GB (%.2f x B), hit rate %.3f"
              % (rows, a.wgs, req * a.n * 8 / 1e9, miss * a.n * 8 / 1e9, miss / uniq, 1 - miss / req))


if __name__ == "__main__":
    main()
