#!/usr/bin/env python3
"""Plans for tools/probe/fetch_replay (a fetch-only replay of a B-row schedule on the team kernel's pipeline), pwtk stand-in, n = 256:
  teams.bin  : the team2 format's own rounds and launch grid (8 slots per round, 64 rows of C written at a team's end, 2 workgroups per CU)
  strips.bin : the sliding-window scheme of DESIGN.md section 8 / tools/sliding_window_model.py -- a unit is a strip of one tooth, a round
               is every stream's row at one offset along the tooth (5 slots), one row of C written per round, 3 workgroups per CU, every
               strip of an XCD resident at once
usage: make_plans.py OUTDIR [--L 400]"""
import argparse
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))


def write_plan(path, S, nrowB, nrowC, wgs_per_cu, write_at_end, alanes, reads, units, queues, cbase, cn):
    """units: list of int arrays [nr, S] (row numbers, every slot filled); queues: 8 lists of unit numbers, or a list of such
    (alternative launch grids, replayed one after the other)."""
    if not isinstance(queues[0][0] if len(queues[0]) else 0, (list, np.ndarray)):
        queues = [queues]
    nunit = len(units)
    nr = np.array([len(u) for u in units], dtype=np.int32)
    off = np.zeros(nunit, dtype=np.int64)
    off[1:] = np.cumsum(nr[:-1])
    nrounds = int(nr.sum())
    qlen = max(len(q) for qs in queues for q in qs)
    Q = -np.ones((len(queues), 8, qlen), dtype=np.int32)
    for k, qs in enumerate(queues):
        for x, q in enumerate(qs):
            Q[k, x, :len(q)] = q
    rows = np.concatenate([np.asarray(u, dtype=np.int32).reshape(-1) for u in units])
    assert rows.min() >= 0 and rows.max() < nrowB
    with open(path, "wb") as f:
        f.write(struct.pack("13i", 0x46524550, nunit, qlen, S, nrowB, nrowC, wgs_per_cu, write_at_end, alanes, reads, nrounds & 0x7FFFFFFF, nrounds >> 31, len(queues)))
        f.write(nr.tobytes())
        f.write(off.tobytes())
        f.write(np.asarray(cbase, dtype=np.int32).tobytes())
        f.write(np.asarray(cn, dtype=np.int32).tobytes())
        f.write(Q.tobytes())
        f.write(rows.tobytes())
    print("%s: %d units, %d rounds of %d slots, %.2f requests per row of B" % (path, nunit, nrounds, S, nrounds * S / nrowB))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--L", type=int, default=400)
    ap.add_argument("--wgs", type=int, default=3, help="workgroups per CU of the strip plan")
    ap.add_argument("--orders", action="store_true", help="also teams_orders.bin: the teams under alternative processing orders")
    ap.add_argument("--strides", type=int, nargs=2, default=[1200, 36000], help="--orders: the lattice's two strides in rows (pwtk stand-in 1200 36000; fem3d(56) 168 9408)")
    ap.add_argument("--matrix", default="pwtk", help="pwtk | shell | fem3d | kkt (bench.py's stand-ins; anything but pwtk: the teams plan only)")
    a = ap.parse_args()
    os.makedirs(a.outdir, exist_ok=True)
    import l2sim
    import sliding_window_model as sw
    from crp_spmm_amd import gen, hip
    import bench
    name = {"pwtk": "pwtk", "shell": "pwtk_shell"}.get(a.matrix, a.matrix)
    _, _, m, k, rp, ci, va = bench.build_matrix(name, None)
    perm, info = hip.locality_order_host(rp, ci, k) if a.matrix == "shell" else (None, None)       # (the stand-in the library re-orders)
    if perm is not None and info is not None:
        import scipy.sparse as sp                  # (the library builds its formats on the re-ordered rows)
        Ap = sp.csr_matrix((va, ci, rp), shape=(m, k))[perm]
        rp, ci, va = Ap.indptr.astype(np.int32), Ap.indices.astype(np.int32), Ap.data
    t = hip.team2_format_host(rp, ci, va)
    rounds = l2sim.team_rounds(t)
    units = []
    for r in rounds:
        r = r.copy()
        first = r[r != l2sim.NOCOL][0]
        r[r == l2sim.NOCOL] = first                 # an empty slot fetches a row of the team (as the kernel does)
        units.append(r)
    tg = t["tgrid"]
    queues = [[int(g) for g in run if g >= 0] for run in tg]
    tp = t["tpanel"]
    # C rows: 64 per team; the replay writes them to a contiguous block per team (the addresses differ from the kernel's, the bytes do not)
    cbase = np.arange(len(units)) * 64
    cn = np.array([int((tp[g] >= 0).sum()) * 8 for g in range(len(units))])
    parts = int(t["tinfo"][:, 2].sum())
    reads = int(round(parts / (sum(len(u) for u in units) * 8.0)))
    # value bytes per wave and round: 12 bytes per nonzero (value + its share of the records)
    ab = 12.0 * len(ci) / (sum(len(u) for u in units) * 8)
    tag = "" if a.matrix == "pwtk" else "_" + a.matrix
    write_plan(os.path.join(a.outdir, "teams%s.bin" % tag), 8, k, len(units) * 64, 2, 1, max(1, min(8, int(round(ab / 16)))), max(1, reads), units, queues, cbase, cn)
    if a.orders:
        # alternative processing orders of the same teams, the family csrc/team_order.cpp searches with its L2 model: XCD boxes of team
        # columns (pa x pb), blocks of bt positions x ba x bb columns, blocks and teams in either nesting; every order is cut into eight
        # runs of equal rounds like the format's own.  Queue 0 is the format's grid.
        nt = len(units)
        first = np.array([int(tp[g][tp[g] >= 0].min()) * 8 for g in range(nt)])
        D1, D2 = a.strides
        A, Bc, Tt = ((first % D2) // D1) // 2, (first // D2) // 2, (first % D1) // 16
        na, nb = int(A.max()) + 1, int(Bc.max()) + 1
        cands, names, seen = [queues], ["format"], set()
        boxes = [(8, 1), (4, 2), (2, 4)] + ([(1, 8)] if nb >= 8 else [])
        blocks = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 2), (1000, 1000)] + ([(2, 4), (4, 4), (4, 6), (6, 4), (8, 8)] if nb >= 8 else [])
        for (pa, pb) in boxes:
            box = np.minimum(A * pa // na, pa - 1) * pb + np.minimum(Bc * pb // nb, pb - 1)
            for bt in [1, 2, 3, 4, 6, 8, 12, 1000]:
                for (ba, bb) in blocks:
                    for flags in range(4):
                        tb, ab_, bb_ = Tt // bt, A // ba, Bc // bb
                        inner = (A, Bc, Tt) if flags & 1 else (Tt, A, Bc)             # fastest key first
                        outer = (ab_, bb_, tb) if flags & 2 else (tb, ab_, bb_)
                        order = np.lexsort(inner + outer + (box,))
                        key = order.tobytes()
                        if key in seen:
                            continue
                        seen.add(key)
                        cands.append(l2sim.xcd_queues(rounds, [int(g) for g in order]))
                        names.append("boxes %dx%d blocks %dx%dx%d flags %d" % (pa, pb, bt, ba, bb, flags))
        write_plan(os.path.join(a.outdir, "teams%s_orders.bin" % tag), 8, k, len(units) * 64, 2, 1, max(1, min(8, int(round(ab / 16)))), max(1, reads), units, cands, cbase, cn)
        with open(os.path.join(a.outdir, "teams%s_orders.txt" % tag), "w") as f:
            for k, nm in enumerate(names):
                f.write("%d %s\n" % (k, nm))
        print("  %d candidate orders" % len(cands))
    print("  parts per wave and round %.2f -> %d ring reads; value bytes per wave and round %.0f" % (parts / (sum(len(u) for u in units) * 8.0), max(1, reads), ab))
    if a.matrix != "pwtk":
        return

    # strips
    sw.BY_OFFSET = True
    nseg = sw.D1 // a.L
    NI, NJ, D1, D2 = sw.NI, sw.NJ, sw.D1, sw.D2
    strips, sunits, scb, scn = [], [], [], []
    nrow_strips = 0
    for s in range(nseg):
        for j in range(NJ + 1):                       # (the stand-in's last slab j = 6 is partial: rows up to 217917)
            for i in range(NI):
                base = j * D2 + i * D1 + s * a.L
                if base >= m:
                    continue
                L = min(a.L, m - base)
                strips.append((s, i, j, base, L))
    sw.NJ = NJ + 1
    for (s, i, j, base, L) in strips:
        st = sw.strip_stream(i, j, s * a.L, a.L)
        u = np.empty((len(st), 5), dtype=np.int64)
        for k, rr in enumerate(st):
            rr = [x for x in rr if 0 <= x < m]
            if not rr:
                rr = [min(base, m - 1)]
            u[k, :len(rr)] = rr
            u[k, len(rr):] = rr[0]                  # (a round with fewer than five streams: the first row again)
        sunits.append(u)
        scb.append(base)
        scn.append(min(L, len(st)))
    per_seg = 8.0 / nseg
    queues = [[] for _ in range(8)]
    for k, (s, i, j, base, L) in enumerate(strips):
        queues[min(7, int(s * per_seg + i * per_seg / NI))].append(k)
    print("strips per XCD:", [len(q) for q in queues])
    ab = 12.0 * len(ci) / (sum(len(u) for u in sunits) * 8)
    write_plan(os.path.join(a.outdir, "strips_L%d.bin" % a.L), 5, m, m, a.wgs, 0, max(1, min(8, int(round(ab / 16)))), max(1, reads), sunits, queues, scb, scn)
    print("  value bytes per wave and round %.0f" % ab)


if __name__ == "__main__":
    main()
