#!/usr/bin/env python3
"""Write the seeded stand-in matrices (crp-spmm_amd/gen.py) as Matrix-Market files
for the example drivers, e.g.

    tools/gen_mtx.py --kind banded_fem --m 217918 --out /tmp/pwtk_standin.mtx
    mpiexec -np 1 examples/test_rp_spmm.exe /tmp/pwtk_standin.mtx 256 5 0 1

Symmetric kinds are written as "real symmetric" (lower triangle); --general writes every entry."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="banded_fem", choices=("banded_fem", "kkt3d", "fem3d", "er"))
    ap.add_argument("--m", type=int, default=217918, help="rows (banded_fem, er)")
    ap.add_argument("--g", type=int, default=32, help="grid edge (kkt3d, fem3d)")
    ap.add_argument("--deg", type=int, default=32, help="nonzeros per row (er)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--general", action="store_true")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    from crp_spmm_amd import gen
    kw = {} if a.seed is None else {"seed": a.seed}
    if a.kind == "banded_fem":
        rp, ci, va = gen.banded_fem(a.m, **kw)
        sym = True
    elif a.kind == "kkt3d":
        rp, ci, va = gen.kkt3d(a.g, **kw)
        sym = True
    elif a.kind == "fem3d":
        rp, ci, va = gen.fem3d(a.g, **kw)
        sym = True
    else:
        rp, ci, va = gen.erdos_renyi(a.m, a.m, a.deg, **kw)
        sym = False
    m = len(rp) - 1
    rows = np.repeat(np.arange(m, dtype=np.int64), np.diff(rp))
    cols = ci.astype(np.int64)
    if sym and not a.general:
        keep = cols <= rows
        rows, cols, va = rows[keep], cols[keep], va[keep]
    with open(a.out, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real %s\n" % ("symmetric" if sym and not a.general else "general"))
        f.write("%% %s stand-in written by tools/gen_mtx.py (seeded, see crp-spmm_amd/gen.py)\n" % a.kind)
        f.write("%d %d %d\n" % (m, m, rows.size))
        step = 1 << 20
        for s in range(0, rows.size, step):
            e = min(rows.size, s + step)
            blk = np.char.add(np.char.add(np.char.add((rows[s:e] + 1).astype(str), " "),
                                          np.char.add((cols[s:e] + 1).astype(str), " ")),
                              np.char.mod("%.17g", va[s:e]))
            f.write("\n".join(blk.tolist()))
            f.write("\n")
    print("%s: %d x %d, %d stored entries" % (a.out, m, m, rows.size))


if __name__ == "__main__":
    main()
