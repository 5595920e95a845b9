#!/usr/bin/env python3
"""Generates tests/golden/mat_redist_P{2,4}.json with the REFERENCE's own mat_redist engine
(oracle/_ref/ref_mat_redist_dump = our driver + /root/reference/src/{mat_redist,dev_type,utils}.c
compiled unmodified), run under MPICH in the build container:

    python tests/golden/make_golden_redist.py

Each fixture holds the scenarios (rectangles per rank), and per scenario and rank the plan fields
of struct mat_redist_engine (src/mat_redist.h:22-34) and the redistributed block."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def scenarios(P, M, N, rng):
    out = []
    rows = np.linspace(0, M, P + 1).astype(int)
    # 1: 1D row blocks -> 1D column blocks
    cols = np.linspace(0, N, P + 1).astype(int)
    out.append([[rows[r], 0, rows[r + 1] - rows[r], N, 0, cols[r], M, cols[r + 1] - cols[r]] for r in range(P)])
    # 2: gather everything on rank 0 (examples/test_para2d_spmm.c:186-200), others request nothing
    pr = 2 if P % 2 == 0 else 1
    pc = P // pr
    rr = np.linspace(0, M, pr + 1).astype(int)
    cc = np.linspace(0, N, pc + 1).astype(int)
    sc = []
    for r in range(P):
        i, j = r // pc, r % pc
        req = [0, 0, M, N] if r == 0 else [0, 0, 0, 0]
        sc.append([rr[i], cc[j], rr[i + 1] - rr[i], cc[j + 1] - cc[j]] + req)
    out.append(sc)
    # 3: every rank requests the whole matrix (requests may overlap, sources may not)
    out.append([[rows[r], 0, rows[r + 1] - rows[r], N, 0, 0, M, N] for r in range(P)])
    # 4: uneven row blocks with an empty owner -> random rectangles (partial coverage)
    cut = sorted(rng.choice(np.arange(1, M), size=P - 2, replace=False).tolist()) if P > 2 else []
    bounds = [0] + cut + [M, M][: (P + 1 - len(cut) - 1)]
    bounds = (bounds + [M] * (P + 1))[:P + 1]
    sc = []
    for r in range(P):
        r0, c0 = int(rng.integers(0, M - 1)), int(rng.integers(0, N - 1))
        sc.append([bounds[r], 0, bounds[r + 1] - bounds[r], N, r0, c0, int(rng.integers(1, M - r0 + 1)),
                   int(rng.integers(1, N - c0 + 1))])
    out.append(sc)
    # 5: 2D grid -> transposed grid assignment
    sc = []
    for r in range(P):
        i, j = r // pc, r % pc
        i2, j2 = r % pr, r // pr
        sc.append([rr[i], cc[j], rr[i + 1] - rr[i], cc[j + 1] - cc[j], rr[i2], cc[j2], rr[i2 + 1] - rr[i2], cc[j2 + 1] - cc[j2]])
    out.append(sc)
    return [[[int(x) for x in rect] for rect in s] for s in out]


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_mat_redist_dump")
    assert os.path.exists(exe), "run `make -C oracle` in the build container first"
    env = dict(os.environ, PATH="/opt/conda/bin:" + os.environ["PATH"])
    for P in (2, 4):
        M, N = 37, 23
        sc = scenarios(P, M, N, np.random.default_rng(P))
        path = os.path.join(HERE, "_redist_scen.txt")
        with open(path, "w") as f:
            f.write("%d %d %d %d\n" % (len(sc), P, M, N))
            for s in sc:
                for rect in s:
                    f.write(" ".join(map(str, rect)) + "\n")
        out = subprocess.run(["mpiexec", "-np", str(P), exe, path], capture_output=True, text=True, env=env, check=True).stdout
        os.remove(path)
        res = {}
        cur = None
        for line in out.splitlines():
            t = line.split()
            if not t:
                continue
            if t[0] == "S":
                cur = res.setdefault(t[1], {}).setdefault(t[3], {})
                cur.update(n_proc_send=int(t[5]), n_proc_recv=int(t[7]), send_cnt=int(t[9]), recv_cnt=int(t[11]))
            else:
                cur[t[0]] = [int(x) for x in t[1:]]
        json.dump({"P": P, "M": M, "N": N, "scenarios": sc, "expected": res},
                  open(os.path.join(HERE, "mat_redist_P%d.json" % P), "w"))
        print("P=%d: %d scenarios" % (P, len(sc)))


if __name__ == "__main__":
    main()
