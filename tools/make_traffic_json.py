#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the PMC summaries of tools/prof_r04.sh (tools/prof_pmc.sh output): per workload the
bytes that crossed the L2's memory side per launch = 2 x FETCH_SIZE (gfx950 correction for 16-B/lane streams,
MI355X_MICROARCH.md, HBM section) + WRITE_SIZE.  bench.py prints them as roofline.traffic with their source.

usage: make_traffic_json.py profiles/r04_pmc_summary_*.txt > profiles/r04_traffic.json"""
import json
import re
import sys

WORK = {"pwtk_standin": ("pwtk", 256, "f64"), "shell_standin": ("pwtk_shell", 256, "f64"), "kkt": ("kkt", 256, "f64"),
        "fem3d_f32": ("fem3d", 1024, "f32"), "fem3d": ("fem3d", 1024, "f64"), "queen_size_f32": ("fem3d_queen", 1024, "f32")}
entries = []
for path in sys.argv[1:]:
    tag = re.sub(r".*r[0-9][0-9]_pmc_summary_|\.txt$", "", path)
    if tag not in WORK:
        continue
    txt = open(path).read()
    sym = re.search(r"^== (.*)$", txt, re.M).group(1)
    fetch = float(re.search(r"FETCH_SIZE\s+mean ([0-9.e+]+)", txt).group(1))
    write = float(re.search(r"WRITE_SIZE\s+mean ([0-9.e+]+)", txt).group(1))
    hit, miss = (float(re.search(r"%s\s+mean ([0-9.e+]+)" % c, txt).group(1)) for c in ("TCC_HIT_sum", "TCC_MISS_sum"))
    m, n, dt = WORK[tag]
    entries.append({"matrix": m, "n": n, "dtype": dt, "kernel": "team2-R8", "kernel_symbol": "crp::" + sym,
                    "FETCH_SIZE_KB_reported": fetch, "FETCH_SIZE_bytes_corrected_x2": int(2 * fetch * 1024), "WRITE_SIZE_KB": write,
                    "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024), "L2_hit_rate": hit / (hit + miss),
                    "source": "%s (rocprofv3 --pmc, one counter group per pass, tools/prof_pmc.sh via tools/prof_r04.sh; a run of this "
                              "round on the final code, not of the bench invocation that prints it)" % path})
print(json.dumps({"note": "gfx950: FETCH_SIZE reports half the bytes of 16-B/lane streams (MI355X_MICROARCH.md, HBM section; confirmed "
                          "this round by TCC_EA0_RDREQ_128B: all fabric reads of these kernels are 128-byte requests counted as 64), so it is "
                          "doubled; WRITE_SIZE is exact. Infinity-Cache hits are counted in both.", "entries": entries}, indent=1))
