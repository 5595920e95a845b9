#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs produced by tools/prof_pmc.sh / prof_mem.sh: per kernel name, the
mean of every counter over its dispatches (kernels whose name holds argv[2], default "spmm")."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
pattern = sys.argv[2] if len(sys.argv) > 2 else "spmm"      # substring a kernel name must hold
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if pattern not in name:
                continue
            short = name.split("(")[0].replace("void crp::", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, ctrs in acc.items():
    print("==", k)
    for c, v in sorted(ctrs.items()):
        print("  %-32s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
    if "FETCH_SIZE" in ctrs:
        f = sum(ctrs["FETCH_SIZE"]) / len(ctrs["FETCH_SIZE"])
        print("  -> FETCH_SIZE KB %.0f = %.1f MB as reported; x2 gfx950 correction for 16-B/lane streams = %.1f MB"
              % (f, f / 1024, 2 * f / 1024))
    if "WRITE_SIZE" in ctrs:
        w = sum(ctrs["WRITE_SIZE"]) / len(ctrs["WRITE_SIZE"])
        print("  -> WRITE_SIZE = %.1f MB" % (w / 1024))
    # memory side (tools/prof_mem.sh)
    mean = lambda c: sum(ctrs[c]) / len(ctrs[c])
    if "TCC_EA0_RDREQ_LEVEL_sum" in ctrs and "TCC_EA0_RDREQ_sum" in ctrs and mean("TCC_EA0_RDREQ_sum") > 0:
        print("  -> mean fabric read latency %.0f TCC cycles (RDREQ_LEVEL / RDREQ)" % (mean("TCC_EA0_RDREQ_LEVEL_sum") / mean("TCC_EA0_RDREQ_sum")))
    if "TCC_EA0_WRREQ_LEVEL_sum" in ctrs and "TCC_EA0_WRREQ_sum" in ctrs and mean("TCC_EA0_WRREQ_sum") > 0:
        print("  -> mean fabric write latency %.0f TCC cycles (WRREQ_LEVEL / WRREQ)" % (mean("TCC_EA0_WRREQ_LEVEL_sum") / mean("TCC_EA0_WRREQ_sum")))
    if all(c in ctrs for c in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
        b = 32 * mean("TCC_EA0_RDREQ_32B_sum") + 64 * mean("TCC_EA0_RDREQ_64B_sum") + 128 * mean("TCC_EA0_RDREQ_128B_sum")
        rest = mean("TCC_EA0_RDREQ_sum") - mean("TCC_EA0_RDREQ_32B_sum") - mean("TCC_EA0_RDREQ_64B_sum") - mean("TCC_EA0_RDREQ_128B_sum")
        print("  -> fabric read bytes by request size: %.1f MB (+ %.0f requests of no listed size)" % (b / 1e6, rest))
