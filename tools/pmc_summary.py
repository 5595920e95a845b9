#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs produced by tools/prof_pmc.sh: per kernel name, the
mean of every counter over its dispatches (SpMM kernels only)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "spmm" not in name:
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
