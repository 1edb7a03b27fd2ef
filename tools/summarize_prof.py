#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8:
            print("  ", ",".join(row))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== pmc:", os.path.basename(d))
        for k, cs in acc.items():
            for c, v in cs.items():
                print(f"   {k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
