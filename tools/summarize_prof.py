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

# ---- HBM traffic of the dominant kernel, per launch (bench.py reads profiles/pmc_traffic.json) ----
import json
def mean_counter(dirname, kernel_sub, counter):
    for f in glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
        if v:
            return sum(v) / len(v)
    return None
KERNEL = os.environ.get("RCV_PROF_KERNEL", "k_filter_rows_mfma")   # the dominant kernel of bench.py
fetch_kb = mean_counter("pmc_FETCH_SIZE", KERNEL, "FETCH_SIZE")
write_kb = mean_counter("pmc_WRITE_SIZE", KERNEL, "WRITE_SIZE")
if fetch_kb and write_kb:
    # units: KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B although wide reads
    # are 128-B requests -> double it.  WRITE_SIZE checked against k_synth's known 1 592 524 800 B in the same runs.
    rd, wr = fetch_kb * 1024 * 2, write_kb * 1024
    synth_w = mean_counter("pmc_WRITE_SIZE", "k_synth", "WRITE_SIZE")
    res = {"kernel": KERNEL, "filter2d_i8_7x7_hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
           "fetch_size_raw_kib": fetch_kb, "write_size_raw_kib": write_kb,
           "calibration": {"k_synth_write_size_kib": synth_w, "k_synth_known_bytes": 64 * 2160 * 3840 * 3},
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B read requests as 64 B); separate --pmc passes"}
    print("== traffic:", json.dumps(res))
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)
