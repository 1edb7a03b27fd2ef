#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of one op (tools/profile_ops.sh) to the rows of its dominant kernel."""
import csv
import glob
import os
import sys

out, ksub, alg, pat = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
print(f"op: {pat}    kernel: *{ksub}*    algorithmic bytes per launch: {alg:.0f}")
avg_ns = None
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if ksub in row["Name"]:
            calls = int(row["Calls"])
            if avg_ns is None or calls > best_calls:
                avg_ns, best_calls = float(row["AverageNs"]), calls
                print(f"kernel-trace --stats: calls={row['Calls']} avg={float(row['AverageNs'])/1e3:.1f} us min={float(row['MinNs'])/1e3:.1f} us "
                      f"max={float(row['MaxNs'])/1e3:.1f} us   ({row['Name'][:90]})")
vals = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if ksub in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for c, v in acc.items():
            # the largest launches are the benchmarked batch (smaller ones come from other rows of the same --only filter)
            v = sorted(v)[len(v) // 2:]
            vals[c] = sum(v) / len(v)
            print(f"pmc {c:24s} n={len(v):3d} mean={vals[c]:.6g}")
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    rd, wr = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024     # FETCH_SIZE x2: gfx950 counts 128-B reads as 64 B
    print(f"HBM traffic per launch: read {rd/1e9:.3f} GB + write {wr/1e9:.3f} GB = {(rd+wr)/1e9:.3f} GB  vs algorithmic {alg/1e9:.3f} GB  "
          f"(x{(rd+wr)/alg:.3f})")
    if avg_ns:
        print(f"achieved: {alg/avg_ns:.0f} GB/s algorithmic = {alg/avg_ns/80:.1f} % of 8 TB/s (profiled run, clocks not settled);  "
              f"{(rd+wr)/avg_ns:.0f} GB/s of real traffic")
if "GRBM_GUI_ACTIVE" in vals and avg_ns:
    print(f"clock: GRBM_GUI_ACTIVE / 8 XCDs / duration = {vals['GRBM_GUI_ACTIVE']/8/avg_ns:.2f} GHz")
