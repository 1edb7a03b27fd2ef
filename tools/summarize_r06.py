#!/usr/bin/env python3
"""Condense tools/profile_r06.sh's output: kernel stats of the bench command with every call forced into ONE launch (`onelaunch`: RCV_FR_SPLIT=0 -- a kernel's
duration IS the launch time, for the headline kernel and the dominant kernel of every other_configs record) and in the library's default form (`split`: a
64-frame call = two 32-frame launches on the context's two streams: their overlap from the start / end timestamps, time per CALL = two launches), and the
PMC traffic per 64-frame launch from two passes of the one-launch command (-> pmc_traffic.json)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
ALG = 3185049600


def rows_of(d, suffix):
    for f in glob.glob(os.path.join(out, d, "**", "*" + suffix), recursive=True):
        return list(csv.DictReader(open(f)))
    return []


for d in ("onelaunch", "split"):
    print(f"==== {d}: bench line")
    try:
        line = [l for l in open(os.path.join(out, d + ".json")) if l.startswith("{")][-1]
        j = json.loads(line)
        r = j["roofline"]
        print(f"   value {j['value']} Mpix/s  ms_per_step {j['ms_per_step']}  in_flight {r['in_flight']}  launch_ms {r['launch_ms']}  frac {r['frac']}  single_stream {r.get('single_stream')}")
    except Exception as e:  # noqa: BLE001
        print("   (no bench line)", e)
    st = rows_of(d, "kernel_stats.csv")
    if d == "split":
        print("   (library default: a filter2D / Harris call of 16+ frames = two half-batch launches on two streams, never joined per call; the kernels' own durations are those of "
              "32-frame launches running side by side -- the one-launch durations are in the onelaunch section)")
    for row in st[:8]:
        print("   stats:", row.get("Name", "")[:86], "calls", row.get("Calls"), "avg ns", row.get("AverageNs"), "min", row.get("MinNs"), "max", row.get("MaxNs"))
    try:
        for k, rec in j.get("other_configs", {}).items():
            rr = rec["roofline"]
            print(f"   other_configs[{k}]: {rr['kernel']}  launch_ms {rr['launch_ms']}  frac {rr['frac']}")
    except Exception:  # noqa: BLE001
        pass
    tr = [r for r in rows_of(d, "kernel_trace.csv") if "k_filter_rows_chain" in r.get("Kernel_Name", "")]   # (the headline kernel only: k_filter_rows_mfma is the "3f" record)
    if not tr:
        continue
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the sustained window: the longest run of launches without an idle gap (a gap = the previous kernels all ended > 50 us before the next start)
    runs, cur, end = [], [], 0
    for r in tr:
        if cur and int(r["Start_Timestamp"]) - end > 50000:
            runs.append(cur)
            cur = []
        cur.append(r)
        end = max(end, int(r["End_Timestamp"]))
    runs.append(cur)
    w = max(runs, key=len)
    w = w[4:-4] if len(w) > 40 else w          # (without the run's ramp-up and tail)
    t0, t1 = int(w[0]["Start_Timestamp"]), int(w[-1]["Start_Timestamp"])      # start to start: len(w) - 1 launch intervals
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in w]
    span = (t1 - t0) / 1e6
    nl = len(w) - 1
    per_call = 2 if d == "split" else 1          # launches per 64-frame call
    busy2 = 0
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in w] + [(int(r["End_Timestamp"]), -1) for r in w])
    depth, last, hist = 0, ev[0][0], {}
    for t, dlt in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += dlt
        last = t
    tot = sum(hist.values())
    print(f"   trace: {nl} back-to-back launches in {span:.3f} ms wall = {span / nl * per_call:.4f} ms per 64-frame call = {ALG / (span / nl * per_call) / 1e6 / 8000:.4f} of 8 TB/s;"
          f" mean kernel duration {sum(dur) / len(dur) / 1e6:.4f} ms")
    print("   time with k kernels of this kind running:", {k: f"{100.0 * v / tot:.1f} %" for k, v in sorted(hist.items())})
    qs = sorted({r.get("Queue_Id", "?") for r in w})
    print(f"   queues: {qs}")
    print("   six consecutive launches (queue, start us, end us, duration us):")
    base = int(w[10]["Start_Timestamp"])
    for r in w[10:16]:
        s, e = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base
        print(f"      q{r.get('Queue_Id', '?')}  {s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f}")

traffic = {}


def mean_counter(d, kernel_sub, counter):
    v = [float(r["Counter_Value"]) for r in rows_of(d, "counter_collection.csv") if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v) if v else None


KEYS = {"3": ("k_filter_rows_chain", "filter2d_i8_7x7_hbm_bytes_per_launch"), "3s": ("k_sobel_rows", "sobel_bgr_hbm_bytes_per_launch"),
        "3f": ("k_filter_rows_mfma", "filter2d_sobel_fused_hbm_bytes_per_launch"), "4": ("k_warp_resize_stage", "warp_resize_fused_hbm_bytes_per_launch"),
        "5": ("k_harris_fused", "harris_pipeline_hbm_bytes_per_launch")}


def synth_calibration():
    """k_synth writes known byte counts: the largest launches of the run are the 64 x 4K BGR batches"""
    v = sorted(float(r["Counter_Value"]) for r in rows_of("pmc_WRITE_SIZE", "counter_collection.csv") if "k_synth" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE")
    return v[len(v) // 2] if v else None


synth_w = synth_calibration()
print("==== PMC traffic per launch (FETCH_SIZE x 2: gfx950 counts 128-B read requests as 64 B, MI355X_MICROARCH.md; WRITE_SIZE as is, KiB)")
print(f"   calibration: k_synth WRITE_SIZE (median launch) {synth_w} KiB; a 64 x 4K BGR batch is {64 * 2160 * 3840 * 3 / 1024} KiB, a 32 x 8K batch {32 * 4320 * 7680 * 3 / 1024} KiB")
for cfg, (ksub, key) in KEYS.items():
    f, w = mean_counter("pmc_FETCH_SIZE", ksub, "FETCH_SIZE"), mean_counter("pmc_WRITE_SIZE", ksub, "WRITE_SIZE")
    if f is None or w is None:
        print(f"   config {cfg}: no counters")
        continue
    rd, wr = f * 1024 * 2, w * 1024
    traffic[key] = int(rd + wr)
    traffic[key.replace("_hbm_bytes_per_launch", "_read_bytes")] = int(rd)
    traffic[key.replace("_hbm_bytes_per_launch", "_write_bytes")] = int(wr)
    print(f"   config {cfg} ({ksub}): read {rd / 1e9:.4f} GB  written {wr / 1e9:.4f} GB  total {(rd + wr) / 1e9:.4f} GB")
traffic["note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of the bench command with RCV_FR_SPLIT=0 (one launch per call; tools/profile_r06.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B read requests as 64 B)"
traffic["calibration"] = {"k_synth_write_size_kib": synth_w, "k_synth_known_bytes": 64 * 2160 * 3840 * 3}
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("== traffic:", json.dumps(traffic))
