#!/usr/bin/env python3
"""profiles/ops_bench.json (tools/bench_ops.py --cpu) -> markdown table on stdout / profiles/ops_table.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "ops_bench.json")
d = json.load(open(src))
# what bounds each op (DESIGN_HISTORY.md 4): ops whose fixed arithmetic exceeds what the vector ALU can issue at HBM rate are VALU-bound by
# construction; for those the HBM percentage is information, not the target (rocprofv3 issue-slot figures: profiles/r02_op_*.txt)
BOUND = [("filter2D 7x7 f32", "VALU (49 dependent fmaf per sample)"), ("sigma=1.5", "VALU (14 fmaf per sample)"),
         ("cornerHarris blockSize 3", "VALU / stores"), ("blockSize 3", "VALU (46 instr/px: 95 % of issue slots)"), ("Harris pipeline", "VALU 70-80 % busy + the launch's ragged end (DESIGN 6.3)"), ("cornerHarris", "stores (whole lines, non-temporal: r06_harris_resp_stores.txt)"),
         ("warpAffine + resize", "HBM read stream: every line of the source holds taps, 3.2 GB of unique lines at ~5.4 TB/s (staging alone 0.58-0.60 ms; DESIGN 6.2)"), ("warpAffine bilinear f32", "HBM (LDS-staged f32 patch, no conversions)"), ("(rot 7deg) on a GRAY", "per-workgroup set-up + stores (four frames per LDS pass; round 4: border tiles staged too, ablation in r04_warp_store_wait_ablation.txt)"), ("warpAffine", "VALU 63-80 % at a 1.95-2.15 GHz clock + 6.8 GB of real traffic at 4.5 TB/s (LDS-staged taps; DESIGN 0.3 item 6)"),
         ("rectangle", "launch latency"), ("text blend", "launch latency"), ("batch=1", "launch latency (L3-resident)"),
         ("640x480", "launch latency"), ("resize 8K -> 1080p", "HBM (line granularity: 56 MB/frame must be fetched for 31 MB used)")]


def bound_of(r):
    for key, b in BOUND:
        if key in r["op"] or key in r["config"]:
            return b
    return "HBM"


out = ["| op | config | frames | ms / launch | Gpix/s | algorithmic TB/s | % of 8 TB/s | % of the VALU roofline (instr/px @ measured clock) | bound | CPU oracle Mpix/s (cores) |", "|---|---|---|---|---|---|---|---|---|---|"]
for r in d["rows"]:
    cpu = r.get("cpu")
    cpu_s = f"{cpu['mpix_s']:.0f} ({cpu['cores']})" if cpu else ""
    hbm = f"{r['alg_gb_s'] / 1e3:.2f} | {r['frac_hbm_peak'] * 100:.1f}" if r["alg_bytes_per_px"] else "– | –"
    valu = f"{r['frac_valu_bound'] * 100:.0f} ({r['valu_instr_per_px']:g} @ {r['shader_mhz'] / 1e3:.2f} GHz)" if "frac_valu_bound" in r else ""
    out.append(f"| {r['op']} | {r['config']} | {r['frames']} | {r['ms_per_launch']:.4f} | {r['mpix_s'] / 1e3:.0f} | {hbm} | {valu} | {bound_of(r)} | {cpu_s} |")
text = "\n".join(out) + "\n"
sys.stdout.write(text)
if len(sys.argv) <= 1:
    with open(os.path.join(ROOT, "profiles", "ops_table.md"), "w") as f:
        f.write("One MI355X, device-resident batches, sustained clocks (`python tools/bench_ops.py --cpu`; raw data: `ops_bench.json`).\n"
                "Algorithmic bytes per pixel as DESIGN_HISTORY.md section 4 states them; launch-bound rows carry no roofline figure.  VALU roofline: the\n"
                "op's VALU instructions per pixel (packed FMAs count once) x pixels / (1024 SIMDs x 16 lanes per cycle x the shader clock\n"
                "measured under that op) / measured time -- min(HBM, VALU) is the bound of a row.\n\n" + text)
