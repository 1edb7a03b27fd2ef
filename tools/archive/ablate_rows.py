#!/usr/bin/env python3
"""tools/ablate_rows.py -- A/B timings of the north-star launch (64 x 4K BGR, 7x7 i8) inside ONE process on ONE box: the
row-streaming MFMA kernel against the round-1 strip kernel, its tuning knobs (waves per CU, bands per slot, row pairs in
flight) and -- on a profiling build, `make -C rustcv_amd/csrc EXTRA=-DRCV_ABLATE` -- its ablations (no stores / no loads / no
MFMA ...), next to plain device copies of the same bytes.  Every variant is timed three times in rotation (box-to-box and
minute-to-minute drift is several per cent on this pool; only same-run medians compare).  Writes gpurun_out/ablate_rows.json.

    python tools/ablate_rows.py [--ablate] [--copies]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402

L = _ffi.lib()
BL = _ffi.bench_lib()   # copy / store / clock probes: librustcv_hip_bench.so, not part of the product library
KNOBS = ("RCV_F7_ROWS", "RCV_FR_WPC", "RCV_FR_ROUNDS", "RCV_FR_PP", "RCV_FR_ORDER", "RCV_FR_BPF")


def setenv(env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    L.rcv__debug_reload_knobs()


def timeit(ctx, fn, steps=150, settle_ms=80.0):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(8):
            fn()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(steps):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ablate", action="store_true", help="profiling build: add the no-stores / no-loads / no-MFMA variants")
    ap.add_argument("--copies", action="store_true", help="add plain device copies / reads / writes of the same bytes")
    ap.add_argument("--fine", action="store_true", help="finer sweep of the bands-per-slot knob")
    ap.add_argument("--rounds", action="store_true", help="sweep of the band height (rounds of bands per wave slot)")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    from bench import bench_kernel7
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    k = bench_kernel7()
    alg = n * rows * cols * 6
    nbytes = n * rows * cols * 3
    flt = lambda: device.filter2d(src, dst, k, shift=6)   # noqa: E731
    variants = [("strip kernel (round 1)", {"RCV_F7_ROWS": 0}, 0, flt), ("row-streaming kernel (default: 10 waves/CU, 8 rounds, PP=3)", {}, 0, flt)]
    variants.append(("rows, bands dealt round-robin to the XCDs", {"RCV_FR_ORDER": 1}, 0, flt))
    for wpc in (8, 12):
        variants.append((f"rows {wpc} waves/CU", {"RCV_FR_WPC": wpc}, 0, flt))
    for r in (1, 4, 16):
        variants.append((f"rows rounds={r}", {"RCV_FR_ROUNDS": r}, 0, flt))
    if "--fine" in sys.argv:
        for b in (10, 12, 14, 15, 16, 18, 20, 21, 22, 24, 27, 30, 32, 36, 40, 43, 45, 48, 54, 60):
            variants.append((f"rows, {b} bands per frame ({2160 / b:.1f} rows)", {"RCV_FR_BPF": b}, 0, flt))
    if a.rounds:
        for r in (12, 16, 20, 24, 32, 40, 48, 64, 96, 128):
            variants.append((f"rows rounds={r}", {"RCV_FR_ROUNDS": r}, 0, flt))
    if a.ablate:
        for pp in (2, 4):
            variants.append((f"rows PP={pp}", {"RCV_FR_PP": pp}, 0, flt))
        for flags, nm in ((8, "plain stores"), (1, "no stores"), (2, "no loads"), (3, "compute only"), (4, "no MFMA"), (5, "loads only"), (6, "stores only"),
                          (16, "lane-ordered stores (bpermute)"), (18, "lane-ordered stores only"),
                          (32, "exclusive loads (24 B per lane; wrong data)"), (96, "exclusive nt loads"), (128, "no halo rows (wrong data)"),
                          (224, "exclusive nt loads, no halo rows"), (36, "no MFMA, exclusive loads"), (100, "no MFMA, exclusive nt loads"),
                          (132, "no MFMA, no halo rows"), (228, "no MFMA, exclusive nt loads, no halo rows"),
                          (37, "loads only, exclusive"), (101, "loads only, exclusive nt"), (229, "loads only, exclusive nt, no halo")):
            variants.append((f"rows, {nm}", {}, flags, flt))
    if a.copies:
        names = {0: "hipMemcpy D2D", 1: "copy sweep", 2: "copy block-contiguous", 3: "copy sweep nt", 5: "copy block nt", 6: "read only", 7: "write only",
                 8: "copy XCD-local sweep", 9: "copy XCD-local sweep nt"}
        for variant, grid in ((0, 1), (1, 1024), (1, 2048), (2, 1024), (3, 512), (3, 2048), (5, 512), (5, 2048), (8, 1024), (8, 2048), (8, 4096), (9, 1024), (9, 2048), (9, 4096), (6, 2048), (7, 32768)):
            def cp(variant=variant, grid=grid):
                assert BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, variant, grid) == 0
            variants.append((f"{names[variant]} g={grid}", {}, 0, cp))
    res = {v[0]: [] for v in variants}
    for rep in range(3):
        for tag, env, flags, fn in variants:
            setenv(env)
            L.rcv__debug_set(flags)
            res[tag].append(timeit(ctx, fn, settle_ms=80.0 if rep else 150.0))
    L.rcv__debug_set(0)
    setenv({})
    for tag, v in res.items():
        ms = sorted(v)[1]
        print(f"{tag:60s} median {ms:.4f} ms  ({' '.join(f'{x:.4f}' for x in v)})  {alg / ms / 1e6:8.1f} GB/s alg.  frac {alg / ms / 1e6 / 8000:.4f}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ablate_rows.json"), "w"), indent=1)
    src.free()
    dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
