#!/usr/bin/env python3
"""tools/batch_cliffs.py -- (round 5) batch-size sweep of the 4K BGR 7x7 integer filter2D, n = 8 .. 72 frames per launch:
us per frame on the default plan (chained bands for every n since round 5) and with RCV_FR_CHAIN=0 (one band per wave), same process.
A step of more than 3 % between neighbouring n is a cliff.  Also the integer GaussianBlur 7x7 (two weight tables) at n = 64, both kernels.

    python tools/batch_cliffs.py [--ns 8,9,...] [--launches 30]
"""
import argparse, ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import bench_kernel7

ap = argparse.ArgumentParser()
ap.add_argument("--ns", default="8,12,15,16,17,24,31,32,33,40,47,48,49,56,60,62,63,64,65,66,68,72")
ap.add_argument("--launches", type=int, default=30)
a = ap.parse_args()
L = _ffi.lib(); ctx = rcv.Context(0)
ROWS, COLS = 2160, 3840
ns = [int(v) for v in a.ns.split(",") if v]
nmax = max(ns + [64])
src = device.DeviceBatch(ctx, nmax, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, nmax, ROWS, COLS, 3)
device.synth(src, 0, 0x5EED0003, 0)
k = bench_kernel7()

def timed(fn):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for _ in range(4): fn()
        ctx.sync()
    ms = C.c_float(); L.rcv_timer_start(ctx.handle)
    for _ in range(a.launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / a.launches

def knob(v):
    if v is None: os.environ.pop("RCV_FR_CHAIN", None)
    else: os.environ["RCV_FR_CHAIN"] = str(v)
    L.rcv__debug_reload_knobs()

print("4K BGR 7x7 filter2D, us per frame (frac of 8 TB/s at 6 B/px); chained = default plan, plain = RCV_FR_CHAIN=0")
prev = None
for n in ns:
    s, d = src.view(0, n), dst.view(0, n)
    row = []
    for kn in ((None, 0) if not os.environ.get('BC_ONLY') else ((None,) if os.environ['BC_ONLY'] == 'c' else (0,))):
        knob(kn)
        L.rcv__debug_kernels_reset()
        t = statistics.median(timed(lambda: device.filter2d(s, d, k, shift=6)) for _ in range(3))
        row.append((t, L.rcv__debug_kernels().decode().split(";")[0][:28]))
    row = row * 2 if len(row) == 1 else row
    us = [r[0] * 1000 / n for r in row]
    step = "" if prev is None else f"  step vs previous n: {100 * (us[0] / prev - 1):+.1f} %"
    print(f"  n = {n:3d}   chained {us[0]:7.3f} us ({ROWS * COLS * 6 / us[0] / 1e3 / 8000:.3f})   plain {us[1]:7.3f} us ({ROWS * COLS * 6 / us[1] / 1e3 / 8000:.3f})   {row[0][1]} / {row[1][1]}{step}")
    prev = us[0]
s, d = src.view(0, 64), dst.view(0, 64)
for kn, name in ((None, "chained"), (0, "plain")):
    knob(kn)
    L.rcv__debug_kernels_reset()
    t = statistics.median(timed(lambda: device.gaussian_blur(s, d, 7, 0.0)) for _ in range(3))
    print(f"  GaussianBlur 7x7 int, 64 x 4K, {name:8s} {t:.4f} ms  frac {64 * ROWS * COLS * 6 / t / 1e6 / 8000:.4f}   {L.rcv__debug_kernels().decode().split(';')[0]}")
knob(None)
