#!/usr/bin/env python3
"""tools/ab_3f.py -- one launch of config 3 whole ("3f": 7x7 filter2D -> gray -> Sobel, 64 x 4K) through the product entry of THIS tree: ms per launch,
medians of 5 x 40 launches.  Run it from two checkouts on one box for an A/B of library versions (the round-6 group-of-four layout against the
240-px strips of rounds 3-5: profiles/r06_3f_ab.txt)."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from rustcv_amd._ffi import RCV_16S
from bench import bench_kernel7
L = _ffi.lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
dx = device.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_16S); dy = device.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_16S)
device.synth(src, 0, 0x5EED0003, 0)
k = bench_kernel7()
def timed(fn, launches=40):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.05:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
for name, fn in (("3f", lambda: device.filter2d_sobel(src, dx, dy, k, 6)), ("3s", lambda: device.sobel(src, dx, dy))):
    v = [timed(fn) for _ in range(5)]
    m = statistics.median(v)
    print(f"  {os.path.basename(ROOT):10s} {name}  {m:.4f} ms  frac at 7 B/px {n * ROWS * COLS * 7 / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in v]}", flush=True)
