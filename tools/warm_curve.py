#!/usr/bin/env python3
"""tools/warm_curve.py -- (round 6) does the box speed up under sustained load?  The north-star launch (64 x 4K BGR 7x7, one stream) back to back
for `seconds` (default 60), ms per launch of every window of 100 launches against the time since the first launch, with the shader clock
(rcv__clock_probe) every few seconds; then 10 s idle and 20 s more.  One line per window group."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import bench_kernel7
L = _ffi.lib(); BL = _ffi.bench_lib()
n, ROWS, COLS = 64, 2160, 3840
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = rcv.Context(0)
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
device.synth(src, 0, 0x5EED0003, 0)
k = bench_kernel7()
def window(launches=100):
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): device.filter2d(src, dst, k, shift=6)
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
def run(total, label):
    t0 = time.perf_counter(); nxt = 0.0; acc = []
    while True:
        t = time.perf_counter() - t0
        if t >= total: break
        acc.append(window())
        if t >= nxt:
            print(f"  {label} t = {t:5.1f} s   {statistics.median(acc):.4f} ms per launch  frac {n * ROWS * COLS * 6 / statistics.median(acc) / 1e6 / 8000:.4f}   (min {min(acc):.4f}, {len(acc)} windows)", flush=True)
            acc = []; nxt += 1.0 if t < 10 else 5.0
run(secs, "load")
time.sleep(10.0)
run(20.0, "after 10 s idle")
