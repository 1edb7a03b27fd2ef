#!/usr/bin/env python3
"""tools/ab_split_product.py -- (round 6) the product's filter2D call of 64 x 4K BGR frames on ONE context with the call run as two halves on the context's two
streams (default) and as one launch (RCV_FR_SPLIT=0), same process, same buffers, alternating; HIP events on the context's stream around 100 calls (the stop
event waits for the half stream: rcv_timer_stop comes through rcv_bind), five rotations, medians."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import bench_kernel7
L = _ffi.lib()
n, ROWS, COLS = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 2160, 3840
ctx = rcv.Context(0)
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
device.synth(src, 0, 0x5EED0003, 0)
k = bench_kernel7()
def timed(launches=100):
    fn = lambda: device.filter2d(src, dst, k, shift=6)
    t = time.perf_counter()
    while time.perf_counter() - t < 0.06:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
res = {}
for r in range(5):
    for name, v in (("one launch per call (RCV_FR_SPLIT=0)", "0"), ("two halves on two streams (default)", "-1")):
        os.environ["RCV_FR_SPLIT"] = v
        L.rcv__debug_reload_knobs()
        res.setdefault(name, []).append(timed())
base = statistics.median(res["one launch per call (RCV_FR_SPLIT=0)"])
for name, v in res.items():
    m = statistics.median(v)
    print(f"  {n} frames  {name:40s} {m:.4f} ms per call  frac {n * ROWS * COLS * 6 / m / 1e6 / 8000:.4f}  {100 * (m / base - 1):+.2f} %   {['%.4f' % x for x in v]}", flush=True)
