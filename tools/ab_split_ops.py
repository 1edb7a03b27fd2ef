#!/usr/bin/env python3
"""tools/ab_split_ops.py -- (round 6) every BASELINE batch op on ONE context with its call run as two halves on the context's two streams (default) and as one
launch (RCV_FR_SPLIT=0): same process, same buffers, alternating; HIP events on the context's stream around the calls (rcv_timer_stop comes through
rcv_bind: the stop event waits for the half stream), five rotations, medians.  Ops = bench.py's lanes: 3 (filter2D), 3s (Sobel of BGR), 3f (filter -> gray ->
Sobel), 4 (warp -> resize, 32 x 8K), 5 (Harris pipeline).   usage: ab_split_ops.py [cfg ...]"""
import argparse, ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi
import bench
L = _ffi.lib()
ctx = rcv.Context(0)
a = bench.parse([])
cfgs = [(int(x) if x.isdigit() else x) for x in (sys.argv[1:] or ["3", "3s", "3f", "4", "5"])]
def timed(fn, launches):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.06:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
for cfg in cfgs:
    c = bench.CONFIGS[cfg]
    ln = bench.Lane(a, cfg, ctx, 0, n=c["batch"])
    res = {}
    for r in range(5):
        for name, v in (("one launch per call (RCV_FR_SPLIT=0)", "0"), ("two halves on two streams (default)", "-1")):
            os.environ["RCV_FR_SPLIT"] = v
            L.rcv__debug_reload_knobs()
            res.setdefault(name, []).append(timed(ln.step, 60))
    base = statistics.median(res["one launch per call (RCV_FR_SPLIT=0)"])
    for name, v in res.items():
        m = statistics.median(v)
        print(f"  config {str(cfg):3s} {name:40s} {m:.4f} ms  frac {c['batch'] * c['alg_bytes'] / m / 1e6 / 8000:.4f}  {100 * (m / base - 1):+.2f} %   {['%.4f' % x for x in v]}", flush=True)
    ln.free()
