#!/usr/bin/env python3
"""tools/ab_in_flight.py -- (round 6) the north-star launch with 1, 2, 3 batches in flight (contexts of one device, own buffers each), SAME process, rotations:
ms per 64-frame call from the group timer (events on every context's stream).  With one context the call runs as two halves on that context's two streams
(RCV_FR_SPLIT default) or as one launch (=0)."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi, multigpu
import bench
L = _ffi.lib()
a = bench.parse([])
g = multigpu.NativeGroup.in_flight(0, 3)
lanes = [bench.Lane(a, 3, g.ctxs[j], 64 * j, n=64) for j in range(3)]
def run(F, calls=60):
    def step():
        for ln in lanes[:F]: ln.step()
    t = time.perf_counter()
    while time.perf_counter() - t < 0.06:
        for _ in range(8): step()
        g.sync()
    g.timer_start()
    for _ in range(calls): step()
    return g.timer_stop() / (calls * F)
res = {}
for r in range(5):
    for name, F, split in (("1 context, one launch per call", 1, "0"), ("1 context, call as two halves", 1, "-1"), ("2 batches in flight", 2, "-1"), ("3 batches in flight", 3, "-1")):
        os.environ["RCV_FR_SPLIT"] = split
        L.rcv__debug_reload_knobs()
        res.setdefault(name, []).append(run(F))
base = statistics.median(res["1 context, one launch per call"])
for name, v in res.items():
    m = statistics.median(v)
    print(f"  {name:34s} {m:.4f} ms per 64 frames  frac {64 * 2160 * 3840 * 6 / m / 1e6 / 8000:.4f}  {100 * (m / base - 1):+.2f} %   {['%.4f' % x for x in v]}", flush=True)
