#!/usr/bin/env python3
"""tools/ablate_chain_halo.py -- (round 5) what the chained-band kernel's halo rows cost: 64 x 4K BGR 7x7, the filter and its memory-only
variant with and without the 6 halo rows of every band (without: the output is not a filtered image), band heights 16 .. 64."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import bench_kernel7
from tools._rows import Rows
L = _ffi.lib(); _ffi.bench_lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
device.synth(src, 0, 0x5EED0003, 0)
rows = Rows(ctx, src, dst, bench_kernel7())
def timed(fn, launches=60):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.04:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
res = {}
for r in range(3):
    for hgt in (16, 32, 64):
        for name, dbg in (("filter", 0), ("filter, no halo rows", 128), ("memory only", 4), ("memory only, no halo rows", 132)):
            res.setdefault((hgt, name), []).append(timed(rows.fn(chain=1, chain_rows=hgt, dbg=dbg)))
nbytes = n * ROWS * COLS * 3
for (hgt, name), v in res.items():
    m = statistics.median(v)
    print(f"  chained {hgt:3d} rows  {name:28s} {m:.4f} ms  {2 * nbytes / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in v]}")
