#!/usr/bin/env python3
"""tools/placement_variance.py -- (round 6) does WHERE the batches lie decide the launch time?  One process, the north-star launch (64 x 4K BGR 7x7, one stream) on
freshly allocated source / destination batches, ten times: (a) free and allocate again (the allocator may hand back the same range), (b) keep a growing pile of
dummy allocations of odd sizes alive so that every pair lands somewhere else.  Printed: device addresses (mod 2 MiB / 1 GiB) and ms per launch."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import bench_kernel7
L = _ffi.lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
k = bench_kernel7()
def timed(fn, launches=100):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.08:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
pile = []
for mode in ("free + allocate again", "with a growing pile of odd-sized allocations in between"):
    print(mode)
    for it in range(8):
        if mode.startswith("with"):
            pile.append(device.DeviceBatch(ctx, 1, 1, (37 + 61 * it) * 1024 * 1024 + 4096 * (it + 1), 1))
        src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
        device.synth(src, 0, 0x5EED0003, 0)
        v = [timed(lambda: device.filter2d(src, dst, k, shift=6)) for _ in range(3)]
        m = statistics.median(v)
        a, b = src.ptr.value, dst.ptr.value
        print(f"  src {a:#014x} (mod 2M {a % (1 << 21):#08x}, mod 1G {a % (1 << 30):#010x})  dst {b:#014x} (mod 2M {b % (1 << 21):#08x})   {m:.4f} ms  frac {n * ROWS * COLS * 6 / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in v]}", flush=True)
        src.free(); dst.free()
