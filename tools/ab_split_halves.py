#!/usr/bin/env python3
"""tools/ab_split_halves.py -- (round 6) would a 64-frame call gain from running as two 32-frame launches on two streams that are never joined per call?  Same
process, SAME buffers: (a) one context, 64-frame launches back to back; (b) two contexts of the device, each filtering one half of the same source / destination
batches (frames 0-31 / 32-63), enqueued alternately without any synchronisation between calls (what a lazily joined pair of internal streams would do); (c) the
same with four quarters on four contexts.  Wall time per 64 frames over 100 calls, rotations, medians.  (Round 3 measured the halves with a fork / join per call:
5 % SLOWER.)"""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi, device, multigpu
from bench import bench_kernel7
L = _ffi.lib()
n, ROWS, COLS = 64, 2160, 3840
g = multigpu.NativeGroup.in_flight(0, 4)
ctxs = g.ctxs
src = device.DeviceBatch(ctxs[0], n, ROWS, COLS, 3); dst = device.DeviceBatch(ctxs[0], n, ROWS, COLS, 3)
device.synth(src, 0, 0x5EED0003, 0)
k = bench_kernel7(); kp = k.ctypes.data_as(C.POINTER(C.c_int8))
def part(ctx, first, cnt):
    bs, bd = src.view(first, cnt).as_rcv(), dst.view(first, cnt).as_rcv()
    h = ctx.handle
    def f():
        rc = L.rcv_filter2d_i8_batch(h, C.byref(bs), C.byref(bd), kp, 7, 6)
        assert rc == 0, rc
    return f
whole = [part(ctxs[0], 0, 64)]
halves = [part(ctxs[0], 0, 32), part(ctxs[1], 32, 32)]
quarters = [part(ctxs[i], 16 * i, 16) for i in range(4)]
def run(parts, calls=100):
    def call():
        for p in parts: p()
    t = time.perf_counter()
    while time.perf_counter() - t < 0.08:
        for _ in range(8): call()
        g.sync()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(calls): call()
    g.sync()
    return (time.perf_counter() - t0) * 1e3 / calls
res = {}
for r in range(5):
    for name, parts in (("one 64-frame launch per call", whole), ("two 32-frame halves on two streams, never joined", halves), ("four 16-frame quarters on four streams", quarters)):
        res.setdefault(name, []).append(run(parts))
base = statistics.median(res["one 64-frame launch per call"])
for name, v in res.items():
    m = statistics.median(v)
    print(f"  {name:52s} {m:.4f} ms per 64 frames  frac {n * ROWS * COLS * 6 / m / 1e6 / 8000:.4f}  {100 * (m / base - 1):+.2f} %   {['%.4f' % x for x in v]}", flush=True)
