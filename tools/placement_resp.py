#!/usr/bin/env python3
"""tools/placement_resp.py -- (round 6) the launches that store the f32 response differ by up to 12 % from process to process (profiles/r06_harris_resp_stores.txt).
Does WHERE the buffers lie decide it?  One process, cornerHarris gray -> f32 (64 x 4K) on freshly allocated batches: (a) free + allocate again, (b) with a pad of
p x 2 MiB allocated between the gray and the response batch, p = 0 .. 40 (moves the response relative to the source in 2-MiB steps), (c) with pads of odd sizes.
Printed: device addresses and ms per launch."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from rustcv_amd._ffi import RCV_32F
L = _ffi.lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
def timed(fn, launches=60):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.05:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
def one(pad_bytes, label):
    gray = device.DeviceBatch(ctx, n, ROWS, COLS, 1)
    pad = device.DeviceBatch(ctx, 1, 1, pad_bytes, 1) if pad_bytes else None
    resp = device.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_32F)
    device.synth(gray, 1, 0x5EED0005, 0)
    v = [timed(lambda: device.corner_harris(gray, resp, 2, 0.04)) for _ in range(3)]
    a, b = gray.ptr.value, resp.ptr.value
    print(f"  {label:26s} gray {a:#014x}  resp {b:#014x}  resp - gray = {(b - a) / (1 << 20):9.2f} MiB (mod 64 MiB: {((b - a) % (1 << 26)) / (1 << 20):6.2f})   {statistics.median(v):.4f} ms   {['%.4f' % x for x in v]}", flush=True)
    gray.free(); resp.free()
    if pad: pad.free()
print("free + allocate again")
for it in range(4): one(0, "no pad")
print("pads of p x 2 MiB between the two batches")
for p in list(range(0, 20)) + [24, 32, 40, 48, 64, 96, 128]: one(p << 21, f"pad {2 * p} MiB")
print("pads of odd sizes")
for it in range(6): one((37 + 61 * it) * 1024 * 1024 + 4096 * (it + 1), f"pad {(37 + 61 * it)} MiB + {4 * (it + 1)} KiB")
