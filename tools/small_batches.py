#!/usr/bin/env python3
"""tools/small_batches.py -- every op on 1 / 2 / 4 / 8 / 16 / 64 frames of 3840x2160: microseconds per frame (a single frame is a latency
problem: one launch, every SIMD one wave; the last column is batch 1 against batch 64).  GPU box; output kept as profiles/r02_small_batches.txt."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
L = _ffi.lib(); ctx = rcv.Context(0)
def t(fn, steps=60):
    for _ in range(150): fn()
    ctx.sync()
    ms = C.c_float(); L.rcv_timer_start(ctx.handle)
    for _ in range(steps): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / steps
r = np.random.default_rng(3); k7 = r.integers(-8, 9, size=(7, 7)).astype(np.int8); k7[3, 3] = 40
th = np.deg2rad(7.0); c, sn = np.cos(th), np.sin(th); cx, cy = 1920.0, 1080.0
M = np.array([c, -sn, cx - c * cx + sn * cy, sn, c, cy - sn * cx - c * cy], np.float32)
rows, cols = 2160, 3840
res = {}
for n in (1, 2, 4, 8, 16, 64):
    bgr = device.DeviceBatch(ctx, n, rows, cols, 3); bgr2 = device.DeviceBatch(ctx, n, rows, cols, 3)
    gray = device.DeviceBatch(ctx, n, rows, cols, 1); gray2 = device.DeviceBatch(ctx, n, rows, cols, 1)
    dx = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S); dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    resp = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F); small = device.DeviceBatch(ctx, n, 540, 960, 3)
    device.synth(bgr, 1, 5, 0); device.synth(gray, 1, 3, 0)
    ops = [("filter7", lambda: device.filter2d(bgr, bgr2, k7, shift=6)), ("gauss5int", lambda: device.gaussian_blur(bgr, bgr2, 5, 0.0)),
           ("gauss5f32", lambda: device.gaussian_blur(bgr, bgr2, 5, 1.5)), ("bgr2gray", lambda: device.cvt_color(bgr, gray2, _ffi.RCV_BGR2GRAY)),
           ("sobel", lambda: device.sobel(gray, dx, dy)), ("sobelBGR", lambda: device.sobel(bgr, dx, dy)), ("filt>sobel", lambda: device.filter2d_sobel(bgr, dx, dy, k7, shift=6)),
           ("harris", lambda: device.harris_pipeline(bgr, gray2, None, 2, 0.04, 1e-4)), ("harris3", lambda: device.harris_pipeline(bgr, gray2, None, 3, 0.04, 1e-4)),
           ("cornerH", lambda: device.corner_harris(gray, resp, 2, 0.04)), ("nms", lambda: device.nms3x3(resp, gray2, 1e-4)),
           ("warp", lambda: device.warp_affine(bgr, bgr2, M)), ("warpgray", lambda: device.warp_affine(gray, gray2, M)),
           ("resize4x", lambda: device.resize(bgr, small)), ("grayfilt7", lambda: device.filter2d(gray, gray2, k7, shift=6))]
    for name, fn in ops:
        res.setdefault(name, []).append(t(fn) * 1e3 / n)
    for b in (bgr, bgr2, gray, gray2, dx, dy, resp, small): b.free()
print("us per 4K frame at batch     1      2      4      8     16     64")
for name, v in res.items():
    print("%-12s" % name, " ".join("%6.1f" % x for x in v), "  x%.2f" % (v[0] / v[-1]))
