#!/usr/bin/env python3
"""tools/cliffs.py -- every op on an aligned 4K batch and on odd / unaligned variants of the same size (3839x2159, 3838x2160, 3836x2160), with
the kernel each call dispatched: a survey of where a shape falls from a streaming kernel to a per-sample one.  Run on a GPU box."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
L = _ffi.lib()
ctx = rcv.Context(0)
def t(fn, steps=10):
    for _ in range(3): fn()
    ctx.sync()
    ms = C.c_float(); L.rcv_timer_start(ctx.handle)
    for _ in range(steps): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / steps
def kern(fn):
    L.rcv__debug_kernels_reset(); fn(); ctx.sync(); return L.rcv__debug_kernels().decode()[:90]
n = 16
k7 = (np.arange(49, dtype=np.int8).reshape(7, 7) % 9 - 4).astype(np.int8)
kf = np.full((5, 5), 1 / 25, np.float32)
M = np.array([0.9925, -0.1219, 300.0, 0.1219, 0.9925, -200.0], np.float32)
for (rows, cols) in ((2160, 3840), (2159, 3839), (2160, 3838), (2160, 3836)):
    B = lambda ch, depth=_ffi.RCV_8U, r=rows, c=cols: device.DeviceBatch(ctx, n, r, c, ch, depth)
    bgr, bgr2, gray, gray2 = B(3), B(3), B(1), B(1)
    dx, dy, resp, mask = B(1, _ffi.RCV_16S), B(1, _ffi.RCV_16S), B(1, _ffi.RCV_32F), B(1)
    half = device.DeviceBatch(ctx, n, rows // 2, cols // 2, 3)
    bgra = B(4)
    odd = device.DeviceBatch(ctx, n, 1441, 2561, 3)
    device.synth(bgr, 1, 7, 0); device.synth(gray, 1, 8, 0)
    ops = [("filter2D 7x7 i8", lambda: device.filter2d(bgr, bgr2, k7, shift=6)), ("Gaussian 5x5 int", lambda: device.gaussian_blur(bgr, bgr2, 5, 0.0)),
           ("Gaussian 7 sigma 1.5", lambda: device.gaussian_blur(bgr, bgr2, 7, 1.5)), ("filter2D 5x5 f32", lambda: device.filter2d(bgr, bgr2, kf)),
           ("bgr2gray", lambda: device.cvt_color(bgr, gray2, _ffi.RCV_BGR2GRAY)), ("Sobel gray", lambda: device.sobel(gray, dx, dy)),
           ("Harris pipeline b2", lambda: device.harris_pipeline(bgr, mask, None, 2, 0.04, 1e-4)), ("Harris pipeline b3", lambda: device.harris_pipeline(bgr, mask, None, 3, 0.04, 1e-4)),
           ("cornerHarris b2", lambda: device.corner_harris(gray, resp, 2, 0.04)), ("NMS", lambda: device.nms3x3(resp, mask, 1e-4)),
           ("warpAffine", lambda: device.warp_affine(bgr, bgr2, M)), ("warp gray", lambda: device.warp_affine(gray, gray2, M)),
           ("resize to half", lambda: device.resize(bgr, half)), ("resize to 2561x1441", lambda: device.resize(bgr, odd)),
           ("gray filter 7x7", lambda: device.filter2d(gray, gray2, k7, shift=6)),
           ("BGRA -> BGR", lambda: device.cvt_color(bgra, bgr2, _ffi.RCV_BGRA2BGR)), ("BGR -> RGB", lambda: device.cvt_color(bgr, bgr2, _ffi.RCV_BGR2RGB)),
           ("BGR -> BGRX", lambda: device.cvt_color(bgr, bgra, _ffi.RCV_BGR2BGRX)),
           ("Sobel of BGR", lambda: device.sobel(bgr, dx, dy)), ("filter2D 7x7 -> gray -> Sobel", lambda: device.filter2d_sobel(bgr, dx, dy, k7, shift=6)), ("fused warp + 2x down-scale", lambda: device.warp_affine_resize(bgr, half, M, 2 * (rows // 2), 2 * (cols // 2)))]
    print(f"---- {cols} x {rows}, {n} frames")
    for name, fn in ops:
        print(f"{name:24s} {t(fn):8.3f} ms   {kern(fn)}", flush=True)
    for b in (bgr, bgr2, gray, gray2, dx, dy, resp, mask, half, odd, bgra): b.free()
