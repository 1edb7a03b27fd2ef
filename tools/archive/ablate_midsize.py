#!/usr/bin/env python3
"""tools/ablate_midsize.py -- launches between the one-wave-per-SIMD plans and the fill-the-GPU plans: 8 and 16 frames of 4K through the
register-window kernels with their segment-height knobs swept (us per frame; 0 = the library's own plan)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402

L = _ffi.lib()
ctx = rcv.Context(0)
rows, cols = 2160, 3840


def timed(fn, launches=80):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.06:
        for _ in range(8):
            fn()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(launches):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / launches


def sweep(name, knob, values, fn, n):
    out = []
    for v in values:
        os.environ.pop(knob, None)
        if v:
            os.environ[knob] = str(v)
        L.rcv__debug_reload_knobs()
        out.append(timed(fn) * 1e3 / n)
    os.environ.pop(knob, None)
    L.rcv__debug_reload_knobs()
    print(f"{name:22s} n={n:2d}  " + "  ".join(f"{v}:{t:5.2f}" for v, t in zip(values, out)), flush=True)


for n in (4, 8, 16, 32):
    bgr = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(bgr, 1, 7, 0)
    gray = device.DeviceBatch(ctx, n, rows, cols, 1)
    device.synth(gray, 1, 8, 0)
    dx, dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    resp = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F)
    mask = device.DeviceBatch(ctx, n, rows, cols, 1)
    sweep("Sobel gray", "RCV_SOBEL_SEG", (0, 16, 24, 32, 48, 64, 96), lambda: device.sobel(gray, dx, dy), n)
    sweep("Sobel of BGR", "RCV_SOBEL_SEG", (0, 16, 24, 32, 48, 64, 96), lambda: device.sobel(bgr, dx, dy), n)
    sweep("cornerHarris", "RCV_HARRIS_SEG_ROWS", (0, 32, 48, 64, 96, 128, 192, 270), lambda: device.corner_harris(gray, resp, 2, 0.04), n)
    sweep("NMS", "RCV_NMS_SEG", (0, 16, 24, 32, 48, 64, 96), lambda: device.nms3x3(resp, mask, 1e-4), n)
    sweep("Harris pipeline", "RCV_HARRIS_SEG_ROWS", (0, 32, 48, 64, 96, 128, 192, 270), lambda: device.harris_pipeline(bgr, mask, None, 2, 0.04, 1e-4), n)
    for b in (bgr, gray, dx, dy, resp, mask):
        b.free()
