#!/usr/bin/env python3
"""tools/small_filter_latency.py -- the integer filters on launches of a few frames: the strip kernel (latency variant) against the
row-streaming MFMA kernel with its per-SIMD band plan (RCV_F7_ROWS=1), microseconds per launch, back-to-back launches."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from tools.config2_latency import setenv  # noqa: E402

L = _ffi.lib()


def main():
    ctx = rcv.Context(0)
    r = np.random.default_rng(3)
    k7 = r.integers(-8, 9, size=(7, 7)).astype(np.int8)
    k5 = r.integers(-8, 9, size=(5, 5)).astype(np.int8)

    def t(fn, steps=600):
        for _ in range(100):
            fn()
        ctx.sync()
        ms = C.c_float()
        L.rcv_timer_start(ctx.handle)
        for _ in range(steps):
            fn()
        L.rcv_timer_stop(ctx.handle, C.byref(ms))
        return ms.value / steps * 1e3
    print(f"{'shape':8s} {'n':>2s}  {'op':28s} {'default':>9s} {'strip':>9s} {'rows':>9s}   kernel taken by default")
    for rows, cols, tag in ((1080, 1920, "1080p"), (2160, 3840, "4K")):
        for n in (1, 2, 4, 8, 16):
            bgr, bgr2 = device.DeviceBatch(ctx, n, rows, cols, 3), device.DeviceBatch(ctx, n, rows, cols, 3)
            gray, gray2 = device.DeviceBatch(ctx, n, rows, cols, 1), device.DeviceBatch(ctx, n, rows, cols, 1)
            yuyv = device.DeviceBatch(ctx, n, rows, cols, 2)
            dx, dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
            device.synth(bgr, 1, 5, 0)
            device.synth(gray, 1, 6, 0)
            device.synth(yuyv, 2, 7, 0)
            ops = [("filter2D 7x7 BGR", lambda: device.filter2d(bgr, bgr2, k7, shift=6)), ("filter2D 5x5 BGR", lambda: device.filter2d(bgr, bgr2, k5, shift=6)),
                   ("GaussianBlur 5x5 BGR", lambda: device.gaussian_blur(bgr, bgr2, 5, 0.0)), ("GaussianBlur 7x7 BGR (2 tables)", lambda: device.gaussian_blur(bgr, bgr2, 7, 0.0)),
                   ("filter2D 7x7 gray", lambda: device.filter2d(gray, gray2, k7, shift=6)), ("fused YUYV -> filter 7x7", lambda: device.filter2d_yuyv(yuyv, bgr2, k7, shift=6)),
                   ("fused filter -> gray -> Sobel", lambda: device.filter2d_sobel(bgr, dx, dy, k7, shift=6))]
            for name, fn in ops:
                res = []
                for env in ({}, {"RCV_F7_ROWS": 0, "RCV_GAUSS_ROWS": 0}, {"RCV_F7_ROWS": 1, "RCV_GAUSS_ROWS": 0}):
                    setenv(env)
                    if not env:
                        L.rcv__debug_kernels_reset()
                        fn()
                        kn = L.rcv__debug_kernels().decode()
                    res.append(sorted(t(fn) for _ in range(3))[1])
                print(f"{tag:8s} {n:2d}  {name:28s} {res[0]:9.2f} {res[1]:9.2f} {res[2]:9.2f}   {kn[:50]}", flush=True)
            setenv({})
            for b in (bgr, bgr2, gray, gray2, yuyv, dx, dy):
                b.free()
    ctx.close()


if __name__ == "__main__":
    main()
