#!/usr/bin/env python3
"""tools/ablate_ops.py -- same-run A/B of the register-window kernels' block order (RCV_XCD_ORDER) and, on a profiling build,
of their store flavour: Sobel, Harris pipeline, NMS on 64 x 4K.  Every variant three times in rotation, medians."""
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


def setenv(env):
    for k in ("RCV_XCD_ORDER", "RCV_HARRIS_SEG_ROWS", "RCV_SOBEL_WGS"):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    L.rcv__debug_reload_knobs()


def timeit(ctx, fn, steps=100, settle_ms=60.0):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(8):
            fn()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(steps):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / steps


def main():
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    gray = device.DeviceBatch(ctx, n, rows, cols, 1)
    bgr = device.DeviceBatch(ctx, n, rows, cols, 3)
    dx, dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    mask = device.DeviceBatch(ctx, n, rows, cols, 1)
    resp = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F)
    device.synth(gray, 1, 3, 0)
    device.synth(bgr, 1, 5, 0)
    device.corner_harris(gray, resp, 2, 0.04)
    ops = [("Sobel gray", lambda: device.sobel(gray, dx, dy), 5), ("Sobel of BGR", lambda: device.sobel(bgr, dx, dy), 7),
           ("Harris pipeline BGR", lambda: device.harris_pipeline(bgr, mask, None, 2, 0.04, 1e-4), 4),
           ("NMS 3x3", lambda: device.nms3x3(resp, mask, 1e-4), 5), ("BGR2GRAY", lambda: device.cvt_color(bgr, gray, _ffi.RCV_BGR2GRAY), 4)]
    variants = []
    for name, fn, bpp in ops:
        variants.append((name + " | XCD-contiguous (default)", {}, 0, fn, bpp))
        variants.append((name + " | plain block order", {"RCV_XCD_ORDER": 0}, 0, fn, bpp))
    for w in (2, 3, 4, 5):
        variants.append((f"Sobel gray | {w} workgroups per CU", {"RCV_SOBEL_WGS": w}, 0, ops[0][1], 5))
        variants.append((f"Sobel of BGR | {w} workgroups per CU", {"RCV_SOBEL_WGS": w}, 0, ops[1][1], 7))
    if "--ablate" in sys.argv:
        variants.append(("Sobel gray | plain stores", {}, 4, ops[0][1], 5))
    res = {v[0]: [] for v in variants}
    for rep in range(3):
        for tag, env, flags, fn, bpp in variants:
            setenv(env)
            L.rcv__debug_set(flags)
            res[tag].append(timeit(ctx, fn))
    L.rcv__debug_set(0)
    setenv({})
    for tag, env, flags, fn, bpp in variants:
        ms = sorted(res[tag])[1]
        print(f"{tag:50s} median {ms:.4f} ms  ({' '.join(f'{x:.4f}' for x in res[tag])})  {n * rows * cols * bpp / ms / 1e6 / 8000 * 100:5.1f} % of 8 TB/s", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
