#!/usr/bin/env python3
"""tools/ablate_warp.py -- same-run A/B of the warpAffine kernel on config 4's batch (32 x 8K BGR, rotate 7 degrees): tile order
(RCV_XCD_ORDER), frames per workgroup (RCV_WARP_FPG) and the identity / small-angle maps for comparison.  Three rotations, medians."""
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
KNOBS = ("RCV_XCD_ORDER", "RCV_WARP_FPG", "RCV_WARP_LDS")


def setenv(env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    L.rcv__debug_reload_knobs()


def timeit(ctx, fn, steps=40, settle_ms=60.0):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(4):
            fn()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(steps):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / steps


def rot(deg, cx, cy, tx, ty):
    t = np.deg2rad(deg)
    c, s = np.cos(t), np.sin(t)
    return np.array([c, -s, cx - c * cx + s * cy + tx, s, c, cy - s * cx - c * cy + ty], np.float32)


def main():
    ctx = rcv.Context(0)
    n, rows, cols = 32, 4320, 7680
    src, dst = device.DeviceBatch(ctx, n, rows, cols, 3), device.DeviceBatch(ctx, n, rows, cols, 3)
    small = device.DeviceBatch(ctx, n, 1080, 1920, 3)
    gsrc, gdst = device.DeviceBatch(ctx, n, rows, cols, 1), device.DeviceBatch(ctx, n, rows, cols, 1)
    device.synth(src, 1, 4, 0)
    device.synth(gsrc, 1, 5, 0)
    M7 = rot(7.0, cols / 2, rows / 2, 13.25, -8.5)
    M1 = rot(1.0, cols / 2, rows / 2, 13.25, -8.5)
    M45 = rot(45.0, cols / 2, rows / 2, 0.0, 0.0)
    Mid = np.array([1, 0, 0.3, 0, 1, 0.7], np.float32)
    w = lambda M: (lambda: device.warp_affine(src, dst, M))   # noqa: E731
    variants = [("warp 7 deg (default)", {}, w(M7))]
    variants += [("warp 7 deg, gather kernel (no LDS staging)", {"RCV_WARP_LDS": 0}, w(M7)), ("warp 7 deg, plain tile order", {"RCV_XCD_ORDER": 0}, w(M7))]
    for f in (1, 2, 4, 8, 16):
        variants.append((f"warp 7 deg, {f} frames per workgroup", {"RCV_WARP_FPG": f}, w(M7)))
        variants.append((f"warp 7 deg, {f} frames per workgroup, gather kernel", {"RCV_WARP_FPG": f, "RCV_WARP_LDS": 0}, w(M7)))
    variants += [("warp translation (0.3, 0.7)", {}, w(Mid)), ("warp translation, gather kernel", {"RCV_WARP_LDS": 0}, w(Mid)),
                 ("warp 1 deg", {}, w(M1)), ("warp 1 deg, gather kernel", {"RCV_WARP_LDS": 0}, w(M1)),
                 ("warp 45 deg", {}, w(M45)), ("warp 45 deg, gather kernel", {"RCV_WARP_LDS": 0}, w(M45)), ("warp 45 deg, plain tile order", {"RCV_XCD_ORDER": 0}, w(M45)),
                 ("warp 90 deg", {}, w(rot(90.0, cols / 2, rows / 2, 0.0, 0.0))), ("warp 90 deg, gather kernel", {"RCV_WARP_LDS": 0}, w(rot(90.0, cols / 2, rows / 2, 0.0, 0.0))),
                 ("warp translation (13.25, -8.5)", {}, w(np.array([1, 0, 13.25, 0, 1, -8.5], np.float32))),
                 ("warp translation (100.5, 0)", {}, w(np.array([1, 0, 100.5, 0, 1, 0], np.float32))),
                 ("warp translation (0, 100.5)", {}, w(np.array([1, 0, 0, 0, 1, 100.5], np.float32))),
                 ("warp 0.1 deg", {}, w(rot(0.1, cols / 2, rows / 2, 0.0, 0.0))),
                 ("warp scale 0.9", {}, w(np.array([0.9, 0, 10, 0, 0.9, 10], np.float32))),
                 ("warp scale 1.1", {}, w(np.array([1.1, 0, 10, 0, 1.1, 10], np.float32))),
                 ("warp shear x", {}, w(np.array([1, 0.2, 0, 0, 1, 0], np.float32))),
                 ("warp gray 7 deg", {}, lambda: device.warp_affine(gsrc, gdst, M7)),
                 ("fused warp + 4x down-scale", {}, lambda: device.warp_affine_resize(src, small, M7, rows, cols)),
                 ]
    if "--quick" in sys.argv:
        variants = [variants[0], [v for v in variants if v[0].startswith("warp translation")][0]]
    res = {v[0]: [] for v in variants}
    for rep in range(3):
        for tag, env, fn in variants:
            setenv(env)
            res[tag].append(timeit(ctx, fn, settle_ms=50.0 if rep else 100.0))
    setenv({})
    alg = n * rows * cols * 6
    for tag, v in res.items():
        ms = sorted(v)[1]
        print(f"{tag:58s} median {ms:.4f} ms  ({' '.join(f'{x:.4f}' for x in v)})  {alg / ms / 1e6:8.1f} GB/s  {alg / ms / 1e6 / 80:.1f} %", flush=True)
    for b in (src, dst, small, gsrc, gdst):
        b.free()
    ctx.close()


if __name__ == "__main__":
    main()
