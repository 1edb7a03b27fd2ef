#!/usr/bin/env python3
"""tools/ablate_segs.py -- rows per segment of the register-window kernels on 64 x 4K (Harris pipeline, cornerHarris, NMS): with the
XCD-contiguous block order shorter segments keep what one XCD has in flight more compact (the Sobel kernel gained 5 % from 68 -> 24
rows); every segment also streams its halo rows again.  Three rotations, medians."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from tools.ablate_sweep import timeit  # noqa: E402

L = _ffi.lib()
KN = ("RCV_HARRIS_SEG_ROWS", "RCV_NMS_SEG")


def setenv(env):
    for k in KN:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    L.rcv__debug_reload_knobs()


def main():
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    gray = device.DeviceBatch(ctx, n, rows, cols, 1)
    bgr = device.DeviceBatch(ctx, n, rows, cols, 3)
    mask = device.DeviceBatch(ctx, n, rows, cols, 1)
    resp = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F)
    device.synth(gray, 1, 3, 0)
    device.synth(bgr, 1, 5, 0)
    device.corner_harris(gray, resp, 2, 0.04)
    ops = [("Harris pipeline BGR", "RCV_HARRIS_SEG_ROWS", lambda: device.harris_pipeline(bgr, mask, None, 2, 0.04, 1e-4), 4),
           ("cornerHarris gray -> f32", "RCV_HARRIS_SEG_ROWS", lambda: device.corner_harris(gray, resp, 2, 0.04), 5),
           ("NMS 3x3", "RCV_NMS_SEG", lambda: device.nms3x3(resp, mask, 1e-4), 5)]
    variants = []
    for name, knob, fn, bpp in ops:
        variants.append((f"{name:26s} plan", {}, fn, bpp))
        for seg in ((6, 8, 10, 12, 16, 20, 24, 64) if knob == "RCV_NMS_SEG" else (90, 135)):
            variants.append((f"{name:26s} seg={seg}", {knob: seg}, fn, bpp))
    res = {v[0]: [] for v in variants}
    for rep in range(3):
        for tag, env, fn, bpp in variants:
            setenv(env)
            res[tag].append(timeit(ctx, fn, steps=50, settle_ms=40.0))
    setenv({})
    for tag, env, fn, bpp in variants:
        ms = sorted(res[tag])[1]
        print(f"{tag:40s} {ms:.4f} ms   {n * rows * cols * bpp / ms / 1e6 / 8000 * 100:5.1f} % of 8 TB/s", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
