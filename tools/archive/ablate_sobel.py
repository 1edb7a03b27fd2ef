#!/usr/bin/env python3
"""tools/ablate_sobel.py -- the Sobel kernel on 64 x 4K gray: non-temporal against plain stores, rows per segment, workgroups per CU
(same-run A/B, three rotations, medians)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from tools.ablate_sweep import timeit  # noqa: E402

L = _ffi.lib()
KN = ("RCV_SOBEL_SEG", "RCV_SOBEL_PLAIN", "RCV_SOBEL_WGS", "RCV_XCD_ORDER")


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
    dx = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    device.synth(gray, 1, 3, 0)
    device.synth(bgr, 1, 4, 0)
    alg = n * rows * cols * 5
    variants = [("default (nt, plan: 68 rows, 3 WG/CU)", {})]
    for seg in (10, 12, 16, 20, 24, 27, 30, 36, 40, 45, 50, 54, 60, 68, 90):
        variants.append((f"nt    seg={seg:3d} wgs=3", {"RCV_SOBEL_SEG": seg, "RCV_SOBEL_WGS": 3}))
    for seg in (20, 30):
        variants.append((f"nt    seg={seg:3d} wgs=2", {"RCV_SOBEL_SEG": seg, "RCV_SOBEL_WGS": 2}))
        variants.append((f"nt    seg={seg:3d} wgs=4", {"RCV_SOBEL_SEG": seg, "RCV_SOBEL_WGS": 4}))
        variants.append((f"plain seg={seg:3d} wgs=3", {"RCV_SOBEL_SEG": seg, "RCV_SOBEL_WGS": 3, "RCV_SOBEL_PLAIN": 1}))
        variants.append((f"nt    seg={seg:3d} wgs=3, plain block order", {"RCV_SOBEL_SEG": seg, "RCV_SOBEL_WGS": 3, "RCV_XCD_ORDER": 0}))
    res = {v[0]: [] for v in variants}
    resb = {}
    for rep in range(3):
        for tag, env in variants:
            setenv(env)
            res[tag].append(timeit(ctx, lambda: device.sobel(gray, dx, dy), steps=50, settle_ms=40.0 if rep else 80.0))
            resb.setdefault(tag, []).append(timeit(ctx, lambda: device.sobel(bgr, dx, dy), steps=30, settle_ms=30.0))
    setenv({})
    for tag, v in res.items():
        ms = sorted(v)[1]
        msb = sorted(resb[tag])[1]
        print(f"{tag:44s} gray {ms:.4f} ms  {alg / ms / 1e6:7.1f} GB/s alg.  frac {alg / ms / 1e6 / 8000:.4f}    BGR source {msb:.4f} ms  frac {alg / 5 * 7 / msb / 1e6 / 8000:.4f}", flush=True)
    for b in (gray, bgr, dx, dy):
        b.free()
    ctx.close()


if __name__ == "__main__":
    main()
