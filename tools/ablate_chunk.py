#!/usr/bin/env python3
"""tools/ablate_chunk.py -- the north-star batch (64 x 4K, 7x7 i8) as ONE launch against consecutive launches of c frames on the same
stream (RCV_FR_CHUNK), with the band knobs that matter for each; three rotations, medians."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from tools.ablate_sweep import timeit  # noqa: E402

L = _ffi.lib()
KN = ("RCV_FR_CHUNK", "RCV_FR_BAND_ROWS", "RCV_FR_BPF", "RCV_FR_WPC")


def setenv(env):
    for k in KN:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    L.rcv__debug_reload_knobs()


def main():
    from bench import bench_kernel7
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    k = bench_kernel7()
    alg = n * rows * cols * 6
    flt = lambda: device.filter2d(src, dst, k, shift=6)   # noqa: E731
    setenv({})
    flt()
    ref = [dst.download_frame(i) for i in (0, 37, 63)]
    variants = [("one launch of 64 frames (default)", {})]
    for c in (2, 4, 6, 8, 12, 16, 32):
        variants.append((f"launches of {c:2d} frames", {"RCV_FR_CHUNK": c}))
    for br in (64, 90, 108, 127, 135, 180, 270):
        variants.append((f"launches of  8 frames, bands of {br} rows", {"RCV_FR_CHUNK": 8, "RCV_FR_BAND_ROWS": br}))
    for c, br in ((4, 64), (4, 54), (16, 270), (16, 240)):
        variants.append((f"launches of {c:2d} frames, bands of {br} rows", {"RCV_FR_CHUNK": c, "RCV_FR_BAND_ROWS": br}))
    res = {v[0]: [] for v in variants}
    for tag, env in variants:
        setenv(env)
        dst.memset(0)
        flt()
        ok = all(np.array_equal(dst.download_frame(i), r) for i, r in zip((0, 37, 63), ref))
        if not ok:
            print("MISMATCH", tag)
    for rep in range(3):
        for tag, env in variants:
            setenv(env)
            res[tag].append(timeit(ctx, flt, steps=60, settle_ms=40.0 if rep else 80.0))
    setenv({})
    for tag, v in res.items():
        ms = sorted(v)[1]
        print(f"{tag:50s} {ms:.4f} ms  ({' '.join(f'{x:.4f}' for x in v)})  {alg / ms / 1e6:7.1f} GB/s  frac {alg / ms / 1e6 / 8000:.4f}", flush=True)
    src.free()
    dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
