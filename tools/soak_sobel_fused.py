#!/usr/bin/env python3
"""tools/soak_sobel_fused.py [N] -- (round 6) N seeded random cases of the one-launch filter2D -> gray -> Sobel (rcv_filter2d_i8_sobel_batch) against
the oracle's composition: widths that are multiples of 4 from 16 to 4200 (every place of a row's end inside a group of 960 pixels, inside a wave's 256 and
inside the seam windows), 4 .. 300 rows, 1 .. 9 frames (bands that cross frames), padded source / gradient rows, kernels 3 / 5 / 7, shifts 0 .. 8,
saturating frames.  Prints the kernels used and the number of mismatches.  Run on a GPU box."""
import os, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from rustcv_amd._ffi import RCV_16S
from oracle import pyoracle as oracle
L = _ffi.lib()
ctx = rcv.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad, used = 0, Counter()
for case in range(N):
    rng = np.random.default_rng(0x50BE1 + case)
    near = int(rng.choice([256, 512, 768, 944, 960, 976, 1024, 1920, 2880, 3840]))
    cols = int(4 * rng.integers(4, 1050)) if case % 2 else max(16, near + 4 * int(rng.integers(-5, 6)))
    rows = int(rng.integers(4, 300)) if cols < 2000 else int(rng.integers(4, 60))
    n = int(rng.integers(1, 10)) if rows * cols < 400000 else int(rng.integers(1, 3))
    ks = int(rng.choice([3, 5, 7]))
    shift = int(rng.integers(0, 9))
    k = rng.integers(-20, 21, size=(ks, ks)).astype(np.int8) if shift else rng.integers(-2, 3, size=(ks, ks)).astype(np.int8)
    frames = rng.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    if case % 7 == 0:
        frames[0] = (rng.integers(0, 2, size=(rows, cols, 1)) * 255).astype(np.uint8)
    spad, gpad = int(rng.choice([0, 4, 12, 64])), int(rng.choice([0, 8, 16, 120]))
    src = device.DeviceBatch(ctx, n, rows, cols, 3, step=cols * 3 + spad)
    src.upload(frames)
    dx = device.DeviceBatch(ctx, n, rows, cols, 1, RCV_16S, step=cols * 2 + gpad)
    dy = device.DeviceBatch(ctx, n, rows, cols, 1, RCV_16S, step=cols * 2 + gpad)
    dx.memset(0x7B); dy.memset(0x7B)
    L.rcv__debug_kernels_reset()
    device.filter2d_sobel(src, dx, dy, k, shift)
    used[L.rcv__debug_kernels().decode().split("(")[1].split(")")[0][:40] if "(" in L.rcv__debug_kernels().decode() else "?"] += 1
    gx, gy = dx.download(), dy.download()
    for i in range(n):
        wx, wy = oracle.sobel(oracle.bgr2gray(oracle.filter2d_i8(frames[i], k, shift)))
        if not (np.array_equal(gx[i].reshape(rows, cols), wx.reshape(rows, cols)) and np.array_equal(gy[i].reshape(rows, cols), wy.reshape(rows, cols))):
            bad += 1
            print(f"MISMATCH case {case}: n={n} rows={rows} cols={cols} ks={ks} shift={shift} frame {i}", flush=True)
            break
    for b in (src, dx, dy):
        b.free()
print(f"{N} cases, {bad} mismatches; kernels: {dict(used)}")
sys.exit(1 if bad else 0)
