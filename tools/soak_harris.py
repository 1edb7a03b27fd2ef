#!/usr/bin/env python3
"""tools/soak_harris.py [N] -- N seeded random (shape, source kind, block size, row padding, threshold) cases of cornerHarris and
of the Harris pipeline (mask, and mask + response) against the oracle with the library's own dispatch: aligned and ragged shapes,
BGR / gray sources, every block size, black/white images among the random ones (window sums at their maximum).  Prints the kernels
used and the number of mismatches.  GPU box."""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402

L = _ffi.lib()
ctx = rcv.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
FIXED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
used = Counter()
for case in range(N):
    rng = np.random.default_rng(0x4A2215 + case)
    block = FIXED or int(rng.integers(1, 8))
    rows = int(rng.integers(8, 300)) if not FIXED or case % 4 else int(rng.integers(130, 900))
    cols = int(rng.integers(1, 130)) * 8 if case % 3 else int(rng.integers(8, 1100))
    if FIXED and case % 7 == 0: cols = int(rng.choice([488, 496, 504, 984, 992, 1000, 1488, 1496]))   # around the 496-pixel strips
    ch = 3 if case % 2 else 1
    if FIXED and case % 5 == 1:
        ch, cols = 2, cols + (cols & 1)     # packed YUYV
    n = int(rng.integers(1, 4)) if not FIXED else int(rng.integers(1, 6))
    if case % 5 == 0:      # black / white: |Ix|, |Iy| up to 1020
        frames = (rng.integers(0, 2, size=(n, rows, cols, ch)) * 255).astype(np.uint8)
    else:
        frames = rng.integers(0, 256, size=(n, rows, cols, ch), dtype=np.uint8)
    pad = int(rng.choice([0, 0, 8, 16, 3])) if cols % 8 == 0 else int(rng.choice([0, 1, 5]))
    src = device.DeviceBatch(ctx, n, rows, cols, ch, step=cols * ch + pad)
    src.upload(frames)
    thr = float(rng.choice([1e-6, 1e-4, 1e-2, 0.0])) if not FIXED else float(rng.choice([1e-6, 1e-4, 1e-2, 0.0, -np.inf, np.nan, 3e-3]))
    k = 0.04 if case % 4 else 0.06
    want_resp = case % 3 == 0
    mask = device.DeviceBatch(ctx, n, rows, cols, 1)
    resp = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F) if (want_resp or ch == 1) else None
    ok = True
    L.rcv__debug_kernels_reset()
    device.harris_pipeline(src, mask, resp if want_resp else None, block, k, thr)
    ctx.sync()
    kn = L.rcv__debug_kernels().decode().split(";")[0]
    used[kn] += 1
    gm = mask.download()
    gr = resp.download() if want_resp else None
    for i in range(n):
        if ch == 2:
            bgr = np.zeros(rows * cols * 3, np.uint8)
            oracle.yuv422_to_bgr_strided(np.ascontiguousarray(frames[i]).reshape(-1), cols * 2, rows, cols, False, bgr)
            gray = oracle.bgr2gray(bgr.reshape(rows, cols, 3))
        else:
            gray = frames[i, :, :, 0] if ch == 1 else oracle.bgr2gray(frames[i])
        wr = oracle.corner_harris(gray, block, k)
        wm = oracle.nms3x3(wr, thr)
        if not np.array_equal(gm[i].reshape(rows, cols), wm.reshape(rows, cols)) or (want_resp and not np.array_equal(gr[i].reshape(rows, cols).view(np.uint32), wr.view(np.uint32))):
            ok = False
    if ch == 1:   # cornerHarris proper
        L.rcv__debug_kernels_reset()
        device.corner_harris(src, resp, block, k)
        ctx.sync()
        used[L.rcv__debug_kernels().decode().split(";")[0]] += 1
        gr = resp.download()
        for i in range(n):
            if not np.array_equal(gr[i].reshape(rows, cols).view(np.uint32), oracle.corner_harris(frames[i, :, :, 0], block, k).view(np.uint32)):
                ok = False
    if not ok:
        bad += 1
        print("MISMATCH", case, kn, (rows, cols, ch, n, block, pad, thr, want_resp), flush=True)
    for b in (src, mask, resp):
        if b is not None:
            b.free()
print(f"soak: {N} cases, kernels {dict(used)}, {bad} mismatches")
