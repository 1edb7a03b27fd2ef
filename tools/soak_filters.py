#!/usr/bin/env python3
"""tools/soak_filters.py [N] -- N seeded random (shape, alignment, batch, kernel) cases of the integer filter2D / GaussianBlur on BGR and
gray images against the oracle with the library's own dispatch (no knobs except RCV_FR_CHAIN=1 on every fifth BGR case: a batch of 8 / 16 / 24 frames of >= 64 rows:
k_filter_rows_chain, launch after launch on one context = its ticket accounting): every width (multiples of 16, of 4, odd), packed and
padded rows, weights inside and beyond the i8 range.  Prints the kernels used and the number of mismatches.  Run on a GPU box."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
used = Counter()
for case in range(N):
    rng = np.random.default_rng(0xF117E2 + case)
    ch = 3 if case % 4 else 1
    kind = case % 3
    cols = int(16 * rng.integers(1, 130)) if kind == 0 else (int(4 * rng.integers(4, 520)) if kind == 1 else int(rng.integers(16, 2100)))
    rows = int(rng.integers(4, 260))
    n = int(rng.integers(1, 5))
    if ch == 3 and case % 5 == 4:   # (round 4) shapes the persistent ticket-queue row kernel takes: batches of 8k frames, >= 64 rows
        n, rows = 8 * int(rng.integers(1, 4)), int(rng.integers(64, 420))
    os.environ.pop("RCV_FR_CHAIN", None)
    if ch == 3 and case % 5 == 4:
        os.environ["RCV_FR_CHAIN"] = "1"   # (every eligible launch, not only the ones that fill the GPU)
    L.rcv__debug_reload_knobs()
    ks = int(rng.choice([3, 5, 7]))
    pad = int(rng.choice([0, 0, 4, 16, 1]))
    step = cols * ch + pad
    frames = rng.integers(0, 256, size=(n, rows, cols, ch), dtype=np.uint8)
    src = device.DeviceBatch(ctx, n, rows, cols, ch, step=step)
    dst = device.DeviceBatch(ctx, n, rows, cols, ch, step=step)
    src.upload(frames)
    img = lambda i: frames[i] if ch == 3 else frames[i, :, :, 0]   # noqa: E731
    k = rng.integers(-30, 31, size=(ks, ks)).astype(np.int8)
    shift = int(rng.integers(0, 9))
    for tag, fn, ref in (("filter2D", lambda: device.filter2d(src, dst, k, shift=shift), lambda a: oracle.filter2d_i8(a, k, shift)),
                         ("gaussian", lambda: device.gaussian_blur(src, dst, ks, 0.0), lambda a: oracle.gaussian_blur(a, ks, 0.0))):
        dst.memset(0x3C)
        L.rcv__debug_kernels_reset()
        fn()
        ctx.sync()
        kn = L.rcv__debug_kernels().decode().split(";")[0].split("<")[0].strip("(")
        used[kn] += 1
        got = dst.download()
        for i in range(n):
            if not np.array_equal(got[i], ref(img(i))):
                bad += 1
                print("MISMATCH", case, tag, kn, (rows, cols, ch, n, ks, pad, shift), flush=True)
                break
    src.free()
    dst.free()
print(f"soak: {N} cases, kernels {dict(used)}, {bad} mismatches")
