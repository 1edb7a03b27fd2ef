#!/usr/bin/env python3
"""tools/soak_resize.py [N] -- N seeded random (source shape, destination shape, batch, paddings) cases of the BGR / gray bilinear
resize against the oracle with the library's own dispatch; prints the kernels used and the number of mismatches.  GPU box."""
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
bad = 0
used = Counter()
for case in range(N):
    rng = np.random.default_rng(0x5E512E + case)
    ch = 3 if case % 5 else 1
    sr, sc = int(rng.integers(4, 500)), int(rng.integers(4, 900))
    if case % 7 == 0:
        dr, dc = sr // 2 or 1, sc // 2 or 1          # near-exact 2x
    else:
        dr, dc = int(rng.integers(1, 500)), int(rng.integers(1, 900))
    n = int(rng.integers(1, 4))
    frames = rng.integers(0, 256, size=(n, sr, sc, ch), dtype=np.uint8)
    src = device.DeviceBatch(ctx, n, sr, sc, ch, step=sc * ch + int(rng.choice([0, 0, 1, 4, 7])))
    dst = device.DeviceBatch(ctx, n, dr, dc, ch, step=dc * ch + int(rng.choice([0, 0, 1, 4, 5])))
    src.upload(frames)
    dst.memset(0x77)
    L.rcv__debug_kernels_reset()
    device.resize(src, dst)
    ctx.sync()
    kn = L.rcv__debug_kernels().decode().split(";")[0]
    used[kn] += 1
    got = dst.download()
    for i in range(n):
        want = oracle.resize(frames[i] if ch == 3 else frames[i, :, :, 0], dr, dc)
        if not np.array_equal(got[i], want):
            bad += 1
            print("MISMATCH", case, kn, (sr, sc, dr, dc, ch, n), int((got[i] != want).sum()), flush=True)
            break
    src.free()
    dst.free()
print(f"soak: {N} cases, kernels {dict(used)}, {bad} mismatches")
