#!/usr/bin/env python3
"""tools/ablate_streams.py -- how much of the distance between the north-star kernel and the best plain copy is the NUMBER OF STREAMS?
The nt sweep copy (U = 2) of the benchmark's 2 x 1.59 GB with the buffer cut into N contiguous regions, each swept by its own share of
512 / 576 workgroups, N = 1 ... 576 -- a band-streaming stencil keeps ~136 row streams going -- next to the kernel and its memory-only
variant in the same run."""
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
BL = _ffi.bench_lib()
ctx = rcv.Context(0)
n, rows, cols = 64, 2160, 3840
src = device.DeviceBatch(ctx, n, rows, cols, 3)
dst = device.DeviceBatch(ctx, n, rows, cols, 3)
device.synth(src, 1, 0x5EED0003, 0)
nbytes = n * rows * cols * 3


def timed(fn, launches=100):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.06:
        for _ in range(8):
            fn()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(launches):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / launches


def report(name, ms):
    print(f"{name:58s} {ms:8.4f} ms  {2 * nbytes / ms / 1e6:8.1f} GB/s  {2 * nbytes / ms / 1e6 / 8000:6.3f}", flush=True)


k = np.arange(-24, 25, dtype=np.int8).reshape(7, 7)
kp = k.ctypes.data_as(C.POINTER(C.c_int8))
bs, bd = src.as_rcv(), dst.as_rcv()


def filt():
    assert L.rcv_filter2d_i8_batch(ctx.handle, C.byref(bs), C.byref(bd), kp, 7, 6) == 0


for rep in range(2):
    report("k_filter_rows_mfma (the benchmark launch)", timed(filt))
    L.rcv__debug_set(4)
    report("  its memory-only variant", timed(filt))
    L.rcv__debug_set(0)
    for wgs in (512, 576):
        for ns in (1, 2, 8, 16, 32, 64, 128, 192, 288, 576) if wgs == 576 else (1, 2, 8, 16, 32, 64, 128, 256, 512):
            if wgs % ns or (nbytes // 16) % ns:
                continue
            for variant in (40, 41):
                if variant == 41 and (wgs % 8 or (wgs // 8) % (wgs // ns)):
                    continue
                g = wgs | (ns << 16)

                def cp():
                    rc = BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, variant, g)
                    assert rc == 0, rc
                report(f"nt sweep U=2, {wgs} workgroups, {ns:3d} streams" + (" (a stream on one XCD)" if variant == 41 else ""), timed(cp))
