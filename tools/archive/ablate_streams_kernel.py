#!/usr/bin/env python3
"""tools/ablate_streams_kernel.py -- the stream-count question asked of the kernel itself (profiling build: make EXTRA=-DRCV_ABLATE):
fewer bands in flight (RCV_FR_WPC caps the waves per CU through an unused LDS request: 8 / 6 / 4 / 3 / 2 = 136 / 102 / 68 / 51 / 34
concurrent bands) with deeper prefetch per wave (RCV_FR_PP = 3 / 4 / 6 / 8 row pairs) to keep the bytes in flight.  Same process,
each variant three times in rotation."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from bench import bench_kernel7  # noqa: E402

L = _ffi.lib()
ctx = rcv.Context(0)
n, rows, cols = 64, 2160, 3840
src = device.DeviceBatch(ctx, n, rows, cols, 3)
dst = device.DeviceBatch(ctx, n, rows, cols, 3)
device.synth(src, 1, 0x5EED0003, 0)
nbytes = n * rows * cols * 3
k = bench_kernel7()
kp = k.ctypes.data_as(C.POINTER(C.c_int8))
bs, bd = src.as_rcv(), dst.as_rcv()


def filt():
    assert L.rcv_filter2d_i8_batch(ctx.handle, C.byref(bs), C.byref(bd), kp, 7, 6) == 0


def timed(launches=120):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.08:
        for _ in range(8):
            filt()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(launches):
        filt()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / launches


variants = [("default (8 waves per CU, PP 3)", {})]
for wpc in (8, 6, 4, 3, 2):
    for pp in (3, 4, 6, 8):
        for rounds in ((8,) if wpc >= 6 else (8, 16)):
            variants.append((f"WPC {wpc}  PP {pp}  rounds {rounds}", {"RCV_FR_WPC": wpc, "RCV_FR_PP": pp, "RCV_FR_ROUNDS": rounds}))
res = {name: [] for name, _ in variants}
for rep in range(3):
    for name, env in variants:
        for kk in ("RCV_FR_WPC", "RCV_FR_PP", "RCV_FR_ROUNDS"):
            os.environ.pop(kk, None)
        for kk, v in env.items():
            os.environ[kk] = str(v)
        L.rcv__debug_reload_knobs()
        res[name].append(timed())
for name, _ in variants:
    t = sorted(res[name])
    print(f"{name:40s} median {t[1]:.4f} ms  ({2 * nbytes / t[1] / 1e6 / 8000:.3f})   {' '.join('%.4f' % x for x in res[name])}", flush=True)
