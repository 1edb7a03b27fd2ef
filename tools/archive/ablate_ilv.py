#!/usr/bin/env python3
"""tools/ablate_ilv.py -- (profiling build: make EXTRA=-DRCV_ABLATE) the north-star kernel's memory pattern with TWO waves per (band, strip)
taking alternate output row pairs (both read every input row; half as many bands in flight at the same occupancy) against its
ordinary memory-only variant and the kernel itself.  Each variant three times in rotation."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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


variants = [("kernel", 0, {}), ("memory-only (one wave per band x strip)", 4, {}),
            ("memory-only, rounds 4 (bands twice as tall)", 4, {"RCV_FR_ROUNDS": 4}),
            ("memory-only, TWO waves per band x strip, rounds 4 (same wave count)", 7, {"RCV_FR_ROUNDS": 4}),
            ("memory-only, TWO waves per band x strip, rounds 8", 7, {"RCV_FR_ROUNDS": 8}),
            ("memory-only, TWO waves per band x strip, rounds 4, no taper", 7, {"RCV_FR_ROUNDS": 4, "RCV_FR_TAPER": 0}),
            ("memory-only, TWO waves, rounds 4, 2-wave workgroups (the pair on one CU)", 7, {"RCV_FR_ROUNDS": 4, "RCV_FR_WPB": 2}),
            ("memory-only, TWO waves, rounds 4, 8-wave workgroups", 7, {"RCV_FR_ROUNDS": 4, "RCV_FR_WPB": 8})]
res = {v[0]: [] for v in variants}
for rep in range(3):
    for name, flag, env in variants:
        for kk in ("RCV_FR_ROUNDS", "RCV_FR_TAPER", "RCV_FR_WPB"):
            os.environ.pop(kk, None)
        for kk, v in env.items():
            os.environ[kk] = str(v)
        L.rcv__debug_reload_knobs()
        L.rcv__debug_set(flag)
        res[name].append(timed())
        L.rcv__debug_set(0)
for name, _, _ in variants:
    t = sorted(res[name])
    print(f"{name:76s} median {t[1]:.4f} ms  ({2 * nbytes / t[1] / 1e6 / 8000:.3f})   {' '.join('%.4f' % x for x in res[name])}", flush=True)
