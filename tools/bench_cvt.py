#!/usr/bin/env python3
"""tools/bench_cvt.py -- every cvtColor code at 4K on 32 device-resident frames: ms per launch, algorithmic GB/s and the kernel each call took
(the flat reference codes, the stride-aware capture formats of row f2, the display / codec swizzles of row f4)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402

L = _ffi.lib()
ctx = rcv.Context(0)
n, rows, cols = 32, 2160, 3840


def timed(fn, launches=60):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.08:
        for _ in range(4):
            fn()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(launches):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / launches


def B(r, c, ch, pad=0):
    return device.DeviceBatch(ctx, n, r, c, ch, step=c * ch + pad)


cases = [("YUYV2BGR (flat)", _ffi.RCV_YUYV2BGR, lambda: (B(rows, cols, 2), B(rows, cols, 3)), 5),
         ("YUYV2BGR_TWIN", _ffi.RCV_YUYV2BGR_TWIN, lambda: (B(rows, cols, 2), B(rows, cols, 3)), 5),
         ("BGRA2BGR (flat)", _ffi.RCV_BGRA2BGR, lambda: (B(rows, cols, 4), B(rows, cols, 3)), 7),
         ("RGB2BGR", _ffi.RCV_RGB2BGR, lambda: (B(rows, cols, 3), B(rows, cols, 3)), 6),
         ("BGR2GRAY", _ffi.RCV_BGR2GRAY, lambda: (B(rows, cols, 3), B(rows, cols, 1)), 4),
         ("BGR2GRAY padded steps", _ffi.RCV_BGR2GRAY, lambda: (B(rows, cols, 3, 64), B(rows, cols, 1, 64)), 4),
         ("BGR2BGRX", _ffi.RCV_BGR2BGRX, lambda: (B(rows, cols, 3), B(rows, cols, 4)), 7),
         ("BGR2RGB", _ffi.RCV_BGR2RGB, lambda: (B(rows, cols, 3), B(rows, cols, 3)), 6),
         ("YUYV2BGR_STRIDED padded", _ffi.RCV_YUYV2BGR_STRIDED, lambda: (B(rows, cols, 2, 64), B(rows, cols, 3, 64)), 5),
         ("UYVY2BGR_STRIDED padded", _ffi.RCV_UYVY2BGR_STRIDED, lambda: (B(rows, cols, 2, 64), B(rows, cols, 3, 64)), 5),
         ("BGRA2BGR_STRIDED padded", _ffi.RCV_BGRA2BGR_STRIDED, lambda: (B(rows, cols, 4, 64), B(rows, cols, 3, 64)), 7)]
# (NV12 needs a source whose capacity covers the chroma rows behind `rows` luma rows: host-Mat form only, not timed here)
for name, code, mk, bpp in cases:
    try:
        s, d = mk()
    except Exception as e:  # noqa: BLE001
        print(f"{name:28s} alloc failed: {e}")
        continue
    s.memset(0x55)

    def fn():
        device.cvt_color(s, d, code)
    try:
        L.rcv__debug_kernels_reset()
        fn()
        ctx.sync()
        k = L.rcv__debug_kernels().decode()
        ms = timed(fn)
        gbs = n * rows * cols * bpp / ms / 1e6
        print(f"{name:28s} {ms:8.4f} ms  {gbs:8.1f} GB/s  {gbs / 80:5.1f} %   {k}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name:28s} failed: {e}")
    s.free()
    d.free()
