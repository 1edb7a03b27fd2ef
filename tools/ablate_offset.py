#!/usr/bin/env python3
"""tools/ablate_offset.py -- does the distance between the source and the destination batch matter?  The north-star launch with
the destination placed at different byte offsets inside one large allocation (same process, three rotations, medians)."""
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


def view(ctx, base, off, n, rows, cols, ch):
    b = device.DeviceBatch.__new__(device.DeviceBatch)
    b.ctx, b.n, b.rows, b.cols, b.channels, b.depth = ctx, n, rows, cols, ch, _ffi.RCV_8U
    b.step = cols * ch
    b.frame_cap = rows * b.step
    b.frame_stride = b.frame_cap
    b.nbytes = n * b.frame_stride
    b.ptr = C.c_void_p(base + off)
    b.free = lambda: None
    return b


def timeit(ctx, fn, steps=150, settle_ms=80.0):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(8):
            fn()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(steps):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / steps


def main():
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    nb = n * rows * cols * 3
    pad = 96 << 20
    p = C.c_void_p()
    _ffi.check(L.rcv_malloc(ctx.handle, 2 * nb + 2 * pad, C.byref(p)), "rcv_malloc")
    base = p.value
    src = view(ctx, base, 0, n, rows, cols, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    k = bench_kernel7()
    first = (nb + 4095) // 4096 * 4096
    offs = [0, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 3 << 19, 1 << 21, 3 << 20, 1 << 22, 5 << 20, 1 << 23,
            (1 << 23) + 4096, 1 << 24, (1 << 24) + 65536, 1 << 25, 1 << 26]
    res = {o: [] for o in offs}
    for rep in range(3):
        for o in offs:
            dst = view(ctx, base, first + o, n, rows, cols, 3)
            res[o].append(timeit(ctx, lambda: device.filter2d(src, dst, k, shift=6), settle_ms=60.0 if rep else 120.0))
    for o in offs:
        v = sorted(res[o])
        print(f"dst = src_end + {o:10d} B   median {v[1]:.4f} ms  ({' '.join(f'{x:.4f}' for x in res[o])})", flush=True)
    L.rcv_free(ctx.handle, p)
    ctx.close()


if __name__ == "__main__":
    main()
