#!/usr/bin/env python3
"""tools/ablate_walk3.py -- (round 4) what separates the strip-walker copy from the kernel's own memory-only variant at short bands?
Walker flags: persistent static waves (bands g, g + ng, ... per wave, all waves resident) / every byte requested by two lanes and
two rows per request (the MFMA kernel's overlapping 48-byte lane windows).  64 x 4K, 32-row and 103-row bands, three rotations."""
import ctypes as C
import statistics
import sys
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROWS, COLS = 2160, 3840


def main():
    import torch  # noqa: F401
    import rustcv_amd as rcv
    from rustcv_amd import _ffi, device
    L, BL = _ffi.lib(), _ffi.bench_lib()
    n = 64
    nbytes = n * ROWS * COLS * 3
    ctx = rcv.Context(0)
    src = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    device.synth(src, 0, 0x5EED0003, 0)

    def timed(fn, launches=60):
        t = time.perf_counter()
        while time.perf_counter() - t < 0.04:
            for _ in range(8):
                fn()
            ctx.sync()
        ms = C.c_float(0.0)
        L.rcv_timer_start(ctx.handle)
        for _ in range(launches):
            fn()
        L.rcv_timer_stop(ctx.handle, C.byref(ms))
        return ms.value / launches

    def walker(W, depth, rounds, flags):
        def fn():
            rc = BL.rcv__stripwalk(ctx.handle, dst.ptr, src.ptr, n, ROWS, COLS * 3, COLS * 3, W, depth, rounds, 8, 6, flags)
            assert rc == 0, (rc, W, depth, flags)
        return fn

    variants = []
    for rounds in (10, 32, 64):
        for depth in (2, 4):
            for flags, name in ((0, "dynamic"), (2, "persistent static"), (4, "dup, 2 rows per request"), (6, "dup + persistent")):
                variants.append((f"walk W=768 rounds={rounds:2d} depth={depth} {name}", walker(768, depth, rounds, flags)))
    res = {nm: [] for nm, _ in variants}
    for r in range(3):
        for nm, fn in variants:
            res[nm].append(timed(fn))
    for nm, _ in variants:
        m = statistics.median(res[nm])
        print(f"  {nm:60s} {m:.4f} ms  {2 * nbytes / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in res[nm]]}")


if __name__ == "__main__":
    main()
