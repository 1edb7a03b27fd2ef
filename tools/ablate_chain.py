#!/usr/bin/env python3
"""tools/ablate_chain.py -- (round 4) the chained-band kernel (k_filter_rows_chain) against the one-band-per-wave kernel
(k_filter_rows_mfma, RCV_FR_CHAIN=0), band heights 16 .. 128 rows, each with its memory-only variant and the strip-walker copy
at the same band height.  64 x 4K BGR 7x7; same process, three rotations, medians."""
import ctypes as C
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROWS, COLS = 2160, 3840


def main():
    import torch  # noqa: F401
    import rustcv_amd as rcv
    from rustcv_amd import _ffi, device
    from bench import bench_kernel7
    L, BL = _ffi.lib(), _ffi.bench_lib()
    n = 64
    nbytes = n * ROWS * COLS * 3
    k = bench_kernel7()
    kp = k.ctypes.data_as(C.POINTER(C.c_int8))
    ctx = rcv.Context(0)
    src = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    bs, bd = src.as_rcv(), dst.as_rcv()

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

    def filt():
        rc = L.rcv_filter2d_i8_batch(ctx.handle, C.byref(bs), C.byref(bd), kp, 7, 6)
        assert rc == 0, rc
    from tools._rows import Rows
    rows = Rows(ctx, src, dst, k)

    res = {}
    heights = (24, 32, 40, 64)
    for r in range(3):
        res.setdefault(("one band per wave (103 rows, tapered)", "filter"), []).append(timed(rows.fn(chain=0)))
        res.setdefault(("one band per wave (103 rows, tapered)", "memonly"), []).append(timed(rows.fn(chain=0, dbg=4)))
        for hgt in heights:
            L.rcv__debug_kernels_reset()
            res.setdefault((f"chained {hgt:3d} rows", "filter"), []).append(timed(rows.fn(chain=1, chain_rows=hgt)))
            assert "k_filter_rows_chain" in L.rcv__debug_kernels().decode()
            res.setdefault((f"chained {hgt:3d} rows", "memonly"), []).append(timed(rows.fn(chain=1, chain_rows=hgt, dbg=4)))
            rounds = max(1, round(ROWS / hgt * 64 * 15 / 2048))

            def walk():
                rc = BL.rcv__stripwalk(ctx.handle, dst.ptr, src.ptr, n, ROWS, COLS * 3, COLS * 3, 768, 4, rounds, 8, 6, 4)   # kernel-like: dup lanes, 2 rows per request, 4 in flight
                assert rc == 0, rc
            res.setdefault((f"chained {hgt:3d} rows", "walker"), []).append(timed(walk))
    # the default plan, as shipped
    L.rcv__debug_kernels_reset()
    t = timed(filt, 200)
    print(f"default plan: {L.rcv__debug_kernels().decode().split(';')[0]}  {t:.4f} ms  frac {2 * nbytes / t / 1e6 / 8000:.4f}")
    for key in res:
        m = statistics.median(res[key])
        print(f"  {key[0]:40s} {key[1]:24s} {m:.4f} ms  {2 * nbytes / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in res[key]]}")


if __name__ == "__main__":
    main()
