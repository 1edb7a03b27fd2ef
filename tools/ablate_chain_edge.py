#!/usr/bin/env python3
"""tools/ablate_chain_edge.py -- (round 4) chained-band kernel: share of the waves that start on the edge-strip queue (weight of an edge
strip against an interior one, in %) x band height; filter and memory-only variant, three rotations."""
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
    L = _ffi.lib()
    n = 64
    nbytes = n * ROWS * COLS * 3
    k = bench_kernel7()
    kp = k.ctypes.data_as(C.POINTER(C.c_int8))
    ctx = rcv.Context(0)
    src = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    bs, bd = src.as_rcv(), dst.as_rcv()

    from tools._rows import Rows
    rows = Rows(ctx, src, dst, k)

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

    res = {}
    for r in range(3):
        res.setdefault(("one band per wave", 0, 103, "filter"), []).append(timed(rows.fn(chain=0)))
        for wgt in (80, 100, 115, 140, 200):
            for hgt in (24, 32):
                res.setdefault(("chained", wgt, hgt, "filter"), []).append(timed(rows.fn(chain=1, chain_rows=hgt, edge_pct=wgt)))
                res.setdefault(("chained", wgt, hgt, "memonly"), []).append(timed(rows.fn(chain=1, chain_rows=hgt, edge_pct=wgt, dbg=4)))
    for key in res:
        m = statistics.median(res[key])
        print(f"  {key[0]:18s} edge weight {key[1]:3d} % rows {key[2]:3d} {key[3]:8s} {m:.4f} ms  {2 * nbytes / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in res[key]]}")


if __name__ == "__main__":
    main()
