#!/usr/bin/env python3
"""tools/ablate_chain_misc.py -- (round 4) chained-band kernel, small levers in one run: items per ticket (1 / 2 / 4), row pairs in flight
(2 / 3 / 4), band height 28 / 32 / 36; filter and (where it exists) memory-only variant; three rotations, medians."""
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
    from tools._rows import Rows
    L = _ffi.lib()
    n = 64
    nbytes = n * ROWS * COLS * 3
    ctx = rcv.Context(0)
    src = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    rows = Rows(ctx, src, dst, bench_kernel7())

    def timed(fn, launches=80):
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

    variants = [("one band per wave", dict(chain=0)), ("chain default (32 rows, 1 item per ticket, 3 pairs in flight)", dict(chain=1)),
                ("chain memory-only", dict(chain=1, dbg=4)),
                ("chain 2 items per ticket", dict(chain=1, wpb=2)), ("chain 2 items per ticket, memory-only", dict(chain=1, wpb=2, dbg=4)),
                ("chain 4 items per ticket", dict(chain=1, wpb=4)),
                ("chain 4 pairs in flight", dict(chain=1, pp=4)), ("chain 2 pairs in flight", dict(chain=1, pp=2)),
                ("chain 4 pairs in flight, 2 items per ticket", dict(chain=1, pp=4, wpb=2)),
                ("chain 28 rows", dict(chain=1, chain_rows=28)), ("chain 36 rows", dict(chain=1, chain_rows=36)),
                ("chain edge weight 108 %", dict(chain=1, edge_pct=108)), ("chain edge weight 125 %", dict(chain=1, edge_pct=125))]
    fns = [(nm, rows.fn(**t)) for nm, t in variants]
    res = {nm: [] for nm, _ in fns}
    for r in range(3):
        for nm, fn in fns:
            res[nm].append(timed(fn))
    for nm, _ in fns:
        m = statistics.median(res[nm])
        print(f"  {nm:68s} {m:.4f} ms  {2 * nbytes / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in res[nm]]}")


if __name__ == "__main__":
    main()
