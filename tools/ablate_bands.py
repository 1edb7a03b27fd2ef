#!/usr/bin/env python3
"""tools/ablate_bands.py -- (round 4) band height: the north-star kernel, its memory-only variant and the strip-walker copy at the
SAME bands per frame (tune bpf; 21 = the default plan, 103-row bands).  The walker says short bands (a compact window per XCD)
are worth 6-10 %; the kernel does not show it -- is that its per-band prologue (tables, halo rows, pipeline fill)?
Same process, three rotations, medians."""
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

    from tools._rows import Rows
    rows = Rows(ctx, src, dst, k)

    bpfs = (0, 27, 34, 45, 68, 90, 135)
    res = {}
    for r in range(3):
        for bpf in bpfs:
            for taper in ((1, 0) if bpf else (1,)):
                res.setdefault((bpf, taper, "filter"), []).append(timed(rows.fn(chain=0, bpf=bpf, taper=taper)))
                res.setdefault((bpf, taper, "memonly"), []).append(timed(rows.fn(chain=0, bpf=bpf, taper=taper, dbg=4)))
            rounds = max(1, round((bpf or 21) * 64 * 15 / 2048))

            def walk():
                rc = BL.rcv__stripwalk(ctx.handle, dst.ptr, src.ptr, n, ROWS, COLS * 3, COLS * 3, 768, 2, rounds, 8, 6, 0)
                assert rc == 0, rc
            res.setdefault((bpf, rounds, "walker"), []).append(timed(walk))
    print("bands per frame (0 = default plan: 21, tapered) | taper / walker rounds | what | median ms | frac of 8 TB/s | samples")
    for key in sorted(res, key=lambda t: (t[0], t[2], t[1])):
        m = statistics.median(res[key])
        print(f"  bpf {key[0]:4d}  {key[1]:3d}  {key[2]:8s} {m:.4f} ms  {2 * nbytes / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in res[key]]}")


if __name__ == "__main__":
    main()
