#!/usr/bin/env python3
"""tools/ablate_walk.py -- (round 4) which strip geometry does this memory system like for a row-streaming
stencil at FULL occupancy?  `rcv__stripwalk` (librustcv_hip_bench.so) is the stencil's access pattern without the stencil: a wave
owns a strip of W bytes of a band of rows and walks down it, 3072 bytes per request (= 3072 / W strip-rows), `depth` requests in
flight, every byte stored once (non-temporal).  The north-star kernel is W = 768 (15 strips per 4K row: its 2 048 resident waves walk
~136 bands at a time); W = 384 / 256 / 192 put the same waves on 68 / 45 / 34 bands, W = 1536 / 3072 on 273 / 546.

Same process, three rotations, medians; the filter and its own memory-only variant run in every rotation as the controls.
Also: F = 1 .. 4 batches in flight (F contexts on the device, each a full 64-frame filter launch stream) -- ms per launch.

    python tools/ablate_walk.py [--frames 64] [--launches 60] [--rot 3]
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROWS, COLS = 2160, 3840


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--launches", type=int, default=60)
    ap.add_argument("--rot", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ablate_walk.json"))
    a = ap.parse_args()
    import torch  # noqa: F401  (HIP runtime first)
    import rustcv_amd as rcv
    from rustcv_amd import _ffi, device, multigpu
    from bench import bench_kernel7
    L, BL = _ffi.lib(), _ffi.bench_lib()
    n = a.frames
    nbytes = n * ROWS * COLS * 3
    k = bench_kernel7()
    kp = k.ctypes.data_as(C.POINTER(C.c_int8))

    # ---- F batches in flight ----
    inflight = {}
    for F in (1, 2):
        g = multigpu.NativeGroup.in_flight(0, F)
        bufs = []
        for j in range(F):
            s = device.DeviceBatch(g.ctxs[j], n, ROWS, COLS, 3)
            d = device.DeviceBatch(g.ctxs[j], n, ROWS, COLS, 3)
            device.synth(s, 0, 0x5EED0003, 64 * j)
            bufs.append((s, d, s.as_rcv(), d.as_rcv()))
        g.sync()

        def step():
            for j in range(F):
                rc = L.rcv_filter2d_i8_batch(g.ctxs[j].handle, C.byref(bufs[j][2]), C.byref(bufs[j][3]), kp, 7, 6)
                assert rc == 0, rc
        t = time.perf_counter()
        while time.perf_counter() - t < 0.15:
            for _ in range(8):
                step()
            g.sync()
        ms = []
        for _ in range(a.rot):
            g.timer_start()
            for _ in range(200):
                step()
            ms.append(g.timer_stop() / (200 * F))
        inflight[F] = ms
        print(f"in flight F={F}: ms per 64-frame launch {['%.4f' % m for m in ms]}  median {statistics.median(ms):.4f}  "
              f"frac {n * ROWS * COLS * 6 / statistics.median(ms) / 1e6 / 8000:.4f}", flush=True)
        for s, d, _, _ in bufs:
            s.free()
            d.free()
        g.close()

    ctx = rcv.Context(0)
    src = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    bs, bd = src.as_rcv(), dst.as_rcv()

    def timed(fn, launches):
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
    memonly = Rows(ctx, src, dst, k).fn(dbg=4)

    def walker(W, depth, wpc, rounds, halo, ntl=0):
        def fn():
            rc = BL.rcv__stripwalk(ctx.handle, dst.ptr, src.ptr, n, ROWS, COLS * 3, COLS * 3, W, depth, rounds, wpc, halo, ntl)
            assert rc == 0, (rc, W, depth)
        return fn

    def copy(variant, grid):
        def fn():
            rc = BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, variant, grid)
            assert rc == 0, rc
        return fn

    variants = [("filter (k_filter_rows_mfma)", filt), ("filter memory-only", "memonly"), ("copy sweep U=2 nt g=512", copy(21, 512)),
                ("copy 2-region sweep", copy(40, 512 | (2 << 16)))]
    for W in (768, 384, 256, 128):
        for rounds in (8, 16, 32):
            variants.append((f"walk W={W} depth=2 wpc=8 rounds={rounds} halo=6", walker(W, 2, 8, rounds, 6)))
    variants += [("walk W=768 depth=4 wpc=8 rounds=16 halo=6", walker(768, 4, 8, 16, 6)),
                 ("walk W=384 depth=4 wpc=8 rounds=16 halo=6", walker(384, 4, 8, 16, 6)),
                 ("walk W=384 depth=1 wpc=8 rounds=16 halo=6", walker(384, 1, 8, 16, 6)),
                 ("walk W=384 depth=2 wpc=12 rounds=16 halo=6", walker(384, 2, 12, 16, 6)),
                 ("walk W=384 depth=2 wpc=6 rounds=16 halo=6", walker(384, 2, 6, 16, 6)),
                 ("walk W=256 depth=2 wpc=12 rounds=16 halo=6", walker(256, 2, 12, 16, 6)),
                 ("walk W=768 depth=2 wpc=8 rounds=16 halo=0", walker(768, 2, 8, 16, 0))]
    res = {name: [] for name, _ in variants}
    for r in range(a.rot):
        for name, fn in variants:
            if fn == "memonly":
                ms = timed(memonly, a.launches)
            else:
                ms = timed(fn, a.launches)
            res[name].append(ms)
    print(f"{n} x 4K BGR, {a.launches} launches per sample, {a.rot} rotations; frac = 2 x {nbytes} B / ms / 8 TB/s")
    for name, _ in variants:
        m = statistics.median(res[name])
        print(f"  {name:58s} {m:.4f} ms  frac {2 * nbytes / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in res[name]]}")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"inflight_ms_per_launch": inflight, "walk_ms": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
