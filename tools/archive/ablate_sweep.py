#!/usr/bin/env python3
"""tools/ablate_sweep.py -- traversal orders of the north-star launch (64 x 4K BGR, 7x7 i8) inside ONE process on ONE box.

Round 3, VERDICT item 1: the one plain copy that beats the row kernel is the single global sweep (the whole GPU inside one
moving window).  This tool times the kernel's own traversal orders against each other and against that copy:

    order 0   every XCD its own contiguous eighth of the bands (the default)
    order 1   bands dealt round-robin to the XCDs, strips of a band neighbours on one L2: a global raster sweep whose window
              is (resident waves / strips) bands high -- short bands = a window of a few MB
    (order 2 = plain raster over (band, strip) and RCV_FR_PERSIST = persistent waves sweeping the item list with a static stride
     were measured with the first round-3 build -- profiles/r03_ablate_sweep_orders.txt -- and removed: never better than order 1)
    bpf       bands per frame (2160 / bpf rows per band)

plus the kernel's memory-only variant (no MFMA) and the best plain copies.  Every variant three times in rotation; medians
compare.  Writes gpurun_out/ablate_sweep.json.
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402

L = _ffi.lib()
BL = _ffi.bench_lib()   # copy / store / clock probes: librustcv_hip_bench.so, not part of the product library
KNOBS = ("RCV_F7_ROWS", "RCV_FR_WPC", "RCV_FR_ROUNDS", "RCV_FR_PP", "RCV_FR_ORDER", "RCV_FR_BPF", "RCV_FR_PERSIST", "RCV_FR_WPB", "RCV_FR_GATE")


def setenv(env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    L.rcv__debug_reload_knobs()


def timeit(ctx, fn, steps=100, settle_ms=60.0):
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
    from bench import bench_kernel7
    quick = "--quick" in sys.argv
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    k = bench_kernel7()
    alg = n * rows * cols * 6
    nbytes = n * rows * cols * 3
    flt = lambda: device.filter2d(src, dst, k, shift=6)   # noqa: E731
    variants = [("default (order 0, 21 bands per frame)", {}, 0, flt)]
    bpfs = (21, 43, 68, 135, 270) if not quick else (21, 135)
    for order in (0, 1):
        for bpf in bpfs:
            if order == 0 and bpf == 21:
                continue
            variants.append((f"order {order}, bpf {bpf} ({2160 / bpf:.0f} rows)", {"RCV_FR_ORDER": order, "RCV_FR_BPF": bpf}, 0, flt))
    variants.append(("memory-only (no MFMA), default order", {}, 4, flt))
    variants.append(("memory-only, order 1 bpf 135", {"RCV_FR_ORDER": 1, "RCV_FR_BPF": 135}, 4, flt))
    variants.append(("memory-only, order 1 bpf 270", {"RCV_FR_ORDER": 1, "RCV_FR_BPF": 270}, 4, flt))
    names = {1: "copy sweep", 3: "copy sweep nt", 5: "copy block nt", 9: "copy XCD-local sweep nt"}
    for variant, grid in ((3, 512), (3, 1024), (3, 2048), (1, 1024), (5, 2048), (9, 2048)):
        def cp(variant=variant, grid=grid):
            assert BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, variant, grid) == 0
        variants.append((f"{names[variant]} g={grid}", {}, 0, cp))
    res = {v[0]: [] for v in variants}
    for rep in range(3):
        for tag, env, flags, fn in variants:
            setenv(env)
            L.rcv__debug_set(flags)
            res[tag].append(timeit(ctx, fn, settle_ms=60.0 if rep else 120.0))
    L.rcv__debug_set(0)
    setenv({})
    # correctness of the new orders (the timed variants must be the real computation): frame 0 / 63 rows against the default order
    import numpy as np
    device.filter2d(src, dst, k, shift=6)
    ref = dst.download_frame(0) if hasattr(dst, "download_frame") else None
    if ref is not None:
        for env in ({"RCV_FR_ORDER": 1, "RCV_FR_BPF": 135}, {"RCV_FR_ORDER": 1, "RCV_FR_BPF": 270}, {"RCV_FR_ORDER": 1, "RCV_FR_BPF": 43, "RCV_FR_WPB": 4}):
            setenv(env)
            dst.memset(0)
            device.filter2d(src, dst, k, shift=6)
            got = dst.download_frame(0)
            print("check", env, "frame 0 equal:", bool(np.array_equal(got, ref)), flush=True)
        setenv({})
    for tag, v in res.items():
        ms = sorted(v)[1]
        print(f"{tag:50s} median {ms:.4f} ms  ({' '.join(f'{x:.4f}' for x in v)})  {alg / ms / 1e6:8.1f} GB/s  frac {alg / ms / 1e6 / 8000:.4f}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ablate_sweep.json"), "w"), indent=1)
    src.free()
    dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
