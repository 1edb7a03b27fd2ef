#!/usr/bin/env python3
"""tools/boxprobe.py -- what kind of box is this?  The boxes of this pool differ by up to 8 % on the north-star launch.  For the
row kernel, its memory-only variant and the best plain copies: time per launch AND the shader clock sampled WHILE that launch
stream runs (rcv__clock_probe: s_memtime against the 100 MHz s_memrealtime on a concurrent one-wave kernel).
Appends one JSON line to gpurun_out/boxprobe.jsonl."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from tools.ablate_sweep import setenv  # noqa: E402

L = _ffi.lib()
BL = _ffi.bench_lib()   # copy / store / clock probes: librustcv_hip_bench.so, not part of the product library


def clock(ctx, us=3000):
    f = C.c_float()
    assert BL.rcv__clock_probe(ctx.handle, us, C.byref(f)) == 0
    return round(float(f.value), 1)


def timed_with_clock(ctx, fn, steps=120, settle_ms=60.0):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(8):
            fn()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(steps):
        fn()
    mhz = clock(ctx, 20000)   # 20 ms inside the ~70 ms the queued launches take
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / steps, mhz


def main():
    from bench import bench_kernel7
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    k = bench_kernel7()
    alg = n * rows * cols * 6
    nbytes = n * rows * cols * 3
    flt = lambda: device.filter2d(src, dst, k, shift=6)   # noqa: E731

    def cp(v, g):
        return lambda: BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, v, g)
    out = {"idle_mhz": clock(ctx)}
    variants = [("filter", {}, 0, flt), ("memory-only", {}, 4, flt), ("filter wpb=4", {"RCV_FR_WPB": 4}, 0, flt),
                ("memory-only wpb=4", {"RCV_FR_WPB": 4}, 4, flt), ("strip kernel (round 1)", {"RCV_F7_ROWS": 0}, 0, flt),
                ("copy sweep U=2 nt both g=512", {}, 0, cp(21, 512)), ("copy sweep U=4 plain g=256", {}, 0, cp(10, 256)),
                ("copy sweep U=8 nt both g=256", {}, 0, cp(17, 256)), ("copy sweep U=4 nt both g=512", {}, 0, cp(3, 512)),
                ("read only g=2048", {}, 0, cp(6, 2048)), ("write only g=32768", {}, 0, cp(7, 32768))]
    for rep in range(2):
        for tag, env, flags, fn in variants:
            setenv(env)
            L.rcv__debug_set(flags)
            ms, mhz = timed_with_clock(ctx, fn)
            out.setdefault(tag, []).append([round(ms, 4), mhz])
    L.rcv__debug_set(0)
    setenv({})
    for tag, v in out.items():
        if tag == "idle_mhz":
            print(f"idle clock {v} MHz")
            continue
        ms = min(x[0] for x in v)
        print(f"{tag:36s} {ms:.4f} ms  {alg / ms / 1e6:8.1f} GB/s  frac {alg / ms / 1e6 / 8000:.4f}   clock under load {' / '.join(str(x[1]) for x in v)} MHz   ({' '.join(str(x[0]) for x in v)})", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "boxprobe.jsonl"), "a") as f:
        f.write(json.dumps(out) + "\n")
    src.free()
    dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
