#!/usr/bin/env python3
"""tools/ablate_depth.py -- (profiling build, make EXTRA=-DRCV_ABLATE) the row kernel and its memory-only variant over row pairs
in flight (RCV_FR_PP) x waves per CU (RCV_FR_WPC): does the memory system prefer FEWER waves with DEEPER streams, as the plain
sweep copies do (tools/ablate_copy.py: 4 waves per CU x 8 accesses beat 16 x 2 over the same window by 15 %)?"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from tools.ablate_sweep import setenv, timeit  # noqa: E402

L = _ffi.lib()
BL = _ffi.bench_lib()   # copy / store / clock probes: librustcv_hip_bench.so, not part of the product library


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
    variants = [("filter default", {}, 0, flt), ("memory-only default", {}, 4, flt)]
    for pp in (2, 3, 4, 5, 6, 8):
        for wpc in (3, 4, 5, 6, 8):
            env = {"RCV_FR_PP": pp, "RCV_FR_WPC": wpc}
            variants.append((f"memory-only PP={pp} wpc={wpc}", env, 4, flt))
            variants.append((f"filter      PP={pp} wpc={wpc}", env, 0, flt))

    def cp(v, g):
        return lambda: BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, v, g)
    variants.append(("copy sweep U=2 nt both g=512", {}, 0, cp(21, 512)))
    variants.append(("copy sweep U=8 nt both g=256", {}, 0, cp(17, 256)))
    variants.append(("copy sweep U=4 plain g=256", {}, 0, cp(10, 256)))
    res = {v[0]: [] for v in variants}
    for rep in range(3):
        for tag, env, flags, fn in variants:
            setenv(env)
            L.rcv__debug_set(flags)
            res[tag].append(timeit(ctx, fn, steps=60, settle_ms=40.0 if rep else 80.0))
    L.rcv__debug_set(0)
    setenv({})
    for tag, v in res.items():
        ms = sorted(v)[1]
        print(f"{tag:40s} median {ms:.4f} ms  ({' '.join(f'{x:.4f}' for x in v)})  {alg / ms / 1e6:8.1f} GB/s  frac {alg / ms / 1e6 / 8000:.4f}", flush=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ablate_depth.json"), "w"), indent=1)
    src.free()
    dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
