#!/usr/bin/env python3
"""tools/ablate_wpb.py -- waves per workgroup of the row kernel (RCV_FR_WPB): 1 (every wave its own workgroup, wherever the
dispatcher puts it) against 2 / 4 / 8 neighbouring strips of one band on ONE CU; the full kernel and its memory-only variant,
with the band-height and occupancy knobs that interact with it.  Three rotations, medians."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
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
    variants = []
    for wpb in (1, 2, 4, 8):
        for bpf in (0, 16, 27):
            env = {"RCV_FR_WPB": wpb}
            if bpf:
                env["RCV_FR_BPF"] = bpf
            variants.append((f"filter      wpb={wpb} bpf={bpf or 21}", env, 0, flt))
            variants.append((f"memory-only wpb={wpb} bpf={bpf or 21}", env, 4, flt))
        for order in (1,):
            variants.append((f"filter      wpb={wpb} order 1 bpf 43", {"RCV_FR_WPB": wpb, "RCV_FR_ORDER": 1, "RCV_FR_BPF": 43}, 0, flt))

    def cp(v, g):
        return lambda: BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, v, g)
    variants.append(("copy sweep U=2 nt both g=512", {}, 0, cp(21, 512)))
    variants.append(("copy sweep U=8 nt both g=256", {}, 0, cp(17, 256)))
    variants.append(("copy sweep U=4 plain g=256", {}, 0, cp(10, 256)))
    variants.append(("copy sweep U=4 nt both g=512 (round-2 ceiling)", {}, 0, cp(3, 512)))
    # correctness first: every wpb against wpb = 1
    setenv({})
    device.filter2d(src, dst, k, shift=6)
    ref = [dst.download_frame(i) for i in (0, 37, 63)]
    for wpb in (2, 4, 8):
        setenv({"RCV_FR_WPB": wpb})
        dst.memset(0)
        device.filter2d(src, dst, k, shift=6)
        ok = all(np.array_equal(dst.download_frame(i), r) for i, r in zip((0, 37, 63), ref))
        print(f"wpb={wpb}: frames 0/37/63 equal to wpb=1: {ok}", flush=True)
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
        print(f"{tag:48s} median {ms:.4f} ms  ({' '.join(f'{x:.4f}' for x in v)})  {alg / ms / 1e6:8.1f} GB/s  frac {alg / ms / 1e6 / 8000:.4f}", flush=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ablate_wpb.json"), "w"), indent=1)
    src.free()
    dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
