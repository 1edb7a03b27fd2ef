#!/usr/bin/env python3
"""tools/ablate_copy.py -- what a plain device copy of the north-star bytes (2 x 1.59 GB) is worth on THIS box as a function of
its shape: workgroups (bytes in flight), accesses in flight per thread, non-temporal loads / stores -- next to the row kernel
and its memory-only variant (rcv__debug_set(4): the kernel's loads and stores with nothing in between).  Three rotations,
medians.  Writes gpurun_out/ablate_copy.json."""
import ctypes as C
import json
import os
import sys
import time

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
    variants = [("filter default", {}, 0, flt), ("filter memory-only", {}, 4, flt)]
    for wpc in (4, 6, 8):
        variants.append((f"filter memory-only wpc {wpc}", {"RCV_FR_WPC": wpc}, 4, flt))
    for bpf in (16, 30, 43):
        variants.append((f"filter memory-only bpf {bpf}", {"RCV_FR_BPF": bpf}, 4, flt))
        variants.append((f"filter bpf {bpf}", {"RCV_FR_BPF": bpf}, 0, flt))
    ntn = ("plain", "nt loads", "nt stores", "nt both")
    for ui, U in ((2, 2), (0, 4), (1, 8)):
        for nt in range(4):
            for g in (256, 384, 512, 768, 1024):
                v = 10 + 4 * ui + nt

                def cp(v=v, g=g):
                    assert BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, v, g) == 0
                variants.append((f"sweep U={U} {ntn[nt]:9s} g={g}", {}, 0, cp))
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
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ablate_copy.json"), "w"), indent=1)
    src.free()
    dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
