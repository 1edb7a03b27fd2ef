#!/usr/bin/env python3
"""tools/ablate_stores.py -- the store side of the Sobel kernel: 64 x 4K gray -> two i16 planes = 2.12 GB of
stores per launch, which the kernel writes at 4.8-5.3 TB/s while a plain write sweep of the same bytes reaches 6.1-6.6.  A pure
store kernel with the Sobel kernel's traversal (strip-walking waves, rcv__storebench) over: strip width 1 / 2 KB per plane and row,
one plane of double-width rows against two planes, non-temporal against plain, rows per segment, workgroups per CU, runs of r rows per
plane before switching planes -- next to the real kernel and its ablations."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from tools.ablate_sweep import timeit  # noqa: E402

L = _ffi.lib()
BL = _ffi.bench_lib()   # copy / store / clock probes: librustcv_hip_bench.so, not part of the product library


def main():
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    gray = device.DeviceBatch(ctx, n, rows, cols, 1)
    dx = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    device.synth(gray, 1, 3, 0)
    nbytes = 2 * n * rows * cols * 2
    rb = cols * 2   # 7680 bytes per plane row
    res = {}

    def run(tag, fn, b=nbytes):
        v = sorted(timeit(ctx, fn, steps=60, settle_ms=50.0) for _ in range(3))
        res[tag] = v
        print(f"{tag:86s} {v[1]:.4f} ms   {b / v[1] / 1e6:7.1f} GB/s of stores", flush=True)
    run("Sobel kernel (1 B read + 4 B written per px)", lambda: device.sobel(gray, dx, dy))
    run("write-only sweep (membench 7, g=32768) of the same 2.12 GB", lambda: BL.rcv__membench(ctx.handle, dx.ptr, dx.ptr, nbytes // 2, 7, 32768), nbytes // 2 * 1)
    for nt in (1, 0):
        for chunks in (1, 2):   # 7680 = 7.5 KB: use 7168-byte rows (7 strips of 1 KB) / 6144 (3 strips of 2 KB) -- whole strips only
            rbw = (rb // (1024 * chunks)) * 1024 * chunks
            for planes in (2, 1):
                for seg in (32, 68, 135):
                    for wgs in (0, 3, 2):
                        for pair in (1, 4):
                            if planes == 1 and pair > 1:
                                continue
                            if (seg != 68 or wgs != 3) and pair > 1:
                                continue
                            b = n * rows * rbw * planes      # (one plane: half the bytes -- compare the RATES)
                            run(f"store strips {'nt   ' if nt else 'plain'} {chunks} KB per wave-row, {planes} plane(s), {seg:3d} rows per segment, {wgs or 'max'} WG/CU, runs of {pair}",
                                lambda chunks=chunks, planes=planes, seg=seg, wgs=wgs, pair=pair, nt=nt, rbw=rbw:
                                BL.rcv__storebench(ctx.handle, dx.ptr, dy.ptr, n, rows, rbw, rb, chunks, planes, seg, nt, 1, wgs, pair), b)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ablate_stores.json"), "w"), indent=1)
    for b in (gray, dx, dy):
        b.free()
    ctx.close()


if __name__ == "__main__":
    main()
