#!/usr/bin/env python3
"""tools/ablate_phase.py -- does it matter WHEN reads and writes reach the memory system?  The nt sweep copy (a) as it is, (b)
with every wave de-synchronised by random sleeps, (c) clock-gated: the whole GPU issues loads only in even and stores only in
odd half periods of the chip-wide 100 MHz counter."""
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
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    alg = n * rows * cols * 6
    nbytes = n * rows * cols * 3

    def cp(v, g, half=0):
        def f():
            rc = BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, v, g | (half << 16))
            assert rc == 0, rc
        return f
    variants = []
    US = {0: 2, 1: 4, 2: 8, 3: 16}
    for ui, g in ((0, 512), (1, 256), (2, 256), (1, 512), (2, 512), (3, 256), (3, 128), (2, 128)):
        variants.append((f"sweep U={US[ui]:2d} g={g}                 ", cp(100 + 10 * ui, g)))
        variants.append((f"sweep U={US[ui]:2d} g={g} desynchronised  ", cp(200 + 10 * ui, g)))
        for half in (50, 100, 150, 200, 300, 400):
            variants.append((f"sweep U={US[ui]:2d} g={g} gated half={half / 100:.1f} us", cp(300 + 10 * ui, g, half)))
    res = {v[0]: [] for v in variants}
    for rep in range(3):
        for tag, fn in variants:
            res[tag].append(timeit(ctx, fn, steps=40, settle_ms=30.0 if rep else 60.0))
    for tag, v in res.items():
        ms = sorted(v)[1]
        print(f"{tag:44s} median {ms:.4f} ms  ({' '.join(f'{x:.4f}' for x in v)})  {alg / ms / 1e6:8.1f} GB/s  frac {alg / ms / 1e6 / 8000:.4f}", flush=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ablate_phase.json"), "w"), indent=1)
    src.free()
    dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
