#!/usr/bin/env python3
"""tools/ablate_sob.py -- the fused filter2D -> gray -> Sobel launch (config 3 in one launch) on 64 x 4K: 240-pixel strips with plain
stores against line-aligned 192-pixel strips with non-temporal stores (RCV_FR_SOB192), same run, three rotations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from tools.ablate_sweep import timeit  # noqa: E402

L = _ffi.lib()


def setenv(env):
    for k in ("RCV_FR_SOB192", "RCV_FR_BPF", "RCV_FR_TAPER"):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    L.rcv__debug_reload_knobs()


def main():
    from bench import bench_kernel7
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    bgr = device.DeviceBatch(ctx, n, rows, cols, 3)
    dx = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    device.synth(bgr, 1, 4, 0)
    k = bench_kernel7()
    alg = n * rows * cols * 7
    fn = lambda: device.filter2d_sobel(bgr, dx, dy, k, shift=6)   # noqa: E731
    variants = [("240-px strips, plain stores", {"RCV_FR_SOB192": 0}), ("192-px strips, nt stores", {"RCV_FR_SOB192": 1}),
                ("192-px strips, nt stores, 27 bands per frame", {"RCV_FR_SOB192": 1, "RCV_FR_BPF": 27}),
                ("192-px strips, nt stores, 16 bands per frame", {"RCV_FR_SOB192": 1, "RCV_FR_BPF": 16}),
                ("240-px strips, equal bands", {"RCV_FR_SOB192": 0, "RCV_FR_TAPER": 0})]
    res = {v[0]: [] for v in variants}
    for rep in range(3):
        for tag, env in variants:
            setenv(env)
            res[tag].append(timeit(ctx, fn, steps=50, settle_ms=40.0 if rep else 80.0))
    setenv({})
    for tag, v in res.items():
        ms = sorted(v)[1]
        print(f"{tag:50s} {ms:.4f} ms  {alg / ms / 1e6:7.1f} GB/s alg.  frac {alg / ms / 1e6 / 8000:.4f}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
