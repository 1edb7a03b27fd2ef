#!/usr/bin/env python3
"""tools/ab_gauss_sigma.py -- (round 5) GaussianBlur(sigma > 0) on 64 x 4K: the row-pair kernel (k_gauss_f32_pairs) against the one-row kernel
(k_filter_f32_stream, RCV_GAUSS_ROWS=0), same process, alternating, medians; outputs compared byte for byte."""
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device

L = _ffi.lib(); ctx = rcv.Context(0)
def knob(v):
    if v is None: os.environ.pop("RCV_GAUSS_ROWS", None)
    else: os.environ["RCV_GAUSS_ROWS"] = str(v)
    L.rcv__debug_reload_knobs()
def timed(fn, launches=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(4): fn()
        ctx.sync()
    ms = C.c_float(); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
for (n, rows, cols, ch) in ((64, 2160, 3840, 3), (64, 2160, 3840, 1), (64, 1080, 1920, 3)):
    src = device.DeviceBatch(ctx, n, rows, cols, ch); a = device.DeviceBatch(ctx, n, rows, cols, ch); b = device.DeviceBatch(ctx, n, rows, cols, ch)
    device.synth(src, 0, 77, 0)
    for ks, sigma in ((3, 0.8), (5, 1.1), (7, 1.5), (9, 2.0), (11, 2.5)):
        knob(None); L.rcv__debug_kernels_reset(); device.gaussian_blur(src, a, ks, sigma); ctx.sync(); k1 = L.rcv__debug_kernels().decode().split(";")[0]
        knob(0); L.rcv__debug_kernels_reset(); device.gaussian_blur(src, b, ks, sigma); ctx.sync(); k0 = L.rcv__debug_kernels().decode().split(";")[0]
        same = all(np.array_equal(a.download_frame(i), b.download_frame(i)) for i in (0, n // 2, n - 1))
        t = {0: [], 1: []}
        for r in range(3):
            knob(None); t[1].append(timed(lambda: device.gaussian_blur(src, a, ks, sigma)))
            knob(0); t[0].append(timed(lambda: device.gaussian_blur(src, b, ks, sigma)))
        m1, m0 = statistics.median(t[1]), statistics.median(t[0])
        px = n * rows * cols
        print(f"{n} x {cols}x{rows}x{ch}  {ks} taps sigma {sigma}:  pairs {m1:.4f} ms ({px * (ch * 2) / m1 / 1e6 / 8000:.3f} of 8 TB/s)   one-row {m0:.4f} ms   ratio {m1 / m0:.3f}   same bytes {same}   [{k1[:40]} | {k0[:40]}]", flush=True)
    knob(None)
    for x in (src, a, b): x.free()
