"""The fast kernels must be the ones that run: every op below has a slow generic fallback that produces the same bytes, so a
dispatch condition that silently stops matching would leave the parity tests green.  Each case times the op on a BASELINE-
shaped batch (device events on the context stream, after a warm-up) and asserts a per-frame budget several times above the
fast kernel's measured time and several times below the generic kernel's."""
import ctypes as C
import time

import numpy as np
import pytest

import rustcv_amd as rcv
from rustcv_amd import _ffi, device

pytestmark = pytest.mark.gpu


def _ms_per_call(ctx, fn, steps=6):
    L = _ffi.lib()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.05:      # warm-up: clocks, caches, lazily built tables
        fn()
        ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(steps):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / steps


def test_fast_kernels_are_dispatched(ctx):
    n, rows, cols = 8, 2160, 3840
    B = lambda ch, depth=_ffi.RCV_8U, r=rows, c=cols: device.DeviceBatch(ctx, n, r, c, ch, depth)   # noqa: E731
    bgr, bgr2, gray, gray2, yuyv = B(3), B(3), B(1), B(1), B(2)
    dx, dy, resp, mask = B(1, _ffi.RCV_16S), B(1, _ffi.RCV_16S), B(1, _ffi.RCV_32F), B(1)
    small = B(3, r=540, c=960)
    device.synth(bgr, 1, 7, 0)
    device.synth(gray, 1, 8, 0)
    device.synth(yuyv, 2, 9, 0)
    k7 = np.arange(49, dtype=np.int8).reshape(7, 7) - 24
    kf = np.full((7, 7), 1 / 49, np.float32)
    M = np.array([0.9925, -0.1219, 300.0, 0.1219, 0.9925, -200.0], np.float32)
    # (name, call, budget in ms for the 8-frame 4K batch: ~4x the fast kernel, far below the generic one)
    cases = [
        ("filter2D 7x7 BGR (MFMA strip kernel)", lambda: device.filter2d(bgr, bgr2, k7, shift=6), 0.40),
        ("filter2D 7x7 gray (strip kernel)", lambda: device.filter2d(gray, gray2, k7, shift=6), 0.16),
        ("GaussianBlur 7x7 int BGR (two-table strip kernel)", lambda: device.gaussian_blur(bgr, bgr2, 7, 0.0), 0.45),
        ("fused YUYV -> filter2D", lambda: device.filter2d_yuyv(yuyv, bgr2, k7, shift=6), 0.40),
        ("filter2D 7x7 f32 BGR (stream kernel)", lambda: device.filter2d(bgr, bgr2, kf), 1.6),
        ("GaussianBlur sigma BGR (separable stream kernel)", lambda: device.gaussian_blur(bgr, bgr2, 7, 1.5), 0.7),
        ("Sobel gray", lambda: device.sobel(gray, dx, dy), 0.5),
        ("Sobel of BGR", lambda: device.sobel(bgr, dx, dy), 0.5),
        ("Harris pipeline BGR", lambda: device.harris_pipeline(bgr, mask, None, 2, 0.04, 1e-4), 0.5),
        ("Harris pipeline YUYV", lambda: device.harris_pipeline(yuyv, mask, None, 2, 0.04, 1e-4), 0.5),
        ("Harris pipeline gray", lambda: device.harris_pipeline(gray, mask, None, 2, 0.04, 1e-4), 0.3),
        ("cornerHarris gray", lambda: device.corner_harris(gray, resp, 2, 0.04), 0.5),
        ("NMS 3x3", lambda: device.nms3x3(resp, mask, 1e-4), 0.25),
        ("warpAffine BGR", lambda: device.warp_affine(bgr, bgr2, M), 0.8),
        ("warpAffine gray", lambda: device.warp_affine(gray, gray2, M), 0.5),
        ("resize BGR 4K -> 960x540 (box)", lambda: device.resize(bgr, small), 0.2),
        ("fused warp -> 4x down-scale", lambda: device.warp_affine_resize(bgr, small, M, rows, cols), 0.5),
        ("cvtColor BGR2GRAY", lambda: device.cvt_color(bgr, gray2, _ffi.RCV_BGR2GRAY), 0.25),
        ("cvtColor YUYV2BGR", lambda: device.cvt_color(yuyv, bgr2, _ffi.RCV_YUYV2BGR), 0.3),
    ]
    # a packed 1080-pixel-wide (portrait) BGR batch: rows are only 4-byte aligned, so the strip kernel does not apply -- the
    # integer filters must take the streaming kernel's exact integer mode, not the generic per-sample kernel (13x slower)
    pw, pw2 = device.DeviceBatch(ctx, n, 1920, 1080, 3), device.DeviceBatch(ctx, n, 1920, 1080, 3)
    device.synth(pw, 1, 10, 0)
    cases += [
        ("filter2D 7x7 i8, packed 1080-wide BGR (streaming kernel, integer mode)", lambda: device.filter2d(pw, pw2, k7, shift=6), 0.4),
        ("GaussianBlur 5x5 int, packed 1080-wide BGR", lambda: device.gaussian_blur(pw, pw2, 5, 0.0), 0.25),
    ]
    slow = []
    for name, fn, budget in cases:
        ms = _ms_per_call(ctx, fn)
        if ms > budget:                       # one retry after a longer run-up (a cold or briefly busy device is not a dispatch bug)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                fn()
                ctx.sync()
            ms = _ms_per_call(ctx, fn, steps=12)
        print(f"{name:52s} {ms:7.3f} ms  (budget {budget})")
        if ms > budget:
            slow.append((name, round(ms, 3), budget))
    for b in (bgr, bgr2, gray, gray2, yuyv, dx, dy, resp, mask, small, pw, pw2):
        b.free()
    assert not slow, slow
