"""The fast kernels must be the ones that run: every op below has a slow generic fallback that produces the same bytes, so a
dispatch condition that silently stops matching would leave the parity tests green.  The library logs the name of every kernel
it launches (RCV_LAUNCH in rcv_internal.h; rcv__debug_kernels / rcv__debug_kernels_reset): each case runs the op on a
BASELINE-shaped batch and asserts WHICH kernel was dispatched.  Timings are printed for information only -- nothing here
asserts wall-clock."""
import ctypes as C

import numpy as np
import pytest

import rustcv_amd as rcv
from rustcv_amd import _ffi, device

pytestmark = pytest.mark.gpu


def _ms_per_call(ctx, fn, steps=4):
    L = _ffi.lib()
    fn()
    ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(steps):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / steps


def _kernels_of(ctx, fn):
    L = _ffi.lib()
    L.rcv__debug_kernels_reset()
    fn()
    ctx.sync()
    return L.rcv__debug_kernels().decode()


def test_fast_kernels_are_dispatched(ctx):
    n, rows, cols = 8, 2160, 3840
    B = lambda ch, depth=_ffi.RCV_8U, r=rows, c=cols: device.DeviceBatch(ctx, n, r, c, ch, depth)   # noqa: E731
    bgr, bgr2, gray, gray2, yuyv = B(3), B(3), B(1), B(1), B(2)
    dx, dy, resp, mask = B(1, _ffi.RCV_16S), B(1, _ffi.RCV_16S), B(1, _ffi.RCV_32F), B(1)
    small = B(3, r=540, c=960)
    gray16, gray16b = device.DeviceBatch(ctx, 16, rows, cols, 1), device.DeviceBatch(ctx, 16, rows, cols, 1)
    device.synth(gray16, 1, 12, 0)
    one = device.DeviceBatch(ctx, 1, 1080, 1920, 3)
    one2 = device.DeviceBatch(ctx, 1, 1080, 1920, 3)
    gone, gone2 = device.DeviceBatch(ctx, 1, 1080, 1920, 1), device.DeviceBatch(ctx, 1, 1080, 1920, 1)
    one4k, one4k2 = device.DeviceBatch(ctx, 1, 2160, 3840, 3), device.DeviceBatch(ctx, 1, 2160, 3840, 3)
    device.synth(bgr, 1, 7, 0)
    device.synth(gray, 1, 8, 0)
    device.synth(yuyv, 2, 9, 0)
    device.synth(one, 1, 11, 0)
    k7 = np.arange(49, dtype=np.int8).reshape(7, 7) - 24
    kf = np.full((7, 7), 1 / 49, np.float32)
    M = np.array([0.9925, -0.1219, 300.0, 0.1219, 0.9925, -200.0], np.float32)
    # (name, call, substring of the kernel that must have been launched)
    cases = [
        ("filter2D 7x7 BGR, 8 x 4K (row-streaming MFMA kernel, chained bands)", lambda: device.filter2d(bgr, bgr2, k7, shift=6), "k_filter_rows_chain<7"),
        ("GaussianBlur 5x5 int BGR, 8 x 4K (row-streaming MFMA kernel, chained bands)", lambda: device.gaussian_blur(bgr, bgr2, 5, 0.0), "k_filter_rows_chain<5"),
        ("filter2D 7x7 BGR, one 1080p frame (strip kernel, latency variant)", lambda: device.filter2d(one, one2, k7, shift=6), "k_filter7_mfma<0, 0, 0, true>"),
        ("filter2D 7x7 gray, 16 x 4K (row-streaming kernel, gray variant)", lambda: device.filter2d(gray16, gray16b, k7, shift=6), "k_filter_rows_mfma<KS, 3, 0, 0, 2>"),
        ("filter2D 7x7 gray, 8 x 4K (row-streaming kernel, gray variant, per-SIMD band plan)", lambda: device.filter2d(gray, gray2, k7, shift=6), "k_filter_rows_mfma<KS, 3, 0, 0, 2>"),
        ("filter2D 7x7 gray, one 1080p frame (strip kernel, gray variant, latency variant)", lambda: device.filter2d(gone, gone2, k7, shift=6), "k_filter7_mfma<0, 0, 2, true>"),
        ("GaussianBlur 5x5 int BGR, one 1080p frame (BASELINE config 2: register-window kernel)", lambda: device.gaussian_blur(one, one2, 5, 0.0), "k_gauss_rows<5, 3>"),
        ("GaussianBlur 5x5 int BGR, one 4K frame (row-streaming MFMA kernel, per-SIMD band plan)", lambda: device.gaussian_blur(one4k, one4k2, 5, 0.0), "k_filter_rows_mfma<"),
        ("GaussianBlur 7x7 int gray (strip kernel, gray variant, two tables)", lambda: device.gaussian_blur(gray, gray2, 7, 0.0), "k_filter7_mfma<0, 2, 2>"),
        ("GaussianBlur 7x7 int BGR, 8 x 4K (row-streaming kernel, two weight tables)", lambda: device.gaussian_blur(bgr, bgr2, 7, 0.0), "k_filter_rows_mfma<KS, 3, 0, KS == 7 ? kCentre7 : kAll>"),
        ("GaussianBlur 7x7 int BGR, one 1080p frame (two-table strip kernel)", lambda: device.gaussian_blur(one, one2, 7, 0.0), "k_filter7_mfma<0, 2>"),
        ("fused YUYV -> filter2D, 8 x 4K (row-streaming kernel, conversion in registers)", lambda: device.filter2d_yuyv(yuyv, bgr2, k7, shift=6), "k_filter_rows_mfma<KS, 3, 0, 0, 1>"),
        ("filter2D 7x7 f32 BGR (stream kernel)", lambda: device.filter2d(bgr, bgr2, kf), "k_filter_f32_stream<"),
        ("GaussianBlur sigma BGR (separable stream kernel)", lambda: device.gaussian_blur(bgr, bgr2, 7, 1.5), "k_filter_f32_stream<"),
        ("Sobel gray", lambda: device.sobel(gray, dx, dy), "k_sobel_rows<"),
        ("Sobel of BGR", lambda: device.sobel(bgr, dx, dy), "k_sobel_rows<0, true>"),
        ("Harris pipeline BGR", lambda: device.harris_pipeline(bgr, mask, None, 2, 0.04, 1e-4), "k_harris_fused<false, 0>"),
        ("Harris pipeline YUYV", lambda: device.harris_pipeline(yuyv, mask, None, 2, 0.04, 1e-4), "k_harris_fused<false, 1>"),
        ("Harris pipeline gray", lambda: device.harris_pipeline(gray, mask, None, 2, 0.04, 1e-4), "k_harris_fused<false, 2>"),
        ("cornerHarris gray", lambda: device.corner_harris(gray, resp, 2, 0.04), "k_harris_fused<true, 2, false>"),
        ("NMS 3x3", lambda: device.nms3x3(resp, mask, 1e-4), "k_nms3x3_rows"),
        ("warpAffine BGR", lambda: device.warp_affine(bgr, bgr2, M), "k_warp_affine_lds<3, false>"),
        ("warpAffine gray", lambda: device.warp_affine(gray, gray2, M), "k_warp_gray_lds4<4>"),
        ("resize BGR 4K -> 960x540 (box)", lambda: device.resize(bgr, small), "k_resize_box<4>"),
        ("fused warp -> 4x down-scale, 8 frames (row pieces staged through LDS)", lambda: device.warp_affine_resize(bgr, small, M, rows, cols), "k_warp_resize_stage<4"),
        ("cvtColor BGR2GRAY", lambda: device.cvt_color(bgr, gray2, _ffi.RCV_BGR2GRAY), "k_bgr2gray16"),
        ("cvtColor YUYV2BGR", lambda: device.cvt_color(yuyv, bgr2, _ffi.RCV_YUYV2BGR), "k_yuyv2bgr_vec"),
    ]
    # a packed 1080-pixel-wide (portrait) BGR batch: rows are only 4-byte aligned and the width is not a multiple of 16 -- the
    # row-streaming MFMA kernel takes widths that are a multiple of 4 (whatever the size of the launch: the strip kernel does not
    # apply); a one-channel 1084-wide image takes the streaming kernel's exact integer mode, never the per-sample kernel
    pw, pw2 = device.DeviceBatch(ctx, n, 1920, 1080, 3), device.DeviceBatch(ctx, n, 1920, 1080, 3)
    pg, pg2 = device.DeviceBatch(ctx, n, 1920, 1084, 1), device.DeviceBatch(ctx, n, 1920, 1084, 1)
    device.synth(pw, 1, 10, 0)
    device.synth(pg, 1, 15, 0)
    cases += [
        ("filter2D 7x7 i8, packed 1080-wide BGR (row-streaming MFMA kernel)", lambda: device.filter2d(pw, pw2, k7, shift=6), "k_filter_rows_mfma<"),
        ("GaussianBlur 5x5 int, packed 1080-wide BGR", lambda: device.gaussian_blur(pw, pw2, 5, 0.0), "k_filter_rows_mfma<"),
        ("filter2D 7x7 i8, packed 1084-wide gray", lambda: device.filter2d(pg, pg2, k7, shift=6), "k_filter_"),
    ]
    # odd widths of packed images (the reference's Mat::new gives step = cols * channels: rows are then only byte-aligned): the
    # streaming kernel's unaligned instantiation, the register-window kernels' ragged instantiations -- never the per-sample kernels
    ow, ow2 = device.DeviceBatch(ctx, n, 1079, 1919, 3), device.DeviceBatch(ctx, n, 1079, 1919, 3)
    og, og2 = device.DeviceBatch(ctx, n, 1079, 1919, 1), device.DeviceBatch(ctx, n, 1079, 1919, 1)
    odx, ody, om = device.DeviceBatch(ctx, n, 1079, 1919, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, n, 1079, 1919, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, n, 1079, 1919, 1)
    device.synth(ow, 1, 13, 0)
    device.synth(og, 1, 14, 0)
    kf5 = np.full((5, 5), 1 / 25, np.float32)
    cases += [
        ("filter2D 7x7 i8, packed 1919-wide BGR (odd width: the MFMA kernel's any-width instantiation)", lambda: device.filter2d(ow, ow2, k7, shift=6), "k_filter_rows_mfma<KS, 3, 0, 0, 3>"),
        ("GaussianBlur 5x5 int, packed 1919-wide BGR", lambda: device.gaussian_blur(ow, ow2, 5, 0.0), "k_filter_rows_mfma<KS, 3, 0, 0, 3>"),
        ("GaussianBlur 7x7 int, packed 1919-wide BGR (two weight tables)", lambda: device.gaussian_blur(ow, ow2, 7, 0.0), "k_filter_rows_mfma<KS, 3, 0, KS == 7 ? kCentre7 : kAll, 3>"),
        ("GaussianBlur 7x7 sigma 1.5, packed 1919-wide BGR", lambda: device.gaussian_blur(ow, ow2, 7, 1.5), "k_filter_f32_stream<"),
        ("filter2D 5x5 f32, packed 1919-wide BGR", lambda: device.filter2d(ow, ow2, kf5), "k_filter_f32_stream<"),
        ("filter2D 7x7 i8, packed 1919-wide gray", lambda: device.filter2d(og, og2, k7, shift=6), "k_filter_gray_dot4<"),
        ("Sobel, packed 1919-wide gray", lambda: device.sobel(og, odx, ody), "k_sobel_rows<0, false, true>"),
        ("Harris pipeline, packed 1919-wide BGR", lambda: device.harris_pipeline(ow, om, None, 2, 0.04, 1e-4), "k_harris_fused<false, 0, true, true>"),
    ]
    oddsz = device.DeviceBatch(ctx, n, 1441, 2561, 3)
    cases += [
        ("resize BGR 4K -> 2561x1441 (odd destination width)", lambda: device.resize(bgr, oddsz), "k_resize_bgr"),
        ("resize packed 1919-wide BGR -> 960x540", lambda: device.resize(ow, small), "k_resize_bgr"),
        ("warpAffine between packed 1919-wide BGR images (byte-aligned rows)", lambda: device.warp_affine(ow, ow2, M), "k_warp_affine_lds<3, true>"),
        ("cvtColor BGR2GRAY, packed 1919-wide", lambda: device.cvt_color(ow, og2, _ffi.RCV_BGR2GRAY), "k_bgr2gray"),
    ]
    # block sizes other than 2: streaming Sobel + the register-window response kernel + streaming NMS
    cases += [
        ("cornerHarris blockSize 3 (gray): one launch", lambda: device.corner_harris(gray, resp, 3, 0.04), "k_harris_blocks_fused<"),
        ("Harris pipeline blockSize 5 (BGR): one launch", lambda: device.harris_pipeline(bgr, mask, None, 5, 0.04, 1e-4), "k_harris_blocks_fused<"),
        ("Harris pipeline blockSize 3, packed 1919-wide BGR: Sobel planes + window kernel", lambda: device.harris_pipeline(ow, om, None, 3, 0.04, 1e-4), "k_harris_resp_rows<"),
    ]
    wrong = []
    for name, fn, want in cases:
        launched = _kernels_of(ctx, fn)
        ms = _ms_per_call(ctx, fn)
        print(f"{name:72s} {ms:7.3f} ms   {launched}")
        if want not in launched or "generic" in launched:
            wrong.append((name, want, launched))
    for b in (bgr, bgr2, gray, gray2, yuyv, dx, dy, resp, mask, small, pw, pw2, one, one2, gone, gone2, one4k, one4k2, gray16, gray16b, ow, ow2, og, og2, odx, ody, om, pg, pg2, oddsz):
        b.free()
    assert not wrong, wrong


def test_row_streaming_kernel_by_size(ctx, knob):
    """the row-streaming kernel takes launches that fill the GPU; RCV_F7_ROWS=0 / 1 force the strip kernel / the row kernel"""
    k7 = np.arange(49, dtype=np.int8).reshape(7, 7) - 24
    big_s, big_d = device.DeviceBatch(ctx, 8, 2160, 3840, 3), device.DeviceBatch(ctx, 8, 2160, 3840, 3)
    small_s, small_d = device.DeviceBatch(ctx, 1, 480, 640, 3), device.DeviceBatch(ctx, 1, 480, 640, 3)
    device.synth(big_s, 0, 1, 0)
    device.synth(small_s, 0, 2, 0)
    assert "k_filter_rows_chain<" in _kernels_of(ctx, lambda: device.filter2d(big_s, big_d, k7, shift=6))   # (whole frames per XCD: chained bands)
    c = big_d.download().copy()
    knob("RCV_FR_CHAIN", 0)
    assert "k_filter_rows_mfma<" in _kernels_of(ctx, lambda: device.filter2d(big_s, big_d, k7, shift=6))
    assert np.array_equal(c, big_d.download())
    assert "k_filter7_mfma<" in _kernels_of(ctx, lambda: device.filter2d(small_s, small_d, k7, shift=6))
    a = big_d.download().copy()
    knob("RCV_F7_ROWS", 0)
    assert "k_filter7_mfma<" in _kernels_of(ctx, lambda: device.filter2d(big_s, big_d, k7, shift=6))
    assert np.array_equal(a, big_d.download())          # both kernels: the same bytes
    b = small_d.download().copy()
    knob("RCV_F7_ROWS", 1)
    assert "k_filter_rows_mfma<" in _kernels_of(ctx, lambda: device.filter2d(small_s, small_d, k7, shift=6))
    assert np.array_equal(b, small_d.download())
    for x in (big_s, big_d, small_s, small_d):
        x.free()
