"""rustcv::imgproc on the MI355X backend.

`Point`, `Rect`, `Scalar` and `rectangle` mirror the reference one to one
(rustcv/src/imgproc/drawing.rs:8-106: same names, argument order and clipping/guard behaviour).
Everything else in this module (gaussian_blur ... harris_pipeline) does NOT exist in the reference
(SURVEY.md F1); names follow the OpenCV functions the reference says it wants parity with, and the
semantics are SURVEY.md 8-A.  All functions take host `Mat`s (upload -> HIP kernel -> download);
the device-resident batch forms are in rustcv_amd.device.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from .core import Mat, default_context


@dataclass(frozen=True)
class Point:  # drawing.rs:8-17
    x: int
    y: int


@dataclass(frozen=True)
class Rect:  # drawing.rs:20-37
    x: int
    y: int
    width: int
    height: int


@dataclass(frozen=True)
class Scalar:  # drawing.rs:39-60 (v0 = blue, v1 = green, v2 = red)
    v0: int
    v1: int
    v2: int

    @staticmethod
    def all(v):
        return Scalar(v, v, v)


def _ctx(ctx):
    return (ctx or default_context()).handle


def rectangle(mat: Mat, rect: Rect, color: Scalar, thickness: int, ctx=None):
    """In-place rectangle outline, border grown inward (drawing.rs:67-106)."""
    m = mat._as_rcv()
    _ffi.check(_ffi.lib().rcv_rectangle(_ctx(ctx), C.byref(m), rect.x, rect.y, rect.width, rect.height,
                                        color.v0, color.v1, color.v2, thickness), "rcv_rectangle")


def blend_glyphs(mat: Mat, glyphs, color: Scalar, ctx=None):
    """The per-pixel half of `put_text` (drawing.rs:123-163): alpha-blend rasterised glyphs into the Mat, in order.

    `glyphs`: iterable of `(min_x, min_y, coverage)` -- what the reference's loop has in hand for each positioned glyph:
    `glyph.pixel_bounding_box().min` (:134) and the values `glyph.draw` yields (:136) as an (h, w) float32 array.  Layout
    and rasterisation (rusttype on a font blob, third-party) stay with the caller."""
    tbl, n, cov = _ffi.pack_glyphs(glyphs)
    m = mat._as_rcv()
    _ffi.check(_ffi.lib().rcv_blend_glyphs(_ctx(ctx), C.byref(m), tbl, n, cov.ctypes.data_as(C.POINTER(C.c_float)), cov.size,
                                           color.v0, color.v1, color.v2), "rcv_blend_glyphs")


# ---- build-defined ops ----------------------------------------------------------------------------

def cvt_color(src: Mat, dst: Mat, code: int, ctx=None):
    s, d = src._as_rcv(), dst._as_rcv()
    return _ffi.check(_ffi.lib().rcv_cvt_color(_ctx(ctx), code, C.byref(s), C.byref(d)), "rcv_cvt_color")


def gaussian_blur(src: Mat, dst: Mat, ksize: int, sigma: float = 0.0, ctx=None):
    s, d = src._as_rcv(), dst._as_rcv()
    _ffi.check(_ffi.lib().rcv_gaussian_blur(_ctx(ctx), C.byref(s), C.byref(d), ksize, float(sigma)), "rcv_gaussian_blur")


def filter2d(src: Mat, dst: Mat, kernel, shift: int = 0, delta: float = 0.0, ctx=None):
    """int8 kernel -> integer path (`shift` = power-of-two normaliser); float32 kernel -> f32 path (`delta`)."""
    k = np.ascontiguousarray(kernel)
    if k.ndim != 2 or k.shape[0] != k.shape[1]:
        raise ValueError("kernel must be square")
    s, d = src._as_rcv(), dst._as_rcv()
    if k.dtype == np.int8:
        _ffi.check(_ffi.lib().rcv_filter2d_i8(_ctx(ctx), C.byref(s), C.byref(d), k.ctypes.data_as(C.POINTER(C.c_int8)),
                                              k.shape[0], shift), "rcv_filter2d_i8")
    elif k.dtype == np.float32:
        _ffi.check(_ffi.lib().rcv_filter2d_f32(_ctx(ctx), C.byref(s), C.byref(d), k.ctypes.data_as(C.POINTER(C.c_float)),
                                               k.shape[0], float(delta)), "rcv_filter2d_f32")
    else:
        raise TypeError("kernel dtype must be int8 or float32")


def filter2d_yuyv(src_yuyv: Mat, dst_bgr: Mat, kernel, shift: int = 0, ctx=None):
    """fused YUYV -> BGR -> integer filter2D (== cvt_color(YUYV2BGR_STRIDED) then filter2d, in one launch)"""
    k = np.ascontiguousarray(kernel, dtype=np.int8)
    s, d = src_yuyv._as_rcv(), dst_bgr._as_rcv()
    _ffi.check(_ffi.lib().rcv_filter2d_i8_yuyv(_ctx(ctx), C.byref(s), C.byref(d), k.ctypes.data_as(C.POINTER(C.c_int8)), k.shape[0], shift),
               "rcv_filter2d_i8_yuyv")


def filter2d_sobel(src_bgr: Mat, dx: Mat, dy: Mat, kernel, shift: int = 0, ctx=None):
    """fused integer filter2D -> BGR2GRAY -> Sobel (== filter2d then sobel on its output, in one launch)"""
    k = np.ascontiguousarray(kernel, dtype=np.int8)
    s, a, b = src_bgr._as_rcv(), dx._as_rcv(), dy._as_rcv()
    _ffi.check(_ffi.lib().rcv_filter2d_i8_sobel(_ctx(ctx), C.byref(s), C.byref(a), C.byref(b), k.ctypes.data_as(C.POINTER(C.c_int8)), k.shape[0], shift),
               "rcv_filter2d_i8_sobel")


def sobel(src: Mat, dx: Mat, dy: Mat, ctx=None):
    s, a, b = src._as_rcv(), dx._as_rcv(), dy._as_rcv()
    _ffi.check(_ffi.lib().rcv_sobel(_ctx(ctx), C.byref(s), C.byref(a), C.byref(b)), "rcv_sobel")


def resize(src: Mat, dst: Mat, ctx=None):
    s, d = src._as_rcv(), dst._as_rcv()
    _ffi.check(_ffi.lib().rcv_resize(_ctx(ctx), C.byref(s), C.byref(d)), "rcv_resize")


def warp_affine(src: Mat, dst: Mat, M, ctx=None):
    m = np.ascontiguousarray(M, dtype=np.float32).reshape(6)
    s, d = src._as_rcv(), dst._as_rcv()
    _ffi.check(_ffi.lib().rcv_warp_affine(_ctx(ctx), C.byref(s), C.byref(d), m.ctypes.data_as(C.POINTER(C.c_float))),
               "rcv_warp_affine")


def warp_affine_resize(src: Mat, dst: Mat, M, mid_rows: int, mid_cols: int, ctx=None):
    """resize(warp_affine(src -> mid_rows x mid_cols), dst) in one call (fused for exact 2x / 4x BGR down-scales)."""
    m = np.ascontiguousarray(M, dtype=np.float32).reshape(6)
    s, d = src._as_rcv(), dst._as_rcv()
    _ffi.check(_ffi.lib().rcv_warp_affine_resize(_ctx(ctx), C.byref(s), C.byref(d), m.ctypes.data_as(C.POINTER(C.c_float)),
                                                 int(mid_rows), int(mid_cols)), "rcv_warp_affine_resize")


def corner_harris(gray: Mat, resp: Mat, block_size: int = 2, k: float = 0.04, ctx=None):
    s, d = gray._as_rcv(), resp._as_rcv()
    _ffi.check(_ffi.lib().rcv_corner_harris(_ctx(ctx), C.byref(s), C.byref(d), block_size, float(k)), "rcv_corner_harris")


def nms3x3(resp: Mat, mask: Mat, thr: float, ctx=None):
    s, d = resp._as_rcv(), mask._as_rcv()
    _ffi.check(_ffi.lib().rcv_nms3x3(_ctx(ctx), C.byref(s), C.byref(d), float(thr)), "rcv_nms3x3")


def harris_pipeline(bgr: Mat, mask: Mat, resp: Mat = None, block_size: int = 2, k: float = 0.04, thr: float = 0.0, ctx=None):
    s, m = bgr._as_rcv(), mask._as_rcv()
    r = resp._as_rcv() if resp is not None else None
    _ffi.check(_ffi.lib().rcv_harris_pipeline(_ctx(ctx), C.byref(s), C.byref(m), C.byref(r) if r is not None else None,
                                              block_size, float(k), float(thr)), "rcv_harris_pipeline")
