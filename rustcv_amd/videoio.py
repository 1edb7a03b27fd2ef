"""The per-pixel leaf of rustcv::videoio::VideoCapture::read on the MI355X backend.

`yuyv_to_bgr` / `bgra_to_bgr` keep the reference's private-helper signatures
(rustcv/src/videoio/mod.rs:344,385: `(src: &[u8], dest: &mut [u8], width, height)`), including the
silent return when a buffer is short.  `decode_into` reproduces the FourCC dispatch and Mat sizing
of `read` (mod.rs:191-258) for the formats that are pure per-pixel work; MJPEG is a third-party
codec and out of scope.  Camera I/O is not part of this package.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .core import Mat, default_context


def fourcc(a, b, c, d):
    """rustcv-core/src/pixel_format.rs:10-12"""
    return ord(a) | (ord(b) << 8) | (ord(c) << 16) | (ord(d) << 24)


YUYV, BGRA, MJPEG = fourcc(*"YUYV"), fourcc(*"BGRA"), fourcc(*"MJPG")
RGB3, BGR3, BGR4 = fourcc(*"RGB3"), fourcc(*"BGR3"), fourcc(*"BGR4")


def _flat(buf):
    a = np.ascontiguousarray(buf, dtype=np.uint8).reshape(-1)
    m = _ffi.rcv_mat()
    m.data = a.ctypes.data if a.size else None
    m.cap, m.step, m.rows, m.cols = a.size, a.size, 1 if a.size else 0, a.size
    m.channels, m.depth, m.device, m.reserved = 1, _ffi.RCV_8U, _ffi.RCV_HOST, 0
    return a, m


def _convert(code, src, dest, width, height, ctx):
    if not (isinstance(dest, np.ndarray) and dest.dtype == np.uint8 and dest.flags.c_contiguous):
        raise TypeError("dest must be a contiguous uint8 array (it is written in place)")
    sa, sm = _flat(src)
    d = _ffi.rcv_mat()
    flat = dest.reshape(-1)
    d.data = flat.ctypes.data if flat.size else None
    d.cap, d.step, d.rows, d.cols = flat.size, width * 3, height, width
    d.channels, d.depth, d.device, d.reserved = 3, _ffi.RCV_8U, _ffi.RCV_HOST, 0
    h = (ctx or default_context()).handle
    return _ffi.check(_ffi.lib().rcv_cvt_color(h, code, C.byref(sm), C.byref(d)), "rcv_cvt_color")


def yuyv_to_bgr(src, dest, width, height, ctx=None):
    """mod.rs:344-371.  Returns True if converted, False on the reference's silent no-op."""
    return _convert(_ffi.RCV_YUYV2BGR, src, dest, width, height, ctx) == _ffi.RCV_OK


def bgra_to_bgr(src, dest, width, height, ctx=None):
    """mod.rs:385-399."""
    return _convert(_ffi.RCV_BGRA2BGR, src, dest, width, height, ctx) == _ffi.RCV_OK


def rgb_to_bgr(src, dest, ctx=None):
    """rustcv-camera/src/decode.rs:213-219 (whole pixels of the shorter buffer)."""
    return _convert(_ffi.RCV_RGB2BGR, src, dest, 0, 0, ctx) == _ffi.RCV_OK


def decode_into(mat: Mat, data, fcc: int, width: int, height: int, ctx=None):
    """The body of `VideoCapture::read` after a frame arrives (mod.rs:191-258): (re)size `mat` to a
    packed BGR frame, then dispatch on FourCC.  Unknown formats copy when the length matches."""
    target = width * height * 3
    if mat.data.size != target:
        mat.data = np.zeros(target, dtype=np.uint8)  # mod.rs:193-195
    mat.rows, mat.cols, mat.channels, mat.step, mat.depth = height, width, 3, width * 3, _ffi.RCV_8U
    code = C.c_int(0)
    rc = _ffi.lib().rcv_fourcc_to_code(fcc, C.byref(code))
    if fcc == MJPEG:
        raise NotImplementedError("MJPEG decode is a third-party codec (turbojpeg / image); out of scope")
    if rc == 0 and code.value in (_ffi.RCV_YUYV2BGR, _ffi.RCV_BGRA2BGR) and fcc != BGR4:
        _convert(code.value, data, mat.data, width, height, ctx)
    else:
        a = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        if a.size == target:  # mod.rs:253-257 "Assume RGB/BGR or Copy"
            mat.data[:] = a
    return True
