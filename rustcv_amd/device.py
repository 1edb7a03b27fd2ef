"""Device-resident frame batches: the measured path.

A `DeviceBatch` owns one hipMalloc'd block holding n equally shaped frames on ONE GPU; the batch
entry points of the C ABI run asynchronously on the owning context's stream.  Frames are independent
for every op, so multi-GPU = one Context + DeviceBatch per GPU over a frame range, no collective.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .core import Context, Mat

_ESZ = {_ffi.RCV_8U: 1, _ffi.RCV_16S: 2, _ffi.RCV_32F: 4}
_DT = {_ffi.RCV_8U: np.uint8, _ffi.RCV_16S: np.int16, _ffi.RCV_32F: np.float32}


class DeviceBatch:
    def __init__(self, ctx: Context, n, rows, cols, channels, depth=_ffi.RCV_8U, step=None, frame_stride=None, frame_cap=None):
        self.ctx, self.n, self.rows, self.cols, self.channels, self.depth = ctx, int(n), int(rows), int(cols), int(channels), depth
        rowb = self.cols * self.channels * _ESZ[depth]
        self.step = int(step) if step is not None else rowb
        self.frame_cap = int(frame_cap) if frame_cap is not None else self.rows * self.step
        fs = self.frame_cap if frame_stride is None else int(frame_stride)
        self.frame_stride = (fs + 255) // 256 * 256 if frame_stride is None else fs
        self.nbytes = max(self.n, 1) * self.frame_stride
        p = C.c_void_p()
        _ffi.check(_ffi.lib().rcv_malloc(ctx.handle, self.nbytes, C.byref(p)), "rcv_malloc")
        self.ptr = p

    def view(self, first, n):
        """frames [first, first + n) of this batch as a batch of their own (same memory; free() only the parent)"""
        assert 0 <= first and n >= 0 and first + n <= self.n
        v = object.__new__(DeviceBatch)
        v.__dict__.update(self.__dict__)
        v.n = int(n)
        v.ptr = C.c_void_p((self.ptr.value or 0) + int(first) * self.frame_stride)
        v.nbytes = max(v.n, 1) * self.frame_stride
        v._parent = self      # keeps the allocation alive; a view never frees it
        return v

    def as_rcv(self):
        b = _ffi.rcv_batch()
        m = b.frame0
        m.data, m.cap, m.step = self.ptr, self.frame_cap, self.step
        m.rows, m.cols = self.rows, self.cols
        m.channels, m.depth, m.device, m.reserved = self.channels, self.depth, _ffi.RCV_DEVICE, 0
        b.frame_stride, b.n, b.reserved = self.frame_stride, self.n, 0
        return b

    def upload_bytes(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1)
        assert raw.size <= self.nbytes
        _ffi.check(_ffi.lib().rcv_upload(self.ctx.handle, self.ptr, raw.ctypes.data, raw.size), "rcv_upload")

    def download_bytes(self):
        raw = np.empty(self.nbytes, dtype=np.uint8)
        _ffi.check(_ffi.lib().rcv_download(self.ctx.handle, raw.ctypes.data, self.ptr, raw.size), "rcv_download")
        return raw

    def upload(self, frames):
        """frames: array [n, rows, cols(, ch)] of the batch dtype."""
        a = np.asarray(frames, dtype=_DT[self.depth]).reshape(self.n, self.rows, self.cols * self.channels)
        rowb = self.cols * self.channels * _ESZ[self.depth]
        raw = np.zeros((self.n, self.frame_stride), dtype=np.uint8)
        v = raw[:, : self.rows * self.step].reshape(self.n, self.rows, self.step)
        v[:, :, :rowb] = a.view(np.uint8).reshape(self.n, self.rows, rowb)
        self.upload_bytes(raw)

    def download(self):
        raw = self.download_bytes()[: self.n * self.frame_stride].reshape(self.n, self.frame_stride)
        rowb = self.cols * self.channels * _ESZ[self.depth]
        v = raw[:, : self.rows * self.step].reshape(self.n, self.rows, self.step)[:, :, :rowb]
        a = np.ascontiguousarray(v).view(_DT[self.depth]).reshape(self.n, self.rows, self.cols, self.channels)
        return a[..., 0] if self.channels == 1 else a

    def download_frame(self, i):
        """one frame of the batch as [rows, cols(, ch)] (the other frames stay on the device)"""
        if not 0 <= i < self.n:
            raise IndexError(i)
        raw = np.empty(self.rows * self.step, dtype=np.uint8)
        _ffi.check(_ffi.lib().rcv_download(self.ctx.handle, raw.ctypes.data, self.ptr.value + i * self.frame_stride, raw.size), "rcv_download")
        rowb = self.cols * self.channels * _ESZ[self.depth]
        a = np.ascontiguousarray(raw.reshape(self.rows, self.step)[:, :rowb]).view(_DT[self.depth]).reshape(self.rows, self.cols, self.channels)
        return a[..., 0] if self.channels == 1 else a

    def memset(self, value=0):
        _ffi.check(_ffi.lib().rcv_memset(self.ctx.handle, self.ptr, value, self.nbytes), "rcv_memset")

    def free(self):
        if getattr(self, "_parent", None) is None and self.ptr is not None and self.ctx._h is not None:
            _ffi.lib().rcv_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ImportedBuffer:
    """A DMA-BUF mapped into the context's device address space without a copy (`rcv_import_dmabuf`; the consuming side of the
    reference's `AsDmaBuf::as_dmabuf_fd`, rustcv-core/src/frame.rs:58-65).  `.ptr` is device memory for `DeviceBatch`-style views;
    the caller keeps ownership of its fd."""

    def __init__(self, ctx: Context, fd: int, nbytes: int, offset: int = 0):
        self.ctx, self.nbytes = ctx, int(nbytes)
        h, p = C.c_void_p(), C.c_void_p()
        _ffi.check(_ffi.lib().rcv_import_dmabuf(ctx.handle, int(fd), int(offset), self.nbytes, C.byref(h), C.byref(p)), "rcv_import_dmabuf")
        self._h, self.ptr = h, p
        self._views = 0           # live DeviceBatch views of the mapping (as_batch)

    def as_batch(self, n, rows, cols, channels, depth=_ffi.RCV_8U, step=None, frame_stride=None):
        """view the imported bytes as n frames (no allocation: the view does not own the memory)"""
        b = DeviceBatch.__new__(DeviceBatch)
        b.ctx, b.n, b.rows, b.cols, b.channels, b.depth = self.ctx, int(n), int(rows), int(cols), int(channels), depth
        rowb = b.cols * b.channels * _ESZ[depth]
        b.step = int(step) if step is not None else rowb
        b.frame_cap = b.rows * b.step
        b.frame_stride = int(frame_stride) if frame_stride is not None else b.frame_cap
        b.nbytes = max(b.n, 1) * b.frame_stride
        if b.nbytes > self.nbytes:
            raise ValueError("view larger than the imported buffer")
        b.ptr = self.ptr
        b.free = lambda: None     # the import owns the mapping
        b._owner = self           # ... and the view keeps the import alive: dropping `imp` must not unmap memory a view still points at
        self._views += 1
        import weakref
        weakref.finalize(b, ImportedBuffer._view_gone, self)
        return b

    @staticmethod
    def _view_gone(imp):
        imp._views -= 1

    def release(self):
        """unmap.  CONTRACT (since round 3): raises RuntimeError while views made by as_batch are alive -- they would be dangling device
        pointers -- so drop (or `del`) the views first: `view = imp.as_batch(...); ...; del view; imp.release()`."""
        if self._h is not None and getattr(self, "_views", 0) > 0:
            raise RuntimeError(f"ImportedBuffer.release(): {self._views} view(s) of the mapping are still alive")
        if self._h is not None:
            _ffi.lib().rcv_import_release(self._h)
            self._h, self.ptr = None, None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _h(b):
    return b.ctx.handle


def synth(dst: DeviceBatch, family, seed, frame_base=0):
    b = dst.as_rcv()
    _ffi.check(_ffi.lib().rcv_synth_batch(_h(dst), C.byref(b), family, seed, frame_base), "rcv_synth_batch")


def cvt_color(src, dst, code):
    a, b = src.as_rcv(), dst.as_rcv()
    return _ffi.check(_ffi.lib().rcv_cvt_color_batch(_h(src), code, C.byref(a), C.byref(b)), "rcv_cvt_color_batch")


def rectangle(mats, rect, color, thickness):
    b = mats.as_rcv()
    _ffi.check(_ffi.lib().rcv_rectangle_batch(_h(mats), C.byref(b), rect.x, rect.y, rect.width, rect.height,
                                              color.v0, color.v1, color.v2, thickness), "rcv_rectangle_batch")


def blend_glyphs(mats, glyphs, color):
    """put_text's blend (drawing.rs:137-160) of the same rasterised glyphs into every frame of a resident batch."""
    tbl, n, cov = _ffi.pack_glyphs(glyphs)
    b = mats.as_rcv()
    _ffi.check(_ffi.lib().rcv_blend_glyphs_batch(_h(mats), C.byref(b), tbl, n, cov.ctypes.data_as(C.POINTER(C.c_float)), cov.size,
                                                 color.v0, color.v1, color.v2), "rcv_blend_glyphs_batch")


def gaussian_blur(src, dst, ksize, sigma=0.0):
    a, b = src.as_rcv(), dst.as_rcv()
    _ffi.check(_ffi.lib().rcv_gaussian_blur_batch(_h(src), C.byref(a), C.byref(b), ksize, float(sigma)), "rcv_gaussian_blur_batch")


def filter2d(src, dst, kernel, shift=0, delta=0.0):
    k = np.ascontiguousarray(kernel)
    a, b = src.as_rcv(), dst.as_rcv()
    if k.dtype == np.int8:
        _ffi.check(_ffi.lib().rcv_filter2d_i8_batch(_h(src), C.byref(a), C.byref(b), k.ctypes.data_as(C.POINTER(C.c_int8)),
                                                    k.shape[0], shift), "rcv_filter2d_i8_batch")
    elif k.dtype == np.float32:
        _ffi.check(_ffi.lib().rcv_filter2d_f32_batch(_h(src), C.byref(a), C.byref(b), k.ctypes.data_as(C.POINTER(C.c_float)),
                                                     k.shape[0], float(delta)), "rcv_filter2d_f32_batch")
    else:
        raise TypeError("kernel dtype must be int8 or float32")


def filter2d_yuyv(src_yuyv, dst_bgr, kernel, shift=0):
    """fused capture pipeline: YUYV (2-channel batch) -> BGR -> integer filter2D in one launch"""
    k = np.ascontiguousarray(kernel, dtype=np.int8)
    a, b = src_yuyv.as_rcv(), dst_bgr.as_rcv()
    _ffi.check(_ffi.lib().rcv_filter2d_i8_yuyv_batch(_h(src_yuyv), C.byref(a), C.byref(b), k.ctypes.data_as(C.POINTER(C.c_int8)),
                                                     k.shape[0], shift), "rcv_filter2d_i8_yuyv_batch")


def filter2d_sobel(src_bgr, dx, dy, kernel, shift=0):
    """fused integer filter2D -> BGR2GRAY -> Sobel of a BGR batch: the i16 gradients of the filtered image in one launch"""
    k = np.ascontiguousarray(kernel, dtype=np.int8)
    a, b, c = src_bgr.as_rcv(), dx.as_rcv(), dy.as_rcv()
    _ffi.check(_ffi.lib().rcv_filter2d_i8_sobel_batch(_h(src_bgr), C.byref(a), C.byref(b), C.byref(c), k.ctypes.data_as(C.POINTER(C.c_int8)),
                                                      k.shape[0], shift), "rcv_filter2d_i8_sobel_batch")


def sobel(src, dx, dy):
    a, b, c = src.as_rcv(), dx.as_rcv(), dy.as_rcv()
    _ffi.check(_ffi.lib().rcv_sobel_batch(_h(src), C.byref(a), C.byref(b), C.byref(c)), "rcv_sobel_batch")


def resize(src, dst):
    a, b = src.as_rcv(), dst.as_rcv()
    _ffi.check(_ffi.lib().rcv_resize_batch(_h(src), C.byref(a), C.byref(b)), "rcv_resize_batch")


def warp_affine(src, dst, M):
    m = np.ascontiguousarray(M, dtype=np.float32).reshape(6)
    a, b = src.as_rcv(), dst.as_rcv()
    _ffi.check(_ffi.lib().rcv_warp_affine_batch(_h(src), C.byref(a), C.byref(b), m.ctypes.data_as(C.POINTER(C.c_float))),
               "rcv_warp_affine_batch")


def warp_affine_resize(src, dst, M, mid_rows, mid_cols):
    """resize(warp_affine(src -> mid_rows x mid_cols), dst); fused when mid is exactly 2x / 4x dst (BGR)."""
    m = np.ascontiguousarray(M, dtype=np.float32).reshape(6)
    a, b = src.as_rcv(), dst.as_rcv()
    _ffi.check(_ffi.lib().rcv_warp_affine_resize_batch(_h(src), C.byref(a), C.byref(b), m.ctypes.data_as(C.POINTER(C.c_float)),
                                                       int(mid_rows), int(mid_cols)), "rcv_warp_affine_resize_batch")


def corner_harris(gray, resp, block_size=2, k=0.04):
    a, b = gray.as_rcv(), resp.as_rcv()
    _ffi.check(_ffi.lib().rcv_corner_harris_batch(_h(gray), C.byref(a), C.byref(b), block_size, float(k)), "rcv_corner_harris_batch")


def nms3x3(resp, mask, thr):
    a, b = resp.as_rcv(), mask.as_rcv()
    _ffi.check(_ffi.lib().rcv_nms3x3_batch(_h(resp), C.byref(a), C.byref(b), float(thr)), "rcv_nms3x3_batch")


def harris_pipeline(bgr, mask, resp=None, block_size=2, k=0.04, thr=0.0):
    a, m = bgr.as_rcv(), mask.as_rcv()
    r = resp.as_rcv() if resp is not None else None
    _ffi.check(_ffi.lib().rcv_harris_pipeline_batch(_h(bgr), C.byref(a), C.byref(m), C.byref(r) if r is not None else None,
                                                    block_size, float(k), float(thr)), "rcv_harris_pipeline_batch")
