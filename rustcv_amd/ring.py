"""Pinned-host staging ring (`rcv_ring`, include/rustcv_hip.h): `depth` frames in flight, H2D / kernels / D2H of
different frames overlapping.  The streaming counterpart of the reference's capture loop
(rustcv/src/videoio/mod.rs:83-112, `VideoCapture::read`), "next" row f3 of SURVEY.md 8(f).

    ring = StagingRing(ctx, 3, (2160, 3840, 2), (2160, 3840, 3))
    op = lambda c, din, dout: L.rcv_cvt_color(c, _ffi.RCV_YUYV2BGR_STRIDED, din, dout)
    for frame in frames:
        if ring.full():
            out = ring.retire()
        ring.submit(frame, op)
    while ring.in_flight(): out = ring.retire()

The op receives the raw context handle and two `rcv_mat` pointers describing DEVICE buffers; it must only enqueue work
on the context's stream (every rcv_* call does) and return the status code.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .core import Context

_DEPTH = {np.dtype(np.uint8): _ffi.RCV_8U, np.dtype(np.int16): _ffi.RCV_16S, np.dtype(np.float32): _ffi.RCV_32F}
_NP = {_ffi.RCV_8U: np.uint8, _ffi.RCV_16S: np.int16, _ffi.RCV_32F: np.float32}


def _host_mat(a: np.ndarray, depth):
    if a.ndim == 2:
        a = a[:, :, None]
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("frame must be C-contiguous")
    m = _ffi.rcv_mat()
    m.data = a.ctypes.data
    m.step = a.strides[0]
    m.cap = a.nbytes
    m.rows, m.cols, m.channels = a.shape
    m.depth, m.device = depth, _ffi.RCV_HOST
    return m


class StagingRing:
    def __init__(self, ctx: Context, depth, in_shape, out_shape, in_dtype=np.uint8, out_dtype=np.uint8):
        self._ctx = ctx
        self._in_shape, self._out_shape = tuple(in_shape), tuple(out_shape)
        self._in_depth, self._out_depth = _DEPTH[np.dtype(in_dtype)], _DEPTH[np.dtype(out_dtype)]
        h = C.c_void_p()
        _ffi.check(_ffi.lib().rcv_ring_create(ctx.handle, int(depth), *self._in_shape, self._in_depth, *self._out_shape, self._out_depth,
                                              C.byref(h)), "rcv_ring_create")
        self._h, self.depth = h, int(depth)
        self._cbs = []   # keep the ctypes thunks of in-flight ops alive

    def in_flight(self):
        return _ffi.lib().rcv_ring_in_flight(self._h)

    def full(self):
        return self.in_flight() >= self.depth

    def input_view(self):
        """numpy view (rows, cols, ch) of the pinned input buffer the next submit(None, op) uploads; row stride may exceed cols*ch"""
        m = _ffi.rcv_mat()
        _ffi.check(_ffi.lib().rcv_ring_input(self._h, C.byref(m)), "rcv_ring_input")
        dt = np.dtype(_NP[m.depth])
        buf = (C.c_uint8 * m.cap).from_address(m.data)
        rows = np.frombuffer(buf, dtype=np.uint8).reshape(m.rows, m.step)[:, : m.cols * m.channels * dt.itemsize]
        return rows.view(dt).reshape(m.rows, m.cols, m.channels)

    def submit(self, frame, op):
        """frame: ndarray (rows, cols, ch) or None (input_view() was filled in place); op(ctx_handle, dev_in_ptr, dev_out_ptr) -> status"""
        def thunk(c, din, dout, _user):
            try:
                return int(op(c, din, dout) or 0)
            except Exception:   # an exception must not cross the C boundary
                return _ffi.RCV_ERR_ARG
        cb = _ffi.RING_OP(thunk)
        self._cbs.append(cb)
        if len(self._cbs) > 2 * self.depth:
            self._cbs = self._cbs[-2 * self.depth:]
        if frame is None:
            rc = _ffi.lib().rcv_ring_submit(self._h, None, cb, None)
        else:
            a = np.ascontiguousarray(frame)
            m = _host_mat(a, self._in_depth)
            rc = _ffi.lib().rcv_ring_submit(self._h, C.byref(m), cb, None)
        _ffi.check(rc, "rcv_ring_submit")

    def retire(self, out=None, copy=True):
        """wait for the oldest frame; returns it as an ndarray: a copy, or -- copy=False -- a view of the ring's pinned buffer of
        the slot just retired.  That slot is the one the NEXT submit() may reuse (always, when the ring was full), so a view is
        valid only until the next submit(): consume it first."""
        r, c, ch = self._out_shape
        dt = np.dtype(_NP[self._out_depth])
        if not copy:
            pm = _ffi.rcv_mat()
            if _ffi.check(_ffi.lib().rcv_ring_retire(self._h, None, C.byref(pm)), "rcv_ring_retire") == _ffi.RCV_NOOP:
                raise IndexError("retire() on an empty ring")
            buf = (C.c_uint8 * pm.cap).from_address(pm.data)
            rows = np.frombuffer(buf, dtype=np.uint8).reshape(pm.rows, pm.step)[:, : c * ch * dt.itemsize]
            return rows.view(dt).reshape(r, c, ch)
        if out is None:
            out = np.empty((r, c, ch), dt)
        m = _host_mat(out, self._out_depth)
        rc = _ffi.check(_ffi.lib().rcv_ring_retire(self._h, C.byref(m), None), "rcv_ring_retire")
        if rc == _ffi.RCV_NOOP:
            raise IndexError("retire() on an empty ring")
        return out

    def close(self):
        if self._h is not None:
            _ffi.lib().rcv_ring_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
