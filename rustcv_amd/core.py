"""Host-side mirror of rustcv::core::mat::Mat (reference rustcv/src/core/mat.rs:6-53).

`data` is a 1-D numpy uint8 array standing in for `Vec<u8>`; rows/cols/step/channels keep the
reference's meaning (step = bytes per row, >= cols*channels).  Same constructor names.
"""
import ctypes as C

import numpy as np

from . import _ffi

_DEPTH_DTYPE = {_ffi.RCV_8U: np.uint8, _ffi.RCV_16S: np.int16, _ffi.RCV_32F: np.float32}


class Mat:
    __slots__ = ("data", "rows", "cols", "step", "channels", "depth")

    def __init__(self, rows=0, cols=0, channels=0, depth=_ffi.RCV_8U, step=None, data=None):
        esz = np.dtype(_DEPTH_DTYPE[depth]).itemsize
        self.rows, self.cols, self.channels, self.depth = int(rows), int(cols), int(channels), depth
        self.step = int(step) if step is not None else self.cols * self.channels * esz
        if data is None:
            data = np.zeros(self.rows * self.step, dtype=np.uint8)  # mat.rs:18-29: zero-filled, step = cols*channels
        self.data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)

    # --- reference constructors / accessors (mat.rs:18-51) -------------------------------------
    @staticmethod
    def new(rows, cols, channels):
        return Mat(rows, cols, channels)

    @staticmethod
    def empty():
        return Mat(0, 0, 0)

    def is_empty(self):
        return self.data.size == 0 or self.rows == 0 or self.cols == 0

    def row_bytes(self, row):
        start = row * self.step
        return self.data[start:start + self.cols * self.channels]

    # --- numpy conveniences (not in the reference) ---------------------------------------------
    @staticmethod
    def from_array(a, step=None):
        """HxW or HxWxC array of uint8/int16/float32 -> Mat (optionally with a padded step)."""
        a = np.asarray(a)
        depth = {np.dtype(np.uint8): _ffi.RCV_8U, np.dtype(np.int16): _ffi.RCV_16S, np.dtype(np.float32): _ffi.RCV_32F}[a.dtype]
        if a.ndim == 2:
            a = a[:, :, None]
        rows, cols, ch = a.shape
        rowb = cols * ch * a.dtype.itemsize
        step = rowb if step is None else int(step)
        buf = np.zeros(rows * step, dtype=np.uint8)
        if rows and cols:
            buf.reshape(rows, step)[:, :rowb] = np.ascontiguousarray(a).view(np.uint8).reshape(rows, rowb)
        return Mat(rows, cols, ch, depth, step, buf)

    def to_array(self):
        """The pixel payload (padding dropped) as an HxWxC array (HxW when channels == 1)."""
        dt = np.dtype(_DEPTH_DTYPE[self.depth])
        rowb = self.cols * self.channels * dt.itemsize
        if self.rows == 0 or self.cols == 0:
            return np.zeros((self.rows, self.cols, self.channels), dt)
        a = self.data[: self.rows * self.step].reshape(self.rows, self.step)[:, :rowb]
        a = np.ascontiguousarray(a).view(dt).reshape(self.rows, self.cols, self.channels)
        return a[:, :, 0] if self.channels == 1 else a

    def clone(self):
        return Mat(self.rows, self.cols, self.channels, self.depth, self.step, self.data.copy())

    def _as_rcv(self):
        m = _ffi.rcv_mat()
        m.data = self.data.ctypes.data if self.data.size else None
        m.cap = self.data.size
        m.step = self.step
        m.rows, m.cols = self.rows, self.cols
        m.channels, m.depth, m.device, m.reserved = self.channels, self.depth, _ffi.RCV_HOST, 0
        return m

    def __repr__(self):
        return f"Mat(rows={self.rows}, cols={self.cols}, channels={self.channels}, step={self.step})"


class Context:
    """One (device, stream, staging workspace) handle -- `rcv_ctx`.  Not thread-safe; contexts on
    different devices are independent (reference FFI precedent: bridge.h:4-7)."""

    def __init__(self, device=0):
        L = _ffi.lib()
        h = C.c_void_p()
        _ffi.check(L.rcv_ctx_create(int(device), C.byref(h)), f"rcv_ctx_create(device={device})")
        self._h, self.device = h, int(device)

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError("Context already closed")
        return self._h

    def sync(self):
        _ffi.check(_ffi.lib().rcv_sync(self.handle), "rcv_sync")

    def close(self):
        if self._h is not None:
            _ffi.lib().rcv_ctx_destroy(self._h)
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


def device_count():
    n = C.c_int(0)
    rc = _ffi.lib().rcv_device_count(C.byref(n))
    return n.value if rc == 0 else 0


_default_ctx = None


def default_context():
    """Lazily created process-wide context on device $LOCAL_RANK (or 0)."""
    global _default_ctx
    if _default_ctx is None:
        import os
        _default_ctx = Context(int(os.environ.get("RUSTCV_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default_ctx
