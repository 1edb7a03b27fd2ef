"""Zero-copy DMA-BUF import (rcv_import_dmabuf): the consuming side of the reference's declared `AsDmaBuf::as_dmabuf_fd`
(rustcv-core/src/frame.rs:58-65).  No capture backend is running here, so the producer is the GPU runtime itself: a device
allocation is exported as a DMA-BUF fd (hsa_amd_portable_export_dmabuf), imported back through the ABI, and the two views of the
same memory are held against each other -- kernels read the imported mapping and write through it, the caller's fd stays open and
usable after the import is released.  If this box cannot export a DMA-BUF at all the producer half is skipped with the reason."""
import ctypes as C
import os

import numpy as np
import pytest

import rustcv_amd as rcv
from rustcv_amd import _ffi, device

pytestmark = pytest.mark.gpu


def _export_dmabuf(ptr, nbytes):
    try:
        hsa = C.CDLL("libhsa-runtime64.so.1", mode=C.RTLD_GLOBAL)
    except OSError:
        hsa = C.CDLL("/opt/rocm/lib/libhsa-runtime64.so.1", mode=C.RTLD_GLOBAL)
    fn = hsa.hsa_amd_portable_export_dmabuf
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    fd, off = C.c_int(-1), C.c_uint64(0)
    st = fn(ptr, nbytes, C.byref(fd), C.byref(off))
    if st != 0 or fd.value < 0:
        pytest.skip(f"this box cannot export a DMA-BUF (hsa_amd_portable_export_dmabuf status {st})")
    return fd.value, off.value, hsa


def test_import_dmabuf_argument_checks(ctx):
    L = _ffi.lib()
    h, p = C.c_void_p(), C.c_void_p()
    assert L.rcv_import_dmabuf(ctx.handle, -1, 0, 4096, C.byref(h), C.byref(p)) == _ffi.RCV_ERR_ARG
    assert L.rcv_import_dmabuf(ctx.handle, 0, 0, 0, C.byref(h), C.byref(p)) == _ffi.RCV_ERR_ARG
    assert L.rcv_import_dmabuf(ctx.handle, 0, 0, 4096, None, C.byref(p)) == _ffi.RCV_ERR_ARG
    assert L.rcv_import_dmabuf(ctx.handle, 0, 2**64 - 1, 4096, C.byref(h), C.byref(p)) == _ffi.RCV_ERR_SIZE   # offset + bytes wraps
    r, w = os.pipe()                                # a descriptor that is no DMA-BUF: a clean error, nothing leaked
    rc = L.rcv_import_dmabuf(ctx.handle, r, 0, 4096, C.byref(h), C.byref(p))
    assert rc in (_ffi.RCV_ERR_DEVICE, _ffi.RCV_ERR_OOM) and not h.value and not p.value
    os.close(r)
    os.close(w)
    L.rcv_import_release(None)
    ctx.sync()                                      # the context is still healthy


def test_import_dmabuf_round_trip(ctx, oracle, rng):
    n, rows, cols = 2, 96, 256
    src = device.DeviceBatch(ctx, n, rows, cols, 3)             # the "capture buffer": a device allocation exported as a DMA-BUF
    frames = rng.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    src.upload(frames)
    fd, off, hsa = _export_dmabuf(src.ptr, src.nbytes)
    imp = device.ImportedBuffer(ctx, fd, src.nbytes, off)      # (the runtime sub-allocates: the buffer sits at `off` inside the DMA-BUF)
    assert imp.ptr.value and imp.ptr.value != src.ptr.value      # a second mapping of the same memory
    view = imp.as_batch(n, rows, cols, 3, frame_stride=src.frame_stride)
    gray = device.DeviceBatch(ctx, n, rows, cols, 1)
    device.cvt_color(view, gray, _ffi.RCV_BGR2GRAY)              # a kernel reads the imported mapping
    got = gray.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.bgr2gray(frames[i]))
    k = rng.integers(-9, 10, size=(7, 7)).astype(np.int8)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.filter2d(src, dst, k, shift=5)
    device.filter2d(dst, view, k, shift=5)                       # ... and a kernel writes through it
    ctx.sync()
    back = src.download()                                        # read through the ORIGINAL allocation
    for i in range(n):
        assert np.array_equal(back[i], oracle.filter2d_i8(oracle.filter2d_i8(frames[i], k, 5), k, 5))
    with pytest.raises(RuntimeError):
        imp.release()                                            # a live view would become a dangling device pointer
    del view
    imp.release()
    os.fstat(fd)                                                 # the caller's fd was not closed by the import
    hsa.hsa_amd_portable_close_dmabuf(fd)
    for b in (src, gray, dst):
        b.free()


def test_import_view_keeps_the_mapping_alive_and_size_is_checked(ctx, oracle, rng):
    """round-2 advisor findings: (1) `ImportedBuffer(...).as_batch(...)` with the import itself dropped must stay valid -- the view
    owns a reference; (2) a DMA-BUF smaller than offset + bytes is refused instead of mapped past its end"""
    n, rows, cols = 1, 64, 256
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    frames = rng.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    src.upload(frames)
    fd, off, hsa = _export_dmabuf(src.ptr, src.nbytes)
    view = device.ImportedBuffer(ctx, fd, src.nbytes, off).as_batch(n, rows, cols, 3, frame_stride=src.frame_stride)   # the import is a temporary
    import gc
    gc.collect()
    gray = device.DeviceBatch(ctx, n, rows, cols, 1)
    device.cvt_color(view, gray, _ffi.RCV_BGR2GRAY)
    assert np.array_equal(gray.download()[0], oracle.bgr2gray(frames[0]))
    del view                                                     # ... and the mapping goes with its last view
    gc.collect()
    total = os.lseek(fd, 0, os.SEEK_END)
    L = _ffi.lib()
    h, p = C.c_void_p(), C.c_void_p()
    if total > 0:                                                # the exporter reports its size: one byte too many is refused
        assert L.rcv_import_dmabuf(ctx.handle, fd, 0, total + 1, C.byref(h), C.byref(p)) == _ffi.RCV_ERR_SIZE and not h.value and not p.value
        assert L.rcv_import_dmabuf(ctx.handle, fd, total, 1, C.byref(h), C.byref(p)) == _ffi.RCV_ERR_SIZE
    os.fstat(fd)
    hsa.hsa_amd_portable_close_dmabuf(fd)
    src.free()
    gray.free()


def test_import_keeps_its_context_alive():
    c = rcv.Context(0)
    buf = device.DeviceBatch(c, 1, 64, 64, 1)
    fd, off, hsa = _export_dmabuf(buf.ptr, buf.nbytes)
    imp = device.ImportedBuffer(c, fd, buf.nbytes, off)
    buf.free()
    c.close()          # deferred: the import still uses the context's device
    imp.release()      # the context goes here
    hsa.hsa_amd_portable_close_dmabuf(fd)
    c2 = rcv.Context(0)
    c2.sync()
    c2.close()
