"""GPU tests of launch graphs (rcv_graph_*): a recorded chain replays with the same results as direct calls, on fresh buffer
contents, with its own copy of per-call constants; entry points that must synchronise refuse to be recorded."""
import ctypes as C

import numpy as np
import pytest

import rustcv_amd as rcv
from rustcv_amd import _ffi, device, imgproc
from rustcv_amd.core import Mat
from rustcv_amd.imgproc import Rect, Scalar

pytestmark = pytest.mark.gpu


def test_graph_replays_reference_loop(ctx, oracle, rng):
    """config 0 chain (examples/camera_demo.rs:50-76): YUYV -> BGR, then rectangle, on 4 VGA frames; three replays on new frames"""
    n, rows, cols = 4, 480, 640
    src = device.DeviceBatch(ctx, n, rows, cols, 2)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    with ctx.capture() as g:
        device.cvt_color(src, dst, _ffi.RCV_YUYV2BGR)
        device.rectangle(dst, Rect(200, 150, 240, 240), Scalar(0, 255, 0), 2)
    for it in range(3):
        frames = rng.integers(0, 256, size=(n, rows, cols, 2), dtype=np.uint8)
        src.upload(frames)
        g.launch()
        got = dst.download()
        for i in range(n):
            want = np.zeros(rows * cols * 3, np.uint8)
            assert oracle.yuyv_to_bgr(frames[i].reshape(-1), want, cols, rows)
            oracle.rectangle(want, rows, cols, cols * 3, 200, 150, 240, 240, 0, 255, 0, 2)
            assert np.array_equal(got[i].reshape(-1), want)
    g.close()
    src.free()
    dst.free()


def test_graph_owns_its_filter_weights(ctx, oracle, rng):
    """the MFMA filter's weight table is copied into the graph: calls with other kernels between replays do not leak in"""
    rows, cols = 96, 256
    k1 = rng.integers(-9, 10, size=(7, 7)).astype(np.int8)
    k2 = rng.integers(-9, 10, size=(5, 5)).astype(np.int8)
    src = device.DeviceBatch(ctx, 2, rows, cols, 3)
    mid = device.DeviceBatch(ctx, 2, rows, cols, 3)
    gray = device.DeviceBatch(ctx, 2, rows, cols, 1)
    other = device.DeviceBatch(ctx, 2, rows, cols, 3)
    frames = rng.integers(0, 256, size=(2, rows, cols, 3), dtype=np.uint8)
    src.upload(frames)
    device.filter2d(src, other, k2, shift=4)          # warm the shared cache with a different kernel
    with ctx.capture() as g:
        device.filter2d(src, mid, k1, shift=6)
        device.cvt_color(mid, gray, _ffi.RCV_BGR2GRAY)
    for it in range(2):
        device.filter2d(src, other, k2, shift=4)      # overwrites the context's cached table
        g.launch()
        got = gray.download()
        for i in range(2):
            assert np.array_equal(got[i].reshape(rows, cols), oracle.bgr2gray(oracle.filter2d_i8(frames[i], k1, 6)))
        assert np.array_equal(other.download()[0], oracle.filter2d_i8(frames[0], k2, 4))
        frames = rng.integers(0, 256, size=(2, rows, cols, 3), dtype=np.uint8)
        src.upload(frames)
    g.close()
    for b in (src, mid, gray, other):
        b.free()


def test_graph_refuses_synchronising_calls(ctx, rng):
    L = _ffi.lib()
    img = rng.integers(0, 256, size=(16, 32, 3), dtype=np.uint8)
    d = device.DeviceBatch(ctx, 1, 16, 32, 3)
    d.upload(img[None])
    with ctx.capture() as g:
        with pytest.raises(rcv.RcvError) as e:           # host mats stage and synchronise
            imgproc.gaussian_blur(Mat.from_array(img), Mat(16, 32, 3), 3, 0.0, ctx)
        assert e.value.code == _ffi.RCV_ERR_UNSUPPORTED
        assert L.rcv_sync(ctx.handle) == _ffi.RCV_ERR_UNSUPPORTED
        assert L.rcv_graph_begin(ctx.handle) == _ffi.RCV_ERR_ARG      # no nesting
        with pytest.raises(rcv.RcvError) as e:           # per-call host tables (glyph boxes, coverage) cannot be replayed
            device.blend_glyphs(d, [(0, 0, np.ones((2, 2), np.float32))], imgproc.Scalar(1, 2, 3))
        assert e.value.code == _ffi.RCV_ERR_UNSUPPORTED
        device.gaussian_blur(d, d2 := device.DeviceBatch(ctx, 1, 16, 32, 3), 3, 0.0)
    g.launch()
    ctx.sync()
    assert L.rcv_graph_launch(ctx.handle, None) == _ffi.RCV_ERR_ARG
    gg = C.c_void_p()
    assert L.rcv_graph_end(ctx.handle, C.byref(gg)) == _ffi.RCV_ERR_ARG   # nothing being recorded
    g.close()
    d.free()
    d2.free()


def test_graph_owns_its_workspace(ctx, oracle, rng):
    """An op that goes through the context workspace (unfused Harris pipeline, block 3: gray + gradients + response through
    `ws`) is recorded; afterwards a LARGER call of the same kind makes the context free and re-allocate its grow-only workspace.
    The replay must not depend on that memory: a graph owns the workspace its ops used while they were recorded."""
    rows, cols = 64, 96
    src = device.DeviceBatch(ctx, 1, rows, cols, 3)
    mask = device.DeviceBatch(ctx, 1, rows, cols, 1)
    frame = rng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
    src.upload(frame[None])
    device.harris_pipeline(src, mask, None, 3, 0.04, 1e-6)            # run once: the context workspace exists at this size
    with ctx.capture() as g:
        device.harris_pipeline(src, mask, None, 3, 0.04, 1e-6)
    want = oracle.harris_pipeline(frame, 3, 0.04, 1e-6)
    for rounds in range(3):
        big_s = device.DeviceBatch(ctx, 2, 300 + 200 * rounds, 512, 3)       # grows ws: free + hipMalloc of the context buffer
        big_m = device.DeviceBatch(ctx, 2, 300 + 200 * rounds, 512, 1)
        device.synth(big_s, 1, 5 + rounds, 0)
        device.harris_pipeline(big_s, big_m, None, 3, 0.04, 1e-6)
        mask.memset(7)
        g.launch()
        assert np.array_equal(mask.download()[0], want), rounds
        big_s.free()
        big_m.free()
    g.close()
    src.free()
    mask.free()


def test_graph_and_ring_outlive_their_context():
    """rcv_ctx_destroy on a context with live graphs / rings only drains it; the children stay usable for destruction and the
    context's memory goes with the last of them (Python: Context.close() is explicit, Graph / StagingRing finalisers run later)"""
    c = rcv.Context(0)
    a, b = device.DeviceBatch(c, 1, 32, 64, 3), device.DeviceBatch(c, 1, 32, 64, 1)
    with c.capture() as g:
        device.cvt_color(a, b, _ffi.RCV_BGR2GRAY)
    g.launch()
    c.sync()
    ring = rcv.StagingRing(c, 2, (32, 64, 3), (32, 64, 1))
    a.free()
    b.free()
    c.close()            # children alive: deferred
    g.close()            # must not touch freed memory
    ring.close()         # last child: the context is finalised here
    c2 = rcv.Context(0)  # the device is still usable
    c2.sync()
    c2.close()


def test_free_is_refused_while_recording(ctx):
    L = _ffi.lib()
    d = device.DeviceBatch(ctx, 1, 16, 32, 3)
    d2 = device.DeviceBatch(ctx, 1, 16, 32, 3)
    with ctx.capture() as g:
        assert L.rcv_free(ctx.handle, d.ptr) == _ffi.RCV_ERR_UNSUPPORTED     # would synchronise the recording stream
        device.gaussian_blur(d, d2, 3, 0.0)
    g.launch()
    ctx.sync()
    g.close()
    d.free()
    d2.free()
