"""GPU tests of the pinned-host staging ring ("next" row f3): frames streamed through H2D -> op -> D2H with `depth` in flight
must come back in order and equal to the oracle of the same op; capacity, empty and error behaviour of the C ABI."""
import ctypes as C

import numpy as np
import pytest

import rustcv_amd as rcv
from rustcv_amd import _ffi

pytestmark = pytest.mark.gpu


def _expect_yuyv_filter(oracle, frame, k, shift):
    rows, cols = frame.shape[:2]
    bgr = np.zeros(rows * cols * 3, np.uint8)
    oracle.yuv422_to_bgr_strided(frame.reshape(-1), cols * 2, rows, cols, False, bgr)
    return oracle.filter2d_i8(bgr.reshape(rows, cols, 3), k, shift)


@pytest.mark.parametrize("depth", [1, 2, 4])
@pytest.mark.parametrize("rows,cols", [(48, 64), (37, 250), (270, 480)])
def test_ring_streams_capture_pipeline(ctx, oracle, rng, depth, rows, cols):
    """YUYV frame -> BGR -> 7x7 integer filter, the capture-side chain of the reference's read() loop, 9 frames through the ring"""
    L = _ffi.lib()
    k = rng.integers(-9, 10, size=(7, 7)).astype(np.int8)
    kp = k.ctypes.data_as(C.POINTER(C.c_int8))
    frames = rng.integers(0, 256, size=(9, rows, cols, 2), dtype=np.uint8)
    op = lambda c, din, dout: L.rcv_filter2d_i8_yuyv(c, din, dout, kp, 7, 6)
    got = []
    with rcv.StagingRing(ctx, depth, (rows, cols, 2), (rows, cols, 3)) as ring:
        for f in frames:
            if ring.full():
                got.append(ring.retire())
            ring.submit(f, op)
            assert 1 <= ring.in_flight() <= depth
        while ring.in_flight():
            got.append(ring.retire())
    assert len(got) == len(frames)
    for f, g in zip(frames, got):
        assert np.array_equal(g, _expect_yuyv_filter(oracle, f, k, 6))


def test_ring_in_place_input_and_pinned_output(ctx, oracle, rng):
    """producer writes into the ring's pinned input (no host copy); consumer reads the pinned output view"""
    L = _ffi.lib()
    rows, cols = 40, 72
    frames = rng.integers(0, 256, size=(5, rows, cols, 4), dtype=np.uint8)
    op = lambda c, din, dout: L.rcv_cvt_color(c, _ffi.RCV_BGRA2BGR_STRIDED, din, dout)
    with rcv.StagingRing(ctx, 2, (rows, cols, 4), (rows, cols, 3)) as ring:
        outs = []
        for f in frames:
            if ring.full():
                outs.append(ring.retire(copy=False).copy())
            ring.input_view()[...] = f
            ring.submit(None, op)
        while ring.in_flight():
            outs.append(ring.retire(copy=False).copy())
    for f, g in zip(frames, outs):
        assert np.array_equal(g, f[:, :, :3])


def test_ring_capacity_empty_and_errors(ctx, rng):
    L = _ffi.lib()
    rows, cols = 16, 32
    frame = rng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
    ok = lambda c, din, dout: L.rcv_gaussian_blur(c, din, dout, 3, 0.0)
    with rcv.StagingRing(ctx, 2, (rows, cols, 3), (rows, cols, 3)) as ring:
        with pytest.raises(IndexError):
            ring.retire()
        ring.submit(frame, ok)
        ring.submit(frame, ok)
        with pytest.raises(rcv.RcvError) as e:
            ring.submit(frame, ok)
        assert e.value.code == _ffi.RCV_ERR_BUSY
        with pytest.raises(rcv.RcvError) as e:
            ring.input_view()
        assert e.value.code == _ffi.RCV_ERR_BUSY
        a = ring.retire()
        # a failing op is reported by submit, and the slot is still retired in order
        with pytest.raises(rcv.RcvError) as e:
            ring.submit(frame, lambda c, din, dout: L.rcv_gaussian_blur(c, din, dout, 4, 0.0))   # even ksize
        assert e.value.code == _ffi.RCV_ERR_ARG
        assert ring.in_flight() == 2
        b = ring.retire()
        ring.retire()
        assert ring.in_flight() == 0
        assert np.array_equal(a, b)
        # wrong frame shape
        with pytest.raises(rcv.RcvError):
            ring.submit(frame[:, :16], ok)
    with pytest.raises(rcv.RcvError):
        rcv.StagingRing(ctx, 0, (4, 4, 3), (4, 4, 3))
    with pytest.raises(rcv.RcvError):
        rcv.StagingRing(ctx, 2, (4, 0, 3), (4, 4, 3))


def test_ring_i16_output(ctx, oracle, rng):
    """depth/dtype plumbing: gray u8 in, Sobel dx as i16 out (dy goes to a scratch device buffer owned by the test)"""
    L = _ffi.lib()
    rows, cols = 33, 64
    frames = rng.integers(0, 256, size=(4, rows, cols, 1), dtype=np.uint8)
    scratch = rcv.device.DeviceBatch(ctx, 1, rows, cols, 1, _ffi.RCV_16S)
    dy = scratch.as_rcv().frame0

    def op(c, din, dout):
        return L.rcv_sobel(c, din, dout, C.byref(dy))

    outs = []
    with rcv.StagingRing(ctx, 3, (rows, cols, 1), (rows, cols, 1), np.uint8, np.int16) as ring:
        for f in frames:
            if ring.full():
                outs.append(ring.retire())
            ring.submit(f, op)
        while ring.in_flight():
            outs.append(ring.retire())
    for f, g in zip(frames, outs):
        dx, _ = oracle.sobel(f[:, :, 0])
        assert np.array_equal(g[:, :, 0], dx)
    scratch.free()
