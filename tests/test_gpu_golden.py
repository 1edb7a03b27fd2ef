"""The committed golden fixtures through every HIP entry point.

tests/golden/ops_small.npz holds small seeded inputs and, for every op of the path, the expected output bytes (frozen by
tests/golden/make_golden.py; the CPU test tests/test_oracle.py::test_oracle_matches_golden keeps the oracle pinned to
the same file).  Here the SAME inputs go through the C ABI on the GPU -- host Mats: upload, HIP kernel, download -- and must
reproduce the stored outputs bit for bit, without the oracle in the loop.  kat_reference.json: the reference's own test
vectors (rustcv-camera/src/decode.rs:234-273) and the hand-derived BT.601 triples."""
import json
import os

import numpy as np
import pytest

import rustcv_amd as rcv
from rustcv_amd import _ffi, imgproc, videoio
from rustcv_amd.core import Mat

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "ops_small.npz"))


def _run(fn, src, rows, cols, ch, depth=_ffi.RCV_8U):
    dst = Mat(rows, cols, ch, depth)
    fn(Mat.from_array(src), dst)
    return dst.to_array()


def test_golden_pointwise(ctx, G):
    out = np.zeros(24 * 10 * 3, np.uint8)
    videoio.yuyv_to_bgr(G["yuyv"], out, 24, 10, ctx)
    assert np.array_equal(out, G["yuyv_bgr"])
    out = np.zeros(21 * 3, np.uint8)
    videoio.bgra_to_bgr(G["bgra"], out, 21, 1, ctx)
    assert np.array_equal(out, G["bgra_bgr"])
    m = Mat.from_array(G["bgr"])
    imgproc.rectangle(m, imgproc.Rect(5, 4, 30, 20), imgproc.Scalar(0, 255, 0), 2, ctx)
    assert np.array_equal(m.data, G["rect_5_4_30_20_t2"])
    got = _run(lambda s, d: imgproc.cvt_color(s, d, _ffi.RCV_BGR2GRAY, ctx), G["bgr"], 37, 48, 1)
    assert np.array_equal(got, G["bgr2gray"])


def test_golden_filters(ctx, G, knob):
    bgr = G["bgr"]
    for ks in (3, 5, 7):
        assert np.array_equal(_run(lambda s, d: imgproc.gaussian_blur(s, d, ks, 0.0, ctx), bgr, 37, 48, 3), G[f"gauss{ks}"]), ks
    assert np.array_equal(_run(lambda s, d: imgproc.gaussian_blur(s, d, 5, 1.2, ctx), bgr, 37, 48, 3), G["gauss5_s1p2"])
    assert np.array_equal(_run(lambda s, d: imgproc.filter2d(s, d, G["k7"], shift=6, ctx=ctx), bgr, 37, 48, 3), G["filter7_s6"])
    assert np.array_equal(_run(lambda s, d: imgproc.filter2d(s, d, G["kf3"], delta=0.25, ctx=ctx), bgr, 37, 48, 3), G["filter3_f32"])
    # the same integer filters on the row-streaming MFMA kernel (by default it only takes launches that fill the GPU)
    knob("RCV_F7_ROWS")
    assert np.array_equal(_run(lambda s, d: imgproc.filter2d(s, d, G["k7"], shift=6, ctx=ctx), bgr, 37, 48, 3), G["filter7_s6"])
    for ks in (3, 5):
        assert np.array_equal(_run(lambda s, d: imgproc.gaussian_blur(s, d, ks, 0.0, ctx), bgr, 37, 48, 3), G[f"gauss{ks}"]), ks


def test_golden_gradients_and_corners(ctx, G):
    gray = G["gray"]
    dx, dy = Mat(29, 41, 1, _ffi.RCV_16S), Mat(29, 41, 1, _ffi.RCV_16S)
    imgproc.sobel(Mat.from_array(gray), dx, dy, ctx)
    assert np.array_equal(dx.to_array(), G["sobel_dx"]) and np.array_equal(dy.to_array(), G["sobel_dy"])
    resp = Mat(29, 41, 1, _ffi.RCV_32F)
    imgproc.corner_harris(Mat.from_array(gray), resp, 2, 0.04, ctx)
    assert np.array_equal(resp.to_array().view(np.uint32), G["harris_b2"].view(np.uint32))   # f32 response: the same BITS
    mask = Mat(29, 41, 1)
    imgproc.nms3x3(Mat.from_array(G["harris_b2"]), mask, 1e-4, ctx)
    assert np.array_equal(mask.to_array(), G["nms"])


def test_golden_geometry(ctx, G):
    bgr = G["bgr"]
    assert np.array_equal(_run(lambda s, d: imgproc.resize(s, d, ctx), bgr, 9, 12, 3), G["resize_9x12"])
    assert np.array_equal(_run(lambda s, d: imgproc.resize(s, d, ctx), bgr, 50, 70, 3), G["resize_50x70"])
    assert np.array_equal(_run(lambda s, d: imgproc.warp_affine(s, d, G["warp_M"], ctx), bgr, 37, 48, 3), G["warp"])
    # RCV_32F (round 3): the stored Harris response map through the f32 warp / resize, against the STORED bits
    resp = Mat.from_array(G["harris_b2"])
    out = Mat(29, 41, 1, _ffi.RCV_32F)
    imgproc.warp_affine(resp, out, G["warp_M"], ctx)
    assert np.array_equal(out.to_array().view(np.uint32), G["warp_f32"].view(np.uint32))
    out = Mat(17, 23, 1, _ffi.RCV_32F)
    imgproc.resize(resp, out, ctx)
    assert np.array_equal(out.to_array().view(np.uint32), G["resize_f32_17x23"].view(np.uint32))


def test_golden_synthetic_frames(ctx, G):
    from rustcv_amd import device
    for name, (rows, cols, fam, frame) in {"synth_scene": (24, 40, 1, 2), "synth_noise": (8, 8, 0, 0)}.items():
        b = device.DeviceBatch(ctx, 1, rows, cols, 3)
        device.synth(b, fam, 0x5EED0003, frame)
        assert np.array_equal(b.download()[0], G[name]), name
        b.free()


def test_reference_known_answers_on_the_gpu(ctx):
    kat = json.load(open(os.path.join(HERE, "golden", "kat_reference.json")))
    for t in kat["reference_tests"]:
        if "yuyv" in t:
            out = np.zeros(t["w"] * t["h"] * 3, np.uint8)
            videoio.yuyv_to_bgr(np.array(t["yuyv"], np.uint8), out, t["w"], t["h"], ctx)
            assert (out > 240).all() if t["check"] == "all > 240" else (out < 10).all(), t["name"]
        else:
            rgb = np.array(t["rgb"], np.uint8)
            out = np.zeros_like(rgb)
            videoio.rgb_to_bgr(rgb, out, ctx)
            assert out.tolist() == t["bgr"], t["name"]
    for (y, u, v), bgr in kat["yuv_to_bgr_hand_derived"]:
        out = np.zeros(6, np.uint8)
        videoio.yuyv_to_bgr(np.array([y, u, y, v], np.uint8), out, 2, 1, ctx)
        assert out.tolist() == bgr + bgr, (y, u, v)
