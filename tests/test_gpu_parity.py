"""GPU parity: every HIP entry point of librustcv_hip.so vs the CPU oracle, through the C ABI.

Bit-exact for all u8 / i16 outputs and -- because the f32 evaluation order is fixed on both
sides -- bit-exact for the f32 Harris response too (tolerance stated where it is looser).
"""
import ctypes as C

import os

import numpy as np
import pytest

import rustcv_amd as rcv
from rustcv_amd import _ffi, device, imgproc, videoio
from rustcv_amd.core import Mat
from rustcv_amd.imgproc import Rect, Scalar

pytestmark = pytest.mark.gpu

# soak runs: RCV_SOAK=N multiplies the case count of the seeded random tests, RCV_SOAK_SEED shifts their seeds
_SOAK = max(1, int(os.environ.get("RCV_SOAK", "10")))   # default: 200-400 cases per random test, a few seconds in total
_SOAK_SEED = int(os.environ.get("RCV_SOAK_SEED", "0"))

SHAPES = [(1, 1), (1, 7), (9, 1), (2, 2), (3, 5), (17, 33), (48, 64), (61, 127), (128, 240), (37, 515)]


def rand_img(rng, rows, cols, ch):
    a = rng.integers(0, 256, size=(rows, cols, ch), dtype=np.uint8)
    return a[:, :, 0] if ch == 1 else a


# ---- a1: YUYV -> BGR -----------------------------------------------------------------------------

def test_yuyv_reference_tests(ctx):
    # rustcv-camera/src/decode.rs:234-265 replayed on the GPU path
    d = np.zeros(6, np.uint8)
    assert videoio.yuyv_to_bgr(np.array([235, 128, 235, 128], np.uint8), d, 2, 1, ctx)
    assert (d > 240).all()
    assert videoio.yuyv_to_bgr(np.array([16, 128, 16, 128], np.uint8), d, 2, 1, ctx)
    assert (d < 10).all()


def test_yuyv_exhaustive_all_yuv_triples(ctx, oracle):
    """All 2^24 (Y,U,V) triples (SURVEY.md §4 tier 3), 16.7 M macropixels, vector + scalar paths."""
    y, u, v = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    src = np.stack([y, u, np.roll(y, 1, axis=0), v], axis=-1).reshape(-1)  # Y0,U,Y1,V with Y1 != Y0
    w, h = 4096, 8192
    assert src.size == w * h * 2
    want = np.zeros(w * h * 3, np.uint8)
    assert oracle.yuyv_to_bgr(src, want, w, h)
    got = np.zeros_like(want)
    assert videoio.yuyv_to_bgr(src, got, w, h, ctx)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("w,h", [(2, 1), (3, 1), (5, 3), (640, 480), (641, 3), (18, 1), (16, 1), (1, 1), (0, 0), (7, 9)])
@pytest.mark.parametrize("code,variant", [(_ffi.RCV_YUYV2BGR, 0), (_ffi.RCV_YUYV2BGR_TWIN, 1)])
def test_yuyv_shapes_and_guards(ctx, oracle, rng, w, h, code, variant):
    for slack_src, slack_dst in [(0, 0), (5, 7), (-1, 0), (0, -1)]:
        src = rng.integers(0, 256, size=max(w * h * 2 + slack_src, 0), dtype=np.uint8)
        dst0 = rng.integers(0, 256, size=max(w * h * 3 + slack_dst, 0), dtype=np.uint8)
        want = dst0.copy()
        ran = oracle.yuyv_to_bgr(src, want, w, h, variant)
        got = Mat(h, w, 3, data=dst0.copy())
        got.data = dst0.copy()
        s = Mat(1, src.size, 1, data=src)
        sm, dm = s._as_rcv(), got._as_rcv()
        rc = _ffi.lib().rcv_cvt_color(ctx.handle, code, C.byref(sm), C.byref(dm))
        if variant == 0 and src.size >= w * h * 2 and dst0.size < (w * h // 2) * 6:
            assert rc == _ffi.RCV_ERR_SIZE  # reference facade would panic; we refuse
            continue
        assert rc == (_ffi.RCV_OK if ran else _ffi.RCV_NOOP), (rc, ran)
        assert np.array_equal(got.data, want)


# ---- a2 / a4 ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("npx", [0, 1, 15, 16, 17, 640 * 480, 100003])
def test_bgra_to_bgr(ctx, oracle, rng, npx):
    src = rng.integers(0, 256, size=npx * 4, dtype=np.uint8)
    want = np.full(npx * 3 + 5, 7, np.uint8)
    got = want.copy()
    w, h = (npx, 1) if npx else (0, 0)
    assert oracle.bgra_to_bgr(src, want, w, h, 0)
    assert videoio.bgra_to_bgr(src, got, w, h, ctx)
    assert np.array_equal(got, want)
    # facade guard: short src -> silent no-op
    if npx:
        got2 = np.full(npx * 3, 9, np.uint8)
        assert not videoio.bgra_to_bgr(src[:-1], got2, w, h, ctx)
        assert (got2 == 9).all()


def test_rgb_to_bgr_reference_vector(ctx):
    # rustcv-camera/src/decode.rs:267-273 -- the one exact vector the reference holds
    d = np.zeros(6, np.uint8)
    videoio.rgb_to_bgr(np.array([255, 0, 0, 0, 255, 0], np.uint8), d, ctx)
    assert d.tolist() == [0, 0, 255, 0, 255, 0]


@pytest.mark.parametrize("nsrc,ndst", [(48, 48), (49, 50), (3 * 1000 + 2, 3 * 999), (3 * 100003, 3 * 100003), (0, 6), (2, 9)])
def test_rgb_to_bgr(ctx, oracle, rng, nsrc, ndst):
    src = rng.integers(0, 256, size=nsrc, dtype=np.uint8)
    want = np.full(ndst, 3, np.uint8)
    got = want.copy()
    oracle.rgb_to_bgr(src, want)
    videoio.rgb_to_bgr(src, got, ctx)
    assert np.array_equal(got, want)


# ---- "next" rows f2 / f4 -----------------------------------------------------------------------------

@pytest.mark.parametrize("rows,cols,extra", [(1, 1, 0), (3, 5, 2), (48, 64, 0), (37, 515, -7), (480, 640, 0)])
def test_bgr_to_u32_display_buffer(ctx, oracle, rng, rows, cols, extra):
    """highgui mat_to_u32_buffer (highgui/mod.rs:125-141): flat chunks_exact(3), zero-filled tail"""
    data = rng.integers(0, 256, size=max(rows * cols * 3 + extra, 0), dtype=np.uint8)
    src = Mat(rows, cols, 3, data=data)
    src.data = data
    dst = Mat(rows, cols, 4)
    dst.data[:] = 0x77
    imgproc.cvt_color(src, dst, _ffi.RCV_BGR2BGRX, ctx)
    assert np.array_equal(dst.data.view(np.uint32), oracle.bgr_to_u32(data, rows * cols))


@pytest.mark.parametrize("rows,cols,pad", [(1, 1, 0), (3, 5, 1), (48, 64, 16), (37, 516, 8)])
def test_bgr_rows_to_rgb(ctx, oracle, rng, rows, cols, pad):
    """imwrite's BGR -> RGB loop over row_bytes (imgcodecs/mod.rs:51-63): honours step, packs the output"""
    step = cols * 3 + pad
    data = rng.integers(0, 256, size=rows * step, dtype=np.uint8)
    src = Mat(rows, cols, 3, step=step, data=data)
    dst = Mat(rows, cols, 3)
    imgproc.cvt_color(src, dst, _ffi.RCV_BGR2RGB, ctx)
    assert np.array_equal(dst.data, oracle.bgr_to_rgb_rows(data, step, rows, cols))


@pytest.mark.parametrize("rows,cols,pad", [(1, 2, 0), (3, 5, 2), (48, 64, 16), (37, 516, 8), (33, 130, 4)])
@pytest.mark.parametrize("uyvy", [False, True])
def test_yuv422_strided(ctx, oracle, rng, rows, cols, pad, uyvy):
    step = cols * 2 + pad
    data = rng.integers(0, 256, size=rows * step, dtype=np.uint8)
    src = Mat(rows, cols, 2, step=step, data=data)
    dst = Mat(rows, cols, 3)
    dst.data[:] = 0x55
    want = dst.data.copy()
    oracle.yuv422_to_bgr_strided(data, step, rows, cols, uyvy, want)
    imgproc.cvt_color(src, dst, _ffi.RCV_UYVY2BGR_STRIDED if uyvy else _ffi.RCV_YUYV2BGR_STRIDED, ctx)
    assert np.array_equal(dst.data, want)


@pytest.mark.parametrize("rows,cols,spad,dpad", [(1, 1, 0, 0), (3, 5, 4, 1), (48, 64, 16, 0), (37, 516, 48, 12), (33, 130, 8, 4), (20, 64, 3, 0)])
def test_bgra_strided(ctx, oracle, rng, rows, cols, spad, dpad):
    """f2: BGRA with a real row stride (AVF reports bytes_per_row) -> BGR with its own step == bgra_to_bgr applied row by row"""
    sstep, dstep = cols * 4 + spad, cols * 3 + dpad
    data = rng.integers(0, 256, size=rows * sstep, dtype=np.uint8)
    src = Mat(rows, cols, 4, step=sstep, data=data)
    dst = Mat(rows, cols, 3, step=dstep, data=np.full(rows * dstep, 0x55, np.uint8))
    want = dst.data.copy()
    for r in range(rows):
        row = np.zeros(cols * 3, np.uint8)
        assert oracle.bgra_to_bgr(data[r * sstep: r * sstep + cols * 4], row, cols, 1)
        want[r * dstep: r * dstep + cols * 3] = row
    assert imgproc.cvt_color(src, dst, _ffi.RCV_BGRA2BGR_STRIDED, ctx) == _ffi.RCV_OK
    assert np.array_equal(dst.data, want)      # padding bytes untouched
    with pytest.raises(Exception):
        imgproc.cvt_color(Mat(rows, cols, 3), dst, _ffi.RCV_BGRA2BGR_STRIDED, ctx)


@pytest.mark.parametrize("rows,cols,pad", [(1, 2, 0), (2, 2, 0), (3, 5, 3), (48, 64, 16), (37, 516, 8), (33, 130, 2)])
def test_nv12(ctx, oracle, rng, rows, cols, pad):
    step = cols + (cols & 1) + pad
    data = rng.integers(0, 256, size=step * (rows + (rows + 1) // 2), dtype=np.uint8)
    src = Mat(rows, cols, 1, step=step, data=data)
    src.data = data
    dst = Mat(rows, cols, 3)
    want = dst.data.copy()
    assert oracle.nv12_to_bgr(data, step, rows, cols, want)
    assert imgproc.cvt_color(src, dst, _ffi.RCV_NV12_2BGR, ctx) == _ffi.RCV_OK
    assert np.array_equal(dst.data, want)
    short = Mat(rows, cols, 1, step=step, data=data[:-1])
    short.data = data[:-1]
    dst.data[:] = 9
    assert imgproc.cvt_color(short, dst, _ffi.RCV_NV12_2BGR, ctx) == _ffi.RCV_NOOP and (dst.data == 9).all()


def test_decode_into_dispatch(ctx, oracle, rng):
    w, h = 64, 48
    m = Mat.empty()
    yuyv = rng.integers(0, 256, size=w * h * 2, dtype=np.uint8)
    videoio.decode_into(m, yuyv, videoio.YUYV, w, h, ctx)
    want = np.zeros(w * h * 3, np.uint8)
    oracle.yuyv_to_bgr(yuyv, want, w, h)
    assert (m.rows, m.cols, m.channels, m.step) == (h, w, 3, w * 3) and np.array_equal(m.data, want)
    bgra = rng.integers(0, 256, size=w * h * 4, dtype=np.uint8)
    videoio.decode_into(m, bgra, videoio.BGRA, w, h, ctx)
    oracle.bgra_to_bgr(bgra, want, w, h)
    assert np.array_equal(m.data, want)
    raw = rng.integers(0, 256, size=w * h * 3, dtype=np.uint8)
    videoio.decode_into(m, raw, videoio.BGR3, w, h, ctx)  # unknown -> copy when the length matches
    assert np.array_equal(m.data, raw)


# ---- a3: rectangle ---------------------------------------------------------------------------------

RECTS = [
    (Rect(200, 150, 240, 240), 2), (Rect(0, 0, 640, 480), 1), (Rect(-10, -10, 50, 50), 3), (Rect(600, 440, 100, 100), 4),
    (Rect(10, 10, 1, 1), 1), (Rect(10, 10, 5, 3), 9), (Rect(630, 5, 10, 400), 20), (Rect(700, 5, 10, 10), 2),
    (Rect(5, 5, 0, 10), 2), (Rect(5, 5, 10, 10), 0), (Rect(5, 5, 10, 10), -1), (Rect(0, 470, 640, 10), 15), (Rect(3, 3, 600, 6), 700),
]


@pytest.mark.parametrize("rect,thick", RECTS)
@pytest.mark.parametrize("pad", [0, 13])
def test_rectangle(ctx, oracle, rng, rect, thick, pad):
    rows, cols = 480, 640
    step = cols * 3 + pad
    base = rng.integers(0, 256, size=rows * step, dtype=np.uint8)
    want = base.copy()
    oracle.rectangle(want, rows, cols, step, rect.x, rect.y, rect.width, rect.height, 0, 255, 7, thick)
    m = Mat(rows, cols, 3, step=step, data=base.copy())
    imgproc.rectangle(m, rect, Scalar(0, 255, 7), thick, ctx)
    assert np.array_equal(m.data, want)


def test_calls_without_ctx_use_the_default_context(oracle, rng):
    # imgproc.* / videoio.* without ctx= go through core.default_context()
    rows, cols = 48, 64
    base = rng.integers(0, 256, size=rows * cols * 3, dtype=np.uint8)
    want = base.copy()
    oracle.rectangle(want, rows, cols, cols * 3, 5, 6, 30, 20, 9, 8, 7, 1)
    m = Mat(rows, cols, 3, data=base.copy())
    imgproc.rectangle(m, Rect(5, 6, 30, 20), Scalar(9, 8, 7), 1)
    assert np.array_equal(m.data, want)
    img = rand_img(rng, 40, 52, 3)
    k = rng.integers(-8, 9, size=(3, 3)).astype(np.int8)
    got = Mat.new(40, 52, 3)
    imgproc.filter2d(Mat.from_array(img), got, k, shift=4)
    assert np.array_equal(got.to_array(), oracle.filter2d_i8(img, k, 4))


def test_rectangle_short_vec_guard(ctx, oracle, rng):
    # data.len() shorter than rows*step: only the `idx+2 < len` guard protects memory (drawing.rs:82)
    rows, cols, step = 20, 30, 90
    base = rng.integers(0, 256, size=rows * step - 100, dtype=np.uint8)
    want = base.copy()
    oracle.rectangle(want, rows, cols, step, 2, 2, 26, 17, 1, 2, 3, 2)
    m = Mat(rows, cols, 3, step=step, data=base.copy())
    imgproc.rectangle(m, Rect(2, 2, 26, 17), Scalar(1, 2, 3), 2, ctx)
    assert np.array_equal(m.data, want)


# ---- f4: put_text's blend (drawing.rs:137-160) -------------------------------------------------------

def _rand_glyphs(rng, rows, cols, n, special=True):
    """boxes around and across the Mat (overlapping, clipped, outside, empty), coverage in [0, 1] with special values"""
    out = []
    for _ in range(n):
        h, w = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        gx, gy = int(rng.integers(-45, cols + 10)), int(rng.integers(-45, rows + 10))
        cov = rng.random((h, w), dtype=np.float32)
        if special and cov.size:
            flat = cov.reshape(-1)
            k = rng.integers(0, flat.size, size=max(1, flat.size // 6))
            flat[k] = rng.choice(np.array([0.0, 1.0, 0.5, -0.25, 1.25, 1e-40, np.nan, np.inf, -np.inf, 0.99999994], np.float32), size=k.size)
        out.append((gx, gy, cov))
    return out


@pytest.mark.parametrize("rows,cols,pad", [(1, 1, 0), (48, 64, 0), (61, 127, 13), (128, 240, 5), (480, 640, 0)])
def test_blend_glyphs_host_mat(ctx, oracle, rows, cols, pad):
    for case in range(_SOAK):
        rng = np.random.default_rng(7000 + 97 * case + rows + _SOAK_SEED)
        step = cols * 3 + pad
        base = rng.integers(0, 256, size=rows * step, dtype=np.uint8)
        glyphs = _rand_glyphs(rng, rows, cols, int(rng.integers(0, 30)))
        b, g, r = (int(v) for v in rng.integers(0, 256, 3))
        want = base.copy()
        oracle.blend_glyphs(want, rows, cols, step, glyphs, b, g, r)
        m = Mat(rows, cols, 3, step=step, data=base.copy())
        imgproc.blend_glyphs(m, glyphs, Scalar(b, g, r), ctx)
        assert np.array_equal(m.data, want), (rows, cols, pad, case)


def test_blend_glyphs_batch_many_glyphs_and_canaries(ctx, oracle):
    """more glyphs than one launch takes (ordered chunks), the same text on every frame of a padded resident batch"""
    rng = np.random.default_rng(424242 + _SOAK_SEED)
    n, rows, cols = 3, 200, 320
    bt = _canary_batch(ctx, n, rows, cols, 3)
    raw = np.full(bt.nbytes, 0xCD, np.uint8)
    for i in range(n):
        v = raw[i * bt.frame_stride: i * bt.frame_stride + rows * bt.step].reshape(rows, bt.step)
        v[:, : cols * 3] = rng.integers(0, 256, size=(rows, cols * 3), dtype=np.uint8)
    bt.upload_bytes(raw)
    glyphs = [(x, y, np.ascontiguousarray(c[:12, :12])) for x, y, c in _rand_glyphs(rng, rows, cols, 2500, special=False)]
    device.blend_glyphs(bt, glyphs, Scalar(9, 200, 255))
    want = raw.copy()
    for i in range(n):
        fr = want[i * bt.frame_stride: i * bt.frame_stride + rows * bt.step]      # a view: blended in place, padding included
        oracle.blend_glyphs(fr, rows, cols, bt.step, glyphs, 9, 200, 255)
    assert np.array_equal(bt.download_bytes(), want)
    bt.free()


def test_blend_glyphs_text_line_on_1080p(ctx, oracle):
    """the shape of a real put_text call: one line of ~40 overlapping soft-edged boxes on a 1080p frame"""
    rng = np.random.default_rng(5150 + _SOAK_SEED)
    rows, cols = 1080, 1920
    base = rng.integers(0, 256, size=rows * cols * 3, dtype=np.uint8)
    yy, xx = np.mgrid[0:34, 0:24].astype(np.float32)
    blob = np.clip(1.4 - np.hypot((xx - 11.5) / 9, (yy - 16.5) / 14), 0, 1).astype(np.float32)   # anti-aliased ellipse
    glyphs = [(100 + 20 * i, 500 + (i % 3), blob) for i in range(40)]                            # advance < box width: boxes overlap
    want = base.copy()
    oracle.blend_glyphs(want, rows, cols, cols * 3, glyphs, 0, 255, 0)
    m = Mat(rows, cols, 3, data=base.copy())
    imgproc.blend_glyphs(m, glyphs, Scalar(0, 255, 0), ctx)
    assert np.array_equal(m.data, want)
    assert (m.data != base).sum() > 10000


def test_blend_glyphs_errors(ctx):
    L = _ffi.lib()
    m = Mat(8, 8, 3)
    a = m._as_rcv()
    cov = (C.c_float * 16)(*([0.5] * 16))

    def call(mat, gl, n, ncov=16):
        tbl = (_ffi.rcv_glyph * max(1, len(gl)))(*gl)
        return L.rcv_blend_glyphs(ctx.handle, C.byref(mat), tbl, n, cov, ncov, 1, 2, 3)

    G = _ffi.rcv_glyph
    assert call(a, [G(0, 0, 4, 4, 0)], 1) == _ffi.RCV_OK
    assert call(a, [G(0, 0, 4, 4, 1)], 1) == _ffi.RCV_ERR_SIZE            # coverage range leaves the array
    assert call(a, [G(0, 0, 5, 4, 0)], 1) == _ffi.RCV_ERR_SIZE
    assert call(a, [G(0, 0, 1, 1, 17)], 1) == _ffi.RCV_ERR_SIZE
    assert call(a, [G(0, 0, -1, 4, 0)], 1) == _ffi.RCV_ERR_ARG
    assert call(a, [G(0, 0, 4, 4, 0)], -1) == _ffi.RCV_ERR_ARG
    assert call(a, [G(2**31 - 1, -2**31, 4, 4, 0)], 1) == _ffi.RCV_OK     # far outside: nothing to do
    assert call(a, [], 0) == _ffi.RCV_OK
    assert L.rcv_blend_glyphs(ctx.handle, C.byref(a), None, 1, cov, 16, 1, 2, 3) == _ffi.RCV_ERR_ARG
    g1 = Mat(8, 8, 1)._as_rcv()
    assert call(g1, [G(0, 0, 4, 4, 0)], 1) == _ffi.RCV_ERR_UNSUPPORTED    # the reference hard-codes 3 channels
    short = Mat(8, 8, 3)
    short.data = short.data[:-1]
    assert call(short._as_rcv(), [G(0, 0, 4, 4, 0)], 1) == _ffi.RCV_ERR_SIZE   # where the reference would panic
    before = m.data.copy()
    assert call(a, [G(0, 0, 2, 2, 0), G(0, 0, 4, 4, 1)], 2) == _ffi.RCV_ERR_SIZE and np.array_equal(m.data, before)   # all or nothing


# ---- build-defined ops -------------------------------------------------------------------------------

@pytest.mark.parametrize("rows,cols", SHAPES)
def test_bgr2gray(ctx, oracle, rng, rows, cols):
    img = rand_img(rng, rows, cols, 3)
    for pad in (0, 5):
        src = Mat.from_array(img, step=cols * 3 + pad)
        dst = Mat(rows, cols, 1, step=cols + (3 if pad else 0))
        imgproc.cvt_color(src, dst, _ffi.RCV_BGR2GRAY, ctx)
        assert np.array_equal(dst.to_array(), oracle.bgr2gray(img))


@pytest.mark.parametrize("rows,cols", SHAPES)
@pytest.mark.parametrize("ch", [1, 3])
@pytest.mark.parametrize("ksize", [3, 5, 7])
def test_gaussian_int(ctx, oracle, rng, rows, cols, ch, ksize):
    img = rand_img(rng, rows, cols, ch)
    src = Mat.from_array(img, step=cols * ch + 16)
    dst = Mat(rows, cols, ch)
    imgproc.gaussian_blur(src, dst, ksize, 0.0, ctx)
    assert np.array_equal(dst.to_array(), oracle.gaussian_blur(img, ksize, 0.0))


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 5), (17, 33), (61, 127)])
@pytest.mark.parametrize("ksize,sigma", [(3, 0.8), (5, 1.1), (9, 2.0), (15, 3.3)])
def test_gaussian_sigma(ctx, oracle, rng, rows, cols, ksize, sigma):
    img = rand_img(rng, rows, cols, 3)
    src, dst = Mat.from_array(img), Mat(rows, cols, 3)
    imgproc.gaussian_blur(src, dst, ksize, sigma, ctx)
    assert np.array_equal(dst.to_array(), oracle.gaussian_blur(img, ksize, sigma))  # bit-exact: same fmaf chains


@pytest.mark.parametrize("rows,cols", SHAPES)
@pytest.mark.parametrize("ch", [1, 3])
@pytest.mark.parametrize("ksize,shift", [(1, 0), (3, 4), (5, 0), (7, 6), (7, 9)])
def test_filter2d_i8(ctx, oracle, rng, rows, cols, ch, ksize, shift):
    img = rand_img(rng, rows, cols, ch)
    k = rng.integers(-128, 128, size=(ksize, ksize), dtype=np.int8) if shift == 9 else rng.integers(-8, 9, size=(ksize, ksize)).astype(np.int8)
    src = Mat.from_array(img, step=cols * ch + 7)
    dst = Mat(rows, cols, ch)
    imgproc.filter2d(src, dst, k, shift=shift, ctx=ctx)
    assert np.array_equal(dst.to_array(), oracle.filter2d_i8(img, k, shift))


# shapes the MFMA strip kernel takes (ch=3, cols % 16 == 0, rows >= 4): strip / segment / tile-edge cases
MFMA_SHAPES = [(4, 16), (5, 32), (16, 16), (17, 48), (33, 240), (40, 256), (19, 496), (300, 272), (131, 720), (260, 16)]


@pytest.mark.parametrize("rows,cols", MFMA_SHAPES)
@pytest.mark.parametrize("ksize,shift", [(7, 6), (5, 3), (3, 0), (7, 0)])
@pytest.mark.parametrize("pad", [0, 32])
@pytest.mark.parametrize("pipelined", [False, True, "rows"])
def test_filter2d_i8_mfma_path(ctx, oracle, rng, knob, rows, cols, ksize, shift, pad, pipelined):
    # small launches take the strip kernel's latency variant; RCV_F7_NO_LAT sends the same shapes through the pipelined one,
    # RCV_F7_ROWS=1 through the row-streaming kernel (which by default only takes launches that fill the GPU)
    if pipelined == "rows":
        knob("RCV_F7_ROWS")
    elif pipelined:
        knob("RCV_F7_NO_LAT")
    img = rand_img(rng, rows, cols, 3)
    k = rng.integers(-128, 128, size=(ksize, ksize), dtype=np.int8) if shift == 0 else rng.integers(-9, 10, size=(ksize, ksize)).astype(np.int8)
    src = Mat.from_array(img, step=cols * 3 + pad)
    dst = Mat(rows, cols, 3, step=cols * 3 + (pad // 2))
    dst.data[:] = 0xAB  # padding bytes must stay untouched
    imgproc.filter2d(src, dst, k, shift=shift, ctx=ctx)
    assert np.array_equal(dst.to_array(), oracle.filter2d_i8(img, k, shift))
    if pad:
        padbytes = dst.data.reshape(rows, dst.step)[:, cols * 3:]
        assert (padbytes == 0xAB).all()


@pytest.mark.parametrize("variant", ["bgr", "gray", "yuyv", "sobel", "dual"])
@pytest.mark.parametrize("chain", [None, 0])
def test_row_kernel_tapered_bands_every_frame(ctx, oracle, knob, variant, chain):
    """round 3: launches that fill the GPU several times over cut the tail of every XCD's band list into half- and quarter-height
    bands (tapered bands).  32 frames of 1080p -- the smallest BASELINE-shaped batch that takes this path -- EVERY frame against the
    oracle, for each source flavour of the row kernel (BGR, one-channel, packed YUYV, the fused filter -> gray -> Sobel launch, two
    weight tables); frame boundaries inside XCD ranges included (n % 8 == 0: four frames per XCD).  Round 4: the plain BGR launch of this
    size takes the chained-band kernel by default; RCV_FR_CHAIN=0 keeps the tapered one-band-per-wave kernel covered"""
    if chain is not None:
        knob("RCV_FR_CHAIN", chain)
    knob("RCV_GAUSS_ROWS", 0)
    n, rows, cols = 32, 1080, 1920
    r = np.random.default_rng(991 + _SOAK_SEED)
    k = r.integers(-8, 9, size=(7, 7)).astype(np.int8)
    k[3, 3] = 33
    L = _ffi.lib()
    L.rcv__debug_kernels_reset()
    ch, fam = {"bgr": (3, 1), "dual": (3, 1), "sobel": (3, 1), "gray": (1, 1), "yuyv": (2, 2)}[variant]
    src = device.DeviceBatch(ctx, n, rows, cols, ch)
    device.synth(src, fam, 77, 0)
    frames = src.download()     # (the oracle works on what the device holds)
    if variant == "sobel":
        dx, dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
        device.filter2d_sobel(src, dx, dy, k, 6)
        gx, gy = dx.download(), dy.download()
        for i in range(n):
            wx, wy = oracle.sobel(oracle.bgr2gray(oracle.filter2d_i8(frames[i], k, 6)))
            assert np.array_equal(gx[i], wx.reshape(rows, cols)) and np.array_equal(gy[i], wy.reshape(rows, cols)), i
        outs = [dx, dy]
    else:
        dst = device.DeviceBatch(ctx, n, rows, cols, 1 if variant == "gray" else 3)
        if variant == "dual":
            device.gaussian_blur(src, dst, 7, 0.0)
        elif variant == "yuyv":
            device.filter2d_yuyv(src, dst, k, shift=6)
        else:
            device.filter2d(src, dst, k, shift=6)
        got = dst.download()
        for i in range(n):
            if variant == "dual":
                want = oracle.gaussian_blur(frames[i], 7, 0.0)
            elif variant == "yuyv":
                bgr = np.zeros(rows * cols * 3, np.uint8)
                oracle.yuyv_to_bgr(frames[i].reshape(-1), bgr, cols, rows)
                want = oracle.filter2d_i8(bgr.reshape(rows, cols, 3), k, 6)
            else:
                want = oracle.filter2d_i8(frames[i], k, 6)
            assert np.array_equal(got[i], want.reshape(got[i].shape)), (variant, i, np.argwhere(got[i] != want.reshape(got[i].shape))[:3])
        outs = [dst]
    launched = L.rcv__debug_kernels().decode()
    assert ("k_filter_rows_chain<" if variant == "bgr" and chain is None else "k_filter_rows_mfma<") in launched, launched
    for b in outs + [src]:
        b.free()


CHAIN_SHAPES = [(8, 64, 256), (8, 70, 272), (8, 97, 1084), (16, 131, 1040), (8, 200, 3840), (24, 66, 16), (8, 65, 772),
                # round 5: any frame count -- the batch's bands are dealt to the XCDs in eight runs that may cut through a frame
                (1, 300, 772), (3, 131, 1040), (7, 64, 256), (9, 70, 272), (13, 97, 1084), (63, 64, 528)]


@pytest.mark.parametrize("n,rows,cols", CHAIN_SHAPES)
@pytest.mark.parametrize("band_rows", [0, 8, 13])
def test_filter_rows_chained_bands(ctx, oracle, knob, n, rows, cols, band_rows):
    """round 4: k_filter_rows_chain -- persistent waves that walk SHORT bands of their strip back to back through one register ring (the
    next band's rows, halo included, are in flight while the last rows of the current one are computed; the steps whose window straddles
    two bands store nothing).  Frame counts that are and are not multiples of 8 (round 5: an XCD's run of bands may start and end inside
    a frame; one frame alone); band heights 8 / 13 / 32 rows incl. odd heights and a last band that is
    shorter; one strip, a partial last strip, widths with every residue of 4 mod 16; padded steps / frame strides with canaries; every
    frame of the batch against the oracle for ksize 3 / 5 / 7 and the integer Gaussian 5x5 (the same launch path)."""
    knob("RCV_FR_CHAIN", 1)
    knob("RCV_F7_ROWS", 1)
    knob("RCV_GAUSS_ROWS", 0)     # (small Gaussian launches would take the register-window kernel)
    if band_rows:
        knob("RCV_FR_CHAIN_ROWS", band_rows)
    r = np.random.default_rng(9000 + 31 * rows + cols + band_rows + _SOAK_SEED)
    frames = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    frames[n // 2, : rows // 2] = 255
    frames[n - 1, :, : min(cols, 40)] = 0
    src = device.DeviceBatch(ctx, n, rows, cols, 3, step=cols * 3 + 16, frame_stride=rows * (cols * 3 + 16) + 64)
    src.upload(frames)
    L = _ffi.lib()
    for ks, sh in ((7, 6), (5, 3), (3, 0)):
        k = r.integers(-9, 10, size=(ks, ks)).astype(np.int8) if sh else r.integers(-128, 128, size=(ks, ks), dtype=np.int8)
        dst = _canary_batch(ctx, n, rows, cols, 3, pad=32)
        L.rcv__debug_kernels_reset()
        device.filter2d(src, dst, k, shift=sh)
        assert f"k_filter_rows_chain<{ks}" in L.rcv__debug_kernels().decode(), L.rcv__debug_kernels().decode()
        got = dst.download()
        for i in range(n):
            want = oracle.filter2d_i8(frames[i], k, sh)
            assert np.array_equal(got[i], want), (ks, i, np.argwhere(got[i] != want)[:4])
        _assert_canaries(dst)
        dst.free()
    # the integer Gaussian 5x5 (one table: chained); 7x7 (two weight tables) stays on the one-band-per-wave kernel -- measured slower chained
    for gk, kern in ((5, "k_filter_rows_chain<5"), (7, "k_filter_rows_mfma<")):
        dst = _canary_batch(ctx, n, rows, cols, 3, pad=32)
        L.rcv__debug_kernels_reset()
        device.gaussian_blur(src, dst, gk, 0.0)
        assert kern in L.rcv__debug_kernels().decode(), L.rcv__debug_kernels().decode()
        got = dst.download()
        for i in range(n):
            assert np.array_equal(got[i], oracle.gaussian_blur(frames[i], gk, 0.0)), ("gauss", gk, i)
        _assert_canaries(dst)
        dst.free()
    src.free()


@pytest.mark.parametrize("drop", [0, 5])
def test_filter_rows_chain_reports_an_xcd_without_waves(oracle, knob, drop):
    """round 6: the chained kernel's lists are per XCD and only drawn by waves that RUN on that XCD; an XCD that
    receives no waves (a CU mask, a partition that still reports 256 CUs) used to leave an eighth of the batch unfiltered without a word.
    Now every launch checks that its predecessor drew all of its items (first wave of the launch) and every host-side wait enqueues the
    check of the last launch; a short queue comes back as RCV_ERR_DEVICE, once, and the context stays on the one-band-per-wave kernel.
    Fault injection RCV_FR_CHAIN_DROP_XCD: the waves on that XCD leave at once.  Never stale output without an error."""
    import rustcv_amd as rcv
    from rustcv_amd._ffi import RcvError
    knob("RCV_FR_CHAIN", 1)
    knob("RCV_F7_ROWS", 1)
    c = rcv.Context(0)
    r = np.random.default_rng(77 + drop + _SOAK_SEED)
    n, rows, cols = 16, 80, 1040
    frames = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    k = r.integers(-9, 10, size=(7, 7)).astype(np.int8)
    src = device.DeviceBatch(c, n, rows, cols, 3)
    src.upload(frames)
    dst = device.DeviceBatch(c, n, rows, cols, 3)
    L = _ffi.lib()
    want = [oracle.filter2d_i8(frames[i], k, 5) for i in range(n)]
    # (a) healthy launches: no false alarm, whichever path does the check (successor launch or the wait)
    for reps in (1, 3):
        for _ in range(reps):
            device.filter2d(src, dst, k, shift=5)
        c.sync()
    got = dst.download()
    assert all(np.array_equal(got[i], want[i]) for i in range(n))
    # (b) one faulty launch, then a wait: the wait reports it
    knob("RCV_FR_CHAIN_DROP_XCD", drop)
    dst.memset(0)
    L.rcv__debug_kernels_reset()
    device.filter2d(src, dst, k, shift=5)
    assert "k_filter_rows_chain<7" in L.rcv__debug_kernels().decode()
    with pytest.raises(RcvError) as e:
        c.sync()
    assert e.value.code == -4   # RCV_ERR_DEVICE
    got = dst.download()        # (reported once: the download itself succeeds)
    assert any(not np.array_equal(got[i], want[i]) for i in range(n)), "the injected fault left no trace: the hook did not work"
    # (c) the context has left the chained kernel for good: correct bytes from the one-band-per-wave kernel, hook still set
    dst.memset(0)
    L.rcv__debug_kernels_reset()
    device.filter2d(src, dst, k, shift=5)
    c.sync()
    assert "k_filter_rows_mfma<" in L.rcv__debug_kernels().decode() and "chain" not in L.rcv__debug_kernels().decode()
    got = dst.download()
    assert all(np.array_equal(got[i], want[i]) for i in range(n))
    c.close()
    # (d) a fresh context, faulty launches queued back to back without a wait: a LATER call reports the fault (successor's check)
    c2 = rcv.Context(0)
    src2 = device.DeviceBatch(c2, n, rows, cols, 3)
    src2.upload(frames)
    dst2 = device.DeviceBatch(c2, n, rows, cols, 3)
    with pytest.raises(RcvError):
        for _ in range(200):
            device.filter2d(src2, dst2, k, shift=5)
        c2.sync()
    c2.sync()
    src2.free(); dst2.free()
    c2.close()


@pytest.mark.parametrize("taper", [0, 3 + 256 * 2, 1 + 256 * 1, 6])
@pytest.mark.parametrize("n,rows,cols", [(16, 100, 1040), (9, 64, 256), (24, 131, 772)])
def test_filter_rows_chain_tapered_tail(ctx, oracle, n, rows, cols, taper):
    """round 6: the last bands of every XCD's run drawn as half- and quarter-height items (FRArgs::tp1 / tp2); through the measurement entry,
    where the plan is an argument (the product's own plan is covered by every other chained test); every frame against the oracle"""
    import ctypes as C
    BL = _ffi.bench_lib()
    r = np.random.default_rng(515 + rows + taper + _SOAK_SEED)
    frames = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    k = r.integers(-9, 10, size=(7, 7)).astype(np.int8)
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    src.upload(frames)
    dst = _canary_batch(ctx, n, rows, cols, 3, pad=32)
    bs, bd = src.as_rcv(), dst.as_rcv()
    for chain_rows in (0, 16):
        rc = BL.rcv__filter_rows_bench(ctx.handle, C.byref(bs), C.byref(bd), k.ctypes.data_as(C.POINTER(C.c_int8)), 7, 5,
                                       _ffi.rows_tune(chain=1, taper=taper, chain_rows=chain_rows), None)
        assert rc == 0
        got = dst.download()
        for i in range(n):
            want = oracle.filter2d_i8(frames[i], k, 5)
            assert np.array_equal(got[i], want), (chain_rows, i, np.argwhere(got[i] != want)[:4])
        _assert_canaries(dst)
    src.free(); dst.free()


def test_filter_split_call_two_halves_two_streams(oracle, knob):
    """round 6: a chained filter2D call of 16+ BGR frames runs as two launches -- frames [0, n/2) on the context's stream, the rest on its half
    stream -- and NOTHING is joined per call (the halves of consecutive calls hide each other's tail: profiles/r06_split_halves.txt).  Every other
    entry point joins first, so what a caller can observe stays ordered on the one stream.  Checked here: (a) the split happens (two chained launches
    per call) and every frame equals the oracle's, odd frame counts too; (b) a dependent chain of calls without a wait in between (call 2 reads
    what call 1 wrote, call 3 writes what call 2 still reads) gives the oracle's double filter; (c) a call whose source is the previous
    destination SHIFTED by a few frames -- its halves depend on the OTHER stream's pending work -- still does; (d) other entry points (Sobel of
    the result, a download) right behind a split call see its whole result; (e) RCV_FR_SPLIT=0 and a second busy context of the device keep the
    call in one launch, same bytes."""
    import rustcv_amd as rcv
    knob("RCV_FR_CHAIN", 1)
    knob("RCV_F7_ROWS", 1)
    c = rcv.Context(0)
    L = _ffi.lib()
    r = np.random.default_rng(2024 + _SOAK_SEED)
    rows, cols = 64, 272
    k = r.integers(-9, 10, size=(7, 7)).astype(np.int8)
    k2 = r.integers(-9, 10, size=(5, 5)).astype(np.int8)
    f1 = lambda fr: oracle.filter2d_i8(fr, k, 5)
    for n in (16, 48, 57):
        frames = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
        src = device.DeviceBatch(c, n, rows, cols, 3)
        src.upload(frames)
        a = device.DeviceBatch(c, n, rows, cols, 3)
        b = device.DeviceBatch(c, n + 6, rows, cols, 3)
        a.memset(0); b.memset(0)
        c.sync()
        # (a)
        L.rcv__debug_kernels_reset()
        device.filter2d(src, a, k, shift=5)
        assert L.rcv__debug_kernels().decode().count("k_filter_rows_chain<7") == 2, L.rcv__debug_kernels().decode()
        got = a.download()
        want1 = [f1(frames[i]) for i in range(n)]
        assert all(np.array_equal(got[i], want1[i]) for i in range(n))
        # (b) src -> a -> b' -> a, no wait in between: RAW and WAR across calls, each half against its own predecessor
        bv = b.view(0, n)
        device.filter2d(src, a, k, shift=5)
        device.filter2d(a, bv, k2, shift=4)
        device.filter2d(bv, a, k, shift=5)
        got = a.download()
        for i in (0, n // 2 - 1, n // 2, n - 1):
            assert np.array_equal(got[i], f1(oracle.filter2d_i8(want1[i], k2, 4))), ("chain", n, i)
        # (c) the destination of call 1, read SHIFTED by 6 frames by call 2: frame j of call 2's first half is frame j + 6 of call 1's output,
        # and the frames around n / 2 - 6 .. n / 2 were written by call 1's OTHER half
        b.memset(0)
        device.filter2d(src, b.view(6, n), k, shift=5)            # b[6 + i] = f1(frames[i])
        device.filter2d(b.view(0, n), a, k2, shift=4)             # a[j] = f2(b[j]) ; b[j] = f1(frames[j - 6]) for j >= 6, zeros below
        got = a.download()
        zero = np.zeros((rows, cols, 3), np.uint8)
        for j in (0, 5, 6, n // 2 - 7, n // 2 - 1, n // 2, n // 2 + 5, n - 1):
            assert np.array_equal(got[j], oracle.filter2d_i8(want1[j - 6] if j >= 6 else zero, k2, 4)), ("shifted", n, j)
        # (d) another entry point right behind a split call
        gray_dx = device.DeviceBatch(c, n, rows, cols, 1, _ffi.RCV_16S)
        gray_dy = device.DeviceBatch(c, n, rows, cols, 1, _ffi.RCV_16S)
        device.filter2d(src, a, k, shift=5)
        device.sobel(a, gray_dx, gray_dy)
        gx = gray_dx.download()
        for i in (0, n // 2, n - 1):
            assert np.array_equal(gx[i].reshape(rows, cols), oracle.sobel(oracle.bgr2gray(want1[i]))[0].reshape(rows, cols)), ("sobel behind", n, i)
        # (e) one launch when asked / when another context of the device is busy
        knob("RCV_FR_SPLIT", 0)
        L.rcv__debug_kernels_reset()
        device.filter2d(src, a, k, shift=5)
        assert L.rcv__debug_kernels().decode().count("k_filter_rows_chain<7") == 1
        assert np.array_equal(a.download()[n - 1], want1[n - 1])
        knob("RCV_FR_SPLIT", -1)
        c2 = rcv.Context(0)
        junk = device.DeviceBatch(c2, 1, 64, 64, 3)
        junk.memset(1)                       # c2 has enqueued and not waited: busy
        c.sync()
        L.rcv__debug_kernels_reset()
        device.filter2d(src, a, k, shift=5)
        assert L.rcv__debug_kernels().decode().count("k_filter_rows_chain<7") == 1, "split although a second context is busy"
        c2.sync()
        c.sync()
        L.rcv__debug_kernels_reset()
        device.filter2d(src, a, k, shift=5)
        assert L.rcv__debug_kernels().decode().count("k_filter_rows_chain<7") == 2
        assert np.array_equal(a.download()[n // 2], want1[n // 2])
        junk.free()
        c2.close()
        for x in (src, a, b, gray_dx, gray_dy):
            x.free()
    c.close()


@pytest.mark.parametrize("want_resp", [False, True])
def test_harris_pipeline_split_call(oracle, knob, want_resp):
    """round 6: the fused Harris pipeline of 16+ frames runs as two halves on the context's two streams as well (-1.6 %; the Sobel, filter -> Sobel and
    warp -> resize calls measured slower split and are not: tools/ab_split_ops.py).  Two launches per call, every frame equals the oracle's mask (and
    response); a call right behind it that reads the masks (NMS input of another op: a download here) sees both halves; RCV_FR_SPLIT=0: one launch."""
    import rustcv_amd as rcv
    c = rcv.Context(0)
    L = _ffi.lib()
    n, rows, cols = 19, 70, 496
    frames = np.stack([oracle.synth_frame(rows, cols, 3, 1, 0x5EED0005, i) for i in range(n)])
    src = device.DeviceBatch(c, n, rows, cols, 3)
    src.upload(frames)
    mask = device.DeviceBatch(c, n, rows, cols, 1)
    resp = device.DeviceBatch(c, n, rows, cols, 1, _ffi.RCV_32F) if want_resp else None
    for split in (-1, 0):
        knob("RCV_FR_SPLIT", split)
        mask.memset(7)
        c.sync()
        L.rcv__debug_kernels_reset()
        device.harris_pipeline(src, mask, resp, 2, 0.04, 1e-4)
        device.harris_pipeline(src, mask, resp, 2, 0.04, 1e-4)     # (back to back: the second call's halves behind the first call's)
        assert L.rcv__debug_kernels().decode().count("k_harris_fused") == (4 if split else 2), L.rcv__debug_kernels().decode()
        got = mask.download()
        gr = resp.download() if want_resp else None
        for i in range(n):
            want = oracle.harris_pipeline(frames[i], 2, 0.04, 1e-4, want_resp=want_resp)
            wm = want[0] if want_resp else want
            assert np.array_equal(got[i].reshape(rows, cols), wm.reshape(rows, cols)), (split, i)
            if want_resp:
                assert np.array_equal(gr[i].reshape(rows, cols).view(np.uint32), want[1].reshape(rows, cols).view(np.uint32)), (split, i)
    for b in (src, mask) + ((resp,) if want_resp else ()):
        b.free()
    c.close()


def test_filter_rows_chain_ticket_accounting(oracle, knob):
    """the chained-band kernel's ticket counters (round 5: four sets; launch i draws from set i % 4, found zero, and zeroes set (i + 2) % 4;
    no host-side count of what a launch draws).  Sixty launches of four different geometries (different item counts, one / three edge
    strips, a single-strip image whose interior queue is empty) and three kernel sizes queued back to back WITHOUT a sync on a fresh
    context, every result against the oracle: a counter / base mismatch would make later launches skip or repeat items"""
    import rustcv_amd as rcv
    knob("RCV_FR_CHAIN", 1)
    knob("RCV_F7_ROWS", 1)
    c = rcv.Context(0)
    r = np.random.default_rng(4711 + _SOAK_SEED)
    shapes = [(8, 64, 256), (8, 90, 1040), (16, 72, 772), (8, 130, 2304)]
    srcs, frames = [], []
    for n, rows, cols in shapes:
        f = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
        b = device.DeviceBatch(c, n, rows, cols, 3)
        b.upload(f)
        srcs.append(b)
        frames.append(f)
    L = _ffi.lib()
    jobs = []
    for it in range(60):
        si = int(r.integers(0, len(shapes)))
        ks = int(r.choice([3, 5, 7]))
        k = r.integers(-9, 10, size=(ks, ks)).astype(np.int8)
        n, rows, cols = shapes[si]
        dst = device.DeviceBatch(c, n, rows, cols, 3)
        L.rcv__debug_kernels_reset()
        device.filter2d(srcs[si], dst, k, shift=5)            # no sync: the queue holds launches of every geometry
        assert "k_filter_rows_chain<" in L.rcv__debug_kernels().decode()
        jobs.append((si, k, dst))
    c.sync()
    for si, k, dst in jobs:
        got = dst.download()
        for i in (0, shapes[si][0] - 1):
            assert np.array_equal(got[i], oracle.filter2d_i8(frames[si][i], k, 5)), (si, k.shape, i)
        dst.free()
    for b in srcs:
        b.free()
    c.close()


def test_row_kernel_weight_table_cache(ctx, oracle, knob):
    """round 3: the row-streaming kernel caches the banded weight tables of FOUR kernels per context and uploads a
    new one stream-ordered without synchronising; a caller cycling through six kernels (more than the cache holds: entries are
    replaced while earlier launches that read them are still queued) gets every result right, with no sync between the calls"""
    knob("RCV_F7_ROWS")
    knob("RCV_GAUSS_ROWS", 0)
    r = np.random.default_rng(4242 + _SOAK_SEED)
    n, rows, cols = 2, 96, 272
    frames = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    src.upload(frames)
    kernels = [(r.integers(-9, 10, size=(ks, ks)).astype(np.int8), ks, sh) for ks, sh in ((7, 6), (5, 4), (3, 2), (7, 5), (5, 0), (7, 7))]
    outs = [device.DeviceBatch(ctx, n, rows, cols, 3) for _ in range(3 * len(kernels) + 3)]
    calls = []
    for rep in range(3):
        for k, ks, sh in kernels:
            device.filter2d(src, outs[len(calls)], k, shift=sh)            # no sync: the queue holds launches of all six kernels
            calls.append(("f", k, sh))
    for ks in (3, 5, 7):
        device.gaussian_blur(src, outs[len(calls)], ks, 0.0)               # (7: two weight tables)
        calls.append(("g", ks, 0))
    ctx.sync()
    for out, (kind, k, sh) in zip(outs, calls):
        got = out.download()
        for i in range(n):
            want = oracle.filter2d_i8(frames[i], k, sh) if kind == "f" else oracle.gaussian_blur(frames[i], k, 0.0)
            assert np.array_equal(got[i], want), (kind, sh)
    for b in outs + [src]:
        b.free()


GAUSS_ROWS_SHAPES = [(3, 16, 3), (5, 32, 3), (33, 240, 3), (40, 336, 3), (19, 496, 3), (70, 672, 3), (300, 1008, 3), (9, 2000, 3), (1080, 1920, 3),
                     (3, 32, 1), (17, 992, 1), (40, 1008, 1), (64, 4000, 1), (7, 48, 1)]


@pytest.mark.parametrize("rows,cols,ch", GAUSS_ROWS_SHAPES)
@pytest.mark.parametrize("ksize", [3, 5])
@pytest.mark.parametrize("seg", [0, 4, 5, 13])
def test_gaussian_int_rows_kernel(ctx, oracle, rng, knob, rows, cols, ch, ksize, seg):
    """round 3: the register-window integer Gaussian for small launches (rcv_gauss_rows.hip; BASELINE config 2).  Strips of 992
    bytes: one strip, a second strip of ONE lane (336 px BGR / 1008 px gray), a strip that ends exactly at the row end (992 px gray);
    segment seams at every height (RCV_GR_SEG), rows fewer than one segment; batch of 3 with padded steps and canaries; a
    saturated region (sums reach 255 * D exactly); the full 1080p frame of config 2."""
    knob("RCV_GAUSS_ROWS")
    if seg:
        knob("RCV_GR_SEG", seg)
    n = 1 if rows >= 1000 else 3
    img = rand_img(rng, rows, cols, ch).reshape(rows, cols, ch)
    img[: rows // 3] = 255
    img[:, :2] = 7
    frames = np.stack([np.roll(img, 3 * i, axis=1) if i < 2 else img[::-1].copy() for i in range(n)])
    src = device.DeviceBatch(ctx, n, rows, cols, ch, step=cols * ch + 32)
    dst = _canary_batch(ctx, n, rows, cols, ch, pad=48)
    src.upload(frames if ch == 3 else frames[..., 0])
    _ffi.lib().rcv__debug_kernels_reset()
    device.gaussian_blur(src, dst, ksize, 0.0)
    assert "k_gauss_rows<" in _ffi.lib().rcv__debug_kernels().decode()
    got = dst.download()
    for i in range(n):
        want = oracle.gaussian_blur(frames[i] if ch == 3 else frames[i][..., 0], ksize, 0.0)
        assert np.array_equal(got[i], want), (i, np.argwhere(got[i] != want)[:5])
    _assert_canaries(dst)
    src.free()
    dst.free()


def test_gaussian_int_rows_kernel_takes_config2_by_default(ctx, oracle):
    """BASELINE config 2 (one 1080p BGR frame, 5x5, sigma 0) runs on the register-window kernel without any knob; launches that
    fill the GPU several times over stay on the MFMA kernels; ksize 7 (sums beyond 16 bits) and unaligned rows never take it"""
    L = _ffi.lib()
    one, one2 = device.DeviceBatch(ctx, 1, 1080, 1920, 3), device.DeviceBatch(ctx, 1, 1080, 1920, 3)
    device.synth(one, 0, 0x5EED0002, 0)

    def kernel_of(fn):
        L.rcv__debug_kernels_reset()
        fn()
        return L.rcv__debug_kernels().decode()
    assert "k_gauss_rows<5, 3>" in kernel_of(lambda: device.gaussian_blur(one, one2, 5, 0.0))
    assert np.array_equal(one2.download()[0], oracle.gaussian_blur(one.download()[0], 5, 0.0))
    assert "k_gauss_rows<3, 3>" in kernel_of(lambda: device.gaussian_blur(one, one2, 3, 0.0))
    assert "k_gauss_rows" not in kernel_of(lambda: device.gaussian_blur(one, one2, 7, 0.0))
    big, big2 = device.DeviceBatch(ctx, 64, 1080, 1920, 3), device.DeviceBatch(ctx, 64, 1080, 1920, 3)
    assert "k_gauss_rows" not in kernel_of(lambda: device.gaussian_blur(big, big2, 5, 0.0))
    odd, odd2 = device.DeviceBatch(ctx, 1, 64, 1920, 3, step=1920 * 3 + 4), device.DeviceBatch(ctx, 1, 64, 1920, 3)
    assert "k_gauss_rows" not in kernel_of(lambda: device.gaussian_blur(odd, odd2, 5, 0.0))
    for b in (one, one2, big, big2, odd, odd2):
        b.free()


@pytest.mark.parametrize("rows,cols", [(4, 16), (33, 240), (19, 496), (300, 272)])
@pytest.mark.parametrize("ksize", [3, 5, 7])
@pytest.mark.parametrize("full_tables", [False, True, "rows"])
def test_gaussian_int_mfma_path(ctx, oracle, rng, knob, rows, cols, ksize, full_tables):
    """integer GaussianBlur on the MFMA strip kernel.  ksize 7 has weights up to 324: two weight tables, by default the
    centre split K = K1 + 2*T2 (second table in kernel rows 2..4 only), with RCV_F7_DUAL_FULL the general K = 4Q + R"""
    knob("RCV_GAUSS_ROWS", 0)   # (these small shapes would otherwise take the register-window kernel of rcv_gauss_rows.hip)
    if full_tables == "rows":
        knob("RCV_F7_ROWS")   # (ksize 3 / 5: weights inside i8 -> the row-streaming kernel; 7 stays on the two-table strip kernel)
    elif full_tables:
        knob("RCV_F7_DUAL_FULL")
    img = rand_img(rng, rows, cols, 3)
    img[: rows // 2] = 255  # saturating region: sums reach 255 * D exactly
    src, dst = Mat.from_array(img), Mat(rows, cols, 3)
    imgproc.gaussian_blur(src, dst, ksize, 0.0, ctx)
    assert np.array_equal(dst.to_array(), oracle.gaussian_blur(img, ksize, 0.0))


@pytest.mark.parametrize("rows,cols", [(4, 16), (5, 32), (33, 240), (40, 256), (19, 496), (70, 512), (300, 272), (9, 10), (6, 770)])
@pytest.mark.parametrize("ksize,shift", [(7, 6), (3, 0), (5, 4)])
@pytest.mark.parametrize("kernel", ["strip", "rows"])
def test_fused_yuyv_filter(ctx, oracle, rng, knob, rows, cols, ksize, shift, kernel):
    """f1: YUYV -> BGR -> filter2D in one launch == the two-step oracle composition (fused MFMA path when cols % 16 == 0 -- the
    strip kernel for these small launches, the row-streaming kernel with RCV_F7_ROWS=1 -- the unfused HIP path otherwise), batch
    of 2 with padded steps"""
    if kernel == "rows":
        knob("RCV_F7_ROWS")
    n = 2
    k = rng.integers(-9, 10, size=(ksize, ksize)).astype(np.int8)
    src = device.DeviceBatch(ctx, n, rows, cols, 2, step=cols * 2 + 16)
    dst = _canary_batch(ctx, n, rows, cols, 3, pad=16)
    frames = rng.integers(0, 256, size=(n, rows, cols, 2), dtype=np.uint8)
    src.upload(frames)
    device.filter2d_yuyv(src, dst, k, shift=shift)
    got = dst.download()
    for i in range(n):
        bgr = np.zeros(rows * cols * 3, np.uint8)
        oracle.yuv422_to_bgr_strided(frames[i].reshape(-1), cols * 2, rows, cols, False, bgr)
        assert np.array_equal(got[i], oracle.filter2d_i8(bgr.reshape(rows, cols, 3), k, shift))
    _assert_canaries(dst)
    src.free()
    dst.free()


@pytest.mark.parametrize("pipelined", [False, True, "rows"])
def test_filter2d_i8_mfma_random_shapes(ctx, oracle, knob, pipelined):
    """(latency variant of the kernel for these small launches, and -- RCV_F7_NO_LAT -- the pipelined one)  40 x RCV_SOAK seeded random cases for the MFMA strip kernel: widths 16..1040 (multiples of 16: partial last strips, one to five strips),
    heights 4..150 (one to several 16-row steps, ragged last step), ksize 3/5/7, weights over the full i8 range, shifts 0..12,
    padded steps, batch 1..3, BGR and YUYV sources"""
    if pipelined == "rows":
        knob("RCV_F7_ROWS")   # (BGR and YUYV sources on the row-streaming kernel)
    elif pipelined:
        knob("RCV_F7_NO_LAT")
    r = np.random.default_rng(0xF17E7 + _SOAK_SEED)
    for case in range(40 * _SOAK):
        cols = 16 * int(r.integers(1, 66))
        rows = int(r.integers(4, 151))
        ksize = int(r.choice([3, 5, 7]))
        shift = int(r.integers(0, 13))
        n = int(r.integers(1, 4))
        k = r.integers(-128, 128, size=(ksize, ksize)).astype(np.int8)
        yuyv = bool(case % 4 == 3)
        sch = 2 if yuyv else 3
        pad = 16 * int(r.integers(0, 3))
        src = device.DeviceBatch(ctx, n, rows, cols, sch, step=cols * sch + pad)
        dst = _canary_batch(ctx, n, rows, cols, 3, pad=16)
        frames = r.integers(0, 256, size=(n, rows, cols, sch), dtype=np.uint8)
        src.upload(frames)
        if yuyv:
            device.filter2d_yuyv(src, dst, k, shift=shift)
        else:
            device.filter2d(src, dst, k, shift=shift)
        got = dst.download()
        for i in range(n):
            if yuyv:
                bgr = np.zeros(rows * cols * 3, np.uint8)
                oracle.yuv422_to_bgr_strided(frames[i].reshape(-1), cols * 2, rows, cols, False, bgr)
                ref = bgr.reshape(rows, cols, 3)
            else:
                ref = frames[i]
            assert np.array_equal(got[i], oracle.filter2d_i8(ref, k, shift)), (case, rows, cols, ksize, shift, n, yuyv)
        _assert_canaries(dst)
        src.free()
        dst.free()


def test_gray_dot4_filter_random_shapes(ctx, oracle):
    """20 x RCV_SOAK seeded random cases for the one-channel dot4 streaming kernel: widths 12..1000 (multiples of 4), heights 1..150,
    ksize 3/5/7, weights over the i8 range, integer GaussianBlur 3/5/7 (7x7 takes the split-weight path), padded steps, batch 1..3"""
    r = np.random.default_rng(0x6A47 + _SOAK_SEED)
    for case in range(20 * _SOAK):
        cols = 4 * int(r.integers(3, 251))
        rows = int(r.integers(1, 151))
        n = int(r.integers(1, 4))
        ksize = int(r.choice([3, 5, 7]))
        pad = 4 * int(r.integers(0, 4))
        frames = r.integers(0, 256, size=(n, rows, cols, 1), dtype=np.uint8)
        src = device.DeviceBatch(ctx, n, rows, cols, 1, step=cols + pad)
        dst = _canary_batch(ctx, n, rows, cols, 1, pad=int(r.choice([4, 8, 12])))
        src.upload(frames)
        if case % 3 == 2:
            device.gaussian_blur(src, dst, ksize, 0.0)
            want = [oracle.gaussian_blur(frames[i, :, :, 0], ksize, 0.0) for i in range(n)]
        else:
            k = r.integers(-128, 128, size=(ksize, ksize)).astype(np.int8)
            shift = int(r.integers(0, 13))
            device.filter2d(src, dst, k, shift=shift)
            want = [oracle.filter2d_i8(frames[i, :, :, 0], k, shift) for i in range(n)]
        got = dst.download()
        for i in range(n):
            assert np.array_equal(got[i].reshape(rows, cols), want[i]), (case, rows, cols, ksize)
        _assert_canaries(dst)
        src.free()
        dst.free()


GRAY_MFMA_COLS = [16, 32, 48, 64, 240, 752, 768, 784, 800, 816, 1520, 1536, 1552, 1568, 2320]


@pytest.mark.parametrize("cols", GRAY_MFMA_COLS)
@pytest.mark.parametrize("dot4", [False, True])
@pytest.mark.parametrize("pipelined", [False, True, "rows"])
def test_gray_filter_strip_kernel(ctx, oracle, knob, cols, dot4, pipelined):
    """one-channel images on 16-byte aligned rows take the MFMA strip kernel's gray variant (768-pixel strips of 48 tiles):
    widths around the strip seams (one to three + strips; a last strip with 1, 2, 3, ... tiles -- every position ntiles % 3 of the
    right border inside a lane's 48-pixel chunk -- and full last strips, whose border sits in the halo piece), heights across
    several 16-row steps, ksize 3/5/7, full-range weights, the two-table integer Gaussian, padded steps, batches.
    RCV_F7_NO_GRAY sends the same cases through the dot4 streaming kernel."""
    if dot4:
        if pipelined:
            pytest.skip("the dot4 kernel has one variant")
        knob("RCV_F7_NO_GRAY")
    if pipelined == "rows":
        knob("RCV_F7_ROWS")     # the row-streaming kernel's gray variant (three 256-pixel blocks per strip; two-table weights: strip kernel)
    elif pipelined:
        knob("RCV_F7_NO_LAT")   # (small launches would otherwise all take the strip kernel's latency variant)
    r = np.random.default_rng(0x6A4700 + cols + _SOAK_SEED)
    for case in range(max(2, _SOAK // 2)):
        rows = int(r.integers(4, 120))
        n = int(r.integers(1, 4))
        ksize = int(r.choice([3, 5, 7]))
        spad, dpad = 16 * int(r.integers(0, 3)), 16 * int(r.integers(0, 3))
        frames = r.integers(0, 256, size=(n, rows, cols, 1), dtype=np.uint8)
        if case % 4 == 1:
            frames[:, :, cols // 2:] = 255       # saturating half
        src = device.DeviceBatch(ctx, n, rows, cols, 1, step=cols + spad)
        dst = _canary_batch(ctx, n, rows, cols, 1, pad=dpad)
        src.upload(frames)
        if case % 3 == 2:
            device.gaussian_blur(src, dst, ksize, 0.0)
            want = [oracle.gaussian_blur(frames[i, :, :, 0], ksize, 0.0) for i in range(n)]
        else:
            k = r.integers(-128, 128, size=(ksize, ksize)).astype(np.int8)
            shift = int(r.integers(0, 13))
            device.filter2d(src, dst, k, shift=shift)
            want = [oracle.filter2d_i8(frames[i, :, :, 0], k, shift) for i in range(n)]
        got = dst.download()
        for i in range(n):
            assert np.array_equal(got[i].reshape(rows, cols), want[i]), (case, rows, cols, ksize)
        _assert_canaries(dst)
        src.free()
        dst.free()


def test_gray_filter_strip_kernel_4k_batch(ctx, oracle):
    """BASELINE-sized gray batch (4K x 8 frames, 5 full strips per row): row slabs of two frames against the oracle"""
    rows, cols, n = 2160, 3840, 8
    src = device.DeviceBatch(ctx, n, rows, cols, 1)
    dst = device.DeviceBatch(ctx, n, rows, cols, 1)
    device.synth(src, 0, 0x5EED0077, 0)
    k = oracle.bench_kernel7()
    device.filter2d(src, dst, k, shift=6)
    frames, got = src.download(), dst.download()
    for i in (0, n - 1):
        for y0, y1 in ((0, 40), (1060, 1100), (2120, 2160)):
            lo, hi = max(0, y0 - 3), min(rows, y1 + 3)
            want = oracle.filter2d_i8(frames[i][lo:hi], k, 6) if (lo == 0 or hi == rows) else None
            if want is None:
                want = oracle.filter2d_i8(frames[i][lo:hi], k, 6)[y0 - lo: y1 - lo]
                assert np.array_equal(got[i][y0:y1], want), (i, y0)
            elif lo == 0:
                assert np.array_equal(got[i][y0:y1], want[: y1 - y0]), (i, y0)
            else:
                assert np.array_equal(got[i][y0:y1], want[y0 - lo:]), (i, y0)
    src.free()
    dst.free()


@pytest.mark.parametrize("ch", [1, 3])
def test_integer_filters_on_the_streaming_kernel(ctx, oracle, ch):
    """integer filter2D / GaussianBlur on shapes the MFMA strip kernel refuses (packed rows that are only 4-byte aligned, widths
    that are not a multiple of 16 -- e.g. a packed 1080-pixel-wide BGR image): they run on the streaming f32 kernel in its
    exact integer mode.  Full-range weights (negative sums, saturation at both ends), shifts 0..12, integer Gaussian 3/5/7."""
    r = np.random.default_rng(0x51E4 + ch + _SOAK_SEED)
    for case in range(8 * _SOAK):
        cols = int(r.choice([20, 36, 100, 1080, 340, 52, 1364])) if ch == 3 else 4 * int(r.integers(3, 300)) + (0 if case % 2 else 4)
        if ch == 3 and (cols * 3) % 4:
            cols += 4 - cols % 4
        rows = int(r.integers(1, 90))
        n = int(r.integers(1, 3))
        ksize = int(r.choice([3, 5, 7]))
        frames = r.integers(0, 256, size=(n, rows, cols, ch), dtype=np.uint8)
        if case % 5 == 0:
            frames[:, :, : cols // 2] = 255
        pad = 4 * int(r.integers(0, 3)) + (4 if (cols * ch) % 16 == 0 else 0)      # never a 16-byte aligned step
        src = device.DeviceBatch(ctx, n, rows, cols, ch, step=cols * ch + pad)
        dst = _canary_batch(ctx, n, rows, cols, ch, pad=4 * int(r.integers(1, 3)) + (0 if (cols * ch) % 16 else 0))
        src.upload(frames)
        fr = [frames[i] if ch == 3 else frames[i, :, :, 0] for i in range(n)]
        if case % 3 == 2:
            device.gaussian_blur(src, dst, ksize, 0.0)
            want = [oracle.gaussian_blur(f, ksize, 0.0) for f in fr]
        else:
            k = r.integers(-128, 128, size=(ksize, ksize)).astype(np.int8)
            shift = int(r.integers(0, 13))
            device.filter2d(src, dst, k, shift=shift)
            want = [oracle.filter2d_i8(f, k, shift) for f in fr]
        got = dst.download()
        for i in range(n):
            assert np.array_equal(got[i].reshape(want[i].shape), want[i]), (case, rows, cols, ksize)
        _assert_canaries(dst)
        src.free()
        dst.free()


def test_register_window_kernels_random_shapes(ctx, oracle):
    """24 x RCV_SOAK seeded random shapes through the Sobel (gray and BGR source) and Harris (BGR and YUYV source) sliding-window kernels:
    widths 8..1600 (multiples of 8: one to four strips, partial last strip), heights 4..260 (several row segments), padded steps"""
    r = np.random.default_rng(0x50BE1 + _SOAK_SEED)
    for case in range(24 * _SOAK):
        cols = 8 * int(r.integers(1, 201))
        rows = int(r.integers(4, 261))
        n = int(r.integers(1, 3))
        bgr = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
        pad = 8 * int(r.integers(0, 3))
        if case % 2 == 0:     # Sobel, gray or BGR source
            from_bgr = bool(case % 4 == 0)
            src = device.DeviceBatch(ctx, n, rows, cols, 3 if from_bgr else 1, step=cols * (3 if from_bgr else 1) + pad)
            grays = np.stack([oracle.bgr2gray(f) for f in bgr])
            src.upload(bgr if from_bgr else grays[..., None])
            dx = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_16S, pad=16)
            dy = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_16S, pad=16)
            device.sobel(src, dx, dy)
            gx, gy = dx.download(), dy.download()
            for i in range(n):
                wx, wy = oracle.sobel(grays[i])
                assert np.array_equal(gx[i], wx) and np.array_equal(gy[i], wy), (case, rows, cols, from_bgr)
            _assert_canaries(dx)
            _assert_canaries(dy)
            for b in (src, dx, dy):
                b.free()
        else:                 # Harris pipeline, BGR or YUYV source, with the response
            from_yuyv = bool(case % 4 == 1)
            if from_yuyv:
                yuyv = r.integers(0, 256, size=(n, rows, cols, 2), dtype=np.uint8)
                src = device.DeviceBatch(ctx, n, rows, cols, 2, step=cols * 2 + pad)
                src.upload(yuyv)
                refs = []
                for i in range(n):
                    t = np.zeros(rows * cols * 3, np.uint8)
                    oracle.yuv422_to_bgr_strided(yuyv[i].reshape(-1), cols * 2, rows, cols, False, t)
                    refs.append(t.reshape(rows, cols, 3))
            else:
                src = device.DeviceBatch(ctx, n, rows, cols, 3, step=cols * 3 + pad)
                src.upload(bgr)
                refs = list(bgr)
            mask = _canary_batch(ctx, n, rows, cols, 1, pad=8)
            resp = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F)
            device.harris_pipeline(src, mask, resp, 2, 0.04, 1e-4)
            gm, gr = mask.download(), resp.download()
            for i in range(n):
                wm, wr = oracle.harris_pipeline(refs[i], 2, 0.04, 1e-4, True)
                assert np.array_equal(gr[i].view(np.uint32), wr.view(np.uint32)), (case, rows, cols, from_yuyv)
                assert np.array_equal(gm[i], wm), (case, rows, cols, from_yuyv)
            _assert_canaries(mask)
            for b in (src, mask, resp):
                b.free()


def test_geometry_kernels_random_maps(ctx, oracle):
    """20 x RCV_SOAK seeded random affine maps / scales through the BGR warp, general resize and fused warp->down-scale kernels (output widths
    multiples of 4): rotations, shears, scales 0.3..3, translations that push part or all of the footprint outside"""
    r = np.random.default_rng(0x6E07 + _SOAK_SEED)
    for case in range(20 * _SOAK):
        sr, sc = int(r.integers(6, 200)), int(r.integers(6, 300))
        dr, dc = int(r.integers(2, 120)), 4 * int(r.integers(1, 60))
        img = r.integers(0, 256, size=(sr, sc, 3), dtype=np.uint8)
        th, sx, sy = r.uniform(-3.2, 3.2), r.uniform(0.3, 3.0), r.uniform(0.3, 3.0)
        M = np.array([sx * np.cos(th), -sy * np.sin(th) + r.uniform(-0.2, 0.2), r.uniform(-0.6 * sc, 0.6 * sc),
                      sx * np.sin(th), sy * np.cos(th), r.uniform(-0.6 * sr, 0.6 * sr)], np.float32)
        src = device.DeviceBatch(ctx, 1, sr, sc, 3)
        src.upload(img[None])
        dst = _canary_batch(ctx, 1, dr, dc, 3, pad=8)
        device.warp_affine(src, dst, M)
        assert np.array_equal(dst.download()[0], oracle.warp_affine(img, M, dr, dc)), (case, "warp", sr, sc, dr, dc)
        _assert_canaries(dst)
        device.resize(src, dst)
        assert np.array_equal(dst.download()[0], oracle.resize(img, dr, dc)), (case, "resize", sr, sc, dr, dc)
        _assert_canaries(dst)
        S = 2 if case % 2 else 4
        device.warp_affine_resize(src, dst, M, S * dr, S * dc)
        assert np.array_equal(dst.download()[0], oracle.resize(oracle.warp_affine(img, M, S * dr, S * dc), dr, dc)), (case, "fused", S)
        _assert_canaries(dst)
        src.free()
        dst.free()


def test_warp_affine_gray_random_maps(ctx, oracle):
    """(and random scales through the one-channel resize kernel)  20 x RCV_SOAK seeded random affine maps through the one-channel warp kernel (output widths multiples of 4): interior waves
    (aligned 8-byte tap windows), border waves (per-pixel path), footprints partly or wholly outside, padded steps, batch of 2"""
    r = np.random.default_rng(0x6E0761 + _SOAK_SEED)
    for case in range(20 * _SOAK):
        sr, sc = int(r.integers(6, 260)), int(r.integers(8, 600))
        dr, dc = int(r.integers(2, 150)), 4 * int(r.integers(1, 130))
        img = r.integers(0, 256, size=(2, sr, sc, 1), dtype=np.uint8)
        th, sx, sy = r.uniform(-3.2, 3.2), r.uniform(0.3, 3.0), r.uniform(0.3, 3.0)
        if case % 3 == 0:     # near-identity maps keep most waves on the interior path
            th, sx, sy = r.uniform(-0.2, 0.2), r.uniform(0.9, 1.1), r.uniform(0.9, 1.1)
        M = np.array([sx * np.cos(th), -sy * np.sin(th) + r.uniform(-0.2, 0.2), r.uniform(-0.3 * sc, 0.3 * sc),
                      sx * np.sin(th), sy * np.cos(th), r.uniform(-0.3 * sr, 0.3 * sr)], np.float32)
        src = device.DeviceBatch(ctx, 2, sr, sc, 1, step=sc + int(r.choice([0, 3, 4, 13])))
        src.upload(img)
        dst = _canary_batch(ctx, 2, dr, dc, 1, pad=8)
        device.warp_affine(src, dst, M)
        got = dst.download()
        for i in range(2):
            assert np.array_equal(got[i], oracle.warp_affine(img[i, :, :, 0], M, dr, dc)), (case, i, sr, sc, dr, dc)
        _assert_canaries(dst)
        device.resize(src, dst)          # the one-channel general resize kernel (any scale, up and down)
        got = dst.download()
        for i in range(2):
            assert np.array_equal(got[i], oracle.resize(img[i, :, :, 0], dr, dc)), (case, "resize", i, sr, sc, dr, dc)
        _assert_canaries(dst)
        src.free()
        dst.free()


def test_filter2d_i8_mfma_batch_4k_properties(ctx, oracle):
    """Full-size frames (BASELINE configs[2] shape, small batch): (1) rows of frame 0 against the oracle on
    slabs; (2) linearity: filter(K1) + filter(K2) == filter(K1+K2) where nothing saturates (shift 0 is not
    usable at u8, so use a delta kernel: identity must reproduce the input exactly)."""
    rows, cols, n = 2160, 3840, 3
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    ident = np.zeros((7, 7), np.int8)
    ident[3, 3] = 64
    device.filter2d(src, dst, ident, shift=6)
    a, b = src.download(), dst.download()
    assert np.array_equal(a, b)                      # identity kernel: output == input, every frame
    # shifted delta: output == input shifted with REFLECT_101 (checks halo + borders at full size)
    sh = np.zeros((7, 7), np.int8)
    sh[0, 6] = 1                                     # tap (ky=0,kx=6): out[y,x] = in[y-3, x+3]
    device.filter2d(src, dst, sh, shift=0)
    b = dst.download()
    yy = np.abs(np.arange(rows) - 3)
    xx = np.arange(cols) + 3
    xx = np.where(xx >= cols, 2 * cols - 2 - xx, xx)
    assert np.array_equal(b, a[:, yy][:, :, xx])
    # bench kernel on slabs of frame 1 vs the oracle (top, a segment seam, bottom)
    k = oracle.bench_kernel7()
    device.filter2d(src, dst, k, shift=6)
    b = dst.download()
    want = oracle.filter2d_i8(a[1], k, 6)
    assert np.array_equal(b[1], want)
    src.free()
    dst.free()


def test_filter2d_i8_extremes(ctx, oracle):
    img = np.full((40, 56, 3), 255, np.uint8)
    for k in (np.full((7, 7), 127, np.int8), np.full((7, 7), -128, np.int8)):
        for shift in (0, 6, 24):
            dst = Mat(40, 56, 3)
            imgproc.filter2d(Mat.from_array(img), dst, k, shift=shift, ctx=ctx)
            assert np.array_equal(dst.to_array(), oracle.filter2d_i8(img, k, shift))


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 5), (17, 33), (61, 127)])
@pytest.mark.parametrize("ksize", [1, 3, 7])
def test_filter2d_f32(ctx, oracle, rng, rows, cols, ksize):
    img = rand_img(rng, rows, cols, 3)
    k = (rng.standard_normal((ksize, ksize)) / ksize).astype(np.float32)
    src, dst = Mat.from_array(img), Mat(rows, cols, 3)
    imgproc.filter2d(src, dst, k, delta=0.5, ctx=ctx)
    assert np.array_equal(dst.to_array(), oracle.filter2d_f32(img, k, 0.5))


def test_f32_stream_kernels_random_shapes(ctx, oracle):
    """30 x RCV_SOAK seeded random cases for the f32 streaming kernels (dense filter2D and separable Gaussian): 1 and 3 channels, ksize 3..11,
    row bytes a multiple of 4 or of 8 (4- and 8-byte-per-thread variants, EDGE threads), 8-byte aligned and unaligned steps"""
    r = np.random.default_rng(0xF32 + _SOAK_SEED)
    for case in range(30 * _SOAK):
        ch = int(r.choice([1, 3]))
        cols = 4 * int(r.integers(3, 120)) if ch == 1 else 4 * int(r.integers(3, 80))
        rows = int(r.integers(1, 120))
        n = int(r.integers(1, 3))
        pad = int(r.choice([0, 4, 8, 16]))
        img = r.integers(0, 256, size=(n, rows, cols, ch), dtype=np.uint8)
        src = device.DeviceBatch(ctx, n, rows, cols, ch, step=cols * ch + pad)
        dst = _canary_batch(ctx, n, rows, cols, ch, pad=int(r.choice([4, 8])))
        src.upload(img)
        if case % 2 == 0:
            ksize = int(r.choice([3, 5, 7]))
            k = (r.standard_normal((ksize, ksize)) / ksize).astype(np.float32)
            delta = float(r.uniform(-20, 20))
            device.filter2d(src, dst, k, delta=delta)
            want = [oracle.filter2d_f32(img[i].reshape(rows, cols, ch) if ch > 1 else img[i, :, :, 0], k, delta) for i in range(n)]
        else:
            ksize = int(r.choice([3, 5, 7, 9, 11]))
            sigma = float(r.uniform(0.4, 3.0))
            device.gaussian_blur(src, dst, ksize, sigma)
            want = [oracle.gaussian_blur(img[i].reshape(rows, cols, ch) if ch > 1 else img[i, :, :, 0], ksize, sigma) for i in range(n)]
        got = dst.download()
        for i in range(n):
            g = got[i] if ch > 1 else got[i].reshape(rows, cols)
            assert np.array_equal(g, want[i].reshape(g.shape)), (case, rows, cols, ch, ksize, pad)
        _assert_canaries(dst)
        src.free()
        dst.free()


@pytest.mark.parametrize("rows,cols", [(1, 4), (2, 8), (7, 12), (33, 64), (61, 128), (40, 300)])
@pytest.mark.parametrize("ch", [1, 3])
@pytest.mark.parametrize("ksize", [3, 5, 7])
def test_filter2d_f32_stream_path(ctx, oracle, rng, rows, cols, ch, ksize):
    """shapes the streaming f32 kernel takes (cols*ch % 4 == 0): row-segment seams, edge threads, tiny images"""
    img = rand_img(rng, rows, cols, ch)
    k = (rng.standard_normal((ksize, ksize)) / ksize).astype(np.float32)
    n = 2
    src = device.DeviceBatch(ctx, n, rows, cols, ch, step=cols * ch + 4)
    dst = device.DeviceBatch(ctx, n, rows, cols, ch)
    frames = np.stack([img, img[::-1].copy()])
    src.upload(frames)
    device.filter2d(src, dst, k, delta=-3.5)
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.filter2d_f32(frames[i], k, -3.5))
    src.free()
    dst.free()


@pytest.mark.parametrize("rows,cols", [(1, 4), (7, 12), (61, 128), (40, 300)])
@pytest.mark.parametrize("ch", [1, 3])
@pytest.mark.parametrize("ksize,sigma", [(3, 0.7), (5, 1.3), (7, 1.9), (9, 2.2), (11, 3.0), (13, 3.0)])
def test_gaussian_sigma_stream_path(ctx, oracle, rng, rows, cols, ch, ksize, sigma):
    img = rand_img(rng, rows, cols, ch)
    src, dst = Mat.from_array(img), Mat(rows, cols, ch)
    imgproc.gaussian_blur(src, dst, ksize, sigma, ctx)
    assert np.array_equal(dst.to_array(), oracle.gaussian_blur(img, ksize, sigma))


@pytest.mark.parametrize("rows,cols", SHAPES)
def test_sobel(ctx, oracle, rng, rows, cols):
    img = rand_img(rng, rows, cols, 1)
    src = Mat.from_array(img, step=cols + 3)
    dx, dy = Mat(rows, cols, 1, _ffi.RCV_16S), Mat(rows, cols, 1, _ffi.RCV_16S, step=cols * 2 + 6)
    imgproc.sobel(src, dx, dy, ctx)
    wx, wy = oracle.sobel(img)
    assert np.array_equal(dx.to_array(), wx) and np.array_equal(dy.to_array(), wy)


@pytest.mark.parametrize("rows,cols", [(2, 8), (3, 16), (9, 496), (40, 504), (33, 1000), (70, 3840), (300, 64)])
def test_sobel_rows_path(ctx, oracle, rng, rows, cols):
    """shapes the register-sliding-window Sobel takes (cols % 8 == 0): strip seams at 496 px, segment seams, edges"""
    img = rand_img(rng, rows, cols, 1)
    img[:, 0] = 255
    img[:, -1] = 0
    n = 3
    src = device.DeviceBatch(ctx, n, rows, cols, 1, step=cols + 8)
    dx = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S, step=cols * 2 + 16)
    frames = np.stack([img, img[::-1].copy(), np.roll(img, 5, axis=1)])
    src.upload(frames)
    dx.memset(0x11)
    dy.memset(0x22)
    device.sobel(src, dx, dy)
    gx, gy = dx.download(), dy.download()
    for i in range(n):
        wx, wy = oracle.sobel(frames[i])
        assert np.array_equal(gx[i], wx) and np.array_equal(gy[i], wy)
    for b in (src, dx, dy):
        b.free()


@pytest.mark.parametrize("rows,cols", [(2, 8), (9, 16), (33, 504), (40, 512), (70, 520), (19, 1032), (300, 64), (5, 10), (7, 3)])
def test_sobel_of_bgr_source(ctx, oracle, rng, rows, cols):
    """f1: Sobel on a BGR image == Sobel(BGR2GRAY(.)) of the oracle; fused register kernel when cols % 8 == 0, gray through the
    workspace otherwise; batch of 2, padded steps, canaries around the i16 outputs"""
    n = 2
    src = device.DeviceBatch(ctx, n, rows, cols, 3, step=cols * 3 + 8 if (cols * 3) % 8 == 0 else None)
    dx = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_16S, pad=16)
    dy = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_16S, pad=16)
    frames = rng.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    src.upload(frames)
    device.sobel(src, dx, dy)
    gx, gy = dx.download(), dy.download()
    for i in range(n):
        wx, wy = oracle.sobel(oracle.bgr2gray(frames[i]))
        assert np.array_equal(gx[i], wx) and np.array_equal(gy[i], wy)
    _assert_canaries(dx)
    _assert_canaries(dy)
    for b in (src, dx, dy):
        b.free()


@pytest.mark.parametrize("rows,cols", [(4, 16), (9, 20), (33, 248), (40, 252), (31, 256), (37, 260), (21, 496), (18, 500), (26, 504), (70, 520),
                                       (64, 748), (19, 1032), (130, 1920), (300, 64), (5, 1000), (7, 3), (12, 18), (3, 64),
                                       (23, 764), (11, 768), (14, 772), (29, 944), (10, 956), (35, 960), (13, 964), (9, 972), (12, 976), (16, 1016),
                                       (27, 1024), (8, 1028), (15, 1916), (22, 1924), (9, 2884), (6, 3844)])
@pytest.mark.parametrize("ksize", [3, 5, 7])
def test_filter2d_sobel_fused(ctx, oracle, rows, cols, ksize):
    """f1: filter2D -> BGR2GRAY -> Sobel in one launch == sobel(bgr2gray(filter2d_i8(.))) of the oracle, bit
    for bit.  The SOB instantiation of the row-streaming MFMA kernel takes BGR images with a width that is a multiple of 4 (>= 16,
    >= 4 rows) on 4-byte aligned rows.  Round 6 layout: workgroups of four waves own 960 output pixels (60 regular windows + the two seam
    windows left and right of the group in wave 3's spare slots, wave-seam gray values through LDS) -- widths around the wave seams (256,
    512, 768), around the groups' ends (960, 1920, 2880, 3840) and around 944 / 976 (the seam windows' own ends) put the row's end into
    every place of a group; every other shape runs the two ordinary launches through the side buffer.  Batch of 3 (bands cross
    frames), padded steps, canaries around the i16 outputs; black/white frames drive the gradients to +-1020."""
    n = 3
    r = np.random.default_rng(rows * 1009 + cols * 7 + ksize + _SOAK_SEED)
    frames = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    frames[1] = (r.integers(0, 2, size=(rows, cols, 1)) * 255).astype(np.uint8)   # saturated edges
    k = r.integers(-8, 9, size=(ksize, ksize)).astype(np.int8)
    k[ksize // 2, ksize // 2] = 40
    shift = 6
    aligned = cols % 4 == 0
    src = device.DeviceBatch(ctx, n, rows, cols, 3, step=cols * 3 + (12 if aligned else 0))
    src.upload(frames)
    dx = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_16S, pad=16)
    dy = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_16S, pad=16)
    launched = _kernels_launched(ctx, lambda: device.filter2d_sobel(src, dx, dy, k, shift))
    fused = aligned and cols >= 16 and rows >= 4
    assert ("k_filter_rows_mfma<KS, kSobPP, 0, 0, 0, 1>" in launched) == fused, launched
    assert ("k_sobel_rows" in launched or "k_sobel" in launched) == (not fused), launched
    gx, gy = dx.download(), dy.download()
    for i in range(n):
        wx, wy = oracle.sobel(oracle.bgr2gray(oracle.filter2d_i8(frames[i], k, shift)))
        assert np.array_equal(gx[i].reshape(rows, cols), wx.reshape(rows, cols)), ("dx", i, np.argwhere(gx[i].reshape(rows, cols) != wx.reshape(rows, cols))[:4])
        assert np.array_equal(gy[i].reshape(rows, cols), wy.reshape(rows, cols)), ("dy", i, np.argwhere(gy[i].reshape(rows, cols) != wy.reshape(rows, cols))[:4])
    _assert_canaries(dx)
    _assert_canaries(dy)
    for b in (src, dx, dy):
        b.free()


def test_filter2d_sobel_fused_unequal_plane_layouts(ctx, oracle):
    """gradient planes with different row steps, or rows that are not 8-byte aligned, take the two-launch path: same bytes"""
    n, rows, cols = 2, 40, 512
    r = np.random.default_rng(77 + _SOAK_SEED)
    frames = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    k = r.integers(-6, 7, size=(5, 5)).astype(np.int8)
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    src.upload(frames)
    for stepx, stepy in ((cols * 2 + 16, cols * 2 + 32), (cols * 2 + 2, cols * 2 + 2)):
        dx = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S, step=stepx)
        dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S, step=stepy)
        launched = _kernels_launched(ctx, lambda: device.filter2d_sobel(src, dx, dy, k, 5))
        assert "k_filter_rows_mfma<KS, kSobPP, 0, 0, 0, 1>" not in launched and "k_sobel" in launched, launched
        gx, gy = dx.download(), dy.download()
        for i in range(n):
            wx, wy = oracle.sobel(oracle.bgr2gray(oracle.filter2d_i8(frames[i], k, 5)))
            assert np.array_equal(gx[i].reshape(rows, cols), wx.reshape(rows, cols)) and np.array_equal(gy[i].reshape(rows, cols), wy.reshape(rows, cols))
        dx.free()
        dy.free()
    src.free()


def test_filter2d_sobel_host_mat_and_arguments(ctx, oracle, rng):
    img = rand_img(rng, 31, 48, 3)
    k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], np.int8)
    dx, dy = Mat(31, 48, 1, _ffi.RCV_16S), Mat(31, 48, 1, _ffi.RCV_16S)
    imgproc.filter2d_sobel(Mat.from_array(img), dx, dy, k, 4, ctx)
    wx, wy = oracle.sobel(oracle.bgr2gray(oracle.filter2d_i8(img, k, 4)))
    assert np.array_equal(dx.to_array().reshape(31, 48), wx.reshape(31, 48)) and np.array_equal(dy.to_array().reshape(31, 48), wy.reshape(31, 48))
    with pytest.raises(Exception):   # one-channel source
        imgproc.filter2d_sobel(Mat.from_array(rand_img(rng, 8, 16, 1)), Mat(8, 16, 1, _ffi.RCV_16S), Mat(8, 16, 1, _ffi.RCV_16S), k, 4, ctx)
    with pytest.raises(Exception):   # shape mismatch
        imgproc.filter2d_sobel(Mat.from_array(img), Mat(31, 40, 1, _ffi.RCV_16S), dy, k, 4, ctx)
    with pytest.raises(Exception):   # even kernel size
        imgproc.filter2d_sobel(Mat.from_array(img), dx, dy, np.ones((2, 2), np.int8), 0, ctx)


def test_sobel_of_bgr_source_host_mat(ctx, oracle, rng):
    img = rand_img(rng, 31, 48, 3)
    dx, dy = Mat(31, 48, 1, _ffi.RCV_16S), Mat(31, 48, 1, _ffi.RCV_16S)
    imgproc.sobel(Mat.from_array(img), dx, dy, ctx)
    wx, wy = oracle.sobel(oracle.bgr2gray(img))
    assert np.array_equal(dx.to_array(), wx) and np.array_equal(dy.to_array(), wy)
    with pytest.raises(Exception):
        imgproc.sobel(Mat.from_array(rand_img(rng, 8, 8, 4)), Mat(8, 8, 1, _ffi.RCV_16S), Mat(8, 8, 1, _ffi.RCV_16S), ctx)


@pytest.mark.parametrize("src_shape,dst_shape", [((48, 64), (12, 16)), ((48, 64), (48, 64)), ((17, 33), (40, 71)), ((61, 127), (13, 9)),
                                                 ((1, 1), (5, 7)), ((4, 4), (1, 1)), ((128, 240), (32, 60)), ((30, 50), (31, 49))])
@pytest.mark.parametrize("ch", [1, 3, 4])
def test_resize(ctx, oracle, rng, src_shape, dst_shape, ch):
    img = rand_img(rng, *src_shape, ch)
    src, dst = Mat.from_array(img, step=src_shape[1] * ch + 5), Mat(*dst_shape, ch)
    imgproc.resize(src, dst, ctx)
    want = oracle.resize(img, *dst_shape)
    assert np.array_equal(dst.to_array(), want)  # f32 path with a fixed op order: bit-exact (north_star allows 1 ULP)
    if src_shape == (48, 64) and dst_shape == (12, 16):  # exact 4x == (a+b+c+d+2)>>2 of the centre 2x2
        a = img.astype(np.int32).reshape(12, 4, 16, 4, -1)
        box = (a[:, 1, :, 1] + a[:, 1, :, 2] + a[:, 2, :, 1] + a[:, 2, :, 2] + 2) >> 2
        assert np.array_equal(want.reshape(12, 16, -1), box)


@pytest.mark.parametrize("src_shape,dst_shape", [((48, 64), (48, 64)), ((720, 1280), (480, 852)), ((61, 127), (40, 96)), ((17, 33), (40, 72)),
                                                 ((30, 4), (7, 8)), ((5, 5), (9, 260)), ((270, 480), (181, 324)), ((9, 1000), (3, 12)),
                                                 ((100, 100), (1, 4)), ((3, 700), (5, 1400))])
def test_resize_bgr_fast_path(ctx, oracle, rng, src_shape, dst_shape):
    """general-scale BGR kernel (output width a multiple of 4): up- and down-scales, right-edge waves, tiny sources, batch of 2"""
    n = 2
    src = device.DeviceBatch(ctx, n, src_shape[0], src_shape[1], 3)
    dst = _canary_batch(ctx, n, dst_shape[0], dst_shape[1], 3, pad=8)
    frames = rng.integers(0, 256, size=(n,) + src_shape + (3,), dtype=np.uint8)
    src.upload(frames)
    device.resize(src, dst)
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.resize(frames[i], dst_shape[0], dst_shape[1]))
    _assert_canaries(dst)
    src.free()
    dst.free()


@pytest.mark.parametrize("scale", [2, 4])
@pytest.mark.parametrize("dshape", [(1, 4), (5, 8), (27, 240), (270, 480)])
def test_resize_box_fast_path(ctx, oracle, rng, scale, dshape):
    """exact 2x / 4x down-scale kernel == the general bilinear oracle, batch of 2"""
    dr, dc = dshape
    n = 2
    src = device.DeviceBatch(ctx, n, dr * scale, dc * scale, 3)
    dst = device.DeviceBatch(ctx, n, dr, dc, 3)
    frames = rng.integers(0, 256, size=(n, dr * scale, dc * scale, 3), dtype=np.uint8)
    src.upload(frames)
    device.resize(src, dst)
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.resize(frames[i], dr, dc))
    src.free()
    dst.free()


def test_warp_and_resize_8k_to_1080p_slab(ctx, oracle):
    """BASELINE configs[3] shapes on one frame: 8K rotate-7-degrees warp (rows of the result vs the oracle) and 8K -> 1080p"""
    rows, cols = 4320, 7680
    src = device.DeviceBatch(ctx, 1, rows, cols, 3)
    device.synth(src, 1, 0x5EED0004, 0)
    frame = src.download()[0]
    M = _rot(7.0, cols / 2, rows / 2, 13.25, -8.5)
    dst = device.DeviceBatch(ctx, 1, rows, cols, 3)
    device.warp_affine(src, dst, M)
    warped = oracle.warp_affine(frame, M, rows, cols)
    assert np.array_equal(dst.download()[0], warped)
    small = device.DeviceBatch(ctx, 1, 1080, 1920, 3)
    device.resize(src, small)
    assert np.array_equal(small.download()[0], oracle.resize(frame, 1080, 1920))
    # "next" row f1: the fused warp + 4x down-scale equals the two-step composition
    device.warp_affine_resize(src, small, M, rows, cols)
    assert np.array_equal(small.download()[0], oracle.resize(warped, 1080, 1920))
    for b in (src, dst, small):
        b.free()


def _rot(deg, cx, cy, tx, ty):
    t = np.deg2rad(deg)
    c, s = np.cos(t), np.sin(t)
    return np.array([c, -s, cx - c * cx + s * cy + tx, s, c, cy - s * cx - c * cy + ty], np.float32)


@pytest.mark.parametrize("shape", [(48, 64), (61, 127), (1, 1), (5, 3)])
@pytest.mark.parametrize("ch", [1, 3, 4])
@pytest.mark.parametrize("M", [np.array([1, 0, 0, 0, 1, 0], np.float32), np.array([1, 0, 0.5, 0, 1, 0.25], np.float32),
                               np.array([1, 0, -30.75, 0, 1, 1000], np.float32), np.array([0.5, 0.1, 3, -0.2, 1.7, -4], np.float32),
                               "rot7", np.array([1e9, 0, 0, 0, 1e-9, 0], np.float32)])
def test_warp_affine(ctx, oracle, rng, shape, ch, M):
    if isinstance(M, str):
        M = _rot(7.0, shape[1] / 2, shape[0] / 2, 13.25, -8.5)
    img = rand_img(rng, *shape, ch)
    src, dst = Mat.from_array(img), Mat(shape[0] + 3, shape[1] + 5, ch)
    imgproc.warp_affine(src, dst, M, ctx)
    assert np.array_equal(dst.to_array(), oracle.warp_affine(img, M, shape[0] + 3, shape[1] + 5))


@pytest.mark.parametrize("scale", [2, 4])
@pytest.mark.parametrize("dshape", [(24, 32), (17, 260), (3, 4)])
@pytest.mark.parametrize("M", ["ident", "shift", "rot7", "shrink", "far"])
def test_warp_affine_resize_fused(ctx, oracle, rng, scale, dshape, M):
    """fused warpAffine + exact down-scale == resize(warp_affine(.)) of the oracle, batch of 2 (interior, border and outside waves)"""
    dr, dc = dshape
    mr, mc = dr * scale, dc * scale
    sr, sc = mr + 5, mc + 9            # source a little larger than the intermediate image
    Ms = {"ident": np.array([1, 0, 4, 0, 1, 2], np.float32), "shift": np.array([1, 0, 0.5, 0, 1, 0.25], np.float32),
          "rot7": _rot(7.0, mc / 2, mr / 2, 3.25, -1.5), "shrink": np.array([0.5, 0.1, 3, -0.2, 0.7, 4], np.float32),
          "far": np.array([1, 0, 1e6, 0, 1, 0], np.float32)}[M]
    n = 2
    src = device.DeviceBatch(ctx, n, sr, sc, 3)
    dst = device.DeviceBatch(ctx, n, dr, dc, 3)
    frames = rng.integers(0, 256, size=(n, sr, sc, 3), dtype=np.uint8)
    src.upload(frames)
    device.warp_affine_resize(src, dst, Ms, mr, mc)
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.resize(oracle.warp_affine(frames[i], Ms, mr, mc), dr, dc))
    src.free()
    dst.free()


@pytest.mark.parametrize("n,fpg,kernel", [(4, 0, "quad"), (5, 0, "quad"), (9, 4, "quad"), (9, 8, "quad"), (17, 0, "quad"), (19, 16, "quad"), (6, 0, "single")])
@pytest.mark.parametrize("M", ["rot7", "rot-20", "shear", "shift", "flip", "big"])
@pytest.mark.parametrize("dshape", [(70, 200), (33, 131)])
def test_warp_affine_gray_four_frames_per_pass(ctx, oracle, rng, knob, n, fpg, kernel, M, dshape):
    """round 3: one-channel warpAffine with the SAME pixel of four consecutive frames in one LDS dword (k_warp_gray_lds4): frame
    counts that are not a multiple of 4, groups with a short last pass, ragged destination widths (131), tiles at the source border
    and outside (gather path), against the one-frame kernel's results (RCV_WARP_GRAY4=0) and the oracle"""
    if kernel == "single":
        knob("RCV_WARP_GRAY4", 0)
    if fpg:
        knob("RCV_WARP_FPG", fpg)
    dr, dc = dshape
    sr, sc = dr + 21, dc + 40
    Ms = {"rot7": _rot(7.0, dc / 2, dr / 2, 13.25, 9.5), "rot-20": _rot(-20.0, dc / 2, dr / 2, 20.0, 12.0),
          "shear": np.array([1, 0.25, 3.5, -0.125, 1, 18.25], np.float32), "shift": np.array([1, 0, 7.5, 0, 1, 3.25], np.float32),
          "flip": np.array([-1, 0, dc + 5.5, 0, -1, dr + 3.25], np.float32), "big": np.array([3, 0, 0, 0, 3, 0], np.float32)}[M]
    src = device.DeviceBatch(ctx, n, sr, sc, 1, step=(sc + 3) // 4 * 4 + 4)
    dst = _canary_batch(ctx, n, dr, dc, 1, pad=9)
    frames = rng.integers(0, 256, size=(n, sr, sc), dtype=np.uint8)
    src.upload(frames)
    L = _ffi.lib()
    L.rcv__debug_kernels_reset()
    device.warp_affine(src, dst, Ms)
    names = L.rcv__debug_kernels().decode()
    assert ("k_warp_gray_lds4" in names) == (kernel != "single" and M != "big"), names   # ("big": the patch of a tile does not fit -> no LDS plan)
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.warp_affine(frames[i], Ms, dr, dc)), (kernel, M, n, i)
    _assert_canaries(dst)
    src.free()
    dst.free()


@pytest.mark.parametrize("kernel,fpg", [("box", 0)])
@pytest.mark.parametrize("M", ["rot7", "rot-20", "shear", "ident", "flip", "grow", "big"])
def test_warp_affine_resize_fused_lds_tiles(ctx, oracle, rng, knob, kernel, fpg, M):
    """the fused warp -> 4x gather kernel (interior tiles, tiles at the source border and outside, ragged last tile row / column) produces
    the oracle's resize(warp_affine(.)) bit for bit.  (Round 3's LDS-staged variant of this launch was removed in round 4: slower.)"""
    if fpg:
        knob("RCV_WARP_FPG", fpg)
    dr, dc = 52, 100                  # 4 x 7 tiles, the last row / column of tiles ragged (dc % 4 == 0)
    mr, mc = 4 * dr, 4 * dc
    sr, sc = mr + 37, mc + 22
    Ms = {"rot7": _rot(7.0, mc / 2, mr / 2, 13.25, 9.5), "rot-20": _rot(-20.0, mc / 2, mr / 2, 20.0, 30.0),
          "shear": np.array([1, 0.25, 3.5, -0.125, 1, 60.25], np.float32), "ident": np.array([1, 0, 4, 0, 1, 2], np.float32),
          "flip": np.array([-1, 0, mc + 5.5, 0, -1, mr + 3.25], np.float32), "grow": np.array([0.5, 0, 40.3, 0, 0.5, 20.7], np.float32),
          "big": np.array([3, 0, 0, 0, 3, 0], np.float32)}[M]           # (patch of a tile too large to stage: gather tiles only)
    n = 5
    src = device.DeviceBatch(ctx, n, sr, sc, 3, step=sc * 3 + 1 + (-(sc * 3 + 1)) % 4)
    dst = _canary_batch(ctx, n, dr, dc, 3, pad=8)
    frames = rng.integers(0, 256, size=(n, sr, sc, 3), dtype=np.uint8)
    src.upload(frames)
    device.warp_affine_resize(src, dst, Ms, mr, mc)
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.resize(oracle.warp_affine(frames[i], Ms, mr, mc), dr, dc)), (kernel, M, i)
    _assert_canaries(dst)
    src.free()
    dst.free()


@pytest.mark.parametrize("plan", [(1, 1, 32, 0, 0), (1, 2, 32, 0, 0), (1, 3, 16, 1, 0), (1, 5, 64, 2, 0), (1, 8, 32, 2, 2), (1, 4, 32, 1, 3), (0, 0, 0, 1, 2), (0, 0, 0, 2, 3),
                                  (2, 1, 0, 0, 0), (2, 2, 0, 1, 0), (2, 3, 0, 2, 2), (2, 5, 0, 0, 0), (2, 8, 0, 1, 3)])
@pytest.mark.parametrize("scale", [2, 4])
@pytest.mark.parametrize("M", ["rot7", "rot-20", "shear", "ident", "flip", "far"])
def test_warp_affine_resize_measurement_variants(ctx, oracle, rng, plan, scale, M):
    """round 5: the kernels tools/ablate_warp_resize.py times through rcv__warp_resize_bench (librustcv_hip_bench.so) -- the frame-loop
    kernel (frames per wave 1 .. 8 on a 5-frame batch: every tail of the two-deep pipeline; wave tiles 16 / 32 / 64 wide) and the
    XCD-contiguous / synchronous-stripes tile orders of both kernels -- produce the oracle's resize(warp_affine(.)) bit for bit:
    interior waves, waves at the source border and outside, ragged last tile row / column, canaries around the destination"""
    variant, fpg, ww, order, strip = plan
    dr, dc = 52, 100
    mr, mc = scale * dr, scale * dc
    sr, sc = mr + 37, mc + 22
    Ms = {"rot7": _rot(7.0, mc / 2, mr / 2, 13.25, 9.5), "rot-20": _rot(-20.0, mc / 2, mr / 2, 20.0, 30.0),
          "shear": np.array([1, 0.25, 3.5, -0.125, 1, 60.25], np.float32), "ident": np.array([1, 0, 4, 0, 1, 2], np.float32),
          "flip": np.array([-1, 0, mc + 5.5, 0, -1, mr + 3.25], np.float32), "far": np.array([1, 0, 1e6, 0, 1, 0], np.float32)}[M]
    n = 5
    src = device.DeviceBatch(ctx, n, sr, sc, 3, step=sc * 3 + (-(sc * 3)) % 4 + 4)
    dst = _canary_batch(ctx, n, dr, dc, 3, pad=8)
    frames = rng.integers(0, 256, size=(n, sr, sc, 3), dtype=np.uint8)
    src.upload(frames)
    a, b = src.as_rcv(), dst.as_rcv()
    m = np.ascontiguousarray(Ms, dtype=np.float32)
    _ffi.check(_ffi.bench_lib().rcv__warp_resize_bench(ctx.handle, C.byref(a), C.byref(b), m.ctypes.data_as(C.POINTER(C.c_float)), scale, variant, fpg, ww,
                                                       order, strip, -1), "rcv__warp_resize_bench")
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.resize(oracle.warp_affine(frames[i], Ms, mr, mc), dr, dc)), (plan, M, i)
    _assert_canaries(dst)
    src.free()
    dst.free()


@pytest.mark.parametrize("sc", [1340, 1344])
def test_warp_affine_resize_staged_tight_capacity(ctx, oracle, rng, sc):
    """a source whose stated capacity ends with the last pixel of the last row (cap = (rows - 1) * step + cols * 3, the reference's
    Vec::len() of a packed Mat): the staged kernel's last chunk of that row runs past the capacity; the buffer range check is per
    dword, so every tap byte below it still arrives (rows of 4020 bytes: not a multiple of 16; 4032: the zero-filled path)"""
    dr, dc = 40, 328
    mr, mc = 4 * dr, 4 * dc
    sr = mr + 3
    n = 3
    M = np.array([1, 0, float(sc - mc) - 0.5, 0, 1, float(sr - mr) - 0.5], np.float32)     # the map ends on the source's last row / last columns
    step = sc * 3
    src = device.DeviceBatch(ctx, n, sr, sc, 3, step=step, frame_cap=(sr - 1) * step + sc * 3)
    frames = rng.integers(1, 256, size=(n, sr, sc, 3), dtype=np.uint8)
    src.upload(frames)
    dst = _canary_batch(ctx, n, dr, dc, 3, pad=8)
    a, b = src.as_rcv(), dst.as_rcv()
    _ffi.check(_ffi.bench_lib().rcv__warp_resize_bench(ctx.handle, C.byref(a), C.byref(b), M.ctypes.data_as(C.POINTER(C.c_float)), 4, 2, 3, 0, 0, 0, -1),
               "rcv__warp_resize_bench")
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.resize(oracle.warp_affine(frames[i], M, mr, mc), dr, dc)), (sc, i)
    _assert_canaries(dst)
    src.free()
    dst.free()


def test_warp_affine_resize_plan_check_is_cached_per_map(ctx, oracle, rng):
    """the host-side plan check of the staged kernel is cached in the context, keyed by matrix and geometry: alternating maps and a
    changed destination size on one context must each get their own verdict (kernel names) and the oracle's bytes"""
    dr, dc = 38, 200
    mr, mc = 4 * dr, 4 * dc
    sr, sc, n = mr + 33, mc + 27, 16
    fits, steep = _rot(7.0, mc / 2, mr / 2, 13.25, 9.5), _rot(-20.0, mc / 2, mr / 2, 20.0, 30.0)
    src = device.DeviceBatch(ctx, n, sr, sc, 3, step=sc * 3 + (-(sc * 3)) % 4)
    frames = rng.integers(0, 256, size=(n, sr, sc, 3), dtype=np.uint8)
    src.upload(frames)
    for M, kern, (r_, c_) in ((fits, "k_warp_resize_stage<4", (dr, dc)), (steep, "k_warp_resize_box<4", (dr, dc)), (fits, "k_warp_resize_stage<4", (dr, dc)),
                              (fits, "k_warp_resize_stage<4", (dr - 4, dc - 8)), (steep, "k_warp_resize_box<4", (dr - 4, dc - 8))):
        dst = _canary_batch(ctx, n, r_, c_, 3, pad=8)
        names = _kernels_launched(ctx, lambda: device.warp_affine_resize(src, dst, M, 4 * r_, 4 * c_))
        assert kern in names, (names, kern)
        got = dst.download()
        for i in (0, n - 1):
            assert np.array_equal(got[i], oracle.resize(oracle.warp_affine(frames[i], M, 4 * r_, 4 * c_), r_, c_)), (kern, i)
        _assert_canaries(dst)
        dst.free()
    src.free()


@pytest.mark.parametrize("fpg,order,strip", [(1, 0, 0), (3, 3, 2 + 256 * 4), (5, 1, 0), (2, 0, 131072)])
@pytest.mark.parametrize("M", ["left", "right", "top", "bottom", "corner", "rot7out", "far", "farneg", "nan", "inf", "flip", "zero", "graze-1", "grazecols", "big"])
def test_warp_affine_resize_staged_border_tiles(ctx, oracle, rng, fpg, order, strip, M):
    """round 5: tiles at the source border on the staged path (rows of a multiple of 16 bytes: `zfill`): chunks outside the image are not
    fetched and arrive as zeros (the constant border tap by tap), samples the specification zeroes without reading taps get zero weights
    on a pixel outside the image.  Maps that leave the source on every side and at a corner, tiles wholly outside, maps thousands of
    pixels away, NaN / inf / all-zero matrices, a mirrored map, coordinates grazing -1 and cols exactly, a magnifying map (footprint
    too large: fallback); strip + 131072: the same launch with border tiles on the gather path.  Oracle bytes, canaries."""
    dr, dc = 44, 264
    mr, mc = 4 * dr, 4 * dc
    sr, sc = 150, 1040                      # 1040 * 3 = 3120 = 16 * 195
    c7, s7 = np.cos(np.deg2rad(7.0)), np.sin(np.deg2rad(7.0))
    Ms = {"left": [1, 0, -300.25, 0, 1, 3.5], "right": [1, 0, 400.75, 0, 1, 2.25], "top": [1, 0, 5.5, 0, 1, -70.5], "bottom": [1, 0, 7.25, 0, 1, 60.75],
          "corner": [1, 0, -200.5, 0, 1, -90.25], "rot7out": [c7, -s7, -40.5, s7, c7, -30.25], "far": [1, 0, 1e6, 0, 1, 0], "farneg": [1, 0, -3e5, 0, 1, -2e5],
          "nan": [np.nan, 0, 0, 0, 1, 0], "inf": [1, np.inf, 3, 0, 1, 0], "flip": [-1, 0, mc + 5.5 - 30, 0, -1, mr + 3.25 - 20], "zero": [0, 0, 5.5, 0, 0, 7.25],
          "graze-1": [1, 0, -2.0, 0, 1, -2.0], "grazecols": [1, 0, float(sc - mc + 1), 0, 1, float(sr - mr + 1)], "big": [2.5, 0, 0, 0, 2.5, 0]}[M]
    Ms = np.array(Ms, np.float32)
    n = 5
    src = device.DeviceBatch(ctx, n, sr, sc, 3, step=sc * 3 + 16)
    dst = _canary_batch(ctx, n, dr, dc, 3, pad=8)
    frames = rng.integers(1, 256, size=(n, sr, sc, 3), dtype=np.uint8)     # (no zero pixels: a wrongly fetched tap shows)
    src.upload(frames)
    a, b = src.as_rcv(), dst.as_rcv()
    _ffi.check(_ffi.bench_lib().rcv__warp_resize_bench(ctx.handle, C.byref(a), C.byref(b), Ms.ctypes.data_as(C.POINTER(C.c_float)), 4, 2, fpg, 0,
                                                       order, strip, -1), "rcv__warp_resize_bench")
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.resize(oracle.warp_affine(frames[i], Ms, mr, mc), dr, dc)), (fpg, order, strip, M, i)
    _assert_canaries(dst)
    src.free()
    dst.free()


@pytest.mark.parametrize("scale", [4, 2])
@pytest.mark.parametrize("n", [8, 13, 23, 37])
@pytest.mark.parametrize("M", ["rot7", "rot-3", "shear", "rot-20"])
def test_warp_affine_resize_product_dispatch(ctx, oracle, rng, knob, n, M, scale):
    """round 5: rcv_warp_affine_resize_batch sends 4x and 2x launches of 8+ frames to k_warp_resize_stage when the map's tile footprints fit
    (frame groups of <= 12: one group, an uneven pair, two, four groups), everything else and RCV_WARP_LDS=0 to k_warp_resize_box;
    both produce the oracle's bytes"""
    dr, dc = 38, 200
    mr, mc = scale * dr, scale * dc
    sr, sc = mr + 33, mc + 27
    Ms = {"rot7": _rot(7.0, mc / 2, mr / 2, 13.25, 9.5), "rot-3": _rot(-3.0, mc / 2, mr / 2, 16.5, 21.25),
          "shear": np.array([1, 0.0625, 3.5, -0.03125, 1, 30.25], np.float32), "rot-20": _rot(-20.0, mc / 2, mr / 2, 20.0, 30.0)}[M]
    src = device.DeviceBatch(ctx, n, sr, sc, 3, step=sc * 3 + (-(sc * 3)) % 4)
    frames = rng.integers(0, 256, size=(n, sr, sc, 3), dtype=np.uint8)
    src.upload(frames)
    want = [oracle.resize(oracle.warp_affine(frames[i], Ms, mr, mc), dr, dc) for i in range(n)]
    # (at 4x a 20-degree rotation spans more source rows per tile than the staged kernel holds: the host's plan check keeps it on the gather
    #  kernel; the 2x footprint of the same map fits)
    for lds_knob, kern in ((None, f"k_warp_resize_stage<{scale}" if (M != "rot-20" or scale == 2) else f"k_warp_resize_box<{scale}"), (0, f"k_warp_resize_box<{scale}")):
        if lds_knob is not None:
            knob("RCV_WARP_LDS", lds_knob)
        dst = _canary_batch(ctx, n, dr, dc, 3, pad=8)
        names = _kernels_launched(ctx, lambda: device.warp_affine_resize(src, dst, Ms, mr, mc))
        assert kern in names, names
        got = dst.download()
        for i in range(n):
            assert np.array_equal(got[i], want[i]), (kern, n, M, i)
        _assert_canaries(dst)
        dst.free()
    # seven frames: the gather kernel whatever the knob says
    v = src.view(0, 7)
    dst = _canary_batch(ctx, 7, dr, dc, 3, pad=8)
    assert f"k_warp_resize_box<{scale}" in _kernels_launched(ctx, lambda: device.warp_affine_resize(v, dst, Ms, mr, mc))
    dst.free()
    src.free()


@pytest.mark.parametrize("fpg,order,strip", [(1, 0, 0), (2, 1, 0), (3, 0, 65536), (7, 2, 0), (4, 0, 0), (2, 3, 2 + 256 * 2), (5, 3, 4 + 256 * 3), (3, 3, 1 + 256 * 7)])
@pytest.mark.parametrize("scale", [2, 4])
@pytest.mark.parametrize("M", ["rot7", "rot-3", "rot12", "shear", "ident", "flip", "shrink", "grow"])
def test_warp_affine_resize_staged_row_pieces(ctx, oracle, rng, fpg, order, strip, scale, M):
    """round 5: k_warp_resize_stage (exact row pieces of the tile's footprint fetched global -> LDS, unaligned 8-byte tap reads, two buffers,
    frames walked per tile) on frames large enough that most tiles are interior: odd frame counts against every group size (the tails
    of the two-buffer loop), rows that are 4- but not 16-byte aligned (chunks straddle lines), maps whose footprint fits (staged) and
    does not fit (steeper rotation, magnification: the workgroup falls back), ragged last tile row / column, canaries; tile orders: raster,
    XCD-contiguous, synchronous stripes and blocks of bw x bh tiles (strip = bw + 256 bh; blocks that overhang the tile grid); strip + 65536:
    row pieces packed without the odd-slot rule"""
    dr, dc = 70, 328
    mr, mc = scale * dr, scale * dc
    sr, sc = mr + 45, mc + 31
    Ms = {"rot7": _rot(7.0, mc / 2, mr / 2, 13.25, 9.5), "rot-3": _rot(-3.0, mc / 2, mr / 2, 16.5, 21.25), "rot12": _rot(12.0, mc / 2, mr / 2, 14.0, 20.0),
          "shear": np.array([1, 0.0625, 3.5, -0.03125, 1, 30.25], np.float32), "ident": np.array([1, 0, 4, 0, 1, 2], np.float32),
          "flip": np.array([-1, 0, mc + 5.5, 0, -1, mr + 3.25], np.float32), "shrink": np.array([0.5, 0, 40.3, 0, 0.5, 20.7], np.float32),
          "grow": np.array([1.25, 0, 2.5, 0, 1.0625, 1.75], np.float32)}[M]
    n = 5
    src = device.DeviceBatch(ctx, n, sr, sc, 3, step=sc * 3 + (-(sc * 3)) % 4 + 4)
    dst = _canary_batch(ctx, n, dr, dc, 3, pad=8)
    frames = rng.integers(0, 256, size=(n, sr, sc, 3), dtype=np.uint8)
    src.upload(frames)
    a, b = src.as_rcv(), dst.as_rcv()
    m = np.ascontiguousarray(Ms, dtype=np.float32)
    _ffi.check(_ffi.bench_lib().rcv__warp_resize_bench(ctx.handle, C.byref(a), C.byref(b), m.ctypes.data_as(C.POINTER(C.c_float)), scale, 2, fpg, 0,
                                                       order, strip, -1), "rcv__warp_resize_bench")
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.resize(oracle.warp_affine(frames[i], Ms, mr, mc), dr, dc)), (fpg, order, strip, M, i)
    _assert_canaries(dst)
    src.free()
    dst.free()


@pytest.mark.parametrize("M", [[np.nan, 0, 0, 0, 1, 0], [1, 0, np.nan, 0, 1, 0], [np.inf, 0, 0, 0, 1, 0], [1, np.inf, 3, 0, 1, 0],
                               [1, 0, 0, -np.inf, 1, 0], [0, 0, 5.5, 0, 0, 7.25], [1e-30, 0, 1, 0, 1e-30, 2], [-1, 0, 63, 0, -1, 31],
                               [1, 0, -0.999, 0, 1, -0.999], [1, 0, 0.999, 0, 1, 0.999], [1.0000001, 0, -1, 0, 1, -1]])
def test_warp_affine_bgr_kernel_degenerate_matrices(ctx, oracle, rng, M):
    """BGR fast kernels (width a multiple of 4: interior / outside / border paths and the fused down-scale) on NaN, inf,
    singular, mirrored and edge-grazing maps: same bytes as the oracle"""
    M = np.array(M, np.float32)
    rows, cols, n = 32, 64, 2
    frames = rng.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    src.upload(frames)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.warp_affine(src, dst, M)
    got = dst.download()
    small = device.DeviceBatch(ctx, n, rows // 2, cols // 2, 3)
    device.warp_affine_resize(src, small, M, rows, cols)
    got2 = small.download()
    for i in range(n):
        want = oracle.warp_affine(frames[i], M, rows, cols)
        assert np.array_equal(got[i], want)
        assert np.array_equal(got2[i], oracle.resize(want, rows // 2, cols // 2))
    for b in (src, dst, small):
        b.free()


@pytest.mark.parametrize("rows,cols", [(40, 20), (33, 36), (64, 1080), (50, 1084), (37, 1088), (29, 1092), (45, 252), (31, 260), (70, 264), (19, 268), (23, 772),
                                       (41, 1548), (16, 16), (90, 28),
                                       (40, 17), (33, 37), (64, 1079), (50, 1081), (37, 1082), (29, 1083), (45, 253), (31, 254), (70, 255), (19, 257), (23, 258),
                                       (41, 259), (90, 29), (12, 18), (25, 19), (30, 21), (30, 22), (30, 23), (44, 769), (44, 770), (44, 771), (27, 511), (27, 513),
                                       (64, 1919), (20, 3839)])
def test_row_streaming_kernel_on_any_width(ctx, oracle, knob, rows, cols):
    """the row-streaming MFMA kernel on BGR widths with every residue mod 16: multiples of 4 on 4-byte aligned rows (a packed
    1080-pixel-wide frame; the right-border chunk holds 0 / 4 / 8 / 12 valid pixels) and ANY other width on byte-aligned rows (the
    SRC = 3 instantiation: aligned dwords + v_alignbyte on the load side, byte-granular border repair with 0..15 valid pixels --
    for 14 and 15 the mirrored pixels spill into the next chunk --, unaligned stores, the row's last 1-3 pixels byte by byte);
    partial last windows and strips; filter2D 3 / 5 / 7 and the integer GaussianBlur (two weight tables at 7) against the
    oracle; frames start at any byte"""
    knob("RCV_F7_ROWS", 1)
    n = 3
    fa = 4 if cols % 4 == 0 else 1   # frame alignment
    for ks in (3, 5, 7):
        rng = np.random.default_rng(1000 * rows + 7 * cols + ks + _SOAK_SEED)
        frames = rng.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
        k = rng.integers(-20, 21, size=(ks, ks)).astype(np.int8)
        src = device.DeviceBatch(ctx, n, rows, cols, 3, frame_stride=rows * cols * 3 + fa * int(rng.integers(0, 3)))
        dst = device.DeviceBatch(ctx, n, rows, cols, 3, frame_stride=rows * cols * 3 + fa * int(rng.integers(1, 3)))
        src.upload(frames)
        dst.memset(0xEE)
        launched = _kernels_launched(ctx, lambda: device.filter2d(src, dst, k, shift=6))
        assert "k_filter_rows_mfma<" in launched, launched
        raw = dst.download_bytes()[: n * dst.frame_stride].reshape(n, dst.frame_stride)
        for i in range(n):
            assert np.array_equal(raw[i, : rows * cols * 3].reshape(rows, cols, 3), oracle.filter2d_i8(frames[i], k, 6)), (ks, i)
        assert (raw[:, rows * cols * 3:] == 0xEE).all(), "gap between frames overwritten"
        launched = _kernels_launched(ctx, lambda: device.gaussian_blur(src, dst, ks, 0.0))
        assert "k_filter_rows_mfma<" in launched, launched
        got = dst.download()
        for i in range(n):
            assert np.array_equal(got[i], oracle.gaussian_blur(frames[i], ks, 0.0)), ("gauss", ks, i)
        src.free()
        dst.free()


@pytest.mark.parametrize("rows,cols", [(37, 41), (64, 333), (129, 1919), (30, 1021), (200, 47)])
@pytest.mark.parametrize("ch", [1, 3])
def test_stream_filters_on_byte_aligned_rows(ctx, oracle, rows, cols, ch):
    """odd widths of packed images (step = cols * channels: rows start at any byte, frames too): GaussianBlur sigma > 0, f32
    filter2D run the streaming kernel's unaligned instantiation (window fetched as aligned dwords + one v_alignbyte shift per
    row, unaligned dword stores, the row's last 1-3 bytes by the edge launch), the BGR integer filters the row-streaming MFMA
    kernel's any-width instantiation, the one-channel integer filters the dot4 kernel's -- bit for bit against the oracle, padding between frames
    untouched, and never the per-sample kernels"""
    n = 3
    r = np.random.default_rng(rows * 4099 + cols * 7 + ch + _SOAK_SEED)
    frames = r.integers(0, 256, size=(n, rows, cols, ch), dtype=np.uint8)
    fs = rows * cols * ch + int(r.integers(1, 4)) * 2 + 1          # odd frame stride: every frame starts at another alignment
    src = device.DeviceBatch(ctx, n, rows, cols, ch, frame_stride=fs)
    src.upload(frames)
    img = lambda i: frames[i] if ch == 3 else frames[i, :, :, 0]   # noqa: E731

    def check(tag, fn, ref, mfma=False):
        dst = device.DeviceBatch(ctx, n, rows, cols, ch, frame_stride=fs + 2)
        dst.memset(0xCD)
        launched = _kernels_launched(ctx, lambda: fn(dst))
        # BGR integer filters: the row-streaming MFMA kernel's any-width instantiation; everything else: the streaming VALU kernel
        want = ("k_filter_rows_mfma<" if ch == 3 and cols >= 16 else "k_filter_gray_dot4<" if ch == 1 and cols >= 12 else "k_filter_f32_stream<") if mfma else "k_filter_f32_stream<"
        assert want in launched and "generic" not in launched, (tag, launched)
        raw = dst.download_bytes()[: n * dst.frame_stride].reshape(n, dst.frame_stride)
        for i in range(n):
            got = raw[i, : rows * cols * ch].reshape(rows, cols, ch)
            want = ref(img(i))
            assert np.array_equal(got if ch == 3 else got[:, :, 0], want), (tag, i)
        assert (raw[:, rows * cols * ch:] == 0xCD).all(), (tag, "gap between frames overwritten")
        dst.free()

    for ks in (3, 5, 7):
        k = r.integers(-9, 10, size=(ks, ks)).astype(np.int8)
        check(f"filter2D i8 {ks}", lambda d, k=k: device.filter2d(src, d, k, shift=5), lambda a, k=k: oracle.filter2d_i8(a, k, 5), mfma=True)
        check(f"Gaussian int {ks}", lambda d, ks=ks: device.gaussian_blur(src, d, ks, 0.0), lambda a, ks=ks: oracle.gaussian_blur(a, ks, 0.0), mfma=True)
        check(f"Gaussian sigma {ks}", lambda d, ks=ks: device.gaussian_blur(src, d, ks, 1.3), lambda a, ks=ks: oracle.gaussian_blur(a, ks, 1.3))
        kf = (r.standard_normal((ks, ks)) / ks).astype(np.float32)
        check(f"filter2D f32 {ks}", lambda d, kf=kf: device.filter2d(src, d, kf, delta=1.5), lambda a, kf=kf: oracle.filter2d_f32(a, kf, 1.5))
    src.free()


@pytest.mark.parametrize("shape", [(3, 37, 64, 3), (2, 61, 136, 1), (1, 9, 40, 3), (2, 5, 96, 3), (1, 2, 64, 1), (2, 130, 200, 3), (1, 1, 48, 3)])
@pytest.mark.parametrize("ks", [3, 5, 7, 9, 11])
def test_gaussian_sigma_row_pair_kernel(ctx, oracle, rng, knob, shape, ks):
    """round 5: k_gauss_f32_pairs (the separable pass on row pairs: {row r, row r + 1} halves, output row pairs, RAD + 1 pairs in flight)
    against the oracle on images of a few rows (fewer than the taps), odd heights (a half-used last output pair), row segments of
    every parity, 1 / 3 channels, every tap count; the integer Gaussian (sigma = 0) through the same kernel where its taps allow;
    RCV_GAUSS_ROWS=0 sends the same call to the one-row kernel"""
    n, rows, cols, ch = shape
    frames = rng.integers(0, 256, size=(n, rows, cols, ch) if ch > 1 else (n, rows, cols), dtype=np.uint8)
    src = device.DeviceBatch(ctx, n, rows, cols, ch)
    src.upload(frames)
    for sigma in (1.3, 0.8 if ks <= 7 else 2.6):
        want = [oracle.gaussian_blur(frames[i], ks, sigma) for i in range(n)]
        for val, kern in ((1, "k_gauss_f32_pairs<"), (0, "k_filter_f32_stream<")):
            knob("RCV_GAUSS_ROWS", val)
            dst = _canary_batch(ctx, n, rows, cols, ch, pad=8)
            names = _kernels_launched(ctx, lambda: device.gaussian_blur(src, dst, ks, sigma))
            assert kern in names, (names, shape, ks)
            got = dst.download()
            for i in range(n):
                assert np.array_equal(got[i], want[i]), (kern, shape, ks, sigma, i)
            _assert_canaries(dst)
            dst.free()
    src.free()


def _kernels_launched(ctx, fn):
    L = _ffi.lib()
    L.rcv__debug_kernels_reset()
    fn()
    ctx.sync()
    return L.rcv__debug_kernels().decode()


@pytest.mark.parametrize("ch", [3, 1])
@pytest.mark.parametrize("case", range(14))
def test_warp_affine_bgr_lds_staged_kernel(ctx, oracle, knob, case, ch):
    """the LDS-staged warp, BGR and one-channel (source patch of a 64 x 32 tile copied to LDS, taps from LDS) on maps whose patch fits --
    rotations of any angle, scales, shears, translations -- over images with interior tiles, border tiles and tiles wholly
    outside the source; batches split into frame groups of 1 / 2 / 3 / 8 frames (incomplete last group); same bytes as the
    oracle and as the gather kernel (RCV_WARP_LDS=0)"""
    rng = np.random.default_rng(0xC0FFEE + case)
    sr, sc = int(rng.integers(150, 420)), int(rng.integers(200, 640))
    dr, dc = int(rng.integers(100, 400)), (4 * int(rng.integers(40, 150)) if case % 2 == 0 else int(rng.integers(160, 600)))   # odd cases: any width
    n = int(rng.integers(1, 6))
    kind = case % 7
    if kind == 0:
        M = _rot(float(rng.uniform(-180, 180)), sc / 2, sr / 2, float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20)))
    elif kind == 1:
        M = _rot(float(rng.choice([7.0, 45.0, 90.0, -90.0, 180.0, 0.1])), sc / 2, sr / 2, 13.25, -8.5)
    elif kind == 2:
        M = np.array([1, 0, float(rng.uniform(-5, 5)), 0, 1, float(rng.uniform(-5, 5))], np.float32)
    elif kind == 3:
        sx, sy = float(rng.uniform(0.6, 1.5)), float(rng.uniform(0.6, 1.5))
        M = np.array([sx, 0, float(rng.uniform(0, 9)), 0, sy, float(rng.uniform(0, 9))], np.float32)
    elif kind == 4:
        M = np.array([1, float(rng.uniform(-0.5, 0.5)), 3.5, float(rng.uniform(-0.5, 0.5)), 1, 2.25], np.float32)
    elif kind == 5:   # mirror + rotation
        M = _rot(float(rng.uniform(-30, 30)), sc / 2, sr / 2, 0.0, 0.0) * np.array([-1, 1, 1, -1, 1, 1], np.float32) + np.array([0, 0, sc - 3, 0, 0, 0], np.float32)
    else:             # general affine near the identity
        M = (np.array([1, 0, 0, 0, 1, 0]) + rng.uniform(-0.25, 0.25, 6) * np.array([1, 1, 40, 1, 1, 40])).astype(np.float32)
    M = np.asarray(M, np.float32)
    frames = rng.integers(0, 256, size=(n, sr, sc, ch), dtype=np.uint8)
    src = device.DeviceBatch(ctx, n, sr, sc, ch, step=sc * ch + (int(rng.integers(0, 4)) if ch == 1 and case % 3 == 0 else 0))   # (one channel: rows of any alignment)
    dst = device.DeviceBatch(ctx, n, dr, dc, ch, step=dc * ch + (4 if case % 2 == 0 else 1) * int(rng.integers(0, 3)))
    src.upload(frames)
    want = [oracle.warp_affine(frames[i] if ch == 3 else frames[i, :, :, 0], M, dr, dc).reshape(dr, dc, ch) for i in range(n)]
    for fpg in (0, 1, 2, 3, 8):
        if fpg:
            knob("RCV_WARP_FPG", fpg)
        dst.memset(0xAB)
        launched = _kernels_launched(ctx, lambda: device.warp_affine(src, dst, M))
        # (one channel, >= 4 frames, 4-byte aligned source rows: the four-frames-per-pass kernel)
        assert ("k_warp_affine_lds<%d" % ch) in launched or (ch == 1 and n >= 4 and "k_warp_gray_lds4" in launched), (launched, M)
        got = dst.download()
        for i in range(n):
            assert np.array_equal(got[i].reshape(dr, dc, ch), want[i]), (case, fpg, i, M.tolist(), (sr, sc, dr, dc))
    knob("RCV_WARP_LDS", 0)
    dst.memset(0xCD)
    launched = _kernels_launched(ctx, lambda: device.warp_affine(src, dst, M))
    assert launched.split(";") == ["k_warp_affine_bgr" if ch == 3 else "k_warp_affine_gray"], launched
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i].reshape(dr, dc, ch), want[i]), (case, "gather", i)
    src.free()
    dst.free()


def test_warp_affine_bgr_large_patch_takes_the_gather_kernel(ctx, oracle, rng):
    """a map whose source patch per tile exceeds the LDS budget (4x down-scale) stays on the gather kernel"""
    M = np.array([4.0, 0.3, 1.5, -0.2, 4.0, 2.5], np.float32)
    frames = rng.integers(0, 256, size=(2, 300, 400, 3), dtype=np.uint8)
    src, dst = device.DeviceBatch(ctx, 2, 300, 400, 3), device.DeviceBatch(ctx, 2, 64, 96, 3)
    src.upload(frames)
    launched = _kernels_launched(ctx, lambda: device.warp_affine(src, dst, M))
    assert launched.split(";") == ["k_warp_affine_bgr"], launched
    got = dst.download()
    for i in range(2):
        assert np.array_equal(got[i], oracle.warp_affine(frames[i], M, 64, 96))
    src.free()
    dst.free()


@pytest.mark.parametrize("ch,mid,dshape", [(3, (50, 70), (20, 28)), (1, (48, 64), (24, 32)), (3, (48, 66), (24, 33)), (4, (40, 40), (10, 10))])
def test_warp_affine_resize_unfused_shapes(ctx, oracle, rng, ch, mid, dshape):
    """shapes the fused kernel does not take (non-integer factor, 1/4 channels, width not a multiple of 4) run warp then resize"""
    img = rand_img(rng, 45, 61, ch)
    M = np.array([0.98, -0.1, 2.5, 0.1, 0.98, -1.25], np.float32)
    src, dst = Mat.from_array(img), Mat(dshape[0], dshape[1], ch)
    imgproc.warp_affine_resize(src, dst, M, mid[0], mid[1], ctx)
    assert np.array_equal(dst.to_array(), oracle.resize(oracle.warp_affine(img, M, mid[0], mid[1]), dshape[0], dshape[1]))


def test_warp_affine_resize_host_fused(ctx, oracle, rng):
    img = rand_img(rng, 130, 200, 3)
    M = _rot(-11.0, 100, 65, 0.5, 0.75)
    src, dst = Mat.from_array(img), Mat(32, 48, 3)
    imgproc.warp_affine_resize(src, dst, M, 128, 192, ctx)
    assert np.array_equal(dst.to_array(), oracle.resize(oracle.warp_affine(img, M, 128, 192), 32, 48))
    with pytest.raises(Exception):
        imgproc.warp_affine_resize(src, dst, M, 0, 192, ctx)


@pytest.mark.parametrize("rows,cols", SHAPES)
@pytest.mark.parametrize("block", [1, 2, 3, 5])
def test_corner_harris(ctx, oracle, rng, rows, cols, block):
    img = rand_img(rng, rows, cols, 1)
    src, dst = Mat.from_array(img), Mat(rows, cols, 1, _ffi.RCV_32F)
    imgproc.corner_harris(src, dst, block, 0.04, ctx)
    want = oracle.corner_harris(img, block, 0.04)
    assert np.array_equal(dst.to_array().view(np.uint32), want.view(np.uint32))  # bit-exact f32


@pytest.mark.parametrize("block", [1, 2, 3, 4, 5, 7])
@pytest.mark.parametrize("cols", [512, 509])
def test_corner_harris_saturated_gradients(ctx, oracle, block, cols):
    """Black/white images drive |Ix|, |Iy| to 1020 and the window sums to their maximum (block^2 * 1020^2): blocks <= 4 hold them
    in f32 (< 2^24: exact), larger ones in i32.  Stripes of period 4 (|Ix| = 1020 on half the columns), the same transposed,
    a checkerboard of 2x2 cells, random binary pixels -- response bit for bit against the oracle, aligned (one-launch kernel)
    and ragged width (planes + window kernel)."""
    rows = 64
    r = np.random.default_rng(block * 131 + cols + _SOAK_SEED)
    yy, xx = np.mgrid[0:rows, 0:cols]
    imgs = [((xx // 2) % 2) * 255, ((yy // 2) % 2) * 255, (((xx // 2) + (yy // 2)) % 2) * 255, r.integers(0, 2, size=(rows, cols)) * 255,
            ((xx // 3) % 2) * 255, np.where(xx < cols // 2, 0, 255)]
    gray = np.stack(imgs).astype(np.uint8)[..., None]
    n = gray.shape[0]
    src = device.DeviceBatch(ctx, n, rows, cols, 1)
    src.upload(gray)
    resp = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_32F, pad=32 if cols % 8 == 0 else 4)
    device.corner_harris(src, resp, block, 0.04)
    got = resp.download()
    for i in range(n):
        assert np.array_equal(got[i].view(np.uint32), oracle.corner_harris(gray[i, :, :, 0], block, 0.04).view(np.uint32)), ("cornerHarris", i)
    _assert_canaries(resp)
    mask = _canary_batch(ctx, n, rows, cols, 1, pad=8)
    device.harris_pipeline(src, mask, None, block, 0.04, 1e-3)
    gm = mask.download()
    for i in range(n):
        wr = oracle.corner_harris(gray[i, :, :, 0], block, 0.04)
        assert np.array_equal(gm[i].reshape(rows, cols), oracle.nms3x3(wr, 1e-3).reshape(rows, cols)), ("mask", i)
    _assert_canaries(mask)
    for b in (src, resp, mask):
        b.free()


@pytest.mark.parametrize("rows,cols", [(7, 16), (9, 496), (40, 504), (33, 1000), (130, 3840), (300, 64), (61, 120), (8, 24), (45, 497), (64, 1919), (23, 9), (31, 503)])
@pytest.mark.parametrize("block", [1, 3, 4, 5, 6, 7])
def test_corner_harris_any_block_streaming_kernels(ctx, oracle, rows, cols, block):
    """block sizes other than 2 on any width >= 8: aligned shapes run the one-launch kernel (gray conversion and Sobel in front of
    the window, gradients of the mirrored columns from the neighbouring lanes), ragged ones (byte-aligned rows) streaming Sobel
    into i16 planes with mirrored margins + the register-window response kernel (vertical running sums in a register ring, horizontal
    sliding sums with DPP halos, row segments) -- cornerHarris from gray and the pipeline from BGR (mask, mask + response;
    the 3x3 NMS inside the same kernel) -- bit for bit against the oracle; batch of 3, padded steps.  Images with fewer than
    block + 2 rows stay on the per-sample kernels."""
    n = 3
    r = np.random.default_rng(rows * 7919 + cols * 31 + block + _SOAK_SEED)
    gray = r.integers(0, 256, size=(n, rows, cols, 1), dtype=np.uint8)
    gray[:, rows // 3: rows // 3 + 3, cols // 4: cols // 4 + 5] = 255      # a few real corners
    src = device.DeviceBatch(ctx, n, rows, cols, 1, step=(cols + 15) // 16 * 16 + 16) if cols % 8 == 0 else device.DeviceBatch(ctx, n, rows, cols, 1)
    src.upload(gray)
    resp = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_32F, pad=32 if cols % 8 == 0 else 4)
    launched = _kernels_launched(ctx, lambda: device.corner_harris(src, resp, block, 0.04))
    # aligned shapes: ONE launch (gray -> Sobel -> window sums -> response [-> NMS]); ragged ones: Sobel into planes + the window kernel
    fused = cols % 8 == 0 and cols >= 16 and rows >= max(block + 4, 8)
    streaming = not fused and rows >= block + 2
    assert ("k_harris_blocks_fused" in launched) == fused, launched
    assert ("k_harris_resp_rows" in launched) == streaming and ("k_sobel_rows" in launched) == streaming, launched
    got = resp.download()
    for i in range(n):
        assert np.array_equal(got[i].view(np.uint32), oracle.corner_harris(gray[i, :, :, 0], block, 0.04).view(np.uint32)), ("cornerHarris", i)
    _assert_canaries(resp)
    bgr = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    bgr[:, rows // 2: rows // 2 + 4, cols // 3: cols // 3 + 6] = (255, 255, 255)
    sb = device.DeviceBatch(ctx, n, rows, cols, 3, step=cols * 3 + 8)
    sb.upload(bgr)
    thr = 1e-5
    for want_resp in (False, True):
        mask = _canary_batch(ctx, n, rows, cols, 1, pad=8)
        resp2 = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F) if want_resp else None
        launched = _kernels_launched(ctx, lambda: device.harris_pipeline(sb, mask, resp2, block, 0.04, thr))
        assert ("k_harris_blocks_fused" in launched) == fused and ("k_harris_resp_rows" in launched) == streaming, launched
        assert ("k_nms3x3" in launched) == (not streaming and not fused), launched   # the NMS runs inside the window kernels
        gm = mask.download()
        for i in range(n):
            wm, wr = oracle.harris_pipeline(bgr[i], block, 0.04, thr, True)
            assert np.array_equal(gm[i], wm), ("pipeline mask", want_resp, i)
            if want_resp:
                assert np.array_equal(resp2.download()[i].view(np.uint32), wr.view(np.uint32)), ("pipeline response", i)
        _assert_canaries(mask)
        mask.free()
        if resp2 is not None:
            resp2.free()
    for b in (src, resp, sb):
        b.free()


def test_harris_general_block_kernel_at_block2(ctx, oracle, knob):
    """RCV_HARRIS_GENERAL=1 routes blockSize 2 through the general-block one-launch kernel as well: same mask and response as the
    dedicated kernel and the oracle (the dedicated kernel stays the default: 0.56 against 0.71 ms on 64 4K frames)"""
    rows, cols, n = 150, 1000, 2
    r = np.random.default_rng(4321 + _SOAK_SEED)
    bgr = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    bgr[:, 40:47, 300:309] = 255
    sb = device.DeviceBatch(ctx, n, rows, cols, 3)
    sb.upload(bgr)
    mask, resp = device.DeviceBatch(ctx, n, rows, cols, 1), device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F)
    knob("RCV_HARRIS_GENERAL", 1)
    launched = _kernels_launched(ctx, lambda: device.harris_pipeline(sb, mask, resp, 2, 0.04, 1e-5))
    assert "k_harris_blocks_fused" in launched, launched
    gm, gr = mask.download(), resp.download()
    for i in range(n):
        wm, wr = oracle.harris_pipeline(bgr[i], 2, 0.04, 1e-5, True)
        assert np.array_equal(gm[i], wm) and np.array_equal(gr[i].view(np.uint32), wr.view(np.uint32)), i
    for b in (sb, mask, resp):
        b.free()


@pytest.mark.parametrize("thr", [float("nan"), float("inf"), float("-inf"), 0.0, -1e-3, 3.0e-6])
def test_harris_pipeline_block3_threshold_edge_values(ctx, oracle, thr):
    """the NMS inside the general-block response kernel at threshold edge values (NaN keeps nothing, -inf keeps every local
    maximum) against the oracle"""
    rows, cols, n = 37, 72, 2
    r = np.random.default_rng(77 + _SOAK_SEED)
    bgr = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    bgr[:, 10:14, 20:26] = 255
    sb, mask = device.DeviceBatch(ctx, n, rows, cols, 3), device.DeviceBatch(ctx, n, rows, cols, 1)
    sb.upload(bgr)
    launched = _kernels_launched(ctx, lambda: device.harris_pipeline(sb, mask, None, 3, 0.04, thr))
    assert "k_harris_blocks_fused" in launched and "k_nms3x3" not in launched, launched
    gm = mask.download()
    for i in range(n):
        assert np.array_equal(gm[i], oracle.harris_pipeline(bgr[i], 3, 0.04, thr)), (thr, i)
    # the same through the two-launch path (a source whose rows are only byte-aligned)
    sb2 = device.DeviceBatch(ctx, n, rows, cols, 3, step=cols * 3 + 1)
    sb2.upload(bgr)
    launched = _kernels_launched(ctx, lambda: device.harris_pipeline(sb2, mask, None, 3, 0.04, thr))
    assert "k_harris_resp_rows" in launched and "k_nms3x3" not in launched, launched
    gm = mask.download()
    for i in range(n):
        assert np.array_equal(gm[i], oracle.harris_pipeline(bgr[i], 3, 0.04, thr)), (thr, "planes", i)
    sb.free()
    sb2.free()
    mask.free()


@pytest.mark.parametrize("want_resp", [False, True])
def test_harris_pipeline_frame_of_more_than_4gb(ctx, oracle, want_resp):
    """round 6: the aligned instantiations of k_harris_fused address a frame's rows with 32-bit scalar offsets (buffer loads / stores); a frame whose
    rows x step reaches 4 GB goes to the 64-bit-pointer instantiation instead.  A 70 x 64 BGR image with a row step of 64 MiB (4.6 GB of mostly
    padding): the rows beyond 4 GB would wrap onto the first ones if it did not."""
    rows, cols, step = 70, 64, 64 << 20
    img = oracle.synth_frame(rows, cols, 3, 1, 0x5EED0005, 5)
    src = device.DeviceBatch(ctx, 1, rows, cols, 3, step=step)
    L = _ffi.lib()
    for r in range(rows):
        row = np.ascontiguousarray(img[r].reshape(-1))
        _ffi.check(L.rcv_upload(ctx.handle, C.c_void_p(src.ptr.value + r * step), row.ctypes.data, row.size), "rcv_upload")
    mask = device.DeviceBatch(ctx, 1, rows, cols, 1)
    resp = device.DeviceBatch(ctx, 1, rows, cols, 1, _ffi.RCV_32F) if want_resp else None
    mask.memset(7)
    launched = _kernels_launched(ctx, lambda: device.harris_pipeline(src, mask, resp, 2, 0.04, 1e-4))
    assert "k_harris_fused<" in launched and "true, true>" in launched, launched   # (the RAG = true instantiation: 64-bit row pointers)
    want = oracle.harris_pipeline(img, 2, 0.04, 1e-4, want_resp=want_resp)
    wm = want[0] if want_resp else want
    assert np.array_equal(mask.download()[0].reshape(rows, cols), wm.reshape(rows, cols))
    if want_resp:
        assert np.array_equal(resp.download()[0].reshape(rows, cols).view(np.uint32), want[1].reshape(rows, cols).view(np.uint32))
    for b in (src, mask) + ((resp,) if want_resp else ()):
        b.free()


@pytest.mark.parametrize("rows,cols", [(4, 8), (9, 496), (40, 504), (33, 1000), (130, 3840), (300, 64), (61, 120), (7, 12)])
def test_corner_harris_and_pipeline_from_gray(ctx, oracle, rows, cols):
    """a one-channel source: cornerHarris (response only) and the pipeline (mask, mask + response) on the fused register-window
    kernel where the shape allows (block 2, cols % 8 == 0), the generic kernels elsewhere; batch of 3, padded steps"""
    n = 3
    r = np.random.default_rng(rows * 7919 + cols + _SOAK_SEED)
    frames = r.integers(0, 256, size=(n, rows, cols, 1), dtype=np.uint8)
    frames[:, rows // 3: rows // 3 + 3, cols // 4: cols // 4 + 5] = 255      # a few real corners
    src = device.DeviceBatch(ctx, n, rows, cols, 1, step=(cols + 15) // 16 * 16 + 16)
    src.upload(frames)
    resp = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F)
    device.corner_harris(src, resp, 2, 0.04)
    got = resp.download()
    want = [oracle.corner_harris(frames[i, :, :, 0], 2, 0.04) for i in range(n)]
    for i in range(n):
        assert np.array_equal(got[i].view(np.uint32), want[i].view(np.uint32)), ("cornerHarris", i)
    thr = 1e-5
    for want_resp in (False, True):
        mask = _canary_batch(ctx, n, rows, cols, 1, pad=8)
        resp2 = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F) if want_resp else None
        device.harris_pipeline(src, mask, resp2, 2, 0.04, thr)
        gm = mask.download()
        for i in range(n):
            assert np.array_equal(gm[i], oracle.nms3x3(want[i], thr)), ("pipeline mask", want_resp, i)
        if want_resp:
            gr = resp2.download()
            for i in range(n):
                assert np.array_equal(gr[i].view(np.uint32), want[i].view(np.uint32))
            resp2.free()
        _assert_canaries(mask)
        mask.free()
    src.free()
    resp.free()


@pytest.mark.parametrize("rows,cols", [(1, 4), (2, 8), (5, 256), (70, 260), (130, 1024), (65, 1028), (33, 2052)])
def test_nms3x3_streaming_kernel(ctx, oracle, rows, cols):
    """the row-streaming NMS kernel (16-byte aligned f32 rows, cols % 4 == 0): wave and block seams, row-segment seams,
    plateaus (ties are kept), +-inf, NaN (a NaN neighbour or centre is never kept), thresholds incl. NaN; batch of 2, padded steps"""
    r = np.random.default_rng(rows * 131 + cols + _SOAK_SEED)
    n = 2
    resp = r.standard_normal((n, rows, cols)).astype(np.float32)
    resp[r.random((n, rows, cols)) < 0.2] = 0.5
    resp[r.random((n, rows, cols)) < 0.01] = np.nan
    resp[r.random((n, rows, cols)) < 0.01] = np.inf
    resp[r.random((n, rows, cols)) < 0.01] = -np.inf
    src = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F, step=cols * 4 + 32)
    src.upload(resp[..., None])
    for thr in (0.0, 0.5, -np.inf, np.nan):
        mask = _canary_batch(ctx, n, rows, cols, 1, pad=12)
        device.nms3x3(src, mask, thr)
        got = mask.download()
        for i in range(n):
            assert np.array_equal(got[i], oracle.nms3x3(resp[i], thr)), (thr, i)
        _assert_canaries(mask)
        mask.free()
    src.free()


@pytest.mark.parametrize("rows,cols", SHAPES)
def test_nms3x3(ctx, oracle, rng, rows, cols):
    resp = rng.standard_normal((rows, cols)).astype(np.float32)
    resp[rng.random((rows, cols)) < 0.2] = 0.5  # plateaus: ties must be kept (>=)
    src, dst = Mat.from_array(resp), Mat(rows, cols, 1)
    imgproc.nms3x3(src, dst, 0.1, ctx)
    assert np.array_equal(dst.to_array(), oracle.nms3x3(resp, 0.1))


@pytest.mark.parametrize("rows,cols", SHAPES + [(200, 300)])
@pytest.mark.parametrize("block", [2, 3])
@pytest.mark.parametrize("want_resp", [False, True])
def test_harris_pipeline(ctx, oracle, rows, cols, block, want_resp):
    img = oracle.synth_frame(rows, cols, 3, 1, 0x5EED0005, 3)
    src, mask = Mat.from_array(img), Mat(rows, cols, 1)
    resp = Mat(rows, cols, 1, _ffi.RCV_32F) if want_resp else None
    thr = 1e-4
    imgproc.harris_pipeline(src, mask, resp, block, 0.04, thr, ctx)
    if want_resp:
        wm, wr = oracle.harris_pipeline(img, block, 0.04, thr, True)
        assert np.array_equal(resp.to_array().view(np.uint32), wr.view(np.uint32))
    else:
        wm = oracle.harris_pipeline(img, block, 0.04, thr)
    assert np.array_equal(mask.to_array(), wm)


@pytest.mark.parametrize("rows,cols", [(4, 8), (5, 16), (9, 496), (40, 504), (33, 1000), (130, 3840), (300, 64)])
@pytest.mark.parametrize("want_resp", [False, True])
def test_harris_fused_path(ctx, oracle, rows, cols, want_resp):
    """shapes the fused register-window kernel takes (block 2, cols % 8 == 0, rows >= 4), batch of 3, padded steps"""
    n = 3
    src = device.DeviceBatch(ctx, n, rows, cols, 3, step=cols * 3 + 8)
    mask = device.DeviceBatch(ctx, n, rows, cols, 1, step=cols + 8)
    resp = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F) if want_resp else None
    device.synth(src, 1, 0x5EED0005, 11)
    mask.memset(0x33)
    thr = 1e-4
    device.harris_pipeline(src, mask, resp, 2, 0.04, thr)
    frames, got = src.download(), mask.download()
    gr = resp.download() if want_resp else None
    for i in range(n):
        wm, wr = oracle.harris_pipeline(frames[i], 2, 0.04, thr, True)
        if want_resp:
            assert np.array_equal(gr[i].view(np.uint32), wr.view(np.uint32))
        assert np.array_equal(got[i], wm)
    assert got.any() or rows * cols < 4096  # the scene has real corners
    for b in (src, mask) + ((resp,) if want_resp else ()):
        b.free()


def test_harris_fused_threshold_edge_values(ctx, oracle):
    """`rc > thr` with thresholds at the edges of the float line: +-0 (flat regions have a response of exactly 0 and must not
    pass), a denormal, the exact value of a local maximum present in the image (strict inequality), +-inf and NaN"""
    rows, cols = 64, 256
    img = np.zeros((rows, cols, 3), np.uint8)
    img[:, :] = 40                                   # flat background: response exactly 0
    rng = np.random.default_rng(77 + _SOAK_SEED)
    for _ in range(12):                              # bright rectangles: corners and edges
        y, x = int(rng.integers(2, rows - 12)), int(rng.integers(2, cols - 20))
        img[y:y + int(rng.integers(3, 10)), x:x + int(rng.integers(3, 18))] = rng.integers(80, 256, 3)
    _, resp = oracle.harris_pipeline(img, 2, 0.04, 0.0, True)
    peak = float(resp.max())
    assert peak > 0 and (resp == 0).any()
    thrs = [0.0, -0.0, 1e-45, -1e-45, peak, float(np.nextafter(np.float32(peak), np.float32(0))), -np.inf, np.inf, np.nan, 1e-9]
    src = device.DeviceBatch(ctx, 1, rows, cols, 3)
    mask = device.DeviceBatch(ctx, 1, rows, cols, 1)
    src.upload(img[None])
    for thr in thrs:
        device.harris_pipeline(src, mask, None, 2, 0.04, thr)
        want = oracle.harris_pipeline(img, 2, 0.04, thr)
        assert np.array_equal(mask.download()[0], want), thr
    want_peak = oracle.harris_pipeline(img, 2, 0.04, peak)
    assert want_peak[resp == peak].sum() == 0        # strict: the maximum itself does not exceed a threshold equal to it
    assert oracle.harris_pipeline(img, 2, 0.04, 0.0)[resp == 0].sum() == 0 and oracle.harris_pipeline(img, 2, 0.04, 0.0).any()
    src.free()
    mask.free()


def test_baseline_configs_at_full_size(ctx, oracle):
    """one or two frames of every BASELINE.json configuration that has no full-size test of its own, bit for bit against the
    oracle: config 1 (640x480 YUYV -> BGR + rectangle), config 2 (1080p 5x5 Gaussian), config 3's Sobel on 4K gray,
    config 5 (4K Harris pipeline with the response).  Full sizes exercise every strip/segment seam and the partial last strip."""
    # config 1
    y = device.DeviceBatch(ctx, 2, 480, 640, 2)
    b = device.DeviceBatch(ctx, 2, 480, 640, 3)
    device.synth(y, 2, 0x5EED0001, 0)
    device.cvt_color(y, b, _ffi.RCV_YUYV2BGR)
    device.rectangle(b, imgproc.Rect(200, 150, 240, 240), imgproc.Scalar(0, 255, 0), 2)
    fy, fb = y.download(), b.download()
    for i in range(2):
        want = np.zeros(480 * 640 * 3, np.uint8)
        assert oracle.yuyv_to_bgr(fy[i].reshape(-1), want, 640, 480)
        oracle.rectangle(want, 480, 640, 640 * 3, 200, 150, 240, 240, 0, 255, 0, 2)
        assert np.array_equal(fb[i].reshape(-1), want)
    y.free(); b.free()
    # config 2
    s = device.DeviceBatch(ctx, 1, 1080, 1920, 3)
    d = device.DeviceBatch(ctx, 1, 1080, 1920, 3)
    device.synth(s, 1, 0x5EED0002, 0)
    device.gaussian_blur(s, d, 5, 0.0)
    assert np.array_equal(d.download()[0], oracle.gaussian_blur(s.download()[0], 5, 0.0))
    s.free(); d.free()
    # config 3 (second half) and config 5 on one 4K frame
    s = device.DeviceBatch(ctx, 1, 2160, 3840, 3)
    g = device.DeviceBatch(ctx, 1, 2160, 3840, 1)
    dx = device.DeviceBatch(ctx, 1, 2160, 3840, 1, _ffi.RCV_16S)
    dy = device.DeviceBatch(ctx, 1, 2160, 3840, 1, _ffi.RCV_16S)
    m = device.DeviceBatch(ctx, 1, 2160, 3840, 1)
    r = device.DeviceBatch(ctx, 1, 2160, 3840, 1, _ffi.RCV_32F)
    device.synth(s, 1, 0x5EED0005, 0)
    device.cvt_color(s, g, _ffi.RCV_BGR2GRAY)
    device.sobel(g, dx, dy)
    device.harris_pipeline(s, m, r, 2, 0.04, 1e-4)
    frame = s.download()[0]
    gray = oracle.bgr2gray(frame)
    assert np.array_equal(g.download()[0], gray)
    wx, wy = oracle.sobel(gray)
    assert np.array_equal(dx.download()[0], wx) and np.array_equal(dy.download()[0], wy)
    wm, wr = oracle.harris_pipeline(frame, 2, 0.04, 1e-4, True)
    assert np.array_equal(r.download()[0].view(np.uint32), wr.view(np.uint32))
    assert np.array_equal(m.download()[0], wm) and wm.any()
    # the same frame at blockSize 3 and 5 (the general-block one-launch kernel: all strips, segment seams, both image edges)
    for block in (3, 5):
        device.harris_pipeline(s, m, r, block, 0.04, 1e-4)
        wm, wr = oracle.harris_pipeline(frame, block, 0.04, 1e-4, True)
        assert np.array_equal(r.download()[0].view(np.uint32), wr.view(np.uint32)), block
        assert np.array_equal(m.download()[0], wm) and wm.any(), block
    for x in (s, g, dx, dy, m, r):
        x.free()


def test_baseline_batch_geometries(ctx, oracle):
    """the launch geometries the benchmark and the driver time: the whole batches BASELINE.json's configs 3, 4 and 5 name per
    GPU (64 x 4K, 32 x 8K, 64 x 4K), first / middle / last frame of each result bit for bit against the oracle.  (One-frame and
    few-frame launches of the same shapes run different band / segment partitions.)"""
    k7 = (np.arange(49, dtype=np.int8).reshape(7, 7) * 5 % 23 - 11).astype(np.int8)
    # config 3: 64 x 4K BGR, 7x7 integer filter2D; gray + Sobel of the same batch
    n, rows, cols = 64, 2160, 3840
    s, d = device.DeviceBatch(ctx, n, rows, cols, 3), device.DeviceBatch(ctx, n, rows, cols, 3)
    g = device.DeviceBatch(ctx, n, rows, cols, 1)
    dx, dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    m = device.DeviceBatch(ctx, n, rows, cols, 1)
    device.synth(s, 1, 0x5EED0003, 0)
    device.filter2d(s, d, k7, shift=6)
    device.cvt_color(s, g, _ffi.RCV_BGR2GRAY)
    device.sobel(g, dx, dy)
    device.harris_pipeline(s, m, None, 2, 0.04, 1e-4)          # config 5
    fx, fy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    device.filter2d_sobel(s, fx, fy, k7, 6)                    # config 3 as one launch: filter2D -> gray -> Sobel
    for i in (0, n // 2 - 1, n - 1):
        frame = s.download_frame(i)
        assert np.array_equal(frame, oracle.synth_frame(rows, cols, 3, 1, 0x5EED0003, i))
        filtered = oracle.filter2d_i8(frame, k7, 6)
        assert np.array_equal(d.download_frame(i), filtered), ("filter2D", i)
        wfx, wfy = oracle.sobel(oracle.bgr2gray(filtered))
        assert np.array_equal(fx.download_frame(i), wfx) and np.array_equal(fy.download_frame(i), wfy), ("filter2D -> Sobel", i)
        gray = oracle.bgr2gray(frame)
        assert np.array_equal(g.download_frame(i), gray), ("gray", i)
        wx, wy = oracle.sobel(gray)
        assert np.array_equal(dx.download_frame(i), wx) and np.array_equal(dy.download_frame(i), wy), ("sobel", i)
        wm = oracle.harris_pipeline(frame, 2, 0.04, 1e-4)
        assert np.array_equal(m.download_frame(i), wm) and wm.any(), ("harris", i)
    for x in (s, d, g, dx, dy, m, fx, fy):
        x.free()
    # config 4: 32 x 8K BGR, warpAffine (rotate 7 degrees) and resize to 1080p, and the fused form
    n, rows, cols = 32, 4320, 7680
    s, w = device.DeviceBatch(ctx, n, rows, cols, 3), device.DeviceBatch(ctx, n, rows, cols, 3)
    small, fused = device.DeviceBatch(ctx, n, 1080, 1920, 3), device.DeviceBatch(ctx, n, 1080, 1920, 3)
    device.synth(s, 1, 0x5EED0004, 0)
    M = _rot(7.0, cols / 2, rows / 2, 13.25, -8.5)
    device.warp_affine(s, w, M)
    device.resize(w, small)
    device.warp_affine_resize(s, fused, M, rows, cols)
    for i in (0, n // 2, n - 1):
        warped = oracle.warp_affine(s.download_frame(i), M, rows, cols)
        assert np.array_equal(w.download_frame(i), warped), ("warp", i)
        want = oracle.resize(warped, 1080, 1920)
        assert np.array_equal(small.download_frame(i), want), ("resize", i)
        assert np.array_equal(fused.download_frame(i), want), ("fused", i)
    for x in (s, w, small, fused):
        x.free()


@pytest.mark.parametrize("rows,cols", [(4, 8), (9, 496), (40, 504), (33, 1000), (130, 3840), (21, 10), (7, 6)])
@pytest.mark.parametrize("want_resp", [False, True])
def test_harris_pipeline_from_yuyv(ctx, oracle, rows, cols, want_resp):
    """config 5 with a YUYV source: == harris_pipeline(yuyv_to_bgr(.)) of the oracle (fused kernel when cols % 8 == 0, the two-stage
    HIP path otherwise), batch of 2, padded steps"""
    n = 2
    src = device.DeviceBatch(ctx, n, rows, cols, 2, step=cols * 2 + 8)
    mask = device.DeviceBatch(ctx, n, rows, cols, 1, step=cols + 8)
    resp = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_32F) if want_resp else None
    device.synth(src, 2, 0x5EED0005, 3)
    mask.memset(0x33)
    thr = 1e-4
    device.harris_pipeline(src, mask, resp, 2, 0.04, thr)
    frames, got = src.download(), mask.download()
    gr = resp.download() if want_resp else None
    for i in range(n):
        bgr = np.zeros(rows * cols * 3, np.uint8)
        oracle.yuv422_to_bgr_strided(frames[i].reshape(-1), cols * 2, rows, cols, False, bgr)
        wm, wr = oracle.harris_pipeline(bgr.reshape(rows, cols, 3), 2, 0.04, thr, True)
        if want_resp:
            assert np.array_equal(gr[i].view(np.uint32), wr.view(np.uint32))
        assert np.array_equal(got[i], wm)
    for b in (src, mask) + ((resp,) if want_resp else ()):
        b.free()


# ---- synthetic frames + device-resident batches -------------------------------------------------------

@pytest.mark.parametrize("family,ch", [(0, 1), (0, 3), (0, 4), (1, 3), (1, 1)])
def test_synth_matches_oracle(ctx, oracle, family, ch):
    rows, cols, n = 70, 333, 3
    b = device.DeviceBatch(ctx, n, rows, cols, ch, step=cols * ch + 9)
    b.memset(0)
    device.synth(b, family, 0x5EED0003, 5)
    got = b.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.synth_frame(rows, cols, ch, family, 0x5EED0003, 5 + i))
    b.free()


def test_synth_yuyv_matches_oracle(ctx, oracle):
    rows, cols, n = 33, 64, 2
    b = device.DeviceBatch(ctx, n, rows, cols, 2)
    device.synth(b, _ffi.RCV_SYNTH_YUYV, 0x5EED0001, 0)
    got = b.download().reshape(n, rows, cols * 2)
    for i in range(n):
        assert np.array_equal(got[i], oracle.synth_yuyv(rows, cols, 0x5EED0001, i))
    b.free()


def test_batch_equals_single_frames(ctx, oracle):
    """Batch entry points: frame i of the batch == the single-frame result (no cross-frame leakage)."""
    rows, cols, n = 96, 160, 5
    k = oracle.bench_kernel7()
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 1, 0x5EED0003, 0)
    device.filter2d(src, dst, k, shift=6)
    got = dst.download()
    frames = src.download()
    for i in range(n):
        assert np.array_equal(frames[i], oracle.synth_frame(rows, cols, 3, 1, 0x5EED0003, i))
        assert np.array_equal(got[i], oracle.filter2d_i8(frames[i], k, 6))
    src.free()
    dst.free()


def test_contexts_driven_from_concurrent_threads(oracle):
    """a context is single-threaded, but different contexts are independent and may be driven from different
    threads at the same time (the one-host-thread-per-GPU model of 8(e)).  Four threads, each with its own context on GPU 0,
    run a mix of ops (different kernels per thread: per-context weight tables, constant areas, workspaces, occupancy caches)
    and every result must equal the oracle."""
    import threading
    import rustcv_amd as rcv
    rows, cols = 96, 256
    errors = []

    def worker(tid):
        try:
            r = np.random.default_rng(900 + tid + _SOAK_SEED)
            c = rcv.Context(0)
            for it in range(6 * _SOAK):
                img = r.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
                src = Mat.from_array(img)
                which = (it + tid) % 4
                if which == 0:
                    ks = int(r.choice([3, 5, 7]))
                    k = r.integers(-20, 21, size=(ks, ks)).astype(np.int8)
                    dst = Mat(rows, cols, 3)
                    imgproc.filter2d(src, dst, k, 5, ctx=c)
                    ok = np.array_equal(dst.to_array(), oracle.filter2d_i8(img, k, 5))
                elif which == 1:
                    mask = Mat(rows, cols, 1)
                    imgproc.harris_pipeline(src, mask, None, 2, 0.04, 1e-7, c)
                    ok = np.array_equal(mask.to_array(), oracle.harris_pipeline(img, 2, 0.04, 1e-7))
                elif which == 2:
                    dst = Mat(rows, cols, 3)
                    imgproc.gaussian_blur(src, dst, 5, 1.2, c)
                    ok = np.array_equal(dst.to_array(), oracle.gaussian_blur(img, 5, 1.2))
                else:
                    M = np.array([0.98, -0.1, 3.5, 0.1, 0.98, -2.25], np.float32)
                    dst = Mat(rows, cols, 3)
                    imgproc.warp_affine(src, dst, M, c)
                    ok = np.array_equal(dst.to_array(), oracle.warp_affine(img, M, rows, cols))
                if not ok:
                    errors.append((tid, it, which))
            c.close()
        except Exception as e:   # noqa: BLE001 -- reported through the list, the thread must not die silently
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("world", [2, 3, 8])
def test_frame_sharded_contexts_equal_one_batch(ctx, oracle, world):
    """rank i of G owns frames [floor(iN/G), floor((i+1)N/G)) on its own context and stream; the concatenation of
    the shards is byte-identical to the single-context result.  (All contexts sit on GPU 0 here: the box has one GPU.)"""
    import rustcv_amd as rcv
    from rustcv_amd import shard
    rows, cols, n = 80, 256, 11
    k = oracle.bench_kernel7()
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 1, 0x5EED0003, 0)
    device.filter2d(src, dst, k, shift=6)
    whole = dst.download()
    parts = []
    for r in range(world):
        f0, f1 = shard.frame_range(n, r, world)
        if f1 == f0:
            continue
        c = rcv.Context(0)
        s = device.DeviceBatch(c, f1 - f0, rows, cols, 3)
        d = device.DeviceBatch(c, f1 - f0, rows, cols, 3)
        device.synth(s, 1, 0x5EED0003, f0)          # frame_base = first global frame of the shard
        device.filter2d(s, d, k, shift=6)
        parts.append(d.download())
        s.free(); d.free(); c.close()
    assert np.array_equal(np.concatenate(parts, axis=0), whole)
    src.free()
    dst.free()


def test_errors_do_not_cross_the_abi(ctx):
    L = _ffi.lib()
    m = Mat(4, 4, 3)
    a, b = m._as_rcv(), Mat(4, 5, 3)._as_rcv()
    k = (C.c_int8 * 9)(*([1] * 9))
    assert L.rcv_filter2d_i8(ctx.handle, C.byref(a), C.byref(b), k, 3, 0) == _ffi.RCV_ERR_ARG      # shape mismatch
    assert L.rcv_filter2d_i8(ctx.handle, C.byref(a), C.byref(a), k, 3, 0) == _ffi.RCV_ERR_ARG      # in-place
    assert L.rcv_filter2d_i8(ctx.handle, C.byref(a), C.byref(b), k, 4, 0) == _ffi.RCV_ERR_ARG      # even ksize
    short = Mat(4, 4, 3)
    short.data = short.data[:-1]
    s = short._as_rcv()
    c = Mat(4, 4, 3)._as_rcv()
    assert L.rcv_filter2d_i8(ctx.handle, C.byref(s), C.byref(c), k, 3, 0) == _ffi.RCV_ERR_SIZE
    assert L.rcv_cvt_color(ctx.handle, 99, C.byref(a), C.byref(c)) == _ffi.RCV_ERR_ARG
    assert L.rcv_filter2d_i8(None, C.byref(a), C.byref(c), k, 3, 0) == _ffi.RCV_ERR_ARG


@pytest.mark.parametrize("rows,cols", [(0, 0), (0, 7), (5, 0)])
def test_empty_images_are_successful_noops(ctx, rows, cols):
    """the reference's Mat::empty() / zero-sized Mats flow through every op without touching memory and without an error"""
    L = _ffi.lib()
    def mat(ch, depth=_ffi.RCV_8U, r=rows, c=cols):
        return Mat(r, c, ch, depth)
    k8 = np.ones((3, 3), np.int8)
    kf = np.ones((3, 3), np.float32) / 9
    M = np.array([1, 0, 0, 0, 1, 0], np.float32)
    imgproc.gaussian_blur(mat(3), mat(3), 5, 0.0, ctx)
    imgproc.gaussian_blur(mat(3), mat(3), 5, 1.1, ctx)
    imgproc.filter2d(mat(3), mat(3), k8, 0, ctx=ctx)
    imgproc.filter2d(mat(1), mat(1), kf, ctx=ctx)
    imgproc.sobel(mat(1), mat(1, _ffi.RCV_16S), mat(1, _ffi.RCV_16S), ctx)
    imgproc.cvt_color(mat(3), mat(1), _ffi.RCV_BGR2GRAY, ctx)
    imgproc.harris_pipeline(mat(3), mat(1), None, 2, 0.04, 0.0, ctx)
    imgproc.corner_harris(mat(1), mat(1, _ffi.RCV_32F), 2, 0.04, ctx)
    imgproc.warp_affine(Mat(4, 4, 3), mat(3), M, ctx)            # empty destination
    imgproc.resize(Mat(4, 4, 3), mat(3), ctx)
    imgproc.rectangle(mat(3), Rect(0, 0, 3, 3), Scalar(1, 2, 3), 1, ctx)
    imgproc.blend_glyphs(mat(3), [(0, 0, np.ones((2, 2), np.float32))], Scalar(1, 2, 3), ctx)
    # device batches with zero frames
    for n in (0,):
        a = device.DeviceBatch(ctx, n, 8, 16, 3)
        b = device.DeviceBatch(ctx, n, 8, 16, 3)
        device.filter2d(a, b, k8, shift=0)
        device.filter2d(a, b, kf)
        device.gaussian_blur(a, b, 7, 0.0)
        device.warp_affine(a, b, M)
        device.resize(a, b)
        device.blend_glyphs(a, [(0, 0, np.ones((2, 2), np.float32))], Scalar(1, 2, 3))
        a.free()
        b.free()
    ctx.sync()


# ---- out-of-bounds canaries: padding between rows and between frames must survive every batch op -------------------

def _canary_batch(ctx, n, rows, cols, ch, depth=_ffi.RCV_8U, pad=32):
    esz = {_ffi.RCV_8U: 1, _ffi.RCV_16S: 2, _ffi.RCV_32F: 4}[depth]
    step = cols * ch * esz + pad
    b = device.DeviceBatch(ctx, n, rows, cols, ch, depth, step=step, frame_stride=rows * step + 512)
    b.memset(0xCD)
    return b


def _assert_canaries(b):
    raw = b.download_bytes()[: b.n * b.frame_stride].reshape(b.n, b.frame_stride)
    esz = {_ffi.RCV_8U: 1, _ffi.RCV_16S: 2, _ffi.RCV_32F: 4}[b.depth]
    rowb = b.cols * b.channels * esz
    body = raw[:, : b.rows * b.step].reshape(b.n, b.rows, b.step)
    assert (body[:, :, rowb:] == 0xCD).all(), "row padding overwritten"
    assert (raw[:, b.rows * b.step:] == 0xCD).all(), "inter-frame gap overwritten"


def test_warp_lds_kernel_tight_last_frame(ctx, oracle):
    """round-2 advisor: a batch whose LAST frame is allocated only up to (rows - 1) * step + cols * 3 (padded rows, the padding of the last
    row not allocated -- what rcv_view guarantees): the LDS-staged warp kernel clamps its staging loads inside that extent"""
    L = _ffi.lib()
    n, rows, cols = 2, 200, 320
    step = cols * 3 + 64
    fs = rows * step
    nbytes = (n - 1) * fs + (rows - 1) * step + cols * 3
    r = np.random.default_rng(606 + _SOAK_SEED)
    frames = r.integers(0, 256, size=(n, rows, cols, 3), dtype=np.uint8)
    raw = np.zeros(nbytes, np.uint8)
    for i in range(n):
        for y in range(rows):
            o = i * fs + y * step
            raw[o: o + cols * 3] = frames[i, y].reshape(-1)
    p = C.c_void_p()
    _ffi.check(L.rcv_malloc(ctx.handle, nbytes, C.byref(p)), "rcv_malloc")
    _ffi.check(L.rcv_upload(ctx.handle, p, raw.ctypes.data, nbytes), "rcv_upload")
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    sb = dst.as_rcv()
    sb.frame0.data, sb.frame0.cap, sb.frame0.step, sb.frame_stride = p, (rows - 1) * step + cols * 3, step, fs
    bd = dst.as_rcv()
    M = np.array([0.9925, -0.1219, 20.0, 0.1219, 0.9925, -12.0], np.float32)
    launched = _kernels_launched(ctx, lambda: _ffi.check(L.rcv_warp_affine_batch(ctx.handle, C.byref(sb), C.byref(bd), M.ctypes.data_as(C.POINTER(C.c_float))), "warp"))
    assert "k_warp_affine_lds<3" in launched, launched
    got = dst.download()
    for i in range(n):
        assert np.array_equal(got[i], oracle.warp_affine(frames[i], M, rows, cols))
    L.rcv_free(ctx.handle, p)
    dst.free()


def _ulp_distance(a, b):
    """distance in units in the last place between two f32 arrays (NaN == NaN; +0 == -0)"""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)      # sign-magnitude -> a monotonic integer line
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    d = np.abs(ia - ib)
    d[both_nan] = 0
    d[np.isnan(a) ^ np.isnan(b)] = 2 ** 40
    return d


F32_GEOM = [(53, 71, 60, 80, 1), (37, 48, 37, 48, 1), (64, 96, 16, 24, 1), (16, 24, 61, 97, 3), (5, 7, 9, 3, 4), (1, 1, 4, 5, 1), (270, 480, 1080, 1920, 1)]


@pytest.mark.parametrize("srows,scols,drows,dcols,ch", F32_GEOM)
def test_resize_f32(ctx, oracle, rng, srows, scols, drows, dcols, ch):
    """SURVEY.md 8-A / north_star: RCV_32F bilinear paths within 1 ULP of the CPU oracle.  The evaluation order is fixed on both
    sides (explicit fmaf, no contraction), so the expectation is bit-exact; the test asserts the 1-ULP bar and reports which."""
    n = 2
    frames = (rng.standard_normal((n, srows, scols, ch)) * rng.choice([1e-3, 1.0, 3e4])).astype(np.float32)
    src = device.DeviceBatch(ctx, n, srows, scols, ch, _ffi.RCV_32F, step=scols * ch * 4 + 16)
    dst = _canary_batch(ctx, n, drows, dcols, ch, _ffi.RCV_32F, pad=32)
    src.upload(frames)
    device.resize(src, dst)
    got = dst.download()
    worst = 0
    for i in range(n):
        want = oracle.resize_f32(frames[i] if ch > 1 else frames[i][..., 0], drows, dcols)
        worst = max(worst, int(_ulp_distance(got[i], want).max()))
    print(f"resize f32 {srows}x{scols} -> {drows}x{dcols} ch={ch}: max ULP distance {worst} ({'bit-exact' if worst == 0 else 'within 1 ULP' if worst <= 1 else 'FAIL'})")
    assert worst <= 1
    _assert_canaries(dst)
    src.free()
    dst.free()


@pytest.mark.parametrize("srows,scols,drows,dcols,ch", F32_GEOM)
@pytest.mark.parametrize("deg,scale", [(7.0, 1.0), (-31.0, 0.8), (90.0, 1.0), (0.0, 1.3)])
def test_warp_affine_f32(ctx, oracle, rng, srows, scols, drows, dcols, ch, deg, scale):
    n = 2
    t = np.deg2rad(deg)
    c, s_ = np.cos(t) / scale, np.sin(t) / scale
    cx, cy = scols / 2, srows / 2
    M = np.array([c, -s_, cx - c * (dcols / 2) + s_ * (drows / 2) + 0.37, s_, c, cy - s_ * (dcols / 2) - c * (drows / 2) - 0.61], np.float32)
    frames = (rng.standard_normal((n, srows, scols, ch)) * 1e-3).astype(np.float32)   # the magnitude of a Harris response map
    src = device.DeviceBatch(ctx, n, srows, scols, ch, _ffi.RCV_32F)
    dst = _canary_batch(ctx, n, drows, dcols, ch, _ffi.RCV_32F, pad=16)
    src.upload(frames)
    device.warp_affine(src, dst, M)
    got = dst.download()
    worst = 0
    for i in range(n):
        want = oracle.warp_affine_f32(frames[i] if ch > 1 else frames[i][..., 0], M, drows, dcols)
        worst = max(worst, int(_ulp_distance(got[i], want).max()))
    print(f"warpAffine f32 rot {deg} scale {scale} {srows}x{scols} -> {drows}x{dcols} ch={ch}: max ULP distance {worst}")
    assert worst <= 1
    _assert_canaries(dst)
    src.free()
    dst.free()


def test_geom_f32_special_values_and_arguments(ctx, oracle, rng):
    """inf / NaN / denormal / signed-zero taps go through the same arithmetic on both sides (NaN where the oracle has NaN); host
    Mats work; mixed depths, misaligned rows and the fused warp -> resize entry point (u8 only) are refused"""
    rows, cols = 40, 56
    img = (rng.standard_normal((rows, cols)) * 1e-2).astype(np.float32)
    img[3, 4], img[10, 20], img[11, 21], img[30, 5], img[31, 6] = np.inf, -np.inf, np.nan, np.float32(1e-42), -0.0
    M = np.array([0.98, -0.17, 2.5, 0.17, 0.98, -3.25], np.float32)
    src, dst = Mat.from_array(img), Mat(rows, cols, 1, _ffi.RCV_32F)
    imgproc.warp_affine(src, dst, M, ctx)
    assert int(_ulp_distance(dst.to_array(), oracle.warp_affine_f32(img, M, rows, cols)).max()) <= 1
    dst2 = Mat(23, 31, 1, _ffi.RCV_32F)
    imgproc.resize(src, dst2, ctx)
    assert int(_ulp_distance(dst2.to_array(), oracle.resize_f32(img, 23, 31)).max()) <= 1
    L = _ffi.lib()
    f = device.DeviceBatch(ctx, 1, rows, cols, 1, _ffi.RCV_32F)
    u = device.DeviceBatch(ctx, 1, rows, cols, 1)
    bf, bu = f.as_rcv(), u.as_rcv()
    Mp = M.ctypes.data_as(C.POINTER(C.c_float))
    assert L.rcv_resize_batch(ctx.handle, C.byref(bf), C.byref(bu)) == _ffi.RCV_ERR_UNSUPPORTED        # f32 -> u8
    assert L.rcv_warp_affine_batch(ctx.handle, C.byref(bu), C.byref(bf), Mp) == _ffi.RCV_ERR_UNSUPPORTED
    assert L.rcv_warp_affine_resize_batch(ctx.handle, C.byref(bf), C.byref(bf), Mp, rows, cols) == _ffi.RCV_ERR_UNSUPPORTED
    mis = f.as_rcv()
    mis.frame0.data = f.ptr.value + 2                                                                  # f32 samples must be 4-byte aligned
    mis.frame0.cap -= 4
    mis.frame0.rows -= 1
    assert L.rcv_resize_batch(ctx.handle, C.byref(mis), C.byref(bf)) == _ffi.RCV_ERR_ARG
    f.free()
    u.free()


def test_image_beyond_4gib(ctx, oracle):
    """one 50 000 x 30 000 BGR image (4.5 GB: in-frame byte offsets exceed 2^32, so the 24-bit / 32-bit offset fast paths must
    step aside): row slabs of filter2D, gray filter2D, warpAffine and the Harris pipeline against the oracle, including rows
    that lie beyond the 4 GiB mark"""
    rows, cols = 50000, 30000
    L = _ffi.lib()
    src = device.DeviceBatch(ctx, 1, rows, cols, 3)
    dst = device.DeviceBatch(ctx, 1, rows, cols, 3)
    device.synth(src, 1, 77, 0)
    k = oracle.bench_kernel7()

    def slab(batch, y0, y1, ch):
        n = (y1 - y0) * batch.step
        raw = np.empty(n, np.uint8)
        _ffi.check(L.rcv_download(ctx.handle, raw.ctypes.data, C.c_void_p(batch.ptr.value + y0 * batch.step), n), "rcv_download")
        return raw.reshape(y1 - y0, cols, ch)

    def check_stencil(sbatch, dbatch, ch, fn, halo):
        for a, b in [(0, 30), (24990, 25030), (47710, 47750), (rows - 30, rows)]:
            lo, hi = max(0, a - halo), min(rows, b + halo)
            s = slab(sbatch, lo, hi, ch)
            want = fn(s if ch > 1 else s[:, :, 0])
            got = slab(dbatch, a, b, ch if dbatch.channels > 1 else 1)
            got = got if dbatch.channels > 1 else got[:, :, 0]
            assert np.array_equal(want[a - lo: a - lo + (b - a)], got), (a, b)

    device.filter2d(src, dst, k, shift=6)
    check_stencil(src, dst, 3, lambda im: oracle.filter2d_i8(im, k, 6), 3)
    g = device.DeviceBatch(ctx, 1, rows, cols, 1)
    g2 = device.DeviceBatch(ctx, 1, rows, cols, 1)
    device.cvt_color(src, g, _ffi.RCV_BGR2GRAY)
    device.filter2d(g, g2, k, shift=6)
    check_stencil(g, g2, 1, lambda im: oracle.filter2d_i8(im, k, 6), 3)
    m = g2
    device.harris_pipeline(src, m, None, 2, 0.04, 1e-4)
    check_stencil(src, m, 3, lambda im: oracle.harris_pipeline(im, 2, 0.04, 1e-4), 8)
    M = np.array([1, 0, 0.5, 0, 1, 0.25], np.float32)   # fractional translation: output row y reads source rows y, y+1
    device.warp_affine(src, dst, M)
    for a, b in [(0, 20), (47720, 47760), (rows - 20, rows)]:
        hi = min(rows, b + 2)
        want = oracle.warp_affine(slab(src, a, hi, 3), M, hi - a, cols)
        keep = (b - a) - (1 if hi == rows else 0)       # the image's last row taps the zero border, the slab's does not
        assert np.array_equal(want[:keep], slab(dst, a, b, 3)[:keep]), (a, b)
    for x in (src, dst, g, g2):
        x.free()


class _Arena:
    """one device allocation + host mirror; images are placed at arbitrary byte offsets / steps / frame strides inside it"""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        self.dev = device.DeviceBatch(ctx, 1, 1, nbytes, 1)
        self.host = np.full(nbytes, 0xCD, np.uint8)

    def place(self, off, n, rows, cols, ch, depth, step, fs, frames=None):
        esz = {_ffi.RCV_8U: 1, _ffi.RCV_16S: 2, _ffi.RCV_32F: 4}[depth]
        rowb = cols * ch * esz
        assert off + (n - 1) * fs + (rows - 1) * step + rowb <= self.nbytes
        if frames is not None:
            raw = np.ascontiguousarray(frames).view(np.uint8).reshape(n, rows, rowb)
            for i in range(n):
                for r in range(rows):
                    o = off + i * fs + r * step
                    self.host[o:o + rowb] = raw[i, r]
        b = _ffi.rcv_batch()
        m = b.frame0
        m.data = self.dev.ptr.value + off
        m.cap = (rows - 1) * step + rowb if rows else 0
        m.step, m.rows, m.cols, m.channels, m.depth, m.device = step, rows, cols, ch, depth, _ffi.RCV_DEVICE
        b.frame_stride, b.n = fs, n
        return b, (off, n, rows, rowb, step, fs)

    def upload(self):
        self.dev.upload_bytes(self.host)

    def check(self, geo, want, dtype):
        """rows of the placed image == want, every other byte of the arena untouched"""
        off, n, rows, rowb, step, fs = geo
        got = self.dev.download_bytes()[: self.nbytes]
        mask = np.ones(self.nbytes, bool)
        w = np.ascontiguousarray(want).view(np.uint8).reshape(n, rows, rowb)
        for i in range(n):
            for r in range(rows):
                o = off + i * fs + r * step
                assert np.array_equal(got[o:o + rowb], w[i, r]), (i, r)
                mask[o:o + rowb] = False
        assert (got[mask] == 0xCD).all(), "bytes outside the image rows were written"

    def free(self):
        self.dev.free()


def test_misaligned_views_random(ctx, oracle):
    """every batch entry point on images that start at odd byte offsets with odd steps and frame strides: the dispatchers must fall
    back from the vector kernels to the generic HIP kernels and still match the oracle without touching a byte outside the rows"""
    L = _ffi.lib()
    r = np.random.default_rng(0xA11 + _SOAK_SEED)
    for case in range(12 * _SOAK):
        rows, cols, n = int(r.integers(1, 40)), int(r.integers(1, 70)), int(r.integers(1, 3))
        op = case % 9
        sch = {0: 3, 1: 2, 2: 4, 3: 1}.get(op, 3)
        if op == 1:
            cols += cols & 1
        srcf = r.integers(0, 256, size=(n, rows, cols, sch), dtype=np.uint8)
        soff, sstep = int(r.integers(0, 16)), cols * sch + int(r.integers(0, 10))
        sfs = rows * sstep + int(r.integers(0, 33))
        src = _Arena(ctx, soff + n * sfs + 64)
        bs, _ = src.place(soff, n, rows, cols, sch, _ffi.RCV_8U, sstep, sfs, srcf)
        src.upload()

        def dst_view(drows, dcols, ch, depth, align):
            esz = {_ffi.RCV_8U: 1, _ffi.RCV_16S: 2, _ffi.RCV_32F: 4}[depth]
            off = int(r.integers(0, 16)) // align * align
            step = dcols * ch * esz + int(r.integers(0, 10)) // align * align
            fs = drows * step + int(r.integers(0, 33)) // align * align
            a = _Arena(ctx, off + n * fs + 64)
            b, geo = a.place(off, n, drows, dcols, ch, depth, step, fs)
            a.upload()
            return a, b, geo

        imgs = [srcf[i] if sch > 1 else srcf[i, :, :, 0] for i in range(n)]
        outs = []
        if op == 0:     # BGR -> gray
            a, b, geo = dst_view(rows, cols, 1, _ffi.RCV_8U, 1)
            assert L.rcv_cvt_color_batch(ctx.handle, _ffi.RCV_BGR2GRAY, C.byref(bs), C.byref(b)) == 0
            outs.append((a, geo, np.stack([oracle.bgr2gray(im) for im in imgs]), np.uint8))
        elif op == 1:   # strided YUYV -> BGR
            a, b, geo = dst_view(rows, cols, 3, _ffi.RCV_8U, 1)
            assert L.rcv_cvt_color_batch(ctx.handle, _ffi.RCV_YUYV2BGR_STRIDED, C.byref(bs), C.byref(b)) == 0
            want = np.zeros((n, rows * cols * 3), np.uint8)
            for i in range(n):
                oracle.yuv422_to_bgr_strided(srcf[i].reshape(-1), cols * 2, rows, cols, False, want[i])
            outs.append((a, geo, want, np.uint8))
        elif op == 2:   # strided BGRA -> BGR
            a, b, geo = dst_view(rows, cols, 3, _ffi.RCV_8U, 1)
            assert L.rcv_cvt_color_batch(ctx.handle, _ffi.RCV_BGRA2BGR_STRIDED, C.byref(bs), C.byref(b)) == 0
            outs.append((a, geo, srcf[..., :3], np.uint8))
        elif op == 3:   # Sobel
            ax, bx, gx = dst_view(rows, cols, 1, _ffi.RCV_16S, 2)
            ay, by, gy = dst_view(rows, cols, 1, _ffi.RCV_16S, 2)
            assert L.rcv_sobel_batch(ctx.handle, C.byref(bs), C.byref(bx), C.byref(by)) == 0
            ws = [oracle.sobel(im) for im in imgs]
            outs += [(ax, gx, np.stack([w[0] for w in ws]), np.int16), (ay, gy, np.stack([w[1] for w in ws]), np.int16)]
        elif op == 4:   # integer Gaussian 5x5
            a, b, geo = dst_view(rows, cols, 3, _ffi.RCV_8U, 1)
            assert L.rcv_gaussian_blur_batch(ctx.handle, C.byref(bs), C.byref(b), 5, 0.0) == 0
            outs.append((a, geo, np.stack([oracle.gaussian_blur(im, 5, 0.0) for im in imgs]), np.uint8))
        elif op == 5:   # f32 Gaussian
            a, b, geo = dst_view(rows, cols, 3, _ffi.RCV_8U, 1)
            assert L.rcv_gaussian_blur_batch(ctx.handle, C.byref(bs), C.byref(b), 5, 1.2) == 0
            outs.append((a, geo, np.stack([oracle.gaussian_blur(im, 5, 1.2) for im in imgs]), np.uint8))
        elif op == 6:   # resize
            dr, dc = int(r.integers(1, 30)), int(r.integers(1, 50))
            a, b, geo = dst_view(dr, dc, 3, _ffi.RCV_8U, 1)
            assert L.rcv_resize_batch(ctx.handle, C.byref(bs), C.byref(b)) == 0
            outs.append((a, geo, np.stack([oracle.resize(im, dr, dc) for im in imgs]), np.uint8))
        elif op == 7:   # warpAffine
            dr, dc = int(r.integers(1, 30)), int(r.integers(1, 50))
            M = np.array([0.9, 0.15, r.uniform(-5, 5), -0.2, 1.1, r.uniform(-5, 5)], np.float32)
            a, b, geo = dst_view(dr, dc, 3, _ffi.RCV_8U, 1)
            assert L.rcv_warp_affine_batch(ctx.handle, C.byref(bs), C.byref(b), M.ctypes.data_as(C.POINTER(C.c_float))) == 0
            outs.append((a, geo, np.stack([oracle.warp_affine(im, M, dr, dc) for im in imgs]), np.uint8))
        else:           # Harris pipeline with the response
            am, bm, gm = dst_view(rows, cols, 1, _ffi.RCV_8U, 1)
            ar, br, gr = dst_view(rows, cols, 1, _ffi.RCV_32F, 4)
            assert L.rcv_harris_pipeline_batch(ctx.handle, C.byref(bs), C.byref(bm), C.byref(br), 2, 0.04, 1e-4) == 0
            ws = [oracle.harris_pipeline(im, 2, 0.04, 1e-4, True) for im in imgs]
            outs += [(am, gm, np.stack([w[0] for w in ws]), np.uint8), (ar, gr, np.stack([w[1] for w in ws]), np.float32)]
        for a, geo, want, dt in outs:
            a.check(geo, want.astype(dt), dt)
            a.free()
        src.free()


@pytest.mark.parametrize("rows,cols", [(20, 48), (37, 256), (130, 496)])
def test_no_out_of_bounds_writes(ctx, oracle, rows, cols):
    n = 2
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 1, 0x5EED0009, 0)
    k = oracle.bench_kernel7()
    d = _canary_batch(ctx, n, rows, cols, 3)
    for fn in (lambda: device.filter2d(src, d, k, shift=6), lambda: device.gaussian_blur(src, d, 5, 0.0),
               lambda: device.gaussian_blur(src, d, 7, 0.0), lambda: device.gaussian_blur(src, d, 7, 1.5),
               lambda: device.filter2d(src, d, (k / 64).astype(np.float32), delta=0.0),
               lambda: device.warp_affine(src, d, np.array([0.99, -0.12, 3.5, 0.12, 0.99, -2.25], np.float32)),
               lambda: device.rectangle(d, Rect(3, 3, cols - 6, rows - 6), Scalar(1, 2, 3), 2)):
        fn()
        _assert_canaries(d)
    g = _canary_batch(ctx, n, rows, cols, 1)
    device.cvt_color(src, g, _ffi.RCV_BGR2GRAY)
    _assert_canaries(g)
    dx, dy = _canary_batch(ctx, n, rows, cols, 1, _ffi.RCV_16S), _canary_batch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    device.sobel(g, dx, dy)
    _assert_canaries(dx)
    _assert_canaries(dy)
    m, r = _canary_batch(ctx, n, rows, cols, 1), _canary_batch(ctx, n, rows, cols, 1, _ffi.RCV_32F)
    device.harris_pipeline(src, m, r, 2, 0.04, 1e-4)
    _assert_canaries(m)
    _assert_canaries(r)
    small = _canary_batch(ctx, n, rows // 2, cols // 2, 3)
    device.resize(src, small)
    _assert_canaries(small)
    odd = _canary_batch(ctx, n, rows // 3 + 1, cols // 3 + 1, 3)
    device.resize(src, odd)
    _assert_canaries(odd)


def test_sobel_ragged_and_unaligned_shapes(ctx, oracle):
    """the register-window Sobel on the shapes real callers produce (Mat::new: step = cols * channels, rustcv/src/core/mat.rs:18-29):
    25 x RCV_SOAK seeded cases, widths 8..1100 that are mostly NOT multiples of 8, odd steps (byte-aligned rows), i16 outputs on
    2-byte-aligned rows with canaries in the padding, gray and BGR sources, batches of 1..3 -- and the kernel that ran is the
    row kernel's RAG instantiation, not a generic per-sample kernel"""
    L = _ffi.lib()
    r = np.random.default_rng(0x50BE1 + _SOAK_SEED)
    for case in range(25 * _SOAK):
        cols = int(r.integers(8, 1101))
        rows = int(r.integers(2, 90))
        ch = 3 if case % 3 == 2 else 1
        n = int(r.integers(1, 4))
        sp = int(r.integers(0, 4)) if case % 4 else 0            # source row padding: odd steps
        op = 2 * int(r.integers(0, 9))                            # i16 output padding: rows stay 2-byte aligned, rarely 16
        src = device.DeviceBatch(ctx, n, rows, cols, ch, step=cols * ch + sp, frame_stride=rows * (cols * ch + sp) + int(r.integers(0, 7)))
        dx = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_16S, pad=op)
        dy = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_16S, pad=op)
        frames = r.integers(0, 256, size=(n, rows, cols, ch), dtype=np.uint8)
        src.upload(frames)
        L.rcv__debug_kernels_reset()
        device.sobel(src, dx, dy)
        gx, gy = dx.download(), dy.download()
        assert "k_sobel_rows<" in L.rcv__debug_kernels().decode(), (rows, cols, ch, L.rcv__debug_kernels().decode())
        for i in range(n):
            wx, wy = oracle.sobel(oracle.bgr2gray(frames[i]) if ch == 3 else frames[i][..., 0])
            assert np.array_equal(gx[i], wx) and np.array_equal(gy[i], wy), (case, rows, cols, ch, n, sp, op)
        _assert_canaries(dx)
        _assert_canaries(dy)
        for b in (src, dx, dy):
            b.free()


def test_harris_ragged_and_unaligned_shapes(ctx, oracle):
    """the fused Harris kernel on widths that are not multiples of 8 and on byte-aligned rows (Mat::new: step = cols * channels):
    20 x RCV_SOAK seeded cases, BGR / YUYV / gray sources, mask only / mask + response / response only (cornerHarris), batches of
    1..3, canaries around the outputs -- mask and f32 response bit for bit, and the kernel that ran is the fused one"""
    L = _ffi.lib()
    r = np.random.default_rng(0x4A221 + _SOAK_SEED)
    for case in range(20 * _SOAK):
        kind = case % 4                                         # 0 BGR, 1 YUYV, 2 gray pipeline, 3 cornerHarris (gray -> response)
        cols = int(r.integers(8, 1050))
        if kind == 1:
            cols += cols & 1                                    # YUYV: whole macropixels
        rows = int(r.integers(4, 70))
        n = int(r.integers(1, 4))
        ch = (3, 2, 1, 1)[kind]
        sp = int(r.integers(0, 4)) if case % 5 else 0
        want_resp = kind == 3 or bool(case & 8)
        src = device.DeviceBatch(ctx, n, rows, cols, ch, step=cols * ch + sp, frame_stride=rows * (cols * ch + sp) + int(r.integers(0, 5)))
        mask = _canary_batch(ctx, n, rows, cols, 1, pad=int(r.integers(0, 6))) if kind != 3 else None
        resp = _canary_batch(ctx, n, rows, cols, 1, depth=_ffi.RCV_32F, pad=4 * int(r.integers(0, 5))) if want_resp else None
        frames = r.integers(0, 256, size=(n, rows, cols, ch), dtype=np.uint8)
        frames[:, rows // 3: rows // 3 + 3, cols // 4: cols // 4 + 5] = 255      # a few real corners
        frames[:, :, -2:] = 0                                                    # structure right at the ragged edge
        src.upload(frames)
        thr = 1e-5
        L.rcv__debug_kernels_reset()
        if kind == 3:
            device.corner_harris(src, resp, 2, 0.04)
        else:
            device.harris_pipeline(src, mask, resp, 2, 0.04, thr)
        gm = mask.download() if mask is not None else None
        gr = resp.download() if resp is not None else None
        assert "k_harris_fused<" in L.rcv__debug_kernels().decode(), (kind, rows, cols, L.rcv__debug_kernels().decode())
        for i in range(n):
            if kind == 0:
                bgr = frames[i]
            elif kind == 1:
                flat = np.zeros(rows * cols * 3, np.uint8)
                oracle.yuv422_to_bgr_strided(frames[i].reshape(-1), cols * 2, rows, cols, False, flat)
                bgr = flat.reshape(rows, cols, 3)
            if kind in (0, 1):
                wm, wr = oracle.harris_pipeline(bgr, 2, 0.04, thr, want_resp=True)
            else:
                wr = oracle.corner_harris(frames[i][..., 0], 2, 0.04)
                wm = oracle.nms3x3(wr, thr)
            if gm is not None:
                assert np.array_equal(gm[i], wm), (case, kind, rows, cols, n, sp)
            if gr is not None:
                assert np.array_equal(gr[i].view(np.uint32), wr.view(np.uint32)), (case, kind, rows, cols, n, sp)
        for b in (mask, resp):
            if b is not None:
                _assert_canaries(b)
                b.free()
        src.free()


# ---- round 3: seeded random-shape soaks of the kernels added this round (RCV_SOAK scales the case counts) -------------------------

def test_gaussian_int_rows_kernel_random_shapes(ctx, oracle, knob):
    """4 x RCV_SOAK random cases for the register-window integer Gaussian: any height, widths with 16-byte rows (1 / 3 channels),
    ksize 3 / 5, segment heights 0 (planned) .. 40, batch 1..4, padded steps, pixel fields with saturated and zero regions"""
    knob("RCV_GAUSS_ROWS")
    r = np.random.default_rng(0x6A55 + _SOAK_SEED)
    L = _ffi.lib()
    for case in range(4 * _SOAK):
        ch = int(r.choice([1, 3]))
        cols = 16 * int(r.integers(1, 90)) if ch == 3 else 16 * int(r.integers(2, 260))
        rows = int(r.integers(3, 120))
        ksize = int(r.choice([3, 5]))
        n = int(r.integers(1, 5))
        seg = int(r.choice([0, 0, 3, 4, 7, 16, 40]))
        if seg:
            knob("RCV_GR_SEG", seg)
        else:
            knob("RCV_GR_SEG", 0)
        frames = r.integers(0, 256, size=(n, rows, cols, ch), dtype=np.uint8)
        frames[:, : rows // 4] = 255
        frames[:, -(rows // 5 + 1):, : cols // 2] = 0
        src = device.DeviceBatch(ctx, n, rows, cols, ch, step=cols * ch + 16 * int(r.integers(0, 4)))
        dst = _canary_batch(ctx, n, rows, cols, ch, pad=16 * int(r.integers(1, 4)))
        src.upload(frames if ch == 3 else frames[..., 0])
        L.rcv__debug_kernels_reset()
        device.gaussian_blur(src, dst, ksize, 0.0)
        assert "k_gauss_rows<" in L.rcv__debug_kernels().decode(), (case, rows, cols, ch)
        got = dst.download()
        for i in range(n):
            want = oracle.gaussian_blur(frames[i] if ch == 3 else frames[i][..., 0], ksize, 0.0)
            assert np.array_equal(got[i], want), (case, rows, cols, ch, ksize, n, seg, i)
        _assert_canaries(dst)
        src.free()
        dst.free()


def test_geometry_f32_random_shapes(ctx, oracle):
    """2 x RCV_SOAK random cases each for the RCV_32F resize and warpAffine: shapes 1..90 px, 1 / 3 / 4 channels, up- and
    down-scales, random affine maps (part of the output outside the source), values of mixed magnitude, padded steps: within
    1 ULP of the oracle (expected and printed: bit-exact)"""
    r = np.random.default_rng(0xF32 + _SOAK_SEED)
    worst = 0
    for case in range(2 * _SOAK):
        ch = int(r.choice([1, 3, 4]))
        srows, scols, drows, dcols = (int(v) for v in r.integers(1, 90, size=4))
        n = int(r.integers(1, 4))
        frames = (r.standard_normal((n, srows, scols, ch)) * r.choice([1e-6, 1e-3, 1.0, 255.0, 3e7])).astype(np.float32)
        src = device.DeviceBatch(ctx, n, srows, scols, ch, _ffi.RCV_32F, step=scols * ch * 4 + 4 * int(r.integers(0, 5)))
        for op in ("resize", "warp"):
            dst = _canary_batch(ctx, n, drows, dcols, ch, _ffi.RCV_32F, pad=4 * int(r.integers(1, 9)))
            src.upload(frames)
            if op == "resize":
                device.resize(src, dst)
            else:
                t = r.uniform(-np.pi, np.pi)
                sc = r.uniform(0.3, 2.5)
                M = np.array([np.cos(t) * sc, -np.sin(t) * sc, r.uniform(-10, scols), np.sin(t) * sc, np.cos(t) * sc, r.uniform(-10, srows)], np.float32)
                device.warp_affine(src, dst, M)
            got = dst.download()
            for i in range(n):
                f = frames[i] if ch > 1 else frames[i][..., 0]
                want = oracle.resize_f32(f, drows, dcols) if op == "resize" else oracle.warp_affine_f32(f, M, drows, dcols)
                dist = int(_ulp_distance(got[i], want).max())
                worst = max(worst, dist)
                assert dist <= 1, (case, op, srows, scols, drows, dcols, ch, i, dist)
            _assert_canaries(dst)
            dst.free()
        src.free()
    print(f"f32 geometry, {2 * _SOAK} random cases: max ULP distance {worst}")


def test_row_kernel_random_batches(ctx, oracle, knob):
    """RCV_SOAK random launches of the row-streaming MFMA kernel in its LARGE-launch regime (chained or tapered bands when the batch is a
    multiple of 8, plain bands otherwise): 8..24 frames of 16k-aligned widths, ksize 3 / 5 / 7,
    random i8 weights and shifts; three frames of each launch against the oracle"""
    r = np.random.default_rng(0x7A9E + _SOAK_SEED)
    L = _ffi.lib()
    for case in range(max(2, _SOAK // 2)):
        knob("RCV_F7_ROWS")
        knob("RCV_FR_CHAIN", int(r.choice([-1, 0, 1])))
        knob("RCV_FR_CHAIN_ROWS", int(r.choice([0, 9, 21, 64])))
        ksize = int(r.choice([3, 5, 7]))
        n = int(r.choice([8, 9, 16, 17, 24]))
        rows, cols = int(r.integers(300, 700)), 16 * int(r.integers(60, 130))
        k = r.integers(-128, 128, size=(ksize, ksize)).astype(np.int8)
        shift = int(r.integers(0, 13))
        src = device.DeviceBatch(ctx, n, rows, cols, 3)
        dst = _canary_batch(ctx, n, rows, cols, 3, pad=16)
        device.synth(src, 1, 0x5EED0100 + case + _SOAK_SEED, 0)
        L.rcv__debug_kernels_reset()
        device.filter2d(src, dst, k, shift=shift)
        assert "k_filter_rows_" in L.rcv__debug_kernels().decode()
        frames = src.download()
        got = dst.download()
        for i in sorted({0, n // 2, n - 1}):
            assert np.array_equal(got[i], oracle.filter2d_i8(frames[i], k, shift)), (case, n, rows, cols, ksize, shift, i)
        _assert_canaries(dst)
        src.free()
        dst.free()


def test_geometry_f32_at_8k(ctx, oracle):
    """north_star's 1-ULP clause at BASELINE configs[3] size: one 4320 x 7680 RCV_32F frame (one channel: a response map) through the
    7-degree warp and the 4x down-scale to 1080p, every pixel against the oracle (expected and printed: 0 ULP)"""
    rows, cols = 4320, 7680
    r = np.random.default_rng(0x8F32)
    frame = (r.standard_normal((rows, cols)) * 1e-3).astype(np.float32)
    src = device.DeviceBatch(ctx, 1, rows, cols, 1, _ffi.RCV_32F)
    src.upload(frame[None, :, :, None])
    M = _rot(7.0, cols / 2, rows / 2, 13.25, -8.5)
    dst = device.DeviceBatch(ctx, 1, rows, cols, 1, _ffi.RCV_32F)
    device.warp_affine(src, dst, M)
    d_warp = int(_ulp_distance(dst.download()[0], oracle.warp_affine_f32(frame, M, rows, cols)).max())
    small = device.DeviceBatch(ctx, 1, 1080, 1920, 1, _ffi.RCV_32F)
    device.resize(src, small)
    d_resize = int(_ulp_distance(small.download()[0], oracle.resize_f32(frame, 1080, 1920)).max())
    print(f"f32 geometry at 8K: warp max ULP distance {d_warp}, resize -> 1080p {d_resize}")
    assert d_warp <= 1 and d_resize <= 1
    for b in (src, dst, small):
        b.free()


@pytest.mark.parametrize("kind", ["bgr", "gray", "f32"])
@pytest.mark.parametrize("M", ["rot30", "shift-out", "zoom-out", "corner", "far-out", "edge-exact"])
def test_warp_affine_border_tiles_staged(ctx, oracle, rng, kind, M):
    """(round 4) tiles whose source patch leaves the source run on the LDS-staged kernels too when the source width is a multiple
    of 4: chunks outside the source are staged as zeros, which is the specification's constant border tap by tap.  Maps that put
    much of the destination outside the source, on its edges (coordinates of exactly -1, 0, cols - 1, cols) and wholly outside;
    f32 sources carry inf / NaN next to the border; the result is the oracle's, and no launch falls back to the gather kernels"""
    n, sr, sc, dr, dc = 5, 200, 360, 230, 412
    Ms = {"rot30": _rot(30.0, sc / 2, sr / 2, 11.25, -7.5), "shift-out": np.array([1, 0, -77.25, 0, 1, 51.5], np.float32),
          "zoom-out": np.array([1.3, 0.02, -40.0, -0.03, 1.3, -35.0], np.float32), "corner": _rot(-12.0, 0.0, 0.0, -20.5, 160.25),
          "far-out": np.array([1, 0, 5000.0, 0, 1, 3.0], np.float32), "edge-exact": np.array([1, 0, -1.0, 0, 1, -1.0], np.float32)}[M]
    if kind == "f32":
        frames = rng.standard_normal((n, sr, sc)).astype(np.float32)
        frames[:, 0, :7] = np.inf; frames[:, -1, -5:] = np.nan; frames[:, 3:6, 0] = -np.inf
        src = device.DeviceBatch(ctx, n, sr, sc, 1, _ffi.RCV_32F)
        dst = _canary_batch(ctx, n, dr, dc, 1, _ffi.RCV_32F, pad=20)
        src.upload(frames[..., None])
        want = [oracle.warp_affine_f32(frames[i], Ms, dr, dc) for i in range(n)]
    else:
        ch = 3 if kind == "bgr" else 1
        frames = rng.integers(1, 256, size=(n, sr, sc, ch), dtype=np.uint8)   # (no zeros inside: a border tap mistaken for data shows)
        src = device.DeviceBatch(ctx, n, sr, sc, ch)
        dst = _canary_batch(ctx, n, dr, dc, ch, pad=20)
        src.upload(frames)
        want = [oracle.warp_affine(frames[i] if ch == 3 else frames[i, :, :, 0], Ms, dr, dc).reshape(dr, dc, ch) for i in range(n)]
    launched = _kernels_launched(ctx, lambda: device.warp_affine(src, dst, Ms))
    assert {"bgr": "k_warp_affine_lds<3", "gray": "k_warp_gray_lds4", "f32": "k_warp_f32_lds"}[kind] in launched, launched
    got = dst.download()
    for i in range(n):
        if kind == "f32":
            assert int(_ulp_distance(got[i].reshape(dr, dc), want[i]).max()) == 0, (kind, M, i)   # (NaN == NaN, whatever its payload)
        else:
            assert np.array_equal(got[i].reshape(want[i].shape), want[i]), (kind, M, i)
    _assert_canaries(dst)


@pytest.mark.parametrize("kind", ["bgr", "gray", "f32"])
@pytest.mark.parametrize("strip", [None, 0, 1, 5, 6, 7, 23, 40])
def test_warp_affine_tile_orders(ctx, oracle, rng, knob, kind, strip):
    """tile orders of the LDS-staged warp kernels (round 4): XCD-contiguous runs walked in vertical strips of `strip` tile columns --
    a grid of 23 x 9 tiles (the last strip narrower, wider than the grid, one column) x 3 frame groups, 17 frames in groups of 16 / 8 /
    the default; every tile written exactly once with the oracle's bytes (canaries around the destination)"""
    n, sr, sc, dr, dc = 17, 300, 1500, 270, 1460   # 23 x 9 tiles of 64 x 32
    M = _rot(7.0, dc / 2, dr / 2, 13.25, 9.5)
    if kind == "f32":
        frames = rng.standard_normal((n, sr, sc)).astype(np.float32)
        src = device.DeviceBatch(ctx, n, sr, sc, 1, _ffi.RCV_32F)
        dst = _canary_batch(ctx, n, dr, dc, 1, _ffi.RCV_32F, pad=20)
        src.upload(frames[..., None])
        want = [oracle.warp_affine_f32(frames[i], M, dr, dc) for i in range(n)]
    else:
        ch = 3 if kind == "bgr" else 1
        frames = rng.integers(0, 256, size=(n, sr, sc, ch), dtype=np.uint8)
        src = device.DeviceBatch(ctx, n, sr, sc, ch)
        dst = _canary_batch(ctx, n, dr, dc, ch, pad=20)
        src.upload(frames)
        want = [oracle.warp_affine(frames[i] if ch == 3 else frames[i, :, :, 0], M, dr, dc).reshape(dr, dc, ch) for i in range(n)]
    for fpg in (0, 8, 16):
        if strip is not None or fpg:
            knob("RCV_WARP_FPG", fpg + (256 * (strip + 1) if strip is not None else 0))
        dst.memset(0xCD)
        launched = _kernels_launched(ctx, lambda: device.warp_affine(src, dst, M))
        assert {"bgr": "k_warp_affine_lds<3", "gray": "k_warp_gray_lds4", "f32": "k_warp_f32_lds"}[kind] in launched, launched
        got = dst.download()
        for i in range(n):
            if kind == "f32":
                assert np.array_equal(got[i].reshape(dr, dc).view(np.uint32), want[i].view(np.uint32)), (kind, strip, fpg, i)
            else:
                assert np.array_equal(got[i].reshape(want[i].shape), want[i]), (kind, strip, fpg, i)
        _assert_canaries(dst)


@pytest.mark.parametrize("fpg,xcd", [(0, 0), (2, 1), (8, 0), (3, 1)])
@pytest.mark.parametrize("M", ["rot7", "rot-20", "shear", "shift", "flip", "big"])
def test_warp_affine_f32_lds_kernel(ctx, oracle, rng, knob, fpg, xcd, M):
    """one-channel RCV_32F warpAffine on the LDS-staged kernel (k_warp_f32_lds): interior tiles from LDS, border / outside tiles through
    the per-pixel code, frame groups with a short last group, both tile orders, ragged widths, padded steps -- within 1 ULP of the oracle
    (expected: bit-exact) and identical to the per-pixel kernel (RCV_WARP_LDS=0)"""
    if fpg or xcd:
        knob("RCV_WARP_FPG", fpg + (256 * (1 + fpg) if xcd else 0))   # xcd: XCD-contiguous runs, strips of `fpg` tile columns
    n, sr, sc, dr, dc = 5, 150, 300, 131, 259
    Ms = {"rot7": _rot(7.0, dc / 2, dr / 2, 13.25, 9.5), "rot-20": _rot(-20.0, dc / 2, dr / 2, 20.0, 12.0),
          "shear": np.array([1, 0.25, 3.5, -0.125, 1, 18.25], np.float32), "shift": np.array([1, 0, 7.5, 0, 1, 3.25], np.float32),
          "flip": np.array([-1, 0, dc + 5.5, 0, -1, dr + 3.25], np.float32), "big": np.array([3, 0, 0, 0, 3, 0], np.float32)}[M]
    frames = (rng.standard_normal((n, sr, sc)) * rng.choice([1e-3, 1.0, 3e4])).astype(np.float32)
    frames[:, 10:20, 30:60] = np.float32(1e-41)          # subnormals come through unchanged
    src = device.DeviceBatch(ctx, n, sr, sc, 1, _ffi.RCV_32F, step=sc * 4 + 12)
    dst = _canary_batch(ctx, n, dr, dc, 1, _ffi.RCV_32F, pad=20)
    src.upload(frames[..., None])
    L = _ffi.lib()
    L.rcv__debug_kernels_reset()
    device.warp_affine(src, dst, Ms)
    assert ("k_warp_f32_lds" in L.rcv__debug_kernels().decode()) == (M != "big")
    got = dst.download().copy()
    worst = 0
    for i in range(n):
        worst = max(worst, int(_ulp_distance(got[i], oracle.warp_affine_f32(frames[i], Ms, dr, dc)).max()))
    assert worst <= 1, worst
    _assert_canaries(dst)
    knob("RCV_WARP_LDS", 0)
    dst.memset(0xCD)
    L.rcv__debug_kernels_reset()
    device.warp_affine(src, dst, Ms)
    assert "k_warp_f32_lds" not in L.rcv__debug_kernels().decode()
    assert np.array_equal(dst.download().view(np.uint32), got.view(np.uint32))
    src.free()
    dst.free()


@pytest.mark.parametrize("pad_src,pad_dst", [(0, 0), (4, 8), (1, 0), (0, 3), (2, 2)])
def test_four_channel_geometry_dword_and_byte_taps(ctx, oracle, rng, pad_src, pad_dst):
    """4-channel u8 images through the generic resize / warpAffine kernels: taps as dwords and the pixel as one dword store on 4-byte
    aligned rows, byte by byte on any other step -- batches of 3, padded steps, against the oracle"""
    n, sr, sc, dr, dc = 3, 57, 83, 41, 122
    frames = rng.integers(0, 256, size=(n, sr, sc, 4), dtype=np.uint8)
    src = device.DeviceBatch(ctx, n, sr, sc, 4, step=sc * 4 + pad_src)
    src.upload(frames)
    M = _rot(7.0, sc / 2, sr / 2, 3.25, -1.5)
    for op in ("resize", "warp"):
        dst = device.DeviceBatch(ctx, n, dr, dc, 4, step=dc * 4 + pad_dst)
        dst.memset(0xEE)
        if op == "resize":
            device.resize(src, dst)
        else:
            device.warp_affine(src, dst, M)
        got = dst.download()
        for i in range(n):
            want = oracle.resize(frames[i], dr, dc) if op == "resize" else oracle.warp_affine(frames[i], M, dr, dc)
            assert np.array_equal(got[i], want), (op, i)
        dst.free()
    src.free()


def test_batches_larger_than_the_whole_baseline_job(ctx, oracle):
    """maximum sizes: 520 4K frames in ONE batch (12.9 GB in, more than config 5's whole 8-GPU job) through the filter, the Harris pipeline
    and the Sobel of a BGR source, 130 8K frames (12.9 GB) through the fused warp + down-scale and the warp: first / middle / last frame
    against the oracle -- frame offsets beyond 4 GB, band and tile counts beyond the BASELINE launches"""
    from bench import bench_kernel7, warp_matrix
    n, rows, cols = 520, 2160, 3840
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 1, 0x5EED0003, 0)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    k = bench_kernel7()
    device.filter2d(src, dst, k, shift=6)
    for i in (0, 259, 519):
        assert np.array_equal(dst.download_frame(i), oracle.filter2d_i8(src.download_frame(i), k, 6)), i
    dst.free()
    mask = device.DeviceBatch(ctx, n, rows, cols, 1)
    device.harris_pipeline(src, mask, None, 2, 0.04, 1e-4)
    for i in (0, 300, 519):
        assert np.array_equal(mask.download_frame(i), oracle.harris_pipeline(src.download_frame(i), 2, 0.04, 1e-4)), i
    dx, dy = device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S), device.DeviceBatch(ctx, n, rows, cols, 1, _ffi.RCV_16S)
    device.sobel(src, dx, dy)
    for i in (0, 519):
        gx, gy = oracle.sobel(oracle.bgr2gray(src.download_frame(i)))
        assert np.array_equal(dx.download_frame(i), gx) and np.array_equal(dy.download_frame(i), gy), i
    for b in (src, mask, dx, dy):
        b.free()
    n8 = 130
    s8 = device.DeviceBatch(ctx, n8, 4320, 7680, 3)
    device.synth(s8, 0, 0x5EED0004, 0)
    d8 = device.DeviceBatch(ctx, n8, 1080, 1920, 3)
    M = warp_matrix()
    device.warp_affine_resize(s8, d8, M, 4320, 7680)
    w8 = device.DeviceBatch(ctx, n8, 4320, 7680, 3)
    device.warp_affine(s8, w8, M)
    for i in (0, 129):
        warped = oracle.warp_affine(s8.download_frame(i), M, 4320, 7680)
        assert np.array_equal(w8.download_frame(i), warped), i
        assert np.array_equal(d8.download_frame(i), oracle.resize(warped, 1080, 1920)), i
    for b in (s8, d8, w8):
        b.free()


def test_frame_larger_than_4_gib():
    """maximum sizes: ONE 36 000 x 40 000 BGR frame (4.32 GB: rows * step > 2^32) through filter2D, both Gaussians, BGR2GRAY, Sobel, the Harris
    pipeline, warpAffine and resize -- the kernels that keep 32-bit in-frame offsets must hand it over; top and bottom 96 rows of every
    result (the bottom ones lie beyond the 4-GiB offset) against the oracle on the matching source slices (tools/huge_frame.py)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "huge_frame.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "mismatches: 0" in r.stdout

