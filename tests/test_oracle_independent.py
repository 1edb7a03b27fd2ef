"""The C oracle against independent second implementations (tests/npref.py, written from SURVEY.md §8-A and the reference's
Rust, not from the C code) on images of at least 200 x 300: the pin of the oracle rows that the reference itself cannot pin
(DESIGN_HISTORY.md §3 lists, per op, which check pins it)."""
import numpy as np
import pytest

import npref


@pytest.fixture(scope="module")
def big(request):
    rng = np.random.default_rng(0x1DE9)
    bgr = rng.integers(0, 256, size=(211, 317, 3), dtype=np.uint8)
    bgr[40:90, 100:180] = 255          # saturated and flat regions: exact ties, zero gradients
    bgr[120:160, 20:90] = 0
    gray = rng.integers(0, 256, size=(203, 301), dtype=np.uint8)
    gray[50:120, 60:200] = (np.add.outer(np.arange(70), np.arange(140)) // 16 % 2 * 200 + 20).astype(np.uint8)   # checkerboard: real corners
    return bgr, gray


@pytest.mark.parametrize("ksize,sigma", [(3, 0.8), (5, 1.2), (7, 2.0), (9, 1.7)])
def test_gaussian_sigma_bit_exact(oracle, big, ksize, sigma):
    bgr, gray = big
    assert np.array_equal(oracle.gaussian_taps_f32(ksize, sigma), npref.gaussian_taps(ksize, sigma))
    assert np.array_equal(oracle.gaussian_blur(bgr, ksize, sigma), npref.gaussian_blur_sigma(bgr, ksize, sigma))
    assert np.array_equal(oracle.gaussian_blur(gray, ksize, sigma), npref.gaussian_blur_sigma(gray[:, :, None], ksize, sigma)[:, :, 0])


@pytest.mark.parametrize("block", [1, 2, 3, 5])
@pytest.mark.parametrize("k", [0.04, 0.06])
def test_corner_harris_bit_exact(oracle, big, block, k):
    _, gray = big
    dx, dy = oracle.sobel(gray)
    rdx, rdy = npref.sobel(gray)
    assert np.array_equal(dx, rdx) and np.array_equal(dy, rdy)
    got, want = oracle.corner_harris(gray, block, k), npref.corner_harris(gray, block, k)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))        # the same f32 bits
    thr = 1e-6
    assert np.array_equal(oracle.nms3x3(got, thr), npref.nms3x3(want, thr))
    if block >= 2:   # (block 1: the structure tensor of a single sample has rank 1, no positive response)
        assert oracle.nms3x3(got, thr).any()                                  # the checkerboard has corners


def _close_to_rounding(got, exact, what):
    """u8 result of round-half-up(f32 evaluation) against the float64 value of the same sampling rule: equal, except where the
    exact value sits within f32 noise of a .5 boundary -- there 1 LSB either way is the north star's tolerance.  The noise: a
    source coordinate of a few hundred carries ~3e-5 of f32 rounding error, times a tap difference of up to 255 -> 1e-2."""
    want = npref.round_half_up_u8(exact).reshape(got.shape)
    diff = got.astype(np.int32) - want.astype(np.int32)
    bad = diff != 0
    assert np.abs(diff).max() <= 1, what
    frac = exact.reshape(got.shape) + 0.5
    dist = np.abs(frac - np.rint(frac))                      # distance of (v + 0.5) from an integer
    assert (dist[bad] < 1e-2).all(), (what, float(dist[bad].max()) if bad.any() else 0.0)
    assert bad.mean() < 5e-3, (what, float(bad.mean()))


@pytest.mark.parametrize("drows,dcols", [(97, 143), (300, 421), (211, 159), (53, 317), (422, 634), (1, 1)])
def test_resize_general_scale(oracle, big, drows, dcols):
    bgr, gray = big
    _close_to_rounding(oracle.resize(bgr, drows, dcols), npref.resize_f64(bgr, drows, dcols), f"resize bgr {drows}x{dcols}")
    _close_to_rounding(oracle.resize(gray, drows, dcols), npref.resize_f64(gray, drows, dcols)[:, :, 0], f"resize gray {drows}x{dcols}")


@pytest.mark.parametrize("deg,tx,ty,scale", [(7.0, 13.25, -8.5, 1.0), (-31.0, 40.0, 10.0, 0.8), (90.0, 0.0, 0.0, 1.0), (3.0, -250.0, 5.0, 1.3), (180.0, 0.3, 0.7, 1.0)])
def test_warp_affine_under_rotation(oracle, big, deg, tx, ty, scale):
    bgr, gray = big
    h, w = bgr.shape[:2]
    t = np.deg2rad(deg)
    c, s = np.cos(t) * scale, np.sin(t) * scale
    cx, cy = (w - 1) / 2, (h - 1) / 2
    M = np.array([c, -s, cx - c * cx + s * cy + tx, s, c, cy - s * cx - c * cy + ty], np.float32)
    for drows, dcols in ((h, w), (150, 400)):
        _close_to_rounding(oracle.warp_affine(bgr, M, drows, dcols), npref.warp_affine_f64(bgr, M, drows, dcols), f"warp {deg} {drows}x{dcols}")
    hg, wg = gray.shape
    _close_to_rounding(oracle.warp_affine(gray, M, hg, wg), npref.warp_affine_f64(gray, M, hg, wg)[:, :, 0], f"warp gray {deg}")


@pytest.mark.parametrize("ch", [1, 3])
def test_f32_geometry_against_float64(oracle, ch):
    """RCV_32F resize / warpAffine (round 3): the f32 oracle against a float64 evaluation of the same sampling rule on a
    203 x 301 field of Harris-response magnitude.  The specification fixes f32 COORDINATES (sx = fmaf(M0, x, fmaf(M1, y, M2)) in
    f32), so the two differ by the coordinate rounding times the local slope plus a few value roundings: bounded well below 1e-4 of
    the field's range -- a wrong tap, weight, clamp or border rule would be off by the order of the range itself."""
    rng = np.random.default_rng(0xF32)
    img = (rng.standard_normal((203, 301, ch)) * 1e-3).astype(np.float32)
    img = img[..., 0] if ch == 1 else img
    span = float(np.abs(img).max())
    for drows, dcols in ((203, 301), (57, 80), (400, 450), (1, 1)):
        got = oracle.resize_f32(img, drows, dcols).reshape(drows, dcols, ch)
        assert np.abs(got - npref.resize_f64(img, drows, dcols)).max() <= 1e-4 * span
    for deg, tx, ty, scale in ((7.0, 13.25, -8.5, 1.0), (-31.0, 40.0, 10.0, 0.8), (90.0, 0.0, 0.0, 1.0), (180.0, 0.3, 0.7, 1.3)):
        t = np.deg2rad(deg)
        c, s = np.cos(t) * scale, np.sin(t) * scale
        cx, cy = 150.0, 101.0
        M = np.array([c, -s, cx - c * cx + s * cy + tx, s, c, cy - s * cx - c * cy + ty], np.float32)
        for drows, dcols in ((203, 301), (150, 400)):
            got = oracle.warp_affine_f32(img, M, drows, dcols).reshape(drows, dcols, ch)
            want = npref.warp_affine_f64(img, M, drows, dcols)
            # at the source's border a coordinate within rounding of -1 / cols switches a whole tap on or off: compare away from it
            bad = np.abs(got - want) > 1e-4 * span
            assert bad.mean() < 1e-3, (deg, drows, dcols, bad.sum())


def test_rectangle_transliteration_including_wraps(oracle):
    """drawing.rs:67-106 line by line in Python (u64 wrapping index arithmetic) against the C restatement: clipped rectangles,
    thickness larger than the rectangle (rows / columns run past the far edge and, through the wrapping index, onto other
    rows), steps that are not a multiple of 3 (wrapped pixels land off the pixel grid), zero and negative thickness"""
    rng = np.random.default_rng(0x4EC7)
    for case in range(400):
        rows, cols = int(rng.integers(1, 40)), int(rng.integers(1, 50))
        step = cols * 3 + int(rng.integers(0, 8))
        x, y = int(rng.integers(-20, cols + 10)), int(rng.integers(-20, rows + 10))
        w, h = int(rng.integers(-3, cols + 25)), int(rng.integers(-3, rows + 25))
        thickness = int(rng.integers(-2, 12)) if case % 3 else int(rng.integers(8, 60))
        cap = rows * step - int(rng.integers(0, 3)) if case % 5 == 0 else rows * step    # a Vec a little shorter than rows * step
        color = tuple(int(v) for v in rng.integers(0, 256, size=3))
        base = rng.integers(0, 256, size=cap, dtype=np.uint8)
        want = base.copy()
        npref.rectangle_rs(want, rows, cols, step, x, y, w, h, color, thickness)
        got = base.copy()
        oracle.rectangle(got, rows, cols, step, x, y, w, h, *color, thickness)
        assert np.array_equal(got, want), (case, rows, cols, step, x, y, w, h, thickness, cap)
    # a 200 x 300 Mat, the config-1 rectangle and a thickness beyond the rectangle's size
    base = rng.integers(0, 256, size=200 * 900, dtype=np.uint8)
    for (x, y, w, h, t) in ((200, 150, 240, 240, 2), (10, 20, 30, 5, 40), (-5, -5, 320, 220, 7), (290, 190, 50, 50, 30)):
        want, got = base.copy(), base.copy()
        npref.rectangle_rs(want, 200, 300, 900, x, y, w, h, (0, 255, 0), t)
        oracle.rectangle(got, 200, 300, 900, x, y, w, h, 0, 255, 0, t)
        assert np.array_equal(got, want), (x, y, w, h, t)
