"""CPU tests of the oracle (no GPU): pin it against everything the reference's own tests hold for this
path, against hand-derived known answers, against scipy's exact integer arithmetic for the build-defined
ops, and against the committed golden fixtures."""
import json
import os

import numpy as np
import pytest
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(GOLD, "kat_reference.json")))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "ops_small.npz"))


# ---- the reference's own tests, replayed on the restatement (rustcv-camera/src/decode.rs:234-273) ----

def test_reference_yuyv_to_bgr_basic(oracle, kat):
    t = kat["reference_tests"][0]
    d = np.zeros(6, np.uint8)
    assert oracle.yuyv_to_bgr(np.array(t["yuyv"], np.uint8), d, t["w"], t["h"], 1)
    assert (d > 240).all()


def test_reference_yuyv_to_bgr_black(oracle, kat):
    t = kat["reference_tests"][1]
    d = np.zeros(6, np.uint8)
    assert oracle.yuyv_to_bgr(np.array(t["yuyv"], np.uint8), d, t["w"], t["h"], 1)
    assert (d < 10).all()


def test_reference_rgb_to_bgr_swap(oracle, kat):
    t = kat["reference_tests"][2]
    d = np.zeros(6, np.uint8)
    oracle.rgb_to_bgr(np.array(t["rgb"], np.uint8), d)
    assert d.tolist() == t["bgr"]


def test_yuv_hand_derived_kats(oracle, kat):
    d = np.zeros(6, np.uint8)
    for (y, u, v), bgr in kat["yuv_to_bgr_hand_derived"]:
        for variant in (0, 1):
            assert oracle.yuyv_to_bgr(np.array([y, u, y, v], np.uint8), d, 2, 1, variant)
            assert d[:3].tolist() == bgr and d[3:].tolist() == bgr


def test_yuyv_matches_formula_exhaustively_in_numpy(oracle):
    """All 2^24 triples against a vectorised numpy restatement of mod.rs:356-363 (independent of the C code)."""
    y, u, v = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    y, u, v = y.reshape(-1).astype(np.int32), u.reshape(-1).astype(np.int32) - 128, v.reshape(-1).astype(np.int32) - 128
    c = 298 * (y - 16) + 128
    want = np.stack([np.clip((c + 516 * u) >> 8, 0, 255), np.clip((c - 100 * u - 208 * v) >> 8, 0, 255),
                     np.clip((c + 409 * v) >> 8, 0, 255)], -1).astype(np.uint8)
    src = np.stack([y, u + 128, y, v + 128], -1).astype(np.uint8).reshape(-1)
    got = np.zeros(src.size // 4 * 6, np.uint8)
    assert oracle.yuyv_to_bgr(src, got, 4096, 8192)
    got = got.reshape(-1, 2, 3)
    assert np.array_equal(got[:, 0], want) and np.array_equal(got[:, 1], want)


def test_yuyv_guards(oracle):
    src, dst = np.zeros(16, np.uint8), np.full(24, 7, np.uint8)
    assert not oracle.yuyv_to_bgr(src[:15], dst, 4, 2, 0) and (dst == 7).all()      # facade: short src -> silent return
    assert not oracle.yuyv_to_bgr(src, dst[:23], 4, 2, 1) and (dst == 7).all()      # twin: short dst -> silent return
    d3 = np.full(9, 7, np.uint8)
    assert oracle.yuyv_to_bgr(np.zeros(6, np.uint8), d3, 3, 1, 0)                   # odd w*h: last pixel untouched
    assert d3[6:].tolist() == [7, 7, 7] and d3[:6].tolist() == [0, 135, 0, 0, 135, 0]


def test_bgra_guards_and_zip(oracle):
    src = np.arange(16, dtype=np.uint8)
    dst = np.full(12, 9, np.uint8)
    assert oracle.bgra_to_bgr(src, dst, 4, 1, 0) and dst.tolist() == [0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14]
    dst[:] = 9
    assert not oracle.bgra_to_bgr(src[:15], dst, 4, 1, 0) and (dst == 9).all()      # facade guard (mod.rs:388-390)
    assert oracle.bgra_to_bgr(src[:15], dst, 0, 0, 1) and dst[:9].tolist() == [0, 1, 2, 4, 5, 6, 8, 9, 10] and dst[9] == 9  # twin zips


def test_rectangle_semantics(oracle):
    rows, cols, step = 12, 16, 48
    img = np.zeros(rows * step, np.uint8)
    oracle.rectangle(img, rows, cols, step, 2, 3, 8, 6, 10, 20, 30, 2)
    a = img.reshape(rows, cols, 3)
    inside = np.zeros((rows, cols), bool)
    inside[3:9, 2:10] = True
    inner = np.zeros((rows, cols), bool)
    inner[5:7, 4:8] = True          # border grows INWARD by `thickness` (drawing.rs:92-105), not centred
    ring = inside & ~inner
    assert (a[ring] == [10, 20, 30]).all() and (a[~ring] == 0).all()
    img2 = np.zeros(rows * step, np.uint8)
    oracle.rectangle(img2, rows, cols, step, -5, -5, 100, 100, 1, 1, 1, 1)          # clipped to the Mat
    b = img2.reshape(rows, cols, 3)
    assert b[0].all() and b[-1].all() and b[:, 0].all() and b[:, -1].all() and not b[1:-1, 1:-1].any()
    img3 = np.zeros(rows * step, np.uint8)
    oracle.rectangle(img3, rows, cols, step, 20, 3, 5, 5, 1, 1, 1, 1)               # fully outside -> untouched
    assert not img3.any()


# ---- build-defined ops against scipy integer arithmetic (BORDER_REFLECT_101 == mode='mirror') ----------

def _corr(img, k):
    img = img.astype(np.int64)
    if img.ndim == 2:
        return ndimage.correlate(img, k.astype(np.int64), mode="mirror")
    return np.stack([ndimage.correlate(img[:, :, c], k.astype(np.int64), mode="mirror") for c in range(img.shape[2])], -1)


def test_reflect101_is_scipy_mirror(oracle):
    assert ndimage.correlate(np.array([1, 2, 3, 4, 5]), np.array([1, 0, 0]), mode="mirror").tolist() == [2, 1, 2, 3, 4]
    L = oracle.lib()
    assert [L.orc_reflect101(i, 5) for i in range(-4, 9)] == [4, 3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1, 0]
    assert [L.orc_reflect101(i, 1) for i in (-3, 0, 7)] == [0, 0, 0]
    assert [L.orc_reflect101(i, 2) for i in (-3, -2, -1, 0, 1, 2, 3, 4)] == [1, 0, 1, 0, 1, 0, 1, 0]


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (13, 17), (40, 31)])
@pytest.mark.parametrize("ch", [1, 3])
def test_gaussian_int_vs_scipy(oracle, rng, shape, ch):
    img = rng.integers(0, 256, size=shape + ((ch,) if ch > 1 else ()), dtype=np.uint8)
    for ks, taps, D in ((3, [1, 2, 1], 16), (5, [1, 4, 6, 4, 1], 256), (7, [2, 7, 14, 18, 14, 7, 2], 4096)):
        k = np.outer(taps, taps)
        assert k.sum() == D
        want = ((_corr(img, k) + D // 2) // D).astype(np.uint8)
        assert np.array_equal(oracle.gaussian_blur(img, ks, 0.0), want)


@pytest.mark.parametrize("ksize,shift", [(1, 0), (3, 2), (5, 7), (7, 6), (7, 0)])
def test_filter2d_i8_vs_scipy(oracle, rng, ksize, shift):
    img = rng.integers(0, 256, size=(23, 37, 3), dtype=np.uint8)
    k = rng.integers(-128, 128, size=(ksize, ksize), dtype=np.int8)
    acc = _corr(img, k) + ((1 << (shift - 1)) if shift else 0)
    want = np.clip(acc >> shift, 0, 255).astype(np.uint8)  # numpy >> on negative int64 is arithmetic (floor)
    assert np.array_equal(oracle.filter2d_i8(img, k, shift), want)


def test_sobel_vs_scipy(oracle, rng):
    img = rng.integers(0, 256, size=(31, 45), dtype=np.uint8)
    dx, dy = oracle.sobel(img)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    assert np.array_equal(dx, _corr(img, kx)) and np.array_equal(dy, _corr(img, kx.T))


def test_bgr2gray_formula(oracle, rng):
    img = rng.integers(0, 256, size=(9, 11, 3), dtype=np.uint8).astype(np.int64)
    want = (1868 * img[:, :, 0] + 9617 * img[:, :, 1] + 4899 * img[:, :, 2] + 8192) >> 14
    assert np.array_equal(oracle.bgr2gray(img.astype(np.uint8)), want.astype(np.uint8))
    assert oracle.bgr2gray(np.full((1, 1, 3), 255, np.uint8))[0, 0] == 255


def test_filter2d_f32_is_the_stated_fmaf_chain(oracle, rng):
    img = rng.integers(0, 256, size=(7, 9, 1), dtype=np.uint8)[:, :, 0]
    k = rng.standard_normal((3, 3)).astype(np.float32)
    pad = np.pad(img, 1, mode="reflect")
    want = np.zeros_like(img)
    for y in range(7):
        for x in range(9):
            acc = np.float32(0.25)
            for ky in range(3):
                for kx in range(3):  # one rounding per step: float64 product+sum of f32 operands, rounded to f32 == fmaf
                    acc = np.float32(np.float64(k[ky, kx]) * np.float64(pad[y + ky, x + kx]) + np.float64(acc))
            want[y, x] = np.clip(np.rint(acc), 0, 255)
    assert np.array_equal(oracle.filter2d_f32(img, k, 0.25), want)


def test_resize_exact_4x_is_box_of_centre_2x2(oracle, rng):
    img = rng.integers(0, 256, size=(32, 48, 3), dtype=np.uint8)
    a = img.astype(np.int32).reshape(8, 4, 12, 4, 3)
    want = (a[:, 1, :, 1] + a[:, 1, :, 2] + a[:, 2, :, 1] + a[:, 2, :, 2] + 2) >> 2
    assert np.array_equal(oracle.resize(img, 8, 12), want.astype(np.uint8))
    assert np.array_equal(oracle.resize(img, 32, 48), img)  # identity scale


def test_warp_affine_identity_and_shift(oracle, rng):
    img = rng.integers(0, 256, size=(10, 12, 3), dtype=np.uint8)
    assert np.array_equal(oracle.warp_affine(img, [1, 0, 0, 0, 1, 0], 10, 12), img)
    sh = oracle.warp_affine(img, [1, 0, 2, 0, 1, -1], 10, 12)  # dst(x,y) = src(x+2, y-1), zeros outside
    want = np.zeros_like(img)
    want[1:, :10] = img[:9, 2:]
    assert np.array_equal(sh, want)
    half = oracle.warp_affine(img, [1, 0, 0.5, 0, 1, 0], 10, 12)  # half-pixel: round-half-up of the 2-tap mean
    nxt = np.concatenate([img[:, 1:], np.zeros((10, 1, 3), np.uint8)], 1).astype(np.int32)
    assert np.array_equal(half, ((img.astype(np.int32) + nxt + 1) >> 1).astype(np.uint8))


def test_harris_response_formula_and_nms(oracle, rng):
    g = rng.integers(0, 256, size=(15, 19), dtype=np.uint8)
    dx, dy = oracle.sobel(g)
    box = np.ones((2, 2), np.int64)
    def bsum(a):  # blockSize 2, anchor 1 -> window offsets -1..0 == scipy origin... use explicit shifts
        p = np.pad(a, ((1, 0), (1, 0)), mode="reflect")
        return p[:-1, :-1] + p[:-1, 1:] + p[1:, :-1] + p[1:, 1:]
    sxx, sxy, syy = bsum(dx.astype(np.int64) ** 2), bsum(dx.astype(np.int64) * dy), bsum(dy.astype(np.int64) ** 2)
    s2 = np.float32((1.0 / (4.0 * 2 * 255.0)) ** 2)
    fa, fb, fc = sxx.astype(np.float32) * s2, sxy.astype(np.float32) * s2, syy.astype(np.float32) * s2
    t3 = fa + fc
    want = (fa * fc - fb * fb) - (np.float32(0.04) * t3) * t3
    got = oracle.corner_harris(g, 2, 0.04)
    assert np.array_equal(got.view(np.uint32), want.astype(np.float32).view(np.uint32))
    m = oracle.nms3x3(got, 0.0)
    pad = np.pad(got, 1, constant_values=-np.inf)
    nb = np.stack([pad[1 + dy:16 + dy, 1 + dx:20 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)])
    assert np.array_equal(m, np.where((got > 0.0) & (got >= nb).all(0), 255, 0).astype(np.uint8))


def test_synth_is_counter_based(oracle):
    a = oracle.synth_frame(16, 24, 3, 0, 0x5EED0003, 7)
    b = oracle.synth_frame(32, 24, 3, 0, 0x5EED0003, 7)
    assert np.array_equal(a, b[:16])                      # independent of image height
    assert oracle.splitmix64(0) == 0xE220A8397B1DCDAF     # published splitmix64 first output for seed 0
    k = oracle.bench_kernel7()
    assert k.shape == (7, 7) and k.min() >= -8 and k.max() <= 8


# ---- golden fixtures ---------------------------------------------------------------------------------------

def test_oracle_matches_golden(oracle, gold):
    bgr, gray = gold["bgr"], gold["gray"]
    out = np.zeros(24 * 10 * 3, np.uint8)
    oracle.yuyv_to_bgr(gold["yuyv"], out, 24, 10)
    assert np.array_equal(out, gold["yuyv_bgr"])
    out = np.zeros(21 * 3, np.uint8)
    oracle.bgra_to_bgr(gold["bgra"], out, 21, 1)
    assert np.array_equal(out, gold["bgra_bgr"])
    r = bgr.copy().reshape(-1)
    oracle.rectangle(r, 37, 48, 144, 5, 4, 30, 20, 0, 255, 0, 2)
    assert np.array_equal(r, gold["rect_5_4_30_20_t2"])
    assert np.array_equal(oracle.bgr2gray(bgr), gold["bgr2gray"])
    for ks in (3, 5, 7):
        assert np.array_equal(oracle.gaussian_blur(bgr, ks, 0.0), gold[f"gauss{ks}"])
    assert np.array_equal(oracle.gaussian_blur(bgr, 5, 1.2), gold["gauss5_s1p2"])
    assert np.array_equal(oracle.bench_kernel7(), gold["k7"])
    assert np.array_equal(oracle.filter2d_i8(bgr, gold["k7"], 6), gold["filter7_s6"])
    assert np.array_equal(oracle.filter2d_f32(bgr, gold["kf3"], 0.25), gold["filter3_f32"])
    dx, dy = oracle.sobel(gray)
    assert np.array_equal(dx, gold["sobel_dx"]) and np.array_equal(dy, gold["sobel_dy"])
    assert np.array_equal(oracle.resize(bgr, 9, 12), gold["resize_9x12"])
    assert np.array_equal(oracle.resize(bgr, 50, 70), gold["resize_50x70"])
    assert np.array_equal(oracle.warp_affine(bgr, gold["warp_M"], 37, 48), gold["warp"])
    assert np.array_equal(oracle.corner_harris(gray, 2, 0.04).view(np.uint32), gold["harris_b2"].view(np.uint32))
    assert np.array_equal(oracle.nms3x3(gold["harris_b2"], 1e-4), gold["nms"])
    assert np.array_equal(oracle.warp_affine_f32(gold["harris_b2"], gold["warp_M"], 29, 41).view(np.uint32), gold["warp_f32"].view(np.uint32))
    assert np.array_equal(oracle.resize_f32(gold["harris_b2"], 17, 23).view(np.uint32), gold["resize_f32_17x23"].view(np.uint32))
    assert np.array_equal(oracle.synth_frame(24, 40, 3, 1, 0x5EED0003, 2), gold["synth_scene"])
    assert np.array_equal(oracle.synth_frame(8, 8, 3, 0, 0x5EED0003, 0), gold["synth_noise"])


# ---- "next" rows f2 / f4 -------------------------------------------------------------------------------

def test_next_row_restatements(oracle, rng):
    # mat_to_u32_buffer: (r << 16) | (g << 8) | b, flat, zero tail (highgui/mod.rs:125-141)
    assert oracle.bgr_to_u32(np.array([1, 2, 3, 4, 5, 6, 7], np.uint8), 3).tolist() == [0x030201, 0x060504, 0]
    # imwrite swizzle honours step (imgcodecs/mod.rs:51-63)
    assert oracle.bgr_to_rgb_rows(np.array([1, 2, 3, 9, 4, 5, 6, 9], np.uint8), 4, 2, 1).tolist() == [3, 2, 1, 6, 5, 4]
    # strided YUYV == the flat reference conversion when the rows are packed
    yuyv = rng.integers(0, 256, size=8 * 6 * 2, dtype=np.uint8)
    a, b = np.zeros(8 * 6 * 3, np.uint8), np.zeros(8 * 6 * 3, np.uint8)
    oracle.yuyv_to_bgr(yuyv, a, 8, 6)
    oracle.yuv422_to_bgr_strided(yuyv, 16, 6, 8, False, b)
    assert np.array_equal(a, b)
    uyvy = yuyv.reshape(-1, 2)[:, ::-1].reshape(-1)
    oracle.yuv422_to_bgr_strided(uyvy, 16, 6, 8, True, b)
    assert np.array_equal(a, b)
    # NV12 with constant chroma == YUYV with the same chroma
    nv = np.concatenate([yuyv.reshape(6, 8, 2)[:, :, 0].reshape(-1), np.full(8 * 3, 128, np.uint8)])
    y2 = yuyv.copy().reshape(-1, 2)
    y2[:, 1] = 128
    oracle.yuyv_to_bgr(y2.reshape(-1), a, 8, 6)
    assert oracle.nv12_to_bgr(nv, 8, 6, 8, b) and np.array_equal(a, b)


# ---- put_text's blend closure (drawing.rs:137-160) -----------------------------------------------------

def _np_blend(px, colour, alpha):
    """the formula in numpy float32 scalars: every operation rounds separately, like the Rust source"""
    f = np.float32
    v = f(colour) * f(alpha) + f(px) * (f(1.0) - f(alpha))
    return 0 if not v > 0 else (255 if v >= 255 else int(v))


def test_blend_hand_derived_kats(oracle, kat):
    for old, colour, alpha, new in kat["blend_hand_derived"]:
        img = np.full(3, old, np.uint8)
        oracle.blend_glyphs(img, 1, 1, 3, [(0, 0, np.array([[alpha]], np.float32))], colour, colour, colour)
        assert img.tolist() == [new] * 3, (old, colour, alpha)
        assert _np_blend(old, colour, alpha) == new


def test_blend_order_clipping_and_special_values(oracle, rng):
    rows, cols, step = 9, 11, 40
    base = rng.integers(0, 256, size=rows * step, dtype=np.uint8)
    glyphs = [(-2, -1, rng.random((4, 5), dtype=np.float32)), (1, 0, rng.random((6, 4), dtype=np.float32)),     # overlap
              (8, 6, rng.random((5, 7), dtype=np.float32)), (20, 2, np.ones((2, 2), np.float32)),                # clipped / outside
              (3, 3, np.array([[np.nan, -0.5, 1.5, 1e-40, np.inf]], np.float32)), (0, 8, np.zeros((1, 11), np.float32))]
    got = base.copy()
    oracle.blend_glyphs(got, rows, cols, step, glyphs, 250, 3, 128)
    want = base.copy()
    with np.errstate(all="ignore"):
        for gx, gy, cov in glyphs:                      # glyph after glyph: later boxes read what earlier ones stored
            for y in range(cov.shape[0]):
                for x in range(cov.shape[1]):
                    px, py = gx + x, gy + y
                    if 0 <= px < cols and 0 <= py < rows:
                        for c, col in enumerate((250, 3, 128)):
                            i = py * step + px * 3 + c
                            want[i] = _np_blend(want[i], col, cov[y, x])
    assert np.array_equal(got, want)
    pad = np.ones(rows * step, bool)
    pad.reshape(rows, step)[:, : cols * 3] = False
    assert np.array_equal(got[pad], base[pad])           # row padding untouched
