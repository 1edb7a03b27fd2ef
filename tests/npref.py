"""Independent second implementations of the build-defined ops, written from the SPECIFICATION and, for
`rectangle`, as a direct transliteration of the reference's Rust (rustcv/src/imgproc/drawing.rs:67-106) -- NOT from
oracle/rcv_oracle.c.  tests/test_oracle_independent.py holds the C oracle against them on >= 200 x 300 images.

Where the spec fixes the evaluation order (GaussianBlur sigma > 0: an fmaf chain per pass; cornerHarris: six separate f32 ops)
the implementation below is bit-exact f32: numpy float32 arrays round after every operation, and fmaf(a, b, c) is formed in
80-bit long double (the product of two f32 is exact there and the sum keeps 64 significant bits) and rounded once to f32.
Where the spec leaves the coordinate arithmetic to the implementation (resize, warpAffine) the reference value is computed in
float64 and the comparison allows the north star's tolerance (1 LSB where the exact value sits next to a rounding boundary).
"""
import numpy as np

LD = np.longdouble
F32 = np.float32


def fmaf(a, b, c):
    """IEEE fused multiply-add on f32 operands (arrays or scalars), one rounding"""
    return (np.asarray(a, F32).astype(LD) * np.asarray(b, F32).astype(LD) + np.asarray(c, F32).astype(LD)).astype(F32)


def reflect101(i, n):
    """gfedcb|abcdefgh|gfedcba ; requires |offset| < n"""
    i = np.asarray(i)
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * n - 2 - i, i)


def _shift2(img, dy, dx):
    """img[reflect(y + dy), reflect(x + dx)] for every (y, x)"""
    h, w = img.shape[:2]
    ys = reflect101(np.arange(h) + dy, h)
    xs = reflect101(np.arange(w) + dx, w)
    return img[ys][:, xs]


# ---- GaussianBlur, sigma > 0 (spec: f32 taps from an f64 normalisation, horizontal then vertical fmaf chains, rintf) ----
def gaussian_taps(ksize, sigma):
    r = ksize // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    t = np.exp(-(x * x) / (2.0 * float(sigma) * float(sigma)))
    return (t / t.sum()).astype(F32)


def gaussian_blur_sigma(img, ksize, sigma):
    taps = gaussian_taps(ksize, sigma)
    r = ksize // 2
    src = img.astype(F32)
    tmp = np.zeros_like(src)
    for i in range(ksize):                      # horizontal pass, tap order, from acc = 0
        tmp = fmaf(taps[i], _shift2(src, 0, i - r), tmp)
    out = np.zeros_like(src)
    for i in range(ksize):                      # vertical pass on the f32 intermediate
        out = fmaf(taps[i], _shift2(tmp, i - r, 0), out)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)     # np.rint: ties to even, like rintf


# ---- Sobel 3x3 and cornerHarris (spec: exact integer gradients and box sums, then six separate f32 operations) ----
def sobel(gray):
    g = gray.astype(np.int32)
    sm_v = _shift2(g, -1, 0) + 2 * g + _shift2(g, 1, 0)       # [1 2 1]^T
    sm_h = _shift2(g, 0, -1) + 2 * g + _shift2(g, 0, 1)       # [1 2 1]
    dx = _shift2(sm_v, 0, 1) - _shift2(sm_v, 0, -1)
    dy = _shift2(sm_h, 1, 0) - _shift2(sm_h, -1, 0)
    return dx, dy


def corner_harris(gray, block, k, aperture=3):
    dx, dy = sobel(gray)
    anchor = block // 2

    def box(a):
        s = np.zeros_like(a)
        for oy in range(block):
            for ox in range(block):
                s = s + _shift2(a, oy - anchor, ox - anchor)
        return s
    sxx, sxy, syy = box(dx * dx), box(dx * dy), box(dy * dy)
    s = 1.0 / (2.0 ** (aperture - 1) * block * 255.0)          # f64
    s2 = F32(s * s)                                            # squared in f64, cast once
    a, b, c = sxx.astype(F32) * s2, sxy.astype(F32) * s2, syy.astype(F32) * s2
    t1 = a * c
    t2 = b * b
    t3 = a + c
    t4 = F32(k) * t3
    t5 = t4 * t3
    return (t1 - t2) - t5


def nms3x3(resp, thr):
    h, w = resp.shape
    p = np.full((h + 2, w + 2), -np.inf, np.float32)
    p[1:-1, 1:-1] = resp
    ok = resp > F32(thr)
    for dy in range(3):
        for dx in range(3):
            if dy == 1 and dx == 1:
                continue
            ok &= resp >= p[dy:dy + h, dx:dx + w]
    return np.where(ok, 255, 0).astype(np.uint8)


# ---- resize / warpAffine: float64 evaluation of the stated sampling rule ----
def resize_f64(img, drows, dcols):
    """bilinear, half-pixel centres, source coordinate clamped to the image, taps clamped to the last row / column; returns the
    UNROUNDED float64 value per output sample"""
    h, w = img.shape[:2]
    src = img.astype(np.float64).reshape(h, w, -1)

    def axis(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        s = np.clip(s, 0.0, n_in - 1.0)
        i0 = np.floor(s).astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, s - i0
    y0, y1, fy = axis(drows, h)
    x0, x1, fx = axis(dcols, w)
    fx = fx[None, :, None]
    top = src[y0][:, x0] + fx * (src[y0][:, x1] - src[y0][:, x0])
    bot = src[y1][:, x0] + fx * (src[y1][:, x1] - src[y1][:, x0])
    return top + fy[:, None, None] * (bot - top)


def warp_affine_f64(img, M, drows, dcols):
    """bilinear, M maps dst -> src, constant border 0 (a tap outside the source contributes 0); UNROUNDED float64 values.
    The matrix entries are taken as the f32 values the caller passes."""
    h, w = img.shape[:2]
    src = img.astype(np.float64).reshape(h, w, -1)
    m = np.asarray(M, F32).astype(np.float64)
    xs, ys = np.meshgrid(np.arange(dcols, dtype=np.float64), np.arange(drows, dtype=np.float64))
    sx = m[0] * xs + m[1] * ys + m[2]
    sy = m[3] * xs + m[4] * ys + m[5]
    x0, y0 = np.floor(sx), np.floor(sy)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok[..., None], v, 0.0)
    p00, p01, p10, p11 = tap(y0, x0), tap(y0, x0 + 1), tap(y0 + 1, x0), tap(y0 + 1, x0 + 1)
    top = p00 + fx * (p01 - p00)
    bot = p10 + fx * (p11 - p10)
    return top + fy * (bot - top)


def round_half_up_u8(v):
    return np.clip(np.floor(v + 0.5), 0, 255).astype(np.uint8)


# ---- rectangle: transliteration of rustcv/src/imgproc/drawing.rs:67-106 (release-mode wrapping arithmetic) ----
_M64 = (1 << 64) - 1


def rectangle_rs(data, rows, cols, step, x, y, w, h, color, thickness):
    """data: mutable flat uint8 array (Vec<u8>); color = (v0, v1, v2).  `r as usize` / `c as usize` of a negative i32 sign-extends
    to 2^64 - |r|, and usize multiplication / addition wrap modulo 2^64 in a release build -- which is how a thickness larger
    than the clipped rectangle lands on in-buffer pixels of OTHER rows (drawing.rs:81-82)."""
    x_min, y_min = max(x, 0), max(y, 0)
    x_max, y_max = min(x + w, cols), min(y + h, rows)
    if x_min >= x_max or y_min >= y_max:
        return
    n = len(data)

    def set_pixel(r, c):
        idx = (((r & _M64) * step) + ((c & _M64) * 3)) & _M64
        if idx + 2 < n:                         # (idx + 2 cannot wrap: idx < 2^64 - 2 whenever the test can pass)
            data[idx], data[idx + 1], data[idx + 2] = color

    for c in range(x_min, x_max):
        for t in range(max(thickness, 0)):
            set_pixel(y_min + t, c)
            set_pixel(y_max - 1 - t, c)
    for r in range(y_min, y_max):
        for t in range(max(thickness, 0)):
            set_pixel(r, x_min + t)
            set_pixel(r, x_max - 1 - t)
