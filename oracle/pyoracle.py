"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle/rcv_oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Arrays in,
arrays out; every function is a thin shim over the C restatement in rcv_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RCV_ORACLE_LIB: another build of the same source -- `make -C oracle asan-test` runs the CPU tests against the ASan / UBSan build)
LIB_PATH = os.environ.get("RCV_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
_lib = None

_u8p, _i8p, _i16p, _f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int8), C.POINTER(C.c_int16), C.POINTER(C.c_float)


def build(force=False):
    src = os.path.join(_HERE, "rcv_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_splitmix64.restype = C.c_uint64
        _lib.orc_splitmix64.argtypes = [C.c_uint64]
        _lib.orc_reflect101.restype = C.c_int
        _lib.orc_threads.restype = C.c_int
        _lib.orc_synth_frame.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64]
        _lib.orc_synth_yuyv.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, C.c_uint64, C.c_uint64]
        _lib.orc_yuyv_to_bgr.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
        _lib.orc_bgra_to_bgr.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
        _lib.orc_rgb_to_bgr.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t]
        _lib.orc_rectangle.argtypes = [_u8p, C.c_size_t, C.c_int32, C.c_int32, C.c_size_t] + [C.c_int32] * 4 + [C.c_uint8] * 3 + [C.c_int32]
        _lib.orc_bgr2gray.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int]
        _lib.orc_gaussian_blur.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        _lib.orc_gaussian_taps_f32.argtypes = [C.c_int, C.c_double, _f32p]
        _lib.orc_filter2d_i8.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, _i8p, C.c_int, C.c_int]
        _lib.orc_filter2d_f32.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_float]
        _lib.orc_sobel.argtypes = [_u8p, C.c_size_t, _i16p, C.c_size_t, _i16p, C.c_size_t, C.c_int, C.c_int]
        _lib.orc_resize.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        _lib.orc_warp_affine.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, _f32p]
        _lib.orc_resize_f32.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_int, _f32p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        _lib.orc_warp_affine_f32.argtypes = [_f32p, C.c_size_t, C.c_int, C.c_int, _f32p, C.c_size_t, C.c_int, C.c_int, C.c_int, _f32p]
        _lib.orc_corner_harris.argtypes = [_u8p, C.c_size_t, _f32p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_float]
        _lib.orc_nms3x3.argtypes = [_f32p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_float]
        _lib.orc_harris_pipeline.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, _f32p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
        _lib.orc_bench_kernel7.argtypes = [_i8p]
        _lib.orc_bgr_to_u32.argtypes = [_u8p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t]
        _lib.orc_bgr_to_rgb_rows.argtypes = [_u8p, C.c_size_t, _u8p, C.c_int, C.c_int]
        _lib.orc_yuv422_to_bgr_strided.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        _lib.orc_nv12_to_bgr.argtypes = [_u8p, C.c_size_t, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int]
        _lib.orc_blend_glyph.restype = None
        _lib.orc_blend_glyph.argtypes = [_u8p, C.c_int32, C.c_int32, C.c_size_t] + [C.c_int32] * 4 + [_f32p] + [C.c_uint8] * 3
    return _lib


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (the GPU boxes report
    256 logical CPUs but run under a 16-CPU quota; 256 OpenMP threads there are 6x slower than 16)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, math.ceil(int(quota) / int(period))))
        except Exception:
            pass
    try:   # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, math.ceil(q / p)))
    except Exception:
        pass
    return max(1, n)


def set_threads(n):
    lib().orc_set_threads(int(n))
    return lib().orc_threads()


def _p(a, t):
    return a.ctypes.data_as(t)


def _img(a):
    """HxW or HxWxC uint8, C-contiguous -> (array, step, rows, cols, ch)"""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    r, c, ch = a.shape
    return a, c * ch, r, c, ch


def _out_like(rows, cols, ch, dtype=np.uint8):
    return np.zeros((rows, cols, ch) if ch > 1 else (rows, cols), dtype=dtype)


# ---- (A) reference restatements ---------------------------------------------------------------

def yuyv_to_bgr(src, dst, width, height, variant=0):
    """src, dst: flat uint8 arrays; dst written in place.  Returns True if the conversion ran."""
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(-1)
    assert dst.dtype == np.uint8 and dst.flags.c_contiguous
    return bool(lib().orc_yuyv_to_bgr(_p(src, _u8p), src.size, _p(dst, _u8p), dst.size, width, height, variant))


def bgra_to_bgr(src, dst, width, height, variant=0):
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(-1)
    assert dst.dtype == np.uint8 and dst.flags.c_contiguous
    return bool(lib().orc_bgra_to_bgr(_p(src, _u8p), src.size, _p(dst, _u8p), dst.size, width, height, variant))


def rgb_to_bgr(src, dst):
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(-1)
    assert dst.dtype == np.uint8 and dst.flags.c_contiguous
    lib().orc_rgb_to_bgr(_p(src, _u8p), src.size, _p(dst, _u8p), dst.size)


def rectangle(data, rows, cols, step, x, y, w, h, b, g, r, thickness):
    """data: flat uint8 (Vec<u8>), modified in place"""
    assert data.dtype == np.uint8 and data.flags.c_contiguous
    lib().orc_rectangle(_p(data, _u8p), data.size, rows, cols, step, x, y, w, h, b, g, r, thickness)


def bgr_to_u32(data, pixel_count):
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    out = np.zeros(pixel_count, np.uint32)
    lib().orc_bgr_to_u32(_p(data, _u8p), data.size, out.ctypes.data_as(C.POINTER(C.c_uint32)), pixel_count)
    return out


def bgr_to_rgb_rows(data, step, rows, cols):
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    out = np.zeros(rows * cols * 3, np.uint8)
    lib().orc_bgr_to_rgb_rows(_p(data, _u8p), step, _p(out, _u8p), rows, cols)
    return out


def yuv422_to_bgr_strided(data, sstep, rows, cols, uyvy, dst):
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    lib().orc_yuv422_to_bgr_strided(_p(data, _u8p), sstep, _p(dst, _u8p), cols * 3, rows, cols, int(uyvy))


def nv12_to_bgr(data, sstep, rows, cols, dst):
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    return bool(lib().orc_nv12_to_bgr(_p(data, _u8p), data.size, sstep, _p(dst, _u8p), cols * 3, rows, cols))


def blend_glyphs(data, rows, cols, step, glyphs, b, g, r):
    """put_text's per-pixel half (drawing.rs:137-160).  data: flat uint8 Mat buffer, modified in place; glyphs: iterable
    of (min_x, min_y, coverage[h, w] float32) in drawing order."""
    assert data.dtype == np.uint8 and data.flags.c_contiguous
    assert rows == 0 or cols == 0 or data.size >= (rows - 1) * step + cols * 3
    for gx, gy, cov in glyphs:
        cov = np.ascontiguousarray(cov, dtype=np.float32)
        h, w = cov.shape
        if h and w:
            lib().orc_blend_glyph(_p(data, _u8p), rows, cols, step, gx, gy, w, h, _p(cov, _f32p), b, g, r)


# ---- (B) build-defined ops ----------------------------------------------------------------------

def bgr2gray(bgr):
    a, st, r, c, ch = _img(bgr)
    assert ch == 3
    out = np.zeros((r, c), np.uint8)
    lib().orc_bgr2gray(_p(a, _u8p), st, _p(out, _u8p), c, r, c)
    return out


def gaussian_blur(img, ksize, sigma=0.0):
    a, st, r, c, ch = _img(img)
    out = _out_like(r, c, ch)
    rc = lib().orc_gaussian_blur(_p(a, _u8p), st, _p(out, _u8p), st, r, c, ch, ksize, float(sigma))
    assert rc == 0, rc
    return out


def gaussian_taps_f32(ksize, sigma):
    t = np.zeros(ksize, np.float32)
    lib().orc_gaussian_taps_f32(ksize, float(sigma), _p(t, _f32p))
    return t


def filter2d_i8(img, k, shift):
    a, st, r, c, ch = _img(img)
    k = np.ascontiguousarray(k, dtype=np.int8)
    out = _out_like(r, c, ch)
    rc = lib().orc_filter2d_i8(_p(a, _u8p), st, _p(out, _u8p), st, r, c, ch, _p(k, _i8p), k.shape[0], shift)
    assert rc == 0, rc
    return out


def filter2d_f32(img, k, delta=0.0):
    a, st, r, c, ch = _img(img)
    k = np.ascontiguousarray(k, dtype=np.float32)
    out = _out_like(r, c, ch)
    rc = lib().orc_filter2d_f32(_p(a, _u8p), st, _p(out, _u8p), st, r, c, ch, _p(k, _f32p), k.shape[0], float(delta))
    assert rc == 0, rc
    return out


def sobel(gray):
    a, st, r, c, ch = _img(gray)
    assert ch == 1
    dx, dy = np.zeros((r, c), np.int16), np.zeros((r, c), np.int16)
    lib().orc_sobel(_p(a, _u8p), st, _p(dx, _i16p), c * 2, _p(dy, _i16p), c * 2, r, c)
    return dx, dy


def resize(img, drows, dcols):
    a, st, r, c, ch = _img(img)
    out = _out_like(drows, dcols, ch)
    lib().orc_resize(_p(a, _u8p), st, r, c, _p(out, _u8p), dcols * ch, drows, dcols, ch)
    return out


def warp_affine(img, M, drows, dcols):
    a, st, r, c, ch = _img(img)
    m = np.ascontiguousarray(M, dtype=np.float32).reshape(6)
    out = _out_like(drows, dcols, ch)
    lib().orc_warp_affine(_p(a, _u8p), st, r, c, _p(out, _u8p), dcols * ch, drows, dcols, ch, _p(m, _f32p))
    return out


def _img_f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 2:
        a = a[:, :, None]
    r, c, ch = a.shape
    return a, c * ch * 4, r, c, ch


def resize_f32(img, drows, dcols):
    """RCV_32F resize: the unrounded interpolated value (same f32 operations, same order as `resize`)"""
    a, st, r, c, ch = _img_f32(img)
    out = _out_like(drows, dcols, ch, np.float32)
    lib().orc_resize_f32(_p(a, _f32p), st, r, c, _p(out, _f32p), dcols * ch * 4, drows, dcols, ch)
    return out


def warp_affine_f32(img, M, drows, dcols):
    a, st, r, c, ch = _img_f32(img)
    m = np.ascontiguousarray(M, dtype=np.float32).reshape(6)
    out = _out_like(drows, dcols, ch, np.float32)
    lib().orc_warp_affine_f32(_p(a, _f32p), st, r, c, _p(out, _f32p), dcols * ch * 4, drows, dcols, ch, _p(m, _f32p))
    return out


def corner_harris(gray, block=2, k=0.04):
    a, st, r, c, ch = _img(gray)
    assert ch == 1
    out = np.zeros((r, c), np.float32)
    rc = lib().orc_corner_harris(_p(a, _u8p), st, _p(out, _f32p), c * 4, r, c, block, float(k))
    assert rc == 0, rc
    return out


def nms3x3(resp, thr):
    resp = np.ascontiguousarray(resp, dtype=np.float32)
    r, c = resp.shape
    out = np.zeros((r, c), np.uint8)
    lib().orc_nms3x3(_p(resp, _f32p), c * 4, _p(out, _u8p), c, r, c, float(thr))
    return out


def harris_pipeline(bgr, block=2, k=0.04, thr=0.0, want_resp=False):
    a, st, r, c, ch = _img(bgr)
    assert ch == 3
    mask = np.zeros((r, c), np.uint8)
    resp = np.zeros((r, c), np.float32) if want_resp else None
    rc = lib().orc_harris_pipeline(_p(a, _u8p), st, _p(mask, _u8p), c, _p(resp, _f32p) if want_resp else None, c * 4, r, c, block,
                                   float(k), float(thr))
    assert rc == 0, rc
    return (mask, resp) if want_resp else mask


# ---- synthetic frames ------------------------------------------------------------------------------

def splitmix64(z):
    return lib().orc_splitmix64(C.c_uint64(z & 0xFFFFFFFFFFFFFFFF))


def synth_frame(rows, cols, ch, family, seed, frame):
    out = np.zeros((rows, cols, ch), np.uint8)
    lib().orc_synth_frame(_p(out, _u8p), cols * ch, rows, cols, ch, family, seed, frame)
    return out[:, :, 0] if ch == 1 else out


def synth_yuyv(rows, cols, seed, frame):
    out = np.zeros((rows, cols * 2), np.uint8)
    lib().orc_synth_yuyv(_p(out, _u8p), cols * 2, rows, cols, seed, frame)
    return out


def bench_kernel7():
    k = np.zeros(49, np.int8)
    lib().orc_bench_kernel7(_p(k, _i8p))
    return k.reshape(7, 7)
