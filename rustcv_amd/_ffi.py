"""ctypes binding of librustcv_hip.so (the C ABI declared in include/rustcv_hip.h).

This is the Python twin of the Rust `extern "C"` block a maintainer would add to the
reference.  There is no fallback of any kind: if the shared library is
missing, or a compute call is made without a gfx950 device, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librustcv_hip.so")

RCV_OK, RCV_NOOP = 0, 1
RCV_ERR_ARG, RCV_ERR_UNSUPPORTED, RCV_ERR_SIZE, RCV_ERR_DEVICE, RCV_ERR_OOM, RCV_ERR_BUSY = -1, -2, -3, -4, -5, -6
RCV_8U, RCV_16S, RCV_32F = 0, 1, 2
RCV_HOST, RCV_DEVICE = 0, 1
RCV_YUYV2BGR, RCV_BGRA2BGR, RCV_RGB2BGR, RCV_YUYV2BGR_TWIN, RCV_BGRA2BGR_TWIN, RCV_BGR2GRAY = range(6)
RCV_BGR2BGRX, RCV_BGR2RGB, RCV_YUYV2BGR_STRIDED, RCV_UYVY2BGR_STRIDED, RCV_NV12_2BGR, RCV_BGRA2BGR_STRIDED = 6, 7, 8, 9, 10, 11
RCV_SYNTH_NOISE, RCV_SYNTH_SCENE, RCV_SYNTH_YUYV = 0, 1, 2


class RcvError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what}: rustcv_hip error {code} ({strerror(code)})")


class rcv_mat(C.Structure):
    _fields_ = [("data", C.c_void_p), ("cap", C.c_size_t), ("step", C.c_size_t),
                ("rows", C.c_int32), ("cols", C.c_int32),
                ("channels", C.c_uint8), ("depth", C.c_uint8), ("device", C.c_uint8), ("reserved", C.c_uint8)]


class rcv_batch(C.Structure):
    _fields_ = [("frame0", rcv_mat), ("frame_stride", C.c_size_t), ("n", C.c_int32), ("reserved", C.c_int32)]


class rcv_glyph(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("offset", C.c_uint64)]


def pack_glyphs(glyphs):
    """[(min_x, min_y, coverage[h, w] float32), ...] -> (rcv_glyph array, n, flat float32 coverage ndarray)"""
    import numpy as np
    glyphs = list(glyphs)
    tbl = (rcv_glyph * max(1, len(glyphs)))()
    covs, off = [], 0
    for i, (gx, gy, cov) in enumerate(glyphs):
        cov = np.ascontiguousarray(cov, dtype=np.float32)
        if cov.ndim != 2:
            raise ValueError("glyph coverage must be a 2-D (h, w) array")
        tbl[i] = rcv_glyph(int(gx), int(gy), cov.shape[1], cov.shape[0], off)
        covs.append(cov.reshape(-1))
        off += cov.size
    flat = np.concatenate(covs) if covs else np.zeros(0, np.float32)
    return tbl, len(glyphs), np.ascontiguousarray(flat, dtype=np.float32)


_P = C.POINTER
_ctx = C.c_void_p
_mat, _bat = _P(rcv_mat), _P(rcv_batch)
_i, _u8, _f, _d, _sz, _u64, _i32 = C.c_int, C.c_uint8, C.c_float, C.c_double, C.c_size_t, C.c_uint64, C.c_int32

# callback of the staging ring: int op(rcv_ctx*, const rcv_mat* dev_in, rcv_mat* dev_out, void* user)
RING_OP = C.CFUNCTYPE(C.c_int, C.c_void_p, _P(rcv_mat), _P(rcv_mat), C.c_void_p)
_ring = C.c_void_p

# name -> (restype, argtypes); the list every symbol in include/rustcv_hip.h must appear in
SIGNATURES = {
    "rcv_ring_create": (_i, [_ctx, _i, _i, _i, _i, _i, _i, _i, _i, _i, _P(_ring)]),
    "rcv_ring_destroy": (None, [_ring]),
    "rcv_ring_in_flight": (_i, [_ring]),
    "rcv_ring_input": (_i, [_ring, _mat]),
    "rcv_ring_submit": (_i, [_ring, _mat, RING_OP, C.c_void_p]),
    "rcv_ring_retire": (_i, [_ring, _mat, _mat]),
    "rcv_import_dmabuf": (_i, [_ctx, _i, _sz, _sz, _P(C.c_void_p), _P(C.c_void_p)]),
    "rcv_import_release": (None, [C.c_void_p]),
    "rcv_abi_version": (_i, []),
    "rcv_strerror": (C.c_char_p, [_i]),
    "rcv_device_count": (_i, [_P(_i)]),
    "rcv_ctx_create": (_i, [_i, _P(_ctx)]),
    "rcv_ctx_destroy": (None, [_ctx]),
    "rcv_sync": (_i, [_ctx]),
    "rcv_ctx_device": (_i, [_ctx]),
    "rcv_group_create": (_i, [_P(C.c_int), _i, _P(C.c_void_p)]),
    "rcv_group_destroy": (None, [C.c_void_p]),
    "rcv_group_size": (_i, [C.c_void_p]),
    "rcv_group_ctx": (_ctx, [C.c_void_p, _i]),
    "rcv_group_sync": (_i, [C.c_void_p]),
    "rcv_group_timer_start": (_i, [C.c_void_p]),
    "rcv_group_timer_stop": (_i, [C.c_void_p, _P(_f)]),
    "rcv_shard_range": (_i, [C.c_int64, _i, _i, _P(C.c_int64), _P(C.c_int64)]),
    "rcv_ctx_stream": (C.c_void_p, [_ctx]),
    "rcv_malloc": (_i, [_ctx, _sz, _P(C.c_void_p)]),
    "rcv_free": (_i, [_ctx, C.c_void_p]),
    "rcv_upload": (_i, [_ctx, C.c_void_p, C.c_void_p, _sz]),
    "rcv_download": (_i, [_ctx, C.c_void_p, C.c_void_p, _sz]),
    "rcv_memset": (_i, [_ctx, C.c_void_p, _i, _sz]),
    "rcv_timer_start": (_i, [_ctx]),
    "rcv_timer_stop": (_i, [_ctx, _P(_f)]),
    "rcv_fourcc_to_code": (_i, [C.c_uint32, _P(_i)]),
    "rcv_gaussian_taps_f32": (_i, [_i, _d, _P(_f)]),
    "rcv_cvt_color": (_i, [_ctx, _i, _mat, _mat]),
    "rcv_cvt_color_batch": (_i, [_ctx, _i, _bat, _bat]),
    "rcv_rectangle": (_i, [_ctx, _mat, _i32, _i32, _i32, _i32, _u8, _u8, _u8, _i32]),
    "rcv_rectangle_batch": (_i, [_ctx, _bat, _i32, _i32, _i32, _i32, _u8, _u8, _u8, _i32]),
    "rcv_blend_glyphs": (_i, [_ctx, _mat, _P(rcv_glyph), _i32, _P(C.c_float), _u64, _u8, _u8, _u8]),
    "rcv_blend_glyphs_batch": (_i, [_ctx, _bat, _P(rcv_glyph), _i32, _P(C.c_float), _u64, _u8, _u8, _u8]),
    "rcv_gaussian_blur": (_i, [_ctx, _mat, _mat, _i, _d]),
    "rcv_gaussian_blur_batch": (_i, [_ctx, _bat, _bat, _i, _d]),
    "rcv_filter2d_i8": (_i, [_ctx, _mat, _mat, _P(C.c_int8), _i, _i]),
    "rcv_filter2d_i8_batch": (_i, [_ctx, _bat, _bat, _P(C.c_int8), _i, _i]),
    "rcv_filter2d_i8_yuyv": (_i, [_ctx, _mat, _mat, _P(C.c_int8), _i, _i]),
    "rcv_filter2d_i8_yuyv_batch": (_i, [_ctx, _bat, _bat, _P(C.c_int8), _i, _i]),
    "rcv_filter2d_f32": (_i, [_ctx, _mat, _mat, _P(_f), _i, _f]),
    "rcv_filter2d_f32_batch": (_i, [_ctx, _bat, _bat, _P(_f), _i, _f]),
    "rcv_filter2d_i8_sobel": (_i, [_ctx, _mat, _mat, _mat, _P(C.c_int8), _i, _i]),
    "rcv_filter2d_i8_sobel_batch": (_i, [_ctx, _bat, _bat, _bat, _P(C.c_int8), _i, _i]),
    "rcv_sobel": (_i, [_ctx, _mat, _mat, _mat]),
    "rcv_sobel_batch": (_i, [_ctx, _bat, _bat, _bat]),
    "rcv_resize": (_i, [_ctx, _mat, _mat]),
    "rcv_resize_batch": (_i, [_ctx, _bat, _bat]),
    "rcv_warp_affine": (_i, [_ctx, _mat, _mat, _P(_f)]),
    "rcv_warp_affine_batch": (_i, [_ctx, _bat, _bat, _P(_f)]),
    "rcv_warp_affine_resize": (_i, [_ctx, _mat, _mat, _P(_f), _i, _i]),
    "rcv_warp_affine_resize_batch": (_i, [_ctx, _bat, _bat, _P(_f), _i, _i]),
    "rcv_corner_harris": (_i, [_ctx, _mat, _mat, _i, _f]),
    "rcv_corner_harris_batch": (_i, [_ctx, _bat, _bat, _i, _f]),
    "rcv_nms3x3": (_i, [_ctx, _mat, _mat, _f]),
    "rcv_nms3x3_batch": (_i, [_ctx, _bat, _bat, _f]),
    "rcv_harris_pipeline": (_i, [_ctx, _mat, _mat, _mat, _i, _f, _f]),
    "rcv_harris_pipeline_batch": (_i, [_ctx, _bat, _bat, _bat, _i, _f, _f]),
    "rcv_synth_batch": (_i, [_ctx, _bat, _i, _u64, _u64]),
}

# test hooks inside the product library (not part of include/rustcv_hip.h): knob reload, dispatch trace, profiling-build flags
DEBUG_SIGNATURES = {
    "rcv__debug_reload_knobs": (None, []),
    "rcv__debug_kernels": (C.c_char_p, []),
    "rcv__debug_kernels_reset": (None, []),
    "rcv__debug_occupancy": (_i, []),
}

# measurement kernels (plain copies / stores of every shape, launch floor, shader-clock probe): librustcv_hip_bench.so, a separate
# library on top of the product one -- bench.py's copy-ceiling leg and tools/ use it, nothing in rustcv_amd does
BENCH_LIB_PATH = os.path.join(_HERE, "librustcv_hip_bench.so")
BENCH_SIGNATURES = {
    "rcv__membench": (_i, [_ctx, C.c_void_p, C.c_void_p, _sz, _i, _i]),
    "rcv__storebench": (_i, [_ctx, C.c_void_p, C.c_void_p, _i, _i, _i, _sz, _i, _i, _i, _i, _i, _i, _i]),
    "rcv__clock_probe": (_i, [_ctx, _i, C.POINTER(C.c_float)]),
    "rcv__stripwalk": (_i, [_ctx, C.c_void_p, C.c_void_p, _i, _i, _i, _sz, _i, _i, _i, _i, _i, _i]),
    # the row-streaming filter with every plan parameter explicit + its measurement instantiations (tune: 15 ints, see ROWS_TUNE)
    "rcv__filter_rows_bench": (_i, [_ctx, _bat, _bat, _P(C.c_int8), _i, _i, _P(C.c_int), C.c_void_p]),
    "rcv__gauss_f32_bench": (_i, [_ctx, _bat, _bat, _i, C.c_double, C.c_uint, _i]),
    "rcv__harris_fused_bench": (_i, [_ctx, _bat, _bat, C.c_float, C.c_float, _i, C.c_void_p, C.POINTER(C.c_int)]),
    "rcv__filter_rows_sobel_bench": (_i, [_ctx, _bat, _bat, _bat, _P(C.c_int8), _i, _i, _P(C.c_int), C.c_void_p]),
    # the fused warp -> down-scale launch: (src, dst, M, S, variant 0 box / 1 frame loop, fpg, ww, xcd, strip, lds (-1: the product's))
    "rcv__warp_resize_bench": (_i, [_ctx, _bat, _bat, _P(_f), _i, _i, _i, _i, _i, _i, _i]),
}

# order of the ints rcv__filter_rows_bench takes (defaults = the product's plan; dbg 4 = the kernel's memory-only variant)
ROWS_TUNE = ("f7_rows", "dual_full", "chain", "chain_rows", "dbg", "wpc", "rounds", "pp", "order", "bpf", "band_rows", "taper", "wpb", "edge_pct", "var")
ROWS_TUNE_DEFAULTS = {"f7_rows": 1, "dual_full": 0, "chain": -1, "chain_rows": 0, "dbg": 0, "wpc": 0, "rounds": 0, "pp": 0, "order": 0, "bpf": 0,
                      "band_rows": 0, "taper": -1, "wpb": 0, "edge_pct": 0, "var": 0}


def rows_tune(**kw):
    """ctypes int[15] for rcv__filter_rows_bench: rows_tune(dbg=4), rows_tune(chain=0, bpf=68), ..."""
    bad = set(kw) - set(ROWS_TUNE)
    if bad:
        raise TypeError(f"unknown tune field(s) {sorted(bad)}")
    v = dict(ROWS_TUNE_DEFAULTS, **kw)
    return (C.c_int * len(ROWS_TUNE))(*[int(v[k]) for k in ROWS_TUNE])


_lib = None


def lib():
    """Load librustcv_hip.so (once).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C rustcv_amd/csrc`.  rustcv_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
            fn = getattr(l, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if l.rcv_abi_version() != 1:
            raise ImportError("librustcv_hip.so ABI version mismatch")
        _lib = l
    return _lib


_bench_lib = None


def bench_lib():
    """Load librustcv_hip_bench.so (once; after the product library, whose context and launch helpers it uses)."""
    global _bench_lib
    if _bench_lib is None:
        lib()
        if not os.path.exists(BENCH_LIB_PATH):
            raise ImportError(f"{BENCH_LIB_PATH} is missing: build it with `make -C rustcv_amd/csrc`")
        l = C.CDLL(BENCH_LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in BENCH_SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _bench_lib = l
    return _bench_lib


def strerror(code):
    return lib().rcv_strerror(int(code)).decode()


def check(code, what):
    """Negative status -> RcvError.  Returns the (non-negative) status otherwise."""
    if code < 0:
        raise RcvError(code, what)
    return code
