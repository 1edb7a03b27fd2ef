#!/usr/bin/env python3
"""tools/soak_warp_resize.py [N] -- N seeded random (shape, map, batch, row padding) cases of the fused warpAffine -> 4x down-scale against the
oracle's resize(warp_affine(.)), through the product entry (8+ frames and a map whose tile footprints fit: k_warp_resize_stage; the same call with RCV_WARP_LDS=0: k_warp_resize_box) and
through the measurement entry with random frames per tile / tile orders; prints the launches per kernel and the number of mismatches."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from oracle import pyoracle as oracle

def rot(deg, cx, cy, tx, ty, sc=1.0):
    t = np.deg2rad(deg); c, s = np.cos(t) * sc, np.sin(t) * sc
    return np.array([c, -s, cx - c * cx + s * cy + tx, s, c, cy - s * cx - c * cy + ty], np.float32)

ctx = rcv.Context(0); L = _ffi.lib(); B = _ffi.bench_lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
import time
T0 = time.time()
bad = 0; counts = {}
for case in range(FIRST, FIRST + N):
    rng = np.random.default_rng(0x5A6E00 + case)
    dr, dc = int(rng.integers(5, 90)), 4 * int(rng.integers(2, 120))
    mr, mc = 4 * dr, 4 * dc
    sr, sc = mr + int(rng.integers(0, 60)) - (40 if case % 5 == 3 else 0), mc + int(rng.integers(0, 60))
    if case % 2: sc = (sc + 15) & ~15   # (every second case: rows of a multiple of 16 bytes -- border tiles staged too)
    n = int(rng.integers(8, 30))
    kind = case % 6
    if kind == 0: M = rot(float(rng.uniform(-14, 14)), mc / 2, mr / 2, float(rng.uniform(0, 30)), float(rng.uniform(0, 30)))
    elif kind == 1: M = rot(7.0, mc / 2, mr / 2, 13.25, -8.5 + 20)
    elif kind == 2: M = np.array([1, 0, float(rng.uniform(0, 25)), 0, 1, float(rng.uniform(0, 25))], np.float32)
    elif kind == 3: M = rot(float(rng.uniform(-8, 8)), mc / 2, mr / 2, 10, 10, float(rng.uniform(0.7, 1.15)))
    elif kind == 4: M = np.array([1, float(rng.uniform(-0.15, 0.15)), 12.5, float(rng.uniform(-0.1, 0.1)), 1, 14.25], np.float32)
    else: M = rot(float(rng.uniform(-180, 180)), mc / 2, mr / 2, float(rng.uniform(-40, 40)), float(rng.uniform(-40, 40)), float(rng.uniform(0.5, 1.5)))
    M = np.asarray(M, np.float32)
    pad = 4 * int(rng.integers(0, 5))
    frames = rng.integers(0, 256, size=(n, sr, sc, 3), dtype=np.uint8)
    src = device.DeviceBatch(ctx, n, sr, sc, 3, step=sc * 3 + (-(sc * 3)) % 4 + pad)
    dst = device.DeviceBatch(ctx, n, dr, dc, 3)
    src.upload(frames)
    want = None
    def check(tag):
        global bad, want
        got = dst.download()
        if want is None: want = [oracle.resize(oracle.warp_affine(frames[i], M, mr, mc), dr, dc) for i in range(n)]
        names = L.rcv__debug_kernels().decode()
        for k in ("k_warp_resize_stage", "k_warp_resize_box"):
            if k in names: counts[k] = counts.get(k, 0) + 1
        for i in range(n):
            if not np.array_equal(got[i], want[i]):
                bad += 1; print("MISMATCH", case, tag, i, M.tolist(), (dr, dc, sr, sc, n), flush=True); break
    for knob in (None, 0):
        os.environ.pop("RCV_WARP_LDS", None)
        if knob is not None: os.environ["RCV_WARP_LDS"] = str(knob)
        L.rcv__debug_reload_knobs(); dst.memset(0x5A); L.rcv__debug_kernels_reset()
        device.warp_affine_resize(src, dst, M, mr, mc); ctx.sync(); check(f"product knob {knob}")
    os.environ.pop("RCV_WARP_LDS", None); L.rcv__debug_reload_knobs()
    order = int(rng.integers(0, 4)); strip = (int(rng.integers(1, 5)) + 256 * int(rng.integers(1, 6))) if order == 3 else int(rng.integers(0, 4))
    fpg = int(rng.integers(1, n + 1))
    bs, bd = src.as_rcv(), dst.as_rcv()
    dst.memset(0xA5); L.rcv__debug_kernels_reset()
    _ffi.check(B.rcv__warp_resize_bench(ctx.handle, C.byref(bs), C.byref(bd), M.ctypes.data_as(C.POINTER(C.c_float)), 4, 2, fpg, 0, order, strip, -1), "bench entry")
    ctx.sync(); check(f"staged fpg {fpg} order {order} strip {strip}")
    src.free(); dst.free()
    if os.environ.get('SOAK_VERBOSE'): print('case', case, 'kind', kind, (dr, dc, sr, sc, n), 'fpg', fpg, 'order', order, 'strip', strip, f'{time.time() - T0:.1f} s', flush=True)
print(f"{N} cases, launches per kernel {counts}, mismatches {bad}")
sys.exit(1 if bad else 0)
