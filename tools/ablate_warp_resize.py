#!/usr/bin/env python3
"""tools/ablate_warp_resize.py -- (round 5) the fused warpAffine -> 4x down-scale launch of BASELINE config 4 (32 x 8K -> 1080p, rotation 7
degrees): the one-launch-per-frame-tile gather kernel (k_warp_resize_box) against the frame-loop kernel (k_warp_resize_loop) over its plan
parameters, same process, rotations + medians, every variant's output compared byte for byte with the box kernel's.

    python tools/ablate_warp_resize.py [--rot 5] [--launches 30] [--plans "v:fpg:ww:xcd:strip:lds,..."]
"""
import argparse, ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device

ap = argparse.ArgumentParser()
ap.add_argument("--rot", type=int, default=5)
ap.add_argument("--launches", type=int, default=30)
ap.add_argument("--deg", type=float, default=7.0)
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--scale", type=int, default=4, help="down-scale factor of the fused launch (4: 8K -> 1080p; 2: 8K -> 4K)")
ap.add_argument("--same-frame", action="store_true", help="every frame reads frame 0 of the source (frame stride 0): the launch without its HBM reads")
ap.add_argument("--plans", default="0:0:0:0:0:-1,1:8:32:1:0:-1,1:16:32:1:0:-1,1:32:32:1:0:-1,1:8:32:0:0:-1,1:16:32:0:0:-1,1:16:16:1:0:-1,1:16:64:1:0:-1,"
                                   "1:16:32:1:4:-1,1:16:32:1:8:-1,1:16:32:1:0:27136,1:16:32:1:0:40000")
a = ap.parse_args()
L = _ffi.lib(); B = _ffi.bench_lib(); ctx = rcv.Context(0)
n, rows, cols = a.n, 4320, 7680
SC = a.scale
s = device.DeviceBatch(ctx, n, rows, cols, 3); d = device.DeviceBatch(ctx, n, rows // SC, cols // SC, 3); ref = device.DeviceBatch(ctx, n, rows // SC, cols // SC, 3)
device.synth(s, 0, 0x5EED0004, 0)
t = np.deg2rad(a.deg); c, sn = np.cos(t), np.sin(t); cx, cy = cols / 2, rows / 2
M = np.array([c, -sn, cx - c * cx + sn * cy + 13.25, sn, c, cy - sn * cx - c * cy - 8.5], np.float32)
Mp = M.ctypes.data_as(C.POINTER(C.c_float))
plans = [tuple(int(v) for v in p.split(":")) for p in a.plans.split(",") if p]

def go(plan, dst):
    bs, bd = s.as_rcv(), dst.as_rcv()
    if plan[0] == 9:   # the product entry point (whatever kernel its dispatch picks)
        _ffi.check(L.rcv_warp_affine_resize_batch(ctx.handle, C.byref(bs), C.byref(bd), Mp, rows, cols), "rcv_warp_affine_resize_batch"); return
    if a.same_frame: plan = plan[:3] + (plan[3] | 256,) + plan[4:]
    _ffi.check(B.rcv__warp_resize_bench(ctx.handle, C.byref(bs), C.byref(bd), Mp, SC, *plan), "rcv__warp_resize_bench")

def timed(plan):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        for _ in range(4): go(plan, d)
        ctx.sync()
    ms = C.c_float(); L.rcv_timer_start(ctx.handle)
    for _ in range(a.launches): go(plan, d)
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / a.launches

go((0, 0, 0, 0, 0, -1), ref); ctx.sync()
want = ref.download()
ok = {}
for p in plans:
    d.memset(0xA5); go(p, d); ctx.sync()
    ok[p] = bool(np.array_equal(d.download(), want)) if p[3] < 16 else None   # (xcd + 16 * dbg: measurement variants write other bytes)
res = {p: [] for p in plans}
for r in range(a.rot):
    for p in plans:
        res[p].append(timed(p))
alg = n * (rows // SC) * (cols // SC) * (30 if SC == 4 else 15)   # (2x: every source pixel of the 2x2 block is sampled: 4 x 3 B read + 3 B written)
print(f"{n} x {cols}x{rows} -> {cols // SC}x{rows // SC} fused warp -> {SC}x down-scale, rot {a.deg} deg, {a.launches} launches per sample, {a.rot} rotations; frac = 30 B per output px / ms / 8 TB/s")
print("  plan = variant (0 box, 1 frame loop, 2 staged row pieces, 9 product entry) : frames per wave : wave width : XCD-contiguous : strip : dynamic LDS (-1 default)")
for p in plans:
    m = statistics.median(res[p])
    print(f"  {':'.join(str(v) for v in p):24s} {m:.4f} ms  frac {alg / m / 1e6 / 8000:.4f}  same bytes as box: {ok[p]}   {['%.4f' % x for x in res[p]]}")
