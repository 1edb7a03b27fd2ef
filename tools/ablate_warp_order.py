#!/usr/bin/env python3
"""tools/ablate_warp_order.py -- (round 4) BGR warpAffine 32 x 8K, same process, rotations + medians: tile order inside an XCD's run
(0 = raster, n = vertical strips of n tile columns) and frames per workgroup, both through the test knob
RCV_WARP_FPG = fpg + 256 * (strip + 1)  (fpg 0: the default; no strip part: the default of 6).

    python tools/ablate_warp_order.py [--rot 5] [--launches 40]
"""
import argparse, ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device

ap = argparse.ArgumentParser()
ap.add_argument("--rot", type=int, default=5)
ap.add_argument("--launches", type=int, default=40)
ap.add_argument("--deg", type=float, default=7.0)
ap.add_argument("--strips", default="0,4,6,8,12,16,24")
ap.add_argument("--fpgs", default="")
ap.add_argument("--kind", default="bgr", help="bgr | gray | f32")
ap.add_argument("--combos", default="", help="fpg:strip pairs, e.g. 8:0,16:8")
ap.add_argument("--shape", default="", help="n,rows,cols (default: 32 (f32: 8) x 8K)")
a = ap.parse_args()
L = _ffi.lib(); ctx = rcv.Context(0)
n, rows, cols = (8 if a.kind == "f32" else 32), 4320, 7680
if a.shape:
    n, rows, cols = (int(v) for v in a.shape.split(","))
ch = 3 if a.kind == "bgr" else 1
if a.kind == "f32":
    s = device.DeviceBatch(ctx, n, rows, cols, 1, depth=_ffi.RCV_32F); d = device.DeviceBatch(ctx, n, rows, cols, 1, depth=_ffi.RCV_32F); s.memset(0x3C)
else:
    s = device.DeviceBatch(ctx, n, rows, cols, ch); d = device.DeviceBatch(ctx, n, rows, cols, ch)
    device.synth(s, 0, 0x5EED0007, 0)
t = np.deg2rad(a.deg); c, sn = np.cos(t), np.sin(t); cx, cy = cols / 2, rows / 2
M = np.array([c, -sn, cx - c * cx + sn * cy + 13.25, sn, c, cy - sn * cx - c * cy - 8.5], np.float32)

def timed(launches):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        for _ in range(4): device.warp_affine(s, d, M)
        ctx.sync()
    ms = C.c_float(); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): device.warp_affine(s, d, M)
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches

variants = [{"RCV_WARP_FPG": str(256 * (int(x) + 1))} for x in a.strips.split(",") if x] + [{"RCV_WARP_FPG": x} for x in a.fpgs.split(",") if x]
names = ["strip %s" % x for x in a.strips.split(",") if x] + ["frames per workgroup %s" % x for x in a.fpgs.split(",") if x]
for c in [c for c in a.combos.split(",") if c]:
    f, st = (int(v) for v in c.split(":"))
    variants.append({"RCV_WARP_FPG": str(f + 256 * (st + 1))})
    names.append("frames per workgroup %d, strip %d" % (f, st))
res = {i: [] for i in range(len(variants))}
for r in range(a.rot):
    for i, env in enumerate(variants):
        os.environ.pop("RCV_WARP_FPG", None)
        os.environ.update(env); L.rcv__debug_reload_knobs()
        res[i].append(timed(a.launches))
px = n * rows * cols
bpp = {"bgr": 6, "gray": 2, "f32": 8}[a.kind]
print(f"{n} x {cols}x{rows} {a.kind} warpAffine rot {a.deg} deg, {a.launches} launches per sample, {a.rot} rotations; frac = {bpp} B/px / ms / 8 TB/s; for gray / f32 any strip value also switches the XCD-contiguous runs on")
for i, env in enumerate(variants):
    m = statistics.median(res[i])
    print(f"  {names[i]:32s} {m:.4f} ms  frac {px * bpp / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in res[i]]}")
