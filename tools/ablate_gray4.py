import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
L = _ffi.lib(); ctx = rcv.Context(0)
def timed(fn, launches=60):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.08:
        for _ in range(4): fn()
        ctx.sync()
    ms = C.c_float(); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
def rot(deg, cx, cy, tx, ty):
    t = np.deg2rad(deg); c, s = np.cos(t), np.sin(t)
    return np.array([c, -s, cx - c*cx + s*cy + tx, s, c, cy - s*cx - c*cy + ty], np.float32)
for (n, rows, cols) in ((4, 2160, 3840), (8, 2160, 3840), (16, 2160, 3840), (64, 2160, 3840), (4, 4320, 7680), (8, 4320, 7680), (32, 4320, 7680), (64, 1080, 1920), (16, 1080, 1920)):
    s = device.DeviceBatch(ctx, n, rows, cols, 1); d = device.DeviceBatch(ctx, n, rows, cols, 1); s.memset(0x55)
    M = rot(7.0, cols/2, rows/2, 13.25, -8.5)
    out = []
    for env in ({"RCV_WARP_GRAY4": "0"}, {}, {"RCV_WARP_FPG": "4"}, {"RCV_WARP_FPG": "8"}, {"RCV_WARP_FPG": "16"}):
        for k in ("RCV_WARP_GRAY4", "RCV_WARP_FPG"): os.environ.pop(k, None)
        os.environ.update(env); L.rcv__debug_reload_knobs()
        out.append(timed(lambda: device.warp_affine(s, d, M)))
    print(f"{n:3d} x {cols}x{rows}: single {out[0]:.4f}  quad default {out[1]:.4f}  fq4 {out[2]:.4f}  fq8 {out[3]:.4f}  fq16 {out[4]:.4f}", flush=True)
    s.free(); d.free()
