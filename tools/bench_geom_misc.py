import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
L = _ffi.lib(); ctx = rcv.Context(0)
def timed(fn, launches=40):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.08:
        for _ in range(4): fn()
        ctx.sync()
    ms = C.c_float(); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
def run(name, s, d, fn, bytes_):
    L.rcv__debug_kernels_reset(); fn(); ctx.sync(); k = L.rcv__debug_kernels().decode()
    ms = timed(fn); print(f"{name:44s} {ms:8.4f} ms {bytes_/ms/1e6:8.1f} GB/s {bytes_/ms/1e6/80:5.1f} %  {k}", flush=True)
t = np.deg2rad(7.0); c, s_ = np.cos(t), np.sin(t); cx, cy = 3840, 2160
M = np.array([c, -s_, cx - c*cx + s_*cy + 13.25, s_, c, cy - s_*cx - c*cy - 8.5], np.float32)
n = 16
for ch, depth, nm in ((1, _ffi.RCV_8U, "u8 gray"), (4, _ffi.RCV_8U, "u8 4ch"), (3, _ffi.RCV_32F, "f32 3ch"), (4, _ffi.RCV_32F, "f32 4ch")):
    esz = 4 if depth == _ffi.RCV_32F else 1
    nn = n if esz == 1 and ch == 1 else 4
    s = device.DeviceBatch(ctx, nn, 4320, 7680, ch, depth); s.memset(0x3c)
    d5 = device.DeviceBatch(ctx, nn, 2880, 5120, ch, depth)
    run(f"resize {nm} 8K->5K", s, d5, lambda: device.resize(s, d5), nn*2880*5120*ch*esz*(1+2.25))
    d5.free()
    d1 = device.DeviceBatch(ctx, nn, 1080, 1920, ch, depth)
    run(f"resize {nm} 8K->1080p", s, d1, lambda: device.resize(s, d1), nn*1080*1920*ch*esz*5)
    d1.free()
    if not (ch == 1 and esz == 1):
        d8 = device.DeviceBatch(ctx, nn, 4320, 7680, ch, depth)
        run(f"warpAffine {nm} 8K", s, d8, lambda: device.warp_affine(s, d8, M), nn*4320*7680*ch*esz*2)
        d8.free()
    s.free()
