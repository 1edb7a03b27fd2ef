"""tools/ab_harris_gray.py <label> -- the Harris pipeline from a one-channel source (64 x 4K gray -> mask), medians of 5 x 40 launches; run with library variants
copied over rustcv_amd/librustcv_hip.so for an A/B (round 6: packed-f32 Sobel stage for aligned gray sources)."""
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.getcwd())
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import SEEDS, HARRIS_THR
L = _ffi.lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
gray = device.DeviceBatch(ctx, n, ROWS, COLS, 1); msk = device.DeviceBatch(ctx, n, ROWS, COLS, 1)
device.synth(gray, 1, SEEDS[5], 0)
def timed(fn, launches=40):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.05:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
v = [timed(lambda: device.harris_pipeline(gray, msk, None, 2, 0.04, HARRIS_THR)) for _ in range(5)]
print(f"  {sys.argv[1]:8s} pipeline gray -> mask  {statistics.median(v):.4f} ms   {['%.4f' % x for x in v]}", flush=True)
