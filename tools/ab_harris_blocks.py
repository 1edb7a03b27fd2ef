"""tools/ab_harris_blocks.py <label> -- the general-block Harris launches that store the f32 response (64 x 4K; cornerHarris blockSize 3 gray -> f32, pipeline blockSize 3
BGR -> mask + response, pipeline -> mask only), medians of 5 x 40 launches; run with library variants copied over rustcv_amd/librustcv_hip.so for an A/B."""
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.getcwd())
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from rustcv_amd._ffi import RCV_32F
from bench import SEEDS, HARRIS_THR
L = _ffi.lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); msk = device.DeviceBatch(ctx, n, ROWS, COLS, 1); resp = device.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_32F)
gray = device.DeviceBatch(ctx, n, ROWS, COLS, 1)
device.synth(src, 1, SEEDS[5], 0); device.synth(gray, 1, SEEDS[5], 0)
def timed(fn, launches=40):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.05:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
for name, fn in (("cornerHarris blockSize 3 (gray -> f32)", lambda: device.corner_harris(gray, resp, 3, 0.04)), ("pipeline blockSize 3, mask + response", lambda: device.harris_pipeline(src, msk, resp, 3, 0.04, HARRIS_THR)),
                 ("pipeline blockSize 3, mask", lambda: device.harris_pipeline(src, msk, None, 3, 0.04, HARRIS_THR))):
    v = [timed(fn) for _ in range(5)]
    print(f"  {sys.argv[1]:6s} {name:40s} {statistics.median(v):.4f} ms   {['%.4f' % x for x in v]}", flush=True)
