#!/usr/bin/env python3
"""tools/ablate_sob.py -- (round 6) where the time of the one-launch config 3 (7x7 filter2D -> gray -> Sobel, "3f") goes: the product launch against
its measurement forms (librustcv_hip_bench.so: rcv__filter_rows_sobel_bench) -- no gradient stores, adds instead of the matrix instructions,
the stores fed with the packed filter output instead of the gray / Sobel arithmetic -- and band plans; the plain filter and the Sobel launch of the
same batch beside it.  64 x 4K BGR, same process, rotations, medians."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from rustcv_amd._ffi import RCV_16S
from bench import bench_kernel7
L = _ffi.lib(); BL = _ffi.bench_lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
dx = device.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_16S); dy = device.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_16S)
device.synth(src, 0, 0x5EED0003, 0)
k = bench_kernel7(); kp = k.ctypes.data_as(C.POINTER(C.c_int8))
bs, bx, by = src.as_rcv(), dx.as_rcv(), dy.as_rcv()
def sob(**tune):
    t = _ffi.rows_tune(**tune)
    def f():
        rc = BL.rcv__filter_rows_sobel_bench(ctx.handle, C.byref(bs), C.byref(bx), C.byref(by), kp, 7, 6, t, None)
        assert rc == 0, (rc, tune)
    return f
def timed(fn, launches=40):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.04:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
VARIANTS = [("3f product launch", sob()),
            ("3f, no gradient stores", sob(dbg=1)),
            ("3f, adds instead of MFMA", sob(dbg=4)),
            ("3f, no stores, no MFMA", sob(dbg=5)),
            ("3f, stores fed with the filter output (no gray / Sobel arithmetic)", sob(dbg=2048)),
            ("3f, neither MFMA nor gray / Sobel arithmetic (loads + stores)", sob(dbg=2052)),
            ("3f, no barrier (the seam values race: a measurement)", sob(dbg=4096)),
            ("3f, three row pairs in flight, two waves per SIMD (184 VGPRs), bands for 10 waves per CU", sob(dbg=8192, wpc=10)),
            ("3f, 10 bands per frame", sob(bpf=10)), ("3f, 28 bands per frame", sob(bpf=28)),
            ("3f, 42 bands per frame (51 rows)", sob(bpf=42)), ("3f, 68 bands per frame (32 rows)", sob(bpf=68)), ("3f, 14 bands per frame", sob(bpf=14)),
            ("3f, bands for 8 waves per CU", sob(wpc=8)), ("3f, bands for 12 waves per CU", sob(wpc=12)),
            ("plain filter2D (chained), 6 B/px", lambda: device.filter2d(src, dst, k, shift=6)),
            ("Sobel of the BGR batch (3s), 7 B/px", lambda: device.sobel(src, dx, dy))]
sel = [int(x) for x in sys.argv[1:]] or range(len(VARIANTS))
res = {}
for r in range(3):
    for i in sel:
        res.setdefault(i, []).append(timed(VARIANTS[i][1]))
base = statistics.median(res[list(res)[0]])
for i, v in res.items():
    m = statistics.median(v)
    print(f"  {VARIANTS[i][0]:72s} {m:.4f} ms  {100 * (m / base - 1):+6.1f} %   frac at 7 B/px {n * ROWS * COLS * 7 / m / 1e6 / 8000:.4f}   {['%.4f' % x for x in v]}", flush=True)
