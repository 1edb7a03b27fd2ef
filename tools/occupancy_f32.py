#!/usr/bin/env python3
"""tools/occupancy_f32.py -- (round 6) what binds the sigma > 0 GaussianBlur (64 x 4K BGR, 7 taps, sigma 1.5)?  The separable f32 launch
through the measurement entry rcv__gauss_f32_bench with an untouched dynamic-LDS request that caps the workgroups (4 waves each = one wave per SIMD) a CU
holds: if the pass were bound by exposed memory latency at its occupancy, every wave taken away would cost its share; if it is bound by the vector ALU's
issue slots, the time stays flat until too few waves are left to fill them.  Both kernels (one-row: 116 VGPRs = 4 waves per SIMD; row-pair: 152 = 3)."""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from oracle import pyoracle as orc
L = _ffi.lib(); BL = _ffi.bench_lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
device.synth(src, 0, 0x5EED0003, 0)
bs, bd = src.as_rcv(), dst.as_rcv()
KS, SIGMA = int(sys.argv[1]) if len(sys.argv) > 1 else 7, 1.5
def run(lds, pairs):
    def f():
        rc = BL.rcv__gauss_f32_bench(ctx.handle, C.byref(bs), C.byref(bd), KS, SIGMA, lds, pairs)
        assert rc == 0, rc
    return f
def timed(fn, launches=30):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.05:
        for _ in range(4): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
want = orc.gaussian_blur(orc.synth_frame(ROWS, COLS, 3, 0, 0x5EED0003, 5)[:200], KS, SIGMA)
CAPS = [(0, "no cap"), (40960 + 512, "3 workgroups per CU"), (55296, "2 workgroups per CU"), (98304, "1 workgroup per CU")]
print(f"GaussianBlur {KS} taps sigma {SIGMA}, {n} x 4K BGR; samples per launch {n * ROWS * COLS * 3}")
for pairs, kname in ((0, "one-row kernel k_filter_f32_stream"), (1, "row-pair kernel k_gauss_f32_pairs")):
    for lds, cname in CAPS:
        f = run(lds, pairs)
        dst.memset(0); f(); ctx.sync()
        ok = np.array_equal(dst.download_frame(5)[:190], want[:190])
        v = [timed(f) for _ in range(3)]
        m = statistics.median(v)
        print(f"  {kname:36s} {cname:22s} {m:.4f} ms   frac of 8 TB/s at 6 B/px {n * ROWS * COLS * 6 / m / 1e6 / 8000:.4f}   exact={ok}   {['%.4f' % x for x in v]}", flush=True)
