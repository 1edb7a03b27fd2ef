#!/usr/bin/env python3
"""tools/sob_timeline.py -- (round 6) when do the waves of ONE launch of config 3 whole ("3f": 7x7 filter2D -> gray -> Sobel, 64 x 4K) run?  Measurement
library, traced form (rcv__filter_rows_sobel_bench, dbg 1024): every wave records the chip-wide 100 MHz counter at its start and after its last
store.  Printed per band plan: the span, the waves' own durations, the wave slots in use over the span, the back-to-back time of the untraced launch.
usage: sob_timeline.py [bands_per_frame ...]   (0 = the product's plan)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from rustcv_amd._ffi import RCV_16S
from bench import bench_kernel7
L = _ffi.lib(); BL = _ffi.bench_lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
dx = device.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_16S); dy = device.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_16S)
device.synth(src, 0, 0x5EED0003, 0)
k = bench_kernel7(); kp = k.ctypes.data_as(C.POINTER(C.c_int8))
bs, bx, by = src.as_rcv(), dx.as_rcv(), dy.as_rcv()
NWMAX = 1 << 18
tr = device.DeviceBatch(ctx, 1, 1, NWMAX * 16, 1)
SLOTS = 12 * 256


def launch(trace=None, **tune):
    rc = BL.rcv__filter_rows_sobel_bench(ctx.handle, C.byref(bs), C.byref(bx), C.byref(by), kp, 7, 6, _ffi.rows_tune(**tune), trace)
    assert rc == 0, (rc, tune)


for bpf in [int(x) for x in (sys.argv[1:] or ["0", "6", "10", "14", "28"])]:
    plan = dict(bpf=bpf) if bpf else {}
    for _ in range(20): launch(**plan)
    ctx.sync()
    spans, lost, durs, occs, nws = [], [], [], [], []
    for rep in range(5):
        tr.memset(0)
        for _ in range(3): launch(**plan)
        launch(trace=tr.ptr, dbg=1024, **plan)
        ctx.sync()
        raw = tr.download_bytes()[: NWMAX * 16].view(np.uint64).reshape(NWMAX, 2).astype(np.int64)
        ok = raw[:, 1] > 0
        t0, t1 = raw[ok, 0], raw[ok, 1]
        nws.append(int(ok.sum()))
        base, end = t0.min(), t1.max()
        spans.append((end - base) / 100.0)
        d = (t1 - t0) / 100.0
        durs.append((np.median(d), np.percentile(d, 5), np.percentile(d, 95), d.min(), d.max()))
        lost.append(100.0 * (1.0 - np.sum(t1 - t0) / (SLOTS * float(end - base))))
        ts = np.linspace(base, end, 21)[:-1] + (end - base) / 40.0
        occs.append([100.0 * np.sum((t0 <= t) & (t1 > t)) / SLOTS for t in ts])
    ms = []
    for _ in range(3):
        t = C.c_float(); L.rcv_timer_start(ctx.handle)
        for _ in range(40): launch(**plan)
        L.rcv_timer_stop(ctx.handle, C.byref(t)); ms.append(t.value / 40)
    med = lambda v: float(np.median(v))
    m = sorted(ms)[1]
    dm = np.median(np.array(durs), axis=0)
    print(f"  bands per frame {bpf:3d} ({nws[0]} waves = {nws[0] / SLOTS:.2f} x {SLOTS} slots)  span {med(spans):6.1f} us  wave duration {dm[0]:6.1f} us (5 %: {dm[1]:6.1f}, 95 %: {dm[2]:6.1f}, "
          f"min {dm[3]:6.1f}, max {dm[4]:6.1f})  idle slots {med(lost):4.1f} %   back to back {m * 1000:6.1f} us, frac at 7 B/px {n * ROWS * COLS * 7 / m / 1e6 / 8000:.4f}", flush=True)
    print("       slots in use (%) in 20 equal steps of the span: " + " ".join(f"{v:3.0f}" for v in np.median(np.array(occs), axis=0)), flush=True)
