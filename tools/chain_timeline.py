#!/usr/bin/env python3
"""tools/chain_timeline.py -- (round 6) when do the 2 048 persistent waves of ONE k_filter_rows_chain launch run?  Measurement library, traced
form (RowsTune var 33): every wave records the chip-wide 100 MHz counter at its start and after its last store.  Printed per plan: the
launch's span, the head (first wave start -> last wave start), the tail (first wave end -> last wave end; time with fewer than 90 % / 50 %
of the waves resident), the wave-time lost to head + tail in % of slots x span, and the back-to-back time of the untraced product form.
usage: chain_timeline.py [n_frames ...]   (default 64 16)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import bench_kernel7
L = _ffi.lib(); BL = _ffi.bench_lib()
ROWS, COLS = 2160, 3840
ctx = rcv.Context(0)
k = bench_kernel7(); kp = k.ctypes.data_as(C.POINTER(C.c_int8))
NW = 2048
tr = device.DeviceBatch(ctx, 1, 1, NW * 16, 1)
PLANS = [("no taper", dict(taper=0)), ("taper 8 halved + 4 quartered", dict(taper=8 + 256 * 4)), ("taper 16 + 8", dict(taper=16 + 256 * 8)),
         ("taper 0 + 8 quartered", dict(taper=256 * 8)), ("taper 24 + 12", dict(taper=24 + 256 * 12))]
for n in [int(x) for x in (sys.argv[1:] or ["64", "16"])]:
    src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    bs, bd = src.as_rcv(), dst.as_rcv()
    def launch(trace=None, **tune):
        rc = BL.rcv__filter_rows_bench(ctx.handle, C.byref(bs), C.byref(bd), kp, 7, 6, _ffi.rows_tune(chain=1, **tune), trace)
        assert rc == 0, rc
    print(f"n = {n} frames of 4K BGR, 7x7")
    for name, plan in PLANS:
        for _ in range(20): launch(**plan)
        ctx.sync()
        spans, heads, tails, t90, t50, lost, xcd_end = [], [], [], [], [], [], []
        for rep in range(5):
            tr.memset(0)
            for _ in range(3): launch(**plan)           # the traced launch runs behind untraced ones, as in a stream of launches
            launch(trace=tr.ptr, var=33, **plan)
            ctx.sync()
            raw = tr.download_bytes()[: NW * 16].view(np.uint64).reshape(NW, 2).astype(np.int64)
            t0, t1 = raw[:, 0], raw[:, 1]
            assert (t1 > 0).all()
            base, end = t0.min(), t1.max()
            spans.append((end - base) / 100.0); heads.append((t0.max() - base) / 100.0); tails.append((end - t1.min()) / 100.0)
            ts = np.linspace(base, end, 401)
            occ = np.array([np.sum((t0 <= t) & (t1 > t)) for t in ts])
            t90.append((end - ts[len(occ) - np.argmax(occ[::-1] >= 0.9 * NW) - 1]) / 100.0)
            t50.append((end - ts[len(occ) - np.argmax(occ[::-1] >= 0.5 * NW) - 1]) / 100.0)
            lost.append(100.0 * (1.0 - np.sum(t1 - t0) / (NW * float(end - base))))
            # per XCD (block b runs on XCD b % 8 on an unpartitioned device): when its first and its last wave left, relative to the launch's end
            xcd_end.append([((t1[x::8].min() - end) / 100.0, (t1[x::8].max() - end) / 100.0) for x in range(8)])
        ms = []
        for _ in range(3):
            t = C.c_float(); L.rcv_timer_start(ctx.handle)
            for _ in range(60): launch(**plan)
            L.rcv_timer_stop(ctx.handle, C.byref(t)); ms.append(t.value / 60)
        med = lambda v: float(np.median(v))
        m = sorted(ms)[1]
        print(f"  {name:30s} span {med(spans):6.1f} us  head {med(heads):4.1f}  tail {med(tails):5.1f} (below 90 %: {med(t90):4.1f}, below 50 %: {med(t50):4.1f})  "
              f"idle slots {med(lost):4.1f} %   back to back {m * 1000:6.1f} us = {m * 1000 / n:.3f} us/frame, frac {n * ROWS * COLS * 6 / m / 1e6 / 8000:.4f}", flush=True)
        xe = np.median(np.array(xcd_end), axis=0)
        print("      per XCD, first .. last wave out (us before the launch's end): " + "  ".join(f"{-a:.0f}..{-b:.0f}" for a, b in xe), flush=True)
    src.free(); dst.free()
