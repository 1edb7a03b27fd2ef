#!/usr/bin/env python3
"""tools/harris_timeline.py -- (round 6) when do the waves of ONE k_harris_fused launch (BASELINE config 5: 64 x 4K BGR -> mask) run?  Measurement
library (rcv__harris_fused_bench): every wave records the chip-wide 100 MHz counter at its start and after its last store, and where it ran
(XCD, CU).  Printed per segment height: the launch's span, the waves' own durations (median, 5 % / 95 %), the wave slots in use over the span
(wave-time / (slots x span)), when each XCD's last wave left relative to the launch's end, and the back-to-back time of the untraced launch.
Last column: the same launch alternating on two contexts (two streams, own buffers each), per launch.
usage: harris_timeline.py [seg_rows ...]   (0 = the product's plan)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import SEEDS, HARRIS_THR
L = _ffi.lib(); BL = _ffi.bench_lib()
N, ROWS, COLS = 64, 2160, 3840
from rustcv_amd import multigpu
grp = multigpu.NativeGroup.in_flight(0, 2)
ctx, ctx2 = grp.ctxs
src = device.DeviceBatch(ctx, N, ROWS, COLS, 3); msk = device.DeviceBatch(ctx, N, ROWS, COLS, 1)
device.synth(src, 1, SEEDS[5], 0)
bs, bm = src.as_rcv(), msk.as_rcv()
src2 = device.DeviceBatch(ctx2, N, ROWS, COLS, 3); msk2 = device.DeviceBatch(ctx2, N, ROWS, COLS, 1)
device.synth(src2, 1, SEEDS[5] + 1, 0)
bs2, bm2 = src2.as_rcv(), msk2.as_rcv()
NSTRIPS = (COLS + 495) // 496
SLOTS = 12 * 256


def launch(seg, trace=None, second=False):
    w = C.c_int(0)
    if second: rc = BL.rcv__harris_fused_bench(ctx2.handle, C.byref(bs2), C.byref(bm2), 0.04, HARRIS_THR, seg, None, C.byref(w))
    else: rc = BL.rcv__harris_fused_bench(ctx.handle, C.byref(bs), C.byref(bm), 0.04, HARRIS_THR, seg, trace, C.byref(w))
    assert rc == 0, rc
    return w.value


for seg in [int(x) for x in (sys.argv[1:] or ["0", "360", "180", "120", "90", "72"])]:
    per_seg = launch(seg)
    segrows = seg if seg > 0 else 135      # (the product's plan for 64 x 4K)
    nsegs = (ROWS + segrows - 1) // segrows
    nw = per_seg * nsegs
    tr = device.DeviceBatch(ctx, 1, 1, nw * 24, 1)
    for _ in range(20): launch(seg)
    grp.sync()
    spans, lost, durs, xcd_end, heads, occs = [], [], [], [], [], []
    for rep in range(5):
        tr.memset(0)
        for _ in range(3): launch(seg)
        launch(seg, tr.ptr)
        grp.sync()
        raw = tr.download_bytes()[: nw * 24].view(np.uint64).reshape(nw, 3)
        t0, t1 = raw[:, 0].astype(np.int64), raw[:, 1].astype(np.int64)
        xcd = (raw[:, 2] >> np.uint64(32)).astype(np.int64)
        assert (t1 > 0).all(), "segment height differs from the plan: pass it explicitly"
        base, end = t0.min(), t1.max()
        spans.append((end - base) / 100.0)
        d = (t1 - t0) / 100.0
        durs.append((np.median(d), np.percentile(d, 5), np.percentile(d, 95)))
        lost.append(100.0 * (1.0 - np.sum(t1 - t0) / (SLOTS * float(end - base))))
        heads.append((np.sort(t0)[min(SLOTS, nw) - 1] - base) / 100.0)
        xcd_end.append([((t1[xcd == x].max() - end) / 100.0 if (xcd == x).any() else 0.0) for x in range(8)])
        ts = np.linspace(base, end, 21)[:-1] + (end - base) / 40.0
        occs.append([100.0 * np.sum((t0 <= t) & (t1 > t)) / SLOTS for t in ts])
    ms = []
    for _ in range(3):
        t = C.c_float(); L.rcv_timer_start(ctx.handle)
        for _ in range(60): launch(seg)
        L.rcv_timer_stop(ctx.handle, C.byref(t)); ms.append(t.value / 60)
    med = lambda v: float(np.median(v))
    m = sorted(ms)[1]
    ms2 = []
    for _ in range(3):
        for _ in range(10):
            launch(seg); launch(seg, second=True)
        grp.sync(); grp.timer_start()
        for _ in range(30):
            launch(seg); launch(seg, second=True)
        ms2.append(grp.timer_stop() / 60)
    m2 = sorted(ms2)[1]
    dm = np.median(np.array(durs), axis=0)
    xe = np.median(np.array(xcd_end), axis=0)
    print(f"  seg {segrows:4d} rows ({nw} waves = {nw / SLOTS:.2f} rounds of {SLOTS})  span {med(spans):6.1f} us  first {min(SLOTS, nw)} waves started within {med(heads):5.1f} us  "
          f"wave duration {dm[0]:6.1f} us (5 %: {dm[1]:6.1f}, 95 %: {dm[2]:6.1f})  idle slots {med(lost):4.1f} %   back to back {m * 1000:6.1f} us, frac {N * ROWS * COLS * 4 / m / 1e6 / 8000:.4f};  two contexts {m2 * 1000:6.1f} us, frac {N * ROWS * COLS * 4 / m2 / 1e6 / 8000:.4f}", flush=True)
    print("       last wave of each XCD left (us before the launch's end): " + "  ".join(f"{-v:5.1f}" for v in xe), flush=True)
    print("       slots in use (%) in 20 equal steps of the span: " + " ".join(f"{v:3.0f}" for v in np.median(np.array(occs), axis=0)), flush=True)
    tr.free() if hasattr(tr, "free") else None
