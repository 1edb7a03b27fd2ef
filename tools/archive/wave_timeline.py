#!/usr/bin/env python3
"""tools/wave_timeline.py -- (profiling build, make EXTRA=-DRCV_ABLATE) when do the waves of ONE north-star launch run?  Every wave
records the chip-wide 100 MHz counter at its start and after its last store; printed: the launch's duration, resident waves over time
(deciles), the head (time until 90 % of the slots are filled) and the tail (time from the moment fewer than 90 % / 50 % of the slots are
busy to the end), per-XCD finish times."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402

L = _ffi.lib()


def main():
    from bench import bench_kernel7
    ctx = rcv.Context(0)
    n, rows, cols = 64, 2160, 3840
    src = device.DeviceBatch(ctx, n, rows, cols, 3)
    dst = device.DeviceBatch(ctx, n, rows, cols, 3)
    device.synth(src, 0, 0x5EED0003, 0)
    k = bench_kernel7()
    nw = 1 << 16
    tr = device.DeviceBatch(ctx, 1, 1, nw * 16, 1)
    for env in ({"RCV_FR_TAPER": 0}, {}, {"RCV_FR_TAPER": 150}, {"RCV_FR_TAPER": 200}, {"RCV_FR_TAPER": 60}):
        for kk in ("RCV_FR_BPF", "RCV_FR_TAPER"):
            os.environ.pop(kk, None)
        for kk, v in env.items():
            os.environ[kk] = str(v)
        L.rcv__debug_reload_knobs()
        L.rcv__debug_set(0)
        for _ in range(30):
            device.filter2d(src, dst, k, shift=6)
        ctx.sync()
        tr.memset(0)
        L.rcv__debug_trace_buffer(tr.ptr)
        L.rcv__debug_set(24)
        device.filter2d(src, dst, k, shift=6)
        ctx.sync()
        L.rcv__debug_set(0)
        L.rcv__debug_trace_buffer(None)
        raw = tr.download_bytes()[: nw * 16].view(np.uint64).reshape(nw, 2)
        live = raw[:, 1] > 0
        t0, t1 = raw[live, 0].astype(np.int64), raw[live, 1].astype(np.int64)
        base, end = t0.min(), t1.max()
        dur = (end - base) / 100.0
        print(f"{env or 'default'}: {live.sum()} waves, launch {dur:.1f} us; wave duration mean {np.mean(t1 - t0) / 100:.1f} us  p5 {np.percentile(t1 - t0, 5) / 100:.1f}  p95 {np.percentile(t1 - t0, 95) / 100:.1f}")
        ts = np.linspace(base, end, 201)
        occ = np.array([np.sum((t0 <= t) & (t1 > t)) for t in ts])
        peak = occ.max()
        print("   resident waves at 0,5,..,100 % of the launch:", " ".join(str(int(x)) for x in occ[::10]))
        head = ts[np.argmax(occ >= 0.9 * peak)] - base
        below90 = ts[len(occ) - np.argmax(occ[::-1] >= 0.9 * peak) - 1]
        below50 = ts[len(occ) - np.argmax(occ[::-1] >= 0.5 * peak) - 1]
        print(f"   peak {peak} waves; head (to 90 % of peak) {head / 100:.1f} us; tail below 90 %: {(end - below90) / 100:.1f} us, below 50 %: {(end - below50) / 100:.1f} us")
        ms = []
        for _ in range(3):
            t = C.c_float()
            L.rcv_timer_start(ctx.handle)
            for _ in range(100):
                device.filter2d(src, dst, k, shift=6)
            L.rcv_timer_stop(ctx.handle, C.byref(t))
            ms.append(t.value / 100)
        print(f"   100 launches back to back: {sorted(ms)[1]:.4f} ms per launch = {n * rows * cols * 6 / sorted(ms)[1] / 1e6 / 8000:.4f} of 8 TB/s")
        print(f"   mean occupancy over the launch: {np.mean(occ) / peak * 100:.1f} % of peak  (wave-time {np.sum(t1 - t0) / 100 / dur / peak * 100:.1f} %)")
        idx = np.nonzero(live)[0]
        for x in range(8):
            m = (idx % 8) == x     # wave w = block * wpb + wave; block b on XCD b % 8 (wpb = 1)
            print(f"   XCD {x}: waves {m.sum():5d}  last end {(t1[m].max() - base) / 100:.1f} us", end="")
        print()
    ctx.close()


if __name__ == "__main__":
    main()
