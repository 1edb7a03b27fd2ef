#!/usr/bin/env python3
"""tools/config2_latency.py -- BASELINE config 2 (one 1080p BGR frame, 5x5 integer Gaussian) is a latency problem: launch-to-launch
microseconds of back-to-back launches for every kernel that can take it, with the band plans of the row kernel."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402

L = _ffi.lib()
BL = _ffi.bench_lib()   # copy / store / clock probes: librustcv_hip_bench.so, not part of the product library
KNOBS = ("RCV_F7_ROWS", "RCV_FR_BAND_ROWS", "RCV_F7_NO_LAT", "RCV_GAUSS_ROWS", "RCV_GR_SEG", "RCV_FR_WPB", "RCV_GR_PLAIN")


def setenv(env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    L.rcv__debug_reload_knobs()


def main():
    ctx = rcv.Context(0)
    from oracle import pyoracle as orc
    # the floor: back-to-back launches of an EMPTY kernel on the same stream
    scratch = device.DeviceBatch(ctx, 1, 16, 16, 1)
    for g in (1, 256, 1024, 2048):
        def nop(g=g):
            BL.rcv__membench(ctx.handle, scratch.ptr, scratch.ptr, 16, 30, g)
        for _ in range(300):
            nop()
        ctx.sync()
        ms = C.c_float()
        L.rcv_timer_start(ctx.handle)
        for _ in range(2000):
            nop()
        L.rcv_timer_stop(ctx.handle, C.byref(ms))
        print(f"empty kernel, {g:4d} workgroups of 64 threads: {ms.value / 2000 * 1e3:.2f} us per launch (launch-to-launch floor)", flush=True)
    # what a plain COPY of one 1080p BGR frame costs launch-to-launch: the floor of any kernel that reads and writes the frame once
    fa, fb = device.DeviceBatch(ctx, 1, 1080, 1920, 3), device.DeviceBatch(ctx, 1, 1080, 1920, 3)
    nb = 1080 * 1920 * 3
    for v, g, nm in ((21, 256, "sweep U=2 nt g=256"), (21, 512, "sweep U=2 nt g=512"), (18, 512, "sweep U=2 plain g=512"), (10, 256, "sweep U=4 plain g=256"), (18, 1024, "sweep U=2 plain g=1024"),
                     (18, 2048, "sweep U=2 plain g=2048"), (0, 1, "hipMemcpyAsync D2D")):
        def cp(v=v, g=g):
            BL.rcv__membench(ctx.handle, fb.ptr, fa.ptr, nb, v, g)
        for _ in range(300):
            cp()
        ctx.sync()
        ms = C.c_float()
        L.rcv_timer_start(ctx.handle)
        for _ in range(2000):
            cp()
        L.rcv_timer_stop(ctx.handle, C.byref(ms))
        print(f"copy of one 1080p BGR frame (6.2 MB), {nm:24s}: {ms.value / 2000 * 1e3:.2f} us per launch", flush=True)
    fa.free()
    fb.free()
    for (rows, cols, tag) in ((1080, 1920, "1080p"), (2160, 3840, "4K")):
        s, d = device.DeviceBatch(ctx, 1, rows, cols, 3), device.DeviceBatch(ctx, 1, rows, cols, 3)
        device.synth(s, 0, 0x5EED0002, 0)
        want = orc.gaussian_blur(s.download()[0], 5, 0.0)

        def t(steps=2000):
            for _ in range(300):
                device.gaussian_blur(s, d, 5, 0.0)
            ctx.sync()
            ms = C.c_float()
            L.rcv_timer_start(ctx.handle)
            for _ in range(steps):
                device.gaussian_blur(s, d, 5, 0.0)
            L.rcv_timer_stop(ctx.handle, C.byref(ms))
            return ms.value / steps * 1e3
        variants = [("default", {}), ("strip kernel, pipelined variant", {"RCV_GAUSS_ROWS": 0, "RCV_F7_NO_LAT": 1}), ("strip kernel, latency variant", {"RCV_GAUSS_ROWS": 0})]
        quick = "--quick" in sys.argv
        for br in (() if quick else (0, 4, 6, 8, 10, 12, 16, 24, 32)):
            env = {"RCV_GAUSS_ROWS": 0, "RCV_F7_ROWS": 1}
            if br:
                env["RCV_FR_BAND_ROWS"] = br
            variants.append((f"row MFMA kernel, band rows {br or 'plan'}", env))
        for seg in (() if quick else (0, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 20, 24, 32)):
            env = {"RCV_GAUSS_ROWS": 1}
            if seg:
                env["RCV_GR_SEG"] = seg
            variants.append((f"register-window kernel, segment rows {seg or 'plan'}", env))
        variants.append(("register-window kernel, plain stores", {"RCV_GAUSS_ROWS": 1, "RCV_GR_PLAIN": 1}))
        for tagv, env in variants:
            setenv(env)
            L.rcv__debug_kernels_reset()
            d.memset(0)
            device.gaussian_blur(s, d, 5, 0.0)
            kn = L.rcv__debug_kernels().decode()
            ok = bool(np.array_equal(d.download()[0], want))
            us = sorted(t() for _ in range(3))[1]
            print(f"{tag:6s} {tagv:44s} {us:7.2f} us per launch   bit-exact {ok}   {kn[:60]}", flush=True)
        setenv({})
        s.free()
        d.free()
    ctx.close()


if __name__ == "__main__":
    main()
