#!/usr/bin/env python3
"""Launch-bound configurations, direct calls vs one recorded launch graph (rcv_graph_*).

config 0 chain (the reference's own loop, examples/camera_demo.rs:50-76): YUYV->BGR + rectangle on ONE 640x480 frame;
config 1: one 1080p 5x5 GaussianBlur; and a 4-op chain on a 1080p frame (blur, gray, Sobel, NMS-free Harris pipeline is
separate) to show how the saving grows with the chain.  Device-resident buffers, HIP-event timing on the context stream.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from rustcv_amd.imgproc import Rect, Scalar  # noqa: E402


def timed(ctx, fn, iters):
    L = _ffi.lib()
    for _ in range(20):
        fn()
    ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(iters):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / iters * 1e3   # us per iteration


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "graph_bench.json"))
    a = ap.parse_args()
    ctx = rcv.Context(0)
    out = []

    def case(name, chain, nlaunch):
        chain()
        ctx.sync()
        direct = timed(ctx, chain, a.iters)
        with ctx.capture() as g:
            chain()
        graph = timed(ctx, g.launch, a.iters)
        g.close()
        rec = {"case": name, "kernels": nlaunch, "direct_us": round(direct, 2), "graph_us": round(graph, 2), "speedup": round(direct / graph, 2)}
        print(json.dumps(rec))
        out.append(rec)

    y = device.DeviceBatch(ctx, 1, 480, 640, 2)
    b = device.DeviceBatch(ctx, 1, 480, 640, 3)
    device.synth(y, 2, 1, 0)
    case("config 0: 640x480 YUYV->BGR + rectangle", lambda: (device.cvt_color(y, b, _ffi.RCV_YUYV2BGR),
                                                              device.rectangle(b, Rect(200, 150, 240, 240), Scalar(0, 255, 0), 2)), 2)
    s = device.DeviceBatch(ctx, 1, 1080, 1920, 3)
    d = device.DeviceBatch(ctx, 1, 1080, 1920, 3)
    gr = device.DeviceBatch(ctx, 1, 1080, 1920, 1)
    dx = device.DeviceBatch(ctx, 1, 1080, 1920, 1, _ffi.RCV_16S)
    dy = device.DeviceBatch(ctx, 1, 1080, 1920, 1, _ffi.RCV_16S)
    device.synth(s, 1, 2, 0)
    case("config 1: 1080p GaussianBlur 5x5, one frame", lambda: device.gaussian_blur(s, d, 5, 0.0), 1)
    case("1080p chain: blur 5x5 -> gray -> Sobel", lambda: (device.gaussian_blur(s, d, 5, 0.0), device.cvt_color(d, gr, _ffi.RCV_BGR2GRAY),
                                                            device.sobel(gr, dx, dy)), 3)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
