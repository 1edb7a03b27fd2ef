#!/usr/bin/env python3
"""PCIe-inclusive streaming rate of the capture-side chain through the staging ring ("next" row f3).

Host YUYV 4K frame -> H2D -> fused YUYV->BGR->7x7 filter2D -> D2H -> host BGR frame, `depth` frames in flight.  The producer
writes into the ring's pinned input in place and the consumer reads the pinned output (no pageable copies), so the
figure is what the PCIe link and the copy engines allow; depth 1 is the serialised upload/compute/download of the plain
host-Mat entry points.  Prints one JSON line per depth and writes profiles/ring_bench.json.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--rows", type=int, default=2160)
    ap.add_argument("--cols", type=int, default=3840)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "ring_bench.json"))
    a = ap.parse_args()
    L = _ffi.lib()
    ctx = rcv.Context(0)
    rng = np.random.default_rng(7)
    frame = rng.integers(0, 256, size=(a.rows, a.cols, 2), dtype=np.uint8)
    k = (np.arange(49, dtype=np.int8).reshape(7, 7) - 24)
    kp = k.ctypes.data_as(C.POINTER(C.c_int8))
    op = lambda c, din, dout: L.rcv_filter2d_i8_yuyv(c, din, dout, kp, 7, 6)
    results = []
    for depth in (1, 2, 3, 4):
        with rcv.StagingRing(ctx, depth, (a.rows, a.cols, 2), (a.rows, a.cols, 3)) as ring:
            def run(nframes):
                sink = 0
                for _ in range(nframes):
                    if ring.full():
                        sink += int(ring.retire(copy=False)[0, 0, 0])
                    ring.input_view()[0, :8, 0] = 1          # the producer touches its buffer; the payload is already there
                    ring.submit(None, op)
                while ring.in_flight():
                    sink += int(ring.retire(copy=False)[0, 0, 0])
                return sink
            for _ in range(depth):                           # fill every slot's pinned input once
                ring.input_view()[...] = frame
                ring.submit(None, op)
            while ring.in_flight():
                ring.retire(copy=False)
            run(10)
            t0 = time.perf_counter()
            run(a.frames)
            dt = time.perf_counter() - t0
        px = a.rows * a.cols
        rec = {"op": "YUYV->BGR->filter2D 7x7 i8 via staging ring", "depth": depth, "frames": a.frames, "fps": a.frames / dt,
               "mpix_per_s": a.frames * px / dt / 1e6, "pcie_bytes_per_frame": px * 5, "pcie_gb_per_s": a.frames * px * 5 / dt / 1e9,
               "ms_per_frame": dt / a.frames * 1e3}
        print(json.dumps(rec))
        results.append(rec)
    json.dump(results, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
