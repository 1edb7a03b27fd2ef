#!/usr/bin/env python3
"""tools/ablate_chain_var.py -- (round 6) code variants of the chained-band kernel (RowsTune::var, measurement library only), same process,
rotations, medians; 64 x 4K BGR 7x7.  Every variant is first checked bit-exactly against the CPU oracle on a small batch (8 frames, chained
forced), so a variant that times well but computes something else is reported as such.
usage: ablate_chain_var.py [var ...]      (default: 1000 0 m; 'm' = the memory-only form; the experiments of profiles/r06_chain_variants.txt
were removed from the kernel after the measurement)"""
import ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import bench_kernel7
from tools._rows import Rows
from oracle import pyoracle as orc
L = _ffi.lib(); _ffi.bench_lib()
NAMES = {0: "product form (op_sel pack)", 1000: "round-5 form (shift + or pack)"}
args = sys.argv[1:] or ["1000", "0", "m"]
ctx = rcv.Context(0)
k = bench_kernel7()
# correctness, small
n8, r8, c8 = 8, 272, 1040
s8 = device.DeviceBatch(ctx, n8, r8, c8, 3); d8 = device.DeviceBatch(ctx, n8, r8, c8, 3)
device.synth(s8, 1, 0x5EED0003, 0)
rows8 = Rows(ctx, s8, d8, k)
frames = s8.download()
want = [orc.filter2d_i8(frames[i], k, 6) for i in (0, 3, 7)]
ok = {}
for a in args:
    if a == "m":
        continue
    d8.memset(0)
    rows8.fn(chain=1, var=int(a))()
    got = d8.download()
    ok[a] = all(np.array_equal(got[i], w) for i, w in zip((0, 3, 7), want))
    print(f"  var {a:>2s}  bit-exact vs oracle on {n8} x {r8} x {c8}: {ok[a]}", flush=True)
s8.free(); d8.free()
n, ROWS, COLS = 64, 2160, 3840
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
device.synth(src, 0, 0x5EED0003, 0)
rows = Rows(ctx, src, dst, k)
def timed(fn, launches=60):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.04:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
res = {}
for r in range(5):
    for a in args:
        fn = rows.fn(chain=1, dbg=4) if a == "m" else rows.fn(chain=1, var=int(a))
        res.setdefault(a, []).append(timed(fn))
nbytes = n * ROWS * COLS * 3
base = statistics.median(res[args[0]])
for a, v in res.items():
    m = statistics.median(v)
    name = "memory-only form" if a == "m" else NAMES.get(int(a), "?")
    print(f"  var {a:>2s}  {name:44s} {m:.4f} ms  frac {2 * nbytes / m / 1e6 / 8000:.4f}  {100 * (m / base - 1):+.2f} %  exact={ok.get(a, '-')}   {['%.4f' % x for x in v]}")
