#!/usr/bin/env python3
"""tools/sweep_chain.py -- (round 6) the chained kernel's plan re-swept on the final round-6 kernel (op_sel pack, tapered tail): band heights 32 .. 128, 6 / 8 / 10 / 12
waves per CU, the memory-only form; 64 x 4K BGR 7x7, same process, three rotations, medians.  Result (one box): the product plan (32 rows, 8 waves) is the best of all;
40 / 64 / 96 rows +1.6 / +2.0 / +3.2 %, 10 / 12 / 6 waves +2.5 / +3.0 / +5.4 %; the filter runs 1.8 % FASTER than its own memory-only form."""
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, '/root/repo') if os.path.exists('/root/repo/bench.py') else sys.path.insert(0, os.getcwd())
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from bench import bench_kernel7
from tools._rows import Rows
L = _ffi.lib(); _ffi.bench_lib()
n, ROWS, COLS = 64, 2160, 3840
ctx = rcv.Context(0)
src = device.DeviceBatch(ctx, n, ROWS, COLS, 3); dst = device.DeviceBatch(ctx, n, ROWS, COLS, 3)
device.synth(src, 0, 0x5EED0003, 0)
rows = Rows(ctx, src, dst, bench_kernel7())
def timed(fn, launches=60):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.04:
        for _ in range(8): fn()
        ctx.sync()
    ms = C.c_float(0.0); L.rcv_timer_start(ctx.handle)
    for _ in range(launches): fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms)); return ms.value / launches
V = [("default (32 rows, taper 8+4)", dict(chain=1)), ("40 rows", dict(chain=1, chain_rows=40)), ("48 rows", dict(chain=1, chain_rows=48)), ("64 rows", dict(chain=1, chain_rows=64)),
     ("64 rows, taper 12+8", dict(chain=1, chain_rows=64, taper=12 + 256 * 8)), ("96 rows, taper 8+8", dict(chain=1, chain_rows=96, taper=8 + 256 * 8)),
     ("128 rows, taper 8+8", dict(chain=1, chain_rows=128, taper=8 + 256 * 8)),
     ("32 rows, 10 waves per CU", dict(chain=1, wpc=10)), ("32 rows, 12 waves per CU", dict(chain=1, wpc=12)), ("32 rows 6 waves per CU", dict(chain=1, wpc=6)),
     ("memory-only 32", dict(chain=1, dbg=4)), ("memory-only 64", dict(chain=1, dbg=4, chain_rows=64))]
res = {}
for r in range(3):
    for name, t in V:
        res.setdefault(name, []).append(timed(rows.fn(**t)))
base = statistics.median(res[V[0][0]])
for name, v in res.items():
    m = statistics.median(v)
    print(f"  {name:34s} {m:.4f} ms  frac {2 * n * ROWS * COLS * 3 / m / 1e6 / 8000:.4f}  {100 * (m / base - 1):+.2f} %  {['%.4f' % x for x in v]}", flush=True)
