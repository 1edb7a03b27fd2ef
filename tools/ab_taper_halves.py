import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.getcwd())
import rustcv_amd as rcv
from rustcv_amd import _ffi, device, multigpu
from bench import bench_kernel7
from tools._rows import Rows
L = _ffi.lib(); _ffi.bench_lib()
n, ROWS, COLS = 64, 2160, 3840
g = multigpu.NativeGroup.in_flight(0, 2)
c0, c1 = g.ctxs
src = device.DeviceBatch(c0, n, ROWS, COLS, 3); dst = device.DeviceBatch(c0, n, ROWS, COLS, 3)
device.synth(src, 0, 0x5EED0003, 0)
k = bench_kernel7()
ra = Rows(c0, src.view(0, 32), dst.view(0, 32), k); rb = Rows(c1, src.view(32, 32), dst.view(32, 32), k)
def run(fa, fb, calls=100):
    def call():
        fa(); fb()
    t = time.perf_counter()
    while time.perf_counter() - t < 0.08:
        for _ in range(8): call()
        g.sync()
    g.sync(); t0 = time.perf_counter()
    for _ in range(calls): call()
    g.sync()
    return (time.perf_counter() - t0) * 1e3 / calls
V = [("taper 8+4 (product)", dict(chain=1, taper=8 + 256 * 4)), ("no taper", dict(chain=1, taper=0)), ("taper 4+2", dict(chain=1, taper=4 + 256 * 2)), ("taper 0+4", dict(chain=1, taper=256 * 4)),
     ("40 rows, taper 8+4", dict(chain=1, chain_rows=40, taper=8 + 256 * 4)), ("24 rows, no taper", dict(chain=1, chain_rows=24, taper=0)), ("28 rows no taper", dict(chain=1, chain_rows=28, taper=0))]
res = {}
for r in range(5):
    for name, t in V:
        res.setdefault(name, []).append(run(ra.fn(**t), rb.fn(**t)))
base = statistics.median(res[V[0][0]])
for name, v in res.items():
    m = statistics.median(v)
    print(f"  halves, {name:24s} {m:.4f} ms per 64 frames  frac {n * ROWS * COLS * 6 / m / 1e6 / 8000:.4f}  {100 * (m / base - 1):+.2f} %   {['%.4f' % x for x in v]}", flush=True)
