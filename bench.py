#!/usr/bin/env python3
"""bench.py -- the north-star measurement (BASELINE.json): 4K u8 BGR 7x7 filter2D, batch 64 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]          one process; N > 1: one host thread + one context per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     one process per GPU (RCCL)

A "step" is one pass of the hot path over one batch: rcv_filter2d_i8_batch on 64 device-resident 3840x2160 BGR frames
(integer 7x7 kernel of SURVEY.md 8(d), >>6, saturate).  Frames are generated ON DEVICE before the timed region (no PCIe in
`value`).  Per-GPU work is fixed (weak scaling): rank r owns frames [64r, 64r+64) (rustcv_amd.shard.frame_range); frames are
independent, so there is no data-path collective -- only a barrier and the max over ranks of the elapsed time (RCCL under
torch.distributed.run, a thread barrier in the one-process form).  `--gpus N` on a node with fewer GPUs fails.

Prints ONE JSON line on rank 0 with the contract keys plus
  "roofline":     the dominant kernel's algorithmic HBM bytes / its SUSTAINED launch time -- HIP events on the kernel's own
                  stream around >= 400 back-to-back launches (launch_ms); the same after an idle gap over 20 launches
                  (launch_ms_first20: boost clocks) for comparison -- against the 8 TB/s HBM3E peak; copy_ceiling_gbs = the best
                  plain device copy of the same 2 x 1.59 GB measured in this run (the rate the memory system of THIS box gives)
  "verified_frames": frames of the LAST timed launch's output compared bit for bit with the CPU oracle (mismatch: exit 1)
  "cpu_baseline": the C oracle (a port: C restatement, the Rust reference cannot be built here) timed on this box's host
                  cores on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS, COLS, CH = 2160, 3840, 3
BATCH = 64
ALG_BYTES_PER_PX = 6                      # BGR u8 read once + BGR u8 written once
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SEED = 0x5EED0003


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="frames per GPU (default 64 = BASELINE configs[2])")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the timed launch's output")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the in-run copy ceiling")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-time budget of the cpu_baseline sample")
    ap.add_argument("--family", type=int, default=0, help="synthetic family: 0 noise (default), 1 scene")
    ap.add_argument("--sustained", type=int, default=400, help="launches of the sustained roofline window (at least --steps)")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed run-up of the same step before the W warmup steps: the GPU needs tens of ms of load to reach its "
                         "sustained clocks")
    return ap.parse_args()


def bench_kernel7():
    """the config-3 kernel (same generator as the oracle's orc_bench_kernel7, restated so the timed path does not touch oracle/)"""
    import numpy as np

    def splitmix64(z):
        M = (1 << 64) - 1
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    return np.array([int((splitmix64(0xF117E2D ^ i) >> 40) % 17) - 8 for i in range(49)], np.int8).reshape(7, 7)


def cpu_baseline(budget_s):
    """Oracle (port) on the host cores, all threads (OpenMP over rows), bounded sample of whole 4K frames."""
    from oracle import pyoracle as orc
    cores = orc.usable_cores()   # affinity capped by the cgroup CPU quota, not os.cpu_count()
    used = orc.set_threads(cores)
    k = orc.bench_kernel7()
    frame = orc.synth_frame(ROWS, COLS, CH, 0, SEED, 0)
    orc.filter2d_i8(frame[:256], k, 6)  # warm the thread pool
    t0 = time.perf_counter()
    frames = 0
    while True:
        orc.filter2d_i8(frame, k, 6)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or frames >= 512:
            break
    mpix = frames * ROWS * COLS / 1e6 / dt
    # one-thread figure on a smaller slab (rows are independent), for the record
    orc.set_threads(1)
    slab = frame[:270]
    t1 = time.perf_counter()
    orc.filter2d_i8(slab, k, 6)
    dt1 = time.perf_counter() - t1
    orc.set_threads(cores)
    return {"value": round(mpix, 2), "unit": "Mpix/s", "cores": used, "kind": "port",
            "sample": f"{frames} whole 4K BGR frame(s), 7x7 i8 filter2D, gcc -O3 -march=native OpenMP over rows, {dt:.1f} s",
            "value_1thread": round(270 * COLS / 1e6 / dt1, 2)}


class Fence:
    """barrier + device sync on both sides of the timed region: RCCL barrier (one process per GPU) or a thread barrier
    (one process, one thread per GPU)"""

    def __init__(self, dist=None, tbarrier=None):
        self.dist, self.tb = dist, tbarrier

    def __call__(self, ctx, torch, device):
        ctx.sync()
        torch.cuda.synchronize(device)
        if self.dist is not None:
            self.dist.barrier()
        if self.tb is not None:
            self.tb.wait()
        torch.cuda.synchronize(device)


def run_rank(a, rank, world, device, ctx, fence, torch):
    """everything one GPU does; returns its measurements"""
    import numpy as np
    from rustcv_amd import _ffi, device as dev, shard

    L = _ffi.lib()
    n = a.batch
    total_frames = n * world
    f0, f1 = shard.frame_range(total_frames, rank, world)  # contiguous frame range of this rank
    assert f1 - f0 == n
    src = dev.DeviceBatch(ctx, n, ROWS, COLS, CH)
    dst = dev.DeviceBatch(ctx, n, ROWS, COLS, CH)
    dev.synth(src, a.family, SEED, f0)
    dst.memset(0)
    ctx.sync()
    k = bench_kernel7()
    kp = k.ctypes.data_as(C.POINTER(C.c_int8))
    bs, bd = src.as_rcv(), dst.as_rcv()

    def step():
        rc = L.rcv_filter2d_i8_batch(ctx.handle, C.byref(bs), C.byref(bd), kp, 7, 6)
        if rc != 0:
            raise SystemExit(f"rcv_filter2d_i8_batch failed: {rc} {_ffi.strerror(rc)}")

    def timed(launches, fn=step):
        ms = C.c_float(0.0)
        L.rcv_timer_start(ctx.handle)            # hipEvent on the stream the kernel is launched on
        for _ in range(launches):
            fn()
        L.rcv_timer_stop(ctx.handle, C.byref(ms))  # records + synchronises the ctx stream
        return float(ms.value)

    L.rcv__debug_kernels_reset()
    step()
    ctx.sync()
    kernel_name = L.rcv__debug_kernels().decode()
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < a.settle_ms:   # untimed: clocks settle under the real load
        for _ in range(8):
            step()
        ctx.sync()
    for _ in range(a.warmup):
        step()
    fence(ctx, torch, device)
    t0 = time.perf_counter()
    ev_ms = timed(a.steps)
    fence(ctx, torch, device)
    elapsed = time.perf_counter() - t0
    res = {"elapsed": elapsed, "ev_ms_steps": ev_ms / a.steps, "kernel": kernel_name}

    # ---- roofline window: sustained clocks.  No idle gap before it (the timed steps just ran), >= 400 launches back to back ----
    ns = max(a.sustained, a.steps)
    res["launch_ms"] = timed(ns) / ns
    res["launches_sustained"] = ns
    # the same launch after an idle gap, over 20 launches: what a short window sees (boost clocks) -- for comparison only
    ctx.sync()
    time.sleep(0.25)
    res["launch_ms_first20"] = timed(20) / 20

    # ---- the benchmarked launch's bytes against the oracle (outside every timed region) ----
    if not a.no_verify:
        from oracle import pyoracle as orc   # the checker, never the thing measured
        bad = []
        frames = sorted({0, n // 2 - 1 if n > 1 else 0, n - 1})
        fb = ROWS * COLS * CH
        got = np.empty(fb, np.uint8)
        for i in frames:
            _ffi.check(L.rcv_download(ctx.handle, got.ctypes.data, dst.ptr.value + i * dst.frame_stride, fb), "rcv_download")
            want = orc.filter2d_i8(orc.synth_frame(ROWS, COLS, CH, a.family, SEED, f0 + i), orc.bench_kernel7(), 6)
            if not np.array_equal(got.reshape(ROWS, COLS, CH), want):
                bad.append(f0 + i)
        res["verified_frames"] = [f0 + i for i in frames]
        res["mismatched_frames"] = bad

    # ---- copy ceiling of this box in this run: plain device copies of the same buffers (dst is scratch from here on) ----
    if not a.no_ceiling:
        nbytes = n * ROWS * COLS * CH
        best, best_name = 0.0, None
        for variant, grid, name in ((0, 1, "hipMemcpyAsync D2D"), (1, 1024, "sweep g=1024"), (1, 2048, "sweep g=2048"), (3, 512, "sweep nt g=512"),
                                    (3, 2048, "sweep nt g=2048"), (2, 1024, "block g=1024"), (5, 2048, "block nt g=2048"), (5, 512, "block nt g=512"),
                                    (8, 1024, "XCD-local sweep g=1024"), (8, 2048, "XCD-local sweep g=2048"), (9, 1024, "XCD-local sweep nt g=1024"),
                                    (9, 2048, "XCD-local sweep nt g=2048")):
            def cp():
                rc = L.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, variant, grid)
                if rc != 0:
                    raise SystemExit(f"rcv__membench failed: {rc}")
            t_run = time.perf_counter()
            while (time.perf_counter() - t_run) * 1e3 < 60.0:   # run-up: the copies get the same warm clocks as the filter
                for _ in range(8):
                    cp()
                ctx.sync()
            ms = timed(100, cp) / 100
            gbs = 2 * nbytes / ms / 1e6
            if gbs > best:
                best, best_name = gbs, name
        res["copy_ceiling_gbs"] = best
        res["copy_ceiling_kernel"] = best_name
    src.free()
    dst.free()
    return res


def report(a, world, results):
    n = a.batch
    total_frames = n * world
    elapsed = max(r["elapsed"] for r in results)
    launch_ms = max(r["launch_ms"] for r in results)
    px_per_step = total_frames * ROWS * COLS
    value = px_per_step * a.steps / elapsed / 1e6
    alg_bytes = n * ROWS * COLS * ALG_BYTES_PER_PX   # per launch, per GPU
    ach = alg_bytes / (launch_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("filter2d_i8_7x7_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": "profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE of this kernel, collected in a separate profiled run (not measured here)",
            "kernel": results[0]["kernel"], "launch_ms": round(launch_ms, 4), "launches": results[0]["launches_sustained"],
            "launch_ms_first20": round(max(r["launch_ms_first20"] for r in results), 4),
            "launch_ms_timed_steps": round(max(r["ev_ms_steps"] for r in results), 4),
            "alg_bytes_per_launch": alg_bytes}
    if "copy_ceiling_gbs" in results[0]:
        ceil = min(r["copy_ceiling_gbs"] for r in results)
        roof["copy_ceiling_gbs"] = round(ceil, 1)
        roof["copy_ceiling_kernel"] = results[0]["copy_ceiling_kernel"]
        roof["frac_of_copy_ceiling"] = round(ach / ceil, 4)
    if world > 1:
        roof["launch_ms_per_gpu"] = [round(r["launch_ms"], 4) for r in results]
    out = {
        "metric": "Mpixels/sec on 4K 7x7 filter2D", "value": round(value, 1), "unit": "Mpix/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic (splitmix64 counter noise, generated on device)",
        "config": {"workload": "4K (3840x2160) u8 BGR 7x7 filter2D, integer weights >>6, batch=64 frames per GPU (BASELINE configs[2])",
                   "frames_per_gpu": n, "global_batch": total_frames, "parallelism": f"frame-sharded x{world}, no collective"},
        "roofline": roof,
    }
    bad = []
    if "verified_frames" in results[0]:
        out["verified_frames"] = sorted(f for r in results for f in r["verified_frames"])
        bad = sorted(f for r in results for f in r["mismatched_frames"])
        out["verified"] = "bit-exact vs the CPU oracle" if not bad else f"MISMATCH in frames {bad}"
    return out, bad


def main():
    a = parse()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    under_launcher = "RANK" in os.environ
    if under_launcher and env_world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={env_world}")

    import torch  # first: librustcv_hip.so then binds to the HIP runtime torch already loaded
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback)")
    import rustcv_amd as rcv

    if under_launcher:
        # ---- one process per GPU (torch.distributed.run): RCCL carries the barrier and the max over ranks ----
        import torch.distributed as dist
        rank, local, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0")), env_world
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        dist.barrier()   # the first collective builds the RCCL communicator (hundreds of ms): keep that out of the run-up
        torch.cuda.synchronize()
        ctx = rcv.Context(local)
        res = run_rank(a, rank, world, local, ctx, Fence(dist=dist), torch)
        gathered = [None] * world
        dist.all_gather_object(gathered, res)
        bad = []
        if rank == 0:
            out, bad = report(a, world, gathered)
            if not a.no_cpu and world == 1:   # the CPU baseline leg runs at N=1 only
                out["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
            print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        ctx.close()
        sys.exit(1 if bad else 0)

    # ---- one process: one context + one host thread per GPU (rustcv_amd.multigpu.DeviceGroup) ----
    have = rcv.device_count()
    if a.gpus < 1 or a.gpus > have:
        raise SystemExit(f"--gpus {a.gpus}: this node exposes {have} GPU(s) to this process")
    world = a.gpus
    group = rcv.DeviceGroup(world)
    fence = Fence(tbarrier=threading.Barrier(world) if world > 1 else None)
    results = group.run(lambda r, ctx: run_rank(a, r, world, group.devices[r], ctx, fence, torch))
    out, bad = report(a, world, results)
    if not a.no_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
    print(json.dumps(out), flush=True)
    group.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
