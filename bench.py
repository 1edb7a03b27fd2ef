#!/usr/bin/env python3
"""bench.py -- the north-star measurement (BASELINE.json): 4K u8 BGR 7x7 filter2D, batch 64 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: rcv_filter2d_i8_batch on 64 device-resident
3840x2160 BGR frames (integer 7x7 kernel of SURVEY.md 8(d), >>6, saturate).  Frames are generated
ON DEVICE before the timed region (no PCIe in `value`).  Per-GPU work is fixed (weak scaling): rank r
owns frames [64r, 64r+64); frames are independent, so there is no data-path collective -- torch
.distributed (RCCL) only carries the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0 with the contract keys plus
  "roofline":     dominant kernel's algorithmic HBM bytes / its average launch time (HIP events on
                  the stream the kernel runs on) against the 8 TB/s HBM3E peak
  "cpu_baseline": the C oracle (a port: C restatement, the Rust reference cannot be built here)
                  timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
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
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-time budget of the cpu_baseline sample")
    ap.add_argument("--family", type=int, default=0, help="synthetic family: 0 noise (default), 1 scene")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed run-up of the same step before the W warmup steps: the GPU needs tens of ms of load to reach its "
                         "sustained clocks (measured: 0.685 ms/step after 5 warmup steps, 0.660 after 200)")
    return ap.parse_args()


def cpu_baseline(budget_s):
    """Oracle (port) on the host cores, all threads (OpenMP over rows), bounded sample of whole 4K frames."""
    import numpy as np
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


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import torch  # first: librustcv_hip.so then binds to the HIP runtime torch already loaded
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    use_dist = world > 1 or "RANK" in os.environ   # under torch.distributed.run even a single rank goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    import numpy as np
    import rustcv_amd as rcv
    from rustcv_amd import _ffi, device
    from rustcv_amd import shard

    L = _ffi.lib()
    ctx = rcv.Context(local)
    n = a.batch
    total_frames = n * world
    f0, f1 = shard.frame_range(total_frames, rank, world)  # contiguous frame range of this rank
    assert f1 - f0 == n
    src = device.DeviceBatch(ctx, n, ROWS, COLS, CH)
    dst = device.DeviceBatch(ctx, n, ROWS, COLS, CH)
    device.synth(src, a.family, SEED, f0)
    dst.memset(0)
    ctx.sync()

    # the config-3 kernel (same generator as the oracle's orc_bench_kernel7, restated in numpy so the
    # timed path does not touch oracle/)
    def splitmix64(z):
        M = (1 << 64) - 1
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    k = np.array([int((splitmix64(0xF117E2D ^ i) >> 40) % 17) - 8 for i in range(49)], np.int8).reshape(7, 7)
    kp = k.ctypes.data_as(C.POINTER(C.c_int8))
    bs, bd = src.as_rcv(), dst.as_rcv()

    def step():
        rc = L.rcv_filter2d_i8_batch(ctx.handle, C.byref(bs), C.byref(bd), kp, 7, 6)
        if rc != 0:
            raise SystemExit(f"rcv_filter2d_i8_batch failed: {rc} {_ffi.strerror(rc)}")

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:   # the first collective builds the RCCL communicator (hundreds of ms): keep that out of the run-up below
        dist.barrier()
        torch.cuda.synchronize()
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < a.settle_ms:   # untimed: clocks settle under the real load
        for _ in range(8):
            step()
        ctx.sync()
    for _ in range(a.warmup):
        step()
    fence()
    ms_ev = C.c_float(0.0)
    t0 = time.perf_counter()
    L.rcv_timer_start(ctx.handle)            # hipEvent on the stream the kernel is launched on
    for _ in range(a.steps):
        step()
    L.rcv_timer_stop(ctx.handle, C.byref(ms_ev))  # records + synchronises the ctx stream
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed, ms_ev.value], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(t[0]), float(t[1])
    else:
        ev_ms = float(ms_ev.value)

    if rank == 0:
        px_per_step = total_frames * ROWS * COLS
        value = px_per_step * a.steps / elapsed / 1e6
        launch_ms = ev_ms / a.steps                      # one kernel launch per step
        alg_bytes = n * ROWS * COLS * ALG_BYTES_PER_PX   # per launch, per GPU
        ach = alg_bytes / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("filter2d_i8_7x7_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixels/sec on 4K 7x7 filter2D", "value": round(value, 1), "unit": "Mpix/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic (splitmix64 counter noise, generated on device)",
            "config": {"workload": "4K (3840x2160) u8 BGR 7x7 filter2D, integer weights >>6, batch=64 frames per GPU (BASELINE configs[2])",
                       "frames_per_gpu": n, "global_batch": total_frames, "parallelism": f"frame-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "filter2d_i8 7x7", "launch_ms": round(launch_ms, 4), "alg_bytes_per_launch": alg_bytes,
                         # context only: the guide's measured device-copy ceiling, 6.29 TB/s
                         "frac_of_copy_ceiling_6290": round(ach / 6290.0, 4)},
        }
        if not a.no_cpu and world == 1:   # the CPU baseline leg runs at N=1 only
            out["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    src.free()
    dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
