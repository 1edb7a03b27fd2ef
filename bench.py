#!/usr/bin/env python3
"""bench.py -- the north-star measurement (BASELINE.json): 4K u8 BGR 7x7 filter2D, batch 64 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]          one process; N > 1: one host thread + one context per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     one process per GPU (RCCL)
    python bench.py --config 4|5 ...     the two configs BASELINE.json names as 8-GPU, same line format, per-GPU batch fixed:
        4: 8K BGR warpAffine (rotation 7 deg) + resize -> 1080p, 32 frames per GPU (the fused launch rcv_warp_affine_resize_batch;
           --unfused: the two launches through an 8K intermediate)
        5: 4K cornerHarris pipeline (BGR -> gray -> Sobel -> response -> 3x3 NMS -> mask), 64 frames per GPU

A "step" is one pass of the hot path over one batch: (config 3, the default) rcv_filter2d_i8_batch on 64 device-resident
3840x2160 BGR frames (integer 7x7 kernel of SURVEY.md 8(d), >>6, saturate).  Frames are generated ON DEVICE before the timed region (no PCIe in
`value`).  Per-GPU work is fixed (weak scaling): rank r owns frames [64r, 64r+64) (rustcv_amd.shard.frame_range); frames are
independent, so there is no data-path collective -- only a barrier and the max over ranks of the elapsed time (RCCL under
torch.distributed.run, a thread barrier in the one-process form).  `--gpus N` on a node with fewer GPUs fails.

Prints ONE JSON line on rank 0 with the contract keys plus
  "roofline":     the dominant kernel's algorithmic HBM bytes / its SUSTAINED launch time -- HIP events on the kernel's own
                  stream around >= 400 back-to-back launches (launch_ms); the same after an idle gap over 20 launches
                  (launch_ms_first20: boost clocks) for comparison -- against the 8 TB/s HBM3E peak; copy_ceiling_gbs = the best
                  plain device copy of the same 2 x 1.59 GB measured in this run (the rate the memory system of THIS box gives);
                  memory_only_gbs = the kernel's own loads and stores with nothing in between (the ceiling of ITS access
                  pattern); shader_mhz_under_load = the shader clock sampled while the sustained launches run
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

# The BASELINE.json configs this file can run.  px = the pixels `value` counts per frame; alg_bytes = SURVEY.md 8(d)'s algorithmic
# bytes per frame of the dominant kernel.
CONFIGS = {
    3: {"metric": "Mpixels/sec on 4K 7x7 filter2D", "batch": 64, "px": ROWS * COLS, "alg_bytes": ROWS * COLS * 6, "dtype": "u8", "bound": "hbm",
        "workload": "4K (3840x2160) u8 BGR 7x7 filter2D, integer weights >>6, batch=64 frames per GPU (BASELINE configs[2])"},
    # fused warp -> exact 4x down-scale: the centre 2x2 warped pixels of each 4x4 block tap a 3x3 source block (27 B) + 3 B written
    # per OUTPUT pixel; `value` counts the 8K pixels of the warped image the launch stands for
    4: {"metric": "Mpixels/sec on 8K warpAffine + resize->1080p", "batch": 32, "px": 4320 * 7680, "alg_bytes": 1080 * 1920 * 30, "dtype": "f32 bilinear on u8", "bound": "hbm",
        "workload": "8K (7680x4320) u8 BGR warpAffine (bilinear, rotation 7 deg + translation, constant border) + resize -> 1080p, batch=32 frames per GPU "
                    "(BASELINE configs[3]: 256 frames over 8 GPUs)"},
    5: {"metric": "Mpixels/sec on 4K cornerHarris pipeline", "batch": 64, "px": ROWS * COLS, "alg_bytes": ROWS * COLS * 4, "dtype": "i16 / i32 / f32 on u8", "bound": "hbm",
        "workload": "4K (3840x2160) u8 BGR cornerHarris pipeline (cvtColor -> Sobel -> response, blockSize 2, k 0.04 -> 3x3 NMS -> u8 mask), batch=64 frames per GPU "
                    "(BASELINE configs[4]: 512 frames over 8 GPUs)"},
}
SEEDS = {3: 0x5EED0003, 4: 0x5EED0004, 5: 0x5EED0005}
HARRIS_THR = 1e-4


def warp_matrix():
    """rotation by 7 degrees about the centre of the 8K frame + (13.25, -8.5), dst -> src"""
    import numpy as np
    t = np.deg2rad(7.0)
    c, s_, cx, cy = np.cos(t), np.sin(t), 7680 / 2, 4320 / 2
    return np.array([c, -s_, cx - c * cx + s_ * cy + 13.25, s_, c, cy - s_ * cx - c * cy - 8.5], np.float32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json config (default 3: the north star)")
    ap.add_argument("--unfused", action="store_true", help="config 4: warpAffine and resize as two launches through an 8K intermediate")
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU (default: the config's per-GPU batch, 64 / 32 / 64)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the timed launch's output")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the in-run copy ceiling")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-time budget of the cpu_baseline sample")
    ap.add_argument("--family", type=int, default=0, help="synthetic family: 0 noise (default), 1 scene")
    ap.add_argument("--sustained", type=int, default=400, help="launches of the sustained roofline window (at least --steps)")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed run-up of the same step before the W warmup steps: the GPU needs tens of ms of load to reach its "
                         "sustained clocks")
    a = ap.parse_args()
    if a.batch <= 0:
        a.batch = CONFIGS[a.config]["batch"]
    return a


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


def oracle_step(orc, cfg, frame):
    """what one frame of config `cfg` must turn into, by the CPU oracle (the checker of the verification leg and the thing timed by
    the cpu_baseline leg -- never part of the measured GPU path)"""
    if cfg == 3:
        return orc.filter2d_i8(frame, orc.bench_kernel7(), 6)
    if cfg == 4:
        return orc.resize(orc.warp_affine(frame, warp_matrix(), 4320, 7680), 1080, 1920)
    return orc.harris_pipeline(frame, 2, 0.04, HARRIS_THR)


def synth_args(cfg, family):
    """(rows, cols, family, seed) of the config's source frames: noise for the filters, the scene family (real corners) for Harris"""
    if cfg == 4:
        return 4320, 7680, family, SEEDS[4]
    return ROWS, COLS, (1 if cfg == 5 else family), SEEDS[cfg]


def cpu_baseline(budget_s, cfg=3, family=0):
    """Oracle (port) on the host cores, all threads (OpenMP over rows), bounded sample of whole frames of the config."""
    from oracle import pyoracle as orc
    cores = orc.usable_cores()   # affinity capped by the cgroup CPU quota, not os.cpu_count()
    used = orc.set_threads(cores)
    rows, cols, fam, seed = synth_args(cfg, family)
    frame = orc.synth_frame(rows, cols, CH, fam, seed, 0)
    orc.filter2d_i8(frame[:256], orc.bench_kernel7(), 6)  # warm the thread pool
    t0 = time.perf_counter()
    frames = 0
    while True:
        oracle_step(orc, cfg, frame)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or frames >= 512:
            break
    mpix = frames * CONFIGS[cfg]["px"] / 1e6 / dt
    what = {3: "4K BGR frame(s), 7x7 i8 filter2D", 4: "8K BGR frame(s), warpAffine + resize -> 1080p (two oracle passes)",
            5: "4K BGR frame(s), Harris pipeline"}[cfg]
    out = {"value": round(mpix, 2), "unit": "Mpix/s", "cores": used, "kind": "port",
           "sample": f"{frames} whole {what}, gcc -O3 -march=native OpenMP over rows, {dt:.1f} s"}
    if cfg == 3:
        # one-thread figure on a smaller slab (rows are independent), for the record
        orc.set_threads(1)
        slab = frame[:270]
        t1 = time.perf_counter()
        orc.filter2d_i8(slab, orc.bench_kernel7(), 6)
        dt1 = time.perf_counter() - t1
        orc.set_threads(cores)
        out["value_1thread"] = round(270 * COLS / 1e6 / dt1, 2)
    return out


class Fence:
    """barrier + device sync on both sides of the timed region: RCCL barrier (one process per GPU) or a thread barrier
    (one process, one thread per GPU: the DeviceGroup's own barrier, which a failing rank aborts -- the others then leave with
    BrokenBarrierError instead of waiting for ever)"""

    def __init__(self, dist=None, tbarrier=None):
        self.dist, self.tb = dist, tbarrier

    def __call__(self, ctx, torch, device):
        ctx.sync()
        torch.cuda.synchronize(device)
        if self.dist is not None:
            self.dist.barrier()
        if self.tb is not None:
            self.tb.wait(timeout=600.0)
        torch.cuda.synchronize(device)


# plain device copies of the batch for the in-run ceiling (rcv__membench variant, workgroups, name).  Round 3 (tools/ablate_copy.py,
# profiles/r03_ablate_copy_shapes.txt): what a copy reaches depends on its shape by 20 % -- few workgroups with 2-8 accesses in
# flight per thread in ONE global sweep are the best on every box, so those are in the list.
CEILING_COPIES = ((0, 1, "hipMemcpyAsync D2D"), (1, 1024, "sweep g=1024"), (3, 512, "sweep U=4 nt g=512"), (5, 2048, "block nt g=2048"),
                  (9, 2048, "XCD-local sweep nt g=2048"), (21, 512, "sweep U=2 nt g=512"), (20, 768, "sweep U=2 nt stores g=768"),
                  (10, 256, "sweep U=4 plain g=256"), (17, 256, "sweep U=8 nt g=256"), (21, 384, "sweep U=2 nt g=384"),
                  # (the sweep cut into 2 / 8 regions: the best copies of tools/ablate_streams.py)
                  (40, 512 | (2 << 16), "2-region sweep U=2 nt g=512"), (40, 512 | (8 << 16), "8-region sweep U=2 nt g=512"))


def run_rank(a, rank, world, device, ctx, fence, torch):
    """everything one GPU does; returns its measurements"""
    import numpy as np
    from rustcv_amd import _ffi, device as dev, shard

    L = _ffi.lib()
    BL = _ffi.bench_lib()   # copy / store / clock probes: librustcv_hip_bench.so, not part of the product library
    n, cfg = a.batch, a.config
    total_frames = n * world
    f0, f1 = shard.frame_range(total_frames, rank, world)  # contiguous frame range of this rank
    assert f1 - f0 == n
    rows, cols, fam, seed = synth_args(cfg, a.family)
    src = dev.DeviceBatch(ctx, n, rows, cols, CH)
    dev.synth(src, fam, seed, f0)
    mid = None
    if cfg == 3:
        dst = dev.DeviceBatch(ctx, n, ROWS, COLS, CH)
        k = bench_kernel7()
        kp = k.ctypes.data_as(C.POINTER(C.c_int8))
        bs, bd = src.as_rcv(), dst.as_rcv()

        def step():
            rc = L.rcv_filter2d_i8_batch(ctx.handle, C.byref(bs), C.byref(bd), kp, 7, 6)
            if rc != 0:
                raise SystemExit(f"rcv_filter2d_i8_batch failed: {rc} {_ffi.strerror(rc)}")
    elif cfg == 4:
        dst = dev.DeviceBatch(ctx, n, 1080, 1920, CH)
        M = warp_matrix()
        if a.unfused:
            mid = dev.DeviceBatch(ctx, n, 4320, 7680, CH)

            def step():
                dev.warp_affine(src, mid, M)
                dev.resize(mid, dst)
        else:
            def step():
                dev.warp_affine_resize(src, dst, M, 4320, 7680)
    else:
        dst = dev.DeviceBatch(ctx, n, ROWS, COLS, 1)

        def step():
            dev.harris_pipeline(src, dst, None, 2, 0.04, HARRIS_THR)
    dst.memset(0)
    ctx.sync()

    def timed(launches, fn=step, probe_us=0):
        ms, mhz = C.c_float(0.0), C.c_float(0.0)
        L.rcv_timer_start(ctx.handle)            # hipEvent on the stream the kernel is launched on
        for _ in range(launches):
            fn()
        if probe_us:                             # shader clock while the queued launches run (a one-wave kernel on the side stream)
            BL.rcv__clock_probe(ctx.handle, probe_us, C.byref(mhz))
        L.rcv_timer_stop(ctx.handle, C.byref(ms))  # records + synchronises the ctx stream
        return (float(ms.value), float(mhz.value)) if probe_us else float(ms.value)

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
    ms, mhz = timed(ns, probe_us=20000)
    res["launch_ms"] = ms / ns
    res["launches_sustained"] = ns
    res["shader_mhz_under_load"] = round(mhz, 1)
    # the same launch after an idle gap, over 20 launches: what a short window sees (boost clocks) -- for comparison only
    ctx.sync()
    time.sleep(0.25)
    res["launch_ms_first20"] = timed(20) / 20

    # ---- the benchmarked launch's bytes against the oracle (outside every timed region) ----
    if not a.no_verify:
        from oracle import pyoracle as orc   # the checker, never the thing measured
        bad = []
        frames = sorted({0, n // 2 - 1 if n > 1 else 0, n - 1})
        for i in frames:
            got = dst.download_frame(i)
            want = oracle_step(orc, cfg, orc.synth_frame(rows, cols, CH, fam, seed, f0 + i))
            if not np.array_equal(got, want):
                bad.append(f0 + i)
        res["verified_frames"] = [f0 + i for i in frames]
        res["mismatched_frames"] = bad

    # ---- ceiling of this box in this run (config 3): plain device copies of the same buffers and the kernel's own memory-only
    # ---- variant (its loads and its stores with nothing in between); dst is scratch from here on ----
    if not a.no_ceiling and cfg == 3:
        nbytes = n * ROWS * COLS * CH

        def runup(fn):
            t_run = time.perf_counter()
            while (time.perf_counter() - t_run) * 1e3 < 60.0:   # run-up: the copies get the same warm clocks as the filter
                for _ in range(8):
                    fn()
                ctx.sync()
        best, best_name = 0.0, None
        for variant, grid, name in CEILING_COPIES:
            def cp():
                rc = BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, variant, grid)
                if rc != 0:
                    raise SystemExit(f"rcv__membench failed: {rc}")
            runup(cp)
            gbs = 2 * nbytes / (timed(100, cp) / 100) / 1e6
            if gbs > best:
                best, best_name = gbs, name
        res["copy_ceiling_gbs"] = best
        res["copy_ceiling_kernel"] = best_name
        L.rcv__debug_set(4)          # k_filter_rows_mfma<.., 260>: the launch's own loads and stores, no arithmetic
        runup(step)
        res["memory_only_gbs"] = 2 * nbytes / (timed(100) / 100) / 1e6
        L.rcv__debug_set(0)
    src.free()
    dst.free()
    if mid is not None:
        mid.free()
    return res


def report(a, world, results):
    n = a.batch
    cfg = getattr(a, "config", 3)
    c = CONFIGS[cfg]
    total_frames = n * world
    elapsed = max(r["elapsed"] for r in results)
    launch_ms = max(r["launch_ms"] for r in results)
    px_per_step = total_frames * c["px"]
    value = px_per_step * a.steps / elapsed / 1e6
    alg_bytes = n * c["alg_bytes"]   # per launch, per GPU
    unfused = cfg == 4 and getattr(a, "unfused", False)
    if unfused:
        alg_bytes = n * (4320 * 7680 * 6 + 1080 * 1920 * 15)   # upper bound of the warp (6 B per 8K px) + the exact-4x resize (15 B per output px)
    ach = alg_bytes / (launch_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get({3: "filter2d_i8_7x7_hbm_bytes_per_launch", 4: "warp_resize_fused_hbm_bytes_per_launch",
                                                  5: "harris_pipeline_hbm_bytes_per_launch"}[cfg] if not unfused else "-")
        except Exception:
            traffic = None
    roof = {"bound": c["bound"], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": "profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE of this kernel, collected in a separate profiled run (not measured here)",
            "kernel": results[0]["kernel"], "launch_ms": round(launch_ms, 4), "launches": results[0]["launches_sustained"],
            "launch_ms_first20": round(max(r["launch_ms_first20"] for r in results), 4),
            "launch_ms_timed_steps": round(max(r["ev_ms_steps"] for r in results), 4),
            "alg_bytes_per_launch": alg_bytes}
    if "shader_mhz_under_load" in results[0]:
        roof["shader_mhz_under_load"] = min(r["shader_mhz_under_load"] for r in results)
    if "copy_ceiling_gbs" in results[0]:
        ceil = min(r["copy_ceiling_gbs"] for r in results)
        roof["copy_ceiling_gbs"] = round(ceil, 1)
        roof["copy_ceiling_kernel"] = results[0]["copy_ceiling_kernel"]
        roof["frac_of_copy_ceiling"] = round(ach / ceil, 4)
    if "memory_only_gbs" in results[0]:
        mo = min(r["memory_only_gbs"] for r in results)
        roof["memory_only_gbs"] = round(mo, 1)          # the kernel's own loads + stores, nothing in between
        roof["frac_of_memory_only"] = round(ach / mo, 4)
    if world > 1:
        roof["launch_ms_per_gpu"] = [round(r["launch_ms"], 4) for r in results]
    config = {"workload": c["workload"], "frames_per_gpu": n, "global_batch": total_frames, "parallelism": f"frame-sharded x{world}, no collective"}
    if cfg == 4:
        config["path"] = "two launches (warpAffine, resize) through an 8K intermediate" if unfused else "one fused launch (rcv_warp_affine_resize_batch)"
        config["output_mpix_s"] = round(total_frames * 1080 * 1920 * a.steps / elapsed / 1e6, 1)
    out = {
        "metric": c["metric"], "value": round(value, 1), "unit": "Mpix/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": c["dtype"],
        "data": "synthetic (splitmix64 counter noise, generated on device)" if cfg != 5 else "synthetic (splitmix64 scene family: ramp + checkerboard + moving square, generated on device)",
        "config": config,
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
                out["cpu_baseline"] = cpu_baseline(a.cpu_seconds, a.config, a.family)
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
    # the group's OWN barrier: DeviceGroup.run aborts it when a rank raises, so the other ranks leave the fence instead of hanging
    fence = Fence(tbarrier=group.barrier if world > 1 else None)
    results = group.run(lambda r, ctx: run_rank(a, r, world, group.devices[r], ctx, fence, torch))
    out, bad = report(a, world, results)
    if not a.no_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(a.cpu_seconds, a.config, a.family)
    print(json.dumps(out), flush=True)
    group.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
