#!/usr/bin/env python3
"""bench.py -- the north-star measurement (BASELINE.json): 4K u8 BGR 7x7 filter2D, batches of 64 frames per launch.

    python bench.py [--gpus N] [--steps K] [--warmup W]          one process; N > 1: one host thread per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     one process per GPU (RCCL)
    python bench.py --config 4|5 ...     the two configs BASELINE.json names as 8-GPU as the headline, same line format:
        4: 8K BGR warpAffine (rotation 7 deg) + resize -> 1080p, 32 frames per launch (the fused launch
           rcv_warp_affine_resize_batch; --unfused: the two launches through an 8K intermediate)
        5: 4K cornerHarris pipeline (BGR -> gray -> Sobel -> response -> 3x3 NMS -> mask), 64 frames per launch

What runs (config 3, the default).  A "step" is one rcv_filter2d_i8_batch of 64 device-resident 3840x2160 BGR frames (integer 7x7 kernel of
SURVEY.md 8(d), >>6, saturate) on ONE context per GPU: BASELINE's literal batch.  Round 6: the library runs such a call as two 32-frame launches on
the context's two streams and joins nothing per call (the halves of consecutive calls hide each other's launch tails; every other entry point joins
first), so one context now gives what rounds 4-5 needed two batches in flight for; `--in-flight 2` is that recipe: F contexts on the one device
(rcv_group_create with a repeated ordinal), each with its OWN 64-frame source and destination, one launch per context and step -- the reference's
caller is a frame STREAM (rustcv/src/videoio/mod.rs:168-265 feeding examples/camera_demo.rs:50-76), so consecutive batches are independent.
`value` = all pixels of all launches / wall-clock.  Frames are generated ON DEVICE before the timed region (no PCIe in `value`).  Per-GPU work is
fixed (weak scaling): rank r owns frames [64 F r, 64 F (r + 1)), its context j the 64 from 64 (F r + j); frames are independent, so there is no
data-path collective -- only a barrier and the max over ranks of the elapsed time (RCCL under torch.distributed.run, a thread barrier in the
one-process form).  `--gpus N` on a node with fewer GPUs fails.

Prints ONE JSON line on rank 0 with the contract keys plus
  "roofline":     the dominant kernel's algorithmic HBM bytes per launch / launch_ms, against the 8 TB/s HBM3E peak.
                  launch_ms = HIP events on the kernels' own streams around >= 400 back-to-back steps, divided by the number of
                  LAUNCHES in the window (F per step): the SUSTAINED time per 64-frame launch with F in flight.  With F > 1 the
                  kernels overlap, so ONE kernel's own duration (what rocprofv3 reports) is about F x launch_ms; the figure is
                  bytes moved / wall time (profiles/r04_inflight_kernel_trace.txt shows the overlap).  single_stream_one_launch_* = the
                  call forced into ONE launch (RCV_FR_SPLIT=0: the figure of rounds 1-5, and what rocprofv3's per-kernel duration compares
                  with); in_flight2_* = two 64-frame batches in flight on two contexts (the headline of rounds 4-5).  copy_ceiling_gbs = the best plain device copy of the
                  same 2 x 1.59 GB measured in this run; memory_only_gbs = the kernel's own loads and stores with nothing in
                  between; shader_mhz_under_load = the shader clock sampled while launches run (a separate window)
  "verified_frames": frames of the LAST timed launches' outputs (every context) compared bit for bit with the CPU oracle
                  (mismatch: exit 1)
  "other_configs": (default run at N = 1 only) compact records measured in the same process, one stream each:
                  "3s" / "3f" the Sobel half of BASELINE configs[2] (rcv_sobel_batch on a BGR batch; the whole config as ONE launch,
                  rcv_filter2d_i8_sobel_batch), "4" and "5" the two 8-GPU configs at their per-GPU batch:
                  {value, ms_per_step, roofline{frac, kernel, launch_ms, alg_bytes_per_launch, traffic, in_flight2_launch_ms, in_flight2_frac},
                  verified, cpu_baseline}  (in_flight2_*: for information, the same launch with a second batch in flight on a second context)
  roofline.single_stream_{launch_ms, achieved, frac}: flat copies of the headline's own figures (F = 1), or the one-context measurement beside an
                  --in-flight 2 headline (then also "value_single_stream")
  "cpu_baseline": the C oracle (a port: C restatement, the Rust reference cannot be built here) timed on this box's host
                  cores on a bounded sample of the same workload.
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

# The BASELINE.json configs this file can run.  px = the pixels `value` counts per frame; alg_bytes = SURVEY.md 8(d)'s algorithmic
# bytes per frame of the dominant kernel.
CONFIGS = {
    3: {"metric": "Mpixels/sec on 4K 7x7 filter2D", "batch": 64, "px": ROWS * COLS, "alg_bytes": ROWS * COLS * 6, "dtype": "u8", "bound": "hbm",
        "workload": "4K (3840x2160) u8 BGR 7x7 filter2D, integer weights >>6, batches of 64 frames (BASELINE configs[2])"},
    # fused warp -> exact 4x down-scale: the centre 2x2 warped pixels of each 4x4 block tap a 3x3 source block (27 B) + 3 B written
    # per OUTPUT pixel (DESIGN_HISTORY.md 4).  `value` counts the 1080p OUTPUT pixels the launch produces ("quote Mpix/s on
    # output pixels and, separately, input pixels"); config.input_mpix_s is the 8K source-pixel rate
    4: {"metric": "Mpixels/sec (1080p output pixels) on 8K warpAffine + resize->1080p", "batch": 32, "px": 1080 * 1920, "alg_bytes": 1080 * 1920 * 30,
        "dtype": "f32 bilinear on u8", "bound": "hbm",
        "workload": "8K (7680x4320) u8 BGR warpAffine (bilinear, rotation 7 deg + translation, constant border) + resize -> 1080p, batches of 32 frames "
                    "(BASELINE configs[3]: 256 frames over 8 GPUs)"},
    5: {"metric": "Mpixels/sec on 4K cornerHarris pipeline", "batch": 64, "px": ROWS * COLS, "alg_bytes": ROWS * COLS * 4, "dtype": "i16 / i32 / f32 on u8", "bound": "hbm",
        "workload": "4K (3840x2160) u8 BGR cornerHarris pipeline (cvtColor -> Sobel -> response, blockSize 2, k 0.04 -> 3x3 NMS -> u8 mask), batches of 64 frames "
                    "(BASELINE configs[4]: 512 frames over 8 GPUs)"},
}
# the Sobel half of BASELINE configs[2] ("7x7 filter2D + Sobel gradient"), records of the default run only (other_configs["3s"] / ["3f"]):
#  "3s": rcv_sobel_batch on a 4K BGR batch (gray conversion fused: 3 B read + 2 x 2 B of i16 gradients written per pixel) -- the second launch of the two-call form;
#  "3f": rcv_filter2d_i8_sobel_batch, the whole config in ONE launch (7x7 filter -> gray -> Sobel; the filtered image never exists in HBM)
CONFIGS["3s"] = {"metric": "Mpixels/sec on 4K Sobel gradient (BGR source)", "batch": 64, "px": ROWS * COLS, "alg_bytes": ROWS * COLS * 7, "dtype": "i16 on u8", "bound": "hbm",
                 "workload": "4K (3840x2160) u8 BGR -> gray -> Sobel 3x3 dx, dy (i16), batches of 64 frames (the Sobel half of BASELINE configs[2])"}
CONFIGS["3f"] = {"metric": "Mpixels/sec on 4K 7x7 filter2D + Sobel gradient, one launch", "batch": 64, "px": ROWS * COLS, "alg_bytes": ROWS * COLS * 7, "dtype": "u8 / i16",
                 "bound": "hbm", "workload": "4K (3840x2160) u8 BGR 7x7 filter2D (integer weights >>6) -> gray -> Sobel dx, dy (i16) fused in one launch, batches of 64 "
                                             "frames (BASELINE configs[2] whole)"}
SEEDS = {3: 0x5EED0003, 4: 0x5EED0004, 5: 0x5EED0005, "3s": 0x5EED0003, "3f": 0x5EED0003}
HARRIS_THR = 1e-4
TRAFFIC_KEYS = {3: "filter2d_i8_7x7_hbm_bytes_per_launch", 4: "warp_resize_fused_hbm_bytes_per_launch", 5: "harris_pipeline_hbm_bytes_per_launch",
                "3s": "sobel_bgr_hbm_bytes_per_launch", "3f": "filter2d_sobel_fused_hbm_bytes_per_launch"}


def warp_matrix():
    """rotation by 7 degrees about the centre of the 8K frame + (13.25, -8.5), dst -> src"""
    import numpy as np
    t = np.deg2rad(7.0)
    c, s_, cx, cy = np.cos(t), np.sin(t), 7680 / 2, 4320 / 2
    return np.array([c, -s_, cx - c * cx + s_ * cy + 13.25, s_, c, cy - s_ * cx - c * cy - 8.5], np.float32)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=(3, 4, 5), help="BASELINE.json config (default 3: the north star)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="batches in flight per GPU = contexts (streams) per GPU, each with its own buffers (default: 2 for config 3, else 1)")
    ap.add_argument("--unfused", action="store_true", help="config 4: warpAffine and resize as two launches through an 8K intermediate")
    ap.add_argument("--batch", type=int, default=0, help="frames per launch (default: the config's per-GPU batch, 64 / 32 / 64)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the timed launch's output")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the in-run copy ceiling and the memory-only variant")
    ap.add_argument("--no-probe", action="store_true", help="skip the shader-clock probe")
    ap.add_argument("--no-others", action="store_true", help="skip the compact config-4 / config-5 records of the default run")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-time budget of the cpu_baseline sample")
    ap.add_argument("--other-cpu-seconds", type=float, default=2.5, help="CPU-time budget of the cpu_baseline sample of each other_configs record")
    ap.add_argument("--family", type=int, default=0, help="synthetic family: 0 noise (default), 1 scene")
    ap.add_argument("--sustained", type=int, default=400, help="steps of the sustained roofline window (at least --steps)")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed run-up of the same step before the W warmup steps: the GPU needs tens of ms of load to reach its "
                         "sustained clocks")
    ap.add_argument("--devices", default="",
                    help="TEST ONLY: explicit device ordinal per rank, e.g. 0,0 = two ranks time-sharing GPU 0 (exercises the N > 1 code on a "
                         "one-GPU box; the line then says n_gpus = the number of DISTINCT devices and contexts = the ranks)")
    ap.add_argument("--fail-rank", type=int, default=-1, help="TEST ONLY: this rank raises after the warmup (a failing rank must end the job non-zero, not hang it)")
    ap.add_argument("--dist-backend", default="nccl", help="TEST ONLY: torch.distributed backend of the one-process-per-GPU form (nccl = RCCL)")
    a = ap.parse_args(argv)
    if a.batch <= 0:
        a.batch = CONFIGS[a.config]["batch"]
    if a.in_flight <= 0:
        a.in_flight = 1   # (round 6: BASELINE's literal batch -- one context; rounds 4-5 ran config 3 with two batches in flight: --in-flight 2)
    a.device_list = [int(d) for d in a.devices.split(",")] if a.devices else None
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
    if cfg == "3s":
        return orc.sobel(orc.bgr2gray(frame))                                              # (dx, dy)
    if cfg == "3f":
        return orc.sobel(orc.bgr2gray(orc.filter2d_i8(frame, orc.bench_kernel7(), 6)))
    return orc.harris_pipeline(frame, 2, 0.04, HARRIS_THR)


def synth_args(cfg, family):
    """(rows, cols, family, seed) of the config's source frames: noise for the filters, the scene family (real corners) for Harris"""
    if cfg in ("3s", "3f"):
        return ROWS, COLS, 0, SEEDS[cfg]
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
            5: "4K BGR frame(s), Harris pipeline", "3s": "4K BGR frame(s), BGR2GRAY + Sobel (two oracle passes)",
            "3f": "4K BGR frame(s), 7x7 i8 filter2D + BGR2GRAY + Sobel (three oracle passes)"}[cfg]
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

    def __call__(self, lanes, torch, device):
        lanes.sync()                      # every context of this rank idle
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


class Lane:
    """one context (= one HIP stream) of a GPU with its own source / destination batch of config `cfg`; step() enqueues one launch"""

    def __init__(self, a, cfg, ctx, frame_base, n=None, unfused=False):
        from rustcv_amd import _ffi, device as dev
        self.cfg, self.ctx, self.base = cfg, ctx, frame_base
        self.n = n = n or a.batch
        rows, cols, fam, seed = synth_args(cfg, a.family)
        self.synth = (rows, cols, fam, seed)
        self.src = dev.DeviceBatch(ctx, n, rows, cols, CH)
        dev.synth(self.src, fam, seed, frame_base)
        self.mid = None
        L = _ffi.lib()
        if cfg == 3:
            self.dst = dst = dev.DeviceBatch(ctx, n, ROWS, COLS, CH)
            self.k = k = bench_kernel7()
            kp = k.ctypes.data_as(C.POINTER(C.c_int8))
            self.bs, self.bd = bs, bd = self.src.as_rcv(), dst.as_rcv()
            h = ctx.handle

            def step():
                rc = L.rcv_filter2d_i8_batch(h, C.byref(bs), C.byref(bd), kp, 7, 6)
                if rc != 0:
                    raise SystemExit(f"rcv_filter2d_i8_batch failed: {rc} {_ffi.strerror(rc)}")
        elif cfg == 4:
            self.dst = dst = dev.DeviceBatch(ctx, n, 1080, 1920, CH)
            M = warp_matrix()
            src = self.src
            if unfused:
                self.mid = mid = dev.DeviceBatch(ctx, n, 4320, 7680, CH)

                def step():
                    dev.warp_affine(src, mid, M)
                    dev.resize(mid, dst)
            else:
                def step():
                    dev.warp_affine_resize(src, dst, M, 4320, 7680)
        elif cfg in ("3s", "3f"):
            from rustcv_amd._ffi import RCV_16S
            self.dst = dx = dev.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_16S)
            self.mid = dy = dev.DeviceBatch(ctx, n, ROWS, COLS, 1, RCV_16S)   # (second output: freed with the lane)
            dy.memset(0)
            src, k = self.src, bench_kernel7()
            if cfg == "3s":
                def step():
                    dev.sobel(src, dx, dy)
            else:
                def step():
                    dev.filter2d_sobel(src, dx, dy, k, 6)
        else:
            self.dst = dst = dev.DeviceBatch(ctx, n, ROWS, COLS, 1)
            src = self.src

            def step():
                dev.harris_pipeline(src, dst, None, 2, 0.04, HARRIS_THR)
        self.step = step
        self.dst.memset(0)
        ctx.sync()

    def verify(self, orc):
        """frames first / middle / last of this lane's LAST launch against the oracle: (checked frame numbers, mismatching ones)"""
        import numpy as np
        rows, cols, fam, seed = self.synth
        n = self.n
        frames = sorted({0, n // 2 - 1 if n > 1 else 0, n - 1})
        bad = []
        for i in frames:
            got = self.dst.download_frame(i)
            want = oracle_step(orc, self.cfg, orc.synth_frame(rows, cols, CH, fam, seed, self.base + i))
            if self.cfg in ("3s", "3f"):     # two outputs: (dx, dy)
                ok = np.array_equal(got, want[0]) and np.array_equal(self.mid.download_frame(i), want[1])
            else:
                ok = np.array_equal(got, want)
            if not ok:
                bad.append(self.base + i)
        return [self.base + i for i in frames], bad

    def free(self):
        for b in (self.src, self.dst, self.mid):
            if b is not None:
                b.free()


def settle(ms, step, sync):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:   # untimed: clocks settle under the real load
        for _ in range(8):
            step()
        sync()


def run_rank(a, rank, world, device, fence, torch):
    """everything one GPU does; returns its measurements"""
    from rustcv_amd import _ffi, multigpu

    L = _ffi.lib()
    n, cfg, F = a.batch, a.config, a.in_flight
    lanes = multigpu.NativeGroup.in_flight(device, max(F, 2))   # contexts on this GPU (rcv_group_create, repeated ordinal): F of them carry a lane, the second one
                                                                # serves the "two batches in flight" legs of the default run
    f0 = rank * F * n                                      # this rank's contiguous frame range [f0, f0 + F n): shard.frame_range(world F n, rank, world)
    lane = [Lane(a, cfg, lanes.ctxs[j], f0 + j * n, unfused=a.unfused) for j in range(F)]
    ctx0 = lanes.ctxs[0]

    def step():                    # one launch on every context, enqueued by this one host thread
        for ln in lane:
            ln.step()

    def timed_group(steps):        # events on every context's stream: ms from the first start to the latest stop
        lanes.timer_start()
        for _ in range(steps):
            step()
        return lanes.timer_stop()

    def timed0(launches, fn=None):   # one stream alone: hipEvents on context 0's stream
        fn = fn or lane[0].step
        ms = C.c_float(0.0)
        L.rcv_timer_start(ctx0.handle)
        for _ in range(launches):
            fn()
        L.rcv_timer_stop(ctx0.handle, C.byref(ms))   # records + synchronises the ctx stream
        return float(ms.value)

    L.rcv__debug_kernels_reset()
    lane[0].step()
    lanes.sync()
    kernel_name = L.rcv__debug_kernels().decode()
    settle(a.settle_ms, step, lanes.sync)
    for _ in range(a.warmup):
        step()
    if rank == a.fail_rank:
        raise RuntimeError(f"--fail-rank {rank}: injected failure")
    fence(lanes, torch, device)
    t0 = time.perf_counter()
    ev_ms = timed_group(a.steps)
    fence(lanes, torch, device)
    elapsed = time.perf_counter() - t0
    res = {"elapsed": elapsed, "ev_ms_steps": ev_ms / a.steps, "kernel": kernel_name, "in_flight": F, "device": int(device)}

    # ---- roofline window: sustained clocks.  No idle gap before it (the timed steps just ran), >= 400 steps back to back ----
    ns = max(a.sustained, a.steps)
    res["launch_ms"] = timed_group(ns) / (ns * F)        # per 64-frame launch with F in flight (bytes moved / wall time)
    res["launches_sustained"] = ns * F
    if F > 1:                                            # the same call alone on ONE context (BASELINE's literal batch): the library's default form
        settle(30.0, lane[0].step, lanes.sync)
        res["single_launch_ms"] = timed0(ns) / ns
    if cfg == 3 and world == 1:
        # ... with the call forced into ONE launch (round 6: by default a filter2D call of 16+ frames runs as two halves on the context's two streams
        # while no other context of the device is busy; a process-wide knob, so this leg runs at N = 1 only)
        prev_knob = os.environ.get("RCV_FR_SPLIT")
        os.environ["RCV_FR_SPLIT"] = "0"
        L.rcv__debug_reload_knobs()
        settle(30.0, lane[0].step, lanes.sync)
        res["single_one_launch_ms"] = timed0(ns) / ns
        if prev_knob is None:
            os.environ.pop("RCV_FR_SPLIT", None)
        else:
            os.environ["RCV_FR_SPLIT"] = prev_knob   # (a profiling run may have set it for the whole process)
        L.rcv__debug_reload_knobs()
        if F == 1 and not a.no_others:
            # ... and with a second batch in flight on a second context of the device (the headline of rounds 4-5), for information
            ln2 = Lane(a, cfg, lanes.ctxs[1], f0 + n, unfused=a.unfused)

            def both():
                lane[0].step()
                ln2.step()
            settle(60.0, both, lanes.sync)
            lanes.timer_start()
            for _ in range(ns // 2):
                both()
            res["in_flight2_launch_ms"] = lanes.timer_stop() / (2 * (ns // 2))
            ln2.free()
            settle(30.0, step, lanes.sync)
    # the same step after an idle gap, over 20 steps: what a short window sees (clocks coming back up) -- for comparison only
    lanes.sync()
    time.sleep(0.25)
    res["launch_ms_first20"] = timed_group(20) / (20 * F)
    if not a.no_probe:
        # shader clock while launches run: a one-wave kernel on the side stream, in its OWN window (the probe blocks the host for 20 ms)
        BL = _ffi.bench_lib()
        settle(30.0, step, lanes.sync)
        mhz = C.c_float(0.0)
        for _ in range(60):
            step()
        BL.rcv__clock_probe(ctx0.handle, 20000, C.byref(mhz))
        lanes.sync()
        res["shader_mhz_under_load"] = round(float(mhz.value), 1)

    # ---- the benchmarked launches' bytes against the oracle (outside every timed region) ----
    if not a.no_verify:
        from oracle import pyoracle as orc   # the checker, never the thing measured
        res["verified_frames"], res["mismatched_frames"] = [], []
        for ln in lane:
            ok, bad = ln.verify(orc)
            res["verified_frames"] += ok
            res["mismatched_frames"] += bad

    # ---- ceiling of this box in this run (config 3): plain device copies of one lane's buffers and the kernel's own memory-only
    # ---- variant (its loads and its stores with nothing in between), one stream; dst is scratch from here on ----
    if not a.no_ceiling and cfg == 3:
        BL = _ffi.bench_lib()   # copy / store / clock probes: librustcv_hip_bench.so, not part of the product library
        nbytes = n * ROWS * COLS * CH
        src, dst = lane[0].src, lane[0].dst
        best, best_name = 0.0, None
        for variant, grid, name in CEILING_COPIES:
            def cp():
                rc = BL.rcv__membench(ctx0.handle, dst.ptr, src.ptr, nbytes, variant, grid)
                if rc != 0:
                    raise SystemExit(f"rcv__membench failed: {rc}")
            settle(60.0, cp, lanes.sync)      # run-up: the copies get the same warm clocks as the filter
            gbs = 2 * nbytes / (timed0(100, cp) / 100) / 1e6
            if gbs > best:
                best, best_name = gbs, name
        res["copy_ceiling_gbs"] = best
        res["copy_ceiling_kernel"] = best_name
        # the kernel's own memory-only variant (k_filter_rows_chain<.., 256> / k_filter_rows_mfma<.., 260>: the launch's loads and stores,
        # no arithmetic) lives in the measurement library, with the flag as an argument of the call: nothing process-wide to flip
        tune = _ffi.rows_tune(dbg=4)
        bs0, bd0, kp0 = lane[0].bs, lane[0].bd, lane[0].k.ctypes.data_as(C.POINTER(C.c_int8))

        def memonly():
            rc = BL.rcv__filter_rows_bench(ctx0.handle, C.byref(bs0), C.byref(bd0), kp0, 7, 6, tune, None)
            if rc != 0:
                raise SystemExit(f"rcv__filter_rows_bench failed: {rc}")
        settle(60.0, memonly, lanes.sync)
        res["memory_only_gbs"] = 2 * nbytes / (timed0(100, memonly) / 100) / 1e6
    for ln in lane:
        ln.free()

    # ---- compact records of the other two BASELINE configs (default run, one GPU): same process, one stream each ----
    if cfg == 3 and world == 1 and not a.no_others:
        res["other_configs"] = {"1": config1_record(a, ctx0), "2": config2_record(a, ctx0)}
        res["other_configs"].update({str(c): other_config(a, c, ctx0, lanes) for c in ("3s", "3f", 4, 5)})
    lanes.close()
    return res


def load_traffic(cfg):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json), or None"""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(tpath)).get(TRAFFIC_KEYS[cfg])
    except Exception:
        return None


def other_config(a, cfg, ctx, lanes, launches=100):
    """BASELINE config 4 or 5 at its per-GPU batch on one stream: settle, `launches` sustained launches between HIP events, first /
    middle / last frame against the oracle.  A compact record in the format of the headline's roofline."""
    from rustcv_amd import _ffi
    L = _ffi.lib()
    c = CONFIGS[cfg]
    ln = Lane(a, cfg, ctx, 0, n=c["batch"])
    L.rcv__debug_kernels_reset()
    ln.step()
    ctx.sync()
    kernel = L.rcv__debug_kernels().decode()
    settle(100.0, ln.step, ctx.sync)
    ms = C.c_float(0.0)
    t0 = time.perf_counter()
    L.rcv_timer_start(ctx.handle)
    for _ in range(launches):
        ln.step()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    wall = time.perf_counter() - t0
    launch_ms = float(ms.value) / launches
    alg = c["batch"] * c["alg_bytes"]
    ach = alg / (launch_ms * 1e-3) / 1e9
    rec = {"metric": c["metric"], "value": round(c["batch"] * c["px"] * launches / wall / 1e6, 1), "unit": "Mpix/s", "steps": launches,
           "ms_per_step": round(wall / launches * 1e3, 4), "frames_per_launch": c["batch"], "dtype": c["dtype"], "workload": c["workload"],
           "roofline": {"bound": c["bound"], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": load_traffic(cfg), "kernel": kernel, "launch_ms": round(launch_ms, 4), "launches": launches, "alg_bytes_per_launch": alg}}
    if cfg == 4:
        rec["input_mpix_s"] = round(c["batch"] * 4320 * 7680 * launches / wall / 1e6, 1)
        rec["path"] = "one fused launch (rcv_warp_affine_resize_batch)"
    if len(lanes.ctxs) > 1:
        # for information: the same launch with a second batch in flight on a second context of the device (what the headline does for
        # config 3); `frac` / `launch_ms` above stay the one-stream figures of the config's literal per-GPU batch
        ln2 = Lane(a, cfg, lanes.ctxs[1], c["batch"], n=c["batch"])

        def both():
            ln.step()
            ln2.step()
        settle(60.0, both, lanes.sync)
        lanes.timer_start()
        for _ in range(launches // 2):
            both()
        ms2 = lanes.timer_stop() / (2 * (launches // 2))
        rec["roofline"]["in_flight2_launch_ms"] = round(ms2, 4)
        rec["roofline"]["in_flight2_frac"] = round(alg / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        ln2.free()
    if cfg == 5:
        # the pipeline's time depends on corner density since the threshold-first NMS (rows without a candidate skip the 3x3 maxima): beside the
        # scene family above, the WORST case -- uniform noise with thr = -inf, every pixel of every row a candidate
        from rustcv_amd import device as dev
        noise = dev.DeviceBatch(ctx, c["batch"], ROWS, COLS, CH)
        dev.synth(noise, 0, SEEDS[5], 0)
        wmask = dev.DeviceBatch(ctx, c["batch"], ROWS, COLS, 1)

        def wstep():
            dev.harris_pipeline(noise, wmask, None, 2, 0.04, float("-inf"))
        settle(60.0, wstep, ctx.sync)
        L.rcv_timer_start(ctx.handle)
        for _ in range(launches):
            wstep()
        L.rcv_timer_stop(ctx.handle, C.byref(ms))
        w_ms = float(ms.value) / launches
        rec["roofline"]["worst_case_launch_ms"] = round(w_ms, 4)
        rec["roofline"]["worst_case_frac"] = round(alg / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        rec["roofline"]["worst_case"] = "noise family, thr = -inf: every row holds candidates (the figures above: scene family, thr 1e-4)"
        if not a.no_verify:
            import numpy as np
            from oracle import pyoracle as orc
            i = c["batch"] - 1
            okw = np.array_equal(wmask.download_frame(i), orc.harris_pipeline(orc.synth_frame(ROWS, COLS, CH, 0, SEEDS[5], i), 2, 0.04, float("-inf")))
            rec["roofline"]["worst_case_verified"] = "bit-exact vs the CPU oracle" if okw else "MISMATCH"
            if not okw:
                rec.setdefault("mismatched_extra", []).append(i)
        noise.free()
        wmask.free()
    if not a.no_verify:
        from oracle import pyoracle as orc
        ok, bad = ln.verify(orc)
        bad = bad + rec.pop("mismatched_extra", [])
        rec["verified_frames"] = ok
        rec["verified"] = "bit-exact vs the CPU oracle" if not bad else f"MISMATCH in frames {bad}"
        rec["mismatched_frames"] = bad
    ln.free()
    if not a.no_cpu:   # the CPU path timed beside every reported throughput (north_star): a short bounded sample of the same workload
        rec["cpu_baseline"] = cpu_baseline(a.other_cpu_seconds, cfg, a.family)
    return rec


def _event_us(ctx, fn, calls):
    """`calls` back-to-back enqueues of fn between HIP events on the context's stream -> microseconds per call (launch to launch)"""
    from rustcv_amd import _ffi
    L = _ffi.lib()
    settle(20.0, fn, ctx.sync)
    ms = C.c_float(0.0)
    L.rcv_timer_start(ctx.handle)
    for _ in range(calls):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return float(ms.value) * 1e3 / calls


def config1_record(a, ctx):
    """BASELINE configs[0]: 640x480 YUYV -> BGR cvtColor + imgproc::rectangle -- the one path the reference really runs
    (rustcv/src/videoio/mod.rs:201-258 -> :344-371, rustcv/src/imgproc/drawing.rs:67-106).  SURVEY.md 8(d): the CPU timing is this config's
    headline (restatement, one thread and all cores, 5 B/px); the GPU side is launch latency: 1.5 MB per frame sits in the caches."""
    import numpy as np
    from rustcv_amd import _ffi, device as dev, imgproc, videoio
    from rustcv_amd.core import Mat
    w, h = 640, 480
    rect, col, thick = imgproc.Rect(200, 150, 240, 240), imgproc.Scalar(0, 255, 0), 2
    rec = {"metric": "640x480 YUYV->BGR cvtColor + rectangle: CPU reference path, GPU latency beside it", "unit": "Mpix/s (CPU) / us per frame (GPU)",
           "workload": "640x480 YUYV -> BGR (BT.601 integer) + rectangle Rect(200,150,240,240) (0,255,0) thickness 2, one frame per call (BASELINE configs[0])",
           "dtype": "i32 on u8", "alg_bytes_per_px": 5}
    # GPU, device-resident: one frame per launch, two launches (cvtColor, rectangle)
    y = dev.DeviceBatch(ctx, 1, h, w, 2)
    o = dev.DeviceBatch(ctx, 1, h, w, 3)
    yuyv = np.random.default_rng(0x5EED0001).integers(0, 256, size=w * h * 2, dtype=np.uint8)   # full-range Y, U, V noise
    y.upload(yuyv.reshape(1, h, w, 2))

    def two():
        dev.cvt_color(y, o, _ffi.RCV_YUYV2BGR)
        dev.rectangle(o, rect, col, thick)
    us = _event_us(ctx, two, 400)
    us_cvt = _event_us(ctx, lambda: dev.cvt_color(y, o, _ffi.RCV_YUYV2BGR), 400)
    rec["gpu_device_resident"] = {"us_per_frame": round(us, 2), "us_cvt_color_alone": round(us_cvt, 2), "launches_per_frame": 2,
                                  "mpix_s": round(w * h / us, 1), "note": "launch-latency bound: no HBM fraction claimed"}
    two()
    got_dev = o.download_frame(0)
    # GPU, host Mats through the facade mirror (upload, two kernels, download, sync per call): what a drop-in VideoCapture::read costs
    m = Mat.empty()
    t0 = time.perf_counter()
    calls = 0
    while calls < 200 and time.perf_counter() - t0 < 1.0:
        videoio.decode_into(m, yuyv, videoio.YUYV, w, h, ctx)
        imgproc.rectangle(m, rect, col, thick, ctx)
        calls += 1
    rec["gpu_host_mat"] = {"us_per_frame": round((time.perf_counter() - t0) * 1e6 / calls, 1), "calls": calls,
                           "note": "host Mat in, host Mat out: two staged calls (H2D + kernel + D2H + sync each), PCIe-inclusive"}
    y.free()
    o.free()
    if not a.no_verify or not a.no_cpu:
        from oracle import pyoracle as orc   # the checker / the timed CPU restatement, never the GPU path
        want = np.zeros(w * h * 3, np.uint8)
        orc.yuyv_to_bgr(yuyv, want, w, h)
        orc.rectangle(want, h, w, w * 3, rect.x, rect.y, rect.width, rect.height, 0, 255, 0, thick)
        if not a.no_verify:
            ok = np.array_equal(got_dev.reshape(-1), want) and np.array_equal(m.data, want)
            rec["verified"] = "bit-exact vs the CPU oracle (device-resident and host-Mat paths)" if ok else "MISMATCH"
            rec["verified_frames"] = [0]
            rec["mismatched_frames"] = [] if ok else [0]
        if not a.no_cpu:
            # the restatement is the reference's scalar loop (one frame = one thread), so "all cores" = one frame stream per core, as a
            # capture server with several cameras would run it (OpenMP over frames / rows); ctypes releases the GIL
            import threading
            cores = orc.usable_cores()
            orc.set_threads(1)
            budget = a.other_cpu_seconds / 2

            def stream(count):
                out = np.zeros(w * h * 3, np.uint8)
                t1 = time.perf_counter()
                frames = 0
                while time.perf_counter() - t1 < budget:
                    orc.yuyv_to_bgr(yuyv, out, w, h)
                    orc.rectangle(out, h, w, w * 3, rect.x, rect.y, rect.width, rect.height, 0, 255, 0, thick)
                    frames += 1
                count.append((frames, time.perf_counter() - t1))
            one = []
            stream(one)
            every = []
            ths = [threading.Thread(target=stream, args=(every,)) for _ in range(cores)]
            t2 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            wall = time.perf_counter() - t2
            total = sum(f for f, _ in every)
            cpu = {"value": round(total * w * h / 1e6 / wall, 1), "unit": "Mpix/s", "cores": cores, "kind": "port",
                   "value_1thread": round(one[0][0] * w * h / 1e6 / one[0][1], 1),
                   "sample": f"{total} frames in {wall:.1f} s on {cores} threads (one frame stream per thread) and {one[0][0]} frames in {one[0][1]:.1f} s on one thread; "
                             "C restatement of the reference's scalar Rust loops (videoio/mod.rs:344-371, drawing.rs:67-106), gcc -O3 -march=native"}
            cpu["gbs_1thread"] = round(cpu["value_1thread"] * 5 / 1e3, 2)
            orc.set_threads(cores)
            rec["cpu_baseline"] = cpu
    return rec


def config2_record(a, ctx):
    """BASELINE configs[1]: ONE 1080p BGR frame, 5x5 integer GaussianBlur, one launch per call.  12.4 MB per frame live in the 256-MiB
    Infinity Cache and ~2 us at the HBM roofline, so this is a LATENCY config (no HBM fraction claimed): microseconds from
    launch to launch on one stream, with the floors of the same run beside it (an empty kernel; a plain copy of the frame)."""
    import numpy as np
    from rustcv_amd import _ffi, device as dev
    L, BL = _ffi.lib(), _ffi.bench_lib()
    rows, cols = 1080, 1920
    src = dev.DeviceBatch(ctx, 1, rows, cols, CH)
    dst = dev.DeviceBatch(ctx, 1, rows, cols, CH)
    dev.synth(src, 0, 0x5EED0002, 0)
    L.rcv__debug_kernels_reset()
    dev.gaussian_blur(src, dst, 5, 0.0)
    ctx.sync()
    kernel = L.rcv__debug_kernels().decode()
    us = min(_event_us(ctx, lambda: dev.gaussian_blur(src, dst, 5, 0.0), 1000) for _ in range(3))
    nbytes = rows * cols * CH

    def mb(variant, grid):
        def f():
            if BL.rcv__membench(ctx.handle, dst.ptr, src.ptr, nbytes, variant, grid) != 0:
                raise SystemExit("rcv__membench failed")
        return f
    floor_empty = min(_event_us(ctx, mb(30, 1024), 1000) for _ in range(2))
    floor_copy = min(min(_event_us(ctx, mb(v, g), 1000) for _ in range(2)) for v, g in ((10, 256), (0, 1)))
    dev.gaussian_blur(src, dst, 5, 0.0)
    rec = {"metric": "us per launch on 1080p u8 BGR 5x5 GaussianBlur, batch=1", "value": round(us, 2), "unit": "us", "higher_is_better": False,
           "mpix_s": round(rows * cols / us, 1), "dtype": "i32 on u8", "frames_per_launch": 1, "kernel": kernel,
           "workload": "1080p (1920x1080) u8 BGR 5x5 GaussianBlur (sigma 0: integer taps [1,4,6,4,1] (x) [1,4,6,4,1] / 256), one frame per launch (BASELINE configs[1])",
           "floors": {"empty_kernel_us": round(floor_empty, 2), "copy_of_the_frame_us": round(floor_copy, 2),
                      "note": "launch to launch on the same stream in the same run: an empty 1 024-workgroup kernel, the better of hipMemcpyAsync D2D and a sweep copy of the 6.2-MB frame"},
           "note": "latency config: the frame pair (12.4 MB) is cache-resident; no HBM fraction claimed"}
    if not a.no_verify:
        from oracle import pyoracle as orc
        ok = np.array_equal(dst.download_frame(0), orc.gaussian_blur(orc.synth_frame(rows, cols, CH, 0, 0x5EED0002, 0), 5, 0.0))
        rec["verified"] = "bit-exact vs the CPU oracle" if ok else "MISMATCH"
        rec["verified_frames"] = [0]
        rec["mismatched_frames"] = [] if ok else [0]
    if not a.no_cpu:
        from oracle import pyoracle as orc
        cores = orc.usable_cores()
        used = orc.set_threads(cores)
        frame = orc.synth_frame(rows, cols, CH, 0, 0x5EED0002, 0)
        orc.gaussian_blur(frame, 5, 0.0)
        t0 = time.perf_counter()
        frames = 0
        while time.perf_counter() - t0 < a.other_cpu_seconds:
            orc.gaussian_blur(frame, 5, 0.0)
            frames += 1
        dt = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": round(dt / frames * 1e6, 1), "unit": "us per frame", "mpix_s": round(frames * rows * cols / 1e6 / dt, 1), "cores": used, "kind": "port",
                               "sample": f"{frames} whole 1080p BGR frames, 5x5 integer Gaussian, gcc -O3 -march=native OpenMP over rows, {dt:.1f} s"}
    src.free()
    dst.free()
    return rec


def report(a, world, results):
    n = a.batch
    cfg = getattr(a, "config", 3)
    c = CONFIGS[cfg]
    F = results[0].get("in_flight", 1)
    total_frames = n * F * world
    elapsed = max(r["elapsed"] for r in results)
    launch_ms = max(r["launch_ms"] for r in results)
    px_per_step = total_frames * c["px"]
    value = px_per_step * a.steps / elapsed / 1e6
    alg_bytes = n * c["alg_bytes"]   # per launch
    unfused = cfg == 4 and getattr(a, "unfused", False)
    if unfused:
        alg_bytes = n * (4320 * 7680 * 6 + 1080 * 1920 * 15)   # upper bound of the warp (6 B per 8K px) + the exact-4x resize (15 B per output px)
    ach = alg_bytes / (launch_ms * 1e-3) / 1e9
    traffic = None if unfused else load_traffic(cfg)
    roof = {"bound": c["bound"], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": ("profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE of this kernel per launch, collected in a separate profiled run (not measured here)"
                               if traffic is not None else "no PMC pass recorded for this launch"),
            "kernel": results[0]["kernel"], "launch_ms": round(launch_ms, 4), "launches": results[0]["launches_sustained"], "in_flight": F,
            "launch_ms_first20": round(max(r["launch_ms_first20"] for r in results), 4),
            "launch_ms_timed_steps": round(max(r["ev_ms_steps"] for r in results) / F, 4),
            "alg_bytes_per_launch": alg_bytes}
    if F > 1:
        roof["launch_ms_note"] = (f"{F} launches in flight on {F} streams: launch_ms = sustained window / launches in it (bytes moved / wall time); one kernel's own "
                                  f"duration under rocprofv3 is about {F} x that because the kernels overlap")
    if "in_flight2_launch_ms" in results[0]:
        i_ms = max(r["in_flight2_launch_ms"] for r in results)
        roof["in_flight2_launch_ms"], roof["in_flight2_frac"] = round(i_ms, 4), round(alg_bytes / (i_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roof["in_flight2_note"] = "for information: two 64-frame batches in flight on two contexts of the device (the headline of rounds 4-5)"
    if F == 1 and "single_one_launch_ms" in results[0]:
        # one context IS the headline: the flat single_stream_* scalars of rounds 5-6 repeat it (a parser that reads them keeps working)
        roof["single_stream_launch_ms"], roof["single_stream_achieved"], roof["single_stream_frac"] = roof["launch_ms"], roof["achieved"], roof["frac"]
        o_ms = max(r["single_one_launch_ms"] for r in results)
        o_ach = alg_bytes / (o_ms * 1e-3) / 1e9
        roof["single_stream_one_launch_ms"], roof["single_stream_one_launch_achieved"], roof["single_stream_one_launch_frac"] = round(o_ms, 4), round(o_ach, 1), round(o_ach / HBM_PEAK_GBS, 4)
        roof["single_stream_note"] = ("the headline is one context with 64-frame calls back to back; single_stream_one_launch_* = the same with RCV_FR_SPLIT=0 (one launch per call: "
                                      "the figure of rounds 1-5; copy_ceiling_gbs and memory_only_gbs are one-launch measurements: compare them with it)")
    if "single_launch_ms" in results[0]:
        s_ms = max(r["single_launch_ms"] for r in results)
        s_ach = alg_bytes / (s_ms * 1e-3) / 1e9
        roof["single_stream"] = {"launch_ms": round(s_ms, 4), "achieved": round(s_ach, 1), "frac": round(s_ach / HBM_PEAK_GBS, 4)}
        # the same three as scalars (a parser that drops nested objects keeps them): BASELINE's literal "batch=64", one stream
        roof["single_stream_launch_ms"], roof["single_stream_achieved"], roof["single_stream_frac"] = round(s_ms, 4), round(s_ach, 1), round(s_ach / HBM_PEAK_GBS, 4)
        roof["single_stream_note"] = ("one context, 64-frame calls back to back.  Round 6: the library runs such a call as two 32-frame launches on the context's two streams "
                                      "(nothing joined per call; every other entry point joins first); single_stream_one_launch_* = the same with RCV_FR_SPLIT=0 (one launch per "
                                      "call: the figure of rounds 1-5).  copy_ceiling_gbs and memory_only_gbs are one-launch measurements: compare them with single_stream_one_launch_achieved")
        if "single_one_launch_ms" in results[0]:
            o_ms = max(r["single_one_launch_ms"] for r in results)
            o_ach = alg_bytes / (o_ms * 1e-3) / 1e9
            roof["single_stream_one_launch_ms"], roof["single_stream_one_launch_achieved"], roof["single_stream_one_launch_frac"] = round(o_ms, 4), round(o_ach, 1), round(o_ach / HBM_PEAK_GBS, 4)
    if "shader_mhz_under_load" in results[0]:
        roof["shader_mhz_under_load"] = min(r["shader_mhz_under_load"] for r in results)
    if "copy_ceiling_gbs" in results[0]:
        ceil = min(r["copy_ceiling_gbs"] for r in results)
        roof["copy_ceiling_gbs"] = round(ceil, 1)
        roof["copy_ceiling_kernel"] = results[0]["copy_ceiling_kernel"]
        roof["frac_of_copy_ceiling"] = round(ach / ceil, 4)
    if "memory_only_gbs" in results[0]:
        mo = min(r["memory_only_gbs"] for r in results)
        roof["memory_only_gbs"] = round(mo, 1)          # the kernel's own loads + stores, nothing in between, one stream
        roof["frac_of_memory_only"] = round(ach / mo, 4)
    if world > 1:
        roof["launch_ms_per_gpu"] = [round(r["launch_ms"], 4) for r in results]
    n_devices = len({r.get("device", i) for i, r in enumerate(results)})
    par = f"frame-sharded x{n_devices}, no collective"
    if F > 1:
        par += f"; {F} batches in flight per GPU ({F} contexts = HIP streams per device, own buffers each)"
    config = {"workload": c["workload"], "frames_per_launch": n, "launches_per_step_per_gpu": F, "frames_per_gpu": n * F, "global_batch": total_frames,
              "parallelism": par}
    if F == 1 and cfg == 3:
        config["call_form"] = ("a filter2D call of 16+ frames runs as two launches (halves of the batch) on the context's two streams, nothing joined per call: library "
                               "default while no other context of the device is busy; RCV_FR_SPLIT=0: one launch per call = roofline.single_stream_one_launch_*")
    if n_devices != world:   # (--devices 0,0: ranks time-sharing a device -- a test of the N > 1 code, not a scaling number)
        config["contexts"] = world
        config["note"] = f"{world} ranks on {n_devices} device(s): TEST of the multi-rank path, not a scaling measurement"
    if cfg == 4:
        config["path"] = "two launches (warpAffine, resize) through an 8K intermediate" if unfused else "one fused launch (rcv_warp_affine_resize_batch)"
        config["input_mpix_s"] = round(total_frames * 4320 * 7680 * a.steps / elapsed / 1e6, 1)
    out = {
        "metric": c["metric"], "value": round(value, 1), "unit": "Mpix/s",
        "n_gpus": n_devices, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": c["dtype"],
        "data": "synthetic (splitmix64 counter noise, generated on device)" if cfg != 5 else "synthetic (splitmix64 scene family: ramp + checkerboard + moving square, generated on device)",
        "config": config,
        "roofline": roof,
    }
    if "single_launch_ms" in results[0]:   # like-for-like with rounds 1-3 and BASELINE's batch=64: one 64-frame launch at a time per GPU
        out["value_single_stream"] = round(world * n * c["px"] / (max(r["single_launch_ms"] for r in results) * 1e-3) / 1e6, 1)
    bad = []
    if "verified_frames" in results[0]:
        out["verified_frames"] = sorted(f for r in results for f in r["verified_frames"])
        bad = sorted(f for r in results for f in r["mismatched_frames"])
        out["verified"] = "bit-exact vs the CPU oracle" if not bad else f"MISMATCH in frames {bad}"
    if "other_configs" in results[0]:
        out["other_configs"] = results[0]["other_configs"]
        for rec in out["other_configs"].values():
            bad += rec.pop("mismatched_frames", [])
    return out, bad


def main():
    a = parse()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    under_launcher = "RANK" in os.environ
    if under_launcher and env_world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={env_world}")
    if a.device_list is not None and len(a.device_list) != a.gpus:
        raise SystemExit(f"--devices names {len(a.device_list)} ranks but --gpus is {a.gpus}")

    import torch  # first: librustcv_hip.so then binds to the HIP runtime torch already loaded
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback)")
    import rustcv_amd as rcv

    if under_launcher:
        # ---- one process per GPU (torch.distributed.run): RCCL carries the barrier and the max over ranks ----
        import torch.distributed as dist
        rank, local, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0")), env_world
        if a.device_list is not None:
            local = a.device_list[rank]
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)
        dist.barrier()   # the first collective builds the RCCL communicator (hundreds of ms): keep that out of the run-up
        torch.cuda.synchronize()
        res = run_rank(a, rank, world, local, Fence(dist=dist), torch)
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
        sys.exit(1 if bad else 0)

    # ---- one process: one host thread per GPU (rustcv_amd.multigpu.DeviceGroup supplies threads + barrier) ----
    have = rcv.device_count()
    devices = a.device_list if a.device_list is not None else list(range(a.gpus))
    if a.gpus < 1 or max(devices) >= have:
        raise SystemExit(f"--gpus {a.gpus}: this node exposes {have} GPU(s) to this process")
    world = a.gpus
    group = rcv.DeviceGroup(devices)
    # the group's OWN barrier: DeviceGroup.run aborts it when a rank raises, so the other ranks leave the fence instead of hanging
    fence = Fence(tbarrier=group.barrier if world > 1 else None)
    results = group.run(lambda r, ctx: run_rank(a, r, world, group.devices[r], fence, torch))
    out, bad = report(a, world, results)
    if not a.no_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(a.cpu_seconds, a.config, a.family)
    print(json.dumps(out), flush=True)
    group.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
