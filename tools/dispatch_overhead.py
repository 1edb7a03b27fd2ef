#!/usr/bin/env python3
"""tools/dispatch_overhead.py -- what the HOST side of the multi-GPU dispatch costs, measured on ONE GPU.

`bench.py --gpus N` drives N contexts from N Python threads of one process (rustcv_amd.multigpu.DeviceGroup).  No 8-GPU node is
available to this build, but the host-side question -- do eight threads issuing ctypes calls through the GIL slow each other
down? -- does not need eight GPUs: G contexts ON DEVICE 0 (G streams), one host thread each, split the SAME total work (64 4K
frames of the north-star filter, 64 / G frames per context per step).  Reported per G:

  wall_ms_per_step   barrier-to-barrier wall time of K steps of all G threads / K    (G = 1: the single-context launch)
  vs_G1              the same / the G = 1 figure: GPU work is constant, so what is above 1.0 is dispatch + the smaller launches
  enqueue_us         host time of one rcv_filter2d_i8_batch call (enqueue only, no sync), mean over threads: the GIL-contended
                     cost of a call; a GPU step takes >= 70 us even at 8 frames, so a thread keeps its GPU busy while enqueue_us
                     stays far below that
Two work splits:
  split   the SAME total work (64 frames) cut into G shards: 64 / G frames per context and launch.  What is above 1.0 here is
          host dispatch PLUS the cost of smaller launches (an 8-frame launch has its own fill and tail); the control row
          "G=1, 8 launches of 8 frames" separates the two.
  full    every context gets the FULL per-GPU batch (64 frames), as in the real 8-GPU run; the one GPU time-shares G times the
          work, so the ideal wall time is G x the G = 1 time.  vs_ideal > 1 is what the host side costs when every thread drives
          a full-size launch stream.
This is a host-dispatch measurement, NOT a scaling claim: everything runs on one device and the report says n_devices = 1.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ROWS, COLS = 2160, 3840


def summarize(raw, full=False):
    """raw: {G: {"wall_s": [..per repetition..], "steps": K, "enqueue_us": [..per thread..]}} -> report rows (pure arithmetic: CPU-tested).
    full: every context ran the full batch, so the ideal wall time of G contexts is G x the G = 1 time."""
    rows, base = [], None
    for G in sorted(raw):
        r = raw[G]
        wall = sorted(r["wall_s"])[len(r["wall_s"]) // 2] / r["steps"] * 1e3
        if G == 1:
            base = wall
        ideal = base * (G if full else 1) if base else None
        rows.append({"contexts": G, "frames_per_context": r["frames"], "wall_ms_per_step": round(wall, 4),
                     "vs_G1": round(wall / ideal, 4) if ideal else None,
                     "enqueue_us": round(sum(r["enqueue_us"]) / len(r["enqueue_us"]), 2), "enqueue_us_max": round(max(r["enqueue_us"]), 2)})
    return {"n_devices": 1, "split": "full batch per context (ideal = G x the G=1 time)" if full else "same total work cut into G shards",
            "note": "G contexts on device 0, one host thread each; host-dispatch cost only", "rows": rows,
            "loss_at_max_G": round(rows[-1]["vs_G1"] - 1.0, 4) if rows and rows[-1]["vs_G1"] is not None else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--total", type=int, default=64, help="frames of the whole job (split) / per context (full)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dispatch_overhead.json"))
    a = ap.parse_args()
    import rustcv_amd as rcv
    from rustcv_amd import _ffi, device, shard
    from bench import bench_kernel7
    L = _ffi.lib()
    k = bench_kernel7()
    kp = k.ctypes.data_as(C.POINTER(C.c_int8))

    def measure(G, frames_of, launches_per_step=1, steps=a.steps):
        """G contexts on device 0; context r filters frames_of(r) frames per launch, launches_per_step launches per step"""
        group = rcv.DeviceGroup([0] * G)
        bufs = []
        for r in range(G):
            nf = frames_of(r)
            s = device.DeviceBatch(group.ctxs[r], nf, ROWS, COLS, 3)
            d = device.DeviceBatch(group.ctxs[r], nf, ROWS, COLS, 3)
            device.synth(s, 0, 0x5EED0003, 64 * r)
            bufs.append((s, d, s.as_rcv(), d.as_rcv()))
        group.sync()
        walls, enq = [], [0.0] * G
        bar = threading.Barrier(G)

        def body(r, ctx):
            s, d, bs, bd = bufs[r]
            for _ in range(20):
                L.rcv_filter2d_i8_batch(ctx.handle, C.byref(bs), C.byref(bd), kp, 7, 6)
            ctx.sync()
            bar.wait()
            t0 = time.perf_counter()
            for _ in range(steps * launches_per_step):
                L.rcv_filter2d_i8_batch(ctx.handle, C.byref(bs), C.byref(bd), kp, 7, 6)
            t1 = time.perf_counter()
            ctx.sync()
            bar.wait()
            t2 = time.perf_counter()
            return t2 - t0, (t1 - t0) / (steps * launches_per_step) * 1e6

        for rep in range(a.reps + 1):
            res = group.run(body)
            if rep == 0:
                continue   # warm-up repetition
            walls.append(max(x[0] for x in res))
            enq = [x[1] for x in res]
        for s, d, _, _ in bufs:
            s.free()
            d.free()
        group.close()
        return {"wall_s": walls, "steps": steps, "enqueue_us": enq, "frames": frames_of(0)}

    def show(out):
        print(out["split"])
        for r in out["rows"]:
            print(f"  G={r['contexts']}  {r['frames_per_context']:2d} frames per context  {r['wall_ms_per_step']:.4f} ms per step  x{r['vs_G1']:.4f} of the ideal   "
                  f"enqueue {r['enqueue_us']:.1f} us per call (max {r['enqueue_us_max']:.1f})", flush=True)
        print(f"  loss at G=8: {out['loss_at_max_G'] * 100:+.2f} %")

    split = summarize({G: measure(G, lambda r, G=G: shard.frame_range(a.total, r, G)[1] - shard.frame_range(a.total, r, G)[0]) for G in (1, 2, 4, 8)})
    show(split)
    ctl = measure(1, lambda r: a.total // 8, launches_per_step=8)
    ctl_ms = sorted(ctl["wall_s"])[len(ctl["wall_s"]) // 2] / ctl["steps"] * 1e3
    split["control_G1_8_launches_of_8_frames_ms_per_step"] = round(ctl_ms, 4)
    split["control_vs_G1"] = round(ctl_ms / split["rows"][0]["wall_ms_per_step"], 4)
    print(f"  control: ONE context, 8 launches of {a.total // 8} frames per step: {ctl_ms:.4f} ms per step = x{split['control_vs_G1']:.4f} of one 64-frame launch "
          f"(the launch-granularity part of the split's loss)")
    full = summarize({G: measure(G, lambda r: a.total, steps=max(20, a.steps // G)) for G in (1, 2, 4, 8)}, full=True)
    show(full)
    print("n_devices = 1: a host-dispatch measurement, not a scaling number")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"n_devices": 1, "split": split, "full": full}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
