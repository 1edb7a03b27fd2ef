#!/usr/bin/env python3
"""tools/bench_ops.py -- every op of the hot path at the BASELINE.json configs, device-resident, one GPU.

For each op: ms per launch (HIP events on the context's stream), Mpix/s, algorithmic GB/s
(SURVEY.md 8(d) bytes per pixel), fraction of the 8 TB/s HBM peak, and -- with --cpu -- the C oracle on
the host cores on a bounded sample.  Writes one JSON document (default profiles/ops_bench.json) and a table.
bench.py stays the single-line north-star bench; this is the per-row evidence for SURVEY.md §8.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import rustcv_amd as rcv  # noqa: E402
from rustcv_amd import _ffi, device  # noqa: E402
from rustcv_amd.imgproc import Rect, Scalar  # noqa: E402

HBM = 8000.0
SEED = 0x5EED0000


def rot_matrix(deg, cx, cy, tx, ty):
    t = np.deg2rad(deg)
    c, s = np.cos(t), np.sin(t)
    return np.array([c, -s, cx - c * cx + s * cy + tx, s, c, cy - s * cx - c * cy + ty], np.float32)


def warp_touched_fraction(M, rows, cols, srows, scols):
    """distinct in-bounds source pixels a bilinear dst->src affine warp touches, counted once, as a
    fraction of the output pixel count (host side, float64 coordinates; a counting aid, not a parity path)."""
    touched = np.zeros((srows, scols), bool)
    xs = np.arange(cols, dtype=np.float64)
    for y0 in range(0, rows, 256):
        ys = np.arange(y0, min(rows, y0 + 256), dtype=np.float64)[:, None]
        sx = M[0] * xs[None, :] + M[1] * ys + M[2]
        sy = M[3] * xs[None, :] + M[4] * ys + M[5]
        inside = (sx > -1) & (sx < scols) & (sy > -1) & (sy < srows)
        fx, fy = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy = fx + dx, fy + dy
                ok = inside & (xx >= 0) & (xx < scols) & (yy >= 0) & (yy < srows)
                touched[yy[ok], xx[ok]] = True
    return float(touched.sum()) / float(rows * cols)


def bench_kernel7():
    def sm(z):
        M = (1 << 64) - 1
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    return np.array([int((sm(0xF117E2D ^ i) >> 40) % 17) - 8 for i in range(49)], np.int8).reshape(7, 7)


SETTLE_MS = 100.0   # untimed run-up per op: the GPU needs tens of ms under load to reach its sustained clocks


def timeit(ctx, fn, steps, warmup):
    import time
    L = _ffi.lib()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < SETTLE_MS:
        for _ in range(4):
            fn()
        ctx.sync()
    for _ in range(warmup):
        fn()
    ctx.sync()
    ms = C.c_float()
    L.rcv_timer_start(ctx.handle)
    for _ in range(steps):
        fn()
    L.rcv_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=float, default=1.0, help="scale the per-GPU batch sizes (1.0 = BASELINE per-GPU batches)")
    ap.add_argument("--cpu", action="store_true", help="also time the oracle on the host (bounded sample)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "ops_bench.json"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    ctx = rcv.Context(0)
    rows = []

    def B(n, r, c, ch, depth=_ffi.RCV_8U):
        return device.DeviceBatch(ctx, max(1, int(round(n * a.scale))), r, c, ch, depth)

    def clock_under(fn):
        """shader clock (MHz) sampled by a one-wave probe on the side stream while ~40 launches of `fn` are queued"""
        f = C.c_float()
        for _ in range(40):
            fn()
        _ffi.bench_lib().rcv__clock_probe(ctx.handle, 3000, C.byref(f))
        ctx.sync()
        return float(f.value)

    def record(name, cfg, n, out_px, bpp, fn, note="", cpu=None, valu=None):
        """valu = VALU instructions per output pixel the op's arithmetic needs at least (packed FMAs count as one): the row then
        carries its VALU roofline -- n * px * valu / (1024 SIMDs x 16 lanes per cycle x measured clock) / measured time"""
        if a.only and a.only.replace("_", " ") not in name + " @ " + cfg:   # "--only Sobel_3x3_->_dx,dy_i16_@_4K": op and config
            return
        ms = timeit(ctx, fn, a.steps, a.warmup)
        mpix = n * out_px / 1e6 / (ms * 1e-3)
        gbs = n * out_px * bpp / (ms * 1e-3) / 1e9
        row = {"op": name, "config": cfg, "frames": n, "ms_per_launch": round(ms, 4), "us_per_frame": round(ms * 1e3 / n, 3),
               "mpix_s": round(mpix, 1), "alg_bytes_per_px": bpp, "alg_gb_s": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM, 4), "note": note}
        if valu is not None:
            mhz = clock_under(fn)
            t_valu = n * out_px * valu / (1024 * 16 * mhz * 1e6)
            row.update({"valu_instr_per_px": valu, "shader_mhz": round(mhz, 1), "valu_bound_ms": round(t_valu * 1e3, 4), "frac_valu_bound": round(t_valu / (ms * 1e-3), 4)})
        if a.cpu and cpu is not None:
            row["cpu"] = cpu()
        rows.append(row)
        vb = f"  VALU bound {row['frac_valu_bound'] * 100:5.1f} % @ {row['shader_mhz']:.0f} MHz" if valu is not None else ""
        print(f"{name:34s} {cfg:34s} n={n:3d} {ms:9.4f} ms  {mpix:11.1f} Mpix/s  {gbs:8.1f} GB/s  {gbs / HBM * 100:5.1f} %{vb}  {row.get('cpu', '')}", flush=True)

    def cpu_time(fn, px, budget=4.0):
        from oracle import pyoracle as orc
        cores = orc.set_threads(orc.usable_cores())
        fn(orc)
        t0, k = time.perf_counter(), 0
        while True:
            fn(orc)
            k += 1
            dt = time.perf_counter() - t0
            if dt > budget or k >= 4096:
                break
        return {"mpix_s": round(k * px / 1e6 / dt, 2), "cores": cores, "frames": k, "kind": "port"}

    # ---- config 1: 640x480 YUYV -> BGR + rectangle (the reference's own path) --------------------------------
    n = 64
    y = B(n, 480, 640, 2)
    o = B(n, 480, 640, 3)
    device.synth(y, _ffi.RCV_SYNTH_YUYV, SEED + 1, 0)
    record("cvtColor YUYV2BGR", "640x480", y.n, 640 * 480, 5, lambda: device.cvt_color(y, o, _ffi.RCV_YUYV2BGR),
           cpu=lambda: cpu_time(lambda orc: orc.yuyv_to_bgr(np.zeros(640 * 480 * 2, np.uint8), np.zeros(640 * 480 * 3, np.uint8), 640, 480), 640 * 480))
    record("rectangle t=2", "640x480 Rect(200,150,240,240)", o.n, 640 * 480, 0, lambda: device.rectangle(o, Rect(200, 150, 240, 240), Scalar(0, 255, 0), 2),
           note="latency-bound: 1 920 pixel writes per frame")
    yy, xx = np.mgrid[0:34, 0:24].astype(np.float32)
    blob = np.clip(1.4 - np.hypot((xx - 11.5) / 9, (yy - 16.5) / 14), 0, 1).astype(np.float32)
    line = [(20 + 20 * i, 400 + (i % 3), blob) for i in range(28)]      # one text line: 28 overlapping soft-edged boxes
    record("text blend, 28 glyphs (put_text)", "640x480", o.n, 640 * 480, 0, lambda: device.blend_glyphs(o, line, Scalar(0, 255, 0)),
           note="f4: latency-bound (91 KB of coverage uploaded per call, ~22 800 blended pixels per frame)",
           cpu=lambda: cpu_time(lambda orc: orc.blend_glyphs(np.zeros(640 * 480 * 3, np.uint8), 480, 640, 640 * 3, line, 0, 255, 0), 640 * 480))
    y.free(); o.free()
    # the same conversions at 4K (bandwidth regime)
    n = 64
    y = B(n, 2160, 3840, 2)
    o = B(n, 2160, 3840, 3)
    device.synth(y, _ffi.RCV_SYNTH_YUYV, SEED + 1, 0)
    record("cvtColor YUYV2BGR", "4K", y.n, 3840 * 2160, 5, lambda: device.cvt_color(y, o, _ffi.RCV_YUYV2BGR))
    k7c = bench_kernel7()
    record("fused YUYV->BGR->filter2D 7x7", "4K batch=64", y.n, 3840 * 2160, 5, lambda: device.filter2d_yuyv(y, o, k7c, shift=6),
           note="f1: one launch, 2 B read + 3 B written per px; the two-step path moves 11 B/px")
    y.free()
    q = B(n, 2160, 3840, 4)
    device.synth(q, 0, SEED + 1, 0)
    record("cvtColor BGRA2BGR", "4K", q.n, 3840 * 2160, 7, lambda: device.cvt_color(q, o, _ffi.RCV_BGRA2BGR))
    q.free()
    g = B(n, 2160, 3840, 1)
    record("cvtColor BGR2GRAY", "4K", o.n, 3840 * 2160, 4, lambda: device.cvt_color(o, g, _ffi.RCV_BGR2GRAY))
    o.free()

    g2 = B(n, 2160, 3840, 1)
    record("filter2D 7x7 i8 on a GRAY image", "4K gray", g.n, 3840 * 2160, 2, lambda: device.filter2d(g, g2, k7c, shift=6),
           note="1 B read + 1 B written per px, 49 MAC per px on v_dot4c_i32_i8")
    record("GaussianBlur 5x5 (sigma=0) on a GRAY image", "4K gray", g.n, 3840 * 2160, 2, lambda: device.gaussian_blur(g, g2, 5, 0.0))
    g2.free()
    # ---- config 3 (second half): Sobel on 4K gray, batch 64 -------------------------------------------------
    dx, dy = B(n, 2160, 3840, 1, _ffi.RCV_16S), B(n, 2160, 3840, 1, _ffi.RCV_16S)
    record("Sobel 3x3 -> dx,dy i16", "4K gray", g.n, 3840 * 2160, 5, lambda: device.sobel(g, dx, dy),
           cpu=lambda: cpu_time(lambda orc: orc.sobel(np.zeros((2160, 3840), np.uint8)), 3840 * 2160))
    o2 = B(n, 2160, 3840, 3)
    device.synth(o2, 1, SEED + 3, 0)
    record("Sobel of a BGR source, fused gray (next row f1)", "4K", o2.n, 3840 * 2160, 7, lambda: device.sobel(o2, dx, dy),
           note="3 B read + 4 B written per px; the two-launch chain BGR2GRAY + Sobel moves 9")
    record("filter2D 7x7 i8 -> gray -> Sobel FUSED (config 3 in one launch, next row f1)", "4K batch=64", o2.n, 3840 * 2160, 7,
           lambda: device.filter2d_sobel(o2, dx, dy, k7c, shift=6),
           note="3 B read + 4 B written per px; filter2D + Sobel-of-BGR as two launches moves 13")
    o2.free()
    dx.free(); dy.free(); g.free()

    # ---- config 2: 1080p BGR 5x5 Gaussian, batch 1 (latency) and batch 64 (bandwidth) ------------------------
    for nb in (1, 64):
        s, d = B(nb, 1080, 1920, 3), B(nb, 1080, 1920, 3)
        device.synth(s, 0, SEED + 2, 0)
        record("GaussianBlur 5x5 (sigma=0)", f"1080p batch={s.n}", s.n, 1920 * 1080, 6, lambda: device.gaussian_blur(s, d, 5, 0.0),
               note="batch 1 is L3-resident: read as latency" if nb == 1 else "",
               cpu=(lambda: cpu_time(lambda orc: orc.gaussian_blur(np.zeros((1080, 1920, 3), np.uint8), 5, 0.0), 1920 * 1080)) if nb == 1 else None)
        s.free(); d.free()

    # ---- config 3: 4K 7x7 filter2D, batch 64 (north star; bench.py is the authoritative line) ---------------
    s, d = B(64, 2160, 3840, 3), B(64, 2160, 3840, 3)
    device.synth(s, 0, SEED + 3, 0)
    k7 = bench_kernel7()
    record("filter2D 7x7 i8 (>>6)", "4K batch=64", s.n, 3840 * 2160, 6, lambda: device.filter2d(s, d, k7, shift=6))
    kf = (k7.astype(np.float32) / 64.0)
    record("filter2D 7x7 f32", "4K batch=64", s.n, 3840 * 2160, 6, lambda: device.filter2d(s, d, kf, delta=0.0),
           note="VALU-bound by construction (49 dependent fmaf per sample), reported for completeness", valu=147 / 2)   # 3 x 49 FMA per px, two per v_pk_fma_f32
    record("GaussianBlur 7x7 (sigma=0)", "4K batch=64", s.n, 3840 * 2160, 6, lambda: device.gaussian_blur(s, d, 7, 0.0))
    record("GaussianBlur 7x7 (sigma=1.5)", "4K batch=64", s.n, 3840 * 2160, 6, lambda: device.gaussian_blur(s, d, 7, 1.5),
           note="f32 separable path, 14 fmaf per sample", valu=42 / 2)
    # ---- config 5: Harris pipeline on 4K BGR, 64 frames per GPU ----------------------------------------------
    m = B(64, 2160, 3840, 1)
    device.synth(s, 1, SEED + 5, 0)
    record("Harris pipeline (BGR->mask)", "4K batch=64/GPU", s.n, 3840 * 2160, 4, lambda: device.harris_pipeline(s, m, None, 2, 0.04, 1e-4),
           cpu=lambda: cpu_time(lambda orc: orc.harris_pipeline(np.zeros((2160, 3840, 3), np.uint8), 2, 0.04, 1e-4), 3840 * 2160),
           valu=23)   # VALU instructions per pixel of the kernel as built (rocprofv3 SQ_INSTS_VALU 1.9e8 per launch, round 6; round 4: 28, round 2: 32)
    yq = B(64, 2160, 3840, 2)
    device.synth(yq, 2, SEED + 5, 0)
    record("Harris pipeline from a YUYV source (config 5 [or YUYV])", "4K batch=64/GPU", yq.n, 3840 * 2160, 3,
           lambda: device.harris_pipeline(yq, m, None, 2, 0.04, 1e-4), note="2 B read + 1 B written per px; YUYV->BGR + pipeline as two launches moves 9")
    yq.free()
    gq = B(64, 2160, 3840, 1)
    device.synth(gq, 1, SEED + 5, 0)
    record("Harris pipeline from a GRAY source", "4K batch=64/GPU", gq.n, 3840 * 2160, 2, lambda: device.harris_pipeline(gq, m, None, 2, 0.04, 1e-4))
    rq = B(64, 2160, 3840, 1, _ffi.RCV_32F)
    record("cornerHarris (gray -> f32 response)", "4K batch=64/GPU", gq.n, 3840 * 2160, 5, lambda: device.corner_harris(gq, rq, 2, 0.04),
           note="the fused kernel without its NMS stage")
    record("cornerHarris blockSize 3 (one launch, general-block window kernel)", "4K batch=64/GPU", gq.n, 3840 * 2160, 5,
           lambda: device.corner_harris(gq, rq, 3, 0.04))
    record("Harris pipeline blockSize 3 (BGR->mask, one launch)", "4K batch=64/GPU", s.n, 3840 * 2160, 4,
           lambda: device.harris_pipeline(s, m, None, 3, 0.04, 1e-4))
    record("NMS 3x3 (f32 response -> mask)", "4K batch=64/GPU", rq.n, 3840 * 2160, 5, lambda: device.nms3x3(rq, m, 1e-4))
    gq.free(); rq.free()
    s.free(); d.free(); m.free()

    # ---- the shapes real callers produce (Mat::new: step = cols * channels): odd widths, byte-aligned rows, next to the aligned
    # ---- shape of the same size -- the register-window kernels' RAG instantiations against their aligned ones ----------------
    for (rr, cc, tag) in ((1080, 1920, "1920x1080 (aligned)"), (1079, 1919, "1919x1079 packed (odd width, byte-aligned rows)")):
        nb = 64
        gg, bb = B(nb, rr, cc, 1), B(nb, rr, cc, 3)
        ddx, ddy, mm = B(nb, rr, cc, 1, _ffi.RCV_16S), B(nb, rr, cc, 1, _ffi.RCV_16S), B(nb, rr, cc, 1)
        device.synth(gg, 1, SEED + 7, 0)
        device.synth(bb, 1, SEED + 7, 0)
        record("Sobel 3x3 -> dx,dy i16", tag, gg.n, rr * cc, 5, lambda: device.sobel(gg, ddx, ddy))
        record("Harris pipeline (BGR->mask)", tag, bb.n, rr * cc, 4, lambda: device.harris_pipeline(bb, mm, None, 2, 0.04, 1e-4))
        for x in (gg, bb, ddx, ddy, mm):
            x.free()

    # ---- config 4: 8K warpAffine + resize -> 1080p, 32 frames per GPU ----------------------------------------
    s, d = B(32, 4320, 7680, 3), B(32, 4320, 7680, 3)
    device.synth(s, 0, SEED + 4, 0)
    M = rot_matrix(7.0, 7680 / 2, 4320 / 2, 13.25, -8.5)
    frac = warp_touched_fraction(M.astype(np.float64), 4320, 7680, 4320, 7680) if not a.only or "warpAffine bil" in a.only.replace("_", " ") or "GRAY" in a.only or "f32" in a.only else 1.0
    record("warpAffine bilinear (rot 7deg)", "8K batch=32/GPU", s.n, 7680 * 4320, 3 + 3 * frac, lambda: device.warp_affine(s, d, M), valu=32,
           note=f"3 B written per output px + 3 B per DISTINCT in-bounds source px touched ({frac:.4f} per output px, counted on the host; upper bound 6)",
           cpu=lambda: cpu_time(lambda orc: orc.warp_affine(np.zeros((4320, 7680, 3), np.uint8), M, 4320, 7680), 7680 * 4320, 6.0))
    d.free()
    sg, dg = B(32, 4320, 7680, 1), B(32, 4320, 7680, 1)
    device.synth(sg, 0, SEED + 4, 0)
    record("warpAffine bilinear (rot 7deg) on a GRAY image", "8K gray batch=32/GPU", sg.n, 7680 * 4320, 1 + frac, lambda: device.warp_affine(sg, dg, M))
    sg.free(); dg.free()
    d = B(32, 1080, 1920, 3)
    record("warpAffine + resize 8K -> 1080p FUSED (next row f1)", "8K batch=32/GPU", s.n, 1920 * 1080, 30,
           lambda: device.warp_affine_resize(s, d, M, 4320, 7680), valu=202,   # (k_warp_resize_stage, SQ_INSTS_VALU 2.09e8 per launch: ~160 in the frame loop + plans + border tiles; the gather kernel: 278)
           note="30 B per OUTPUT px: the centre 2x2 warped pixels of each 4x4 block tap a 3x3 source block (27 B) + 3 B written; "
                "the unfused pair moves 6 B/px of 8K intermediate on top")
    d2 = B(32, 2880, 5120, 3)
    record("resize 8K -> 5K bilinear (general 1.5x)", "8K batch=32/GPU", s.n, 5120 * 2880, 3 + 3 * 2.25, lambda: device.resize(s, d2),
           note="3 B written + 2.25 source px (6.75 B) read per output px")
    d2.free()
    record("resize 8K -> 1080p bilinear", "8K batch=32/GPU", s.n, 1920 * 1080, 15, lambda: device.resize(s, d),
           note="15 B per OUTPUT px: exact 4x touches the centre 2x2 of each 4x4 block",
           cpu=lambda: cpu_time(lambda orc: orc.resize(np.zeros((4320, 7680, 3), np.uint8), 1080, 1920), 1920 * 1080))
    s.free(); d.free()

    # ---- RCV_32F geometry (the 1-ULP rows): a one-channel response map, 8 x 8K ----
    sf, df = B(8, 4320, 7680, 1, _ffi.RCV_32F), B(8, 4320, 7680, 1, _ffi.RCV_32F)
    sf.memset(0x3C)   # (every float 0.0115: timing only)
    record("warpAffine bilinear f32 (rot 7deg)", "8K f32 1ch batch=8", sf.n, 7680 * 4320, 4 + 4 * frac, lambda: device.warp_affine(sf, df, M),
           note="4 B written + 4 B per distinct source sample touched")
    df.free()
    df = B(8, 1080, 1920, 1, _ffi.RCV_32F)
    record("resize f32 8K -> 1080p bilinear", "8K f32 1ch batch=8", sf.n, 1920 * 1080, 20, lambda: device.resize(sf, df),
           note="4 source samples (16 B) read + 4 B written per output px")
    df.free()
    df = B(8, 2880, 5120, 1, _ffi.RCV_32F)
    record("resize f32 8K -> 5K bilinear (general 1.5x)", "8K f32 1ch batch=8", sf.n, 5120 * 2880, 4 + 4 * 2.25, lambda: device.resize(sf, df))
    sf.free(); df.free()

    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"device": "MI355X (gfx950)", "steps": a.steps, "hbm_peak_gb_s": HBM, "rows": rows}, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
