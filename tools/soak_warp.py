#!/usr/bin/env python3
"""tools/soak_warp.py [N] [CH | f32] -- N seeded random (shape, map, batch, frames-per-workgroup) cases of the BGR (CH = 1: one-channel) warpAffine against the oracle,
run from the repo root on a GPU box; prints how many launches took the LDS-staged kernel and the number of mismatches (round 2: 400 cases,
676 launches on the LDS kernel, 0 mismatches)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import rustcv_amd as rcv
from rustcv_amd import _ffi, device
from oracle import pyoracle as oracle

def rot(deg, cx, cy, tx, ty):
    t = np.deg2rad(deg); c, s = np.cos(t), np.sin(t)
    return np.array([c, -s, cx - c * cx + s * cy + tx, s, c, cy - s * cx - c * cy + ty], np.float32)

ctx = rcv.Context(0)
L = _ffi.lib()
bad = 0
nlds = 0
nquad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
F32 = len(sys.argv) > 2 and sys.argv[2] == "f32"   # (round 4) one-channel RCV_32F frames: k_warp_f32_lds, inf / NaN next to the source border
CH = 1 if F32 else (int(sys.argv[2]) if len(sys.argv) > 2 else 3)
for case in range(N):
    rng = np.random.default_rng(0xABCD00 + case)
    sr, sc = int(rng.integers(40, 700)), int(rng.integers(40, 900))
    dr, dc = int(rng.integers(8, 600)), (4 * int(rng.integers(2, 220)) if case % 3 else int(rng.integers(5, 880)))   # every third case: any width
    n = int(rng.integers(1, 10)) if case % 4 else int(rng.integers(4, 22))
    kind = case % 8
    if kind == 0: M = rot(float(rng.uniform(-180, 180)), sc / 2, sr / 2, float(rng.uniform(-50, 50)), float(rng.uniform(-50, 50)))
    elif kind == 1: M = rot(float(rng.choice([7.0, 45.0, 90.0, -90.0, 180.0, 0.1, 0.0])), sc / 2, sr / 2, 13.25, -8.5)
    elif kind == 2: M = np.array([1, 0, float(rng.uniform(-5, 5)), 0, 1, float(rng.uniform(-5, 5))], np.float32)
    elif kind == 3:
        sx, sy = float(rng.uniform(0.3, 2.5)), float(rng.uniform(0.3, 2.5))
        M = np.array([sx, 0, float(rng.uniform(0, 9)), 0, sy, float(rng.uniform(0, 9))], np.float32)
    elif kind == 4: M = np.array([1, float(rng.uniform(-1, 1)), 3.5, float(rng.uniform(-1, 1)), 1, 2.25], np.float32)
    elif kind == 5: M = (np.array([1, 0, 0, 0, 1, 0]) + rng.uniform(-0.5, 0.5, 6) * np.array([1, 1, 80, 1, 1, 80])).astype(np.float32)
    elif kind == 6: M = rot(float(rng.uniform(-180, 180)), sc / 2, sr / 2, 0, 0) * np.float32(rng.uniform(0.5, 1.6))
    else: M = np.array([float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-100, 800)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-100, 600))], np.float32)
    M = np.asarray(M, np.float32)
    if F32:
        if case % 2: sc = (sc + 3) & ~3   # (every second case: a width the border tiles can be staged for)
        frames = (rng.standard_normal((n, sr, sc, 1)) * rng.choice([1e-3, 1.0, 3e4])).astype(np.float32)
        if case % 3 == 0:
            frames[:, 0, : sc // 3] = np.inf; frames[:, -1, sc // 2:] = -np.inf; frames[:, sr // 3: sr // 2, 0] = np.nan; frames[:, :, -1] = np.float32(1e-41)
        src = device.DeviceBatch(ctx, n, sr, sc, 1, _ffi.RCV_32F)
        dst = device.DeviceBatch(ctx, n, dr, dc, 1, _ffi.RCV_32F)
    else:
        frames = rng.integers(0, 256, size=(n, sr, sc, CH), dtype=np.uint8)
        src = device.DeviceBatch(ctx, n, sr, sc, CH, step=sc * CH + (int(rng.integers(0, 4)) if CH == 1 and case % 2 else 0))
        dst = device.DeviceBatch(ctx, n, dr, dc, CH)
    src.upload(frames)
    for fpg in (0, int(rng.integers(1, 9))):
        os.environ.pop("RCV_WARP_FPG", None)
        if fpg: os.environ["RCV_WARP_FPG"] = str(fpg)
        L.rcv__debug_reload_knobs()
        dst.memset(0x5A)
        L.rcv__debug_kernels_reset()
        device.warp_affine(src, dst, M)
        ctx.sync()
        k = L.rcv__debug_kernels().decode()
        nlds += "lds" in k
        nquad += "k_warp_gray_lds4" in k
        got = dst.download()
        for i in range(n):
            if F32:
                want = oracle.warp_affine_f32(frames[i, :, :, 0], M, dr, dc).reshape(dr, dc, 1)
                g, w = got[i].reshape(dr, dc, 1), want
                same = np.array_equal(g.view(np.uint32)[~np.isnan(w)], w.view(np.uint32)[~np.isnan(w)]) and np.array_equal(np.isnan(g), np.isnan(w))
                if same: continue
                bad += 1
                print("MISMATCH", case, fpg, i, k, M.tolist(), (sr, sc, dr, dc, n), flush=True)
                break
            want = oracle.warp_affine(frames[i] if CH == 3 else frames[i, :, :, 0], M, dr, dc).reshape(dr, dc, CH)
            if not np.array_equal(got[i].reshape(dr, dc, CH), want):
                bad += 1
                print("MISMATCH", case, fpg, i, k, M.tolist(), (sr, sc, dr, dc, n), int((got[i].reshape(dr, dc, CH) != want).sum()), flush=True)
                break
    src.free(); dst.free()
print(f"soak: {N} cases, {nlds} launches on the LDS kernels ({nquad} on the four-frames-per-pass one), {bad} mismatches")
