#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz|json.  Run from the repo root:  python tests/golden/make_golden.py

Two kinds of fixture (data only -- inputs and expected outputs):
  kat_reference.json : the known-answer vectors the REFERENCE's own tests hold for this path
        (rustcv-camera/src/decode.rs:234-273: two inequality checks + one exact vector) and ten
        (Y,U,V)->(B,G,R) triples derived by hand from the formula at rustcv/src/videoio/mod.rs:356-363
        (SURVEY.md 8(c)).  These are written out literally below; nothing is computed by the oracle.
  ops_small.npz      : small seeded inputs and the ORACLE's outputs for every op (regression pin of the
        oracle itself and an extra GPU check).  The reference cannot generate these: it is Rust (no
        rustc here) and most of the ops do not exist in it.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as orc  # noqa: E402

KAT = {
    "source": "rustcv-camera/src/decode.rs:234-273 (tests) ; rustcv/src/videoio/mod.rs:356-363 (formula)",
    "reference_tests": [
        {"name": "yuyv_to_bgr_basic", "yuyv": [235, 128, 235, 128], "w": 2, "h": 1, "check": "all > 240"},
        {"name": "yuyv_to_bgr_black", "yuyv": [16, 128, 16, 128], "w": 2, "h": 1, "check": "all < 10"},
        {"name": "rgb_to_bgr_swap", "rgb": [255, 0, 0, 0, 255, 0], "bgr": [0, 0, 255, 0, 255, 0]},
    ],
    "yuv_to_bgr_hand_derived": [
        [[235, 128, 128], [255, 255, 255]], [[16, 128, 128], [0, 0, 0]], [[0, 0, 0], [0, 135, 0]],
        [[255, 255, 255], [255, 125, 255]], [[81, 90, 240], [0, 0, 255]], [[145, 54, 34], [1, 255, 0]],
        [[41, 240, 110], [255, 0, 0]], [[128, 128, 128], [130, 130, 130]], [[255, 0, 0], [20, 255, 74]],
        [[0, 255, 255], [237, 0, 184]],
    ],
}


def main():
    # kat_reference.json also holds hand-derived vectors that were added to the file directly (the put_text blend KATs): keep
    # every key this script does not own
    kat_path = os.path.join(HERE, "kat_reference.json")
    kat = json.load(open(kat_path)) if os.path.exists(kat_path) else {}
    kat.update(KAT)
    json.dump(kat, open(kat_path, "w"), indent=1)
    rng = np.random.default_rng(20260928)
    g = {}
    bgr = rng.integers(0, 256, size=(37, 48, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, size=(29, 41), dtype=np.uint8)
    g["bgr"], g["gray"] = bgr, gray
    yuyv = rng.integers(0, 256, size=24 * 10 * 2, dtype=np.uint8)
    out = np.zeros(24 * 10 * 3, np.uint8)
    orc.yuyv_to_bgr(yuyv, out, 24, 10)
    g["yuyv"], g["yuyv_bgr"] = yuyv, out
    bgra = rng.integers(0, 256, size=21 * 4, dtype=np.uint8)
    out = np.zeros(21 * 3, np.uint8)
    orc.bgra_to_bgr(bgra, out, 21, 1)
    g["bgra"], g["bgra_bgr"] = bgra, out
    rect = bgr.copy().reshape(-1)
    orc.rectangle(rect, 37, 48, 144, 5, 4, 30, 20, 0, 255, 0, 2)
    g["rect_5_4_30_20_t2"] = rect
    g["bgr2gray"] = orc.bgr2gray(bgr)
    for ks in (3, 5, 7):
        g[f"gauss{ks}"] = orc.gaussian_blur(bgr, ks, 0.0)
    g["gauss5_s1p2"] = orc.gaussian_blur(bgr, 5, 1.2)
    k7 = orc.bench_kernel7()
    g["k7"] = k7
    g["filter7_s6"] = orc.filter2d_i8(bgr, k7, 6)
    kf = (rng.standard_normal((3, 3)) / 3).astype(np.float32)
    g["kf3"] = kf
    g["filter3_f32"] = orc.filter2d_f32(bgr, kf, 0.25)
    g["sobel_dx"], g["sobel_dy"] = orc.sobel(gray)
    g["resize_9x12"] = orc.resize(bgr, 9, 12)
    g["resize_50x70"] = orc.resize(bgr, 50, 70)
    M = np.array([0.99254615, -0.12186934, 5.5, 0.12186934, 0.99254615, -3.25], np.float32)
    g["warp_M"] = M
    g["warp"] = orc.warp_affine(bgr, M, 37, 48)
    g["harris_b2"] = orc.corner_harris(gray, 2, 0.04)
    g["nms"] = orc.nms3x3(g["harris_b2"], 1e-4)
    g["synth_scene"] = orc.synth_frame(24, 40, 3, 1, 0x5EED0003, 2)
    g["synth_noise"] = orc.synth_frame(8, 8, 3, 0, 0x5EED0003, 0)
    # RCV_32F geometry (round 3): the Harris response map through the f32 warp / resize (no further random draws above this line
    # were added: the earlier arrays are unchanged)
    g["warp_f32"] = orc.warp_affine_f32(g["harris_b2"], M, 29, 41)
    g["resize_f32_17x23"] = orc.resize_f32(g["harris_b2"], 17, 23)
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **g)
    print("wrote", len(g), "arrays")


if __name__ == "__main__":
    main()
