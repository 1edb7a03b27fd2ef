"""bench.py's aggregation (CPU): the JSON line is built from per-GPU measurements by plain arithmetic -- whole-job throughput from
the slowest GPU's wall time, the roofline from the slowest GPU's sustained launch time, the copy ceiling from the slowest GPU's
best copy, and a mismatching verified frame is reported (bench.py then exits non-zero)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _rank(elapsed, launch_ms, ceiling, frames, bad=()):
    return {"elapsed": elapsed, "ev_ms_steps": launch_ms * 1.01, "kernel": "(k_filter_rows_mfma<KS, PP, 0>)", "launch_ms": launch_ms,
            "launches_sustained": 400, "launch_ms_first20": launch_ms * 1.1, "verified_frames": list(frames), "mismatched_frames": list(bad),
            "copy_ceiling_gbs": ceiling, "copy_ceiling_kernel": "sweep g=1024"}


def test_report_single_gpu():
    a = argparse.Namespace(batch=64, steps=50, warmup=5)
    out, bad = bench.report(a, 1, [_rank(0.0275, 0.55, 6000.0, [0, 31, 63])])
    px = 64 * 2160 * 3840
    assert not bad and out["n_gpus"] == 1 and out["unit"] == "Mpix/s" and out["higher_is_better"] and out["vs_baseline"] is None
    assert abs(out["value"] - px * 50 / 0.0275 / 1e6) < 1 and abs(out["ms_per_step"] - 0.55) < 1e-9
    r = out["roofline"]
    assert r["alg_bytes_per_launch"] == px * 6 == 3185049600 and r["peak"] == 8000.0 and r["bound"] == "hbm"
    assert abs(r["achieved"] - 3185049600 / 0.55e-3 / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert abs(r["frac_of_copy_ceiling"] - r["achieved"] / 6000.0) < 1e-4
    assert out["verified_frames"] == [0, 31, 63] and out["verified"].startswith("bit-exact")
    assert out["dtype"] == "u8" and out["scaling"] == "weak" and "workload" in out["config"] and "model" not in out["config"]


def test_report_takes_the_slowest_gpu_and_flags_mismatches():
    a = argparse.Namespace(batch=64, steps=20, warmup=3)
    res = [_rank(0.0110, 0.55, 6000.0, [0, 31, 63]), _rank(0.0124, 0.61, 5400.0, [64, 95, 127], bad=[95])]
    out, bad = bench.report(a, 2, res)
    assert bad == [95] and "MISMATCH" in out["verified"]
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128
    assert abs(out["value"] - 128 * 2160 * 3840 * 20 / 0.0124 / 1e6) < 1           # whole job over the slowest GPU's wall time
    assert out["roofline"]["launch_ms"] == 0.61 and out["roofline"]["copy_ceiling_gbs"] == 5400.0
    assert out["roofline"]["launch_ms_per_gpu"] == [0.55, 0.61]
    assert out["verified_frames"] == [0, 31, 63, 64, 95, 127]


def test_bench_kernel_matches_the_oracle_generator(oracle):
    import numpy as np
    assert np.array_equal(bench.bench_kernel7(), oracle.bench_kernel7())
