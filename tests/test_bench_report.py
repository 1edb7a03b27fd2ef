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
    assert out["roofline"]["in_flight"] == 1 and "single_stream" not in out["roofline"] and out["config"]["frames_per_gpu"] == 64 and out["config"]["global_batch"] == 64
    assert "two launches" in out["config"]["call_form"] and "RCV_FR_SPLIT=0" in out["config"]["call_form"]
    # round 6: one context is the headline; the one-launch form and the two-batches-in-flight form ride beside it, the flat single_stream_* repeat the headline
    r6 = dict(_rank(0.0275, 0.55, 6000.0, [0, 31, 63]), single_one_launch_ms=0.56, in_flight2_launch_ms=0.545)
    o6, _ = bench.report(a, 1, [r6])
    rf6 = o6["roofline"]
    assert rf6["single_stream_frac"] == rf6["frac"] and rf6["single_stream_launch_ms"] == rf6["launch_ms"]
    assert rf6["single_stream_one_launch_ms"] == 0.56 and abs(rf6["single_stream_one_launch_frac"] - 3185049600 / 0.56e-3 / 1e9 / 8000.0) < 1e-4
    assert rf6["in_flight2_launch_ms"] == 0.545 and abs(rf6["in_flight2_frac"] - 3185049600 / 0.545e-3 / 1e9 / 8000.0) < 1e-4
    r = out["roofline"]
    assert r["alg_bytes_per_launch"] == px * 6 == 3185049600 and r["peak"] == 8000.0 and r["bound"] == "hbm"
    assert abs(r["achieved"] - 3185049600 / 0.55e-3 / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert abs(r["frac_of_copy_ceiling"] - r["achieved"] / 6000.0) < 1e-4
    assert out["verified_frames"] == [0, 31, 63] and out["verified"].startswith("bit-exact")
    assert out["dtype"] == "u8" and out["scaling"] == "weak" and "workload" in out["config"] and "model" not in out["config"]


def test_report_two_batches_in_flight():
    """round 4: F = 2 contexts per GPU, one 64-frame launch each per step.  value counts the pixels of BOTH launches over the wall
    time; roofline.launch_ms is the sustained window divided by the LAUNCHES in it (bytes moved / wall time), the one-stream figure
    sits beside it, and every context's frames are in verified_frames"""
    a = argparse.Namespace(batch=64, steps=20, warmup=5)
    r = dict(_rank(0.0220, 0.548, 6000.0, [0, 31, 63, 64, 95, 127]), in_flight=2, single_launch_ms=0.610, launches_sustained=800, ev_ms_steps=1.10, device=0)
    out, bad = bench.report(a, 1, [r])
    px = 64 * 2160 * 3840
    assert not bad and out["n_gpus"] == 1 and out["config"]["global_batch"] == 128 and out["config"]["frames_per_launch"] == 64
    assert out["config"]["launches_per_step_per_gpu"] == 2 and "2 batches in flight" in out["config"]["parallelism"]
    assert abs(out["value"] - 2 * px * 20 / 0.0220 / 1e6) < 1 and abs(out["ms_per_step"] - 1.1) < 1e-9
    assert abs(out["value"] - 128 * 2160 * 3840 / out["ms_per_step"] / 1e3) < 1          # value == global_batch pixels / ms_per_step
    rf = out["roofline"]
    assert rf["in_flight"] == 2 and rf["alg_bytes_per_launch"] == 3185049600 and rf["launches"] == 800
    assert abs(rf["achieved"] - 3185049600 / 0.548e-3 / 1e9) < 0.1 and abs(rf["frac"] - rf["achieved"] / 8000.0) < 1e-4
    assert rf["single_stream"]["launch_ms"] == 0.61 and abs(rf["single_stream"]["frac"] - 3185049600 / 0.61e-3 / 1e9 / 8000.0) < 1e-4
    # round 5: the one-stream figure also as SCALARS of the roofline object and as a top-level value (a parser that drops nested objects keeps them)
    assert rf["single_stream_launch_ms"] == 0.61 and rf["single_stream_frac"] == rf["single_stream"]["frac"] and rf["single_stream_achieved"] == rf["single_stream"]["achieved"]
    assert all(not isinstance(rf[k], (dict, list)) for k in ("single_stream_launch_ms", "single_stream_frac", "single_stream_achieved"))
    assert abs(out["value_single_stream"] - px / 0.61e-3 / 1e6) < 1 and out["value_single_stream"] < out["value"]
    assert abs(rf["launch_ms_timed_steps"] - 0.55) < 1e-9 and "overlap" in rf["launch_ms_note"]
    assert out["verified_frames"] == [0, 31, 63, 64, 95, 127]
    # round 6: a call of 16+ frames runs as two halves on the context's two streams by default; the line carries the one-launch form beside it
    assert "single_stream_one_launch_frac" not in rf
    r2 = dict(r, single_one_launch_ms=0.620)
    out2, _ = bench.report(a, 1, [r2])
    rf2 = out2["roofline"]
    assert rf2["single_stream_one_launch_ms"] == 0.62 and abs(rf2["single_stream_one_launch_frac"] - 3185049600 / 0.62e-3 / 1e9 / 8000.0) < 1e-4
    assert "two 32-frame launches" in rf2["single_stream_note"] and "RCV_FR_SPLIT=0" in rf2["single_stream_note"]


def test_report_ranks_sharing_a_device_is_labelled_a_test():
    """--devices 0,0 (the N > 1 code on a one-GPU box): n_gpus counts DISTINCT devices, the ranks appear as contexts"""
    a = argparse.Namespace(batch=64, steps=20, warmup=3)
    res = [dict(_rank(0.024, 1.2, 6000.0, [0, 31, 63]), device=0), dict(_rank(0.025, 1.25, 6000.0, [64, 95, 127]), device=0)]
    out, bad = bench.report(a, 2, res)
    assert not bad and out["n_gpus"] == 1 and out["config"]["contexts"] == 2 and "not a scaling" in out["config"]["note"]
    assert out["roofline"]["launch_ms_per_gpu"] == [1.2, 1.25]


def test_report_carries_the_other_configs():
    a = argparse.Namespace(batch=64, steps=20, warmup=3)
    rec = {"metric": "m", "value": 1.0, "roofline": {"frac": 0.4}, "verified": "bit-exact vs the CPU oracle", "mismatched_frames": []}
    r = dict(_rank(0.011, 0.55, 6000.0, [0, 31, 63]), other_configs={"4": dict(rec), "5": dict(rec, mismatched_frames=[7])})
    out, bad = bench.report(a, 1, [r])
    assert set(out["other_configs"]) == {"4", "5"} and bad == [7] and "mismatched_frames" not in out["other_configs"]["5"]


def test_other_configs_cover_the_sobel_half_of_config_3():
    """round 5: BASELINE configs[2] reads "7x7 filter2D + Sobel gradient" -- the default run carries the Sobel launch
    ("3s") and the one-launch form ("3f") beside configs 4 and 5, each with its algorithmic bytes (3 B read + 2 x i16
    written = 7 B per pixel) and an oracle step that is the composition of the oracle's own passes"""
    import numpy as np
    for key in ("3s", "3f"):
        c = bench.CONFIGS[key]
        assert c["batch"] == 64 and c["alg_bytes"] == 2160 * 3840 * 7 and c["px"] == 2160 * 3840 and key in bench.TRAFFIC_KEYS and key in bench.SEEDS
        assert bench.synth_args(key, 0) == (2160, 3840, 0, bench.SEEDS[3])
    assert "Sobel" in bench.CONFIGS["3s"]["metric"] and "one launch" in bench.CONFIGS["3f"]["metric"]
    a = bench.parse([])
    assert a.config == 3 and a.other_cpu_seconds > 0
    import pytest
    with pytest.raises(SystemExit):
        bench.parse(["--config", "3s"])        # records of the default run, not headline configs

    class Orc:   # stand-in that records the composition
        def bench_kernel7(self): return "k7"
        def filter2d_i8(self, f, k, s): return ("filt", f, k, s)
        def bgr2gray(self, f): return ("gray", f)
        def sobel(self, g): return ("sobel", g)
    assert bench.oracle_step(Orc(), "3s", "F") == ("sobel", ("gray", "F"))
    assert bench.oracle_step(Orc(), "3f", "F") == ("sobel", ("gray", ("filt", "F", "k7", 6)))


def test_default_run_names_every_baseline_config():
    """round 6: the driver's line carries a record for EVERY BASELINE config -- "1" (640x480 YUYV -> BGR + rectangle: the
    reference's own path, rustcv/src/videoio/mod.rs:201-258 + imgproc/drawing.rs:67-106; CPU restatement on one thread and all cores, GPU
    latency beside it) and "2" (1080p 5x5, us per launch with the empty-kernel and frame-copy floors of the same run) beside 3s / 3f / 4 / 5,
    and config 5 reports its worst-case launch (the time depends on corner density since the threshold-first NMS).  The GPU side of this is
    tests/test_gpu_bench_ranks.py::test_default_run_carries_the_other_configs_small; here: the wiring."""
    import inspect
    src = inspect.getsource(bench.run_rank)
    assert '"1": config1_record(a, ctx0)' in src and '"2": config2_record(a, ctx0)' in src and '("3s", "3f", 4, 5)' in src
    c1, c2, oc = inspect.getsource(bench.config1_record), inspect.getsource(bench.config2_record), inspect.getsource(bench.other_config)
    for must in ("value_1thread", "gpu_device_resident", "gpu_host_mat", "Rect(200, 150, 240, 240)", "alg_bytes_per_px"):
        assert must in c1, must
    for must in ("empty_kernel_us", "copy_of_the_frame_us", '"higher_is_better": False', "gaussian_blur(src, dst, 5, 0.0)"):
        assert must in c2, must
    for must in ("worst_case_launch_ms", "worst_case_frac", 'float("-inf")'):
        assert must in oc, must


def test_parse_defaults():
    a = bench.parse([])
    assert a.in_flight == 1 and a.batch == 64 and a.gpus == 1 and a.device_list is None   # (round 6: BASELINE's literal batch on one context)
    a = bench.parse(["--config", "5"])
    assert a.in_flight == 1 and a.batch == 64
    a = bench.parse(["--config", "4", "--in-flight", "2", "--gpus", "2", "--devices", "0,0"])
    assert a.in_flight == 2 and a.batch == 32 and a.device_list == [0, 0]


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


def test_report_configs_4_and_5():
    """--config 4 / 5: same line format, the config's own metric, per-GPU batch and algorithmic bytes"""
    r4 = dict(_rank(0.0355, 0.71, 0.0, [0, 15, 31]))
    r4.pop("copy_ceiling_gbs"), r4.pop("copy_ceiling_kernel")
    a = argparse.Namespace(batch=32, steps=50, warmup=5, config=4, unfused=False)
    out, bad = bench.report(a, 1, [r4])
    assert not bad and "8K warpAffine" in out["metric"] and out["config"]["frames_per_gpu"] == 32
    assert out["roofline"]["alg_bytes_per_launch"] == 32 * 1080 * 1920 * 30 and "frac_of_copy_ceiling" not in out["roofline"]
    assert abs(out["value"] - 32 * 1080 * 1920 * 50 / 0.0355 / 1e6) < 1 and "fused" in out["config"]["path"] and "output pixels" in out["metric"]
    assert abs(out["config"]["input_mpix_s"] - 32 * 4320 * 7680 * 50 / 0.0355 / 1e6) < 1
    a.unfused = True
    out, _ = bench.report(a, 1, [r4])
    assert out["roofline"]["alg_bytes_per_launch"] == 32 * (4320 * 7680 * 6 + 1080 * 1920 * 15) and "two launches" in out["config"]["path"]
    a = argparse.Namespace(batch=64, steps=50, warmup=5, config=5)
    out, bad = bench.report(a, 8, [dict(r4, verified_frames=[64 * r, 64 * r + 31, 64 * r + 63]) for r in range(8)])
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 512 and "cornerHarris" in out["metric"]
    assert out["roofline"]["alg_bytes_per_launch"] == 64 * 2160 * 3840 * 4 and len(out["roofline"]["launch_ms_per_gpu"]) == 8
    assert bench.CONFIGS[3]["batch"] == 64 and bench.CONFIGS[4]["batch"] * 8 == 256 and bench.CONFIGS[5]["batch"] * 8 == 512


def test_report_memory_only_and_clock():
    a = argparse.Namespace(batch=64, steps=50, warmup=5)
    r = dict(_rank(0.0275, 0.55, 6000.0, [0, 31, 63]), memory_only_gbs=5800.0, shader_mhz_under_load=2100.0)
    out, _ = bench.report(a, 1, [r])
    assert out["roofline"]["memory_only_gbs"] == 5800.0 and abs(out["roofline"]["frac_of_memory_only"] - out["roofline"]["achieved"] / 5800.0) < 1e-4
    assert out["roofline"]["shader_mhz_under_load"] == 2100.0


def test_warp_matrix_is_the_survey_matrix():
    import numpy as np
    M = bench.warp_matrix()
    t = np.deg2rad(7.0)
    assert M.dtype == np.float32 and abs(M[0] - np.cos(t)) < 1e-6 and abs(M[3] - np.sin(t)) < 1e-6
    # the centre of the frame maps to the centre + the translation
    cx, cy = 3840.0, 2160.0
    assert abs(M[0] * cx + M[1] * cy + M[2] - (cx + 13.25)) < 1e-2 and abs(M[3] * cx + M[4] * cy + M[5] - (cy - 8.5)) < 1e-2


def test_bench_kernel_matches_the_oracle_generator(oracle):
    import numpy as np
    assert np.array_equal(bench.bench_kernel7(), oracle.bench_kernel7())


def test_dispatch_overhead_summary():
    """tools/dispatch_overhead.py's report arithmetic: medians, ratio against G = 1, and it never claims more than one device"""
    from tools import dispatch_overhead as do
    raw = {1: {"wall_s": [0.12, 0.11, 0.13], "steps": 200, "enqueue_us": [6.0], "frames": 64},
           8: {"wall_s": [0.1144, 0.1133, 0.1122], "steps": 200, "enqueue_us": [9.0] * 7 + [17.0], "frames": 8}}
    out = do.summarize(raw)
    assert out["n_devices"] == 1 and [r["contexts"] for r in out["rows"]] == [1, 8]
    assert abs(out["rows"][0]["wall_ms_per_step"] - 0.6) < 1e-9 and out["rows"][0]["vs_G1"] == 1.0
    assert abs(out["rows"][1]["vs_G1"] - 0.1133 / 0.12) < 1e-3 and out["rows"][1]["enqueue_us"] == 10.0 and out["rows"][1]["enqueue_us_max"] == 17.0
    assert abs(out["loss_at_max_G"] - (0.1133 / 0.12 - 1.0)) < 1e-3
    full = do.summarize({1: raw[1], 8: dict(raw[8], wall_s=[0.98, 0.97, 0.99], frames=64)}, full=True)   # 8 x the work on one GPU
    assert abs(full["rows"][1]["vs_G1"] - (0.98 / 200) / (8 * 0.12 / 200)) < 1e-3 and full["n_devices"] == 1
