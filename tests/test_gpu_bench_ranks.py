"""bench.py's N > 1 code paths, executed on the hardware that is here: the driver's SCALE run must not be
their first execution.  Ranks time-share GPU 0 (`--devices 0,0`, a test-only flag: the line then says n_gpus = 1, contexts = 2 --
never a scaling claim).  (a) the one-process form: one host thread per rank, thread barrier, per-rank verification, the fenced
memory-only leg; (b) the one-process-per-GPU form under torch.distributed.run: RCCL barrier + all_gather_object (if RCCL refuses two
ranks on one device the same branch runs over gloo -- the bench code is identical, only the barrier's transport differs); (c) a
failing rank ends the job with a non-zero exit instead of hanging it, in both forms."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "4", "--warmup", "2", "--sustained", "12", "--settle-ms", "20", "--batch", "4", "--no-cpu", "--no-probe", "--in-flight", "2"]   # (two contexts per rank: the rounds 4-5 recipe stays covered)


def _run(cmd, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    return p, line


def _check_two_ranks(out):
    assert out["n_gpus"] == 1 and out["config"]["contexts"] == 2 and "not a scaling" in out["config"]["note"]
    assert out["verified"].startswith("bit-exact")
    # 2 ranks x 2 contexts x 4 frames: first / middle / last frame of every context's batch
    assert out["verified_frames"] == [0, 1, 3, 4, 5, 7, 8, 9, 11, 12, 13, 15]
    assert len(out["roofline"]["launch_ms_per_gpu"]) == 2 and all(t > 0 for t in out["roofline"]["launch_ms_per_gpu"])
    assert out["roofline"]["in_flight"] == 2 and out["config"]["global_batch"] == 16 and "cpu_baseline" not in out and "other_configs" not in out


def test_one_process_two_ranks_on_one_device():
    p, out = _run([sys.executable, "bench.py", "--gpus", "2", "--devices", "0,0"] + SMALL)
    assert p.returncode == 0, p.stderr[-2000:]
    _check_two_ranks(out)
    assert out["roofline"]["memory_only_gbs"] > 0 and out["roofline"]["copy_ceiling_gbs"] > 0    # the fenced legs ran on both ranks


def test_one_process_failing_rank_exits_nonzero():
    p, out = _run([sys.executable, "bench.py", "--gpus", "2", "--devices", "0,0", "--fail-rank", "1", "--no-ceiling"] + SMALL, timeout=300)
    assert p.returncode != 0 and out is None and "injected failure" in p.stderr


def _torchrun(extra, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
            "bench.py", "--gpus", "2", "--devices", "0,0", "--no-ceiling"] + SMALL + extra


def test_torchrun_two_ranks_on_one_device():
    p, out = _run(_torchrun([], 29531))
    backend = "nccl"
    if p.returncode != 0:   # RCCL refuses two ranks on one device ("Duplicate GPU detected"): the same branch over gloo
        backend = "gloo"
        p, out = _run(_torchrun(["--dist-backend", "gloo"], 29532))
    assert p.returncode == 0, (backend, p.stderr[-3000:])
    _check_two_ranks(out)


def test_torchrun_failing_rank_exits_nonzero():
    p, out = _run(_torchrun(["--dist-backend", "gloo", "--fail-rank", "1"], 29533), timeout=300)
    assert p.returncode != 0 and out is None


def test_one_process_eight_ranks_on_one_device():
    """the EIGHT-rank one-process form (the shape of the driver's first SCALE run: 8 host threads, 8 x 2 contexts, the
    thread barrier with 8 parties, per-rank verification and the report over 8 results) executed before the driver does -- all ranks on GPU 0"""
    p, out = _run([sys.executable, "bench.py", "--gpus", "8", "--devices", "0,0,0,0,0,0,0,0", "--no-ceiling"] + SMALL)
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["n_gpus"] == 1 and out["config"]["contexts"] == 8 and "not a scaling" in out["config"]["note"]
    assert out["verified"].startswith("bit-exact")
    # 8 ranks x 2 contexts x 4 frames = 16 lanes: first / middle / last frame of every lane's batch
    assert out["verified_frames"] == sorted(4 * lane + i for lane in range(16) for i in (0, 1, 3)) and out["config"]["global_batch"] == 64
    assert len(out["roofline"]["launch_ms_per_gpu"]) == 8 and all(t > 0 for t in out["roofline"]["launch_ms_per_gpu"])
    assert out["roofline"]["in_flight"] == 2 and "cpu_baseline" not in out and "other_configs" not in out


def test_torchrun_one_rank_rccl():
    """the RCCL transport itself (what carries the barrier and the gather of the per-rank timings on a multi-GPU node): one rank under
    torch.distributed.run with the default backend -- init_process_group("nccl"), barrier, all_gather_object, destroy"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29551",
           "bench.py", "--gpus", "1", "--no-ceiling", "--no-others"] + SMALL
    p, out = _run(cmd, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert out["n_gpus"] == 1 and out["verified"].startswith("bit-exact") and out["roofline"]["in_flight"] == 2 and out["config"]["global_batch"] == 8


def test_torchrun_eight_ranks_on_one_device():
    """the form the driver's SCALE run uses (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N) with EIGHT ranks, all on
    GPU 0: eight processes, barrier + all_gather_object over the process group (RCCL refuses eight ranks on one device: gloo carries the
    same branch), every rank's frames verified, the report over eight results"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29541",
           "bench.py", "--gpus", "8", "--devices", "0,0,0,0,0,0,0,0", "--no-ceiling", "--dist-backend", "gloo"] + SMALL
    p, out = _run(cmd, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert out["n_gpus"] == 1 and out["config"]["contexts"] == 8 and out["verified"].startswith("bit-exact")
    assert out["verified_frames"] == sorted(4 * lane + i for lane in range(16) for i in (0, 1, 3)) and out["config"]["global_batch"] == 64
    assert len(out["roofline"]["launch_ms_per_gpu"]) == 8 and all(t > 0 for t in out["roofline"]["launch_ms_per_gpu"])
    assert "cpu_baseline" not in out and "other_configs" not in out


def test_default_run_carries_the_other_configs_small():
    """the default (N = 1) line at a reduced batch: other_configs has EVERY BASELINE config beside the headline (round 6: "1" the reference's own
    640x480 YUYV -> BGR + rectangle path with its CPU timing on one thread and all cores and the GPU latency; "2" one 1080p frame 5x5 in us per
    launch with the floors of the same run), the Sobel half of config 3 ("3s", "3f") and configs 4 / 5 (round 6: "5" also with the worst-case
    launch -- noise, thr = -inf), every record with its verification and cpu_baseline; the one-stream figure is flattened into scalars"""
    p, out = _run([sys.executable, "bench.py", "--steps", "4", "--warmup", "2", "--sustained", "12", "--settle-ms", "20", "--no-probe", "--no-ceiling",
                   "--cpu-seconds", "1", "--other-cpu-seconds", "0.5"], timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert set(out["other_configs"]) == {"1", "2", "3s", "3f", "4", "5"}
    for key, rec in out["other_configs"].items():
        assert rec["verified"].startswith("bit-exact"), (key, rec["verified"])
        assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] >= 1
        if key in ("1", "2"):
            continue
        assert 0.05 < rec["roofline"]["frac"] < 1.0 and rec["roofline"]["launch_ms"] > 0
        assert rec["roofline"]["in_flight2_launch_ms"] > 0 and 0.05 < rec["roofline"]["in_flight2_frac"] < 1.0
    c1, c2, c5 = out["other_configs"]["1"], out["other_configs"]["2"], out["other_configs"]["5"]
    assert c1["cpu_baseline"]["value_1thread"] > 0 and c1["alg_bytes_per_px"] == 5 and "roofline" not in c1       # the CPU path is this config's headline
    assert 1.0 < c1["gpu_device_resident"]["us_per_frame"] < 200.0 and c1["gpu_host_mat"]["us_per_frame"] > c1["gpu_device_resident"]["us_per_frame"]
    assert c2["unit"] == "us" and c2["higher_is_better"] is False and "roofline" not in c2                            # latency config: no HBM fraction claimed
    assert 1.0 < c2["floors"]["empty_kernel_us"] <= c2["value"] * 1.05 and c2["floors"]["copy_of_the_frame_us"] > 1.0 and c2["value"] < 50.0
    assert c5["roofline"]["worst_case_launch_ms"] >= 0.9 * c5["roofline"]["launch_ms"] and c5["roofline"]["worst_case_verified"].startswith("bit-exact")
    assert out["other_configs"]["3s"]["roofline"]["alg_bytes_per_launch"] == 64 * 2160 * 3840 * 7
    # round 6: the default headline is ONE context (BASELINE's literal batch; the library runs a call as two halves on the context's two streams); the
    # one-launch form and the two-batches-in-flight form ride beside it, the flat single_stream_* scalars repeat the headline
    r = out["roofline"]
    assert r["in_flight"] == 1 and out["config"]["global_batch"] == 64 and out["config"]["launches_per_step_per_gpu"] == 1 and "call_form" in out["config"]
    assert r["single_stream_frac"] == r["frac"] and 0.05 < r["single_stream_one_launch_frac"] < 1.0 and 0.05 < r["in_flight2_frac"] < 1.0
    assert out["verified_frames"] == [0, 31, 63]
    assert out["cpu_baseline"]["value"] > 0
