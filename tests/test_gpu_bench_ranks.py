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
SMALL = ["--steps", "4", "--warmup", "2", "--sustained", "12", "--settle-ms", "20", "--batch", "4", "--no-cpu", "--no-probe"]


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
