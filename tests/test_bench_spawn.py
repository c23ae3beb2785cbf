"""`python bench.py --gpus N` started BARE (no launcher) must start N ranks itself, the way the reference starts
NUM_GPU workers from one command (rtpose.cpp:1463-1472).  CPU check of exactly that plumbing: --dry_dispatch runs
the spawn, the gloo rendezvous, the barrier/MAX-reduce timing and the one-line report without touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry_dispatch", "--steps", "20", "--warmup", "2"] + extra,
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    return [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bare_gpus_2_spawns_two_ranks_and_prints_one_line():
    lines = _run(["--gpus", "2"])
    assert len(lines) == 1                       # rank 0 only
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 2
    assert out["steps_timed"] >= 20              # --steps is a minimum: scaled to --min_seconds, same on every rank
    assert out["value"] > 0
    assert len(out["per_rank_frames_per_s"]) == 2 and all(v > 0 for v in out["per_rank_frames_per_s"])   # a straggler would show here


def test_single_rank_needs_no_launcher():
    lines = _run(["--gpus", "1"])
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1


def test_rccl_failure_falls_back_to_gloo_and_says_so():
    """bench.py asks for the nccl (= RCCL) process group where torch sees a GPU.  RCCL runs on torch's bundled HIP runtime next to the
    engine's system runtime; if it cannot start, every rank must fall back to gloo — not hang, not die — and the line must say which
    backend carried the barrier.  Here (no GPU) --comm nccl fails on every rank at init."""
    lines = _run(["--gpus", "2", "--comm", "nccl", "--broadcast_weights"])
    assert len(lines) == 1
    out = lines[0]
    assert out["comm_backend"] == "gloo" and "nccl failed" in out["comm_note"] and "fell back to gloo" in out["comm_note"]
    assert out["n_gpus"] == 2 and len(out["per_rank_frames_per_s"]) == 2 and out["value"] > 0
    assert out["weight_blob_broadcast_ok"] is True       # the one-time weight-blob broadcast (rank 0 -> all) arrives intact on every rank


def test_auto_comm_is_gloo_without_a_gpu():
    out = _run(["--gpus", "2"])[0]
    assert out["comm_backend"] == "gloo" and out["comm_note"] is None


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_with_weight_broadcast():
    """The N > 1 path of bench.py with REAL engines, as far as one GPU can show it: two ranks under torch.distributed.run, both on
    device 0 (--devices 0,0), gloo process group next to the engines' HIP runtime, rank 0's packed weight blob (arena + Caffe-layout
    floats) broadcast to rank 1 and imported, barrier + MAX-reduce around the timed region, one line from rank 0 with both ranks' rates."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0", "--comm", "gloo", "--broadcast_weights", "--steps", "60", "--warmup", "10",
                        "--min_seconds", "0.5", "--no_cpu_baseline", "--no_sub_results", "--no_parity"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and out["comm_backend"] == "gloo" and out["scaling"] == "weak"
    assert len(out["per_rank_frames_per_s"]) == 2 and min(out["per_rank_frames_per_s"]) > 100
    assert abs(out["value"] - sum(out["per_rank_frames_per_s"])) < 0.25 * out["value"]        # whole-job aggregate (the slowest rank's clock)
    assert out["weight_broadcast"]["bytes"] > 250e6 and 0.05 < out["roofline"]["frac"] < 1
