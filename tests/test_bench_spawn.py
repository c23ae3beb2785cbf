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
