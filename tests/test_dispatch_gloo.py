"""The N>1 path on CPU: two processes over gloo exercise the sharding + barrier/MAX-reduce timing that
bench.py uses across GPUs (one rank per GPU, frames sharded, no data-path collective)."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from caffe_rtpose_amd.dispatch import frame_shard, timed_region, aggregate_fps
    lo, hi = frame_shard(101, rank, world)
    calls = []

    def run(n, base):
        calls.append((n, base))
        time.sleep(0.05 * (rank + 1) * (1 if base else 0))  # rank 1 is the slow one in the timed region

    dt = timed_region(run, steps=10, warmup=3, dist=dist)
    owned = torch.zeros(101, dtype=torch.int32)
    owned[lo:hi] = 1
    dist.all_reduce(owned)
    out.put((rank, lo, hi, dt, calls, owned.tolist(), aggregate_fps(10, world, dt)))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, dt0, calls0, owned0, fps0), (r1, lo1, hi1, dt1, calls1, owned1, fps1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 51, 51, 101)          # contiguous, sizes differ by <= 1
    assert owned0 == [1] * 101 and owned1 == [1] * 101       # every frame owned exactly once
    assert calls0 == [(3, 0), (10, 1 << 20)] == calls1        # W untimed warm-up steps, then EXACTLY K timed
    assert dt0 == dt1 and 0.09 < dt0 < 0.5                    # MAX over ranks (rank 1 sleeps 0.1 s)
    assert fps0 == fps1 == 20 / dt0                            # whole-job aggregate, not per-GPU


def test_shards_cover_everything_for_any_world():
    from caffe_rtpose_amd.dispatch import frame_shard
    for total in (0, 1, 7, 256, 1001):
        for world in (1, 2, 3, 4, 8):
            spans = [frame_shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
