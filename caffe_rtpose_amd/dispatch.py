"""Frame sharding + timing helpers for multi-GPU runs (one process per GPU, torch.distributed).

The reference shards FRAMES across GPUs with full replicas and no collective on the data path
(rtpose.cpp:1463-1472,1107).  In-process that is rtpose.bin's shared queue; across
processes (bench.py under torch.distributed.run) every rank owns a contiguous block of the frame
indices, and the only communication is the barrier / MAX-reduce around the timed region.
"""
import time


def frame_shard(total_frames, rank, world):
    """Contiguous block [lo, hi) of the global frame indices owned by `rank`; blocks differ by <= 1."""
    base, rem = divmod(total_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_region(run_fn, steps, warmup, dist=None, device_sync=None, reduce_device=None, return_local=False):
    """warmup untimed steps, then `steps` steps bracketed by barrier + device sync on both sides.
    Returns the MAX over ranks of the elapsed seconds (what the whole job waited for); with return_local also this rank's own."""
    def fence():
        if device_sync:
            device_sync()
        if dist is not None:
            dist.barrier()
        if device_sync:
            device_sync()

    if warmup:
        run_fn(warmup, 0)
    fence()
    t0 = time.perf_counter()
    run_fn(steps, 1 << 20)
    fence()
    dt = time.perf_counter() - t0
    dt_local = dt
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=reduce_device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return (dt, dt_local) if return_local else dt


def scaled_steps(steps, est_steps_per_s, min_seconds, dist=None):
    """Number of steps to time: at least `steps`, and enough for a timed region of `min_seconds` at the
    estimated per-rank rate; every rank gets the same number (MAX over ranks)."""
    import math
    n = max(int(steps), int(math.ceil(min_seconds * max(est_steps_per_s, 0.0))))
    if dist is not None:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([n], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = int(t.item())
    return n


def aggregate_fps(steps_per_rank, world, seconds):
    """Whole-job throughput: every rank processed steps_per_rank frames (weak scaling)."""
    return steps_per_rank * world / seconds


def broadcast_bytes(buf, src, dist, device="cpu"):
    """One-time broadcast of a host byte blob (numpy uint8, same length on every rank) from rank `src`: the weight blob of
    bench.py --broadcast_weights (the only collective this path can use, and not on the per-frame path).
    With the nccl backend the bytes go host -> device -> RCCL broadcast over xGMI -> host; with gloo over TCP."""
    import numpy as np
    import torch
    t = torch.from_numpy(np.ascontiguousarray(buf, np.uint8))
    if device != "cpu":
        t = t.to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()
