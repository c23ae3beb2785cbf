#!/usr/bin/env python
"""Kernel residency of the PIPELINED loop from device-side stamps (rtp_stamp_probe; no profiler):  python tools/stamp_timeline.py [host|resident] [B] [IN_FLIGHT] [frames]
Per plan step: how long the launch is resident in the pipeline (first workgroup start .. last workgroup end) against the same step alone on the
chip (rtp_profile_steps), and the gap between the end of the previous step of the same batch and its start.  Then the chip-level account
(bench.py stamp_account) and where the time without a convolution kernel goes.  Saves the raw spans to gpurun_out/stamps.npy."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # like bench.py and rtpose.bin: one hardware queue per batch context (engine.cpp "hardware queues")
import numpy as np
import caffe_rtpose_amd as r

mode = sys.argv[1] if len(sys.argv) > 1 else "host"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 7
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 600
spec = importlib.util.spec_from_file_location("rtp_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
cfg = r.Config(net_w=656, net_h=368, precision=r.PREC_MIXED, frames_in_flight=depth, batch_frames=B)
e = r.Engine(cfg)
plan = [ln for ln in r.plan_summary(cfg).splitlines() if ln.startswith("step ")]
alone = e.profile_steps(20)[0]
u8 = [r.synth_frame(1280, 720, i, seed=2) for i in range(8)]
rs = np.random.RandomState(1)
dev = [e.device_frame(rs.randint(0, 256, (1, 3, 368, 656)).astype(np.float32) / 256 - 0.5) for _ in range(8)]


def run(k):
    sub = col = 0
    while col < k:
        while sub < k and e.in_flight() < depth:
            (e.submit_frame(u8[sub % 8], tag=sub) if mode == "host" else e.submit_device(dev[sub % 8], tag=sub))
            sub += 1
        e.collect()
        col += 1


run(100)
e.stamp_probe(1)
import time
t0 = time.perf_counter(); run(frames); e.synchronize(); fps = frames / (time.perf_counter() - t0)
sp = e.stamp_probe(-1)
e.stamp_probe(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "stamps.npy"), sp)
print(f"{mode} B={B} in_flight={depth}: {fps:.1f} frames/s with the probe on, {len(sp)} launches stamped")
acc = bench.stamp_account(sp)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in acc.items() if k != "what"})
# batches = runs of increasing slot ids (one harvest each)
batches, cur = [], []
for row in sp:
    if cur and row[0] <= cur[-1][0]:
        batches.append(np.array(cur)); cur = []
    cur.append(row)
if cur:
    batches.append(np.array(cur))
batches = batches[len(batches) // 10:]
ns = len(plan)
dur = {i: [] for i in range(ns)}
gap = {i: [] for i in range(ns)}
stack, post = [], []
for b in batches:
    conv = b[b[:, 0] < 64]
    if len(conv) < ns - 1:
        continue
    prev_end = None
    for row in conv:
        i = int(row[0])
        dur[i].append(row[2] - row[1])
        if prev_end is not None:
            gap[i].append(row[1] - prev_end)
        prev_end = row[2]
    stack.append(conv[:, 2].max() - conv[:, 1].min())
    pc = b[(b[:, 0] >= 64) & (b[:, 0] < 200)]
    if len(pc):
        post.append(pc[:, 2].max() - pc[:, 1].min())
print(f"{len(batches)} batches: conv stack first start .. last end {np.mean(stack):.0f} us on average (alone: the sum below), post chains {np.mean(post):.0f} us")
tot_p = tot_a = tot_g = 0.0
for i in range(ns):
    if not dur[i]:
        continue
    a = alone[i] * 1e3 if alone is not None and i < len(alone) else float("nan")
    g = np.mean(gap[i]) if gap[i] else 0.0
    tot_p += np.mean(dur[i]); tot_g += g; tot_a += (a if a == a else 0)
    print(f"  step {i:2d} resident {np.mean(dur[i]):7.1f} us (alone {a:6.1f})  gap before {g:6.1f} us   {plan[i][5:90]}")
print(f"sum: resident {tot_p:.0f} us, gaps {tot_g:.0f} us, alone {tot_a:.0f} us")
e.close()
