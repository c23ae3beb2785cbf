#!/usr/bin/env python
"""Pipelined frames/s of one engine for A/B runs: python tools/bench_input.py host|resident B IN_FLIGHT [seconds] [model] [precision].
host = pageable u8 1280x720 frames through rtp_submit_frame (bench.py's headline path), resident = rtp_submit_device.
Runs against the experiments library (RTP_IN_STREAM, RTP_HALF_CHIP, ... are read there)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401
import numpy as np  # noqa: E402
import caffe_rtpose_amd as r  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "host"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 7
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
model = sys.argv[5] if len(sys.argv) > 5 else "coco"
prec = {"mixed": r.PREC_MIXED, "fp16": r.PREC_FP16, "f16x3": r.PREC_F16X3}[sys.argv[6] if len(sys.argv) > 6 else "mixed"]
W, H = (656, 368) if model == "coco" else (496, 368)
e = r.Engine(r.Config(model=0 if model == "coco" else 1, net_w=W, net_h=H, precision=prec, frames_in_flight=depth, batch_frames=B))
u8 = [r.synth_frame(1280, 720, i, seed=2) for i in range(8)]
rs = np.random.RandomState(1)
dev = [e.device_frame(rs.randint(0, 256, (1, 3, H, W)).astype(np.float32) / 256 - 0.5) for _ in range(8)]


def run(k):
    sub = col = 0
    while col < k:
        while sub < k and e.in_flight() < depth:
            if mode == "host":
                e.submit_frame(u8[sub % 8], tag=sub)
            else:
                e.submit_device(dev[sub % 8], tag=sub)
            sub += 1
        e.collect()
        col += 1


run(60)
t = time.perf_counter(); run(200); est = 200 / (time.perf_counter() - t)
n = max(200, int(est * secs))
best = []
for rep in range(2):
    e.synchronize()
    t = time.perf_counter(); run(n); e.synchronize(); best.append(n / (time.perf_counter() - t))
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RTP_") and k != "RTP_LIB")
print(f"[{tag}] {model} {mode} B={B} in_flight={depth}: {best[0]:.1f} / {best[1]:.1f} frames/s")
e.close()
