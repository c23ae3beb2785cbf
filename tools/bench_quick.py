#!/usr/bin/env python
"""Pipelined FPS only (no roofline / cpu baseline): quick A/B of engine variants via RTP_* env vars."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401  (experiments build of the library)
import caffe_rtpose_amd as r
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prec = r.PREC_FP32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else r.PREC_FP16
e = r.Engine(r.Config(num_scales=n, scale_gap=0.15, precision=prec, frames_in_flight=depth))
frames = [(torch.randint(0, 256, (n, 3, 368, 656)).float() / 256 - 0.5).cuda() for _ in range(4)]
torch.cuda.synchronize()
def run(k):
    sub = col = 0
    while col < k:
        while sub < k and e.in_flight() < depth:
            e.submit_device(frames[sub % 4].data_ptr(), sub); sub += 1
        e.collect(); col += 1
run(40)
t = time.perf_counter(); run(300); dt = time.perf_counter() - t
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RTP_"))
print(f"[{tag}] depth {depth} N={n}: {300 / dt:.1f} FPS  ({dt / 300 * 1e3:.3f} ms/frame)")
e.close()
