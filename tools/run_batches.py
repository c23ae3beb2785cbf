#!/usr/bin/env python
"""Whole batches ONE AT A TIME (submit B resident frames, collect them, repeat): the launch sequence of the plan in plan order, for counter passes.
   python tools/run_batches.py B N_SCALES PREC NBATCHES [coco|mpi]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import caffe_rtpose_amd as r  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prec = {"fp32": r.PREC_FP32, "mixed": r.PREC_MIXED, "f16x3": r.PREC_F16X3, "fp16": r.PREC_FP16}[sys.argv[3] if len(sys.argv) > 3 else "mixed"]
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 6
model = sys.argv[5] if len(sys.argv) > 5 else "coco"
kw = dict(model=r.MODEL_MPI_15, net_w=496, net_h=368) if model == "mpi" else {}
e = r.Engine(r.Config(precision=prec, num_scales=N, scale_gap=0.15 if N > 1 else 0.3, frames_in_flight=B, batch_frames=B, **kw))
rs = np.random.RandomState(1)
dev = [e.device_frame(rs.randint(0, 256, (N, 3, e.net_h, e.net_w)).astype(np.float32) / 256 - 0.5) for _ in range(4)]
for b in range(nb):
    for j in range(B):
        e.submit_device(dev[(b * B + j) % 4], tag=b * B + j)
    for j in range(B):
        e.collect()
print(f"{nb} batches of {B} done")
e.close()
