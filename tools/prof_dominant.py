#!/usr/bin/env python
"""Run only the dominant conv launch many times (for rocprofv3 counter passes / A-B env toggles)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401  (experiments build of the library)
import caffe_rtpose_amd as r  # noqa: E402
prec = {"fp32": r.PREC_FP32, "mixed": r.PREC_MIXED, "f16x3": r.PREC_F16X3}.get(sys.argv[1] if len(sys.argv) > 1 else "fp16", r.PREC_FP16)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
scales = int(sys.argv[4]) if len(sys.argv) > 4 else 1                     # usage: prof_dominant.py PREC ITERS [BATCH] [SCALES] [coco|mpi]  (bench.py's pmc_traffic() reads this header)
model = sys.argv[5] if len(sys.argv) > 5 else "coco"
kw = dict(model=r.MODEL_MPI_15, net_w=496, net_h=368) if model == "mpi" else {}
e = r.Engine(r.Config(precision=prec, frames_in_flight=batch, batch_frames=batch, num_scales=scales, scale_gap=0.15 if scales > 1 else 0.3, **kw))
ms, fl = e.bench_dominant_conv(iters)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RTP_"))
print(f"[{tag}] dominant conv {ms * 1e3:.1f} us = {fl / ms / 1e9:.1f} TFLOP/s")
e.close()
