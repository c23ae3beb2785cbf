#!/usr/bin/env python
"""A few synchronous frames through the PRODUCTION path (submit/collect, one in flight), for
rocprofv3 --kernel-trace: per-kernel time without cross-frame overlap."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401  (experiments build of the library)
import caffe_rtpose_amd as r  # noqa: E402
prec = r.PREC_FP32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else r.PREC_FP16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
e = r.Engine(r.Config(precision=prec, num_scales=n, scale_gap=0.15, frames_in_flight=1))
x = _synth.random_frame(n, 368, 656, seed=3)
for _ in range(20):
    e.submit(x)
    e.collect()
print(e.last_stage_ms())
e.close()
