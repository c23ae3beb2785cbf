#!/usr/bin/env python
"""Quick A/B of the conv stack: dominant-kernel time (HIP events) and per-stage ms of one frame."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import _synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401  (experiments build of the library)
import caffe_rtpose_amd as r  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for prec, pname, peak in ((r.PREC_FP16, "fp16", 2500.0), (r.PREC_FP32, "fp32", 157.3)):
    e = r.Engine(r.Config(num_scales=n, precision=prec, frames_in_flight=1, scale_gap=0.15))
    x = _synth.random_frame(n, 368, 656, seed=3)
    e.forward_debug(x)
    d = e.forward_debug(x)
    ms, fl = e.bench_dominant_conv(100)
    print(f"impl={os.environ.get('RTP_CONV_IMPL', 'ring')} {pname} N={n}: dominant conv {ms * 1e3:.1f} us/launch = {fl / ms / 1e9:.1f} TFLOP/s "
          f"({fl / ms / 1e9 / peak * 100:.1f}% of peak); frame stages ms {e.last_stage_ms()}")
    e.close()
