#!/usr/bin/env python
"""Post-processing alone on the chip (production path from low-res maps): nms / connect ms for noise maps and 1 / 5 / 20 planted people."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401
import numpy as np  # noqa: E402
import caffe_rtpose_amd as r  # noqa: E402
import _synth  # noqa: E402

e = r.Engine(r.Config(net_w=656, net_h=368, precision=r.PREC_MIXED, frames_in_flight=2, batch_frames=1))
tables = r.model_tables(0)
cases = [("noise", _synth.smooth_field(e.heat_channels, e.low_h, e.low_w, seed=3)[None])]
for P in (1, 5, 20):
    cases.append((f"P{P}", _synth.people_lowres(0, tables, P, e.low_h, e.low_w, seed=3, N=1)[0]))
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RTP_") and k != "RTP_LIB")
for name, low in cases:
    low = np.ascontiguousarray(low, np.float32)
    t = []
    for it in range(12):
        _, _, n = e.post_from_lowres(low)
        t.append(e.last_stage_ms())
    t = t[2:]
    print(f"[{tag}] {name}: people {n}  nms {np.mean([x['nms'] for x in t]) * 1e3:.1f} us  connect {np.mean([x['connect'] for x in t]) * 1e3:.1f} us")
e.close()
