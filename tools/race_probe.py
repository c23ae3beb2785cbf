#!/usr/bin/env python
"""Determinism probe: the same frame many times, sync vs async-under-load (test tooling)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import _synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401  (experiments build of the library)
import caffe_rtpose_amd as r  # noqa: E402

W, H, N, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
prec = r.PREC_FP32 if (len(sys.argv) > 5 and sys.argv[5] == "fp32") else r.PREC_FP16
e = r.Engine(r.Config(net_w=W, net_h=H, num_scales=N, scale_gap=0.25, frames_in_flight=depth, precision=prec))
x = _synth.random_frame(N, H, W, seed=100)
lows = [e.forward_heatmaps(x) for _ in range(6)]
sync_bad = sum(not np.array_equal(lows[0], l) for l in lows[1:])
ref = e.forward_debug(x)
res = []
sub = 0
while sub < 40 or e.in_flight():
    while sub < 40 and e.in_flight() < depth:
        e.submit(x, tag=sub)
        sub += 1
    res.append(e.collect())
bad = sum(1 for _, n, j in res if n != ref["num_people"] or not np.array_equal(j, ref["joints"][:n]))
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RTP_"))
print(f"[{tag}] {W}x{H} N={N} depth={depth} {'fp32' if prec else 'fp16'}: sync lowres mismatches {sync_bad}/5, async joint mismatches {bad}/40 (ref people {ref['num_people']})")
e.close()
