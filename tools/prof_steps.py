#!/usr/bin/env python
"""Per-step table of the conv plan: tile, workgroups, solo time, TFLOP/s (diagnostics)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401  (experiments build of the library)
import caffe_rtpose_amd as r  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
prec = {"fp32": r.PREC_FP32, "mixed": r.PREC_MIXED, "f16x3": r.PREC_F16X3}.get(sys.argv[3] if len(sys.argv) > 3 else "fp16", r.PREC_FP16)
cfg = r.Config(precision=prec, num_scales=N, scale_gap=0.15, frames_in_flight=B, batch_frames=B)
lines = [l for l in r.plan_summary(cfg).splitlines() if l.startswith("step")]
e = r.Engine(cfg)
ms, gf = e.profile_steps(30)
tot = sum(ms)
print(f"B={B} N={N}: {tot:.3f} ms per batch, {tot / B:.3f} ms per frame, {sum(gf) / tot:.0f} TFLOP/s")
agg = {}
for l, m, g in zip(lines, ms, gf):
    w = l.split()
    key = " ".join(w[w.index("k"):w.index("relu")] + w[w.index("tile"):w.index("dsts")]) if w[1] == "conv" else w[1]
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += m; a[2] += g
    if "-v" in sys.argv:
        print(f"{m * 1e3:8.1f} us {g / m if m else 0:7.0f} TF  {l[:150]}")
for k, (n, m, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{m * 1e3:8.1f} us total {100 * m / tot:5.1f}%  x{n:2d}  {m / n * 1e3:7.1f} us each {g / m if m else 0:6.0f} TF  {k}")
e.close()
