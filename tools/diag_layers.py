#!/usr/bin/env python
"""Per-layer error of the HIP conv stack vs the CPU oracle (test tooling; needs a GPU)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import _oracle as orc  # noqa: E402
import _synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401  (experiments build of the library)
import caffe_rtpose_amd as r  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=656)
ap.add_argument("--h", type=int, default=368)
ap.add_argument("--n", type=int, default=1)
ap.add_argument("--model", type=int, default=0)
args = ap.parse_args()
x = _synth.random_frame(args.n, args.h, args.w, seed=3)
net = None
for prec, pname in ((r.PREC_FP32, "fp32"), (r.PREC_FP16, "fp16")):
    e = r.Engine(r.Config(model=args.model, net_w=args.w, net_h=args.h, num_scales=args.n, precision=prec, frames_in_flight=1, scale_gap=0.15, keep_blobs=1))
    if net is None:
        net = orc.Net(args.model)
        for i in range(len(net.convs)):
            net.set_weights(i, *e.get_conv_weights(i))
        net.forward(x, keep_all=True)
    d = e.forward_debug(x)
    print(f"== {pname} {args.w}x{args.h} N={args.n}: stage ms {e.last_stage_ms()}")
    rows = []
    for name, *_ in net.convs:
        ref = net.blob(name)
        g = e.get_blob(name)
        rows.append((name, float(np.abs(g - ref).max()), float(np.abs(ref).max())))
    for name, err, mag in rows[:4] + rows[10:16] + rows[-6:]:
        print(f"  {name:22s} max|err| {err:.3e}  max|ref| {mag:.3f}  rel {err / mag:.2e}")
    ref = net.blob("concat_stage7")
    err = np.abs(d["lowres"] - ref)
    print(f"  concat_stage7: max|err| {err.max():.3e} mean|err| {err.mean():.3e} max|ref| {np.abs(ref).max():.3f}")
    ref_res = orc.imresize(ref, args.w, args.h, 1.0, 0.15)[0]
    rerr = np.abs(d["resized"] - ref_res)
    print(f"  resized_map (oracle chain vs engine chain): max|err| {rerr.max():.3e}")
    thr = e.get_thresholds()
    pk = orc.nms(ref_res, e.num_parts, e.max_peaks, thr["nms_threshold"])
    print(f"  peak totals oracle {pk[:, 0, 0].astype(int).tolist()}\n  peak totals engine {d['peaks'][:, 0, 0].astype(int).tolist()}")
    e.close()
