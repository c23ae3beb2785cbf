#!/usr/bin/env python
"""Bit-level A/B of engine variants selected by RTP_* environment variables: prints a SHA-256 of the low-res maps (and of a
few intermediate blobs) of seeded frames, for several plans.  Run once per variant and diff the output: kernels that only
change HOW fragments reach the MFMAs (not the order of the multiply-adds) must print identical hashes."""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import caffe_rtpose_amd as r  # noqa: E402
import _synth  # noqa: E402

tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RTP_"))
print(f"# [{tag}]")
QUICK = "--quick" in sys.argv  # one plan per precision (the GPU test of the experiment variants)
for prec_name, prec in (("fp16", r.PREC_FP16), ("mixed", r.PREC_MIXED)) + ((("f16x3", r.PREC_F16X3),) if "--all" in sys.argv else ()):
    for (W, H, N, B) in ((656, 368, 1, 2),) + (() if QUICK else ((656, 368, 3, 1),)) + (((160, 96, 1, 1),) if "--all" in sys.argv else ()):
        e = r.Engine(r.Config(net_w=W, net_h=H, num_scales=N, scale_gap=0.15, precision=prec, frames_in_flight=2 * B, batch_frames=B))
        x = _synth.random_frame(N, H, W, seed=11)
        d = e.forward_debug(x)
        h = hashlib.sha256(d["lowres"].tobytes()).hexdigest()[:16]
        extra = []
        for name in ("pool2_stage1", "conv4_4_CPM", "Mconv5_stage4_L1"):
            try:
                extra.append(hashlib.sha256(e.get_blob(name).tobytes()).hexdigest()[:8])
            except Exception as ex:  # noqa: BLE001
                extra.append("-")
        outs = []
        for f in range(2 * B):  # pipelined path (batched launches)
            e.submit(_synth.random_frame(N, H, W, seed=20 + f), tag=f)
        for f in range(2 * B):
            t, n, j = e.collect()
            outs.append(hashlib.sha256(j.tobytes()).hexdigest()[:8] + f":{n}")
        print(f"{prec_name:6s} {W}x{H} N={N} B={B} lowres {h} nan {int(np.isnan(d['lowres']).sum())} blobs {' '.join(extra)} frames {' '.join(outs)}")
        e.close()
