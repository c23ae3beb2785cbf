#!/usr/bin/env python
"""Bit-level A/B of the rtp_submit_frame path (host u8 frames -> H2D -> device pre-processing -> conv stack -> post-processing): prints one
SHA-256 over the (tag, people, joints) of N frames pushed through a pipelined engine.  Run once per variant (RTP_* environment variables,
RTP_LIB) and compare: variants that only change WHERE / WHEN copies and kernels are issued must print the same hash."""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
import caffe_rtpose_amd as r  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 7
nframes = int(sys.argv[3]) if len(sys.argv) > 3 else 23          # odd: the stream ends in a partial batch
e = r.Engine(r.Config(net_w=320, net_h=176, precision=r.PREC_MIXED, frames_in_flight=depth, batch_frames=B))
frames = [r.synth_frame(640, 480, i, seed=5) for i in range(nframes)]
h = hashlib.sha256()
sub = col = 0
while col < nframes:
    while sub < nframes and e.in_flight() < depth:
        e.submit_frame(frames[sub], tag=sub)
        sub += 1
    tag, n, j = e.collect()
    assert tag == col
    h.update(np.int64(tag).tobytes() + np.int64(n).tobytes() + j.tobytes())
    col += 1
print(f"frames {nframes} B {B} in_flight {depth}: {h.hexdigest()[:24]}")
e.close()
