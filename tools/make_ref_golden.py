#!/usr/bin/env python
"""tests/golden/ref_pin.npz: outputs of the REFERENCE'S OWN code (oracle/_ref/libref.so, built by
oracle/ref_recipe/build_ref.sh from /root/reference) on the seeded cases of tests/_pincases.py.
Run in this container (needs /root/reference); the fixture travels to the GPU box, the reference does not.
Big arrays (the 55 MB resized map) are stored as sha256 + a strided sample; peaks / joints / JSON in full."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pincases as pc  # noqa: E402
import _ref  # noqa: E402

assert _ref.available(), "build oracle/_ref first: bash oracle/ref_recipe/build_ref.sh"
out = {}
tables = {m: _ref.model_tables(m) for m in (0, 1)}
for m in (0, 1):
    out[f"tables_{m}"] = np.array([tables[m][0], tables[m][1]] + tables[m][2] + tables[m][3], np.int32)
for name, (model, low, W, H, start, gap) in pc.lowres_cases(tables).items():
    res, peaks, n, joints = pc.chain(_ref, model, low, W, H, start, gap, pc.disp_of(name))
    out[f"chain_{name}_resized_sha"] = np.frombuffer(pc.digest(res).encode(), np.uint8)
    out[f"chain_{name}_resized_sample"] = res.reshape(-1)[::997].copy()
    out[f"chain_{name}_peaks"] = peaks
    out[f"chain_{name}_joints"] = joints
    out[f"chain_{name}_count"] = np.array([n], np.int32)     # -1: the reference CHECK-fails in connect (portrait net, _pincases.chain)
    print(f"{name}: {n} people, peak totals max {int(peaks[:, 0, 0].max())}")
for nm, (res, peaks) in (("ties", pc.tie_case()), ("single", pc.single_sided_case())):
    n, joints = _ref.connect(0, res, peaks, 64, 656, 368, 1280, 720, pc.THR[0])
    out[f"connect_{nm}_joints"] = joints[:n].copy()
    print(f"connect {nm}: {n} people")
model, low, W, H, start, gap = pc.lowres_cases(tables)["coco_people5"]
out["nms_stale_peaks"] = _ref.nms(_ref.imresize(low, W, H, start, gap)[0], 18, 64, 0.05, pc.stale_peaks(18, 64))
with tempfile.TemporaryDirectory() as d:
    for i, (model, n, joints, scale) in enumerate(pc.json_cases()):
        out[f"json_{i}"] = np.frombuffer(_ref.write_json(d, joints, n, model, float(scale), frame_number=i)[1], np.uint8)
for i, (img, tw, th, normalize) in enumerate(pc.pad_cases()):
    out[f"pad_{i}_sha"] = np.frombuffer(pc.digest(_ref.process_and_pad_image(img, tw, th, normalize)).encode(), np.uint8)
# renderer (renderFunctions.cu on the host): the u8 frames in full (8x8-block display images compress well)
for name, model, img, joints, n, googly in pc.render_cases():
    out[f"render_{name}"] = _ref.render(model, img, joints, n, 656, 368, part_to_show=0, googly=googly)
for name, model, img, maps, parts in pc.view_cases(tables):
    for part in parts:
        out[f"view_{name}_{part}"] = _ref.render(model, img, np.zeros((1, pc.DIMS[model][0], 3), np.float32), 0, maps.shape[2], maps.shape[1],
                                                 part_to_show=part, heatmaps=maps)
# convolution / pooling: caffe_conv (the naive loop of the reference's own tests), im2col_cpu + GEMM, the MAX loop of PoolingLayer::Forward_cpu
for name, x, w, b, pad, stride in pc.conv_cases():
    out[f"conv_{name}_naive"] = _ref.caffe_conv(x, w, b, pad, stride)
    out[f"conv_{name}_im2col"] = _ref.im2col_conv(x, w, b, pad, stride)
for name, x, k, stride, pad in pc.pool_cases():
    out[f"pool_{name}"] = _ref.maxpool(x, k, stride, pad)
path = os.path.join(ROOT, "tests", "golden", "ref_pin.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {os.path.getsize(path)} bytes, {len(out)} arrays")
