#!/usr/bin/env python
"""Generate tests/golden/linevec_layers.json from the reference's deploy prototxts.

Run in the build container only (needs /root/reference); the GPU box and the tests read the
committed JSON.  The fixture is the layer TABLE (name, type, bottoms, tops, conv/nms/resize params)
of model/{coco,mpi}/pose_deploy_linevec.prototxt — the graph the built-in generator must equal.
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "linevec_layers.json")


def parse(path):
    t = open(path).read()
    t = re.sub(r"#.*", "", t)
    layers = []
    for L in re.split(r"\nlayer \{", t)[1:]:
        d = {"name": re.search(r'name: "(.*?)"', L).group(1), "type": re.search(r'type: "(.*?)"', L).group(1),
             "bottoms": re.findall(r'bottom: "(.*?)"', L), "tops": re.findall(r'top: "(.*?)"', L)}
        for key, rx in (("num_output", r"num_output: (\d+)"), ("kernel", r"kernel_size: (\d+)"), ("pad", r"\bpad: (\d+)"),
                        ("stride", r"stride: (\d+)"), ("max_peaks", r"max_peaks: (\d+)"), ("num_parts", r"num_parts: (\d+)"),
                        ("factor", r"factor: (\d+)")):
            m = re.search(rx, L)
            if m:
                d[key] = int(m.group(1))
        m = re.search(r"threshold: ([0-9.]+)", L)
        if m:
            d["threshold"] = float(m.group(1))
        m = re.search(r"pool: (\w+)", L)
        if m:
            d["pool"] = m.group(1)
        layers.append(d)
    return {"inputs": re.findall(r'\ninput: "(.*?)"', "\n" + t), "layers": layers}


if not os.path.isdir(REF):
    sys.exit("no /root/reference here")
out = {"source": "model/{coco,mpi}/pose_deploy_linevec.prototxt of CMU-Perceptual-Computing-Lab/caffe_rtpose",
       "coco": parse(f"{REF}/model/coco/pose_deploy_linevec.prototxt"), "mpi": parse(f"{REF}/model/mpi/pose_deploy_linevec.prototxt")}
json.dump(out, open(OUT, "w"), indent=0)
print("wrote", OUT, len(out["coco"]["layers"]), len(out["mpi"]["layers"]))
