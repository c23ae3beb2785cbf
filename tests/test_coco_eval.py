"""tools/coco_eval.py: the OKS / AP restatement on constructed cases (no COCO data offline)."""
import importlib.util
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("coco_eval", os.path.join(ROOT, "tools", "coco_eval.py"))
ce = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ce)


def _person(rs, cx, cy, size=120.0):
    kp = np.zeros((17, 3))
    kp[:, 0] = cx + rs.uniform(-0.4, 0.4, 17) * size
    kp[:, 1] = cy + rs.uniform(-0.5, 0.5, 17) * size
    kp[:, 2] = 2
    return kp


def _gt(kp, image_id, crowd=0):
    x0, y0, x1, y1 = kp[:, 0].min(), kp[:, 1].min(), kp[:, 0].max(), kp[:, 1].max()
    return {"image_id": image_id, "category_id": 1, "keypoints": kp.reshape(-1).tolist(), "num_keypoints": int((kp[:, 2] > 0).sum()),
            "area": float((x1 - x0) * (y1 - y0)), "bbox": [x0, y0, x1 - x0, y1 - y0], "iscrowd": crowd}


def _dt(kp, image_id, score):
    return {"image_id": image_id, "category_id": 1, "keypoints": kp.reshape(-1).tolist(), "score": score}


def test_oks_definition():
    rs = np.random.RandomState(0)
    kp = _person(rs, 200, 200)
    g = _gt(kp, 1)
    assert abs(ce.oks(g, _dt(kp, 1, 1.0)) - 1.0) < 1e-12
    # one keypoint moved by d: its term is exp(-d^2 / (2 * area * (2 sigma)^2)), the rest stay 1
    d = 7.0
    kp2 = kp.copy()
    kp2[9, 0] += d
    want = (16 + np.exp(-d * d / (2 * g["area"] * (2 * ce.SIGMAS[9]) ** 2))) / 17
    assert abs(ce.oks(g, _dt(kp2, 1, 1.0)) - want) < 1e-12
    # unlabelled ground-truth keypoints do not count
    kp3 = kp.copy()
    kp3[3:, 2] = 0
    g3 = _gt(kp3, 1)
    far = kp.copy()
    far[3:, :2] += 500
    assert abs(ce.oks(g3, _dt(far, 1, 1.0)) - 1.0) < 1e-12


def test_ap_perfect_missed_and_false_positive():
    rs = np.random.RandomState(1)
    gts, dts = {}, {}
    for img in range(1, 7):
        people = [_person(rs, 150 + 250 * i, 200) for i in range(2)]
        gts[img] = [_gt(p, img) for p in people]
        dts[img] = [_dt(p, img, 0.9 - 0.1 * i) for i, p in enumerate(people)]
    r = ce.evaluate(gts, dts, sorted(gts))
    assert abs(r["AP"] - 1.0) < 1e-9 and abs(r["AR"] - 1.0) < 1e-9 and r["num_gt"] == 12
    # drop every second detection: recall 0.5, precision stays 1 up to there -> AP = 51/101 exactly
    half = {k: v[:1] for k, v in dts.items()}
    r = ce.evaluate(gts, half, sorted(gts))
    assert abs(r["AR"] - 0.5) < 1e-9 and abs(r["AP"] - 51 / 101) < 1e-9
    # confident false positives ahead of every true positive halve the precision
    fp = {k: [_dt(_person(rs, 900, 900), k, 0.99), _dt(_person(rs, 1500, 900), k, 0.98)] + v for k, v in dts.items()}
    r = ce.evaluate(gts, fp, sorted(gts))
    assert abs(r["AR"] - 1.0) < 1e-9 and 0.3 < r["AP"] < 0.55
    # jitter lowers AP at the strict thresholds first
    jit = {k: [_dt(np.column_stack([np.asarray(d["keypoints"]).reshape(17, 3)[:, :2] + rs.randn(17, 2) * 6, np.ones(17)]), k, d["score"]) for d in v]
           for k, v in dts.items()}
    r = ce.evaluate(gts, jit, sorted(gts))
    assert r["AP50"] > r["AP75"] >= 0 and r["AP"] < 1.0
    # crowd ground truth neither counts as a miss nor turns its matches into false positives
    gts2 = {k: v + [_gt(_person(rs, 900, 600), k, crowd=1)] for k, v in gts.items()}
    dts2 = {k: v + [_dt(np.asarray(gts2[k][-1]["keypoints"]).reshape(17, 3), k, 0.5)] for k, v in dts.items()}
    r = ce.evaluate(gts2, dts2, sorted(gts2))
    assert abs(r["AP"] - 1.0) < 1e-9 and r["num_gt"] == 12


def test_rtpose_json_to_coco_results():
    joints = np.zeros((18, 3))
    for part in range(18):
        joints[part] = (10 + part, 100 + part, 0.5)
    joints[16] = 0      # right ear missing
    res = ce.bodies_to_results([{"joints": joints.reshape(-1).tolist()}, {"joints": [0.0] * 54}], 42)
    assert len(res) == 1 and res[0]["image_id"] == 42
    kp = np.asarray(res[0]["keypoints"]).reshape(17, 3)
    assert kp[0].tolist() == [10.0, 100.0, 1.0]                 # nose = part 0
    assert kp[1].tolist() == [25.0, 115.0, 1.0]                 # left eye = part 15
    assert kp[5].tolist() == [15.0, 105.0, 1.0] and kp[6].tolist() == [12.0, 102.0, 1.0]   # l/r shoulder = parts 5 / 2
    assert kp[4].tolist() == [0.0, 0.0, 0.0]                    # right ear (part 16) missing
    assert abs(res[0]["score"] - 0.5 * 16 / 17) < 1e-12
