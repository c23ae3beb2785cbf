#!/usr/bin/env python
"""COCO keypoint evaluation of rtpose JSON output.

The reference ships only the image list it was evaluated on (`image_info_val2014_1k.txt`:
index, COCO image id, file name, height, width); the scoring is the COCO keypoint protocol.  This
tool turns a directory of `<stem>.json` files (rtpose.bin --image_dir ... --write_json DIR) into COCO
"results" entries and scores them against a person_keypoints annotation file with a restatement of
the published OKS / AP definition (cocodataset.org #keypoints-eval; pycocotools is not in this
image): per image and OKS threshold greedy matching of detections sorted by score, ignoring crowd /
zero-keypoint ground truth, maxDets 20, 101-point interpolated precision, AP = mean over
OKS 0.50:0.05:0.95, plus AP50 / AP75 / AR.

  rtpose.bin --image_dir val2014 --caffemodel pose_iter_440000.caffemodel --caffeproto pose_deploy_linevec.prototxt \
             --write_json out --no_display --no_frame_drops
  python tools/coco_eval.py --json_dir out --image_list image_info_val2014_1k.txt --annotations person_keypoints_val2014.json

Needs real weights and COCO data, neither of which is available offline; tests/test_coco_eval.py
pins the metric on constructed cases."""
import argparse
import json
import os

import numpy as np

# rtpose COCO-18 part order (modelDescriptorFactory.cpp:36-56) -> COCO's 17 keypoints
# nose, l/r eye, l/r ear, l/r shoulder, l/r elbow, l/r wrist, l/r hip, l/r knee, l/r ankle
RTPOSE_TO_COCO = [0, 15, 14, 17, 16, 5, 2, 6, 3, 7, 4, 11, 8, 12, 9, 13, 10]
SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0
OKS_THRS = np.round(np.arange(0.5, 0.95 + 1e-9, 0.05), 2)
REC_THRS = np.linspace(0.0, 1.0, 101)
MAX_DETS = 20


def bodies_to_results(bodies, image_id):
    """`bodies` of one rtpose JSON ({"joints":[x,y,c]*18}) -> COCO result dicts (x,y,v triples, score)."""
    out = []
    for b in bodies:
        j = np.asarray(b["joints"], np.float64).reshape(-1, 3)
        kp = j[RTPOSE_TO_COCO]
        found = kp[:, 2] > 0
        if not found.any():
            continue
        score = float(kp[found, 2].mean() * found.sum() / 17.0)   # mean part confidence weighted by coverage
        flat = []
        for (x, y, c), f in zip(kp, found):
            flat += [float(x), float(y), 1 if f else 0] if f else [0.0, 0.0, 0]
        out.append({"image_id": image_id, "category_id": 1, "keypoints": flat, "score": score})
    return out


def oks(gt, dt):
    """gt: annotation dict (keypoints 51, area, bbox); dt: result dict.  COCOeval.computeOks for one pair."""
    g = np.asarray(gt["keypoints"], np.float64).reshape(17, 3)
    d = np.asarray(dt["keypoints"], np.float64).reshape(17, 3)
    vg = g[:, 2] > 0
    k1 = int(vg.sum())
    var = (SIGMAS * 2) ** 2
    if k1 > 0:
        dx = d[:, 0] - g[:, 0]
        dy = d[:, 1] - g[:, 1]
    else:  # no labelled keypoints: distance to the (doubled) box
        x, y, w, h = gt["bbox"]
        x0, x1, y0, y1 = x - w, x + w * 2, y - h, y + h * 2
        z = np.zeros(17)
        dx = np.max([z, x0 - d[:, 0]], axis=0) + np.max([z, d[:, 0] - x1], axis=0)
        dy = np.max([z, y0 - d[:, 1]], axis=0) + np.max([z, d[:, 1] - y1], axis=0)
    e = (dx ** 2 + dy ** 2) / var / (gt["area"] + np.spacing(1)) / 2
    if k1 > 0:
        e = e[vg]
    return float(np.sum(np.exp(-e)) / e.shape[0])


def evaluate(gts_by_image, dts_by_image, image_ids):
    """Returns dict(AP, AP50, AP75, AR).  gts: COCO annotation dicts, dts: result dicts."""
    T = len(OKS_THRS)
    scores, matched, ignored = [], [[] for _ in range(T)], [[] for _ in range(T)]
    npos = 0
    for img in image_ids:
        gts = list(gts_by_image.get(img, []))
        dts = sorted(dts_by_image.get(img, []), key=lambda d: -d["score"])[:MAX_DETS]
        gig = np.array([bool(g.get("iscrowd", 0)) or g.get("num_keypoints", 1) == 0 for g in gts], bool)
        order = np.argsort(gig, kind="mergesort")           # non-ignored ground truth first
        gts = [gts[i] for i in order]
        gig = gig[order]
        npos += int((~gig).sum())
        ious = np.array([[oks(g, d) for g in gts] for d in dts]).reshape(len(dts), len(gts))
        for t, thr in enumerate(OKS_THRS):
            gtm = -np.ones(len(gts), int)
            for di in range(len(dts)):
                best, m = min(thr, 1 - 1e-10), -1
                for gi in range(len(gts)):
                    if gtm[gi] >= 0 and not gig[gi]:
                        continue
                    if m > -1 and not gig[m] and gig[gi]:
                        break                                   # already matched a real one; the rest are ignore regions
                    if ious[di, gi] < best:
                        continue
                    best, m = ious[di, gi], gi
                if m >= 0:
                    gtm[m] = di
                matched[t].append(m >= 0)
                ignored[t].append(m >= 0 and bool(gig[m]))
        scores += [d["score"] for d in dts]
    scores = np.array(scores)
    order = np.argsort(-scores, kind="mergesort")
    precision = np.zeros((T, len(REC_THRS)))
    recall = np.zeros(T)
    for t in range(T):
        m = np.array(matched[t], bool)[order]
        ig = np.array(ignored[t], bool)[order]
        tp = np.cumsum(m & ~ig)
        fp = np.cumsum(~m & ~ig)
        if npos == 0:
            continue
        rc = tp / npos
        pr = tp / np.maximum(tp + fp, np.spacing(1))
        recall[t] = rc[-1] if len(rc) else 0.0
        pr = pr.tolist()
        for i in range(len(pr) - 1, 0, -1):                    # monotone envelope
            if pr[i] > pr[i - 1]:
                pr[i - 1] = pr[i]
        inds = np.searchsorted(rc, REC_THRS, side="left")
        for ri, pi in enumerate(inds):
            precision[t, ri] = pr[pi] if pi < len(pr) else 0.0
    return {"AP": float(precision.mean()), "AP50": float(precision[0].mean()), "AP75": float(precision[5].mean()), "AR": float(recall.mean()),
            "num_gt": npos, "num_dt": int(len(scores))}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--json_dir", required=True)
    ap.add_argument("--image_list", required=True, help="image_info_val2014_1k.txt: index, image id, file name, height, width")
    ap.add_argument("--annotations", required=True, help="COCO person_keypoints_*.json")
    ap.add_argument("--results_out", default=None, help="also write the COCO results json here")
    a = ap.parse_args()
    ids = {}
    for line in open(a.image_list):
        f = line.split()
        if len(f) >= 3:
            ids[os.path.splitext(f[2])[0]] = int(f[1])
    ann = json.load(open(a.annotations))
    gts = {}
    for g in ann["annotations"]:
        if g.get("category_id", 1) == 1 and g["image_id"] in set(ids.values()):
            gts.setdefault(g["image_id"], []).append(g)
    dts, results = {}, []
    for stem, iid in ids.items():
        p = os.path.join(a.json_dir, stem + ".json")
        if not os.path.exists(p):
            continue
        r = bodies_to_results(json.load(open(p))["bodies"], iid)
        dts[iid] = r
        results += r
    if a.results_out:
        json.dump(results, open(a.results_out, "w"))
    print(json.dumps(evaluate(gts, dts, sorted(ids.values())), indent=1))


if __name__ == "__main__":
    main()
