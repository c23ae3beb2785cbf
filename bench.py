#!/usr/bin/env python
"""bench.py — frames/sec of the rtpose hot path on MI355X (BASELINE.json metric).

A "step" = one frame of BASELINE.json configs[1] AS WRITTEN ("COCO model 656x368, 1 scale, 1xMI355X, synthetic 720p video"): a
pageable host u8 1280x720 frame -> rtp_submit_frame (pinned copy, H2D, display-fit cubic warp, INTER_AREA pyramid, normalise, pad on
the device: what the producer + processFrame do per frame, rtpose.cpp:322-368, 1127-1133) -> conv stack -> ImResize -> Nms ->
connectLimbsCOCO -> joints on the host.  PCIe and pre-processing are INSIDE the timed region.  The same engine fed net inputs that
are already resident in HBM (rtp_submit_device) is the sub-result `resident_input`.

Multi-GPU: one process per GPU, frames sharded (full replicas, no data-path collective — the reference's --num_gpu dispatcher,
rtpose.cpp:1463-1472) => weak scaling.  `python bench.py --gpus N` started bare re-executes itself under torch.distributed.run with
N ranks; under a launcher it reads RANK/LOCAL_RANK/WORLD_SIZE.  The process group (barrier + MAX-reduce around the timed region,
the agreement on the step count, the gather of the per-rank rates) is RCCL ("nccl") and falls back to gloo if RCCL cannot start
next to the engine's HIP runtime; `comm_backend` in the line says which.  torch touches NO device memory of the data path: frames,
arenas and streams belong to the engine library (rtp_device_alloc).

The timed region is never shorter than --min_seconds (default 2 s) and runs un-instrumented: `--steps K` is a MINIMUM, the region is
repeated with more frames until it is long enough, and the line reports the frames actually timed as `steps` (= `steps_timed`; the flag
as `steps_requested`) with `timed_region_s`.  Defaults per workload are measured optima (profiles/r03_in_flight.txt): COCO 1 scale 7
frames in flight in batches of 2, several scales 3 in flight one frame per launch sequence, MPI 10 in flight in batches of 5.

`roofline` (separate pass right after the timed region): HIP event pairs around EVERY launch of a batch, on the stream the launch runs on,
one batch at a time — the dominant kernel shape's launches give achieved / frac (roofline_block() below), all launches grouped by kernel class
give `roofline.classes` (kernel_classes()); `gpu_busy`: an unprofiled account of when the engine had work on the GPU (rtp_busy_probe,
busy_account()); `parity`: the engine's joints against the full fp32 oracle chain as sets of people, every structural difference traced to
the decision that flipped (tests/_parity.py, tests/_explain.py) AND reproduced by a counterfactual replay of the reference chain with only
those near-tie decisions forced (tests/_replay.py: `replay_identical`), plus the structured leg: planted people + the engine's measured
deviation field through both post-processing chains; `cpu_baseline`: the
OpenMP oracle port and torch-CPU conv2d over the same layers; `latency_ms.p50_single_frame`: one frame alone, commit -> joints on the
host (rtpose.cpp:1430).

The printed line is COMPACT (compact_line(): <= 8 KB, one copy of the per-class rows); every leg's body — `sub_results` (resident input,
MPI 496x368 = BASELINE configs[4] with its own roofline and parity, 3 scales = configs[2], single-pass fp16, exact-f32, post-processing alone
on analytic heat maps), the parity explanations, the notes — goes to `bench_detail.json` next to this file (and under gpurun_out/ on the GPU
box); the line's `summary` holds each leg's headline scalar.
"""
import argparse
import glob
import json
import math
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


PREC_NOTE = {
    "mixed": "fp16 MFMA, fp32 accumulate; conv2_*..conv4_*, refinement stages 4-6 and all 1x1 layers also multiply the fp16 rounding errors of both operands "
             "(a_hi*W_hi + a_lo*W_hi + a_hi*W_lo; the two corrections as MX-scaled fp8 MFMA chunks on the 3x3/7x7 layers, as fp16 passes on the 1x1 layers): "
             "final maps within 1e-3 of the fp32 reference (normalised to max 1), tests/test_precision.py",
    "fp16": "fp16 storage and MFMA everywhere, fp32 accumulate: final maps 2.0-2.6e-3 from the fp32 reference, OUTSIDE the +-1e-3 north-star tolerance",
    "f16x3": "every layer as hi+lo fp16 pairs in three MFMA passes: fp32-class accuracy",
    "fp32": "fp32 storage, exact-f32 MFMA (the reference's arithmetic)",
}
MODELS = {  # name -> (model id, net_w, net_h, parts, max_peaks, nms threshold, conv GFLOP per scale-image)
    "coco": (0, 656, 368, 18, 64, 0.05, 484.634),
    "mpi": (1, 496, 368, 15, 20, 0.2, 361.695),
}


def oracle_frames(eng, num_scales, model, frames, seed0=1):
    """`frames` + 1 synthetic frames through the CPU oracle's fp32 conv stack with the engine's weights (frame 0 = warm-up).
    Returns (net, [(x, lowres, conv seconds)])."""
    import _oracle as orc
    import _synth
    mid, W, H, _, _, _, _ = MODELS[model]
    net = orc.Net(mid)
    for i in range(len(net.convs)):
        w, b = eng.get_conv_weights(i)
        net.set_weights(i, w, b)
    out = []
    for f in range(frames + 1):
        x = _synth.random_frame(num_scales, H, W, seed=seed0 + f)
        t0 = time.time()
        low = net.forward(x)
        out.append((x, low, time.time() - t0))
    return net, out


def torch_cpu_stack(eng, model):
    """The same 92 convolutions (+ ReLU, max-pool, concat) as torch-CPU conv2d (oneDNN, every host core): SURVEY 8d's second,
    faster CPU baseline.  Returns forward(x_numpy) -> lowres numpy [N][C][h][w] in reference channel order."""
    import torch
    import torch.nn.functional as F
    Wt = {}
    for i, (name, _, _, _) in enumerate(eng.conv_layers()):
        w, b = eng.get_conv_weights(i)
        Wt[name] = (torch.from_numpy(w.copy()), torch.from_numpy(b.copy()))

    def conv(name, t, relu=True):
        w, b = Wt[name]
        y = F.conv2d(t, w, b, padding=w.shape[-1] // 2)
        return F.relu_(y) if relu else y

    trunk = ["conv1_1", "conv1_2", "P", "conv2_1", "conv2_2", "P", "conv3_1", "conv3_2", "conv3_3", "conv3_4", "P", "conv4_1", "conv4_2",
             "conv4_3_CPM", "conv4_4_CPM"]
    nstage = max(int(n.split("_stage")[1].split("_")[0]) for n in Wt if n.startswith("Mconv"))

    def forward(x):
        with torch.no_grad():
            t = torch.from_numpy(x)
            for nm in trunk:
                t = F.max_pool2d(t, 2, 2, ceil_mode=True) if nm == "P" else conv(nm, t)
            feat = t
            br = {}
            for L in (1, 2):
                t = feat
                for k in range(1, 6):
                    t = conv(f"conv5_{k}_CPM_L{L}", t, relu=k < 5)
                br[L] = t
            for st in range(2, nstage + 1):
                cat = torch.cat([br[1], br[2], feat], 1)
                nb = {}
                for L in (1, 2):
                    t = cat
                    for k in range(1, 8):
                        t = conv(f"Mconv{k}_stage{st}_L{L}", t, relu=k < 7)
                    nb[L] = t
                br = nb
            return torch.cat([br[2], br[1]], 1).numpy()   # concat_stage7: heat maps first, PAFs second
    return forward


def cpu_baseline(eng, num_scales, model="coco", frames=3, oracle=None):
    """The CPU oracle (a port: the reference's conv stack cannot be built here) timed on this host's cores on a
    bounded sample: 1 warm-up frame, then `frames` frames through conv stack + ImResize + NMS + connect; beside it torch-CPU
    conv2d over the same layers (a faster second baseline)."""
    import numpy as np
    import torch
    import _oracle as orc
    mid, W, H, parts, max_peaks, thr, gflop = MODELS[model]
    _, fr = oracle if oracle is not None else oracle_frames(eng, num_scales, model, frames)
    frames = len(fr) - 1
    tc = tp = 0.0
    for f, (x, low, t_conv) in enumerate(fr):
        t1 = time.time()
        res = orc.imresize(low, W, H, 1.0, 0.3)[0]
        peaks = orc.nms(res, parts, max_peaks, thr)
        orc.connect(mid, res, peaks, max_peaks, W, H, 1280, 720)
        t2 = time.time()
        if f:  # frame 0 = warm-up (page-in, thread pool)
            tc += t_conv
            tp += t2 - t1
    out = {"value": frames / (tc + tp), "unit": "frames/s", "cores": orc.num_threads(), "kind": "port",
           "sample": f"{frames} frames after 1 warm-up, {num_scales} scale(s), {W}x{H} {model.upper()}: conv stack {tc / frames:.2f} s + "
                     f"post-processing {tp / frames:.3f} s per frame, OpenMP fp32 (oracle/rtpose_oracle.cpp)",
           "conv_stack_gflops": gflop * num_scales * frames / tc}
    try:
        fwd = torch_cpu_stack(eng, model)
        tt = 0.0
        dev = 0.0
        for f, (x, low, _) in enumerate(fr):
            t0 = time.time()
            y = fwd(x)
            if f:
                tt += time.time() - t0
            dev = max(dev, float(np.abs(y - low).max() / np.abs(low).max()))
        out["torch_conv"] = {"conv_stack_fps": frames / tt, "conv_stack_gflops": gflop * num_scales * frames / tt, "threads": torch.get_num_threads(),
                             "max_rel_dev_from_oracle": dev,
                             "what": "torch.nn.functional.conv2d / relu / max_pool2d / cat on the CPU (oneDNN, fp32), conv stack only, same frames and weights"}
        out["torch_conv_fps"] = frames / (tt + tp)   # torch conv stack + the oracle's post-processing
    except Exception as ex:  # noqa: BLE001
        out["torch_conv"] = {"error": str(ex)}
    return out


def parity_report(eng, fr, model, num_scales, scale_gap, structured=True, start_scale=1.0):
    """SURVEY section 7 / BASELINE.md section 3: the engine's joints (this precision mode, through rtp_submit / rtp_collect) against the
    full fp32 oracle chain conv -> ImResize -> Nms -> connectLimbs* on the same frames, as SETS of people (tests/_parity.py).
    Units: the synthetic network's maps have a maximum of ~5 where real confidences live in [0, 1]; scores and map errors are
    divided by max|reference map| (= stated for maps normalised to a maximum of 1, the unit of the +-1e-3 tolerance), positions
    are display pixels.  (Scaling the maps themselves into [0, 1], as tests/test_precision.py does for the peak test, leaves the
    noise network without a single person above connectLimbs' thresholds: nothing to compare.)

    Every joint that is the same peak on both sides must be inside +-1 px / +-1e-3 (`numeric_out_of_tol` = 0), and every
    STRUCTURAL difference (the random-weight network's maps are noise: hundreds of maxima, some of them near-ties) is traced by
    tests/_explain.py to the decision that flipped — an NMS compare, a PAF sample against its threshold, a sample coordinate at a
    rounding boundary, two candidates swapping places in the greedy order — whose margin on the REFERENCE side must be below twice
    the deviation measured between the two sides: `structural_explained` == `joints_structural`, anything else is a FAIL.
    `structured` adds the leg on maps that look like poses (structured_parity below)."""
    import numpy as np
    import _oracle as orc
    import _parity
    import _explain
    mid, W, H, parts, max_peaks, _, _ = MODELS[model]
    th = eng.get_thresholds()
    reps, exps, map_err, post_exact, dev = [], [], 0.0, True, None
    for x, ref, _ in fr:
        norm = float(np.abs(ref).max())
        res = orc.imresize(ref, W, H, start_scale, scale_gap)[0]
        peaks = orc.nms(res, parts, max_peaks, th["nms_threshold"])
        nr, jr = orc.connect(mid, res, peaks, max_peaks, W, H, 1280, 720, th)
        eng.submit(x, tag=1)
        eng.flush()
        _, ne, je = eng.collect()
        rep = _parity.people_parity(je[:ne], jr[:nr], tol_px=1.0, tol_c=1e-3, c_norm=norm)
        got = eng.forward_heatmaps(x)
        map_err = max(map_err, float(np.abs(got - ref).max() / norm))
        if dev is None:
            dev = (got - ref) / norm       # the conv stack's measured deviation field, in units of the map maximum
        # decomposition: the reference's post-processing applied to the ENGINE's maps must give the engine's joints bit for bit;
        # whatever differs between the two people sets is then decided by strict compares on maps that differ by <= map_err
        res_e = orc.imresize(got, W, H, start_scale, scale_gap)[0]
        n2, j2 = orc.connect(mid, res_e, orc.nms(res_e, parts, max_peaks, th["nms_threshold"]), max_peaks, W, H, 1280, 720, th)
        post_exact = post_exact and n2 == ne and np.array_equal(j2[:n2], je[:ne])
        ex1 = _explain.explain(mid, res, res_e, max_peaks, W, H, 1280, 720, th, rep["structural"], tol_px=1.0, tol_c=1e-3, c_norm=norm, out_of_tol=rep["out_of_tol"])
        _parity.reclassify(rep, ex1["out_of_tol_is_flip"])   # two neighbouring pixels swapping the role of the maximum = an NMS flip, not a numeric deviation
        exps.append(ex1)
        for k in ("structural", "out_of_tol", "_in_tol_max"):
            rep.pop(k, None)
        reps.append(rep)
    tot = _parity.merge(reps)
    ex = _explain.merge(exps)
    tot["map_max_err"] = map_err
    tot["post_on_engine_maps_bit_exact"] = bool(post_exact)
    tot["structural_explained"] = ex["structural_explained"]
    tot["explain"] = {k: ex[k] for k in ("root_flips", "unexplained", "unexplained_detail", "attribution", "e_map", "e_pos_net_px", "worst_margin_over_allowance",
                                         "replay_forced", "replay_unexplained", "replay_unexplained_detail", "replay_worst_margin_over_allowance")}
    # the per-joint proof (tests/_replay.py): the reference side replayed with the engine's outcome forced at every decision that differs for the
    # same inputs — each a checked near-tie — gives the engine's people exactly
    tot["replay_identical"] = bool(ex["replay_identical"] and ex["replay_unexplained"] == 0)
    tot["units"] = "x, y in display pixels (1280x720); scores and map errors for maps normalised to a maximum of 1; explain.e_map in raw map units"
    tot["reference"] = "CPU oracle, fp32 conv stack -> ImResize -> Nms -> connectLimbs*, same synthetic weights and frames"
    tot["verdict"] = _parity.verdict(tot, map_err=map_err, post_exact=post_exact, explained=ex["structural_explained"] if ex["unexplained"] == 0 else 0)
    if structured and dev is not None:
        try:
            tot["structured"] = structured_parity(eng, model, dev, scale_gap, start_scale)
        except Exception as ex2:  # noqa: BLE001
            tot["structured"] = {"verdict": f"FAIL: {ex2}"}
    return tot


def exact_modes_table(make_engine, device_frames, measure, fr, model, num_scales, scale_gap, batch_frames, in_flight, mixed_row, modes=("f16x3", "fp32")):
    """what an exact mode costs and buys in PEOPLE terms.  The SAME noise frames `fr` (oracle_frames()[1][1:]) through an
    engine of each exact mode (RTP_PREC_F16X3 = every layer as three fp16 passes; RTP_PREC_FP32 = the reference's arithmetic) and through the
    fp32 oracle chain: people / joints matched inside +-1 px / +-1e-3, structural differences, map error — next to the mode's pipelined
    frames/s (resident inputs, the headline's batching).  `mixed_row` = the default mode's figures for the same frames."""
    import numpy as np
    import _oracle as orc
    import _parity
    mid, W, H, parts, max_peaks, _, _ = MODELS[model]
    rows = {"mixed": mixed_row}
    for mode in modes:
        try:
            e = make_engine(mode, num_scales, scale_gap, batch_frames, in_flight)
            th = e.get_thresholds()
            reps, map_err = [], 0.0
            for x, ref, _ in fr:
                norm = float(np.abs(ref).max())
                res = orc.imresize(ref, W, H, 1.0, scale_gap)[0]
                nr, jr = orc.connect(mid, res, orc.nms(res, parts, max_peaks, th["nms_threshold"]), max_peaks, W, H, 1280, 720, th)
                e.submit(x, tag=1)
                e.flush()
                _, ne, je = e.collect()
                rep = _parity.people_parity(je[:ne], jr[:nr], tol_px=1.0, tol_c=1e-3, c_norm=norm)
                for k in ("structural", "out_of_tol", "_in_tol_max"):
                    rep.pop(k, None)
                reps.append(rep)
                map_err = max(map_err, float(np.abs(e.forward_heatmaps(x) - ref).max() / norm))
            tot = _parity.merge(reps)
            f1 = device_frames(e)
            m = measure(e, lambda i, tag: e.submit_device(f1[i % len(f1)], tag=tag), steps=50, warmup=10, in_flight=in_flight, min_seconds=1.0)
            rows[mode] = {"fps_resident": round(m["fps"], 1), "people_ref": tot["people_ref"], "people_engine": tot["people_engine"], "people_matched": tot["people_matched"],
                          "joints_ref": tot["joints_ref"], "joints_matched": tot["joints_matched"], "joints_structural": tot["joints_structural"],
                          "numeric_out_of_tol": tot["numeric_out_of_tol"], "map_max_err": map_err}
            e.close()
        except Exception as ex:  # noqa: BLE001
            rows[mode] = {"error": str(ex)}
    return rows


def structured_parity(eng, model, dev, scale_gap, start_scale=1.0):
    """Conv -> JSON parity on maps that LOOK like pose maps.  No trained weights exist offline, so the maps are
    planted: P = 1 / 5 / 20 stick figures as analytic low-res heat maps + PAFs (tests/_synth.people_lowres, values in [0, 1]).  The
    reference side runs them through the fp32 oracle chain; the engine side gets the same maps PLUS the conv stack's MEASURED deviation
    field — (engine - oracle) low-res maps of a noise frame in this precision mode, in units of that frame's map maximum, scaled into the
    planted maps' range — and runs the production post-processing on the device (rtp_post_from_lowres).

    Required: the same people, every joint that is the same maximum inside +-1 px / +-1e-3 — and NO structural difference except the one
    a smooth peak cannot avoid: its two centre pixels differ by less than the tolerance when the peak's centre lies within ~0.1 pixel of
    the midpoint between them, the maximum then moves to the NEXT pixel and the joint with it (the 7x7 centroid follows the integer
    maximum).  Each such flip must be traced by tests/_explain.py to an NMS compare with a sub-tolerance margin, and with the
    tolerance set to one NET pixel (1.95 display pixels) the people sets must be IDENTICAL: nobody added, lost or re-routed."""
    import numpy as np
    import _oracle as orc
    import _parity
    import _explain
    import _synth
    mid, W, H, parts, max_peaks, _, _ = MODELS[model]
    th = eng.get_thresholds()
    tables = orc.model_tables(mid)
    net_px = max(1280 / W, 720 / H) * 1.06    # one net pixel in display pixels (+ the little the centroid adds)
    out = {"cases": {}, "deviation_max": float(np.abs(dev).max()),
           "what": "planted people + the engine's measured conv deviation -> device post-processing, vs planted people -> fp32 oracle chain"}
    ok, flips_total = True, 0
    for P in (1, 5, 20):
        if start_scale == 1.0:
            low, _people = _synth.people_lowres(mid, tables, P, eng.low_h, eng.low_w, seed=3 + P, N=eng.N)
        else:   # --start_scale != 1: every scale's copy of the people lives in that scale's crop window (imresize_layer.cu:110-113)
            import _pincases
            low = _pincases.scaled_people(mid, tables, P, eng.low_h, eng.low_w, 3 + P, eng.N, start_scale, scale_gap)
        s = float(np.abs(low).max())
        low_e = np.ascontiguousarray(low + dev * np.float32(s), np.float32)
        res = orc.imresize(low, W, H, start_scale, scale_gap)[0]
        nr, jr = orc.connect(mid, res, orc.nms(res, parts, max_peaks, th["nms_threshold"]), max_peaks, W, H, 1280, 720, th)
        _, je, ne = eng.post_from_lowres(low_e)
        rep = _parity.people_parity(je[:ne], jr[:nr], tol_px=1.0, tol_c=1e-3, c_norm=s)
        explained = None
        if rep["joints_structural"] or rep["numeric_out_of_tol"]:
            ex = _explain.explain(mid, res, orc.imresize(low_e, W, H, start_scale, scale_gap)[0], max_peaks, W, H, 1280, 720, th, rep["structural"],
                                  tol_px=1.0, tol_c=1e-3, c_norm=s, out_of_tol=rep["out_of_tol"])
            _parity.reclassify(rep, ex["out_of_tol_is_flip"])
            explained = ex["structural_explained"] if ex["unexplained"] == 0 else 0
        v = _parity.verdict(_parity.merge([rep]), explained=explained)
        wide = _parity.people_parity(je[:ne], jr[:nr], tol_px=net_px, tol_c=1e-3, c_norm=s, pair_px=max(3.0, net_px))
        same_people = wide["people_matched"] == nr == ne and wide["joints_structural"] == 0 and wide["numeric_out_of_tol"] == 0
        flips_total += rep["joints_structural"]
        out["cases"][f"P{P}"] = {"people_ref": nr, "people_engine": ne, "people_matched": rep["people_matched"], "joints_ref": rep["joints_ref"],
                                 "joints_matched": rep["joints_matched"], "numeric_out_of_tol": rep["numeric_out_of_tol"],
                                 "adjacent_pixel_flips": rep["joints_structural"], "flips_explained": explained,
                                 "identical_within_one_net_pixel": bool(same_people),
                                 "max_dx_px": rep["max_dx_px"], "max_dy_px": rep["max_dy_px"], "max_dc": rep["max_dc"], "verdict": v}
        ok = ok and nr >= 1 and same_people and not v.startswith("FAIL")
    out["adjacent_pixel_flips"] = flips_total
    out["verdict"] = ("pass" if flips_total == 0 else f"pass: same people, every joint that is the same maximum inside +-1 px / +-1e-3; {flips_total} joint(s) moved to the "
                      "neighbouring net pixel (two centre pixels of a smooth peak within the tolerance of each other, each traced to its NMS compare)") if ok else \
        "FAIL: " + "; ".join(f"{k}: {c['verdict']}{'' if c['identical_within_one_net_pixel'] else ' / people differ beyond one net pixel'}" for k, c in out["cases"].items()
                             if c["verdict"].startswith("FAIL") or not c["identical_within_one_net_pixel"])
    return out


def pmc_traffic(precision, batch_frames, num_scales, model, suffix=""):
    """HBM bytes per dominant launch from the newest rocprofv3 PMC summary under profiles/ whose header names this
    configuration (tools/collect_profiles.sh writes them: one counter per pass).  FETCH_SIZE/WRITE_SIZE are KiB;
    FETCH_SIZE x2 per the guide's gfx950 correction.  None when no matching profile is committed."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dominant_conv_pmc*.txt")), reverse=True):
        if path.endswith("_2q.txt") != (suffix == "_2q"):
            continue
        try:
            lines = open(path).read().splitlines()
        except OSError:
            continue
        m = re.search(r"prof_dominant\.py\s+(\S+)\s+\d+(?:\s+(\d+))?(?:\s+(\d+))?(?:\s+(coco|mpi)\b)?", lines[0] if lines else "")
        if not m:
            continue
        p_prec, p_b, p_n, p_model = m.group(1), int(m.group(2) or 1), int(m.group(3) or 1), (m.group(4) or "coco")
        if (p_prec, p_b, p_n, p_model) != (precision, batch_frames, num_scales, model):
            continue
        vals = {}
        for ln in lines[1:]:
            mm = re.match(r"(\S+)\s+launches=\s*(\d+)\s+mean=(\S+)\s+(\S+)", ln)
            if mm and "Li7E" in mm.group(4) and mm.group(1) in ("FETCH_SIZE", "WRITE_SIZE"):  # the 7x7 kernels; most launches = 128->128 pairs
                cur = vals.get(mm.group(1))
                if cur is None or int(mm.group(2)) > cur[0]:
                    vals[mm.group(1)] = (int(mm.group(2)), float(mm.group(3)))
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            best = {"bytes": (2 * vals["FETCH_SIZE"][1] + vals["WRITE_SIZE"][1]) * 1024, "source": os.path.relpath(path, ROOT)}
            break
    return best


def roofline_block(dom_ms, dom_n, dom_flops, byp, solo_ms, peak, tr=None, tr2=None):
    """The `roofline` object of the bench line from the event timings of the dominant launches (total ms, launches, FLOPs per launch,
    {passes: (ms, launches)}), the same kernel timed alone (solo_ms per launch) and the PMC traffic records.  A per-launch figure that
    is implausible against the solo timing is REPLACED by it, with the reason in `how`; `frac` never leaves (0, 1]."""
    ms = dom_ms / max(dom_n, 1)
    how = "HIP event pair around every launch of this kernel shape, on the launch's stream, over 40 batches processed one at a time (whole frames, no other frame's kernels on the chip)"
    byp = dict(byp or {})
    max_passes = max(byp) if byp else 1
    if not (dom_n > 0 and solo_ms > 0 and 0.5 * solo_ms <= ms <= 10.0 * max_passes * solo_ms):   # implausible against the same kernel timed alone: say so, never print a fantasy
        how = f"FALLBACK to the solo timing: the per-launch events gave {ms:.6g} ms over {dom_n} launches, outside [0.5x, {10 * max_passes}x] of the solo launch ({solo_ms:.4f} ms)"
        ms, byp, dom_ms, dom_n = solo_ms, {}, solo_ms, 1
    achieved = dom_flops / (ms * 1e-3) if ms > 0 else 0.0
    roof = {"bound": "mfma", "kernel": "conv_ring_kernel 7x7 128->128 (L1+L2 branch pair)", "achieved": achieved / 1e12, "peak": peak / 1e12,
            "unit": "TFLOP/s", "frac": min(max(achieved / peak, 0.0), 1.0), "traffic": tr["bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
            "ms_per_launch": ms, "launches_timed": dom_n, "flops_per_launch": dom_flops, "how": how}
    if tr2:  # the fp8-compensated launches of the same shape read the q blocks and the fp8 weight chunks as well
        roof["traffic_2q"] = tr2["bytes"]
        roof["traffic_2q_source"] = tr2["source"]
    if byp:
        exec_flops = sum(p * n for p, (_, n) in byp.items()) * dom_flops
        roof["by_mfma_passes"] = {str(p): {"launches": n, "ms_per_launch": t / n, "algorithmic_tflops": dom_flops / (t / n * 1e-3) / 1e12,
                                           "executed_mfma_tflops": p * dom_flops / (t / n * 1e-3) / 1e12} for p, (t, n) in byp.items()}
        roof["executed"] = {"mfma_tflops": exec_flops / (dom_ms * 1e-3) / 1e12, "frac_of_peak": exec_flops / (dom_ms * 1e-3) / peak,
                            "note": "matrix-pipe work actually issued, in fp16-pass equivalents (error-compensated launches: 2 or 3 passes per algorithmic flop)"}
    if solo_ms > 0:
        roof["solo"] = {"ms_per_launch": solo_ms, "achieved": dom_flops / (solo_ms * 1e-3) / 1e12, "frac": dom_flops / (solo_ms * 1e-3) / peak,
                        "what": "the plain fp16 instantiation, 200 launches back to back between two events"}
    return roof


def kernel_classes(plan_lines, layers, step_times, net_w, net_h, images_per_launch, peak):
    """`roofline.classes`: the per-step HIP-event timings of the roofline pass (rtp_kernel_timing 3: one pair around EVERY launch of a full
    batch, one batch at a time) grouped by kernel class.  plan_lines: the "step ..." lines of rtp_plan_summary (same order as step_times);
    layers: [(name, cin, cout, k)].  Algorithmic FLOPs per launch = sum over the step's layers of 2 Cout Cin k^2 H W x images per launch
    (SURVEY 8a3's formula; H x W of the layer's resolution level).  Per class: launches timed, average us per launch, ms per batch (the
    class's launches of ONE batch), algorithmic TFLOP/s over the class's time, fraction of the fp16 MFMA peak, share of the batch's summed
    launch time."""
    dims = {n: (cin, cout, k) for n, cin, cout, k in layers}

    def level(name):
        return 0 if name.startswith("conv1_") else 1 if name.startswith("conv2_") else 2 if name.startswith("conv3_") else 3

    rows, order = {}, []
    for ln, (ms, n) in zip(plan_lines, step_times):
        m = re.match(r"step (\w+) (.*?) k (\d+) ", ln)
        if not m or n == 0:
            continue
        kind, k = m.group(1), int(m.group(3))
        names = [t for t in re.findall(r"[A-Za-z0-9_]+", m.group(2)) if t in dims]
        flops = sum(2.0 * dims[t][1] * dims[t][0] * dims[t][2] ** 2 * (net_h >> level(t)) * (net_w >> level(t)) for t in names) * images_per_launch
        q = "2q" if " passes 2q " in ln else "3 passes" if re.search(r" passes 3\w* ", ln) and kind == "conv" else "plain"
        n0 = names[0] if names else ""
        if kind == "first":
            cls = "conv1_1 (conv_first)"
        elif kind == "pw2":
            cls = "branch tails 1x1->1x1 (conv_pw2)"
        elif kind == "conv" and k == 7:
            cls = ("dominant 7x7 128->128 pair " if dims[n0][0] == 128 else f"stage-entry 7x7 {dims[n0][0]}->128 pair ") + q
        elif kind == "conv" and n0.startswith(("conv1_", "conv2_", "conv3_", "conv4_")):
            cls = "trunk 3x3 " + q
        elif kind == "conv":
            cls = f"stage-1 {k}x{k} pair " + q
        else:
            cls = kind
        if cls not in rows:
            rows[cls] = {"steps": 0, "launches": 0, "ms": 0.0, "ms_per_batch": 0.0, "flops": 0.0}
            order.append(cls)
        r_ = rows[cls]
        r_["steps"] += 1
        r_["launches"] += n
        r_["ms"] += ms
        r_["ms_per_batch"] += ms / n
        r_["flops"] += flops * n
    total = sum(r_["ms_per_batch"] for r_ in rows.values()) or 1.0
    out = {}
    for cls in order:
        r_ = rows[cls]
        tf = r_["flops"] / (r_["ms"] * 1e-3) / 1e12 if r_["ms"] > 0 else 0.0
        out[cls] = {"steps_per_batch": r_["steps"], "launches_timed": r_["launches"], "us_per_launch": r_["ms"] / r_["launches"] * 1e3,
                    "ms_per_batch": r_["ms_per_batch"], "tflops": tf, "frac_of_peak": tf * 1e12 / peak, "share_of_batch": r_["ms_per_batch"] / total}
    return out, total


def stamp_dominant(spans, plan_lines, flops_per_launch, peak):
    """The dominant launches seen from the DEVICE: rtp_stamp_probe's {slot, start_us, end_us} triples (slot < 64 = plan step, in the order of
    rtp_plan_summary's "step" lines) of batches processed one at a time -> for the 7x7 128->128 pair steps the mean residency (first
    workgroup's start .. last workgroup's end, one wall clock for all XCDs) per MFMA pass count and over all of them.  No launch latency and
    no event marker inside the span: the quantity a kernel trace reports (profiles/ rocprofv3 stats), measured without a tracer."""
    import numpy as np
    dom = {}
    for i, ln in enumerate(plan_lines):
        if re.match(r"step conv .* k 7 cin_p 128 cout 128 ", ln):
            dom[i] = "2" if " passes 2q " in ln else "3" if re.search(r" passes 3\w* ", ln) else "1"
    acc = {}
    for slot, t0, t1 in np.asarray(spans, np.float64).reshape(-1, 3):
        p_ = dom.get(int(slot))
        if p_ is not None and t1 > t0:
            a = acc.setdefault(p_, [0, 0.0])
            a[0] += 1
            a[1] += t1 - t0
    n = sum(a[0] for a in acc.values())
    if not n:
        return None
    us = sum(a[1] for a in acc.values()) / n
    return {"launches": n, "us_per_launch": us, "us_by_mfma_passes": {p_: a[1] / a[0] for p_, a in sorted(acc.items())},
            "achieved": flops_per_launch / (us * 1e-6) / 1e12, "frac": flops_per_launch / (us * 1e-6) / peak,
            "what": "device-side residency stamps of the same launches (first workgroup start .. last workgroup end), batches one at a time, no tracer"}


def busy_account(spans):
    """rtp_busy_probe's spans -> the share of the wall (first start .. last end, after dropping the first tenth as warm-up) in which the
    engine had work on the GPU: union of every batch's conv-stream span (first input staging .. end of its conv stack) and every frame's
    post-processing chain.  No profiler attached: the events are the ones the per-frame path records anyway."""
    import numpy as np
    if len(spans) < 8:
        return None
    sp = spans[np.argsort(spans[:, 1])]
    sp = sp[len(sp) // 10:]
    lo, hi = float(sp[:, 1].min()), float(sp[:, 2].max())

    def union(a):
        tot, cur_s, cur_e = 0.0, None, None
        for s_, e_ in a:
            if cur_e is None or s_ > cur_e:
                if cur_e is not None:
                    tot += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        return tot + (cur_e - cur_s if cur_e is not None else 0.0)

    allu = union(sp[:, 1:3].tolist())
    conv = sp[sp[:, 0] == 0]
    post = sp[sp[:, 0] == 1]
    wall = max(hi - lo, 1e-9)
    return {"busy_frac": allu / wall, "idle_frac": 1.0 - allu / wall, "conv_streams_busy_frac": union(conv[:, 1:3].tolist()) / wall if len(conv) else None,
            "post_chains_busy_frac": union(post[:, 1:3].tolist()) / wall if len(post) else None,
            "conv_stacks_concurrent_avg": float((conv[:, 2] - conv[:, 1]).sum() / wall) if len(conv) else None,
            "window_ms": wall, "spans": int(len(sp)),
            "what": "UNPROFILED: union of per-stream busy spans (HIP events of the pipelined run: each batch from its first input staging to the end of its conv stack "
                    "on the batch's stream, each frame's post-processing chain + D2H on its own stream) over the wall between the first start and the last end; "
                    "idle_frac = no stream of the engine had anything to run (rocprofv3's timeline of the same loop: profiles/, carries the tracer's overhead)"}


def stamp_account(spans):
    """rtp_stamp_probe's {slot, start_us, end_us} triples (device-side residency stamps: first workgroup start .. last workgroup end of EVERY
    kernel launch of the pipelined run) -> what the chip was doing over the wall between the first start and the last end (first tenth dropped
    as warm-up): `resident_frac` = share of the wall with at least one kernel resident, `idle_frac` = 1 - that (NO kernel on the chip),
    the same for the convolution kernels alone, the average number of kernels resident, and the histogram of that number.  Unprofiled: the
    stamps are two atomics per workgroup; the loop's frame rate with the probe on is reported next to it."""
    import numpy as np
    if spans is None or len(spans) < 16:
        return None
    sp = spans[np.argsort(spans[:, 1])]
    sp = sp[len(sp) // 10:]
    lo, hi = float(sp[:, 1].min()), float(sp[:, 2].max())
    wall = max(hi - lo, 1e-9)
    ev = sorted([(float(a), 1) for a in sp[:, 1]] + [(float(b), -1) for b in sp[:, 2]])
    hist, cur, last, busy = {}, 0, lo, 0.0
    for t, d in ev:
        hist[cur] = hist.get(cur, 0.0) + (t - last)
        if cur > 0:
            busy += t - last
        cur += d
        last = t

    def union(a):
        tot, cs, ce = 0.0, None, None
        for s_, e_ in sorted(a):
            if ce is None or s_ > ce:
                if ce is not None:
                    tot += ce - cs
                cs, ce = s_, e_
            else:
                ce = max(ce, e_)
        return tot + (ce - cs if ce is not None else 0.0)

    conv = sp[sp[:, 0] < 64]
    post = sp[(sp[:, 0] >= 64) & (sp[:, 0] < 200)]
    return {"resident_frac": busy / wall, "idle_frac": 1.0 - busy / wall,
            "conv_resident_frac": union(conv[:, 1:3].tolist()) / wall if len(conv) else None,
            "post_resident_frac": union(post[:, 1:3].tolist()) / wall if len(post) else None,
            "kernels_resident_avg": float((sp[:, 2] - sp[:, 1]).sum() / wall),
            "kernels_resident_hist": {str(k): round(v / wall, 4) for k, v in sorted(hist.items()) if v / wall >= 5e-4},
            "launches": int(len(sp)), "window_ms": wall * 1e-3,
            "what": "device-side stamps (kernels.h KStamp): first workgroup start .. last workgroup end of every kernel launch of the pipelined loop, one clock for all XCDs; "
                    "idle_frac = share of the wall with NO kernel resident on the chip"}


USER_HW_QUEUES = None


def leg_subprocess(flags, timeout=900):
    """One sub-result leg as a bench.py run of its own (fresh HIP runtime: its own hardware-queue count and stream -> queue arrangement, which an
    engine created later in THIS process would not get: the runtime attaches streams to the least-loaded queue, so what an engine sees depends on
    every stream the process has created before).  Returns the leg's detail dict."""
    import tempfile
    fd, path = tempfile.mkstemp(suffix=".json", prefix="rtp_bench_leg_")
    os.close(fd)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if USER_HW_QUEUES is None:
        env.pop("GPU_MAX_HW_QUEUES", None)          # the leg picks the count of ITS batch size
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--no_cpu_baseline", "--no_sub_results", "--steps", "50", "--warmup", "10", "--min_seconds", "1.0",
                            "--detail_out", path] + flags, env=env, capture_output=True, text=True, timeout=timeout)
        if p.returncode != 0:
            return {"error": (p.stderr or p.stdout)[-400:]}
        with open(path) as f:
            return json.load(f)
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


LINE_LIMIT = 8192   # the driver keeps an 8 KB stdout tail: the ONE line must fit it whole


DETAIL_OUT = None   # --detail_out: a leg run by the parent bench.py writes its detail here (and nowhere else)


def write_detail(out):
    """Everything bench.py measured (sub-result bodies, per-leg rooflines, parity explanations, notes) as `bench_detail.json` next to
    bench.py and, where the GPU box merges files back, under gpurun_out/.  Returns the paths written (relative to the repo)."""
    paths = []
    if DETAIL_OUT:
        with open(DETAIL_OUT, "w") as f:
            json.dump(out, f)
        return [DETAIL_OUT]
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if d != ROOT and not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                json.dump(out, f, indent=1)
            paths.append(os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT))
        except OSError:
            pass
    return paths


def compact_line(out, detail_paths=()):
    """The ONE JSON line of the bench contract, <= LINE_LIMIT bytes: the contract's scalar keys, `config` (workload string <= 300 chars),
    ONE copy of the per-class rows (`roofline.classes` = {class: [launches per batch, us per launch, TFLOP/s]}), `cpu_baseline`, `parity`
    and `summary` as scalars.  Everything else lives in bench_detail.json (write_detail)."""
    def g(d, *ks):
        for k in ks:
            d = d.get(k) if isinstance(d, dict) else None
        return d

    def rnd(v, n=4):
        if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
            return v
        v = float(v)
        if not math.isfinite(v):
            return None
        return round(v, n) if abs(v) >= 1e-3 or v == 0 else float(f"{v:.3e}")

    keep = ("metric", "value", "unit", "n_gpus", "steps", "steps_requested", "warmup", "ms_per_step", "timed_region_s", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data")
    line = {k: rnd(out[k], 6) for k in keep if k in out}
    cfg = out.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("precision", "input", "batch_frames", "frames_in_flight", "num_scales", "exec", "hw_queues", "parallelism") if k in cfg}
    line["config"]["workload"] = str(cfg.get("workload", ""))[:300]
    roof = out.get("roofline") or {}
    r_ = {k: rnd(roof.get(k), 5) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_2q", "ms_per_launch", "launches_timed",
                                           "flops_per_launch", "batch_ms_sum_of_launches", "idle_frac_kernel_stamps")
          if k in roof}
    if isinstance(roof.get("how"), str) and roof["how"].startswith("FALLBACK"):
        r_["how"] = roof["how"][:200]
    if g(roof, "solo", "frac") is not None:
        r_["solo_frac"] = rnd(roof["solo"]["frac"])
    byp = roof.get("by_mfma_passes") or {}
    if byp:
        r_["us_by_mfma_passes"] = {p: rnd(v["ms_per_launch"] * 1e3, 2) for p, v in byp.items()}
    if roof.get("device_stamps"):   # the same launches by device-side stamps: [us plain, us compensated, fraction of peak] — what a kernel trace reports
        ds = roof["device_stamps"]
        r_["device_stamps"] = {"us_by_mfma_passes": {p_: rnd(v, 2) for p_, v in ds["us_by_mfma_passes"].items()}, "frac": rnd(ds["frac"])}
    if roof.get("classes"):
        r_["classes"] = {k: [v["steps_per_batch"], round(v["us_per_launch"], 1), round(v["tflops"])] for k, v in roof["classes"].items()}
        r_["classes_columns"] = "launches per batch, us per launch, algorithmic TFLOP/s"
    line["roofline"] = r_
    line["conv_stack_frac"] = rnd(g(out, "conv_stack_whole_frame", "frac"))
    lat = out.get("latency_ms") or {}
    line["latency_ms"] = {k: rnd(lat[k], 3) for k in ("p50_single_frame", "p95_single_frame", "p50_pipelined", "p95_pipelined") if k in lat}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": rnd(cb.get("value"), 4), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": str(cb.get("sample", ""))[:200], "torch_conv_fps": rnd(cb.get("torch_conv_fps"), 3)}
    par = out.get("parity")
    if par:
        line["parity"] = {k: rnd(par.get(k), 6) for k in ("verdict", "frames", "people_ref", "people_engine", "people_matched", "joints_ref", "joints_matched",
                                                        "numeric_out_of_tol", "joints_structural", "structural_explained", "replay_identical",
                                                        "max_dx_px", "max_dy_px", "max_dc", "map_max_err", "post_on_engine_maps_bit_exact") if k in par}
        line["parity"]["verdict"] = str(par.get("verdict", ""))[:120]
        if isinstance(par.get("structured"), dict):
            line["parity"]["structured"] = str(par["structured"].get("verdict", ""))[:60]
        if par.get("exact_modes"):   # {mode: [people matched, people of the fp32 reference, frames/s]} on the same noise frames
            line["parity"]["exact_modes"] = par["exact_modes"]
    for k in ("per_rank_frames_per_s", "comm_backend", "comm_note"):
        if out.get(k) is not None:
            line[k] = [rnd(v, 1) for v in out[k]] if isinstance(out[k], list) else str(out[k])[:200]
    if isinstance(out.get("weight_broadcast"), dict):
        line["weight_broadcast"] = {k: rnd(v, 3) for k, v in out["weight_broadcast"].items() if isinstance(v, (int, float, bool))}
    line["detail"] = list(detail_paths)
    line["summary"] = {k: v for k, v in (out.get("summary") or {}).items() if not isinstance(v, (dict, list))}   # scalars only
    s = json.dumps(line, separators=(",", ":"))
    for drop in (("roofline", "classes_columns"), ("cpu_baseline", "sample"), ("parity", "structured"), ("config", "workload")):   # never reached at today's sizes
        if len(s) <= LINE_LIMIT:
            break
        line[drop[0]].pop(drop[1], None)
        s = json.dumps(line, separators=(",", ":"))
    if len(s) > LINE_LIMIT:
        line["summary"] = {k: v for k, v in line["summary"].items() if k in ("fps", "dominant_frac")}
        s = json.dumps(line, separators=(",", ":"))
    return s


def respawn_under_launcher(args):
    """`python bench.py --gpus N` with no launcher: start N ranks of this script under torch.distributed.run."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def init_comm(prefer, rank, world, local):
    """Process group for the barrier / MAX-reduce / gathers around the timed region (nothing on the data path).  `prefer`: "nccl" (RCCL
    over xGMI: one device per rank), "gloo", or "auto" = nccl where torch sees a GPU.  RCCL runs on torch's bundled HIP runtime next to
    the engine's system runtime; if it cannot initialise or its first collective fails, every rank falls back to gloo on the same store
    (the failure is symmetric: all ranks load the same two runtimes) and the line says so in `comm_backend`.
    Returns (dist module, backend name, note)."""
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    note = None
    if prefer == "auto":
        prefer = "nccl" if torch.cuda.is_available() else "gloo"
    if prefer == "nccl":
        try:
            torch.cuda.set_device(local)            # every "cuda" tensor of the reductions below (dispatch.py) must live on THIS rank's device
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=120))
            t = torch.ones(1, device=torch.device("cuda", local))
            dist.all_reduce(t)                      # the first collective is where a broken RCCL shows
            torch.cuda.synchronize(local)
            if int(t.item()) != world:
                raise RuntimeError(f"all_reduce returned {t.item()} for {world} ranks")
            return dist, "nccl", None
        except Exception as ex:  # noqa: BLE001
            note = f"nccl failed ({type(ex).__name__}: {str(ex).splitlines()[0][:160]}); fell back to gloo"
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            # (same MASTER_PORT: under torch.distributed.run the agent hosts the store there and every worker is a client)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    return dist, "gloo", note


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--min_seconds", type=float, default=2.0, help="lower bound of the timed region; --steps is scaled up to reach it")
    ap.add_argument("--num_scales", type=int, default=1)
    ap.add_argument("--scale_gap", type=float, default=0.3)
    ap.add_argument("--precision", default="mixed", choices=["mixed", "fp16", "f16x3", "fp32"],
                    help="mixed (default) = fp16 MFMA with fp8 error-compensation chunks on the error-dominant layers: the fastest mode inside the "
                         "north-star tolerance (+-1e-3 on maps normalised to 1, tests/test_precision.py); fp16 = single-pass everywhere (2x outside it); "
                         "f16x3 = every layer as three fp16 passes; fp32 = exact-f32 MFMA")
    ap.add_argument("--split_layers", default=None, help="override the split set of --precision mixed (rtp_config.split_layers syntax)")
    ap.add_argument("--calibrate", type=int, default=0, help="K > 0: rtp_calibrate_precision on K synthetic frames at engine creation (the split set is checked on the loaded weights)")
    ap.add_argument("--in_flight", type=int, default=None, help="frames in flight per GPU (default 7 = three launched batches of 2 + one staged frame; "
                    "3 at several scales; 10 for MPI = two batches of 5): measured optima, profiles/r03_in_flight.txt")
    ap.add_argument("--batch_frames", type=int, default=None, help="frames whose conv stacks share one launch sequence (1 = the reference's one frame per Forward); "
                    "default 2 for COCO 656x368 at 1 scale (248 workgroups per 1/8-resolution launch), 1 at several scales, 5 for MPI 496x368 "
                    "(240 workgroups of 128x128 tiles)")
    ap.add_argument("--input", default="host_u8", choices=["host_u8", "resident"], help="host_u8 (default) = BASELINE configs[1] as written: pageable host u8 1280x720 frames through "
                    "rtp_submit_frame (H2D + device pre-processing inside the timed region); resident = net inputs already in HBM (rtp_submit_device)")
    ap.add_argument("--model", default="coco", choices=["coco", "mpi"], help="coco = BASELINE configs[1..3] (656x368); mpi = configs[4] (15 parts, 496x368)")
    ap.add_argument("--exec", dest="exec_mode", default="graph", choices=["graph", "eager"])
    ap.add_argument("--comm", default="auto", choices=["auto", "nccl", "gloo"], help="process group for the barrier / reductions around the timed region (N > 1)")
    ap.add_argument("--devices", default=None, help="device of every rank, e.g. 0,0: lets N ranks share one GPU (a test hook like rtpose.bin --devices; default: rank i uses device LOCAL_RANK)")
    ap.add_argument("--broadcast_weights", action="store_true", help="N > 1: rank 0's packed weight arena is broadcast to the other ranks (rtp_weight_blob_export / import) "
                    "instead of every rank keeping the copy it packed itself; one-time, outside the timed region")
    ap.add_argument("--hw_queues", type=int, default=8, help="GPU_MAX_HW_QUEUES for this process (set before the first HIP call unless the environment already has it): "
                    "with at least as many hardware queues as batch contexts the engine gives every context one stream and so one queue to itself "
                    "(+12 %% frames/s at batches of 2 against the runtime's default 4; engine.cpp 'hardware queues', profiles/r06_experiments.txt)")
    ap.add_argument("--detail_out", default=None, help="(internal) write the detail JSON to this path only: how the parent run collects a leg it started as a subprocess")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_sub_results", action="store_true")
    ap.add_argument("--no_parity", action="store_true", help="skip the people-level parity verdict against the CPU oracle chain (3 frames)")
    ap.add_argument("--dry_dispatch", action="store_true", help="self-test of the multi-rank plumbing without a GPU (no engine): tests/test_bench_spawn.py")
    args = ap.parse_args()
    if args.batch_frames is None:
        args.batch_frames = 5 if args.model == "mpi" else (2 if args.num_scales == 1 else 1)
    global DETAIL_OUT, USER_HW_QUEUES
    DETAIL_OUT = args.detail_out
    USER_HW_QUEUES = os.environ.get("GPU_MAX_HW_QUEUES")       # a value from the caller's environment wins, here and in the legs
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(args.hw_queues))   # (the HIP runtime reads it once, at its first call: torch and the engine are imported below)
    if args.in_flight is None:
        table = {("mpi", 5): 15, ("coco", 2): 7 if args.num_scales == 1 else 6, ("coco", 1): 7 if args.num_scales == 1 else 3}   # measured optima (profiles/r03_in_flight.txt)
        args.in_flight = max(table.get((args.model, args.batch_frames), 2 * args.batch_frames), args.batch_frames)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_launcher(args))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.devices:
        devs = [int(d) for d in args.devices.split(",")]
        if len(devs) != world:
            sys.exit(f"bench.py: --devices names {len(devs)} devices for {world} rank(s)")
        local = devs[rank]
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); using WORLD_SIZE", file=sys.stderr)
    from caffe_rtpose_amd.dispatch import timed_region, aggregate_fps, scaled_steps, broadcast_bytes
    dist, comm, comm_note = None, None, None
    if world > 1:
        dist, comm, comm_note = init_comm(args.comm, rank, world, local)
    red_dev = "cuda" if comm == "nccl" else "cpu"

    if args.dry_dispatch:  # plumbing only: spawn, rendezvous (with the gloo fall-back), barrier/MAX timing, weight-blob broadcast, ONE line from rank 0
        def fake(n, base):
            time.sleep(0.001 * n)
        steps = scaled_steps(args.steps, 1000.0, min(args.min_seconds, 0.2), dist)
        dt, dt_local = timed_region(fake, steps, args.warmup, dist, return_local=True, reduce_device=red_dev)
        per_rank = [steps / dt_local]
        blob_ok = None
        if dist is not None:   # the same gather the GPU path does
            t = torch.zeros(world, dtype=torch.float64, device=red_dev)
            t[rank] = steps / dt_local
            dist.all_reduce(t)
            per_rank = [float(v) for v in t.tolist()]
            if args.broadcast_weights:   # the blob path of --broadcast_weights with a stand-in blob
                blob = np.arange(1 << 16, dtype=np.uint8) if rank == 0 else np.zeros(1 << 16, np.uint8)
                blob = broadcast_bytes(blob, 0, dist, red_dev)
                ok = torch.tensor([int(np.array_equal(blob, np.arange(1 << 16, dtype=np.uint8)))], device=red_dev)
                dist.all_reduce(ok)
                blob_ok = int(ok.item()) == world
        if rank == 0:
            print(json.dumps({"metric": "dispatch self-test (no GPU work)", "value": aggregate_fps(steps, world, dt), "unit": "frames/s", "n_gpus": world,
                              "steps": args.steps, "steps_timed": steps, "warmup": args.warmup, "data": "none", "per_rank_frames_per_s": per_rank,
                              "comm_backend": comm, "comm_note": comm_note, "weight_blob_broadcast_ok": blob_ok}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import caffe_rtpose_amd as r
    pinned_to = None
    if world > 1:   # like rtpose.bin's workers: this rank's threads next to its GPU, so the pinned staging buffers it fills are on that NUMA node
        try:
            cpus = r.device_local_cpus(local)
            if cpus:
                os.sched_setaffinity(0, cpus)
                pinned_to = f"{len(cpus)} CPUs local to GPU {local}"
        except Exception:  # noqa: BLE001
            pass
    seed = 1
    PREC = {"fp16": r.PREC_FP16, "fp32": r.PREC_FP32, "mixed": r.PREC_MIXED, "f16x3": r.PREC_F16X3}

    def make_engine(precision, num_scales, scale_gap, batch_frames, in_flight, model=None, receive=None):
        """receive = (precision id, split rules) of the engine whose packed weights this one will take: created WITHOUT weights
        (rtp_config.defer_weights), on that engine's plan."""
        mid, W, H = MODELS[model or args.model][:3]
        kw = dict(precision=PREC[precision], split_layers=args.split_layers, calibrate_frames=args.calibrate if precision == "mixed" else 0)
        if receive is not None:
            kw = dict(precision=receive[0], split_layers=receive[1], calibrate_frames=-1, defer_weights=1)
        return r.Engine(r.Config(device_id=local, model=mid, net_w=W, net_h=H, num_scales=num_scales, scale_gap=scale_gap,
                                 frames_in_flight=in_flight, batch_frames=batch_frames, synthetic_seed=seed,
                                 exec_mode=r.EXEC_GRAPH if args.exec_mode == "graph" else r.EXEC_EAGER, **kw))

    def device_frames(eng, n=8):
        # synthetic net inputs resident in HBM before the timed region (u8/256-0.5 like process_and_pad_image), in buffers of the ENGINE's runtime
        rs = np.random.RandomState(1234 + rank)
        return [eng.device_frame(rs.randint(0, 256, (eng.N, 3, eng.net_h, eng.net_w)).astype(np.float32) / 256.0 - 0.5) for _ in range(n)]

    def host_frames(n=8):
        return [r.synth_frame(1280, 720, i, seed=2 + rank) for i in range(n)]   # the procedural "synthetic 720p video" 

    def measure(eng, submit, steps, warmup, in_flight, min_seconds):
        """Pipelined submit/collect of `steps` frames (at least min_seconds; no instrumentation inside the region).  Returns a dict."""
        lat = []
        host = {"submit": 0.0, "collect": 0.0, "frames": 0}

        def run(nsteps, base_tag):
            sub = col = 0
            t_sub = {}
            while col < nsteps:
                while sub < nsteps and eng.in_flight() < in_flight:
                    t_sub[sub] = time.perf_counter()
                    submit(sub, base_tag + sub)
                    sub += 1
                    if base_tag:
                        host["submit"] += time.perf_counter() - t_sub[sub - 1]
                t_c = time.perf_counter()
                tag, _, _ = eng.collect()
                assert tag == base_tag + col
                if base_tag:
                    now = time.perf_counter()
                    host["collect"] += now - t_c
                    host["frames"] += 1
                    lat.append(now - t_sub.pop(col))
                col += 1

        run(warmup, 0)
        eng.synchronize()
        t0 = time.perf_counter()
        ncal = max(16, 2 * in_flight)
        run(ncal, 0)                      # calibration pass (untimed): frames/s estimate for the step scaling
        eng.synchronize()
        est = ncal / (time.perf_counter() - t0)
        nsteps = scaled_steps(steps, est * 1.1, min_seconds, dist)
        for attempt in range(4):          # the short calibration pass under-estimates the steady rate: repeat until the region is long enough
            lat.clear()
            host.update(submit=0.0, collect=0.0, frames=0)
            dt, dt_local = timed_region(run, nsteps, 0, dist, eng.synchronize, red_dev, return_local=True)
            if dt >= min_seconds or attempt == 3:
                break
            nsteps = int(math.ceil(nsteps * min_seconds / dt * 1.15))   # dt is the MAX over ranks: every rank takes the same decision
        return {"dt": dt, "dt_local": dt_local, "steps_timed": nsteps, "fps": aggregate_fps(nsteps, world, dt), "lat": list(lat),
                "host_ms": {"submit_calls": host["submit"] / max(host["frames"], 1) * 1e3, "collect_calls_incl_wait": host["collect"] / max(host["frames"], 1) * 1e3}}

    def roofline_pass(eng, frames, batch_frames, precision, num_scales, model):
        """The same plan ONE BATCH AT A TIME (submit batch_frames frames, collect them, repeat) with a HIP event pair around every
        dominant-class launch, recorded on the stream the launch runs on (rtp_kernel_timing; batches are launched eagerly while it is on).
        No other frame's kernels are on the chip, so a pair brackets that launch alone inside whole frames (real layer sequence, real L2
        state): the quantity a rocprofv3 kernel trace averages (profiles/).  Nothing compares clocks of different XCDs."""
        eng.kernel_timing(3)        # an event pair around EVERY step of a full batch; the dominant class's totals keep their meaning
        nb = max(1, batch_frames)
        for b in range(40):
            for j in range(nb):
                eng.submit_device(frames[(b * nb + j) % len(frames)], tag=b * nb + j)
            for j in range(nb):
                eng.collect()
            if b % 8 == 7:
                eng.kernel_timing(-1)   # idle here: fold the event pairs into the totals so the table of pairs never fills
        dom_ms, dom_n, dom_flops = eng.kernel_timing(-1)
        if eng.probe_dropped()["timing_pairs"]:
            raise RuntimeError(f"roofline pass truncated: {eng.probe_dropped()} launches were not timed (rtp_probe_dropped)")
        byp = eng.kernel_timing_by_passes()
        steps_t = eng.kernel_timing_steps()
        eng.kernel_timing(0)
        peak = 157.3e12 if precision == "fp32" else 2.5e15
        solo_ms, _ = eng.bench_dominant_conv(iters=200)   # the plain instantiation back to back, alone on the chip (HIP events around 200 launches)
        roof = roofline_block(dom_ms, dom_n, dom_flops, byp, solo_ms, peak, pmc_traffic(precision, batch_frames, num_scales, model),
                              pmc_traffic(precision, batch_frames, num_scales, model, "_2q"))
        try:
            plan = [ln for ln in r.plan_summary(eng.cfg).splitlines() if ln.startswith("step ")]
            if len(plan) == len(steps_t):
                classes, total = kernel_classes(plan, eng.conv_layers(), steps_t, eng.net_w, eng.net_h, nb * num_scales, peak)
                roof["classes"] = classes
                roof["batch_ms_sum_of_launches"] = total
            # the same launches once more, seen from the device (cross-check of the event pairs against what a kernel trace reports)
            eng.stamp_probe(1)
            for b in range(20):
                for j in range(nb):
                    eng.submit_device(frames[(b * nb + j) % len(frames)], tag=b * nb + j)
                for j in range(nb):
                    eng.collect()
            st = eng.stamp_probe(-1)
            eng.stamp_probe(0)
            roof["device_stamps"] = stamp_dominant(st, plan, dom_flops, peak)
        except Exception as ex:  # noqa: BLE001
            roof["classes_error"] = str(ex)
        return roof

    mid, W, H, _, _, _, gflop = MODELS[args.model]
    wb = None
    if dist is not None and args.broadcast_weights:
        # one-time weight distribution (outside the timed region): rank 0 reads / generates, packs (and calibrates) ONCE; the other ranks learn its
        # plan (precision mode + split set), are created without weights (rtp_config.defer_weights) and import rank 0's packed arena
        t0 = time.perf_counter()
        eng = None
        enc = b""
        if rank == 0:
            eng = make_engine(args.precision, args.num_scales, args.scale_gap, args.batch_frames, args.in_flight)
            rules, mode = eng.split_layers()
            enc = f"{mode}|{rules}".encode()
        nlen = torch.tensor([len(enc)], dtype=torch.int64, device=red_dev)     # the rule list has no fixed size: its length first
        dist.broadcast(nlen, 0)
        meta = np.zeros(int(nlen.item()), np.uint8)
        if rank == 0:
            meta[:] = np.frombuffer(enc, np.uint8)
        meta = broadcast_bytes(meta, 0, dist, red_dev)
        if rank != 0:
            mode, rules = bytes(meta).rstrip(b"\0").decode().split("|", 1)
            eng = make_engine(args.precision, args.num_scales, args.scale_gap, args.batch_frames, args.in_flight, receive=(int(mode), rules))
        blob = eng.weight_blob() if rank == 0 else np.zeros(eng.weight_blob_bytes(), np.uint8)   # (non-zero ranks only need the length: no D2H copy)
        blob = broadcast_bytes(blob, 0, dist, red_dev)
        if rank != 0:
            eng.load_weight_blob(blob)
        wb = {"bytes": int(blob.nbytes), "seconds": time.perf_counter() - t0, "receivers_created_without_weights": True}
        del blob
    else:
        eng = make_engine(args.precision, args.num_scales, args.scale_gap, args.batch_frames, args.in_flight)
    frames = device_frames(eng)
    u8 = host_frames()
    if args.input == "host_u8":
        submit = lambda i, tag: eng.submit_frame(u8[i % len(u8)], tag=tag)       # noqa: E731
    else:
        submit = lambda i, tag: eng.submit_device(frames[i % len(frames)], tag=tag)   # noqa: E731
    m = measure(eng, submit, args.steps, args.warmup, args.in_flight, args.min_seconds)
    per_rank = None
    if dist is not None:   # every rank's own rate: a straggler shows
        t = torch.zeros(world, dtype=torch.float64, device=red_dev)
        t[rank] = m["steps_timed"] / m["dt_local"]
        dist.all_reduce(t)
        per_rank = [float(v) for v in t.tolist()]
    # Roofline pass (separate from the timed region, which runs un-instrumented)
    roof = roofline_pass(eng, frames, args.batch_frames, args.precision, args.num_scales, args.model)
    stage = eng.last_stage_ms()
    busy = None
    if True:        # the same pipelined loop once more with the busy probe on (~0.5 s): when did the engine have work on the GPU — no profiler attached.
        try:        # EVERY rank runs it (measure() holds the ranks' barrier and reductions); rank 0's account goes into the line
            eng.busy_probe(1)
            mb = measure(eng, submit, 200, 20, args.in_flight, 0.5)
            busy = busy_account(eng.busy_probe(-1))
            dropped = eng.probe_dropped()
            if busy and (dropped["busy_spans"] or dropped["busy_graph_frames"]):
                busy["truncated"] = dropped
            eng.busy_probe(0)
            if busy:
                busy["frames_per_s_with_probe"] = mb["fps"] * (1 if world == 1 else 1.0 / world)
        except Exception as ex:  # noqa: BLE001
            busy = {"error": str(ex)}

    stamps = None
    try:        # the same loop with the residency stamps on (~0.5 s): share of the wall with no kernel on the chip, unprofiled
        eng.stamp_probe(1)
        ms_ = measure(eng, submit, 200, 20, args.in_flight, 0.5)
        eng.synchronize()
        stamps = stamp_account(eng.stamp_probe(-1))
        eng.stamp_probe(0)
        if stamps:
            stamps["frames_per_s_with_probe"] = ms_["fps"] * (1 if world == 1 else 1.0 / world)
    except Exception as ex:  # noqa: BLE001
        stamps = {"error": str(ex)}
        try:
            eng.stamp_probe(0)
        except Exception:  # noqa: BLE001
            pass

    if rank == 0:
        # dominant kernel: the paired 7x7 128->128 convolution (40 of the 92 layers, 50% of all FLOPs).  Average launch duration over
        # EVERY launch of that kernel shape (both instantiations: the plain fp16 one and the fp8-compensated one) inside whole frames,
        # from HIP event pairs on the launch's stream.  `achieved` counts ALGORITHMIC flops (2*Cout*Cin*k*k*H*W per image): an error-compensated
        # launch spends 2 (fp16 + fp8 chunks) or 3 (three fp16 passes) pass-times of the matrix pipe on them, reported separately under
        # `executed` (pass-time equivalents: an fp8 chunk takes the time of the fp16 chunk it corrects).
        peak = 157.3e12 if args.precision == "fp32" else 2.5e15
        fps = m["fps"]
        whole = {"achieved": fps / world * gflop * 1e9 * args.num_scales / 1e12, "unit": "TFLOP/s",
                 "frac": fps / world * gflop * 1e9 * args.num_scales / peak}
        src = ("pageable host u8 1280x720 frames through rtp_submit_frame: pinned copy + H2D + device warp / INTER_AREA / normalise / pad inside the timed region"
               if args.input == "host_u8" else "net inputs resident in HBM (rtp_submit_device)")
        out = {
            "metric": f"frames/sec (whole node) at {W}x{H} {args.model.upper()} model",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": m["steps_timed"], "steps_requested": args.steps, "steps_timed": m["steps_timed"],
            "warmup": args.warmup, "ms_per_step": m["dt"] / m["steps_timed"] * 1e3, "timed_region_s": m["dt"], "timed_seconds": m["dt"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f16", "data": "synthetic",
            "config": {"precision": args.precision, "precision_note": PREC_NOTE[args.precision],
                       "workload": f"{args.model.upper()} {W}x{H}, {args.num_scales} scale(s), synthetic 720p video: {src} -> conv stack+ImResize+NMS+connect -> joints on the host; "
                                   f"precision mode {args.precision}, {args.in_flight} frames in flight/GPU in batches of {args.batch_frames}, {args.exec_mode} launches, synthetic weights",
                       "input": args.input, "batch_frames": args.batch_frames, "frames_in_flight": args.in_flight, "num_scales": args.num_scales, "exec": args.exec_mode,
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "parallelism": f"frame-sharded replicas x{world}"},
            "latency_ms": {"p50_pipelined": float(np.percentile(m["lat"], 50) * 1e3), "p95_pipelined": float(np.percentile(m["lat"], 95) * 1e3),
                           "batch_on_device": stage["total"]},
            "host_ms_per_frame": m["host_ms"], "roofline": roof, "conv_stack_whole_frame": whole, "gpu_busy": busy, "kernel_residency": stamps,
        }
        if stamps and "idle_frac" in stamps:
            roof["idle_frac_kernel_stamps"] = round(stamps["idle_frac"], 4)
        if args.precision == "mixed":
            out["config"]["split_layers"] = eng.split_layers()[0]
            if args.calibrate:
                out["config"]["calibration"] = eng.calibration_report()
        if world > 1:
            out["rank0_cpu_affinity"] = pinned_to
            out["comm_backend"] = comm
            if comm_note:
                out["comm_note"] = comm_note
            if wb:
                out["weight_broadcast"] = wb
        if per_rank:
            out["per_rank_frames_per_s"] = per_rank
        if world == 1:
            try:   # one frame ALONE: commit -> joints on the host (the reference's per-frame latency, rtpose.cpp:1430), nothing else in flight
                e1 = make_engine(args.precision, args.num_scales, args.scale_gap, 1, 1)
                lat1 = []
                for i in range(70):
                    t0 = time.perf_counter()
                    e1.submit_frame(u8[i % len(u8)], tag=i)
                    e1.collect()
                    lat1.append(time.perf_counter() - t0)
                lat1 = lat1[10:]
                out["latency_ms"]["p50_single_frame"] = float(np.percentile(lat1, 50) * 1e3)
                out["latency_ms"]["p95_single_frame"] = float(np.percentile(lat1, 95) * 1e3)
                out["latency_ms"]["single_frame_what"] = "batch_frames 1, 1 frame in flight: host u8 720p frame -> rtp_submit_frame -> rtp_collect returns the joints (commit -> joints on host)"
                e1.close()
            except Exception as ex:  # noqa: BLE001
                out["latency_ms"]["single_frame_error"] = str(ex)
        if world == 1 and not args.no_sub_results:
            out["sub_results"] = sub_results(args, r, eng, make_engine, device_frames, host_frames, measure, roofline_pass, frames, np)
        if world == 1 and not (args.no_cpu_baseline and args.no_parity):
            orc_fr = oracle_frames(eng, args.num_scales, args.model, 3 if not args.no_cpu_baseline else 1)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(eng, args.num_scales, args.model, oracle=orc_fr)
            if not args.no_parity:
                try:
                    out["parity"] = parity_report(eng, orc_fr[1][1:], args.model, args.num_scales, args.scale_gap)
                except Exception as ex:  # noqa: BLE001
                    out["parity"] = {"error": str(ex), "verdict": f"FAIL: {ex}"}
                if args.precision == "mixed" and not args.no_sub_results and "error" not in out["parity"]:
                    # the precision / throughput trade as one table: the same frames through the exact modes (bench_detail.json; a short form in the line)
                    par = out["parity"]
                    mixed_row = {"fps_resident": round(float(((out.get("sub_results") or {}).get("resident_input") or {}).get("value") or fps), 1),
                                 **{k: par.get(k) for k in ("people_ref", "people_engine", "people_matched", "joints_ref", "joints_matched", "joints_structural",
                                                            "numeric_out_of_tol")}, "map_max_err": par.get("map_max_err")}
                    table = exact_modes_table(make_engine, device_frames, measure, orc_fr[1][1:], args.model, args.num_scales, args.scale_gap,
                                              args.batch_frames, args.in_flight, mixed_row)
                    out["exact_modes"] = table
                    par["exact_modes"] = {k: [v.get("people_matched"), v.get("people_ref"), v.get("fps_resident")] for k, v in table.items() if "error" not in v}
        # A compact summary of every leg: the LAST key of the line (a truncated tail still shows it) and, as flat scalars, inside `roofline`
        # (records that keep `roofline` but not `sub_results` still show every leg's headline number).
        sr = out.get("sub_results") or {}

        def g(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return d

        rnd = lambda v, n=1: None if v is None else round(float(v), n)   # noqa: E731
        summ = {"fps": rnd(fps), "p50_single_frame_ms": rnd(g(out, "latency_ms", "p50_single_frame"), 3), "p50_pipelined_ms": rnd(g(out, "latency_ms", "p50_pipelined"), 2),
                "dominant_frac": rnd(roof.get("frac"), 4), "resident_fps": rnd(g(sr, "resident_input", "value")),
                "scales3_fps": rnd(g(sr, "scales3_gap0.15", "value")), "scales3_p50_ms": rnd(g(sr, "scales3_gap0.15", "p50_ms"), 2),
                "scales3_dominant_frac": rnd(g(sr, "scales3_gap0.15", "roofline", "frac"), 4),
                "mpi_fps": rnd(g(sr, "mpi_496x368", "value")), "mpi_dominant_frac": rnd(g(sr, "mpi_496x368", "roofline", "frac"), 4),
                "fp16_fps": rnd(g(sr, "precision_fp16_single_pass", "value")), "fp32_fps": rnd(g(sr, "precision_fp32", "value")),
                "cpu_fps": rnd(g(out, "cpu_baseline", "value"), 3), "idle_frac_kernel_stamps": rnd(g(out, "kernel_residency", "idle_frac"), 4),
                "kernels_resident_avg": rnd(g(out, "kernel_residency", "kernels_resident_avg"), 2),
                "parity": (g(out, "parity", "verdict") or "")[:40], "parity_replay_identical": g(out, "parity", "replay_identical"),
                "parity_scales3": (g(sr, "scales3_gap0.15", "parity", "verdict") or "")[:40], "parity_mpi": (g(sr, "mpi_496x368", "parity", "verdict") or "")[:40]}
        out["summary"] = summ
        detail_paths = write_detail(out)
        print(compact_line(out, detail_paths))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def sub_results(args, r, eng, make_engine, device_frames, host_frames, measure, roofline_pass, frames, np):
    """Extra legs on one GPU (each >= 1 s timed); the headline `value` is not affected."""
    import _synth
    res = {}
    short = dict(steps=50, warmup=10, min_seconds=1.0)
    gflop = MODELS[args.model][6]
    # (1) the other input mode of the headline engine: net inputs already resident in HBM (no PCIe, no pre-processing) — or, when the
    #     headline was asked for with --input resident, BASELINE configs[1] as written
    try:
        if args.input == "host_u8":
            m = measure(eng, lambda i, tag: eng.submit_device(frames[i % len(frames)], tag=tag), in_flight=args.in_flight, **short)
            res["resident_input"] = {"value": m["fps"], "unit": "frames/s", "steps_timed": m["steps_timed"], "host_ms_per_frame": m["host_ms"],
                                     "what": "rtp_submit_device: net input already in HBM -> conv stack -> post; no PCIe, no pre-processing (round 3's headline)"}
        else:
            u8 = host_frames()
            m = measure(eng, lambda i, tag: eng.submit_frame(u8[i % 8], tag=tag), in_flight=args.in_flight, **short)
            res["host_u8_720p_frames"] = {"value": m["fps"], "unit": "frames/s", "steps_timed": m["steps_timed"], "host_ms_per_frame": m["host_ms"],
                                          "what": "rtp_submit_frame: pageable host u8 1280x720 frame -> pinned copy -> H2D -> device pre-processing -> conv stack -> post, PCIe inclusive"}
    except Exception as ex:  # noqa: BLE001
        res["resident_input" if args.input == "host_u8" else "host_u8_720p_frames"] = {"error": str(ex)}
    # (2) post-processing alone (production path, from the low-res maps): noise worst case + analytic heat maps with P planted people
    try:
        tables = r.model_tables(0 if args.model == "coco" else 1)
        post = {}
        cases = [("noise", _synth.smooth_field(eng.heat_channels, eng.low_h, eng.low_w, seed=3)[None].repeat(eng.N, 0))]
        for P in (1, 5, 20):
            cases.append((f"P{P}", _synth.people_lowres(0 if args.model == "coco" else 1, tables, P, eng.low_h, eng.low_w, seed=3, N=eng.N)[0]))
        for name, low in cases:
            low = np.ascontiguousarray(low, np.float32)
            t = []
            for _ in range(6):
                _, _, n = eng.post_from_lowres(low)
                t.append(eng.last_stage_ms())
            t = t[1:]
            post[name] = {"people": n, "nms_ms": float(np.mean([x["nms"] for x in t])), "connect_ms": float(np.mean([x["connect"] for x in t]))}
        res["postproc_alone"] = post
    except Exception as ex:  # noqa: BLE001
        res["postproc_alone"] = {"error": str(ex)}
    # The legs below run as bench.py processes of their own (leg_subprocess): each gets the hardware-queue count of its batch size and a fresh
    # stream -> queue arrangement, like a driver-style run of that configuration.
    par = [] if not args.no_parity else ["--no_parity"]

    def pick(d, extra=()):
        if "error" in d:
            return d
        o = {"value": d["value"], "unit": d["unit"], "steps_timed": d["steps"], "p50_ms_pipelined": d["latency_ms"].get("p50_pipelined"),
             "p50_ms": d["latency_ms"].get("p50_pipelined"), "batch_frames": d["config"]["batch_frames"], "frames_in_flight": d["config"]["frames_in_flight"],
             "hw_queues": d["config"].get("hw_queues"), "input": d["config"]["input"], "conv_stack_frac_of_peak": d["conv_stack_whole_frame"]["frac"]}
        for k in extra:
            if k in d:
                o[k] = d[k]
        return o
    # (2b) BASELINE configs[4]: the MPI 15-part model at 496x368, fp16 MFMA path, batches of 5 — from host u8 frames like the headline,
    #      with its own roofline (dominant kernel of ITS plan) and its own parity (connectLimbs, rtpose.cpp:549-751)
    if args.model == "coco" and args.num_scales == 1:
        res["mpi_496x368"] = pick(leg_subprocess(["--model", "mpi", "--precision", args.precision] + par), ("roofline", "parity"))
    # (3) 3 scales, gap 0.15 (BASELINE configs[2], the north-star target configuration): one frame (3 images) per launch sequence, 3 in flight
    if args.num_scales == 1 and args.model == "coco":
        res["scales3_gap0.15"] = pick(leg_subprocess(["--num_scales", "3", "--scale_gap", "0.15", "--precision", args.precision] + par), ("roofline", "parity"))
    # (3b) single-pass fp16 everywhere: the fastest mode, ~2x outside the +-1e-3 tolerance (why it is not the default)
    if args.precision == "mixed" and args.num_scales == 1:
        res["precision_fp16_single_pass"] = pick(leg_subprocess(["--precision", "fp16", "--input", "resident", "--no_parity", "--model", args.model]))
        if "error" not in res["precision_fp16_single_pass"]:
            res["precision_fp16_single_pass"]["note"] = PREC_NOTE["fp16"]
    # (4) the exact-f32 MFMA path (reference arithmetic: fp32 throughout)
    if args.precision != "fp32" and args.num_scales == 1:
        res["precision_fp32"] = pick(leg_subprocess(["--precision", "fp32", "--input", "resident", "--no_parity", "--model", args.model]))
        if "error" not in res["precision_fp32"]:
            res["precision_fp32"]["conv_stack_frac_of_f32_mfma_peak"] = res["precision_fp32"]["conv_stack_frac_of_peak"]
    return res


if __name__ == "__main__":
    main()
