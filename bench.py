#!/usr/bin/env python
"""bench.py — frames/sec of the rtpose hot path on MI355X (BASELINE.json metric).

A "step" = one frame: net input (num_scales x 3 x 368 x 656 fp32, already resident in HBM) ->
conv stack -> ImResize -> Nms -> connectLimbsCOCO -> joints on the host.  Workload = BASELINE.json
configs[1] ("COCO model 656x368, 1 scale, 1xMI355X") unless flags say otherwise.

Multi-GPU: one process per GPU, frames sharded (full replicas, no data-path collective — the reference's
--num_gpu dispatcher, rtpose.cpp:1463-1472) => weak scaling.  `python bench.py --gpus N` started bare
re-executes itself under torch.distributed.run with N ranks (RCCL); under a launcher it reads
RANK/LOCAL_RANK/WORLD_SIZE.

The timed region is never shorter than --min_seconds (default 2 s) and runs un-instrumented: `--steps K` is a MINIMUM, the region is
repeated with more frames until it is long enough, and the line reports the frames actually timed as `steps` (= `steps_timed`; the flag
as `steps_requested`) with `timed_region_s`.  (A 20-frame region with 7 frames in flight is mostly pipeline fill/drain and measures the
host, not the GPU.)  Defaults per workload are measured optima (profiles/r03_in_flight.txt): COCO 1 scale 7 frames in flight in batches of
2, several scales 3 in flight one frame per launch sequence, MPI 10 in flight in batches of 5.

`roofline` (separate pass right after the timed region): HIP event pairs around every launch of the dominant kernel shape, on the stream
the launch runs on, one batch at a time — roofline_block() below; `parity`: the engine's joints against the full fp32 oracle chain as
sets of people (parity_report()); `cpu_baseline`: the OpenMP oracle port and torch-CPU conv2d over the same layers.

On one GPU the line also carries `sub_results`: the same engine fed host u8 720p frames through
rtp_submit_frame (config 2 as written: H2D + device pre-processing inside the timed region), 3 scales
(config 3, the north-star target), single-pass fp16, the exact-f32 path, post-processing alone on analytic heat maps.
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
    net, fr = oracle if oracle is not None else oracle_frames(eng, num_scales, model, frames)
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


def parity_report(eng, fr, model, num_scales, scale_gap):
    """SURVEY section 7 / BASELINE.md section 3: the engine's joints (this precision mode, through rtp_submit / rtp_collect) against the
    full fp32 oracle chain conv -> ImResize -> Nms -> connectLimbs* on the same frames, as SETS of people (tests/_parity.py).
    Units: the synthetic network's maps have a maximum of ~5 where real confidences live in [0, 1]; scores and map errors are
    divided by max|reference map| (= stated for maps normalised to a maximum of 1, the unit of the +-1e-3 tolerance), positions
    are display pixels.  (Scaling the maps themselves into [0, 1], as tests/test_precision.py does for the peak test, leaves the
    noise network without a single person above connectLimbs' thresholds: nothing to compare.)"""
    import numpy as np
    import _oracle as orc
    import _parity
    mid, W, H, parts, max_peaks, thr, _ = MODELS[model]
    reps, map_err, post_exact = [], 0.0, True
    for x, ref, _ in fr:
        norm = float(np.abs(ref).max())
        res = orc.imresize(ref, W, H, 1.0, scale_gap)[0]
        peaks = orc.nms(res, parts, max_peaks, thr)
        nr, jr = orc.connect(mid, res, peaks, max_peaks, W, H, 1280, 720)
        eng.submit(x, tag=1)
        eng.flush()
        _, ne, je = eng.collect()
        reps.append(_parity.people_parity(je[:ne], jr[:nr], tol_px=1.0, tol_c=1e-3, c_norm=norm))
        got = eng.forward_heatmaps(x)
        map_err = max(map_err, float(np.abs(got - ref).max() / norm))
        # decomposition: the reference's post-processing applied to the ENGINE's maps must give the engine's joints bit for bit;
        # whatever differs between the two people sets is then a strict compare (Nms '>' / connect's greedy order) that the
        # conv stack's deviation — inside the map tolerance — decides the other way
        res_e = orc.imresize(got, W, H, 1.0, scale_gap)[0]
        n2, j2 = orc.connect(mid, res_e, orc.nms(res_e, parts, max_peaks, thr), max_peaks, W, H, 1280, 720)
        post_exact = post_exact and n2 == ne and np.array_equal(j2[:n2], je[:ne])
    tot = _parity.merge(reps)
    tot["map_max_err"] = map_err
    tot["post_on_engine_maps_bit_exact"] = bool(post_exact)
    tot["units"] = "x, y in display pixels (1280x720); scores and map errors for maps normalised to a maximum of 1"
    tot["reference"] = "CPU oracle, fp32 conv stack -> ImResize -> Nms -> connectLimbs*, same synthetic weights and frames"
    numeric = tot["max_dc"] <= 1e-3 and tot["max_dx_px"] <= 1.0 and tot["max_dy_px"] <= 1.0 and map_err <= 1e-3 and post_exact
    tot["verdict"] = "FAIL" if not numeric else ("pass" if tot["people_matched"] == tot["people_ref"] == tot["people_engine"] else
                                                 "numeric pass; people sets differ by near-tie compares on the synthetic noise maps (joints_structural)")
    return tot


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
    ap.add_argument("--in_flight", type=int, default=None, help="frames in flight per GPU (default 7 = three launched batches of 2 + one staged frame; "
                    "3 at several scales; 10 for MPI = two batches of 5): measured optima, profiles/r03_in_flight.txt")
    ap.add_argument("--batch_frames", type=int, default=None, help="frames whose conv stacks share one launch sequence (1 = the reference's one frame per Forward); "
                    "default 2 for COCO 656x368 at 1 scale (248 workgroups per 1/8-resolution launch), 1 at several scales, 5 for MPI 496x368 "
                    "(240 workgroups of 128x128 tiles)")

    ap.add_argument("--model", default="coco", choices=["coco", "mpi"], help="coco = BASELINE configs[1..3] (656x368); mpi = configs[4] (15 parts, 496x368)")
    ap.add_argument("--exec", dest="exec_mode", default="graph", choices=["graph", "eager"])
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_sub_results", action="store_true")
    ap.add_argument("--no_parity", action="store_true", help="skip the people-level parity verdict against the CPU oracle chain (3 frames)")
    ap.add_argument("--dry_dispatch", action="store_true", help="self-test of the multi-rank plumbing without a GPU (gloo, no engine): tests/test_bench_spawn.py")
    args = ap.parse_args()
    if args.batch_frames is None:
        args.batch_frames = 5 if args.model == "mpi" else (2 if args.num_scales == 1 else 1)
    if args.in_flight is None:
        args.in_flight = 10 if args.model == "mpi" else (7 if args.num_scales == 1 else 3) if args.batch_frames in (1, 2) else 2 * args.batch_frames

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_launcher(args))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); using WORLD_SIZE", file=sys.stderr)
    from caffe_rtpose_amd.dispatch import timed_region, aggregate_fps, scaled_steps
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_dispatch:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    if args.dry_dispatch:  # plumbing only: spawn, rendezvous, barrier/MAX timing, ONE line from rank 0
        def fake(n, base):
            time.sleep(0.001 * n)
        steps = scaled_steps(args.steps, 1000.0, min(args.min_seconds, 0.2), dist)
        dt, dt_local = timed_region(fake, steps, args.warmup, dist, return_local=True)
        per_rank = [steps / dt_local]
        if dist is not None:   # the same gather the GPU path does (there on the device)
            t = torch.zeros(world, dtype=torch.float64)
            t[rank] = steps / dt_local
            dist.all_reduce(t)
            per_rank = [float(v) for v in t.tolist()]
        if rank == 0:
            print(json.dumps({"metric": "dispatch self-test (no GPU work)", "value": aggregate_fps(steps, world, dt), "unit": "frames/s", "n_gpus": world,
                              "steps": args.steps, "steps_timed": steps, "warmup": args.warmup, "data": "none", "per_rank_frames_per_s": per_rank}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import caffe_rtpose_amd as r
    torch.cuda.set_device(local)
    seed = 1
    PREC = {"fp16": r.PREC_FP16, "fp32": r.PREC_FP32, "mixed": r.PREC_MIXED, "f16x3": r.PREC_F16X3}
    mid, W, H, _, _, _, gflop = MODELS[args.model]

    def make_engine(precision, num_scales, scale_gap, batch_frames, in_flight):
        return r.Engine(r.Config(device_id=local, model=mid, net_w=W, net_h=H, num_scales=num_scales, scale_gap=scale_gap, precision=PREC[precision],
                                 frames_in_flight=in_flight, batch_frames=batch_frames, synthetic_seed=seed, split_layers=args.split_layers,
                                 exec_mode=r.EXEC_GRAPH if args.exec_mode == "graph" else r.EXEC_EAGER))

    def device_frames(num_scales, n=8):
        # synthetic frames, resident in HBM before the timed region (u8/256-0.5 like process_and_pad_image)
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        fr = [(torch.randint(0, 256, (num_scales, 3, H, W), generator=g).float() / 256.0 - 0.5).cuda() for _ in range(n)]
        torch.cuda.synchronize()
        return fr

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
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ncal = max(16, 2 * in_flight)
        run(ncal, 0)                      # calibration pass (untimed): frames/s estimate for the step scaling
        torch.cuda.synchronize()
        est = ncal / (time.perf_counter() - t0)
        nsteps = scaled_steps(steps, est * 1.1, min_seconds, dist)
        for attempt in range(4):          # the short calibration pass under-estimates the steady rate: repeat until the region is long enough
            lat.clear()
            host.update(submit=0.0, collect=0.0, frames=0)
            dt, dt_local = timed_region(run, nsteps, 0, dist, torch.cuda.synchronize, "cuda", return_local=True)
            if dt >= min_seconds or attempt == 3:
                break
            nsteps = int(math.ceil(nsteps * min_seconds / dt * 1.15))   # dt is the MAX over ranks: every rank takes the same decision
        return {"dt": dt, "dt_local": dt_local, "steps_timed": nsteps, "fps": aggregate_fps(nsteps, world, dt), "lat": list(lat),
                "host_ms": {"submit_calls": host["submit"] / max(host["frames"], 1) * 1e3, "collect_calls_incl_wait": host["collect"] / max(host["frames"], 1) * 1e3}}

    eng = make_engine(args.precision, args.num_scales, args.scale_gap, args.batch_frames, args.in_flight)
    frames = device_frames(args.num_scales)
    m = measure(eng, lambda i, tag: eng.submit_device(frames[i % len(frames)].data_ptr(), tag=tag), args.steps, args.warmup, args.in_flight,
                args.min_seconds)
    per_rank = None
    if dist is not None:   # every rank's own rate: a straggler shows
        t = torch.zeros(world, dtype=torch.float64, device="cuda")
        t[rank] = m["steps_timed"] / m["dt_local"]
        dist.all_reduce(t)
        per_rank = [float(v) for v in t.tolist()]
    # Roofline pass (separate from the timed region, which runs un-instrumented): the same plan ONE BATCH AT A TIME (submit
    # batch_frames frames, collect them, repeat) with a HIP event pair around every dominant-class launch, recorded on the stream
    # the launch runs on (rtp_kernel_timing; batches are launched eagerly while it is on).  No other frame's kernels are on the
    # chip, so a pair brackets that launch alone inside whole frames (real layer sequence, real L2 state): the quantity a
    # rocprofv3 kernel trace averages (profiles/).  Nothing compares clocks of different XCDs.
    eng.kernel_timing(2)
    nb = max(1, args.batch_frames)
    for b in range(40):
        for j in range(nb):
            eng.submit_device(frames[(b * nb + j) % len(frames)].data_ptr(), tag=b * nb + j)
        for j in range(nb):
            eng.collect()
    dom_ms, dom_n, dom_flops = eng.kernel_timing(-1)
    byp = eng.kernel_timing_by_passes()
    eng.kernel_timing(0)
    stage = eng.last_stage_ms()

    if rank == 0:
        # dominant kernel: the paired 7x7 128->128 convolution (40 of the 92 layers, 50% of all FLOPs).  Average launch duration over
        # EVERY launch of that kernel shape (both instantiations: the plain fp16 one and the fp8-compensated one) inside whole frames,
        # from in-kernel wall-clock stamps.  `achieved` counts ALGORITHMIC flops (2*Cout*Cin*k*k*H*W per image): an error-compensated
        # launch spends 2 (fp16 + fp8 chunks) or 3 (three fp16 passes) pass-times of the matrix pipe on them, reported separately under
        # `executed` (pass-time equivalents: an fp8 chunk takes the time of the fp16 chunk it corrects).
        peak = 157.3e12 if args.precision == "fp32" else 2.5e15
        solo_ms, _ = eng.bench_dominant_conv(iters=200)   # the plain instantiation back to back, alone on the chip (HIP events around 200 launches)
        roof = roofline_block(dom_ms, dom_n, dom_flops, byp, solo_ms, peak,
                              pmc_traffic(args.precision, args.batch_frames, args.num_scales, args.model),
                              pmc_traffic(args.precision, args.batch_frames, args.num_scales, args.model, "_2q"))
        fps = m["fps"]
        whole = {"achieved": fps / world * gflop * 1e9 * args.num_scales / 1e12, "unit": "TFLOP/s",
                 "frac": fps / world * gflop * 1e9 * args.num_scales / peak}
        out = {
            "metric": f"frames/sec (whole node) at {W}x{H} {args.model.upper()} model",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": m["steps_timed"], "steps_requested": args.steps, "steps_timed": m["steps_timed"],
            "warmup": args.warmup, "ms_per_step": m["dt"] / m["steps_timed"] * 1e3, "timed_region_s": m["dt"], "timed_seconds": m["dt"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f16", "data": "synthetic",
            "config": {"precision": args.precision, "precision_note": PREC_NOTE[args.precision],
                       "workload": f"{args.model.upper()} {W}x{H}, {args.num_scales} scale(s), precision mode {args.precision}, conv stack+ImResize+NMS+connect, {args.in_flight} frames in flight/GPU "
                                   f"in batches of {args.batch_frames}, {args.exec_mode} launches, synthetic weights, inputs resident in HBM",
                       "batch_frames": args.batch_frames, "frames_in_flight": args.in_flight, "num_scales": args.num_scales, "exec": args.exec_mode,
                       "parallelism": f"frame-sharded replicas x{world}"},
            "latency_ms": {"p50_pipelined": float(np.percentile(m["lat"], 50) * 1e3), "p95_pipelined": float(np.percentile(m["lat"], 95) * 1e3),
                           "batch_on_device": stage["total"]},
            "host_ms_per_frame": m["host_ms"], "roofline": roof, "conv_stack_whole_frame": whole,
        }
        if per_rank:
            out["per_rank_frames_per_s"] = per_rank
        if world == 1 and not args.no_sub_results:
            out["sub_results"] = sub_results(args, r, eng, make_engine, device_frames, measure, W, H, gflop, np)
        if world == 1 and not (args.no_cpu_baseline and args.no_parity):
            orc_fr = oracle_frames(eng, args.num_scales, args.model, 3 if not args.no_cpu_baseline else 1)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(eng, args.num_scales, args.model, oracle=orc_fr)
            if not args.no_parity:
                try:
                    out["parity"] = parity_report(eng, orc_fr[1][1:], args.model, args.num_scales, args.scale_gap)
                except Exception as ex:  # noqa: BLE001
                    out["parity"] = {"error": str(ex)}
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def sub_results(args, r, eng, make_engine, device_frames, measure, W, H, gflop, np):
    """Extra legs on one GPU (each >= 1 s timed); the headline `value` is not affected."""
    import _synth
    res = {}
    short = dict(steps=50, warmup=10, min_seconds=1.0)
    # (1) BASELINE configs[1] as written: host u8 1280x720 frames -> H2D -> device warp/INTER_AREA/normalise/pad -> same path
    #     (what processFrame + the producer do per frame, rtpose.cpp:322-368, 1127-1133)
    try:
        u8 = [r.synth_frame(1280, 720, i, seed=2) for i in range(8)]
        m = measure(eng, lambda i, tag: eng.submit_frame(u8[i % 8], tag=tag), in_flight=args.in_flight, **short)
        res["host_u8_720p_frames"] = {"value": m["fps"], "unit": "frames/s", "steps_timed": m["steps_timed"], "host_ms_per_frame": m["host_ms"],
                                      "what": "rtp_submit_frame: pageable host u8 1280x720 frame -> pinned copy -> H2D -> device pre-processing -> conv stack -> post, PCIe inclusive"}
    except Exception as ex:  # noqa: BLE001
        res["host_u8_720p_frames"] = {"error": str(ex)}
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
            for it in range(6):
                _, _, n = eng.post_from_lowres(low)
                t.append(eng.last_stage_ms())
            t = t[1:]
            post[name] = {"people": n, "nms_ms": float(np.mean([x["nms"] for x in t])), "connect_ms": float(np.mean([x["connect"] for x in t]))}
        res["postproc_alone"] = post
    except Exception as ex:  # noqa: BLE001
        res["postproc_alone"] = {"error": str(ex)}
    # (3) 3 scales, gap 0.15 (BASELINE configs[2], the north-star target configuration)
    if args.num_scales == 1 and args.model == "coco":
        try:
            e3 = make_engine(args.precision, 3, 0.15, 1, 3)   # one frame (3 images) per launch sequence, 3 frames in flight: the measured optimum
            f3 = device_frames(3)
            m = measure(e3, lambda i, tag: e3.submit_device(f3[i % len(f3)].data_ptr(), tag=tag), in_flight=3, **short)
            peak = 157.3e12 if args.precision == "fp32" else 2.5e15
            res["scales3_gap0.15"] = {"value": m["fps"], "unit": "frames/s", "steps_timed": m["steps_timed"], "p50_ms": float(np.percentile(m["lat"], 50) * 1e3),
                                      "conv_stack_frac_of_peak": m["fps"] * gflop * 3e9 / peak, "batch_frames": 1, "frames_in_flight": 3}
            if not args.no_parity:
                try:
                    res["scales3_gap0.15"]["parity"] = parity_report(e3, oracle_frames(e3, 3, args.model, 0, seed0=7)[1], args.model, 3, 0.15)
                except Exception as ex:  # noqa: BLE001
                    res["scales3_gap0.15"]["parity"] = {"error": str(ex)}
            e3.close()
            del f3
        except Exception as ex:  # noqa: BLE001
            res["scales3_gap0.15"] = {"error": str(ex)}
    # (3b) single-pass fp16 everywhere: the fastest mode, ~2x outside the +-1e-3 tolerance (why it is not the default)
    if args.precision == "mixed" and args.num_scales == 1:
        try:
            e16 = make_engine("fp16", 1, args.scale_gap, args.batch_frames, args.in_flight)
            f1 = device_frames(1)
            m = measure(e16, lambda i, tag: e16.submit_device(f1[i % len(f1)].data_ptr(), tag=tag), in_flight=args.in_flight, **short)
            res["precision_fp16_single_pass"] = {"value": m["fps"], "unit": "frames/s", "steps_timed": m["steps_timed"], "note": PREC_NOTE["fp16"],
                                                 "conv_stack_frac_of_peak": m["fps"] * gflop * 1e9 / 2.5e15}
            e16.close()
        except Exception as ex:  # noqa: BLE001
            res["precision_fp16_single_pass"] = {"error": str(ex)}
    # (4) the exact-f32 MFMA path (reference arithmetic: fp32 throughout)
    if args.precision != "fp32" and args.num_scales == 1:
        try:
            e32 = make_engine("fp32", 1, args.scale_gap, args.batch_frames, args.in_flight)
            f1 = device_frames(1)
            m = measure(e32, lambda i, tag: e32.submit_device(f1[i % len(f1)].data_ptr(), tag=tag), in_flight=args.in_flight, **short)
            res["precision_fp32"] = {"value": m["fps"], "unit": "frames/s", "steps_timed": m["steps_timed"], "conv_stack_frac_of_f32_mfma_peak": m["fps"] * gflop * 1e9 / 157.3e12}
            e32.close()
        except Exception as ex:  # noqa: BLE001
            res["precision_fp32"] = {"error": str(ex)}
    return res


if __name__ == "__main__":
    main()
