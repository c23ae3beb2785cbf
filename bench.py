#!/usr/bin/env python
"""bench.py — frames/sec of the rtpose hot path on MI355X (BASELINE.json metric).

A "step" = one frame: net input (num_scales x 3 x 368 x 656 fp32, already resident in HBM) ->
conv stack -> ImResize -> Nms -> connectLimbsCOCO -> joints on the host.  Workload = BASELINE.json
configs[1] ("COCO model 656x368, 1 scale, 1xMI355X") unless --num_scales says otherwise.
Multi-GPU: one process per GPU, frames sharded (replicas, no data-path collective) — weak scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


MODELS = {  # name -> (model id, net_w, net_h, parts, max_peaks, nms threshold, conv GFLOP per scale-image)
    "coco": (0, 656, 368, 18, 64, 0.05, 484.634),
    "mpi": (1, 496, 368, 15, 20, 0.2, 361.695),
}


def cpu_baseline(eng, num_scales, model="coco"):
    """The CPU oracle (a port: the reference itself cannot be built here) timed on this host's cores
    on a bounded sample: ONE frame through conv stack + ImResize + NMS + connect."""
    import numpy as np
    import _oracle as orc
    import _synth
    mid, W, H, parts, max_peaks, thr, _ = MODELS[model]
    net = orc.Net(mid)
    for i in range(len(net.convs)):
        w, b = eng.get_conv_weights(i)
        net.set_weights(i, w, b)
    x = _synth.random_frame(num_scales, H, W, seed=1)
    t0 = time.time()
    low = net.forward(x)
    t1 = time.time()
    res = orc.imresize(low, W, H, 1.0, 0.3)[0]
    peaks = orc.nms(res, parts, max_peaks, thr)
    orc.connect(mid, res, peaks, max_peaks, W, H, 1280, 720)
    t2 = time.time()
    return {"value": 1.0 / (t2 - t0), "unit": "frames/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"1 frame, {num_scales} scale(s), {W}x{H} {model.upper()}: conv stack {t1 - t0:.2f}s + postproc {t2 - t1:.2f}s, OpenMP fp32"}


PMC_B2 = (7196, 2895)  # (FETCH_SIZE KiB, WRITE_SIZE KiB) per dominant launch at batch_frames=2, from profiles/r01_dominant_conv_pmc_b2.txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--num_scales", type=int, default=1)
    ap.add_argument("--scale_gap", type=float, default=0.3)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--in_flight", type=int, default=8)
    ap.add_argument("--batch_frames", type=int, default=2, help="frames whose conv stacks share one launch sequence (1 = the reference's one frame per Forward)")
    ap.add_argument("--model", default="coco", choices=["coco", "mpi"], help="coco = BASELINE configs[1..3] (656x368); mpi = configs[4] (15 parts, 496x368)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import caffe_rtpose_amd as r

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)

    seed = 1
    prec = r.PREC_FP16 if args.precision == "fp16" else r.PREC_FP32
    mid, W, H, _, _, _, gflop = MODELS[args.model]
    eng = r.Engine(r.Config(device_id=local, model=mid, net_w=W, net_h=H, num_scales=args.num_scales,
                            scale_gap=args.scale_gap, precision=prec, frames_in_flight=args.in_flight, batch_frames=args.batch_frames, synthetic_seed=seed))
    # synthetic frames, resident in HBM before the timed region (u8/256-0.5 like process_and_pad_image)
    nframes = 8
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    frames = [(torch.randint(0, 256, (args.num_scales, 3, H, W), generator=g).float() / 256.0 - 0.5).cuda() for _ in range(nframes)]
    torch.cuda.synchronize()

    lat = []
    host = {"submit": 0.0, "collect": 0.0, "frames": 0}

    def run(nsteps, base_tag):
        sub = col = 0
        people = 0
        t_sub = {}
        while col < nsteps:
            while sub < nsteps and eng.in_flight() < args.in_flight:
                t_sub[sub] = time.perf_counter()
                eng.submit_device(frames[sub % nframes].data_ptr(), tag=base_tag + sub)
                sub += 1
                if base_tag:
                    host["submit"] += time.perf_counter() - t_sub[sub - 1]
            t_c = time.perf_counter()
            tag, n, _ = eng.collect()
            assert tag == base_tag + col
            if base_tag:
                now = time.perf_counter()
                host["collect"] += now - t_c
                host["frames"] += 1
                lat.append(now - t_sub.pop(col))
            people += n
            col += 1
        return people

    from caffe_rtpose_amd.dispatch import timed_region, aggregate_fps

    def run_timed(nsteps, base_tag):
        # HIP events bracket every dominant-kernel launch of the TIMED steps on the frame's own stream
        eng.kernel_timing(1 if base_tag else 0)
        return run(nsteps, base_tag)

    dt = timed_region(run_timed, args.steps, args.warmup, dist, torch.cuda.synchronize, "cuda")
    dom_ms, dom_n, dom_flops = eng.kernel_timing(0)
    stage = eng.last_stage_ms()

    if rank == 0:
        # dominant kernel: the paired 7x7 128->128 convolution (40 of the 92 layers, 50% of all FLOPs);
        # HIP events on the engine's own stream around `iters` back-to-back launches.
        peak = 2.5e15 if args.precision == "fp16" else 157.3e12
        ms = dom_ms / max(dom_n, 1)   # average launch duration inside the timed, pipelined region
        achieved = dom_flops / (ms * 1e-3)
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 FETCH_SIZE/WRITE_SIZE are in KiB;
        # FETCH_SIZE x2 per the guide's gfx950 correction), keyed by (num_scales, batch_frames); None = not collected
        pmc_kib = {(1, 1): (6458, 1886), (1, 2): PMC_B2}
        fw = pmc_kib.get((args.num_scales, args.batch_frames)) if args.model == "coco" else None
        traffic = (2 * fw[0] + fw[1]) * 1024 if fw else None
        roof = {"bound": "mfma", "kernel": "conv_ring_kernel 7x7 128->128 (L1+L2 branch pair)", "achieved": achieved / 1e12, "peak": peak / 1e12,
                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "ms_per_launch": ms, "launches_timed": dom_n,
                "flops_per_launch": dom_flops}
        solo_ms, _ = eng.bench_dominant_conv(iters=200)   # the same kernel alone on the chip (no other frame sharing the CUs)
        roof["solo"] = {"ms_per_launch": solo_ms, "achieved": dom_flops / (solo_ms * 1e-3) / 1e12, "frac": dom_flops / (solo_ms * 1e-3) / peak}
        fps = aggregate_fps(args.steps, world, dt)
        whole = {"achieved": fps / world * gflop * 1e9 * args.num_scales / 1e12, "unit": "TFLOP/s",
                 "frac": fps / world * gflop * 1e9 * args.num_scales / peak}
        out = {
            "metric": f"frames/sec (whole node) at {W}x{H} {args.model.upper()} model",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.precision == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": f"{args.model.upper()} {W}x{H}, {args.num_scales} scale(s), conv stack+ImResize+NMS+connect, {args.in_flight} frames in flight/GPU in batches of {args.batch_frames}, synthetic weights",
                       "batch_frames": args.batch_frames, "frames_in_flight": args.in_flight, "num_scales": args.num_scales,
                       "parallelism": f"frame-sharded replicas x{world}"},
            "latency_ms": {"p50_pipelined": float(np.percentile(lat, 50) * 1e3), "p95_pipelined": float(np.percentile(lat, 95) * 1e3),
                           "single_frame_device": stage["total"]},
            "host_ms_per_frame": {"submit_calls": host["submit"] / max(host["frames"], 1) * 1e3, "collect_calls_incl_wait": host["collect"] / max(host["frames"], 1) * 1e3},
            "stage_ms_last_frame": stage, "roofline": roof, "conv_stack_whole_frame": whole,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(eng, args.num_scales, args.model)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
