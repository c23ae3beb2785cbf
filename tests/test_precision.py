"""Precision of the conv stack in NORTH-STAR units (BASELINE.json: keypoints within +-1 px / +-1e-3 confidence of
the reference's fp32 CPU path), for the exact plans bench.py times: COCO 656x368 at batch_frames 1 and 2, 3 scales
(gap 0.15) at batch_frames 1 (the 3-scale sub-result) and 2, MPI 496x368 at batch_frames 1, 2 and 5 (bench.py --model mpi) — each against the CPU oracle's fp32 conv stack (base_conv_layer.cpp:257-280 restated,
pinned on the reference's KATs) on the same synthetic weights and frame.

Units: the synthetic network's final maps have max |v| ~ 5; real confidence maps live in [0, 1].  The branch-final
1x1 layers are linear, so their weights and biases are scaled by an exact power of two (engine AND reference) that
brings the maps into that range, and every error below is divided by max|ref| of the scaled maps, i.e. stated for
maps normalised to a maximum of 1.
  * whole maps (57 x 46 x 82 values per scale-image): max |d| <= tol
  * keypoints: ImResize -> Nms on both sides; peaks matched within 1 px must agree to 1e-3 in score
tol: fp32 / f16x3 1e-4, mixed (bench.py's default) 1e-3, pure fp16 3e-3 (measured 2.0-2.6e-3) — the pure fp16 path is OUTSIDE the
north-star tolerance by about 2x, which is why it is not the default."""
import numpy as np
import pytest

import _oracle as orc
import _synth

pytestmark = pytest.mark.gpu

SEEDS = {"coco_1s_b2_wseed5": 5, "coco_3s_wseed9": 9}   # other synthetic weight sets (rtp_config.synthetic_seed; default 1) and other frames
CONFIGS = {  # name -> (model, W, H, num_scales, scale_gap, batch_frames)
    "coco_1s_b1": (0, 656, 368, 1, 0.3, 1),
    "coco_1s_b2": (0, 656, 368, 1, 0.3, 2),
    "coco_3s": (0, 656, 368, 3, 0.15, 1),
    "coco_3s_b2": (0, 656, 368, 3, 0.15, 2),   # the plan bench.py's 3-scale sub-result times (6 images per launch: other tiles than 3)
    "mpi_1s_b1": (1, 496, 368, 1, 0.3, 1),
    "mpi_1s_b2": (1, 496, 368, 1, 0.3, 2),
    "mpi_1s_b5": (1, 496, 368, 1, 0.3, 5),     # bench.py --model mpi: 240 workgroups of 128x128 tiles per 1/8-resolution launch
    "coco_1s_b2_wseed5": (0, 656, 368, 1, 0.3, 2),   # the split set was chosen on weight seed 1: the tolerance must hold on weights it never saw
    "coco_3s_wseed9": (0, 656, 368, 3, 0.15, 1),
    "coco_2s_start080": (0, 656, 368, 2, 0.15, 1),   # round 5: --start_scale 0.8 (START below): scales 0.8 and 0.65, both cropped by ImResize
}
START = {"coco_2s_start080": 0.8}   # rtp_config.start_scale (rtpose.cpp:68); 1.0 elsewhere
TOL = {"fp32": 1e-4, "f16x3": 1e-4, "mixed": 1e-3, "fp16": 3e-3}
_ref_cache = {}


def _prec(r, name):
    return {"fp16": r.PREC_FP16, "fp32": r.PREC_FP32, "mixed": r.PREC_MIXED, "f16x3": r.PREC_F16X3}[name]


def _reference(model, W, H, N, e, wseed=1):
    """fp32 reference maps of the UNSCALED synthetic net for the test frame (cached per geometry and weight seed)."""
    key = (model, W, H, N, wseed)
    if key not in _ref_cache:
        net = orc.Net(model)
        for i in range(len(net.convs)):
            net.set_weights(i, *e.get_conv_weights(i))
        x = _synth.random_frame(N, H, W, seed=3 if wseed == 1 else 100 + wseed)
        _ref_cache[key] = (x, net.forward(x))
    return _ref_cache[key]


def _match_peaks(a, b, max_peaks):
    """peaks [parts][max_peaks+1][3] of engine (a) and reference (b): pairs within 1 px in x and y, and what stayed unmatched
    on either side as (part, x, y, score)."""
    pairs, na, nb, lone_a, lone_b = [], 0, 0, [], []
    for p in range(a.shape[0]):
        ca = a[p, 1:1 + min(int(a[p, 0, 0]), max_peaks)]
        cb = b[p, 1:1 + min(int(b[p, 0, 0]), max_peaks)]
        na += len(ca)
        nb += len(cb)
        used = set()
        for i in range(len(ca)):
            d = np.maximum(np.abs(cb[:, 0] - ca[i, 0]), np.abs(cb[:, 1] - ca[i, 1])) if len(cb) else np.array([])
            hit = False
            for j in np.argsort(d):
                if d[j] > 1.0:
                    break
                if j not in used:
                    used.add(j)
                    pairs.append((ca[i], cb[j]))
                    hit = True
                    break
            if not hit:
                lone_a.append((p, *[float(v) for v in ca[i]]))
        lone_b += [(p, *[float(v) for v in cb[j]]) for j in range(len(cb)) if j not in used]
    return pairs, na, nb, lone_a, lone_b


def _explain(lone, res_here, res_other, thr, norm, who):
    """Why a peak exists on one side only: its margin to the NMS threshold and to its largest 8-neighbour on BOTH maps (a maximum
    whose margin is below the map error flips legitimately: nms_layer.cu:15-46 compares with strict >)."""
    for p, x, y, sc in lone:
        xi, yi = int(round(x)), int(round(y))
        out = []
        for m in (res_here, res_other):
            best = None
            for yy in range(max(yi - 1, 1), min(yi + 2, m.shape[1] - 1)):      # the integer maximum is within 1 px of the centroid
                for xx in range(max(xi - 1, 1), min(xi + 2, m.shape[2] - 1)):
                    v = m[p, yy, xx]
                    nb8 = max(m[p, yy + dy, xx + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy)
                    if best is None or v > best[0]:
                        best = (float(v), float(v - nb8), float(v - thr))
            out.append(best)
        print(f"    {who}-only peak part {p} at ({x:.2f}, {y:.2f}) score {sc / norm:.4f}: here value-max(nb8) {out[0][1] / norm:+.2e}, value-thr {out[0][2] / norm:+.2e};"
              f" other side {out[1][1] / norm:+.2e}, {out[1][2] / norm:+.2e}")


@pytest.mark.parametrize("mode,cfg", [(m, c) for m in ("mixed", "fp16") for c in CONFIGS if m == "mixed" or (c not in SEEDS and c not in START)] +
                         [("f16x3", "coco_1s_b1"), ("fp32", "coco_1s_b2")])
def test_final_maps_and_keypoints_within_tolerance(mode, cfg):
    import caffe_rtpose_amd as r
    model, W, H, N, gap, B = CONFIGS[cfg]
    wseed = SEEDS.get(cfg, 1)
    start = START.get(cfg, 1.0)
    e = r.Engine(r.Config(model=model, net_w=W, net_h=H, num_scales=N, start_scale=start, scale_gap=gap, precision=_prec(r, mode), frames_in_flight=B,
                          batch_frames=B, synthetic_seed=wseed))
    x, ref = _reference(model, W, H, N, e, wseed)
    # bring the maps into the range real confidences live in: exact power-of-two scaling of the linear branch-final layers
    s = float(2.0 ** -np.ceil(np.log2(np.abs(ref).max())))
    last = [i for i, (nm, *_rest) in enumerate(e.conv_layers()) if nm.startswith(("Mconv7_stage6_L", ))]
    assert len(last) == 2
    for i in last:
        w, b = e.get_conv_weights(i)
        e.set_conv_weights(i, w * s, b * s)
    ref_s = ref * np.float32(s)                       # what the reference computes with the scaled layers (exact scaling)
    got = e.forward_heatmaps(x)
    norm = float(np.abs(ref_s).max())
    assert 0.5 <= norm <= 1.0
    err = np.abs(got - ref_s) / norm
    print(f"\n[{mode} {cfg}] final maps normalised to max 1: max err {err.max():.3e}  rms {np.sqrt((err ** 2).mean()):.3e}")
    assert err.max() <= TOL[mode], f"{mode} {cfg}: map error {err.max():.3e} > {TOL[mode]:.1e}"
    # keypoints: both sides through ImResize + Nms (bit-exact kernels), threshold on the scaled maps
    parts, max_peaks = e.num_parts, e.max_peaks
    thr = e.get_thresholds()["nms_threshold"]
    res_e = e.resize(got)
    res_r = orc.imresize(ref_s, W, H, start, gap)[0]
    pk_e = e.nms(res_e)
    pk_r = orc.nms(res_r, parts, max_peaks, thr)
    pairs, na, nb, lone_e, lone_r = _match_peaks(pk_e, pk_r, max_peaks)
    dscore = max(abs(float(a[2]) - float(b[2])) for a, b in pairs) / norm
    print(f"[{mode} {cfg}] {len(pairs)} of {na} (engine) / {nb} (reference) peaks matched within 1 px, max |d score| {dscore:.3e}")
    _explain(lone_e, res_e, res_r, thr, norm, "engine")
    _explain(lone_r, res_r, res_e, thr, norm, "reference")
    # north_star: keypoints match.  The tolerance modes must reproduce >= 99 % of the peak set (what is left are maxima whose margin
    # to a neighbour / the threshold is below the map tolerance, printed above); pure fp16 is outside the tolerance and only reported
    # (at most 1 % of the peaks unmatched — or 3 where a model yields few peaks: MPI's 0.2 threshold leaves 80, one near-tie is 1.25 %)
    allowed = max(3, int((0.01 if mode != "fp16" else 0.05) * max(na, nb)))
    assert nb > 50 and max(na, nb) - len(pairs) <= allowed, f"only {len(pairs)} of {na}/{nb} peaks matched within 1 px"
    assert dscore <= TOL[mode]
    # the batch plan is what was checked: B frames in flight give the same maps as the frame alone
    if B > 1:
        for t in range(B):
            e.submit(x, tag=t)
        res = [e.collect() for _ in range(B)]
        assert all(np.array_equal(res[0][2], q[2]) for q in res[1:])
    e.close()


def _bench_module():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rtp_bench", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("cfg", ["coco_1s_b2", "coco_3s_b2", "mpi_1s_b5", "coco_2s_start080"])
def test_people_level_parity_of_the_benched_mode(cfg):
    """SURVEY section 7 / BASELINE.md section 3 on the configurations bench.py quotes: the engine's joints in the DEFAULT (mixed) mode
    through rtp_submit / rtp_collect vs the full fp32 oracle chain (conv -> ImResize -> Nms -> connectLimbs*), compared as sets of
    people (a person as in rtpose.cpp:1051-1073) — the same function that writes bench.py's `parity` dict.

    What must hold: no joint that is the same maximum on both sides outside +-1 px / +-1e-3; EVERY structural
    difference traced to a decision whose reference-side margin is below twice the measured deviation (tests/_explain.py: NMS compares,
    PAF samples against their threshold, rounding boundaries, order inversions) — none unexplained; and on maps that look like poses
    (planted people + the engine's measured deviation field) the same people, identical within one net pixel."""
    import caffe_rtpose_amd as r
    bench = _bench_module()
    model, W, H, N, gap, B = CONFIGS[cfg]
    start = START.get(cfg, 1.0)
    e = r.Engine(r.Config(model=model, net_w=W, net_h=H, num_scales=N, start_scale=start, scale_gap=gap, precision=r.PREC_MIXED, frames_in_flight=2 * B,
                          batch_frames=B))
    x, ref = _reference(model, W, H, N, e)
    rep = bench.parity_report(e, [(x, ref, 0.0)], "coco" if model == 0 else "mpi", N, gap, start_scale=start)
    print(f"\n[parity {cfg}] " + ", ".join(f"{k}: {v}" for k, v in rep.items() if k not in ("structured", "units", "reference")))
    print(f"[parity {cfg}] structured: {rep['structured']}")
    assert rep["verdict"].startswith(("pass", "numeric pass")), rep["verdict"]
    assert rep["numeric_out_of_tol"] == 0 and rep["max_dc"] <= 1e-3 and rep["max_dx_px"] <= 1.0 and rep["max_dy_px"] <= 1.0   # (no longer tautologies: taken over every paired joint)
    assert rep["map_max_err"] <= 1e-3 and rep["post_on_engine_maps_bit_exact"]
    assert rep["explain"]["unexplained"] == 0, rep["explain"]["unexplained_detail"]
    assert rep["structural_explained"] == rep["joints_structural"]
    # per-joint proof: the reference chain replayed with ONLY the near-tie decisions forced to the engine's outcome yields the engine's people
    assert rep["replay_identical"] and rep["explain"]["replay_unexplained"] == 0, rep["explain"]["replay_unexplained_detail"]
    assert rep["explain"]["replay_worst_margin_over_allowance"] < 0.8
    assert rep["joints_structural"] == 0 or sum(rep["explain"]["root_flips"].values()) > 0
    assert rep["explain"]["worst_margin_over_allowance"] < 0.8    # the flips are near-ties with room to spare, not decisions at the edge of the allowance
    # The random-weight network's maps are noise: hundreds of maxima, some of them near-ties.  Each flip re-numbers the raster-ordered
    # peaks of its part and re-routes connectLimbs' greedy picks, so a few PEOPLE differ structurally although every corresponding joint
    # agrees to a few hundredths of a pixel; the floor below only guards against a silent collapse of the comparison itself.
    # (measured in round 4: 61 of 77 / 38 of 47 / 13 of 13 people, 324 of 348 / 144 of 158 / 45 of 45 joints)
    # round 5, --start_scale 0.8 at 2 scales: 20 of 30 people, 95 of 111 joints (fewer, larger maxima per map: one flip moves more)
    fj, fp = (0.8, 0.6) if cfg in START else (0.85, 0.7)
    assert rep["people_ref"] > (20 if model == 0 else 5) and rep["joints_matched"] >= fj * rep["joints_ref"]
    assert rep["people_matched"] >= fp * rep["people_ref"]
    st = rep["structured"]
    assert st["verdict"].startswith("pass"), st
    assert all(c["identical_within_one_net_pixel"] and c["numeric_out_of_tol"] == 0 and c["people_ref"] == c["people_engine"] >= 1 for c in st["cases"].values())
    e.close()


def test_roofline_timing_is_plausible_and_matches_the_step_profile():
    """bench.py's roofline comes from HIP event pairs around every dominant-class launch (rtp_kernel_timing).  It must be a
    plausible fraction of the fp16 MFMA peak on ANY box (the round-2 driver run printed 9.5e-12 from cross-XCD clock stamps),
    and the per-instantiation averages must agree with the same steps timed alone (rtp_profile_steps) within 25 %."""
    import caffe_rtpose_amd as r
    B = 2
    e = r.Engine(r.Config(net_w=656, net_h=368, precision=r.PREC_MIXED, frames_in_flight=4, batch_frames=B))
    x = _synth.random_frame(1, 368, 656, seed=3)   # host frames: the event pairs bracket the kernel launches, not the H2D copy
    for _ in range(3):   # warm
        for j in range(B):
            e.submit(x, tag=j)
        for j in range(B):
            e.collect()
    e.kernel_timing(2)
    for b in range(10):
        for j in range(B):
            e.submit(x, tag=j)
        for j in range(B):
            e.collect()
    ms, n, flops = e.kernel_timing(-1)
    byp = e.kernel_timing_by_passes()
    e.kernel_timing(0)
    assert n == 10 * 20 and set(byp) == {1, 2}          # 8 plain + 12 fp8-compensated paired 7x7 128->128 launches per batch
    frac = flops / (ms / n * 1e-3) / 2.5e15
    print(f"\n[roofline] {n} launches, {ms / n * 1e3:.1f} us each, {flops / 1e9:.2f} GFLOP -> {frac:.3f} of the fp16 peak; by passes "
          + ", ".join(f"{p}: {t / k * 1e3:.1f} us" for p, (t, k) in byp.items()))
    assert 0.05 < frac < 1.0
    step_ms, step_gf = e.profile_steps(iters=20)
    plan = [ln for ln in r.plan_summary(e.cfg).splitlines() if ln.startswith("step ")]
    assert len(plan) == len(step_ms)
    for passes, tag in ((1, " passes 1 "), (2, " passes 2q ")):
        alone = [t for t, ln in zip(step_ms, plan) if " k 7 cin_p 128 cout 128 " in ln and tag in ln]
        assert alone
        t_alone = sum(alone) / len(alone)
        t_ev = byp[passes][0] / byp[passes][1]
        print(f"[roofline] passes {passes}: events {t_ev * 1e3:.1f} us, step alone {t_alone * 1e3:.1f} us")
        # (an event pair also brackets the dispatch of its launch, ~3-5 us that back-to-back launches overlap: measured 30.5 vs 25.2 us
        # and 50.6 vs 43.7 us; 25 % of the step time + that dispatch)
        assert abs(t_ev - t_alone) <= 0.25 * t_alone + 0.004
    # the graphs captured at creation still serve the next batch after the timing pass
    for j in range(B):
        e.submit(x, tag=j)
    assert [e.collect()[0] for _ in range(B)] == [0, 1]
    e.close()


def test_per_step_timing_and_busy_probe_feed_the_bench_line():
    """rtp_kernel_timing(3) brackets EVERY step of a full batch (bench.py's `roofline.classes`), rtp_busy_probe reads the per-stream busy spans of
    the pipelined loop at collect time (`gpu_busy`): the step rows line up with rtp_plan_summary, every step was timed once per batch, the class
    rows add up to the model's FLOPs, and the union of the spans covers most of a saturated loop."""
    import caffe_rtpose_amd as r
    bench = _bench_module()
    B = 2
    e = r.Engine(r.Config(net_w=656, net_h=368, precision=r.PREC_MIXED, frames_in_flight=6, batch_frames=B))
    x = _synth.random_frame(1, 368, 656, seed=3)
    for _ in range(2):
        for j in range(B):
            e.submit(x, tag=j)
        for j in range(B):
            e.collect()
    e.kernel_timing(3)
    nb = 8
    for b in range(nb):
        for j in range(B):
            e.submit(x, tag=j)
        for j in range(B):
            e.collect()
    ms, n, flops = e.kernel_timing(-1)
    steps = e.kernel_timing_steps()
    e.kernel_timing(0)
    plan = [ln for ln in r.plan_summary(e.cfg).splitlines() if ln.startswith("step ")]
    assert len(steps) == len(plan) and n == nb * 20
    timed = [(t, k) for (t, k), ln in zip(steps, plan) if not ln.startswith("step pack")]
    assert all(k == nb and t > 0 for t, k in timed), [(ln[:40], k) for (t, k), ln in zip(steps, plan) if k != nb]
    classes, total = bench.kernel_classes(plan, e.conv_layers(), steps, 656, 368, B, 2.5e15)
    fl = sum(c["tflops"] * 1e12 * c["ms_per_batch"] * 1e-3 for c in classes.values())
    assert abs(fl / (B * 484.634e9) - 1) < 1e-3 and 1.5 < total < 4.0              # ms of launches per batch of 2 (1.97 back to back; event pairs add their dispatch)
    dom = [c for k, c in classes.items() if k.startswith("dominant")]
    assert len(dom) == 2 and abs(sum(c["ms_per_batch"] * 1 for c in dom) * nb - ms) < 0.05 * ms   # the dominant class's rows ARE rtp_kernel_timing's totals
    print("\n[classes] " + "; ".join(f"{k}: {c['us_per_launch']:.1f} us {c['tflops']:.0f} TF" for k, c in classes.items()))
    # busy probe over a pipelined loop
    assert len(e.busy_probe(1)) == 0
    sub = col = 0
    while col < 120:
        while sub < 120 and e.in_flight() < 6:
            e.submit(x, tag=sub)
            sub += 1
        e.collect()
        col += 1
    spans = e.busy_probe(-1)
    e.busy_probe(0)
    assert len(spans) >= 120 + 50 and set(np.unique(spans[:, 0])) == {0.0, 1.0} and (spans[:, 2] > spans[:, 1]).all() and (spans[:, 1] >= 0).all()
    acc = bench.busy_account(spans)
    print(f"[busy] {acc['busy_frac']:.4f} busy, conv streams {acc['conv_streams_busy_frac']:.3f}, {acc['conv_stacks_concurrent_avg']:.2f} stacks in flight, post chains {acc['post_chains_busy_frac']:.3f}")
    assert 0.9 < acc["busy_frac"] <= 1.0 and acc["conv_stacks_concurrent_avg"] > 1.0
    assert e.probe_dropped() == {"timing_pairs": 0, "busy_spans": 0, "busy_graph_frames": 0}   # nothing above was truncated (rtp_probe_dropped)
    e.submit(x, tag=1)
    with pytest.raises(r.RtpError):       # idle engines only
        e.busy_probe(1)
    e.collect()
    # kernel residency stamps (rtp_stamp_probe): every launch of the pipelined loop leaves {first workgroup start, last workgroup end};
    # results must not depend on the probe (same joints), the stamps must be ordered inside a batch, and the account must be a fraction < 1
    e.submit(x, tag=5)
    _, n_ref, j_ref = e.collect()
    assert len(e.stamp_probe(1)) == 0
    sub = col = 0
    while col < 120:
        while sub < 120 and e.in_flight() < 6:
            e.submit(x, tag=sub)
            sub += 1
        _, n_got, j_got = e.collect()
        assert n_got == n_ref and np.array_equal(j_got, j_ref)
        col += 1
    st = e.stamp_probe(-1)
    e.stamp_probe(0)
    nsteps = len(plan)
    conv = st[st[:, 0] < 64]
    assert len(conv) >= (120 // B - 4) * (nsteps - 1) and (st[:, 2] > st[:, 1]).all()
    assert {64.0, 65.0, 66.0, 67.0, 68.0} <= set(np.unique(st[:, 0]))       # strip, write, pairs, match, assemble of frame 0 of the batches
    d = st[:, 2] - st[:, 1]
    assert 5.0 < np.median(d[st[:, 0] < 64]) < 200.0                          # microseconds: a conv launch lives tens of microseconds
    acc2 = bench.stamp_account(st)
    print(f"[stamps] resident {acc2['resident_frac']:.4f} (conv {acc2['conv_resident_frac']:.3f}, post {acc2['post_resident_frac']:.3f}), "
          f"{acc2['kernels_resident_avg']:.2f} kernels resident on average, hist {acc2['kernels_resident_hist']}")
    assert 0.5 < acc2["resident_frac"] < 1.0 and acc2["idle_frac"] > 0.0
    e.submit(x, tag=6)                     # probe off again: graphs re-captured without slots, same results
    _, n_got, j_got = e.collect()
    assert n_got == n_ref and np.array_equal(j_got, j_ref)
    e.close()
