"""bench.py's host-side bookkeeping that needs no GPU: the per-kernel-class rows of `roofline.classes` (from per-step event timings and
rtp_plan_summary's step lines) and the union-of-spans account behind `gpu_busy`."""
import importlib.util
import json
import os

import numpy as np

import _oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("rtp_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_kernel_classes_cover_every_step_and_sum_to_the_models_flops():
    import caffe_rtpose_amd as r
    b = _bench()
    cfg = r.Config(net_w=656, net_h=368, precision=r.PREC_MIXED, frames_in_flight=7, batch_frames=2)
    plan = [ln for ln in r.plan_summary(cfg).splitlines() if ln.startswith("step ")]
    layers = orc.Net(0).convs
    times = [(0.050 * 40, 40)] * len(plan)            # 50 us per launch, 40 launches each
    classes, total = b.kernel_classes(plan, layers, times, 656, 368, 2, 2.5e15)
    assert abs(total - 0.050 * len(plan)) < 1e-9
    assert sum(c["steps_per_batch"] for c in classes.values()) == len(plan)
    assert classes["dominant 7x7 128->128 pair plain"]["steps_per_batch"] == 8 and classes["dominant 7x7 128->128 pair 2q"]["steps_per_batch"] == 12
    assert classes["stage-entry 7x7 185->128 pair plain"]["steps_per_batch"] == 2 and classes["stage-entry 7x7 185->128 pair 2q"]["steps_per_batch"] == 3
    assert classes["branch tails 1x1->1x1 (conv_pw2)"]["steps_per_batch"] == 6 and classes["conv1_1 (conv_first)"]["steps_per_batch"] == 1
    # algorithmic FLOPs: every class's TFLOP/s x its time adds up to the model's 484.634 GFLOP per image x 2 images
    flops = sum(c["tflops"] * 1e12 * c["ms_per_batch"] * 1e-3 for c in classes.values())
    assert abs(flops / (2 * 484.634e9) - 1) < 1e-4
    dom = classes["dominant 7x7 128->128 pair plain"]
    assert abs(dom["tflops"] - 24.225775616e9 / 50e-6 / 1e12) < 0.5 and abs(sum(c["share_of_batch"] for c in classes.values()) - 1) < 1e-9
    # MPI at batches of 5: the stage-entry shape is 172 -> 128
    cfg = r.Config(model=r.MODEL_MPI_15, net_w=496, net_h=368, precision=r.PREC_MIXED, frames_in_flight=10, batch_frames=5)
    plan = [ln for ln in r.plan_summary(cfg).splitlines() if ln.startswith("step ")]
    classes, _ = b.kernel_classes(plan, orc.Net(1).convs, [(1.0, 10)] * len(plan), 496, 368, 5, 2.5e15)
    flops = sum(c["tflops"] * 1e12 * c["ms_per_batch"] * 1e-3 for c in classes.values())
    assert abs(flops / (5 * 361.695e9) - 1) < 1e-4 and any(k.startswith("stage-entry 7x7 172->128") for k in classes)


def test_busy_account_is_the_union_of_the_spans():
    b = _bench()
    spans = []
    for i in range(100):      # conv spans back to back with a 0.2 ms gap every 1 ms; post chains inside the next span
        spans.append((0, i * 1.0, i * 1.0 + 0.8))
        spans.append((1, i * 1.0 + 0.85, i * 1.0 + 0.95))
    acc = b.busy_account(np.array(spans, np.float32))
    assert abs(acc["busy_frac"] - 0.9) < 0.01 and abs(acc["idle_frac"] - 0.1) < 0.01
    assert abs(acc["conv_streams_busy_frac"] - 0.8) < 0.01 and abs(acc["post_chains_busy_frac"] - 0.1) < 0.01
    two = np.array([(0, i * 0.5, i * 0.5 + 0.9) for i in range(100)], np.float32)   # two stacks at a time
    acc = b.busy_account(two)
    assert acc["busy_frac"] > 0.999 and 1.7 < acc["conv_stacks_concurrent_avg"] < 1.9
    assert b.busy_account(np.zeros((3, 3), np.float32)) is None


def test_the_bench_line_is_compact_and_keeps_the_contract_keys():
    """the driver parses ONE line out of an 8 KB stdout tail.  The round-5 record (27.7 KB as printed then) through
    compact_line() must fit, parse, and still carry the contract's keys with one copy of the per-class rows."""
    import json
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_style_steps20.json")))
    line = b.compact_line(full, ["bench_detail.json"])
    assert "\n" not in line and len(line) <= b.LINE_LIMIT
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity", "summary"):
        assert k in d, k
    assert abs(d["value"] - full["value"]) < 1e-5 and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert len(d["config"]["workload"]) <= 300 and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] <= 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and "traffic" in roof
    assert not [k for k in roof if k.startswith(("cls_", "leg_"))]
    assert len(roof["classes"]) >= 8 and all(len(v) == 3 for v in roof["classes"].values())
    assert "classes_us_tflops" not in d["summary"] and "sub_results" not in d
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] in ("port", "reference")
    assert d["parity"]["replay_identical"] is True and d["parity"]["numeric_out_of_tol"] == 0
    # a pathological record (long strings everywhere) still fits
    fat = json.loads(json.dumps(full))
    fat["config"]["workload"] = "x" * 5000
    fat["cpu_baseline"]["sample"] = "y" * 5000
    fat["roofline"]["classes"] = {f"class {i} " + "z" * 40: v for i, v in enumerate(list(full["roofline"]["classes"].values()) * 4)}
    assert len(b.compact_line(fat)) <= b.LINE_LIMIT


def test_stamp_account_counts_the_wall_with_no_kernel_resident():
    """bench.py's `idle_frac_kernel_stamps`: union of device-side residency spans {slot, start_us, end_us}."""
    b = _bench()
    spans = []
    for i in range(200):      # a conv kernel 80 us of every 100, a post kernel inside the gap for 5 us, one overlapping the conv kernel
        spans.append((3, i * 100.0, i * 100.0 + 80.0))
        spans.append((64, i * 100.0 + 85.0, i * 100.0 + 90.0))
        spans.append((66, i * 100.0 + 40.0, i * 100.0 + 60.0))
    acc = b.stamp_account(np.array(spans, np.float32))
    assert abs(acc["resident_frac"] - 0.85) < 0.01 and abs(acc["idle_frac"] - 0.15) < 0.01
    assert abs(acc["conv_resident_frac"] - 0.80) < 0.01 and abs(acc["post_resident_frac"] - 0.25) < 0.01
    assert abs(acc["kernels_resident_avg"] - 1.05) < 0.02 and abs(acc["kernels_resident_hist"]["2"] - 0.20) < 0.01
    assert b.stamp_account(np.zeros((3, 3), np.float32)) is None


def test_stamp_dominant_reads_the_dominant_launches_from_the_device_stamps():
    """bench.py `roofline.device_stamps`: residency of the 7x7 128->128 pair launches (slot = plan step) per MFMA pass count, and the roofline
    fraction they give — the tracer-free counterpart of the rocprofv3 kernel stats under profiles/."""
    b = _bench()
    plan = ["step first conv1_1 k 3 cin 3 cout 64 relu 1 passes 1 wgs 736",
            "step conv Mconv1_stage2_L1 + Mconv1_stage2_L2 k 7 cin_p 192 cout 128 coutp 128 relu 1 tile 128x128 rowb 128 passes 1 impl ring wgs 124 dsts 1 lowres 0",
            "step conv Mconv2_stage2_L1 + Mconv2_stage2_L2 k 7 cin_p 128 cout 128 coutp 128 relu 1 tile 128x64 rowb 256 passes 1 impl ring wgs 248 dsts 1 lowres 0",
            "step conv Mconv2_stage4_L1 + Mconv2_stage4_L2 k 7 cin_p 128 cout 128 coutp 128 relu 1 tile 128x64 rowb 256 passes 2q impl ring wgs 248 dsts 1 lowres 0",
            "step pw2 Mconv6_stage2_L1 + Mconv6_stage2_L2 -> Mconv7_stage2_L1 + Mconv7_stage2_L2 k 1 cin_p 128 mid 128 cout 38 passes 3aw/3aw tile 64 wgs 248 lowres 0"]
    spans = []
    for i in range(10):
        t = i * 1000.0
        spans += [(0, t, t + 30), (1, t + 40, t + 90), (2, t + 100, t + 125), (3, t + 130, t + 175), (4, t + 180, t + 190), (64, t + 200, t + 250)]
    spans.append((2, 5.0, 5.0))       # an empty stamp pair is skipped
    d = b.stamp_dominant(np.array(spans, np.float32), plan, 24.225775616e9, 2.5e15)
    assert d["launches"] == 20 and abs(d["us_by_mfma_passes"]["1"] - 25.0) < 1e-3 and abs(d["us_by_mfma_passes"]["2"] - 45.0) < 1e-3
    assert abs(d["us_per_launch"] - 35.0) < 1e-3 and abs(d["frac"] - 24.225775616e9 / 35e-6 / 2.5e15) < 1e-6
    assert b.stamp_dominant(np.zeros((0, 3), np.float32), plan, 1.0, 1.0) is None
    out = {"metric": "m", "value": 1.0, "roofline": {"frac": 0.2, "device_stamps": d}}
    line = json.loads(b.compact_line(out))
    assert line["roofline"]["device_stamps"]["us_by_mfma_passes"] == {"1": 25.0, "2": 45.0} and abs(line["roofline"]["device_stamps"]["frac"] - d["frac"]) < 1e-3
