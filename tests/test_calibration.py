"""Load-time precision calibration (rtp_calibrate_precision).  The default split set of RTP_PREC_MIXED was chosen on
the synthetic weight seed 1.  Trained weights arrive through CopyTrainedLayersFrom (net.cpp:750-803) with another spectrum; the
calibration measures the set on the weights that are LOADED (mixed vs F16X3 on the device) and widens it until the error is below its
target.  Here: four weight families the set never saw (tests/_families.py), each checked against the fp32 CPU oracle on a frame that
neither the per-layer rescale nor the calibration used — final maps within the north-star tolerance (1e-3 of the map maximum) — and a
run that starts from a deliberately small set, so that the promotion loop itself is exercised."""
import time

import numpy as np
import pytest

import _families
import _oracle as orc
import _synth

pytestmark = pytest.mark.gpu
W, H = 656, 368


def _oracle_maps(e, x):
    net = orc.Net(0)
    for i in range(len(net.convs)):
        net.set_weights(i, *e.get_conv_weights(i))
    return net.forward(x)


def _err(e, x, ref):
    got = e.forward_heatmaps(x)
    assert np.isfinite(got).all()
    return float(np.abs(got - ref).max() / np.abs(ref).max())


def _fps(e, x, frames=80, in_flight=4):
    for t in range(4):
        e.submit(x, tag=t)
    for _ in range(4):
        e.collect()
    t0 = time.perf_counter()
    sub = col = 0
    while col < frames:
        while sub < frames and e.in_flight() < in_flight:
            e.submit(x, tag=sub)
            sub += 1
        e.collect()
        col += 1
    return frames / (time.perf_counter() - t0)


@pytest.mark.parametrize("family", _families.FAMILIES + ("seed5",))
def test_calibrated_split_set_holds_on_weights_it_never_saw(family):
    import caffe_rtpose_amd as r
    e = r.Engine(r.Config(net_w=W, net_h=H, precision=r.PREC_MIXED, frames_in_flight=4, batch_frames=2, synthetic_seed=5 if family == "seed5" else 1))
    if family != "seed5":
        wts, _ = _families.make(e.conv_layers(), family, 7, _synth.random_frame(1, H, W, seed=41))
        for i, (name, *_r) in enumerate(e.conv_layers()):
            e.set_conv_weights(i, *wts[name])
    x = _synth.random_frame(1, H, W, seed=42)      # a frame nobody tuned anything on
    ref = _oracle_maps(e, x)
    err_default = _err(e, x, ref)
    fps_default = _fps(e, x)
    rules, before, after = e.calibrate_precision(nframes=2, target=0.7e-3)
    err_cal = _err(e, x, ref)
    fps_cal = _fps(e, x)
    print(f"\n[calibration {family}] default set: {err_default:.3e} of the map maximum vs the fp32 oracle ({before:.3e} vs F16X3 on the calibration frames), {fps_default:.0f} frames/s"
          f" -> calibrated \"{rules}\": {err_cal:.3e} ({after:.3e}), {fps_cal:.0f} frames/s")
    print("   ", e.calibration_report())
    assert after <= 0.7e-3 or rules == "@f16x3"
    assert err_cal <= 1e-3, f"{family}: {err_cal:.3e} of the map maximum after calibration"
    assert e.split_layers() == ((rules, r.PREC_MIXED) if rules != "@f16x3" else (e.split_layers()[0], r.PREC_F16X3))
    if family in ("seed5", "student_t"):   # He-scaled activations stay inside the fp8 operands' range: promoting un-split groups must be enough
        assert rules != "@f16x3" and ":x" not in rules
    # the re-planned engine is a working pipeline: full batches through submit / collect give the tap's maps' people
    d = e.forward_debug(x)
    for t in range(2):
        e.submit(x, tag=t)
    res = [e.collect() for _ in range(2)]
    assert [q[0] for q in res] == [0, 1] and res[0][1] == d["num_people"] and np.array_equal(res[0][2], d["joints"][:res[0][1]]) and np.array_equal(res[0][2], res[1][2])
    e.close()


def test_calibration_promotes_layer_groups_until_the_target_is_met():
    """Start from a split set that is far too small (only the 1x1 layers): the error is 2-3x the tolerance.  The calibration must promote
    groups — the one that lowers the error most first — until its target holds, and the result must hold against the fp32 oracle."""
    import caffe_rtpose_amd as r
    e = r.Engine(r.Config(net_w=W, net_h=H, precision=r.PREC_MIXED, frames_in_flight=2, batch_frames=1, split_layers="@1x1"))
    x = _synth.random_frame(1, H, W, seed=43)
    ref = _oracle_maps(e, x)
    err0 = _err(e, x, ref)
    rules, before, after = e.calibrate_precision(nframes=2, target=0.7e-3)
    err1 = _err(e, x, ref)
    print(f"\n[calibration from @1x1] {err0:.3e} -> {err1:.3e} with \"{rules}\"\n   ", e.calibration_report())
    assert err0 > 1.2e-3 and before > 0.7e-3            # the starting point really is outside
    assert rules.startswith("@1x1,") and rules.count(",") >= 2 and after <= 0.7e-3 and err1 <= 1e-3
    assert "promote" in e.calibration_report() and "try +" in e.calibration_report()
    e.close()


def test_calibration_at_engine_creation_and_argument_checks():
    import caffe_rtpose_amd as r
    e = r.Engine(r.Config(net_w=320, net_h=176, precision=r.PREC_MIXED, calibrate_frames=1, calibrate_target=0.0))   # target <= 0: the default 0.7e-3
    rep = e.calibration_report()
    assert rep.startswith("target 0.0007") and "final" in rep
    x = _synth.random_frame(1, 176, 320, seed=3)
    assert _err(e, x, _oracle_maps(e, x)) <= 1e-3
    own = np.stack([_synth.random_frame(1, 176, 320, seed=s) for s in (5, 6)])     # the caller's own sample frames
    rules, _before, a = e.calibrate_precision(frames=own, target=0.7e-3)
    assert a <= 0.7e-3 or rules == "@f16x3"
    e.submit(x, tag=9)
    with pytest.raises(r.RtpError):         # idle engines only
        e.calibrate_precision(nframes=1)
    e.collect()
    e.close()
    e16 = r.Engine(r.Config(net_w=320, net_h=176, precision=r.PREC_FP16))
    with pytest.raises(r.RtpError):         # there is no split set to adjust outside RTP_PREC_MIXED
        e16.calibrate_precision(nframes=1)
    e16.close()


@pytest.mark.parametrize("family", ["lognormal_channels", "decaying_spectrum"])
def test_weights_from_a_file_are_checked_by_default(family, tmp_path):
    """calibration helps only if someone calls it — so with `weights_path != NULL` the engine calls it itself.  Two weight
    families that are 4-7x OUTSIDE +-1e-3 with the default split set, written to a .caffemodel and loaded through an UNTOUCHED default
    configuration (net.cpp:750-803 CopyTrainedLayersFrom is where trained weights arrive): inside the tolerance against the fp32 CPU oracle,
    the set reported by rtp_get_split_layers / the calibration report / one line on stderr; calibrate_frames = -1 keeps the default set
    (and stays outside the tolerance: the opt-out is real)."""
    import caffe_rtpose_amd as r
    proto, model = tmp_path / "net.prototxt", tmp_path / "w.caffemodel"
    e0 = r.Engine(r.Config(net_w=W, net_h=H, precision=r.PREC_MIXED, frames_in_flight=2, batch_frames=1))
    wts, _ = _families.make(e0.conv_layers(), family, 7, _synth.random_frame(1, H, W, seed=41))
    for i, (name, *_r) in enumerate(e0.conv_layers()):
        e0.set_conv_weights(i, *wts[name])
    default_rules = e0.split_layers()[0]
    e0.save_caffemodel(model)
    e0.save_prototxt(proto)
    x = _synth.random_frame(1, H, W, seed=42)
    ref = _oracle_maps(e0, x)
    e0.close()
    t0 = time.perf_counter()
    e = r.Engine(r.Config(proto_path=str(proto), weights_path=str(model), net_w=W, net_h=H))    # nothing else set: rtp_config_default
    dt = time.perf_counter() - t0
    rules, mode = e.split_layers()
    err = _err(e, x, ref)
    print(f"\n[default calibration {family}] engine creation {dt:.1f} s; set \"{rules}\" mode {mode}; {err:.3e} of the map maximum vs the fp32 oracle\n   ", e.calibration_report())
    assert e.calibration_report().startswith("target 0.0007; frames 1;") and "final" in e.calibration_report()
    assert mode == r.PREC_F16X3 or (rules != default_rules and rules.startswith(default_rules))
    assert err <= 1e-3
    e.submit(x, tag=1)
    assert e.collect()[0] == 1
    e.close()
    e = r.Engine(r.Config(proto_path=str(proto), weights_path=str(model), net_w=W, net_h=H, calibrate_frames=-1))
    assert e.calibration_report() == "" and e.split_layers() == (default_rules, r.PREC_MIXED)
    assert _err(e, x, ref) > 1e-3
    e.close()
    with pytest.raises(r.RtpError):
        r.Engine(r.Config(net_w=W, net_h=H, calibrate_frames=-2))
