"""Precision of the conv stack in NORTH-STAR units (BASELINE.json: keypoints within +-1 px / +-1e-3 confidence of
the reference's fp32 CPU path), for the exact plans bench.py times: COCO 656x368 at batch_frames 1 and 2, 3 scales
(gap 0.15), MPI 496x368 — each against the CPU oracle's fp32 conv stack (base_conv_layer.cpp:257-280 restated,
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

CONFIGS = {  # name -> (model, W, H, num_scales, scale_gap, batch_frames)
    "coco_1s_b1": (0, 656, 368, 1, 0.3, 1),
    "coco_1s_b2": (0, 656, 368, 1, 0.3, 2),
    "coco_3s": (0, 656, 368, 3, 0.15, 1),
    "mpi_1s_b2": (1, 496, 368, 1, 0.3, 2),
}
TOL = {"fp32": 1e-4, "f16x3": 1e-4, "mixed": 1e-3, "fp16": 3e-3}
_ref_cache = {}


def _prec(r, name):
    return {"fp16": r.PREC_FP16, "fp32": r.PREC_FP32, "mixed": r.PREC_MIXED, "f16x3": r.PREC_F16X3}[name]


def _reference(model, W, H, N, e):
    """fp32 reference maps of the UNSCALED synthetic net for the test frame (cached per geometry)."""
    key = (model, W, H, N)
    if key not in _ref_cache:
        net = orc.Net(model)
        for i in range(len(net.convs)):
            net.set_weights(i, *e.get_conv_weights(i))
        x = _synth.random_frame(N, H, W, seed=3)
        _ref_cache[key] = (x, net.forward(x))
    return _ref_cache[key]


def _match_peaks(a, b, max_peaks):
    """peaks [parts][max_peaks+1][3] of engine (a) and reference (b): pairs within 1 px in x and y."""
    pairs, na, nb = [], 0, 0
    for p in range(a.shape[0]):
        ca = a[p, 1:1 + min(int(a[p, 0, 0]), max_peaks)]
        cb = b[p, 1:1 + min(int(b[p, 0, 0]), max_peaks)]
        na += len(ca)
        nb += len(cb)
        used = set()
        for i in range(len(ca)):
            d = np.maximum(np.abs(cb[:, 0] - ca[i, 0]), np.abs(cb[:, 1] - ca[i, 1])) if len(cb) else np.array([])
            for j in np.argsort(d):
                if d[j] > 1.0:
                    break
                if j not in used:
                    used.add(j)
                    pairs.append((ca[i], cb[j]))
                    break
    return pairs, na, nb


@pytest.mark.parametrize("mode,cfg", [(m, c) for m in ("mixed", "fp16") for c in CONFIGS] + [("f16x3", "coco_1s_b1"), ("fp32", "coco_1s_b2")])
def test_final_maps_and_keypoints_within_tolerance(mode, cfg):
    import caffe_rtpose_amd as r
    model, W, H, N, gap, B = CONFIGS[cfg]
    e = r.Engine(r.Config(model=model, net_w=W, net_h=H, num_scales=N, scale_gap=gap, precision=_prec(r, mode), frames_in_flight=B, batch_frames=B))
    x, ref = _reference(model, W, H, N, e)
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
    pk_e = e.nms(e.resize(got))
    pk_r = orc.nms(orc.imresize(ref_s, W, H, 1.0, gap)[0], parts, max_peaks, thr)
    pairs, na, nb = _match_peaks(pk_e, pk_r, max_peaks)
    assert nb > 50 and len(pairs) >= 0.9 * min(na, nb), f"only {len(pairs)} of {na}/{nb} peaks matched within 1 px"
    dscore = max(abs(float(a[2]) - float(b[2])) for a, b in pairs) / norm
    print(f"[{mode} {cfg}] {len(pairs)} of {na} (engine) / {nb} (reference) peaks matched within 1 px, max |d score| {dscore:.3e}")
    assert dscore <= TOL[mode]
    # the batch plan is what was checked: B frames in flight give the same maps as the frame alone
    if B > 1:
        for t in range(B):
            e.submit(x, tag=t)
        res = [e.collect() for _ in range(B)]
        assert all(np.array_equal(res[0][2], q[2]) for q in res[1:])
    e.close()
