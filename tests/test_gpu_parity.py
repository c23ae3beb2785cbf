"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.

Post-processing (ImResize, Nms, connectLimbs*) is BIT-EXACT given identical inputs; the conv
stack is floating point: tolerance stated per test (north_star: keypoints within +-1 px /
+-1e-3 confidence).  /root/reference is never read here.
"""
import os

import numpy as np
import pytest

import _oracle as orc
import _synth

pytestmark = pytest.mark.gpu


def _engine(**kw):
    import caffe_rtpose_amd as r
    return r.Engine(r.Config(**kw))


def _oracle_net_from(engine):
    net = orc.Net(engine.cfg.c.model if not engine.cfg.c.proto_path else (0 if engine.num_parts == 18 else 1))
    layers = engine.conv_layers()
    assert [l[0] for l in layers] == [c[0] for c in net.convs]
    for i in range(len(layers)):
        w, b = engine.get_conv_weights(i)
        net.set_weights(i, w, b)
    return net


def _rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


# ------------------------------------------------------------------------------------------
# conv stack
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,W,H,N", [(0, 64, 48, 1), (0, 160, 96, 2), (1, 96, 64, 1)])
def test_conv_stack_fp32_exact_path(model, W, H, N):
    """fp32 MFMA path vs oracle: same arithmetic class (fp32 products, fp32 accumulate, different
    summation order) -> 2e-4 of the blob's max magnitude at every layer, 1e-3 absolute at the end."""
    import caffe_rtpose_amd as r
    e = _engine(model=model, net_w=W, net_h=H, num_scales=N, precision=r.PREC_FP32, scale_gap=0.25, frames_in_flight=1)
    net = _oracle_net_from(e)
    x = _synth.random_frame(N, H, W, seed=7)
    got = e.forward_heatmaps(x)
    net.forward(x, keep_all=True)
    worst = []
    for name, *_ in net.convs:
        ref = net.blob(name)
        g = e.get_blob(name)
        assert g.shape == ref.shape, name
        worst.append((name, _rel_err(g, ref)))
    bad = [(n, v) for n, v in worst if not v < 2e-4]
    assert not bad, f"first diverging layers: {bad[:5]}"
    for name in ("pool1_stage1", "pool2_stage1", "pool3_stage1", "concat_stage2", "concat_stage6"):
        assert _rel_err(e.get_blob(name), net.blob(name)) < 2e-4, name
    ref = net.blob("concat_stage7")
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())
    e.close()


@pytest.mark.parametrize("model,W,H,N", [(0, 64, 48, 1), (0, 160, 96, 2), (1, 96, 64, 2)])
def test_conv_stack_fp16_path(model, W, H, N):
    """fp16 storage / fp32 accumulate vs the fp32 oracle.  Each layer re-rounds activations to 11
    bits, so the error is a random walk over 52 layers: bound 2e-2 of the blob's max magnitude,
    and the first layer (exact u8/256-0.5 inputs, one rounding of the output) at 2e-3."""
    import caffe_rtpose_amd as r
    e = _engine(model=model, net_w=W, net_h=H, num_scales=N, precision=r.PREC_FP16, scale_gap=0.25, frames_in_flight=1, keep_blobs=1)  # every blob tapped
    net = _oracle_net_from(e)
    x = _synth.random_frame(N, H, W, seed=8)
    got = e.forward_heatmaps(x)
    net.forward(x, keep_all=True)
    assert _rel_err(e.get_blob("conv1_1"), net.blob("conv1_1")) < 2e-3
    errs = [(name, _rel_err(e.get_blob(name), net.blob(name))) for name, *_ in net.convs]
    bad = [(n, v) for n, v in errs if not v < 2e-2]
    assert not bad, f"first diverging layers: {bad[:5]}"
    assert _rel_err(got, net.blob("concat_stage7")) < 2e-2
    e.close()


@pytest.mark.parametrize("prec", ["fp16", "mixed", "f16x3"])
@pytest.mark.parametrize("model,W,H,N,B", [(0, 656, 368, 1, 2), (1, 496, 368, 1, 1), (0, 320, 176, 2, 1), (0, 336, 208, 1, 3), (0, 720, 400, 2, 2), (0, 64, 48, 1, 1)])
def test_pooling_fused_into_the_convolution_epilogue_is_bit_identical(prec, model, W, H, N, B):
    """Default plan: conv1_2 / conv2_2 / conv3_4 pool in their epilogue (2-row tiles, conv_ring.hip POOL) and write only the pooled
    blob.  keep_blobs = 1 runs the same layers with the stand-alone pooling launches (pooling_layer.cpp:140-180 restated in
    aux_kernels.hip, compared with the oracle by test_conv_stack_fp32_exact_path).  Same bytes: pooled blobs (their lo / fp8 parts
    included: the export adds them), every later blob, the low-res maps."""
    import caffe_rtpose_amd as r
    P = {"fp16": r.PREC_FP16, "mixed": r.PREC_MIXED, "f16x3": r.PREC_F16X3}[prec]
    kw = dict(model=model, net_w=W, net_h=H, num_scales=N, precision=P, scale_gap=0.2, frames_in_flight=B, batch_frames=B)
    ef = _engine(**kw)
    ek = _engine(keep_blobs=1, **kw)
    plan = r.plan_summary(ef.cfg)
    nf = plan.count("+pool")   # small resolutions put some trunk layers on tiles without a pooled variant: those keep their pooling launch
    # fusion needs 128-pixel tiles and >= 128 columns at that level
    assert nf >= (3 if W >= 640 else 2 if W >= 480 else 1 if W >= 320 else 0)
    # a layer that is not fused keeps its stand-alone pooling launch; keep_blobs = 1 restores all three
    assert nf + plan.count("step pool") == 3 and r.plan_summary(ek.cfg).count("step pool") == 3
    fused_away = [ln.split()[2] for ln in plan.splitlines() if "+pool" in ln]
    def tiles(pl):   # (layer, tile, chunk bytes) of every convolution launch: the two plans must run the same kernels to be comparable bit for bit
        return [(w[2], w[w.index("tile") + 1], w[w.index("rowb") + 1]) for w in (ln.replace(" +pool", "").split() for ln in pl.splitlines() if ln.startswith("step conv"))]
    same_kernels = tiles(plan) == tiles(r.plan_summary(ek.cfg))   # (the tile model credits tiles that can pool: small plans may differ)
    x = _synth.random_frame(N, H, W, seed=21)
    a, b = ef.forward_heatmaps(x), ek.forward_heatmaps(x)
    if W >= 640:
        assert same_kernels
    eq = np.array_equal if same_kernels else (lambda p, q: bool(np.abs(p - q).max() <= 2e-3 * np.abs(q).max()))
    assert eq(a, b)
    for name in ("pool1_stage1", "pool2_stage1", "pool3_stage1", "conv2_1", "conv3_1", "conv4_4_CPM"):
        assert eq(ef.get_blob(name), ek.get_blob(name)), name
    for name in fused_away:
        with pytest.raises(r.RtpError):
            ef.get_blob(name)
    assert ek.get_blob("conv1_2").shape == (N, 64, H, W)
    if B > 1:   # full batches through submit / collect
        for t in range(B):
            ef.submit(x, tag=t)
            ek.submit(x, tag=t)
        ra, rb = [ef.collect() for _ in range(B)], [ek.collect() for _ in range(B)]
        if same_kernels:
            assert all(np.array_equal(p[2], q[2]) and p[1] == q[1] for p, q in zip(ra, rb))
    ef.close()
    ek.close()


# ------------------------------------------------------------------------------------------
# ImResize — bit exact
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,W,H,N,start,gap", [(0, 656, 368, 1, 1.0, 0.3), (0, 656, 368, 3, 1.0, 0.15),
                                                    (1, 496, 368, 2, 1.0, 0.3), (0, 64, 48, 2, 1.0, 0.25),
                                                    # round 5: --start_scale != 1 moves the crop of EVERY scale (imresize_layer.cu:110-113),
                                                    # portrait / large nets, crops down to 14 x 8 low-res cells
                                                    (0, 656, 368, 1, 0.8, 0.15), (0, 656, 368, 2, 0.8, 0.15), (0, 656, 368, 3, 0.8, 0.15),
                                                    (0, 656, 368, 1, 0.65, 0.25), (0, 656, 368, 3, 0.65, 0.25), (1, 496, 368, 2, 0.65, 0.15),
                                                    (0, 368, 656, 2, 0.8, 0.15), (0, 1312, 736, 2, 0.8, 0.15), (0, 64, 48, 2, 0.7, 0.3)])
def test_resize_bit_exact(model, W, H, N, start, gap):
    e = _engine(model=model, net_w=W, net_h=H, num_scales=N, start_scale=start, scale_gap=gap, frames_in_flight=1)
    low = _synth.smooth_field(N * e.heat_channels, H // 8, W // 8, seed=3).reshape(N, e.heat_channels, H // 8, W // 8)
    got = e.resize(low)
    ref = orc.imresize(low, W, H, start, gap)[0]
    assert np.array_equal(got, ref), f"max diff {np.abs(got - ref).max()}"
    e.close()


def test_set_scales_and_submit_frame_refuse_out_of_contract_scales():
    """ImResizeLayer::SetStartScale / SetScaleGap (imresize_layer.cpp) take anything and the producer CHECKs later (rtpose.cpp:363);
    rtp_set_scales refuses what the producer would refuse — RTP_EINVAL, the engine keeps its old scales — and accepts start_scale < 1."""
    import caffe_rtpose_amd as r
    from test_host_cpu import BAD_GEOMETRY
    for kw in BAD_GEOMETRY:
        with pytest.raises(r.RtpError) as ei:
            _engine(**{**dict(net_w=160, net_h=96, frames_in_flight=1), **kw})
        assert ei.value.code == r.RTP_EINVAL, kw
    e = _engine(net_w=160, net_h=96, num_scales=2, scale_gap=0.25, disp_w=320, disp_h=180, frames_in_flight=1)
    low = _synth.smooth_field(2 * e.heat_channels, 12, 20, seed=5).reshape(2, e.heat_channels, 12, 20)
    img = r.synth_frame(320, 180, 0, seed=2)
    before = e.resize(low)
    x0, _, _ = e.debug_preprocess(img)
    nan = float("nan")
    for s, g in ((1.2, 0.25), (0.2, 0.25), (nan, 0.25), (1.0, nan), (0.0, 0.1), (1.0, 1.0), (-1.0, -0.5), (float("inf"), 0.25)):
        with pytest.raises(r.RtpError) as ei:
            e.set_scales(s, g)
        assert ei.value.code == r.RTP_EINVAL, (s, g)
        assert np.array_equal(e.resize(low), before)          # a refused call changes nothing
    assert np.array_equal(e.debug_preprocess(img)[0], x0)
    e.submit_frame(img, tag=3)
    assert e.collect()[0] == 3
    e.set_scales(0.8, 0.3)                                    # in contract: levels 0.8 and 0.5
    assert np.array_equal(e.resize(low), orc.imresize(low, 160, 96, 0.8, 0.3)[0])
    want_x, _, _ = r.preprocess_frame(img, 320, 180, 160, 96, 2, 0.8, 0.3)
    assert np.array_equal(e.debug_preprocess(img)[0], want_x)
    e.submit_frame(img, tag=4)
    t, n, j = e.collect()
    e.submit(want_x, tag=5)
    t2, n2, j2 = e.collect()
    assert (t, t2) == (4, 5) and n == n2 and np.array_equal(j, j2)
    e.close()


# ------------------------------------------------------------------------------------------
# NMS — bit exact, raster order, max_peaks cap, stale slots
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,W,H", [(0, 656, 368), (1, 496, 368), (0, 64, 48)])
def test_nms_bit_exact_noise(model, W, H):
    """Noise maps: far more than max_peaks maxima per part -> exercises the raster-order cap."""
    e = _engine(model=model, net_w=W, net_h=H, frames_in_flight=1)
    res = _synth.smooth_field(e.heat_channels, H, W, seed=4)
    stale = np.full((e.num_parts, e.max_peaks + 1, 3), -7.0, np.float32)
    got = e.nms(res, stale)
    ref = orc.nms(res, e.num_parts, e.max_peaks, e.get_thresholds()["nms_threshold"], stale)
    assert (ref[:, 0, 0] > e.max_peaks).any() or W < 100
    assert np.array_equal(got, ref)
    e.close()


def test_nms_sparse_and_stale_slots():
    e = _engine(frames_in_flight=1)
    tabs = orc.model_tables(0)
    low, _ = _synth.people_lowres(0, tabs, 3, 46, 82, seed=5)
    res = orc.imresize(low, 656, 368, 1.0, 0.3)[0]
    stale = np.full((18, 65, 3), 123.0, np.float32)
    got = e.nms(res, stale)
    ref = orc.nms(res, 18, 64, 0.05, stale)
    assert np.array_equal(got, ref)
    n = int(ref[0, 0, 0])
    assert 0 < n < 64 and (got[0, n + 1:] == 123.0).all()  # unwritten slots keep the caller's data
    # empty map: every count is 0
    z = np.zeros_like(res)
    got0 = e.nms(z)
    assert (got0[:, 0, 0] == 0).all()
    e.close()


# ------------------------------------------------------------------------------------------
# connect + JSON — bit exact
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,W,H,P", [(0, 656, 368, 1), (0, 656, 368, 5), (0, 656, 368, 20), (1, 496, 368, 4)])
def test_postproc_chain_bit_exact(model, W, H, P):
    import caffe_rtpose_amd as r
    e = _engine(model=model, net_w=W, net_h=H, frames_in_flight=1)
    tabs = orc.model_tables(model)
    thr = e.get_thresholds()
    assert thr == orc.default_thresholds(model)
    low, _ = _synth.people_lowres(model, tabs, P, H // 8, W // 8, seed=20 + P)
    res = e.resize(low)
    ref_res = orc.imresize(low, W, H, 1.0, 0.3)[0]
    assert np.array_equal(res, ref_res)
    peaks = e.nms(res)
    ref_peaks = orc.nms(ref_res, e.num_parts, e.max_peaks, thr["nms_threshold"])
    assert np.array_equal(peaks, ref_peaks)
    n, joints = e.connect(res, peaks)
    rn, rj = orc.connect(model, ref_res, ref_peaks, e.max_peaks, W, H, 1280, 720, thr)
    assert n == rn and n >= 1
    assert np.array_equal(joints[:n], rj[:n])
    assert r.format_json(joints, n, e.num_parts, 1.5) == orc.write_json(rj, rn, e.num_parts, 1.5)
    e.close()


def test_connect_ties_and_saturation():
    """Constant PAF => every candidate ties at the same score: the greedy assignment then depends on
    std::sort's tie order, which the device replica must reproduce.  Noise peaks saturate max_peaks."""
    e = _engine(frames_in_flight=1)
    H, W = 368, 656
    res = np.zeros((57, H, W), np.float32)
    res[19:] = 0.70710677  # every PAF channel: unit vector along (1,1)
    rs = np.random.RandomState(11)
    peaks = np.zeros((18, 65, 3), np.float32)
    for p in range(18):
        n = int(rs.randint(3, 30))
        peaks[p, 0, 0] = n
        base = rs.uniform(20, 200)
        for i in range(1, n + 1):
            # partB is always down-right of partA-ish so that the (1,1) field scores > 0.05
            peaks[p, i] = (base + 9 * i + p * 3, base * 0.5 + 9 * i + p * 3, rs.uniform(0.3, 0.9))
    n, joints = e.connect(res, peaks)
    rn, rj = orc.connect(0, res, peaks, 64, W, H, 1280, 720)
    assert n == rn and np.array_equal(joints[:n], rj[:n])
    # saturated / noise case end to end
    noise = _synth.smooth_field(57, H, W, seed=12)
    pk = e.nms(noise)
    assert np.array_equal(pk, orc.nms(noise, 18, 64, 0.05))
    n2, j2 = e.connect(noise, pk)
    rn2, rj2 = orc.connect(0, noise, pk, 64, W, H, 1280, 720)
    assert n2 == rn2 and np.array_equal(j2[:n2], rj2[:rn2])
    e.close()


def test_connect_empty_and_single_sided():
    e = _engine(frames_in_flight=1)
    res = np.zeros((57, 368, 656), np.float32)
    peaks = np.zeros((18, 65, 3), np.float32)
    n, _ = e.connect(res, peaks)
    assert n == 0
    peaks[1, 0, 0] = 2  # only necks: nB == 0 branches create 1-part rows, none reaches 3 parts
    peaks[1, 1] = (100, 100, 0.9)
    peaks[1, 2] = (300, 120, 0.8)
    n, j = e.connect(res, peaks)
    rn, rj = orc.connect(0, res, peaks, 64, 656, 368, 1280, 720)
    assert n == rn == 0
    e.close()


# ------------------------------------------------------------------------------------------
# whole frame through submit/collect
# ------------------------------------------------------------------------------------------
def test_frame_pipeline_matches_taps_and_oracle_postproc():
    """submit/collect (async, 3 frames in flight) == forward_debug (sync) == oracle post-processing
    applied to the engine's own low-res maps (bit exact)."""
    import caffe_rtpose_amd as r
    W, H = 160, 96
    e = _engine(net_w=W, net_h=H, num_scales=2, scale_gap=0.25, frames_in_flight=3, precision=r.PREC_FP16)
    frames = [_synth.random_frame(2, H, W, seed=100 + i) for i in range(5)]
    dbg = [e.forward_debug(f) for f in frames]
    thr = e.get_thresholds()
    for d in dbg:
        ref_res = orc.imresize(d["lowres"], W, H, 1.0, 0.25)[0]
        assert np.array_equal(d["resized"], ref_res)
        ref_peaks = orc.nms(ref_res, 18, 64, thr["nms_threshold"], d["peaks"])
        assert np.array_equal(d["peaks"], ref_peaks)
        rn, rj = orc.connect(0, ref_res, d["peaks"], 64, W, H, 1280, 720, thr)
        assert rn == d["num_people"] and np.array_equal(rj[:rn], d["joints"][:rn])
    out = {}
    i = 0
    while i < len(frames) or e.in_flight():
        while i < len(frames) and e.in_flight() < 3:
            e.submit(frames[i], tag=1000 + i)
            i += 1
        tag, n, joints = e.collect()
        out[tag] = (n, joints)
    assert sorted(out) == [1000 + k for k in range(5)]
    for k in range(5):
        n, joints = out[1000 + k]
        assert n == dbg[k]["num_people"]
        assert np.array_equal(joints, dbg[k]["joints"][:n])
    with pytest.raises(r.RtpError):
        e.collect()  # nothing in flight -> RTP_EAGAIN, not an abort
    e.close()


def test_determinism_under_load_full_res():
    """The same frame submitted 12 times with 4 frames in flight at 656x368 must give bit-identical
    joints every time, and the same as the synchronous path: any missing wait / barrier in the
    pipelined kernels shows up as run-to-run differences once several frames compete for the CUs."""
    import caffe_rtpose_amd as r
    for prec in (r.PREC_FP16, r.PREC_FP32):
        e = _engine(frames_in_flight=4, precision=prec)
        x = _synth.random_frame(1, 368, 656, seed=77)
        ref = e.forward_debug(x)
        low2 = e.forward_heatmaps(x)
        assert np.array_equal(low2, ref["lowres"])
        results = []
        sub = 0
        while sub < 12 or e.in_flight():
            while sub < 12 and e.in_flight() < 4:
                e.submit(x, tag=sub)
                sub += 1
            results.append(e.collect())
        for tag, n, joints in results:
            assert n == ref["num_people"], (prec, tag)
            assert np.array_equal(joints, ref["joints"][:n]), (prec, tag)
        e.close()


def test_end_to_end_fp32_vs_oracle_full_chain():
    """Whole chain on the oracle (conv stack included) vs the fp32 engine: peak COUNTS may differ
    only where a heat value sits within 1e-4 of a threshold/neighbour, so compare the resized maps
    (tolerance) and, with planted-people weights impossible, the joints only when peak sets agree."""
    import caffe_rtpose_amd as r
    W, H = 96, 64
    e = _engine(net_w=W, net_h=H, precision=r.PREC_FP32, frames_in_flight=1)
    net = _oracle_net_from(e)
    x = _synth.random_frame(1, H, W, seed=42)
    d = e.forward_debug(x)
    low = net.forward(x)
    ref_res = orc.imresize(low, W, H, 1.0, 0.3)[0]
    assert np.abs(d["resized"] - ref_res).max() < 1e-3 * max(1.0, np.abs(ref_res).max())
    e.close()


def test_config1_640x480_frame_full_resolution_fp32_vs_full_oracle_chain():
    """BASELINE config 1 geometry: one 640x480 frame, --resolution 1280x720, --net_resolution 656x368,
    1 scale.  The WHOLE chain on the oracle (conv stack included, ~5-10 s of CPU) against the fp32
    engine: maps within 1e-3 of their magnitude; where the peak sets coincide the joints agree to
    +-1 px / +-1e-3 (north_star tolerance).  Discrete decisions (NMS '>' tests) can flip on 1e-5
    differences, so peak totals may differ by a few on noise maps."""
    import caffe_rtpose_amd as r
    W, H = 656, 368
    e = _engine(precision=r.PREC_FP32, frames_in_flight=1)
    net = _oracle_net_from(e)
    img = r.synth_frame(640, 480, 0, seed=1)
    x, _, fs = r.preprocess_frame(img, 1280, 720, W, H, 1, 1.0, 0.3)
    assert fs == 1.5
    d = e.forward_debug(x)
    low = net.forward(x)
    mag = float(np.abs(low).max())
    assert np.abs(d["lowres"] - low).max() < 1e-3 * mag
    ref_res = orc.imresize(low, W, H, 1.0, 0.3)[0]
    assert np.abs(d["resized"] - ref_res).max() < 1e-3 * mag
    thr = e.get_thresholds()
    ref_peaks = orc.nms(ref_res, 18, 64, thr["nms_threshold"])
    tot_ref, tot_eng = ref_peaks[:, 0, 0], d["peaks"][:, 0, 0]
    assert (np.abs(tot_ref - tot_eng) <= np.maximum(3, 0.02 * tot_ref)).all()
    same = [p for p in range(18) if tot_ref[p] == tot_eng[p] and tot_ref[p] <= 64]
    for p in same:
        n = int(tot_ref[p])
        assert np.abs(ref_peaks[p, 1:n + 1, :2] - d["peaks"][p, 1:n + 1, :2]).max(initial=0) < 1.0
        assert np.abs(ref_peaks[p, 1:n + 1, 2] - d["peaks"][p, 1:n + 1, 2]).max(initial=0) < 1e-3 * max(1.0, mag)
    e.close()


def test_weights_roundtrip_caffemodel_and_prototxt(tmp_path):
    """Save weights as .caffemodel + graph as prototxt, reload through --caffeproto/--caffemodel:
    identical low-res maps (net.cpp:750-803 CopyTrainedLayersFrom by layer name)."""
    import caffe_rtpose_amd as r
    W, H = 64, 48
    e = _engine(net_w=W, net_h=H, frames_in_flight=1, synthetic_seed=99)
    x = _synth.random_frame(1, H, W, seed=1)
    a = e.forward_heatmaps(x)
    e.save_caffemodel(tmp_path / "w.caffemodel")
    e.save_prototxt(tmp_path / "net.prototxt")
    e.close()
    e2 = _engine(net_w=W, net_h=H, frames_in_flight=1, proto_path=str(tmp_path / "net.prototxt"),
                 weights_path=str(tmp_path / "w.caffemodel"), calibrate_frames=-1)   # (-1: weights from a file are checked by default and the split set may change)
    b = e2.forward_heatmaps(x)
    assert np.array_equal(a, b)
    e2.close()


# ------------------------------------------------------------------------------------------
# row a1 on the device: u8 frame -> net input (bit-exact with the host restatement)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fw,fh,dw,dh,W,H,N,start,gap", [
    (1280, 720, 1280, 720, 656, 368, 1, 1.0, 0.3),     # reference defaults (rtpose.cpp:50-72)
    (1920, 1080, 1280, 720, 656, 368, 3, 1.0, 0.15),   # down-warp + 3-scale pyramid
    (640, 480, 1280, 720, 656, 368, 2, 1.0, 0.25),     # up-warp with a zero border on the right
    (333, 517, 640, 480, 320, 240, 4, 1.0, 0.2),       # odd sizes, portrait, 4 scales
    (640, 480, 656, 368, 656, 368, 1, 1.0, 0.3),       # display == net: identity level (memcpy branch)
    (1280, 720, 1280, 720, 656, 368, 1, 0.8, 0.15),    # round 5: --start_scale != 1 (s = start_scale - i gap, rtpose.cpp:353-368)
    (1280, 720, 1280, 720, 656, 368, 3, 0.8, 0.15),
    (1920, 1080, 1280, 720, 656, 368, 2, 0.65, 0.25),
    (720, 1280, 720, 1280, 368, 656, 2, 0.8, 0.15),    # portrait net and display
    (1920, 1080, 1920, 1080, 1312, 736, 2, 0.8, 0.15), # large net
])
def test_device_preprocess_bit_exact(fw, fh, dw, dh, W, H, N, start, gap):
    import caffe_rtpose_amd as r
    e = _engine(net_w=W, net_h=H, num_scales=N, start_scale=start, scale_gap=gap, disp_w=dw, disp_h=dh, frames_in_flight=1)
    for idx in (0, 3):
        img = r.synth_frame(fw, fh, idx, seed=11)
        if idx == 3:  # noise: every rounding boundary gets exercised
            img = np.random.default_rng(5).integers(0, 256, img.shape, dtype=np.uint8)
        want_x, want_disp, want_fs = r.preprocess_frame(img, dw, dh, W, H, N, start, gap)
        x, disp, fs = e.debug_preprocess(img)
        assert fs == want_fs
        assert np.array_equal(disp, want_disp)
        assert np.array_equal(x, want_x)
    e.close()


def test_submit_frame_equals_host_preprocess_plus_submit():
    import caffe_rtpose_amd as r
    e = _engine(net_w=320, net_h=176, num_scales=2, scale_gap=0.25, disp_w=640, disp_h=360, frames_in_flight=3)
    imgs = [r.synth_frame(*wh, i, seed=3) for i, wh in enumerate([(800, 600), (640, 360), (1280, 720), (320, 200), (801, 603)])]
    ref = []
    for im in imgs:
        x, _, fs = r.preprocess_frame(im, 640, 360, 320, 176, 2, 1.0, 0.25)
        e.submit(x, tag=1)
        ref.append((e.collect(), fs))
    got = []
    pending = []
    for i, im in enumerate(imgs):  # keep 3 in flight, frames of different sizes reuse the staging buffers
        pending.append(e.submit_frame(im, tag=100 + i))
        if len(pending) == 3:
            got.append((e.collect(), pending.pop(0)))
    while pending:
        got.append((e.collect(), pending.pop(0)))
    for i, ((tag, n, joints), fs) in enumerate(got):
        (_, rn, rj), rfs = ref[i]
        assert tag == 100 + i and n == rn and fs == rfs
        assert np.array_equal(joints, rj)
    e.close()


def test_submit_frame_with_an_enlarging_pyramid_level_runs_on_the_device():
    """display smaller than the net input: cv::resize(INTER_AREA) then enlarges with its bilinear kernel and area-mode coefficients
    (tests/_cvref.py resize_area_enlarging).  Round 5: that level runs on the device too (round 4: host fallback) — same numbers as
    rtp_preprocess_frame, which equals the independent restatement (test_host_preprocess_equals_independent_opencv_restatement)."""
    import caffe_rtpose_amd as r
    import _cvref
    e = _engine(net_w=160, net_h=96, num_scales=2, scale_gap=0.4, disp_w=120, disp_h=80, frames_in_flight=1)   # level 0 enlarges (160x96 from 120x80), level 1 shrinks (96x64)
    img = r.synth_frame(320, 240, 1, seed=9)
    x, disp, fs = r.preprocess_frame(img, 120, 80, 160, 96, 2, 1.0, 0.4)
    want_x, want_disp, want_fs = _cvref.producer_frame(img, 120, 80, 160, 96, 2, 1.0, 0.4, orc.process_and_pad_image)
    assert np.array_equal(x, want_x) and np.array_equal(disp, want_disp) and fs == want_fs
    gx, gdisp, gfs = e.debug_preprocess(img)
    assert gfs == fs and np.array_equal(gdisp, disp) and np.array_equal(gx, x)
    e.submit(x, tag=5)
    want = e.collect()
    fs2 = e.submit_frame(img, tag=5)
    got = e.collect()
    assert fs2 == fs and got[0] == want[0] and got[1] == want[1] and np.array_equal(got[2], want[2])
    e.close()


# ------------------------------------------------------------------------------------------
# frame batching: B frames share one conv launch sequence; per-frame results do not depend on B
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec_name,B,N", [("fp16", 3, 1), ("fp32", 2, 2)])
def test_frame_batching_is_transparent(prec_name, B, N):
    import caffe_rtpose_amd as r
    prec = r.PREC_FP16 if prec_name == "fp16" else r.PREC_FP32
    W, H = 320, 176
    e = _engine(net_w=W, net_h=H, num_scales=N, scale_gap=0.25, disp_w=640, disp_h=360, frames_in_flight=2 * B, batch_frames=B, precision=prec)
    imgs = [r.synth_frame(640, 360, i, seed=21) for i in range(2 * B + 2)]   # 2 full batches + a partial one
    xs = [r.preprocess_frame(im, 640, 360, W, H, N, 1.0, 0.25)[0] for im in imgs]
    want = [e.forward_debug(x) for x in xs]                                   # one frame alone (slot 0, nimg = N)
    assert sum(d["num_people"] for d in want) > 0
    # the fp32 plan against the oracle's conv stack (the batch plan picks other tiles than B = 1)
    if prec_name == "fp32":
        net = _oracle_net_from(e)
        net.forward(xs[0], keep_all=True)
        assert _rel_err(e.forward_heatmaps(xs[0]), net.blob("concat_stage7")) < 2e-4
    got = []
    for i, x in enumerate(xs):           # mixed entry points, FIFO order, partial batch launched by collect
        if i % 2 == 0:
            e.submit(x, tag=i)
        else:
            e.submit_frame(imgs[i], tag=i)
        while e.in_flight() >= 2 * B:
            got.append(e.collect())
    while e.in_flight():
        got.append(e.collect())
    assert [g[0] for g in got] == list(range(len(xs)))
    for (tag, n, joints), d in zip(got, want):
        assert n == d["num_people"]
        assert np.array_equal(joints, d["joints"][:n])
    # flush: a lone frame in an open batch runs without waiting for the batch to fill
    e.submit(xs[0], tag=77)
    e.flush()
    tag, n, joints = e.collect()
    assert tag == 77 and n == want[0]["num_people"] and np.array_equal(joints, want[0]["joints"][:n])
    e.close()


# ------------------------------------------------------------------------------------------
# production post-processing (no resized map in memory) == oracle ImResize -> Nms -> connect
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,W,H,N,start,gap,kind", [
    (0, 656, 368, 1, 1.0, 0.3, "people5"),
    (0, 656, 368, 3, 1.0, 0.15, "people20"),
    (1, 496, 368, 2, 1.0, 0.3, "people4"),
    (0, 656, 368, 1, 1.0, 0.3, "noise"),        # saturates max_peaks: raster-order cap, thousands of PAF candidates
    (0, 64, 48, 2, 1.0, 0.25, "noise"),         # tiny map: strips shorter than 8 rows, borders everywhere
    (1, 496, 368, 1, 1.0, 0.3, "noise"),
    # round 5: --start_scale != 1: the strip kernel's low-res row ranges, crops and the bound pre-pass with padw / padh > 0 at scale 0
    (0, 656, 368, 1, 0.8, 0.15, "noise"),
    (0, 656, 368, 2, 0.8, 0.15, "noise"),
    (0, 656, 368, 3, 0.8, 0.15, "speople5"),
    (0, 656, 368, 1, 0.65, 0.25, "speople5"),    # one scale: the column-skip path on a cropped map
    (0, 656, 368, 3, 0.65, 0.25, "noise"),       # scale 2 = 0.15: a 14 x 8 crop
    (1, 496, 368, 2, 0.8, 0.25, "speople5"),
    (1, 496, 368, 2, 0.65, 0.15, "noise"),
    (0, 1312, 736, 2, 0.8, 0.15, "noise"),       # large net: 164-column low-res rows, LDS-staged PAF planes do not fit
    (0, 1312, 736, 1, 0.8, 0.15, "speople5"),
    (0, 368, 656, 1, 1.0, 0.3, "noise"),         # portrait: the first max_peaks maxima lie in the rows the write kernel's `width` bound keeps
    (0, 64, 48, 2, 0.7, 0.3, "noise"),
])
def test_fused_postproc_from_lowres_bit_exact(model, W, H, N, start, gap, kind):
    e = _engine(model=model, net_w=W, net_h=H, num_scales=N, start_scale=start, scale_gap=gap, frames_in_flight=1)
    tabs = orc.model_tables(model)
    thr = e.get_thresholds()
    h, w = H // 8, W // 8
    if kind == "noise":
        low = (_synth.smooth_field(N * e.heat_channels, h, w, seed=31, scale=1.0).reshape(N, e.heat_channels, h, w)
               + 0.25 * np.random.default_rng(7).standard_normal((N, e.heat_channels, h, w)).astype(np.float32))
    elif kind.startswith("speople"):   # the same people at every scale, planted in each scale's crop window
        import _pincases as pc
        low = pc.scaled_people(model, tabs, int(kind[7:]), h, w, 44, N, start, gap)
    else:
        low, _ = _synth.people_lowres(model, tabs, int(kind[6:]), h, w, seed=44, N=N)
        low = low.reshape(N, e.heat_channels, h, w)
    ref_res = orc.imresize(low, W, H, start, gap)[0]
    ref_peaks = orc.nms(ref_res, e.num_parts, e.max_peaks, thr["nms_threshold"])
    peaks, joints, n = e.post_from_lowres(low)
    assert np.array_equal(peaks, ref_peaks)
    if kind == "noise" and W > 100:
        assert ref_peaks[:, 0, 0].max() >= e.max_peaks        # the cap was exercised
    try:
        rn, rj = orc.connect(model, ref_res, ref_peaks, e.max_peaks, W, H, 1280, 720, thr)
    except Exception:
        rn = None
    if rn is not None:
        assert n == rn
        assert np.array_equal(joints[:n], rj[:n])
    if kind != "noise":
        assert n >= 1
    # and the map-materialising taps agree with it too
    res = e.resize(low)
    assert np.array_equal(res, ref_res)
    n2, j2 = e.connect(res, e.nms(res))
    assert n2 == n and np.array_equal(j2[:n2], joints[:n])
    e.close()


def test_fused_postproc_repeatable_while_another_engine_keeps_the_chip_busy():
    """The production NMS/connect kernels share CUs with MFMA-heavy convolution workgroups of other
    frames.  Same low-res maps in => same bits out, however the chip is loaded (this caught packed-f32
    VALU ops returning different bits inside divergent code under exactly that co-residency)."""
    import threading
    import caffe_rtpose_amd as r
    kw = dict(net_w=320, net_h=176, num_scales=2, scale_gap=0.25, disp_w=640, disp_h=360)
    a = _engine(frames_in_flight=1, **kw)
    b = _engine(frames_in_flight=3, **kw)
    x = r.preprocess_frame(r.synth_frame(800, 600, 0, seed=3), 640, 360, 320, 176, 2, 1.0, 0.25)[0]
    low = a.forward_debug(x)["lowres"]
    pk0, j0, n0 = a.post_from_lowres(low)
    stop = []

    def load():
        pend = 0
        while not stop:
            b.submit(x, tag=0)
            pend += 1
            if pend == 3:
                b.collect()
                pend -= 1
        while pend:
            b.collect()
            pend -= 1

    t = threading.Thread(target=load)
    t.start()
    try:
        bad = 0
        for _ in range(250):
            pk, j, n = a.post_from_lowres(low)
            bad += not (n == n0 and np.array_equal(pk, pk0) and np.array_equal(j, j0))
    finally:
        stop.append(1)
        t.join()
    assert bad == 0, f"{bad} of 250 taps differed under load"
    a.close()
    b.close()


# ------------------------------------------------------------------------------------------
# renderer (§8f-3): pose overlay on the display image == oracle restatement of renderFunctions.cu
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,W,H", [(0, 320, 176), (1, 256, 192)])
def test_rendered_frame_matches_oracle(model, W, H):
    """The device evaluates atan2f/sinf/cosf with ocml, the oracle with glibc: a last-ulp difference
    can flip the `judge <= 1` test of a pixel ON an ellipse boundary, so the comparison allows a few
    boundary pixels (< 0.05%) and requires everything else to be identical."""
    import caffe_rtpose_amd as r
    dw, dh = 640, 360
    e = _engine(model=model, net_w=W, net_h=H, disp_w=dw, disp_h=dh, frames_in_flight=2, render=1)
    thr = orc.default_thresholds(model)
    e.set_thresholds(thr["nms_threshold"], thr["inter_threshold"], thr["inter_min_above"], 2, 0.05)   # keep more "people" of the noise maps
    total = 0
    for i in range(3):
        img = r.synth_frame(800, 600, i, seed=13)
        _, disp, _ = r.preprocess_frame(img, dw, dh, W, H, 1, 1.0, 0.3)
        e.submit_frame(img, tag=i)
        tag, n, joints, got = e.collect_rendered()
        assert tag == i
        want = orc.render_pose(model, disp, joints, n)
        bad = (got != want).any(-1)
        assert bad.mean() < 5e-4, f"{int(bad.sum())} pixels differ"
        if n == 0:
            assert np.array_equal(got, disp)
        total += n
    assert total > 0, "the test frames produced no people: nothing was drawn"
    # frames submitted as float tensors have no display image
    x = r.preprocess_frame(r.synth_frame(800, 600, 0, seed=13), dw, dh, W, H, 1, 1.0, 0.3)[0]
    e.submit(x, tag=9)
    with pytest.raises(r.RtpError):
        e.collect_rendered()
    e.close()


def test_part_to_show_view_in_the_pipeline_matches_oracle():
    """rtp_config.render = 1 + part_to_show: the frame rtp_collect_rendered returns is the oracle's view (itself pinned on the
    reference's render kernels) of THAT frame's resized map — which the pipeline materialises only for this mode."""
    import caffe_rtpose_amd as r
    W, H, dw, dh = 320, 176, 640, 360
    for part in (3, 19, 22):
        e = _engine(model=0, net_w=W, net_h=H, disp_w=dw, disp_h=dh, frames_in_flight=2, render=1 + part)
        img = r.synth_frame(800, 600, 1, seed=13)
        x, disp, _ = r.preprocess_frame(img, dw, dh, W, H, 1, 1.0, 0.3)
        e.submit_frame(img, tag=5)
        tag, n, joints, got = e.collect_rendered()
        resized = e.forward_debug(x)["resized"]
        want = orc.render_view(0, disp, resized, part)
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        if part <= 19:
            assert d.max() == 0, f"part_to_show {part}: {int((d > 0).any(-1).sum())} pixels differ"
        else:   # PAF view: atan2 (ocml vs glibc)
            assert d.max() <= 1 and (d > 0).any(-1).mean() < 1e-3
        assert (got != disp).any()
        e.close()
    with pytest.raises(r.RtpError):
        _engine(model=0, net_w=W, net_h=H, render=1 + 40)


# ------------------------------------------------------------------------------------------
# execution modes: the captured launch plan (hipGraph replay, default) == eager launches
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [1, 2])
def test_graph_replay_equals_eager_launches(B):
    import caffe_rtpose_amd as r
    W, H = 320, 176
    kw = dict(net_w=W, net_h=H, disp_w=640, disp_h=360, frames_in_flight=2 * B, batch_frames=B)
    eg = _engine(exec_mode=r.EXEC_GRAPH, **kw)
    ee = _engine(exec_mode=r.EXEC_EAGER, **kw)
    imgs = [r.synth_frame(640, 360, i, seed=33) for i in range(2 * B + 1)]  # full batches + a partial one (its own capture)
    xs = [r.preprocess_frame(im, 640, 360, W, H, 1, 1.0, 0.3)[0] for im in imgs]

    def run(e, timed):
        e.kernel_timing(1 if timed else 0)
        out = []
        for rep in range(2):  # second pass = pure replays
            for i, x in enumerate(xs):
                if i % 2:
                    e.submit_frame(imgs[i], tag=10 * rep + i)
                else:
                    e.submit(x, tag=10 * rep + i)
                while e.in_flight() >= 2 * B:
                    out.append(e.collect())
            while e.in_flight():
                out.append(e.collect())
        return out, e.kernel_timing(0)

    for timed in (False, True):
        a, ta = run(eg, timed)
        b, tb = run(ee, timed)
        assert [t for t, _, _ in a] == [t for t, _, _ in b]
        assert sum(n for _, n, _ in a) > 0
        for (_, na, ja), (_, nb, jb) in zip(a, b):
            assert na == nb and np.array_equal(ja, jb)
        if timed:  # in-kernel stamps of the dominant launches come back through the graph too
            assert ta[1] == tb[1] and ta[1] > 0 and ta[2] == tb[2]
            assert 0.2 < ta[0] / tb[0] < 5.0
    assert eg.last_stage_ms()["total"] > 0
    eg.close()
    ee.close()


# ------------------------------------------------------------------------------------------
# row a1 / f1 on the device against the independent OpenCV restatement (tests/_cvref.py) + the reference's own
# process_and_pad_image semantics (oracle, pinned on oracle/_ref): not product vs product
# ------------------------------------------------------------------------------------------
def _a1_geoms():
    from test_preprocess_cli import GEOMS
    return GEOMS


@pytest.mark.parametrize("geom", _a1_geoms())
def test_device_preprocess_equals_independent_opencv_restatement(geom):
    import caffe_rtpose_amd as r
    import _cvref
    from test_preprocess_cli import _frame
    fw, fh, dw, dh, W, H, N, start, gap = geom
    e = _engine(net_w=W, net_h=H, num_scales=N, start_scale=start, scale_gap=gap, disp_w=dw, disp_h=dh, frames_in_flight=1)
    img = _frame(fw, fh, seed=fw + fh)
    want_x, want_disp, want_fs = _cvref.producer_frame(img, dw, dh, W, H, N, start, gap, orc.process_and_pad_image)
    x, disp, fs = e.debug_preprocess(img)
    assert fs == want_fs
    assert np.array_equal(disp, want_disp)
    assert np.array_equal(x, want_x)
    e.close()


# ------------------------------------------------------------------------------------------
# experiment variants kept in the tree must keep producing the production bits
# ------------------------------------------------------------------------------------------
def test_ring_kernel_variants_are_bit_identical():
    """RTP_RING_ILV=1 (interleaved A-fragment rows: DPP shifts instead of LDS re-reads, conv_ring.hip) and RTP_HALO_SHARED=0 (a
    halo on both sides of every row) only change HOW operands reach the MFMAs: low-res maps, blobs and joints hash identically
    to the default build (tools/ab_hash.py, separate processes: the switches are read once per process).  The knobs exist in the
    EXPERIMENTS build only (librtpose_mi355x_exp.so, selected by RTP_LIB); the production library ignores them — checked too."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ab_hash.py")

    def run(**env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("RTP_")}   # (the suite itself may be running against another library: RTP_LIB)
        e.update(env)
        out = subprocess.run([sys.executable, tool, "--quick"], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return [l for l in out.stdout.splitlines() if l and not l.startswith("#")]

    exp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "caffe_rtpose_amd", "librtpose_mi355x_exp.so")
    assert os.path.exists(exp), "build the experiments library (make -C caffe_rtpose_amd/csrc)"
    strip = lambda lines: [l.split(" RTP_")[0] if " RTP_" in l else l for l in lines]   # (ab_hash prints the RTP_* environment as a tag)
    base = run()
    assert len(base) == 2 and all("nan 0" in l for l in base)
    assert strip(run(RTP_LIB=exp)) == strip(base)                                # the experiments build without knobs = the production bits
    assert strip(run(RTP_LIB=exp, RTP_RING_ILV="1")) == strip(base)
    assert strip(run(RTP_LIB=exp, RTP_HALO_SHARED="0")) == strip(base)
    assert strip(run(RTP_LIB=exp, RTP_PREP_DEFER="1")) == strip(base)             # the copy-only stream + deferred pre-processing kernels (an experiment: prep_defer defaults to 0,
                                                                                  # copies and kernels stay on the batch's conv stream) vs the default
    assert strip(run(RTP_RING_VAR="12", RTP_DIAG_SKIP_POST="2")) == strip(base)  # the production library does not know these names


# ------------------------------------------------------------------------------------------
# round 4: caller-owned device buffers, one-time weight distribution
# ------------------------------------------------------------------------------------------
def test_engine_owned_device_buffers_feed_submit_device():
    """rtp_device_alloc / upload / free: net inputs resident in HBM that belong to the ENGINE's HIP runtime (bench.py keeps torch out of
    the data path with them).  rtp_submit_device on such a buffer == rtp_submit of the same host tensor, bit for bit."""
    import caffe_rtpose_amd as r
    e = _engine(net_w=160, net_h=96, precision=r.PREC_MIXED, frames_in_flight=4, batch_frames=2)
    xs = [_synth.random_frame(1, 96, 160, seed=s_) for s_ in (1, 2)]
    ptrs = [e.device_frame(x) for x in xs]
    for i, x in enumerate(xs):
        e.submit(x, tag=i)
    want = [e.collect() for _ in xs]
    for i, p in enumerate(ptrs):
        e.submit_device(p, tag=10 + i)
    got = [e.collect() for _ in xs]
    e.synchronize()
    for (t0, n0, j0), (t1, n1, j1) in zip(want, got):
        assert t1 == t0 + 10 and n0 == n1 and np.array_equal(j0, j1)
    e.device_free(ptrs[0])
    with pytest.raises(r.RtpError):
        e.device_free(ptrs[0])          # not (any more) a buffer of this engine
    e.close()                           # the second buffer goes with the engine


def test_weight_blob_and_peer_copy_reproduce_the_source_engine():
    """The optional one-time weight distribution: a replica that takes another engine's PACKED arena — through the host blob
    (bench.py --broadcast_weights) or device to device (rtpose.bin --share_weights) — computes the source's maps bit for bit and reports
    the source's Caffe-layout weights; a blob from another plan is refused."""
    import caffe_rtpose_amd as r
    kw = dict(net_w=160, net_h=96, precision=r.PREC_MIXED, frames_in_flight=2, batch_frames=1)
    src = _engine(synthetic_seed=11, **kw)
    x = _synth.random_frame(1, 96, 160, seed=4)
    want = src.forward_heatmaps(x)
    blob = src.weight_blob()
    a = _engine(synthetic_seed=1, **kw)
    assert not np.array_equal(a.forward_heatmaps(x), want)
    a.load_weight_blob(blob)
    assert np.array_equal(a.forward_heatmaps(x), want)
    assert all(np.array_equal(a.get_conv_weights(i)[0], src.get_conv_weights(i)[0]) for i in (0, 17, 91))
    a.submit(x, tag=3)                  # graphs captured before the import bake the old fp8 weight exponents: they must have been dropped
    src.submit(x, tag=3)
    ra, rs = a.collect(), src.collect()
    assert ra[1] == rs[1] and np.array_equal(ra[2], rs[2])
    b = _engine(synthetic_seed=2, **kw)
    b.copy_weights_from(src)
    assert np.array_equal(b.forward_heatmaps(x), want)
    other = _engine(synthetic_seed=1, net_w=160, net_h=96, precision=r.PREC_FP16, frames_in_flight=2, batch_frames=1)
    with pytest.raises(r.RtpError):
        other.load_weight_blob(blob)    # another plan (precision): refused, nothing written
    with pytest.raises(r.RtpError):
        other.copy_weights_from(src)
    # same sizes, other CONTENTS: a layer split ":w" carries W_lo in its second pass, split ":a" carries W_hi again — equal nchunk and
    # byte counts, another arena.  The plan hash must tell them apart, and likewise ":x" (fp16 instead of fp8 corrections).
    kw3 = dict(net_w=160, net_h=96, precision=r.PREC_MIXED, frames_in_flight=1, batch_frames=1)
    ew = _engine(split_layers="conv2_:w,@1x1", **kw3)
    ea = _engine(split_layers="conv2_:a,@1x1", **kw3)
    ew2 = _engine(split_layers="conv2_:w,@1x1", synthetic_seed=3, **kw3)
    blob_w = ew.weight_blob()
    assert len(blob_w) == ea.weight_blob_bytes()                # the sizes agree, so only the hash can refuse it
    before = ea.forward_heatmaps(x)
    with pytest.raises(r.RtpError):
        ea.load_weight_blob(blob_w)
    with pytest.raises(r.RtpError):
        ea.copy_weights_from(ew)
    assert np.array_equal(ea.forward_heatmaps(x), before)       # refused = nothing written
    ew2.load_weight_blob(blob_w)                                # the same split set takes it
    assert np.array_equal(ew2.forward_heatmaps(x), ew.forward_heatmaps(x))
    # rtp_config.defer_weights (round 5): a RECEIVING replica is created without reading, generating, packing or uploading weights; the net cannot
    # run until the blob / the peer copy arrives, and then it computes the source's maps bit for bit (graphs are captured at delivery)
    recv = _engine(defer_weights=1, frames_in_flight=4, batch_frames=2, **{k: v for k, v in kw.items() if k not in ("frames_in_flight", "batch_frames")})
    for call in (lambda: recv.forward_heatmaps(x), lambda: recv.submit(x, tag=1), lambda: recv.weight_blob(), lambda: recv.calibrate_precision(nframes=1),
                 lambda: recv.set_conv_weights(0, *src.get_conv_weights(0)), lambda: b.copy_weights_from(recv)):
        with pytest.raises(r.RtpError) as ei:
            call()
        assert ei.value.code == r.RTP_EINVAL
    assert np.count_nonzero(recv.get_conv_weights(3)[0]) == 0          # sizes exist, contents do not
    src2 = _engine(synthetic_seed=11, frames_in_flight=4, batch_frames=2, **{k: v for k, v in kw.items() if k not in ("frames_in_flight", "batch_frames")})
    recv.load_weight_blob(src2.weight_blob())
    assert np.array_equal(recv.forward_heatmaps(x), src2.forward_heatmaps(x)) and np.array_equal(recv.get_conv_weights(3)[0], src.get_conv_weights(3)[0])
    for t in range(3):                                                  # a full batch and a trailing partial one through the graphs captured at delivery
        recv.submit(x, tag=t)
        src2.submit(x, tag=t)
    got, ref3 = [recv.collect() for _ in range(3)], [src2.collect() for _ in range(3)]
    assert all(g[0] == q[0] and g[1] == q[1] and np.array_equal(g[2], q[2]) for g, q in zip(got, ref3))
    recv2 = _engine(defer_weights=1, **kw)
    recv2.copy_weights_from(src)                                        # device to device
    assert np.array_equal(recv2.forward_heatmaps(x), want)
    for e in (src, a, b, other, ew, ea, ew2, recv, src2, recv2):
        e.close()


def test_deferred_preprocessing_and_staging_streams_do_not_change_results():
    """rtp_submit_frame variants of the experiments build that only change WHEN / WHERE a frame's H2D copy and pre-processing kernels are
    issued (RTP_PREP_DEFER=1: the kernels are enqueued once the copy has completed and a full batch is launched at the next call;
    RTP_IN_STREAM=1 / 2: a staging stream): the joints of 23 pipelined frames — full batches, the trailing partial batch, batches of 1
    and 2 — hash identically to the production library's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "ab_frames.py")
    exp = os.path.join(root, "caffe_rtpose_amd", "librtpose_mi355x_exp.so")

    def run(args, **env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("RTP_")}
        e.update(env)
        out = subprocess.run([sys.executable, tool] + args, env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return [l for l in out.stdout.splitlines() if l.startswith("frames")][-1].split(": ")[1]

    for args in (["2", "7"], ["1", "3"], ["2", "2"]):
        base = run(args)
        assert run(args, RTP_LIB=exp, RTP_PREP_DEFER="1") == base, args
        assert run(args, RTP_LIB=exp, RTP_PREP_DEFER="0") == base, args
    assert run(["2", "7"], RTP_LIB=exp, RTP_CHAIN_CONNECT="1") == run(["2", "7"])     # pairs -> match -> assemble as one launch (tickets)
    assert run(["2", "7"], RTP_LIB=exp, RTP_IN_STREAM="1") == run(["2", "7"])
    assert run(["2", "7"], RTP_LIB=exp, RTP_IN_STREAM="2", RTP_PREP_DEFER="1") == run(["2", "7"])
    # the hardware-queue count bench.py / rtpose.bin choose for batches of 2 (GPU_MAX_HW_QUEUES=6: which streams share a queue changes, results do not)
    assert run(["2", "7"], GPU_MAX_HW_QUEUES="6") == run(["2", "7"], GPU_MAX_HW_QUEUES="4")
