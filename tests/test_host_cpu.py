"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, the host
logic behind it (graph builder / prototxt parser / plan / caffemodel IO / JSON / preprocessing /
model tables / std::sort replica) agrees with the oracle and with fixtures derived from the
reference.  No compute kernels are called here."""
import ctypes as C
import json
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import _oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtpose_mi355x.h")
GOLDEN = os.path.join(ROOT, "tests", "golden", "linevec_layers.json")


def test_abi_library_exports_every_declared_symbol():
    import caffe_rtpose_amd as r
    from caffe_rtpose_amd import _lib
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(rtp_[a-z0-9_]+)\s*\(", text))
    declared -= {"rtp_config", "rtp_engine"}
    assert len(declared) >= 35
    for name in sorted(declared):
        assert hasattr(r.lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert b"gfx950" in r.lib.rtp_version()


def test_production_library_has_no_experiment_knobs():
    """ablation variants of the ring kernel that compute wrong results on purpose, forced tiles, skipped
    post-processing stages and probes used to be selectable by environment variables in the shipped library.  They live in the second
    build target now (librtpose_mi355x_exp.so, -DRTP_EXPERIMENTS, used by tools/); in the production library the knobs' NAMES are not
    even strings in the binary.  The one variable it reads, RTP_EXEC=eager|graph, chooses between two ways of launching the same
    kernels."""
    lib = os.path.join(ROOT, "caffe_rtpose_amd", "librtpose_mi355x.so")
    exp = os.path.join(ROOT, "caffe_rtpose_amd", "librtpose_mi355x_exp.so")
    names = lambda path: set(re.findall(rb"RTP_[A-Z][A-Z0-9_]+", open(path, "rb").read()))
    consts = set(re.findall(rb"#define (RTP_[A-Z0-9_]+)", open(HEADER, "rb").read()))   # RTP_PREC_MIXED etc. appear in error messages
    prod = names(lib) - consts
    assert prod == {b"RTP_EXEC"}, prod
    removed = {b"RTP_RING_VAR", b"RTP_FORCE_CFG", b"RTP_TILE_OVERRIDE", b"RTP_DIAG_SKIP_POST", b"RTP_RING_SB", b"RTP_RING_SPEC", b"RTP_POST_CUS", b"RTP_NMS_PROBE",
               b"RTP_SPLIT_LAYERS", b"RTP_HALO_SHARED", b"RTP_HALF_CHIP", b"RTP_GRAPH_POST", b"RTP_STREAM_PLAN", b"RTP_MATCH_WGS"}
    assert os.path.exists(exp) and removed <= names(exp)            # the experiments build still has them (tools/, tests of the variants)
    src = open(os.path.join(ROOT, "caffe_rtpose_amd", "csrc", "engine.cpp")).read() + open(os.path.join(ROOT, "caffe_rtpose_amd", "csrc", "conv_ring.hip")).read()
    assert re.findall(r"[^_A-Z]getenv\(\"(RTP_[A-Z_]+)\"\)", src.replace("#ifdef RTP_EXPERIMENTS", "")) .count("RTP_EXEC") == 1


def test_no_cpu_fallback_when_no_device():
    import torch
    import caffe_rtpose_amd as r
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(r.RtpError) as ei:
        r.Engine()
    assert ei.value.code == -19  # RTP_ENODEV
    assert b"no CPU fallback" in r.lib.rtp_last_error(None)


BAD_GEOMETRY = [dict(start_scale=1.2), dict(start_scale=0.0), dict(start_scale=-0.5), dict(start_scale=float("nan")), dict(start_scale=float("inf")),
                dict(num_scales=3, start_scale=0.5, scale_gap=0.25),      # third scale = 0: cv::resize to an empty image in the reference
                dict(num_scales=2, scale_gap=float("nan")), dict(num_scales=2, scale_gap=-0.3),   # second scale 1.3 > 1
                dict(net_w=650), dict(net_h=100), dict(net_w=0), dict(net_h=-16), dict(num_scales=0), dict(num_scales=17)]


def test_out_of_contract_scales_and_net_sizes_are_einval_before_any_device_is_touched():
    """The reference CHECKs (rtpose.cpp:363-364: target <= net resolution) and dies inside cv::resize for a scale <= 0; the C ABI returns
    RTP_EINVAL from rtp_engine_create — also where no GPU exists: argument errors come before RTP_ENODEV — and from
    rtp_preprocess_frame, never a crash or an exception across the boundary."""
    import caffe_rtpose_amd as r
    for kw in BAD_GEOMETRY:
        with pytest.raises(r.RtpError) as ei:
            r.Engine(r.Config(**{**dict(net_w=160, net_h=96, frames_in_flight=1), **kw}))
        assert ei.value.code == r.RTP_EINVAL, kw
    img = np.zeros((48, 64, 3), np.uint8)
    for kw in BAD_GEOMETRY:
        a = {**dict(net_w=160, net_h=96, num_scales=1, start_scale=1.0, scale_gap=0.3), **kw}
        if a["num_scales"] > 16 or a["num_scales"] < 1 or a["net_w"] < 1 or a["net_h"] < 1:
            continue     # (the ctypes mirror cannot allocate an output for these; the host function has no scale cap of its own)
        with pytest.raises(r.RtpError) as ei:
            r.preprocess_frame(img, 320, 180, a["net_w"], a["net_h"], a["num_scales"], a["start_scale"], a["scale_gap"])
        assert ei.value.code == r.RTP_EINVAL, kw
    with pytest.raises(r.RtpError):
        r.preprocess_frame(img, 0, 180, 160, 96)
    # in contract: --start_scale 0.8 --scale_gap 0.15 --num_scales 3 -> levels 0.8, 0.65, 0.5, each padded into the net input
    x, _, _ = r.preprocess_frame(np.full((180, 320, 3), 200, np.uint8), 320, 180, 160, 96, 3, 0.8, 0.15)
    assert x.shape == (3, 3, 96, 160)
    for i, s in enumerate((0.8, 0.65, 0.5)):
        tw, th = int(16 * np.ceil(160 * np.float32(s) / 16)), int(16 * np.ceil(96 * np.float32(s) / 16))
        inside = x[i, 0] != 0
        assert inside.sum() == tw * th and inside[(96 - th) // 2, (160 - tw) // 2] and not inside[0, 0] or (tw, th) == (160, 96)


def test_config_defaults_are_the_reference_flag_defaults():
    # rtpose.cpp:50-72
    import caffe_rtpose_amd as r
    c = r.Config().c
    assert (c.net_w, c.net_h, c.disp_w, c.disp_h, c.num_scales) == (656, 368, 1280, 720, 1)
    assert c.start_scale == 1.0 and abs(c.scale_gap - 0.3) < 1e-7
    assert c.precision == r.PREC_MIXED and c.exec_mode == r.EXEC_GRAPH   # the mode that meets the +-1e-3 tolerance; captured launch plan


def test_model_tables_and_thresholds_match_oracle():
    import caffe_rtpose_amd as r
    for m in (0, 1):
        assert r.model_tables(m) == orc.model_tables(m)
        assert r.default_thresholds(m) == orc.default_thresholds(m)
    with pytest.raises(r.RtpError):
        r.model_tables(7)


def _plan_lines(**kw):
    import caffe_rtpose_amd as r
    return r.plan_summary(r.Config(**kw)).strip().split("\n")


def test_plan_coco_flops_and_pairing():
    lines = _plan_lines()
    assert lines[0] == "model 0 parts 18 max_peaks 64 heat_channels 57"
    gf = {l.split()[0]: float(l.split()[1]) for l in lines if l.startswith(("conv_gflop", "mfma_gflop"))}
    assert gf["conv_gflop"] == pytest.approx(484.634, abs=1e-3)
    # default precision (mixed): split 3x3 / 7x7 layers cost one fp16 pass + one fp8 compensation chunk per channel group (= 2 pass
    # times), split 1x1 layers three fp16 passes
    assert 810 < gf["mfma_gflop"] < 900      # (814.8 when every split k x k layer's tile supports the fp8 chunks, e.g. at batch_frames 2)
    b2 = _plan_lines(batch_frames=2)
    assert float([l for l in b2 if l.startswith("mfma_gflop")][0].split()[1]) == pytest.approx(814.765, abs=1e-2)
    fp16 = _plan_lines(precision=0)
    assert float([l for l in fp16 if l.startswith("mfma_gflop")][0].split()[1]) == pytest.approx(484.634, abs=1e-3)
    convs = [l for l in lines if l.startswith("step conv")]
    pw2 = [l for l in lines if l.startswith("step pw2")]
    # 12 VGG/CPM singles + 5 stage-1 pairs + 5x7 refinement pairs = 52 launches, of which the six branch tails
    # (1x1 -> 1x1: conv5_4/5_5 and Mconv6/Mconv7 of stages 2-6) are fused two layers per launch
    # conv1_1 reads the NCHW image itself (conv_first.hip): no pack step, no im2col tensor
    assert lines.count("step pack") == 0 and sum(l.startswith("step first conv1_1 ") for l in lines) == 1
    assert len(convs) == 11 + 3 + 25 and len(pw2) == 6
    f16x3 = _plan_lines(precision=3)   # split weights: conv1_1 takes the generic route (pack + 1x1 with K = 32)
    assert f16x3.count("step pack") == 1 and any(l.startswith("step conv conv1_1 ") for l in f16x3)
    assert sum(" + " in l for l in convs) == 28 and all(l.count(" + ") == 2 for l in pw2)
    assert any("conv4_4_CPM" in l and "dsts 6" in l for l in convs)  # own tensor + 5 concat slices
    assert [l for l in pw2 if "Mconv7_stage6" in l][0].endswith("lowres 1")
    # the three pooling layers run inside the epilogues of conv1_2 / conv2_2 / conv3_4 (conv_ring.hip POOL); keep_blobs = 1 keeps their launches
    assert sum(l.startswith("step pool") for l in lines) == 0 and sum("+pool" in l for l in convs) == 3
    kept = _plan_lines(keep_blobs=1)
    assert sum(l.startswith("step pool") for l in kept) == 3 and not any("+pool" in l for l in kept)
    # 3 scales: same graph, 3x the work
    l3 = _plan_lines(num_scales=3, scale_gap=0.15)
    assert float([l for l in l3 if l.startswith("conv_gflop")][0].split()[1]) == pytest.approx(3 * 484.634, abs=3e-3)


def _tiles(lines, needle):
    out = set()
    for l in lines:
        if l.startswith("step conv") and needle in l:
            w = l.split()
            out.add((w[w.index("tile") + 1], int(w[w.index("rowb") + 1]), int(w[w.index("wgs") + 1])))
    return out


def test_tile_model_choices_of_the_benched_plans():
    """engine.cpp build_plan scores tiles by a time model (MFMA vs L2->LDS time per K step, partial rounds, dispatch, credit for tiles
    that pool in the epilogue).  The choices below are the ones measured on the GPU (profiles/r03_tile_model.txt, r03_steps.txt)."""
    dom = " k 7 cin_p 128 cout 128 "
    entry = " k 7 cin_p 192 cout 128 "
    # COCO 656x368, batches of 2 (bench.py default): 248 workgroups of 128x64 tiles, 256-byte chunks, for the dominant shape; the stage-entry
    # layers (and conv4_3, conv5_1..3) as half-chip launches of 128x128 tiles: two conv stacks share the chip (RTP_HALF_CHIP, +2 % frames/s)
    b2 = _plan_lines(batch_frames=2)
    assert _tiles(b2, dom) == {("128x64", 256, 248)} and _tiles(b2, entry) == {("128x128", 128, 124)}
    assert _tiles([l for l in b2 if "conv5_" in l], " k 3 cin_p 128 cout 128 ") == {("128x128", 128, 124)}
    # MPI 496x368 (46x65 padded pixels = 24 M tiles): batches of 5 fill the chip with 128x128 tiles (bench.py --model mpi) ...
    m5 = _plan_lines(model=1, net_w=496, net_h=368, batch_frames=5)
    assert _tiles(m5, dom) == {("128x128", 128, 240)}
    # ... and at batches of 2 ONE under-filled round of 128x64 tiles beats two rounds of 128x32 (round 2's rule: 384 workgroups; 1089 -> 1240 frames/s)
    m2 = _plan_lines(model=1, net_w=496, net_h=368, batch_frames=2)
    assert _tiles(m2, dom) == {("128x64", 256, 192)}
    # 3 scales, one frame per launch sequence (bench.py's 3-scale configuration): 186 workgroups of 128x128 tiles, two conv stacks share the chip
    s3 = _plan_lines(num_scales=3, scale_gap=0.15, batch_frames=1)
    assert _tiles(s3, dom) == {("128x128", 128, 186)}
    # the three layers in front of a pooling layer keep 128-pixel tiles of 128-byte chunks (the POOL kernel) in all of them
    for pl in (b2, m5, s3):
        pooled = [l for l in pl if "+pool" in l]
        assert pooled and all(" rowb 128 " in l and (" tile 128x64 " in l or " tile 128x128 " in l) for l in pooled)


def test_plan_mpi_and_errors():
    import caffe_rtpose_amd as r
    lines = _plan_lines(model=1, net_w=496, net_h=368)
    assert lines[0] == "model 1 parts 15 max_peaks 20 heat_channels 44"
    assert float([l for l in lines if l.startswith("conv_gflop")][0].split()[1]) == pytest.approx(361.695, abs=1e-3)
    with pytest.raises(r.RtpError):
        _plan_lines(net_w=650)  # not a multiple of 16
    with pytest.raises(r.RtpError):
        _plan_lines(model=5)


def _golden_table(model):
    g = json.load(open(GOLDEN))["coco" if model == 0 else "mpi"]
    return g


def _parse_prototxt_py(path):
    t = re.sub(r"#.*", "", open(path).read())
    out = []
    for L in re.split(r"\nlayer \{", "\n" + t)[1:]:
        d = {"name": re.search(r'name: "(.*?)"', L).group(1), "type": re.search(r'type: "(.*?)"', L).group(1),
             "bottoms": re.findall(r'bottom: "(.*?)"', L), "tops": re.findall(r'top: "(.*?)"', L)}
        for key, rx in (("num_output", r"num_output: (\d+)"), ("kernel", r"kernel_size: (\d+)"), ("pad", r"\bpad: (\d+)"),
                        ("max_peaks", r"max_peaks: (\d+)"), ("num_parts", r"num_parts: (\d+)")):
            m = re.search(rx, L)
            if m:
                d[key] = int(m.group(1))
        out.append(d)
    return out


@pytest.mark.parametrize("model", [0, 1])
def test_builtin_graph_equals_reference_prototxt_fixture(model, tmp_path):
    """The built-in generator vs the layer table extracted from the reference's
    model/*/pose_deploy_linevec.prototxt (tests/golden/linevec_layers.json, made by
    tools/make_golden_from_reference.py): same layers, order, wiring and parameters."""
    import caffe_rtpose_amd as r
    p = tmp_path / "builtin.prototxt"
    r.write_builtin_prototxt(model, p)
    mine = _parse_prototxt_py(p)
    gold = _golden_table(model)["layers"]
    assert len(mine) == len(gold) == 183
    for a, b in zip(mine, gold):
        if b["type"] == "ReLU":  # ReLU layer names carry no weights; wiring must match
            assert a["type"] == "ReLU" and a["bottoms"] == b["bottoms"] and a["tops"] == b["tops"]
            continue
        for k in ("name", "type", "bottoms", "tops"):
            assert a[k] == b[k], (a, b)
        for k in ("num_output", "kernel", "pad", "max_peaks", "num_parts"):
            assert a.get(k) == b.get(k), (k, a, b)
    s = r.prototxt_summary(p)
    assert s["num_layers"] == 183 and s["num_conv"] == 92
    assert s["heat_channels"] == (57 if model == 0 else 44)


@pytest.mark.parametrize("rel,model", [("model/coco/pose_deploy_linevec.prototxt", 0), ("model/mpi/pose_deploy_linevec.prototxt", 1)])
def test_reference_prototxt_parses_to_the_same_plan(rel, model):
    """--caffeproto on the reference's own file gives the same execution plan as the built-in graph.
    Needs /root/reference (build container only)."""
    import caffe_rtpose_amd as r
    path = os.path.join("/root/reference", rel)
    if not os.path.exists(path):
        pytest.skip("reference tree not present on this machine")
    w, h = (656, 368) if model == 0 else (496, 368)
    a = r.plan_summary(r.Config(model=model, net_w=w, net_h=h))
    b = r.plan_summary(r.Config(proto_path=path, net_w=w, net_h=h))
    assert a == b
    s = r.prototxt_summary(path)
    assert s["num_conv"] == 92 and s["num_parts"] == (18 if model == 0 else 15)
    assert s["nms_threshold"] == pytest.approx(0.05 if model == 0 else 0.6)  # runtime overrides MPI to 0.2, rtpose.cpp:214


def _varint(buf, p):
    v = s = 0
    while True:
        b = buf[p]
        p += 1
        v |= (b & 0x7F) << s
        if not b & 0x80:
            return v, p
        s += 7


def test_caffemodel_writer_is_valid_protobuf_and_reader_roundtrips(tmp_path):
    """Decode the written NetParameter with an independent wire-format walker (caffe.proto:64-95,
    310-330, 10-22) and compare with the generator; then read it back through the C++ reader."""
    import caffe_rtpose_amd as r
    path = tmp_path / "w.caffemodel"
    r.write_synthetic_caffemodel(1, 42, path)
    buf = open(path, "rb").read()
    p = 0
    layers = []
    while p < len(buf):
        key, p = _varint(buf, p)
        fn, wt = key >> 3, key & 7
        assert wt == 2
        n, p = _varint(buf, p)
        if fn == 100:
            layers.append(buf[p:p + n])
        p += n
    assert len(layers) == 92
    # first layer: name, type, 2 blobs
    L = layers[0]
    q = 0
    name = typ = None
    blobs = []
    while q < len(L):
        key, q = _varint(L, q)
        n, q = _varint(L, q)
        if key >> 3 == 1:
            name = L[q:q + n].decode()
        elif key >> 3 == 2:
            typ = L[q:q + n].decode()
        elif key >> 3 == 7:
            blobs.append(L[q:q + n])
        q += n
    assert (name, typ, len(blobs)) == ("conv1_1", "Convolution", 2)
    b0 = blobs[0]
    q = 0
    shape = data = None
    while q < len(b0):
        key, q = _varint(b0, q)
        n, q = _varint(b0, q)
        if key >> 3 == 7:
            sh = b0[q:q + n]
            k2, s2 = _varint(sh, 0)
            n2, s2 = _varint(sh, s2)
            dims = []
            e = s2 + n2
            while s2 < e:
                v, s2 = _varint(sh, s2)
                dims.append(v)
            shape = dims
        elif key >> 3 == 5:
            data = np.frombuffer(b0[q:q + n], "<f4")
        q += n
    assert shape == [64, 3, 3, 3]
    w, b = r.synth_weights(42, "conv1_1", 64, 3, 3)
    assert np.array_equal(data, w.ravel())
    got = r.read_caffemodel_layers(path)
    net = orc.Net(1)
    assert [g["name"] for g in got] == [c[0] for c in net.convs]
    for g, (nm, cin, cout, k) in zip(got, net.convs):
        assert g["num_blobs"] == 2 and g["count0"] == cout * cin * k * k and g["count1"] == cout
        assert np.array_equal(g["head0"], r.synth_weights(42, nm, cout, cin, k)[0].ravel()[:8])


def test_caffemodel_reader_accepts_v1_layers_and_legacy_dims(tmp_path):
    """V1LayerParameter (NetParameter.layers = 2: name = 4, blobs = 6) with legacy num/channels/height/
    width blob dims (caffe.proto:17-21) and UNPACKED repeated floats."""
    import caffe_rtpose_amd as r

    def vi(v):
        out = b""
        while v >= 0x80:
            out += bytes([(v & 0x7F) | 0x80])
            v >>= 7
        return out + bytes([v])

    def field(fn, payload):
        return vi((fn << 3) | 2) + vi(len(payload)) + payload

    data = np.arange(8, dtype="<f4") * 0.5
    blob = b"".join(vi((i << 3) | 0) + vi(d) for i, d in zip((1, 2, 3, 4), (2, 1, 2, 2)))
    blob += b"".join(vi((5 << 3) | 5) + struct.pack("<f", float(x)) for x in data)
    layer = field(4, b"old_conv") + field(6, blob)
    path = tmp_path / "v1.caffemodel"
    open(path, "wb").write(field(1, b"net") + field(2, layer))
    got = r.read_caffemodel_layers(path)
    assert len(got) == 1 and got[0]["name"] == "old_conv" and got[0]["count0"] == 8
    assert np.array_equal(got[0]["head0"], data)


def test_synthetic_weights_statistics_and_determinism():
    import caffe_rtpose_amd as r
    w1, b1 = r.synth_weights(1, "Mconv3_stage4_L1", 128, 128, 7)
    w2, b2 = r.synth_weights(1, "Mconv3_stage4_L1", 128, 128, 7)
    assert np.array_equal(w1, w2) and np.array_equal(b1, b2)
    w3, _ = r.synth_weights(2, "Mconv3_stage4_L1", 128, 128, 7)
    assert not np.array_equal(w1, w3)
    assert abs(float(w1.std()) - np.sqrt(2.0 / (128 * 49))) < 2e-4 and abs(float(w1.mean())) < 1e-4
    assert -0.1 <= b1.min() and b1.max() < 0.1


def test_json_bytes_match_oracle_and_reference_shape():
    """rtpose.cpp:1394-1415: `std::ofstream <<` at default precision == printf("%g")."""
    import caffe_rtpose_amd as r
    rs = np.random.RandomState(0)
    for n, parts, scale in [(0, 18, 1.5), (1, 18, 1.5), (3, 15, 0.6666667), (96, 18, 2.0)]:
        j = (rs.rand(max(n, 1), parts, 3) * np.array([1280, 720, 1])).astype(np.float32)
        j[0, 0] = (0, 0, 0)                       # a missing part
        j[0, 1] = (1234567.0, 1e-7, 0.000123456)  # exponent / precision edge cases of %g
        a = r.format_json(j, n, parts, scale)
        b = orc.write_json(j, n, parts, scale)
        assert a == b
        assert a.startswith(b'{\n"version":0.1,\n"bodies":[\n') and a.endswith(b"]\n}\n")
        assert a.count(b'"joints":[') == n


def test_process_and_pad_image_matches_oracle():
    import caffe_rtpose_amd as r
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    for tw, th, norm in [(64, 48, 1), (53, 37, 0), (100, 90, 1)]:
        a = r.process_and_pad_image(img, tw, th, norm)
        b = orc.process_and_pad_image(img, tw, th, norm)
        assert np.array_equal(a, b)
    padw, padh = (64 - 53) // 2, (48 - 37) // 2
    a = r.process_and_pad_image(img, 64, 48, 1)
    assert a[2, padh, padw] == np.float32(img[0, 0, 2]) / np.float32(256.0) - np.float32(0.5)
    assert a[:, 0, :].max() == 0 and a[:, :, 0].max() == 0  # zero padding
    with pytest.raises(r.RtpError):
        r.process_and_pad_image(img, 40, 48, 1)  # "Image too big for target size."


def test_stdsort_replica_matches_libstdcxx(tmp_path):
    """csrc/stdsort_replica.h (what the connect kernel runs on ties) vs the real std::sort on the
    container type and comparator of rtpose.cpp:144-152,953-954, tie-heavy inputs."""
    exe = tmp_path / "stdsort_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "helpers", "stdsort_check.cpp")])
    out = subprocess.check_output([str(exe), "1500"]).decode()
    assert out.startswith("OK"), out


def test_bit_exact_kernels_contain_no_packed_f32_valu():
    """postproc.hip / preproc.hip / render.hip and the conv kernels must compile without v_pk_*_f32: with them the
    production NMS kernel was not repeatable under MFMA co-residency on gfx950."""
    import shutil
    import subprocess
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "caffe_rtpose_amd", "csrc"), "check-nopk"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    counts = [int(x) for x in out.stdout.split()]
    assert counts == [0] * 7, out.stdout   # postproc, preproc, render, conv_ring, conv_igemm, conv_pw2, conv_first


def test_caffemodel_blobshape_lengths_are_bounds_checked(tmp_path):
    """A BlobShape whose packed-dims length runs past the message (heap over-read) is a parse
    error, not a read beyond the file buffer; a well-formed new-style shape still loads."""
    import caffe_rtpose_amd as r

    def vi(v):
        out = b""
        while v >= 0x80:
            out += bytes([(v & 0x7F) | 0x80])
            v >>= 7
        return out + bytes([v])

    def field(fn, payload):
        return vi((fn << 3) | 2) + vi(len(payload)) + payload

    floats = np.arange(4, dtype="<f4").tobytes()
    good_shape = field(1, bytes([1, 1, 2, 2]))                       # BlobShape.dim packed: 1x1x2x2
    good = field(1, b"n") + field(100, field(1, b"c") + field(7, field(7, good_shape) + field(5, floats)))
    p = tmp_path / "good.caffemodel"
    open(p, "wb").write(good)
    got = r.read_caffemodel_layers(p)
    assert got[0]["name"] == "c" and got[0]["count0"] == 4
    for bad_len in (0x7F, 0xFFFF, 1 << 40):
        bad_shape = vi((1 << 3) | 2) + vi(bad_len) + bytes([1, 1])   # claims bad_len bytes of dims, has 2
        bad = field(1, b"n") + field(100, field(1, b"c") + field(7, field(7, bad_shape) + field(5, floats)))
        p = tmp_path / "bad.caffemodel"
        open(p, "wb").write(bad)
        with pytest.raises(r.RtpError):
            r.read_caffemodel_layers(p)


def test_prototxt_kernel_h_w_spelling(tmp_path):
    """Caffe accepts kernel_size or the kernel_h/kernel_w pair; a square pair must parse, a non-square or
    half-specified one is outside the linevec path."""
    import caffe_rtpose_amd as r
    src = tmp_path / "a.prototxt"
    r.write_builtin_prototxt(r.MODEL_COCO_18, src)
    text = open(src).read()
    assert "kernel_size: 3" in text
    sq = tmp_path / "sq.prototxt"
    open(sq, "w").write(text.replace("kernel_size: 3", "kernel_h: 3 kernel_w: 3", 1))
    assert r.prototxt_summary(sq) == r.prototxt_summary(src)
    for bad in ("kernel_h: 3 kernel_w: 5", "kernel_h: 3", "kernel_w: 3"):
        b = tmp_path / "bad.prototxt"
        open(b, "w").write(text.replace("kernel_size: 3", bad, 1))
        with pytest.raises(r.RtpError):
            r.prototxt_summary(b)


def test_split_precision_plan_and_rule_syntax():
    """RTP_PREC_MIXED / F16X3: which layers run the three-pass hi/lo split is host logic (no GPU needed)."""
    import caffe_rtpose_amd as r
    base = r.plan_summary(r.Config(precision=r.PREC_FP16))
    assert "passes 3" not in base and "mfma_gflop 484.6" in base
    allx = r.plan_summary(r.Config(precision=r.PREC_F16X3))
    assert "passes 1 " not in allx.replace("passes 1 impl reg wgs 3784", "")   # conv1_1: image exact in fp16, weights split only
    s = r.plan_summary(r.Config(precision=r.PREC_MIXED, split_layers="conv4_4:w,*_stage6_L:a,@1x1"))
    lines = {ln.split()[2]: ln for ln in s.splitlines() if ln.startswith(("step conv", "step pw2"))}
    assert " passes 2w " in lines["conv4_4_CPM"] and " passes 2a " in lines["Mconv2_stage6_L1"]   # partial splits stay fp16 passes
    dflt = {ln.split()[2]: ln for ln in r.plan_summary(r.Config(precision=r.PREC_MIXED)).splitlines() if ln.startswith(("step conv", "step pw2"))}
    assert " passes 2q " in dflt["conv3_2"] and " passes 2q " in dflt["Mconv3_stage6_L1"] and " passes 1 " in dflt["Mconv3_stage3_L1"]
    assert " passes 1 " in lines["conv3_1"]
    assert " passes 3aw/3aw " in lines["Mconv6_stage6_L1"]   # the fused 1x1 pair: passes of the first / second layer
    # ":x" (what the load-time calibration switches a group to): both operands split, the corrections as fp16 passes instead of the fp8 chunk;
    # a later rule for the same layers wins over the plain rule in front of it, other groups keep their fp8 chunk
    x = {ln.split()[2]: ln for ln in r.plan_summary(r.Config(precision=r.PREC_MIXED, batch_frames=2, split_layers="conv2_,conv3_,*_stage5_,*_stage5_:x,@1x1")).splitlines()
         if ln.startswith(("step conv", "step pw2"))}
    assert " passes 3aw " in x["Mconv2_stage5_L1"] and " passes 3aw " in x["Mconv1_stage5_L1"] and " passes 2q " in x["conv3_2"] and " passes 1 " in x["Mconv2_stage4_L1"]
    cost = lambda sp: float([ln for ln in r.plan_summary(r.Config(precision=r.PREC_MIXED, split_layers=sp)).splitlines() if ln.startswith("mfma_gflop")][0].split()[1])
    assert cost("*_stage5_:x") > cost("*_stage5_") > cost("@1x1")       # three pass-times > two > one on those layers


def test_eighth_resolution_launches_leave_cus_free():
    """Shared row halo: a 46x82 image is 31 M-tiles of 128 (46 * 85 = 3910 GEMM rows), so at batch_frames = 2
    every 1/8-resolution launch of the default plan is 248 workgroups — not 256, which would need every CU of the chip at once — or, for
    the layers the plan runs as half-chip launches (128x128 tiles: two conv stacks share the chip, engine.cpp RTP_HALF_CHIP), 124."""
    import re
    import caffe_rtpose_amd as r
    steps = [ln for ln in r.plan_summary(r.Config(precision=r.PREC_MIXED, frames_in_flight=8, batch_frames=2)).splitlines() if ln.startswith("step")]
    low = [ln for ln in steps if re.search(r"(conv4_|conv5_|Mconv)", ln)]
    assert len(low) >= 38
    for ln in low:
        wgs = int(re.search(r"wgs (\d+)", ln).group(1))
        assert wgs == 248 or (wgs == 124 and " tile 128x128 " in ln), ln
    dom = [ln for ln in low if " k 7 cin_p 128 " in ln]
    assert len(dom) == 20 and all("wgs 248" in ln and " tile 128x64 " in ln for ln in dom)   # the dominant shape keeps full-chip launches


def test_stream_arrangement_follows_the_hardware_queue_count(monkeypatch):
    """engine.cpp 'hardware queues': a batch context gets ONE stream (staging, conv stack, every frame's post-processing chain) when the HIP
    runtime has at least as many hardware queues (GPU_MAX_HW_QUEUES, default 4) as there are contexts, the per-frame chain streams of rounds 1-5
    otherwise; bench.py and rtpose.bin raise the count to 8 (round 6: +9..12 % frames/s at batches of 2)."""
    def arrangement(**kw):
        ln = [l for l in _plan_lines(**kw) if l.startswith("streams ")]
        assert len(ln) == 1
        w = ln[0].split()
        return int(w[2]), int(w[4]), w[6]
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert arrangement(batch_frames=2, frames_in_flight=7) == (5, 4, "per_frame_chains")       # five contexts on the runtime's four queues
    assert arrangement(batch_frames=2, frames_in_flight=5) == (4, 4, "one_per_context")
    assert arrangement(model=1, net_w=496, net_h=368, batch_frames=5, frames_in_flight=15) == (4, 4, "one_per_context")
    assert arrangement(batch_frames=1, frames_in_flight=3)[2] == "one_per_context"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert arrangement(batch_frames=2, frames_in_flight=7) == (5, 8, "one_per_context")
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "0")                                                   # nonsense falls back to the default
    assert arrangement(batch_frames=2, frames_in_flight=7)[1] == 4


def test_a_config_of_another_header_version_is_refused():
    """include/rtpose_mi355x.h rtp_config.struct_size: rtp_config_default writes the size of the library's struct; rtp_plan_summary and
    rtp_engine_create refuse any other value (a caller compiled against another header version, or one that skipped rtp_config_default)
    with RTP_EINVAL and both sizes in rtp_last_error(NULL), before anything else is read.  The reference's counterpart is a gflags parse
    (rtpose.cpp:50-72): there is no struct to get wrong."""
    import caffe_rtpose_amd as r
    from caffe_rtpose_amd._lib import lib, rtp_config
    good = rtp_config()
    assert lib.rtp_config_default(C.byref(good)) == 0 and good.struct_size == C.sizeof(rtp_config)   # the ctypes mirror has the library's layout
    buf = C.create_string_buffer(1 << 16)
    assert lib.rtp_plan_summary(C.byref(good), buf, len(buf)) > 0
    for bad_size in (0, C.sizeof(rtp_config) - 4, C.sizeof(rtp_config) + 8):
        bad = rtp_config()
        lib.rtp_config_default(C.byref(bad))
        bad.struct_size = bad_size
        assert lib.rtp_plan_summary(C.byref(bad), buf, len(buf)) == r.RTP_EINVAL
        msg = lib.rtp_last_error(None).decode()
        assert f"struct_size is {bad_size}" in msg and str(C.sizeof(rtp_config)) in msg
        h = C.c_void_p()
        assert lib.rtp_engine_create(C.byref(bad), C.byref(h)) == r.RTP_EINVAL and not h.value
