"""tests/golden/ref_pin.npz holds what the REFERENCE'S OWN CODE (oracle/_ref, built from /root/reference by
oracle/ref_recipe/build_ref.sh; fixture written by tools/make_ref_golden.py) returns on the seeded cases of _pincases.py.
  * CPU: the oracle reproduces every stored output bit for bit (also where libref.so is absent);
  * GPU: the HIP engine, through the C-ABI, reproduces them — directly against the reference's outputs, no oracle in
    between: ImResize / Nms / connect taps AND the production path that never materialises the resized map."""
import os

import numpy as np
import pytest

import _oracle as orc
import _pincases as pc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_pin.npz"))


def _tables():
    t = {}
    for m in (0, 1):
        a = G[f"tables_{m}"].tolist()
        nl = a[1]
        t[m] = (a[0], nl, a[2:2 + 2 * nl], a[2 + 2 * nl:2 + 4 * nl])
    return t


def _sha(name):
    return G[name].tobytes().decode()


def test_oracle_conv_and_pool_reproduce_the_reference_outputs():
    """The conv / pool half of the oracle against what the reference's own caffe_conv, im2col_cpu (+ GEMM) and pooling loop returned
    (tools/make_ref_golden.py ran them in the build container; here — and on the GPU box — only their outputs are needed)."""
    for name, x, w, b, pad, stride in pc.conv_cases():
        got = orc.conv2d(x, w, b, pad, stride)
        naive, gemm = G[f"conv_{name}_naive"], G[f"conv_{name}_im2col"]
        scale = max(1.0, float(np.abs(naive).max()))
        assert got.shape == naive.shape and np.abs(got - naive).max() <= 1e-4 * scale and np.abs(got - gemm).max() <= 2e-5 * scale, name
    for name, x, k, stride, pad in pc.pool_cases():
        assert np.array_equal(orc.maxpool(x, k, stride, pad), G[f"pool_{name}"]), name


def test_oracle_reproduces_the_reference_outputs():
    tables = _tables()
    for m in (0, 1):
        assert orc.model_tables(m) == tables[m]
    for name, (model, low, W, H, start, gap) in pc.lowres_cases(tables).items():
        res, peaks, n, joints = pc.chain(orc, model, low, W, H, start, gap, pc.disp_of(name))
        assert pc.digest(res) == _sha(f"chain_{name}_resized_sha"), name
        assert np.array_equal(peaks, G[f"chain_{name}_peaks"], equal_nan=True), name
        assert n == int(G[f"chain_{name}_count"][0]) and np.array_equal(joints, G[f"chain_{name}_joints"]), name
    for nm, (res, peaks) in (("ties", pc.tie_case()), ("single", pc.single_sided_case())):
        n, joints = orc.connect(0, res, peaks, 64, 656, 368, 1280, 720, pc.THR[0])
        assert np.array_equal(joints[:n], G[f"connect_{nm}_joints"])
    for i, (model, n, joints, scale) in enumerate(pc.json_cases()):
        parts = pc.DIMS[model][0]
        assert orc.write_json(joints if n else np.zeros((1, parts, 3), np.float32), n, parts, float(scale)) == G[f"json_{i}"].tobytes()
    for i, (img, tw, th, normalize) in enumerate(pc.pad_cases()):
        assert pc.digest(orc.process_and_pad_image(img, tw, th, normalize)) == _sha(f"pad_{i}_sha")
    for name, model, img, joints, n, googly in pc.render_cases():
        assert np.array_equal(orc.render_pose(model, img, joints, n, googly), G[f"render_{name}"]), name
    for name, model, img, maps, parts in pc.view_cases(tables):
        for part in parts:
            assert np.array_equal(orc.render_view(model, img, maps, part), G[f"view_{name}_{part}"]), (name, part)


def test_host_library_reproduces_the_reference_outputs():
    """The product's host functions (no GPU needed): model tables, process_and_pad_image, JSON writer."""
    import caffe_rtpose_amd as r
    tables = _tables()
    for m in (0, 1):
        assert r.model_tables(m) == tables[m]
    for i, (model, n, joints, scale) in enumerate(pc.json_cases()):
        assert r.format_json(joints, n, pc.DIMS[model][0], float(scale)) == G[f"json_{i}"].tobytes()
    for i, (img, tw, th, normalize) in enumerate(pc.pad_cases()):
        assert pc.digest(r.process_and_pad_image(img, tw, th, normalize)) == _sha(f"pad_{i}_sha")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["coco_noise_1s", "coco_noise_3s", "mpi_noise_1s", "small_noise_2s", "coco_people1", "coco_people5",
                                  "coco_people20", "coco_people5_3s", "mpi_people5"] + pc.ROUND5_CASES)
def test_engine_reproduces_the_reference_outputs(name):
    """Round 5 adds --start_scale 0.8 / 0.65 at 1-3 scales (the crop of scale 0 moves too, imresize_layer.cu:110-113), a portrait and a
    1312x736 net, MPI, and display resolutions other than 1280x720 in connect's output scaling (rtpose.cpp:1051-1073)."""
    import caffe_rtpose_amd as r
    model, low, W, H, start, gap = pc.lowres_cases(_tables())[name]
    N = low.shape[0]
    dw, dh = pc.disp_of(name)
    e = r.Engine(r.Config(model=model, net_w=W, net_h=H, num_scales=N, start_scale=start, scale_gap=gap, disp_w=dw, disp_h=dh, frames_in_flight=1))
    low = np.ascontiguousarray(low, np.float32)
    want_peaks, want_joints, want_n = G[f"chain_{name}_peaks"], G[f"chain_{name}_joints"], int(G[f"chain_{name}_count"][0])
    # taps through the materialised map (the reference's own dataflow)
    res = e.resize(low)
    assert np.array_equal(res.reshape(-1)[::997], G[f"chain_{name}_resized_sample"])
    assert pc.digest(res) == _sha(f"chain_{name}_resized_sha")
    peaks = e.nms(res)
    assert np.array_equal(peaks, want_peaks, equal_nan=True)
    if want_n < 0:
        # the reference CHECK-fails in connect (portrait net: NaN centroids, see _pincases.chain): RTP_ERANGE on every path, never a crash
        with pytest.raises(r.RtpError) as ei:
            e.connect(res, peaks)
        assert ei.value.code == r.RTP_ERANGE
        with pytest.raises(r.RtpError) as ei:
            e.post_from_lowres(low)
        assert ei.value.code == r.RTP_ERANGE
        e.close()
        return
    n, joints = e.connect(res, peaks)
    assert n == want_n and np.array_equal(joints[:n], want_joints)
    # production path: peaks and PAF samples straight from the low-res maps
    p2, j2, n2 = e.post_from_lowres(low)
    assert np.array_equal(p2, want_peaks) and n2 == want_n and np.array_equal(j2, want_joints)
    e.close()


@pytest.mark.gpu
def test_engine_connect_ties_stale_slots_vs_reference_outputs():
    import caffe_rtpose_amd as r
    e = r.Engine(r.Config(frames_in_flight=1))
    for nm, (res, peaks) in (("ties", pc.tie_case()), ("single", pc.single_sided_case())):
        n, joints = e.connect(res, peaks)
        want = G[f"connect_{nm}_joints"]
        assert n == len(want) and np.array_equal(joints[:n], want)
    model, low, W, H, start, gap = pc.lowres_cases(_tables())["coco_people5"]
    res = e.resize(np.ascontiguousarray(low, np.float32))
    assert np.array_equal(e.nms(res, pc.stale_peaks(18, 64)), G["nms_stale_peaks"])
    e.close()


@pytest.mark.gpu
def test_engine_renderer_vs_reference_outputs():
    """rtp_render (render.hip) against the frames the reference's renderFunctions.cu kernels produce (host run, golden fixture).
    Pose overlay: the device evaluates atan2f / sinf / cosf with ocml, the reference run with glibc — a last-ulp difference can flip
    the `judge <= 1` test of a pixel ON an ellipse boundary: < 0.05 % of the pixels may differ, the rest is identical.
    Heat-map views (no libm): identical.  PAF views (atan2 in double, rounded to float): at most a handful of pixels off by one."""
    import caffe_rtpose_amd as r
    tables = _tables()
    for name, model, img, joints, n, googly in pc.render_cases():
        h, w, _ = img.shape
        e = r.Engine(r.Config(model=model, net_w=160, net_h=96, disp_w=w, disp_h=h, frames_in_flight=1))
        got = e.render(img, joints, n, googly=googly)
        want = G[f"render_{name}"]
        bad = (got != want).any(-1)
        assert bad.mean() < 5e-4, f"{name}: {int(bad.sum())} pixels differ"
        if n == 0:
            assert np.array_equal(got, img)
        e.close()
    for name, model, img, maps, parts in pc.view_cases(tables):
        h, w, _ = img.shape
        e = r.Engine(r.Config(model=model, net_w=maps.shape[2], net_h=maps.shape[1], disp_w=w, disp_h=h, frames_in_flight=1))
        for part in parts:
            got = e.render(img, np.zeros((0, 3), np.float32), 0, part_to_show=part, resized=maps)
            want = G[f"view_{name}_{part}"]
            paf = part > pc.DIMS[model][0] + 1 if model == 0 else False
            if not paf:
                assert np.array_equal(got, want), f"{name} part_to_show {part}: {int((got != want).any(-1).sum())} pixels differ"
            else:
                d = np.abs(got.astype(np.int32) - want.astype(np.int32))
                assert d.max() <= 1 and (d > 0).any(-1).mean() < 1e-3, f"{name} PAF view {part}: max diff {d.max()}, {int((d > 0).any(-1).sum())} pixels"
        with pytest.raises(r.RtpError):   # a view past the model's maps (the reference would read beyond its blob)
            e.render(img, np.zeros((0, 3), np.float32), 0, part_to_show=(40 if model == 0 else 45), resized=maps)
        e.close()
