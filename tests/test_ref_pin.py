"""Pins oracle/rtpose_oracle.cpp (the CPU restatement every GPU parity test compares against) on the REFERENCE'S OWN
CODE: oracle/_ref/libref.so = connectLimbs / connectLimbsCOCO / process_and_pad_image / the --write_json block /
ModelDescriptorFactory compiled from /root/reference, and imresize_cubic_kernel / nms_register_kernel /
writeResultKernel compiled from the .cu files and run on the host (oracle/ref_recipe/).  Bit-for-bit on every case.

Runs where /root/reference exists (this container: __graft_entry__.build() builds libref.so).  The GPU box has no
reference tree: there the same outputs come from tests/golden/ref_pin.npz (tests/test_ref_golden.py)."""
import numpy as np
import pytest

import _oracle as orc
import _pincases as pc
import _ref

pytestmark = pytest.mark.skipif(not _ref.available(), reason="oracle/_ref/libref.so not built (needs /root/reference)")


def test_model_tables_are_the_factorys():
    for model in (0, 1):
        assert orc.model_tables(model) == _ref.model_tables(model)


def test_process_and_pad_image_bit_equal():
    for img, tw, th, normalize in pc.pad_cases():
        assert np.array_equal(orc.process_and_pad_image(img, tw, th, normalize), _ref.process_and_pad_image(img, tw, th, normalize))
    with pytest.raises(RuntimeError, match="too big"):   # CHECK_GE(padw, 0) << "Image too big for target size."
        _ref.process_and_pad_image(np.zeros((40, 40, 3), np.uint8), 32, 48, 1)


@pytest.fixture(scope="module")
def tables():
    return {m: _ref.model_tables(m) for m in (0, 1)}


def test_imresize_nms_connect_chain_bit_equal(tables):
    """noise maps (peaks saturate max_peaks), planted people 1/5/20, COCO + MPI, 1-3 scales: every stage output equal."""
    for name, (model, low, W, H, start, gap) in pc.lowres_cases(tables).items():
        a = pc.chain(orc, model, low, W, H, start, gap, pc.disp_of(name))
        b = pc.chain(_ref, model, low, W, H, start, gap, pc.disp_of(name))
        assert np.array_equal(a[0], b[0]), f"{name}: ImResize differs"
        assert np.array_equal(a[1], b[1], equal_nan=True), f"{name}: Nms differs"
        assert a[2] == b[2] and np.array_equal(a[3], b[3]), f"{name}: connect differs ({a[2]} vs {b[2]} people)"
        if "people" in name and name != "portrait_people3_s080_n2":
            assert a[2] >= 1, name
    # a net taller than wide: peaks below row `width` + 3 get NaN centroids (nms_layer.cu:79 bounds rows by width) and connect aborts
    a = pc.chain(_ref, *pc.lowres_cases(tables)["portrait_people3_s080_n2"], pc.disp_of("portrait_people3_s080_n2"))
    assert a[2] == -1 and np.isnan(a[1][:, 1:, :2]).any()


def test_nms_stale_slots_and_count_semantics(tables):
    """Slots beyond the peak count keep their previous contents; slot 0 holds the UNCLAMPED total (nms_layer.cu:110)."""
    model, low, W, H, start, gap = pc.lowres_cases(tables)["coco_people5"]
    res = _ref.imresize(low, W, H, start, gap)[0]
    init = pc.stale_peaks(18, 64)
    a, b = orc.nms(res, 18, 64, 0.05, init), _ref.nms(res, 18, 64, 0.05, init)
    assert np.array_equal(a, b)
    n = int(b[3, 0, 0])
    assert 0 < n < 64 and np.array_equal(b[3, n + 1:], init[3, n + 1:])
    noise = pc.lowres_cases(tables)["coco_noise_1s"]
    pk = _ref.nms(_ref.imresize(noise[1], 656, 368, 1.0, 0.3)[0], 18, 64, 0.05)
    assert pk[:, 0, 0].max() > 64          # saturated: more maxima than slots, count not clamped


def test_connect_ties_and_single_sided_bit_equal():
    for res, peaks in (pc.tie_case(), pc.single_sided_case()):
        a = orc.connect(0, res, peaks, 64, 656, 368, 1280, 720, pc.THR[0])
        b = _ref.connect(0, res, peaks, 64, 656, 368, 1280, 720, pc.THR[0])
        assert a[0] == b[0] and np.array_equal(a[1][:a[0]], b[1][:b[0]])
    assert orc.connect(0, *pc.tie_case(), 64, 656, 368, 1280, 720, pc.THR[0])[0] >= 1


def test_connect_out_of_range_sample_is_a_check_failure():
    """COCO: CHECK_GE(mx, 0) (rtpose.cpp:928) — the reference aborts; engine and oracle report RTP_ERANGE / -1."""
    res = np.zeros((57, 368, 656), np.float32)
    peaks = np.zeros((18, 65, 3), np.float32)
    peaks[1, 0, 0] = peaks[2, 0, 0] = 1
    peaks[1, 1] = (-30.0, 50.0, 0.9)
    peaks[2, 1] = (40.0, 60.0, 0.9)
    with pytest.raises(RuntimeError, match="mx >= 0"):
        _ref.connect(0, res, peaks, 64, 656, 368, 1280, 720, pc.THR[0])


def test_json_bytes_equal(tmp_path):
    for i, (model, n, joints, scale) in enumerate(pc.json_cases()):
        parts = pc.DIMS[model][0]
        name, ref_bytes = _ref.write_json(tmp_path, joints, n, model, float(scale), frame_number=i)
        assert name == f"frame{i:06d}.json"
        assert orc.write_json(joints if n else np.zeros((1, parts, 3), np.float32), n, parts, float(scale)) == ref_bytes
    name, _ = _ref.write_json(tmp_path, np.zeros((1, 18, 3), np.float32), 0, 0, 1.0, frame_number=0, image_path="/data/imgs/COCO_val_0001.jpg")
    assert name == "COCO_val_0001.json"       # <stem>.json for --image_dir (rtpose.cpp:1390-1393)


def test_pose_overlay_bit_equal():
    """orc_render_pose == render_pose_coco_parts / render_pose_29parts of renderFunctions.cu (run on the host, with the reference's
    swapped <<<threadsPerBlock, numBlocks>>> launch shape) between the producer's float canvas and the post-processing thread's
    float -> u8 conversion: identical u8 frames (both sides evaluate atan2f / sinf / cosf with the same libm here)."""
    for name, model, img, joints, n, googly in pc.render_cases():
        want = _ref.render(model, img, joints, n, 656, 368, part_to_show=0, googly=googly)
        got = orc.render_pose(model, img, joints, n, googly)
        assert np.array_equal(got, want), f"{name}: {int((got != want).any(-1).sum())} pixels differ"
        if n:
            assert (want != img).any(), name   # something was drawn
        else:
            assert np.array_equal(want, img)


def test_views_bit_equal(tables):
    """orc_render_view == render_pose_coco_heatmap / _heatmap2 / _affinity / render_pose_29parts_heatmap behind the dispatch of
    render() (rtpose.cpp:270-299) for single parts, the last part (initial value 1), all parts, all PAFs, single PAF pairs."""
    for name, model, img, maps, parts in pc.view_cases(tables):
        for part in parts:
            want = _ref.render(model, img, np.zeros((1, pc.DIMS[model][0], 3), np.float32), 0, maps.shape[2], maps.shape[1], part_to_show=part, heatmaps=maps)
            got = orc.render_view(model, img, maps, part)
            assert np.array_equal(got, want), f"{name} part_to_show {part}: {int((got != want).any(-1).sum())} pixels differ"
            assert (want != img).any(), (name, part)


# ---- the conv / pool half of the oracle, pinned on reference-held code --------------------------------------
def test_conv_equals_the_references_own_caffe_conv_and_im2col_path():
    """orc.conv2d (Caffe's CPU convolution restated: im2col K order, fp32) against (a) caffe_conv, the naive loop the reference's own
    convolution tests trust (test_convolution_layer.cpp:21-139, compiled from the reference tree), at the tolerance those tests use
    (1e-4), and (b) the reference's im2col_cpu (im2col.cpp:19-55) followed by the GEMM / bias shapes of forward_cpu_gemm — same K order,
    so the two agree to the rounding of a different summation blocking."""
    for name, x, w, b, pad, stride in pc.conv_cases():
        got = orc.conv2d(x, w, b, pad, stride)
        naive = _ref.caffe_conv(x, w, b, pad, stride)
        gemm = _ref.im2col_conv(x, w, b, pad, stride)
        assert got.shape == naive.shape == gemm.shape and np.isfinite(naive).all() and np.isfinite(gemm).all(), name
        scale = max(1.0, float(np.abs(naive).max()))
        assert np.abs(got - naive).max() <= 1e-4 * scale, (name, float(np.abs(got - naive).max()))
        assert np.abs(got - gemm).max() <= 2e-5 * scale, (name, float(np.abs(got - gemm).max()))
        assert np.abs(naive - gemm).max() <= 1e-4 * scale, name          # the reference agrees with itself


def test_im2col_buffer_is_the_k_order_the_oracle_documents():
    """K index of the im2col buffer = (cin * kh + r) * kw + s, spatial index = y * W_out + x (im2col.cpp:19-55): the order the oracle's
    convolution (and the engine's weight packing, which starts from Caffe's [cout][cin][kh][kw] blob) assumes."""
    rs = np.random.RandomState(3)
    x = rs.randn(3, 5, 6).astype(np.float32)
    col = _ref.im2col(x, 3, 1)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    for c in range(3):
        for r in range(3):
            for s_ in range(3):
                assert np.array_equal(col[(c * 3 + r) * 3 + s_].reshape(5, 6), xp[c, r:r + 5, s_:s_ + 6])


def test_maxpool_equals_the_references_pooling_loop():
    for name, x, k, stride, pad in pc.pool_cases():
        want = _ref.maxpool(x, k, stride, pad)
        got = orc.maxpool(x, k, stride, pad)
        assert got.shape == want.shape and np.array_equal(got, want), name
    y = _ref.maxpool(next(pc.pool_cases())[1], 2, 1, 0)
    assert np.array_equal(y[0, 0], np.array([[9, 5, 5, 8], [9, 5, 5, 8]], np.float32))   # TestForwardSquare's expected answer
