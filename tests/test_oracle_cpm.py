"""Internal-consistency tests of the oracle's restatements of the CMU-added pieces (ImResize, Nms,
connectLimbs*).  The reference ships no test or golden vector for any of them (SURVEY.md §4
FINDING 4): these are properties that follow from the cited source, NOT pins — parity unpinned."""
import numpy as np

import _oracle as orc
import _synth


def test_imresize_constant_and_shape():
    # Catmull-Rom weights sum to 1 (imresize_layer.cu:14-17): a constant map stays constant
    src = np.full((2, 3, 6, 8), 0.37, np.float32)
    dst = orc.imresize(src, 64, 48, 1.0, 0.25)
    assert dst.shape == (1, 3, 48, 64)  # top N forced to 1 (imresize_layer.cpp:37)
    np.testing.assert_allclose(dst, 0.37, atol=1e-6)


def test_imresize_interpolates_low_res_samples():
    # at the centre of a low-res cell (x = 8*i + 3.5 -> x_on = i exactly is not on the grid; the
    # kernel's sample positions are x_on = (x - 3.5)/8): a linear ramp is reproduced in the interior
    h, w = 6, 10
    ramp = np.tile(np.arange(w, dtype=np.float32), (h, 1))[None, None]
    dst = orc.imresize(ramp, 8 * w, 8 * h, 1.0, 0.3)[0, 0]
    xs = (np.arange(8 * w) - 3.5) / 8.0
    np.testing.assert_allclose(dst[20, 16:-16], xs[16:-16], atol=1e-5)


def test_imresize_multiscale_is_mean_of_cropped_scales():
    rs = np.random.RandomState(0)
    a = rs.rand(1, 2, 46, 82).astype(np.float32)
    two = np.concatenate([a, a], 0)
    # scale 0 uses the full map; with gap 0 scale 1 does too -> mean == single scale
    one = orc.imresize(a, 656, 368, 1.0, 0.3)
    both = orc.imresize(two, 656, 368, 1.0, 0.0)
    np.testing.assert_allclose(both, one, atol=1e-6)


def test_nms_single_peak_centroid_and_count():
    H, W = 40, 60
    m = np.zeros((3, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    m[0] = np.exp(-((xx - 30) ** 2 + (yy - 17) ** 2) / 8.0)
    peaks = orc.nms(m, 2, 5, 0.05)
    assert peaks[0, 0, 0] == 1 and peaks[1, 0, 0] == 0
    assert abs(peaks[0, 1, 0] - 30) < 1e-3 and abs(peaks[0, 1, 1] - 17) < 1e-3
    assert peaks[0, 1, 2] == m[0, 17, 30]
    # border pixels never fire (nms_layer.cu:20): a maximum on the edge is ignored
    m2 = np.zeros((2, H, W), np.float32)
    m2[0, 0, 10] = 1.0
    assert orc.nms(m2, 1, 5, 0.05)[0, 0, 0] == 0


def test_nms_raster_order_cap_and_unclamped_total():
    H, W = 30, 50
    m = np.zeros((2, H, W), np.float32)
    pts = [(5, 3), (20, 3), (40, 4), (10, 12), (30, 20)]
    for k, (x, y) in enumerate(pts):
        m[0, y, x] = 0.5 + 0.05 * k
    peaks = orc.nms(m, 1, 3, 0.05)
    assert peaks[0, 0, 0] == 5  # total is NOT clamped to max_peaks (nms_layer.cu:110)
    got = [(round(float(peaks[0, i, 0])), round(float(peaks[0, i, 1]))) for i in (1, 2, 3)]
    assert got == pts[:3]  # first max_peaks in raster order


def test_connect_recovers_planted_people_and_json():
    for model, (W, H) in ((0, (656, 368)), (1, (496, 368))):
        tabs = orc.model_tables(model)
        thr = orc.default_thresholds(model)
        mp = 64 if model == 0 else 20
        low, people = _synth.people_lowres(model, tabs, 3, H // 8, W // 8, seed=13)
        res = orc.imresize(low, W, H, 1.0, 0.3)[0]
        peaks = orc.nms(res, tabs[0], mp, thr["nms_threshold"])
        n, joints = orc.connect(model, res, peaks, mp, W, H, 1280, 720, thr)
        assert n == 3
        assert (joints[:n, :, 2] > 0).sum() == 3 * tabs[0]
        # joints come back in display coordinates (rtpose.cpp:1061-1062)
        got = sorted(float(j[1, 0]) for j in joints[:n])
        exp = sorted((p[1][0] * 8 + 3.5) * 1280 / W for p in people)
        assert np.allclose(got, exp, atol=12)
        js = orc.write_json(joints, n, tabs[0], 1.0)
        assert js.count(b"joints") == 3


def test_render_pose_oracle_sanity():
    """orc_render_pose (renderFunctions.cu restatement): no people -> the image comes back unchanged
    (float -> u8 of integers); people change pixels only inside their boxes; COCO and MPI tables."""
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, (120, 160, 3)).astype(np.uint8)
    for model, NP in ((0, 18), (1, 15)):
        assert np.array_equal(orc.render_pose(model, img, np.zeros((0, NP, 3), np.float32), 0), img)
        pose = np.zeros((1, NP, 3), np.float32)
        pose[0, :, 0] = 40 + rs.rand(NP) * 60
        pose[0, :, 1] = 30 + rs.rand(NP) * 50
        pose[0, :, 2] = 0.8
        out = orc.render_pose(model, img, pose, 1)
        changed = np.argwhere((out != img).any(-1))
        assert len(changed) > 100
        assert changed[:, 1].min() >= 40 - 60 and changed[:, 1].max() <= 100 + 60 and changed[:, 0].min() >= 0
        # parts below the threshold draw nothing
        pose[0, :, 2] = 0.0
        assert np.array_equal(orc.render_pose(model, img, pose, 1), img)
