"""Pin the CPU oracle against the reference's own known-answer tests.

Every case below restates a test the reference ships for the standard layers on the hot path (and cross-checks torch CPU).  The
stronger pins live in tests/test_ref_pin.py / tests/test_ref_golden.py: convolution, pooling, ImResize, Nms, connectLimbs*, JSON,
process_and_pad_image and the renderer are compared with the reference's OWN code (oracle/_ref/libref.so, compiled from
/root/reference by oracle/ref_recipe/build_ref.sh) and with its stored outputs.
"""
import numpy as np
import pytest
import torch

import _oracle as orc


def test_pool_forward_square():
    # src/caffe/test/test_pooling_layer.cpp:49-103  TestForwardSquare (2x2 kernel, stride 1 default)
    plane = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)
    x = np.tile(plane, (2, 2, 1, 1)).astype(np.float32)
    y = orc.maxpool(x, k=2, stride=1, pad=0)
    assert y.shape == (2, 2, 2, 4)
    exp = np.array([[9, 5, 5, 8], [9, 5, 5, 8]], np.float32)
    assert np.array_equal(y, np.tile(exp, (2, 2, 1, 1)))


def test_pool_ceil_mode_shape():
    # pooling_layer.cpp:90-106: ceil-mode output size; 2x2/s2 on odd sizes keeps the partial window
    x = np.arange(1 * 1 * 5 * 7, dtype=np.float32).reshape(1, 1, 5, 7)
    y = orc.maxpool(x, 2, 2, 0)
    assert y.shape == (1, 1, 3, 4)
    ref = torch.nn.functional.max_pool2d(torch.from_numpy(x), 2, 2, ceil_mode=True).numpy()
    assert np.array_equal(y, ref)
    # even sizes (every linevec resolution is a multiple of 16)
    x = np.random.RandomState(0).randn(2, 3, 16, 32).astype(np.float32)
    assert np.array_equal(orc.maxpool(x), torch.nn.functional.max_pool2d(torch.from_numpy(x), 2, 2).numpy())


def test_relu_property():
    # src/caffe/test/test_neuron_layer.cpp:208-221 TestReLU
    x = np.random.RandomState(1).randn(2, 3, 4, 5).astype(np.float32)
    y = orc.relu(x)
    assert (y >= 0).all()
    assert ((y == 0) | (y == x)).all()


def test_concat_channels():
    # src/caffe/test/test_concat_layer.cpp:143-167 TestForwardChannels
    rs = np.random.RandomState(2)
    a = rs.randn(2, 3, 6, 5).astype(np.float32)
    b = rs.randn(2, 2, 6, 5).astype(np.float32)
    y = orc.concat2(a, b)
    assert np.array_equal(y[:, :3], a) and np.array_equal(y[:, 3:], b)


@pytest.mark.parametrize("k,pad,stride", [(3, 0, 2), (1, 0, 1), (3, 1, 1), (7, 3, 1)])
def test_conv_against_caffe_conv(k, pad, stride):
    # src/caffe/test/test_convolution_layer.cpp:151-166,231-265 (TestSimpleConvolution: 3x3 s2,
    # Gaussian-filled 2x3x6x4 bottom, tolerance 1e-4 vs caffe_conv) and :443-468 (1x1); plus the
    # two pad=(k-1)/2 stride-1 shapes the linevec nets actually use.
    rs = np.random.RandomState(1701)
    x = rs.randn(2, 3, 6, 4).astype(np.float32)
    w = rs.randn(4, 3, k, k).astype(np.float32)
    b = np.full(4, 0.1, np.float32)
    y = orc.conv2d(x, w, b, pad, stride)
    ref = orc.conv2d_naive(x, w, b, pad, pad, stride, stride)
    assert y.shape == ref.shape
    np.testing.assert_allclose(y, ref, atol=1e-4, rtol=0)
    t = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride, pad).numpy()
    np.testing.assert_allclose(y, t, atol=1e-4, rtol=0)


def test_conv_sobel_separable():
    # src/caffe/test/test_convolution_layer.cpp:498-589 TestSobelConvolution: the 3x3 Sobel G_x
    # filter equals the [1 2 1]^T column filter (stride_h 2) followed by the [-1 0 1] row filter
    # (stride_w 2), tolerance 1e-4.
    rs = np.random.RandomState(3)
    x = rs.randn(2, 3, 6, 4).astype(np.float32)
    sob = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float32)
    w = np.tile(sob, (1, 3, 1, 1)).astype(np.float32)
    full = orc.conv2d(x, w, None, 0, 2)
    w1 = np.tile(np.array([[1], [2], [1]], np.float32), (1, 3, 1, 1)).astype(np.float32)
    s1 = orc.conv2d_naive(x, w1, None, 0, 0, 2, 1)
    w2 = np.array([[[[-1, 0, 1]]]], np.float32)
    s2 = orc.conv2d_naive(s1, w2, None, 0, 0, 1, 2)
    np.testing.assert_allclose(full, s2, atol=1e-4, rtol=0)


def test_conv_matches_torch_on_linevec_shapes():
    # independent cross-check (SURVEY.md §8c "Independent cross-check available here")
    rs = np.random.RandomState(4)
    for cin, cout, k in [(3, 64, 3), (185, 128, 7), (128, 38, 1)]:
        x = rs.randn(2, cin, 12, 20).astype(np.float32)
        w = (rs.randn(cout, cin, k, k) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)
        b = rs.uniform(-0.1, 0.1, cout).astype(np.float32)
        y = orc.conv2d(x, w, b, (k - 1) // 2)
        t = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, (k - 1) // 2).numpy()
        np.testing.assert_allclose(y, t, atol=2e-5, rtol=1e-5)


def test_net_topology_counts():
    # model/coco|mpi/pose_deploy_linevec.prototxt: 92 Convolution layers; SURVEY.md §8(a) conv table
    for model, nheat, npaf in [(0, 19, 38), (1, 16, 28)]:
        net = orc.Net(model)
        assert len(net.convs) == 92
        gflop = sum(2.0 * cout * cin * k * k * (46 * 82 if i >= 10 else 0) for i, (_, cin, cout, k) in enumerate(net.convs))
        names = [c[0] for c in net.convs]
        assert names[0] == "conv1_1" and names[-1] == "Mconv7_stage6_L2"
        assert net.convs[names.index("Mconv1_stage2_L1")][1] == 128 + nheat + npaf
        assert net.convs[-1][2] == nheat and net.convs[-2][2] == npaf
    # total conv GFLOP at 656x368 (COCO) = 484.634
    net = orc.Net(0)
    res = {}
    H, W = 368, 656
    hw = {"conv1": (H, W), "conv2": (H // 2, W // 2), "conv3": (H // 4, W // 4)}
    tot = 0.0
    for name, cin, cout, k in net.convs:
        h, w = hw.get(name[:5], (H // 8, W // 8))
        tot += 2.0 * cout * cin * k * k * h * w
    assert abs(tot / 1e9 - 484.634) < 0.01
