"""Deterministic post-processing / host-function cases shared by
  * tests/test_ref_pin.py      oracle == oracle/_ref (the reference's own code), bit for bit      [CPU, needs libref.so]
  * tools/make_ref_golden.py   writes the _ref outputs to tests/golden/ref_pin.npz               [this container]
  * tests/test_ref_golden.py   oracle == golden [CPU] and HIP engine == golden [GPU box, no /root/reference there]
Inputs are regenerated from seeds (numpy RandomState), only OUTPUTS are stored."""
import hashlib

import numpy as np

import _synth

THR = {0: dict(nms_threshold=0.05, inter_threshold=0.05, inter_min_above=9, min_subset_cnt=3, min_subset_score=0.4),   # rtpose.cpp:218-226
       1: dict(nms_threshold=0.2, inter_threshold=0.01, inter_min_above=8, min_subset_cnt=3, min_subset_score=0.4)}    # rtpose.cpp:212-217
DIMS = {0: (18, 64, 57), 1: (15, 20, 44)}  # parts, max_peaks, heat channels


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def crop_of(h, w, start, gap, n):
    """(padh, padw, oh, ow) of scale n inside the h x w low-res map: imresize_layer.cu:110-113 in its own float arithmetic
    (integer w/2, float32 products, floor)."""
    f = np.float32
    t = f(f(1) - f(start)) + f(f(n) * f(gap))
    padw = int(np.floor(f(w // 2) * t))
    padh = int(np.floor(f(h // 2) * t))
    return padh, padw, h - 2 * padh, w - 2 * padw


def scaled_people(model, tables, P, h, w, seed, N, start, gap):
    """Planted people whose scale n copy lives where the producer puts it (rtpose.cpp:353-368: the image resized by start - n gap,
    centre-padded): the same people, planted at the crop's size in the crop's window, zeros (+ tie-breaking noise) around it."""
    out = np.zeros((N, DIMS[model][2], h, w), np.float32)
    rs = np.random.RandomState(seed + 1000)
    out += (rs.rand(*out.shape).astype(np.float32) - 0.5) * 1e-3
    for n in range(N):
        padh, padw, oh, ow = crop_of(h, w, start, gap, n)
        out[n, :, padh:padh + oh, padw:padw + ow] = _synth.people_lowres(model, tables, P, oh, ow, seed=seed)[0][0]
    return out


# cases whose joints are scaled to another display resolution than 1280x720 (rtpose.cpp:1051-1073: peaks * DISP / NET)
CASE_DISP = {"coco_s080_n2_disp640x360": (640, 360), "coco_noise_s080_n1_disp1920x1080": (1920, 1080), "portrait_people3_s080_n2": (720, 1280)}


def disp_of(name):
    return CASE_DISP.get(name, (1280, 720))


def lowres_cases(tables):
    """name -> (model, lowres [N][C][h][w], net_w, net_h, start_scale, scale_gap)"""
    c = {}
    c["coco_noise_1s"] = (0, _synth.smooth_field(57, 46, 82, seed=3)[None], 656, 368, 1.0, 0.3)
    c["coco_noise_3s"] = (0, _synth.smooth_field(3 * 57, 46, 82, seed=3).reshape(3, 57, 46, 82), 656, 368, 1.0, 0.15)
    c["mpi_noise_1s"] = (1, _synth.smooth_field(44, 46, 62, seed=6)[None], 496, 368, 1.0, 0.3)
    c["small_noise_2s"] = (0, _synth.smooth_field(2 * 57, 12, 20, seed=7).reshape(2, 57, 12, 20), 160, 96, 1.0, 0.25)
    for P in (1, 5, 20):
        c[f"coco_people{P}"] = (0, _synth.people_lowres(0, tables[0], P, 46, 82, seed=20 + P)[0], 656, 368, 1.0, 0.3)
    c["coco_people5_3s"] = (0, _synth.people_lowres(0, tables[0], 5, 46, 82, seed=44, N=3)[0], 656, 368, 1.0, 0.15)
    c["mpi_people5"] = (1, _synth.people_lowres(1, tables[1], 5, 46, 62, seed=25)[0], 496, 368, 1.0, 0.3)
    # round 5: --start_scale != 1 (rtpose.cpp:68; ImResizeLayer::SetStartScale moves the crop of EVERY scale incl. n = 0,
    # imresize_layer.cu:110-113), portrait and large net resolutions, other display resolutions in connect (rtpose.cpp:1051-1073)
    noise = lambda N, C, h, w, seed: _synth.smooth_field(N * C, h, w, seed=seed).reshape(N, C, h, w)
    c["coco_noise_s080_n1"] = (0, noise(1, 57, 46, 82, 51), 656, 368, 0.8, 0.15)
    c["coco_noise_s080_n2"] = (0, noise(2, 57, 46, 82, 52), 656, 368, 0.8, 0.15)
    c["coco_noise_s080_n3"] = (0, noise(3, 57, 46, 82, 53), 656, 368, 0.8, 0.15)
    c["coco_noise_s065_n1"] = (0, noise(1, 57, 46, 82, 54), 656, 368, 0.65, 0.25)
    c["coco_noise_s065_n2"] = (0, noise(2, 57, 46, 82, 55), 656, 368, 0.65, 0.25)
    c["coco_noise_s065_n3"] = (0, noise(3, 57, 46, 82, 56), 656, 368, 0.65, 0.25)     # scale 2 = 0.15: a 14 x 8 crop
    c["coco_people5_s080_n3"] = (0, scaled_people(0, tables[0], 5, 46, 82, 57, 3, 0.8, 0.15), 656, 368, 0.8, 0.15)
    c["mpi_people5_s080_n2"] = (1, scaled_people(1, tables[1], 5, 46, 62, 58, 2, 0.8, 0.25), 496, 368, 0.8, 0.25)
    c["mpi_noise_s065_n2"] = (1, noise(2, 44, 46, 62, 59), 496, 368, 0.65, 0.15)
    c["portrait_noise_1s"] = (0, noise(1, 57, 82, 46, 60), 368, 656, 1.0, 0.3)
    c["portrait_people3_s080_n2"] = (0, scaled_people(0, tables[0], 3, 82, 46, 61, 2, 0.8, 0.15), 368, 656, 0.8, 0.15)
    c["large_noise_s080_n2"] = (0, noise(2, 57, 92, 164, 62), 1312, 736, 0.8, 0.15)
    c["coco_s080_n2_disp640x360"] = (0, scaled_people(0, tables[0], 5, 46, 82, 63, 2, 0.8, 0.15), 656, 368, 0.8, 0.15)
    c["coco_noise_s080_n1_disp1920x1080"] = (0, noise(1, 57, 46, 82, 64), 656, 368, 0.8, 0.15)
    return c


ROUND5_CASES = ["coco_noise_s080_n1", "coco_noise_s080_n2", "coco_noise_s080_n3", "coco_noise_s065_n1", "coco_noise_s065_n2", "coco_noise_s065_n3",
                "coco_people5_s080_n3", "mpi_people5_s080_n2", "mpi_noise_s065_n2", "portrait_noise_1s", "portrait_people3_s080_n2",
                "large_noise_s080_n2", "coco_s080_n2_disp640x360", "coco_noise_s080_n1_disp1920x1080"]


def clamp_counts(peaks, max_peaks):
    """The reference loops over nA/nB = peaks[part][0] slots although only max_peaks exist (rtpose.cpp:843,897 vs
    nms_layer.cu:70,110): out of contract.  Oracle and engine clamp inside connect; the reference gets clamped counts."""
    p = peaks.copy()
    p[:, 0, 0] = np.minimum(p[:, 0, 0], max_peaks)
    return p


def chain(impl, model, low, net_w, net_h, start, gap, disp=(1280, 720)):
    """ImResize -> Nms -> connect through `impl` (tests/_oracle.py or tests/_ref.py: same call shapes)."""
    parts, max_peaks, _ = DIMS[model]
    thr = THR[model]
    res = impl.imresize(np.ascontiguousarray(low, np.float32), net_w, net_h, start, gap)[0]
    peaks = impl.nms(res, parts, max_peaks, thr["nms_threshold"])
    try:
        n, joints = impl.connect(model, res, clamp_counts(peaks, max_peaks), max_peaks, net_w, net_h, disp[0], disp[1], thr)
    except (RuntimeError, AssertionError):
        # the reference CHECK-fails (rtpose.cpp:928 CHECK_GE(mx, 0)); the oracle returns its error code; the engine RTP_ERANGE.  Reached by
        # portrait nets: writeResultKernel bounds the centroid window's ROWS by `width` (nms_layer.cu:79), so a peak at y >= width + 3 of
        # a net that is taller than wide divides 0 by 0, and connect rounds the NaN coordinate to INT_MIN.
        return res, peaks, -1, np.zeros((0, parts, 3), np.float32)
    return res, peaks, n, joints[:n].copy()


def tie_case():
    """Constant PAF => every candidate ties at one score: the greedy result depends on std::sort's order of equals."""
    H, W = 368, 656
    res = np.zeros((57, H, W), np.float32)
    res[19:] = 0.70710677
    rs = np.random.RandomState(11)
    peaks = np.zeros((18, 65, 3), np.float32)
    for p in range(18):
        n = int(rs.randint(3, 30))
        peaks[p, 0, 0] = n
        base = rs.uniform(20, 200)
        for i in range(1, n + 1):
            peaks[p, i] = (base + 9 * i + p * 3, base * 0.5 + 9 * i + p * 3, rs.uniform(0.3, 0.9))
    return res, peaks


def single_sided_case():
    res = np.zeros((57, 368, 656), np.float32)
    peaks = np.zeros((18, 65, 3), np.float32)
    peaks[1, 0, 0] = 2
    peaks[1, 1] = (100, 100, 0.9)
    peaks[1, 2] = (300, 120, 0.8)
    peaks[0, 0, 0] = 1          # a nose with no neck candidate on limb (1,0): nA != 0, nB != 0 on that limb only
    peaks[0, 1] = (110, 60, 0.7)
    return res, peaks


def stale_peaks(parts, max_peaks, seed=9):
    rs = np.random.RandomState(seed)
    return rs.uniform(-5, 5, size=(parts, max_peaks + 1, 3)).astype(np.float32)


def json_cases():
    rs = np.random.RandomState(13)
    out = []
    for model, n in ((0, 0), (0, 1), (0, 7), (1, 3)):
        parts = DIMS[model][0]
        j = rs.uniform(0, 1300, size=(max(n, 1), parts, 3)).astype(np.float32)
        j[..., 2] = rs.uniform(0, 1, size=j.shape[:2])
        if n:
            j[0, 0] = (0.0, 0.0, 0.0)
            j[0, 1] = (1e-7, 123456.789, 1.0)
            j[-1, -1] = (1234.5678, 0.000123456, 0.99999994)
        for scale in (1.0, 1.5, 0.5625, 2.0 / 3.0):
            out.append((model, n, j[:n], np.float32(scale)))
    return out


def pad_cases():
    rs = np.random.RandomState(17)
    out = []
    for (ow, oh, tw, th) in ((656, 368, 656, 368), (560, 320, 656, 368), (33, 17, 48, 32), (1, 1, 16, 16)):
        img = rs.randint(0, 256, size=(oh, ow, 3)).astype(np.uint8)
        for normalize in (0, 1):
            out.append((img, tw, th, normalize))
    return out


def blocky_image(rs, w, h):
    """display image made of 8x8 colour blocks (compresses well in the golden fixture, unlike per-pixel noise)"""
    small = rs.randint(0, 256, size=(h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(np.repeat(small, 8, axis=0), 8, axis=1)[:h, :w])


def render_cases():
    """(name, model, display image u8 [h][w][3], joints [n][parts][3] in display coordinates, n, googly): people of every size class
    of render_pose_coco_parts (scale factor 0.33 .. 1), missing parts, joints outside the image, overlapping people."""
    out = []
    for name, model, w, h, n, seed, googly in (("coco", 0, 640, 360, 6, 31, 0), ("coco_googly", 0, 640, 360, 4, 32, 1), ("coco_odd", 0, 333, 201, 3, 33, 0),
                                               ("mpi", 1, 640, 360, 5, 34, 0), ("coco_none", 0, 320, 180, 0, 35, 0)):
        rs = np.random.RandomState(seed)
        img = blocky_image(rs, w, h)
        pose = _synth.COCO_POSE if model == 0 else _synth.MPI_POSE
        parts = DIMS[model][0]
        j = np.zeros((max(n, 1), parts, 3), np.float32)
        for p in range(n):
            size = (0.12, 0.3, 0.55, 0.9, 1.3, 0.2)[p % 6] * h       # small (scale clamp 0.33) .. larger than the 200 px threshold
            cx, cy = rs.uniform(0.1, 0.9) * w, rs.uniform(-0.1, 0.5) * h
            for k, (ux, uy) in pose.items():
                j[p, k] = (cx + (ux - 0.5) * size * 0.6 + rs.uniform(-2, 2), cy + uy * size + rs.uniform(-2, 2), rs.uniform(0.05, 1.0))
            for k in rs.choice(parts, size=3, replace=False):          # parts that were not found
                j[p, k] = (0.0, 0.0, 0.0)
        out.append((name, model, img, j[:n].copy() if n else j[:0].copy(), n, googly))
    return out


def view_cases(tables):
    """(name, model, display image, net-resolution maps [C][net_h][net_w], part_to_show values).  Maps = ImResize of planted people
    plus noise (values beyond [0,1] and [-1,1] exercise the colour-map clamps); canvases that are not multiples of the net size."""
    import _oracle as orc
    out = []
    for name, model, w, h, net_w, net_h, seed, parts in (("coco", 0, 320, 180, 160, 96, 41, (1, 7, 18, 19, 20, 21, 30, 39)),
                                                         ("coco_odd", 0, 333, 201, 160, 96, 42, (2, 19, 20, 25)),
                                                         ("mpi", 1, 320, 180, 128, 96, 43, (1, 15, 16, 17, 44))):
        rs = np.random.RandomState(seed)
        img = blocky_image(rs, w, h)
        C = DIMS[model][2]
        low = _synth.people_lowres(model, tables[model], 3, net_h // 8, net_w // 8, seed=seed)[0] + 0.6 * _synth.smooth_field(C, net_h // 8, net_w // 8, seed=seed + 1)[None]
        maps = orc.imresize(np.ascontiguousarray(low, np.float32), net_w, net_h, 1.0, 0.3)[0]
        out.append((name, model, img, maps, parts))
    return out


def conv_cases():
    """(name, x [N][Cin][H][W], w [Cout][Cin][k][k], b or None, pad, stride): the shapes of the reference's own convolution tests
    (test_convolution_layer.cpp:151-166, 231-265 TestSimpleConvolution: 2x3x6x4 bottom, 4 outputs, 3x3 stride 2; :443-468 1x1) and the
    three kinds of layers on the linevec path at reduced spatial size where the naive loop would take minutes: the input convolution
    (3 -> 64, k 3), a stage-entry 7x7 over the 185-channel concat, a branch-final 1x1 (128 -> 38) at the full 46x82 low-res size."""
    rs = np.random.RandomState(1701)
    g = lambda *s: rs.randn(*s).astype(np.float32)
    yield "gtest_simple_k3_s2", g(2, 3, 6, 4), g(4, 3, 3, 3), np.full(4, 0.1, np.float32), 0, 2
    yield "gtest_1x1", g(2, 3, 6, 4), g(4, 3, 1, 1), np.full(4, 0.1, np.float32), 0, 1
    yield "gtest_no_bias_pad1", g(1, 2, 5, 7), g(3, 2, 3, 3), None, 1, 1
    yield "conv1_1_3_64_k3", (rs.randint(0, 256, (2, 3, 16, 24)) / 256.0 - 0.5).astype(np.float32), (g(64, 3, 3, 3) * np.float32(np.sqrt(2 / 27))), rs.uniform(-0.1, 0.1, 64).astype(np.float32), 1, 1
    yield "Mconv1_185_128_k7", np.maximum(g(1, 185, 8, 10), 0), g(128, 185, 7, 7) * np.float32(np.sqrt(2 / (185 * 49))), rs.uniform(-0.1, 0.1, 128).astype(np.float32), 3, 1
    yield "Mconv7_128_38_k1", np.maximum(g(1, 128, 46, 82), 0), g(38, 128, 1, 1) * np.float32(np.sqrt(2 / 128)), rs.uniform(-0.1, 0.1, 38).astype(np.float32), 0, 1


def pool_cases():
    """(name, x, k, stride, pad): TestForwardSquare's plane (test_pooling_layer.cpp:49-103), the 2x2 / stride 2 pooling of the linevec
    trunk on even and on odd sizes (ceil mode keeps the partial window), a padded case (the clip of pooling_layer.cpp:94-105)."""
    rs = np.random.RandomState(7)
    plane = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)
    yield "gtest_square_k2_s1", np.tile(plane, (2, 2, 1, 1)).astype(np.float32), 2, 1, 0
    yield "linevec_64ch_even", rs.randn(2, 64, 32, 48).astype(np.float32), 2, 2, 0
    yield "odd_5x7_ceil", rs.randn(1, 3, 5, 7).astype(np.float32), 2, 2, 0
    yield "k3_s2_pad1", rs.randn(1, 2, 9, 10).astype(np.float32), 3, 2, 1
