"""Row a1 (producer-side pre-processing) and the rtpose.bin CLI.  The OpenCV primitives are restated
(OpenCV absent, unpinned by the reference): these tests check them against exact-area / identity
properties, not against OpenCV."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "caffe_rtpose_amd", "rtpose.bin")


def _exact_area(img, dw, dh):
    """Area-weighted mean of the source pixels each destination pixel covers, in float64."""
    sh, sw, _ = img.shape

    def weights(s, d):
        W = np.zeros((d, s))
        sc = s / d
        for i in range(d):
            a, b = i * sc, (i + 1) * sc
            for j in range(int(np.floor(a)), min(int(np.ceil(b)), s)):
                W[i, j] = max(0.0, min(b, j + 1) - max(a, j)) / sc
        return W

    Wy, Wx = weights(sh, dh), weights(sw, dw)
    return np.einsum("ij,jkc,lk->ilc", Wy, img.astype(np.float64), Wx)


def test_resize_area_matches_exact_area_integration():
    import caffe_rtpose_amd as r
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (72, 128, 3)).astype(np.uint8)
    for dw, dh in [(64, 36), (48, 32), (66, 37), (128, 72)]:
        got = r.resize_area(img, dw, dh).astype(np.float64)
        ref = _exact_area(img, dw, dh)
        assert np.abs(got - ref).max() <= 0.51  # round-to-nearest of the same area mean
    const = np.full((90, 160, 3), 77, np.uint8)
    assert (r.resize_area(const, 82, 46) == 77).all()
    # the net's own geometry: 1280x720 display -> 656x368 (rtpose.cpp:358-366)
    big = rs.randint(0, 256, (720, 1280, 3)).astype(np.uint8)
    out = r.resize_area(big, 656, 368)
    assert out.shape == (368, 656, 3)
    assert abs(float(out.mean()) - float(big.mean())) < 0.5


def test_warp_display_scale_and_border():
    import caffe_rtpose_amd as r
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (48, 64, 3)).astype(np.uint8)
    same, s = r.warp_display(img, 64, 48)
    assert s == 1.0 and np.array_equal(same, img)                 # scale 1: cubic at integer positions is identity
    out, s = r.warp_display(img, 128, 120)                        # fit: limited by width -> 2x, black band below
    assert s == 2.0 and (out[100:] == 0).all()                     # cubic taps reach 2 source rows past the edge
    assert np.array_equal(out[0:96:2, 0:128:2], img)              # even output pixels sample source pixels exactly
    assert r.lib.rtp_display_fit_scale(640, 480, 1280, 720) == 1.5  # BASELINE config 1


def test_display_fit_warp_is_a_copy_when_the_frame_has_the_display_size():
    """rtpose.cpp:324-336 warps every frame to the display resolution with INTER_CUBIC.  At fit scale 1 the warp samples at integer
    positions, where the cubic weights are (0, 1, 0, 0): a copy.  The engine skips the warp kernel for such frames (a 720p video at the
    default --resolution 1280x720: 20 us per frame) and reads the frame itself as the display image; this is the host restatement of
    OpenCV's arithmetic (tests/_cvref.py pins it) saying that nothing changes, for any content."""
    import caffe_rtpose_amd as r
    rs = np.random.RandomState(11)
    for w, h in ((1280, 720), (640, 480), (333, 517), (16, 16)):
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        out, scale = r.warp_display(img, w, h)
        assert scale == 1.0 and np.array_equal(out, img), (w, h)
    out, scale = r.warp_display(np.full((480, 640, 3), 255, np.uint8), 1280, 720)   # (a frame that does NOT fit is resampled: scale 1.5, zero border)
    assert scale == 1.5 and out.shape == (720, 1280, 3) and (out[:, 960:] == 0).all()


def test_preprocess_frame_layout():
    import caffe_rtpose_amd as r
    img = r.synth_frame(640, 480, 3, seed=1)
    x, disp, fs = r.preprocess_frame(img, 1280, 720, 656, 368, num_scales=3, start_scale=1.0, scale_gap=0.15)
    assert x.shape == (3, 3, 368, 656) and disp.shape == (720, 1280, 3) and fs == 1.5
    # scale i is centred and zero padded (rtpose.cpp:244-268): crop sizes 656x368, 560x320, 464x272
    for i, (tw, th) in enumerate([(656, 368), (560, 320), (464, 272)]):
        pw, ph = (656 - tw) // 2, (368 - th) // 2
        assert (x[i, :, :ph, :] == 0).all() and (x[i, :, :, :pw] == 0).all()
        inner = x[i, :, ph:ph + th, pw:pw + tw]
        assert inner.min() >= -0.5 and inner.max() < 0.5
        assert np.array_equal(inner, r.process_and_pad_image(r.resize_area(disp, tw, th), tw, th, 1))
    with pytest.raises(r.RtpError):
        r.preprocess_frame(img, 1280, 720, 656, 368, num_scales=1, start_scale=1.2)  # CHECK_LE(target_width, NET_RESOLUTION_WIDTH)


def test_pyramid_level_sizes_follow_the_references_float_arithmetic():
    """`float scale = START_SCALE - i*SCALE_GAP; target_width = 16 * ceil(NET_RESOLUTION_WIDTH * scale / 16)` (rtpose.cpp:360-361): the flags are
    doubles, `scale` is a float, and `int * float / int` is FLOAT arithmetic — at scales where net * s lands on a multiple of 16 (0.6 of 320,
    0.75 of 656 x 368 ...) double arithmetic gives another level size.  Sweep of start scales / gaps / nets: the padded region the product
    writes has exactly the reference's size, centred (process_and_pad_image, :239-269)."""
    import caffe_rtpose_amd as r
    img = np.full((45, 80, 3), 200, np.uint8)
    checked = exact_multiples = 0
    for W, H in ((160, 96), (320, 240), (656, 368)):
        for start in (1.0, 0.95, 0.9, 0.85, 0.8, 0.75, 0.7, 0.65, 0.6, 0.55, 0.5):
            for gap in (0.1, 0.15, 0.2, 0.25, 0.3, 0.4):
                sizes = []
                for i in range(4):
                    s32 = np.float32(start - i * gap)
                    if not s32 > 0:
                        break
                    tw = int(16 * np.ceil(np.float32(np.float32(W) * s32) / np.float32(16)))
                    th = int(16 * np.ceil(np.float32(np.float32(H) * s32) / np.float32(16)))
                    if tw > W or th > H:
                        break
                    sizes.append((tw, th))
                    exact_multiples += int(np.ceil(float(W) * float(s32) / 16) != float(np.ceil(np.float32(np.float32(W) * s32) / np.float32(16))))   # double vs float arithmetic
                if not sizes:
                    continue
                x, _, _ = r.preprocess_frame(img, 80, 45, W, H, len(sizes), start, gap)
                for i, (tw, th) in enumerate(sizes):
                    inside = x[i, 0] != 0
                    ys, xs = np.nonzero(inside)
                    assert (xs.min(), xs.max() + 1, ys.min(), ys.max() + 1) == ((W - tw) // 2, (W - tw) // 2 + tw, (H - th) // 2, (H - th) // 2 + th), (W, H, start, gap, i, tw, th)
                    assert inside.sum() == tw * th
                    checked += 1
    assert checked > 300 and exact_multiples >= 1     # the sweep contains levels where double and float arithmetic disagree


def test_ppm_and_bmp_loaders(tmp_path):
    import caffe_rtpose_amd as r
    rs = np.random.RandomState(2)
    bgr = rs.randint(0, 256, (5, 7, 3)).astype(np.uint8)
    with open(tmp_path / "a.ppm", "wb") as f:
        f.write(b"P6\n# comment\n7 5\n255\n" + bgr[:, :, ::-1].tobytes())
    assert np.array_equal(r.load_image(tmp_path / "a.ppm"), bgr)
    stride = (7 * 3 + 3) & ~3
    rows = b"".join(bgr[y].tobytes() + b"\0" * (stride - 21) for y in range(4, -1, -1))
    hdr = b"BM" + (54 + len(rows)).to_bytes(4, "little") + b"\0\0\0\0" + (54).to_bytes(4, "little")
    hdr += (40).to_bytes(4, "little") + (7).to_bytes(4, "little") + (5).to_bytes(4, "little") + (1).to_bytes(2, "little") + (24).to_bytes(2, "little")
    hdr += (0).to_bytes(4, "little") + len(rows).to_bytes(4, "little") + b"\0" * 16
    open(tmp_path / "b.bmp", "wb").write(hdr + rows)
    assert np.array_equal(r.load_image(tmp_path / "b.bmp"), bgr)
    open(tmp_path / "c.jpg", "wb").write(b"\xff\xd8\xff\xe0junk")
    with pytest.raises(r.RtpError):
        r.load_image(tmp_path / "c.jpg")


def test_cli_flag_errors_and_no_device_exit_code(tmp_path):
    assert os.path.exists(BIN), "build rtpose.bin first (__graft_entry__.build())"
    assert subprocess.run([BIN, "--help"], capture_output=True).returncode == 0
    p = subprocess.run([BIN, "--bogus_flag", "1"], capture_output=True)
    assert p.returncode == 1 and b"unknown command line flag" in p.stderr
    p = subprocess.run([BIN, "--resolution", "abc", "--video", "synthetic:64x48:1"], capture_output=True)
    assert p.returncode == 1 and b"resolution format" in p.stderr
    p = subprocess.run([BIN], capture_output=True)
    assert p.returncode == 1 and b"camera" in p.stderr
    # --start_scale / --scale_gap / --net_resolution out of contract (the reference CHECKs, rtpose.cpp:363): an error exit with the reason,
    # with or without a GPU (argument errors come before the device is touched)
    for bad in (["--start_scale", "1.3"], ["--start_scale", "0.5", "--scale_gap", "0.25", "--num_scales", "3"], ["--net_resolution", "100x64"]):
        p = subprocess.run([BIN, "--video", "synthetic:64x48:2", "--model", "coco", "--net_resolution", "64x48", "--write_json", str(tmp_path / "bad")] + bad, capture_output=True)
        assert p.returncode == 1 and (b"does not fit the net resolution" in p.stderr or b"multiples of 16" in p.stderr), (bad, p.stderr[-300:])
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([BIN, "--video", "synthetic:64x48:2", "--model", "coco", "--net_resolution", "64x48",
                            "--write_json", str(tmp_path / "js")], capture_output=True)
        assert p.returncode == 1 and b"no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_cli_json_matches_library_path(tmp_path):
    """rtpose.bin on a synthetic video writes, for every frame, exactly the JSON the library path
    produces for the same frame (and that the oracle's writer produces from the same joints)."""
    import caffe_rtpose_amd as r
    import _oracle as orc
    out = tmp_path / "js"
    p = subprocess.run([BIN, "--video", "synthetic:640x480:6:5", "--model", "coco", "--net_resolution", "160x96", "--resolution", "320x240",
                        "--num_scales", "2", "--scale_gap", "0.25", "--write_json", str(out), "--no_frame_drops", "--no_display", "--num_gpu", "1"],
                       capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    files = sorted(os.listdir(out))
    assert files == [f"frame{i:06d}.json" for i in range(6)]
    e = r.Engine(r.Config(net_w=160, net_h=96, num_scales=2, scale_gap=0.25, disp_w=320, disp_h=240, frames_in_flight=1))
    for i in range(6):
        img = r.synth_frame(640, 480, i, seed=5)
        x, _, fs = r.preprocess_frame(img, 320, 240, 160, 96, 2, 1.0, 0.25)
        d = e.forward_debug(x)
        want = r.format_json(d["joints"], d["num_people"], 18, fs)
        assert want == orc.write_json(d["joints"], d["num_people"], 18, fs)
        assert open(out / files[i], "rb").read() == want
    e.close()


@pytest.mark.gpu
def test_cli_start_scale_and_a_display_smaller_than_the_net(tmp_path):
    """--start_scale (rtpose.cpp:68) below 1 with several scales, and --resolution smaller than --net_resolution (the first pyramid level is
    ENLARGED: OpenCV's area-mode bilinear kernel, on the device since round 5): the CLI's JSON equals the library path fed by the HOST
    restatement of the producer (which equals the independent OpenCV restatement, tests/_cvref.py); out-of-contract scales exit with an error."""
    import caffe_rtpose_amd as r
    out = tmp_path / "js"
    flags = ["--model", "coco", "--net_resolution", "160x96", "--resolution", "128x80", "--num_scales", "3", "--start_scale", "0.9", "--scale_gap", "0.2",
             "--no_frame_drops", "--no_display", "--num_gpu", "1"]
    p = subprocess.run([BIN, "--video", "synthetic:640x480:5:7", "--write_json", str(out)] + flags, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    e = r.Engine(r.Config(net_w=160, net_h=96, num_scales=3, start_scale=0.9, scale_gap=0.2, disp_w=128, disp_h=80, frames_in_flight=1))
    people = 0
    for i in range(5):
        img = r.synth_frame(640, 480, i, seed=7)
        x, _, fs = r.preprocess_frame(img, 128, 80, 160, 96, 3, 0.9, 0.2)      # levels 0.9 (160x96 from 128x80: enlarged), 0.7, 0.5
        d = e.forward_debug(x)
        people += d["num_people"]
        assert open(out / f"frame{i:06d}.json", "rb").read() == r.format_json(d["joints"], d["num_people"], 18, fs), i
    e.close()
    for bad in (["--start_scale", "1.3"], ["--start_scale", "0.5", "--scale_gap", "0.25"], ["--start_scale", "0"]):   # a level above the net / a level of scale 0
        q = subprocess.run([BIN, "--video", "synthetic:640x480:2:7", "--write_json", str(tmp_path / "bad")] + [a for a in flags if a not in ("--start_scale", "0.9", "--scale_gap", "0.2")] + bad,
                           capture_output=True, timeout=120)
        assert q.returncode != 0 and b"does not fit the net resolution" in q.stderr, (bad, q.stderr[-300:])


def test_cpp_net_api_mirror_compiles_and_links(tmp_path):
    """csrc/host_api.h (rtpose::Net & friends, the names rtpose.cpp uses) builds against the C-ABI."""
    src = tmp_path / "use.cpp"
    src.write_text('''
#include "caffe_rtpose_amd/csrc/host_api.h"
int main(int argc, char**) {
  if (argc > 100) {  // never runs: link/compile check only
    rtpose::Net net("model/coco/pose_deploy_linevec.prototxt", rtpose::TEST, 0);
    net.CopyTrainedLayersFrom("model/coco/pose_iter_440000.caffemodel");
    net.blobs()[0]->Reshape({1, 3, 368, 656});
    auto resize = net.layer_by_name<rtpose::ImResizeLayer>("resize");
    resize->SetStartScale(1.f); resize->SetScaleGap(0.3f);
    net.Reshape();
    auto nms = net.layer_by_name<rtpose::NmsLayer>("nms");
    nms->SetThreshold(0.05f);
    net.ForwardFrom(0);
    return nms->GetMaxPeaks() + nms->GetNumParts() + (int)net.blob_by_name("resized_map")->mutable_cpu_data()[0];
  }
  return 0;
}
''')
    exe = tmp_path / "use"
    subprocess.check_call(["g++", "-std=c++17", "-I", ROOT, "-o", str(exe), str(src), "-L", os.path.join(ROOT, "caffe_rtpose_amd"),
                           "-lrtpose_mi355x", "-Wl,-rpath," + os.path.join(ROOT, "caffe_rtpose_amd")])
    assert subprocess.run([str(exe)]).returncode == 0


def test_cli_video_file_errors():
    p = subprocess.run([BIN, "--video", "/nonexistent/clip.y4m", "--model", "coco"], capture_output=True)
    assert p.returncode == 1 and b"Couldn't open video file" in p.stderr
    png = os.path.join(ROOT, "tests", "golden", "codecs", "p_rgb8.png")
    p = subprocess.run([BIN, "--video", png, "--model", "coco"], capture_output=True)
    assert p.returncode == 1 and b"Y4M" in p.stderr


@pytest.mark.gpu
def test_cli_image_dir_and_video_files_match_library_path(tmp_path):
    """--image_dir with JPEG/PNG files and --video with a raw MJPEG stream: one JSON per image stem /
    frame number, byte-identical to decoding + submitting the same frames through the library."""
    import shutil
    import caffe_rtpose_amd as r
    gold = os.path.join(ROOT, "tests", "golden", "codecs")
    d = tmp_path / "imgs"
    d.mkdir()
    names = ["j420_q75.jpg", "j444_q90.jpg", "p_rgb8.png"]
    for n in names:
        shutil.copy(os.path.join(gold, n), d / n)
    out = tmp_path / "js"
    common = ["--model", "coco", "--net_resolution", "160x96", "--resolution", "320x240", "--no_frame_drops", "--no_display", "--num_gpu", "1"]
    p = subprocess.run([BIN, "--image_dir", str(d), "--write_json", str(out)] + common, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    assert sorted(os.listdir(out)) == sorted(n.rsplit(".", 1)[0] + ".json" for n in names)
    e = r.Engine(r.Config(net_w=160, net_h=96, disp_w=320, disp_h=240, frames_in_flight=1))

    def want_json(img):
        fs = e.submit_frame(img, tag=1)
        _, n, joints = e.collect()
        full = np.zeros((96, 18, 3), np.float32)
        full[:n] = joints
        return r.format_json(full, n, 18, fs)

    for n in names:
        assert open(out / (n.rsplit(".", 1)[0] + ".json"), "rb").read() == want_json(r.load_image(d / n)), n
    # raw MJPEG "video": frames 0..2 -> frame%06d.json
    clip = tmp_path / "clip.mjpeg"
    a = open(os.path.join(gold, "j420_q75.jpg"), "rb").read()
    clip.write_bytes(a + a + a)
    out2 = tmp_path / "js2"
    p = subprocess.run([BIN, "--video", str(clip), "--write_json", str(out2)] + common, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    assert sorted(os.listdir(out2)) == [f"frame{i:06d}.json" for i in range(3)]
    w = want_json(r.load_image(os.path.join(gold, "j420_q75.jpg")))
    for i in range(3):
        assert open(out2 / f"frame{i:06d}.json", "rb").read() == w
    e.close()


@pytest.mark.gpu
def test_cli_write_frames_jpeg_equals_library_render(tmp_path):
    """--write_frames: frame%06d.jpg = cv::imwrite(quality 98) of the rendered display frame, byte for byte
    what the library path produces (render + rtp_encode_jpeg), and decodable."""
    import caffe_rtpose_amd as r
    out = tmp_path / "frames"
    p = subprocess.run([BIN, "--video", "synthetic:640x480:3:5", "--model", "coco", "--net_resolution", "160x96", "--resolution", "320x240",
                        "--write_frames", str(out), "--no_frame_drops", "--no_display", "--num_gpu", "1"], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    assert sorted(os.listdir(out)) == [f"frame{i:06d}.jpg" for i in range(3)]
    e = r.Engine(r.Config(net_w=160, net_h=96, disp_w=320, disp_h=240, frames_in_flight=1, render=1))
    for i in range(3):
        e.submit_frame(r.synth_frame(640, 480, i, seed=5), tag=i)
        _, n, joints, img = e.collect_rendered()
        data = open(out / f"frame{i:06d}.jpg", "rb").read()
        assert data == r.encode_jpeg(img, 98)
        assert r.decode_image(data).shape == (240, 320, 3)
    e.close()
    p = subprocess.run([BIN, "--video", "synthetic:64x48:1", "--model", "coco", "--write_frames", str(tmp_path / "x"), "--host_preprocess"], capture_output=True)
    assert p.returncode == 1 and b"--write_frames needs" in p.stderr
    # --part_to_show N: the files hold the heat-map / PAF view render() draws (rtpose.cpp:270-299) = rtp_config.render = 1 + N in the library
    out2 = tmp_path / "views"
    p = subprocess.run([BIN, "--video", "synthetic:640x480:2:5", "--model", "coco", "--net_resolution", "160x96", "--resolution", "320x240",
                        "--write_frames", str(out2), "--part_to_show", "19", "--no_frame_drops", "--no_display", "--num_gpu", "1"], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    e = r.Engine(r.Config(net_w=160, net_h=96, disp_w=320, disp_h=240, frames_in_flight=1, render=1 + 19))
    for i in range(2):
        e.submit_frame(r.synth_frame(640, 480, i, seed=5), tag=i)
        _, n, joints, img = e.collect_rendered()
        data = open(out2 / f"frame{i:06d}.jpg", "rb").read()
        assert data == r.encode_jpeg(img, 98)
        assert data != open(out / f"frame{i:06d}.jpg", "rb").read()
    e.close()
    p = subprocess.run([BIN, "--video", "synthetic:64x48:1", "--model", "coco", "--write_frames", str(tmp_path / "y"), "--part_to_show", "40", "--no_display"], capture_output=True)
    assert p.returncode != 0   # outside the COCO model's maps


def test_cli_reorderer_order_drops_and_window(tmp_path):
    """Row a12 without a GPU: the CLI's re-orderer (buffer_and_order, rtpose.cpp:1214-1273) on hand-fed frames."""
    exe = tmp_path / "reorder_check"
    src = os.path.join(ROOT, "tests", "helpers", "reorder_check.cpp")
    libdir = os.path.join(ROOT, "caffe_rtpose_amd")
    p = subprocess.run(["g++", "-O1", "-std=c++17", src, "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lrtpose_mi355x",
                        "-Wl,-rpath," + libdir, "-lpthread", "-o", str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok ") == 5


def test_cli_validates_pipeline_depth_flags():
    """--frames_in_flight 0 used to hang (the worker never pops, the producer spins on back-pressure): now a flag error."""
    for flag, val in (("--frames_in_flight", "0"), ("--batch_frames", "0"), ("--batch_frames", "99")):
        p = subprocess.run([BIN, "--video", "synthetic:64x48:2", "--model", "coco", flag, val], capture_output=True, timeout=30)
        assert p.returncode == 1 and flag[2:].encode() in p.stderr
    p = subprocess.run([BIN, "--video", "synthetic:64x48:2", "--model", "coco", "--num_gpu", "2", "--devices", "0"], capture_output=True, timeout=30)
    assert p.returncode == 1 and b"--devices" in p.stderr


def _worker_counts(stderr):
    import re
    return [int(m.group(1)) for m in re.finditer(rb"worker \d+ \(GPU \d+\) processed (\d+) frames", stderr)]


@pytest.mark.gpu
def test_cli_two_workers_share_one_queue(tmp_path):
    """--num_gpu 2 (both workers on device 0 through the --devices hook): two engines pull from the ONE input queue
    (rtpose.cpp:1463-1472, 1107), every frame is written once, the JSON equals the one-worker run byte for byte."""
    outs = []
    for ngpu, extra in ((1, []), (2, ["--devices", "0,0"])):
        out = tmp_path / f"js{ngpu}"
        p = subprocess.run([BIN, "--video", "synthetic:640x480:24:7", "--model", "coco", "--net_resolution", "160x96", "--resolution", "320x240",
                            "--write_json", str(out), "--no_frame_drops", "--no_display", "--num_gpu", str(ngpu)] + extra, capture_output=True, timeout=600)
        assert p.returncode == 0, p.stderr.decode()
        counts = _worker_counts(p.stderr)
        assert len(counts) == ngpu and sum(counts) == 24
        if ngpu == 2:
            assert min(counts) > 0, f"one worker never got a frame: {counts}"
        outs.append(out)
    files = sorted(os.listdir(outs[0]))
    assert files == [f"frame{i:06d}.json" for i in range(24)] == sorted(os.listdir(outs[1]))
    for f in files:
        assert open(outs[0] / f, "rb").read() == open(outs[1] / f, "rb").read()


@pytest.mark.gpu
def test_cli_four_engines_seven_in_flight_on_one_device(tmp_path):
    """What the first 8-GPU run does, as far as one GPU can show it: --num_gpu 4 --devices 0,0,0,0 at the benched
    resolution with 7 frames in flight in batches of 2 per engine — four engines created, calibration-free, graph-captured in four
    threads AT ONCE, 4 x 4 batch contexts and 4 x 8 streams on one device, --share_weights copies worker 0's packed arena into the
    other three.  Every frame is written once and the JSON equals the one-worker run byte for byte (per-frame results do not depend on
    which engine or which batch a frame lands in)."""
    outs = []
    for ngpu, extra in ((1, []), (4, ["--devices", "0,0,0,0", "--share_weights"])):
        out = tmp_path / f"js{ngpu}"
        p = subprocess.run([BIN, "--video", "synthetic:640x480:56:9", "--model", "coco", "--net_resolution", "656x368", "--write_json", str(out), "--no_frame_drops",
                            "--no_display", "--num_gpu", str(ngpu), "--frames_in_flight", "7", "--batch_frames", "2"] + extra, capture_output=True, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        counts = _worker_counts(p.stderr)
        assert len(counts) == ngpu and sum(counts) == 56, p.stderr.decode()[-2000:]
        if ngpu == 4:
            assert min(counts) > 0, f"a worker never got a frame: {counts}"
            assert b"share_weights: 3 worker(s) took worker 0's packed weights" in p.stderr
        outs.append(out)
    files = sorted(os.listdir(outs[0]))
    assert files == [f"frame{i:06d}.json" for i in range(56)] == sorted(os.listdir(outs[1]))
    for f in files:
        assert open(outs[0] / f, "rb").read() == open(outs[1] / f, "rb").read(), f


@pytest.mark.gpu
def test_cli_frame_drops_and_no_frame_drops(tmp_path):
    """processFrame drops a frame that waited > 0.1 s for a GPU (rtpose.cpp:1112-1124) and the re-orderer skips its
    index; --no_frame_drops disables that.  A slow worker (test hook: 60 ms per frame) makes the queue back up."""
    import re
    common = [BIN, "--video", "synthetic:320x240:40:3", "--model", "coco", "--net_resolution", "160x96", "--resolution", "320x240",
              "--no_display", "--num_gpu", "1", "--frames_in_flight", "1", "--test_worker_delay_ms", "60"]
    out = tmp_path / "drop"
    p = subprocess.run(common + ["--write_json", str(out)], capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()
    m = re.search(rb"frames produced (\d+), written (\d+), dropped (\d+)", p.stderr)
    produced, written, dropped = (int(x) for x in m.groups())
    assert produced == 40 and dropped > 0 and written + dropped == produced
    assert len(os.listdir(out)) == written            # dropped frames leave no file; nothing is written twice
    out2 = tmp_path / "nodrop"
    p = subprocess.run(common + ["--write_json", str(out2), "--no_frame_drops"], capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()
    m = re.search(rb"frames produced (\d+), written (\d+), dropped (\d+)", p.stderr)
    assert tuple(int(x) for x in m.groups()) == (40, 40, 0)
    assert sorted(os.listdir(out2)) == [f"frame{i:06d}.json" for i in range(40)]


# ------------------------------------------------------------------------------------------
# row a1 against something that is NOT the product: tests/_cvref.py, an independent numpy restatement of OpenCV's
# published warpAffine(INTER_CUBIC) / resize(INTER_AREA) (2-D fixed-point table, one rounding per pixel, area fast path)
# ------------------------------------------------------------------------------------------
GEOMS = [  # (frame w, h, display w, h, net w, h, scales, start, gap)
    (640, 480, 1280, 720, 656, 368, 1, 1.0, 0.3),      # BASELINE config 1: 640x480 jpg, enlarging display warp (s = 1.5)
    (1280, 720, 1280, 720, 656, 368, 3, 1.0, 0.15),    # configs 2-4: identity warp, 3 pyramid levels
    (1920, 1080, 1280, 720, 656, 368, 2, 1.0, 0.25),   # shrinking warp (s = 2/3)
    (500, 375, 640, 360, 320, 176, 1, 1.0, 0.3),       # portrait-ish frame: right part of the display stays black
    (333, 500, 1312, 736, 656, 368, 1, 1.0, 0.3),      # display = 2x the net: resizeAreaFast_ 2x2
    (320, 240, 960, 528, 320, 176, 1, 1.0, 0.3),       # display = 3x the net: resizeAreaFast_ general
    # round 5: --start_scale != 1 (rtpose.cpp:68, 353-368: s = start_scale - i gap for EVERY level incl. the first), portrait, large
    (1280, 720, 1280, 720, 656, 368, 1, 0.8, 0.15),    # 528x304 inside 656x368: the first level is padded too
    (1280, 720, 1280, 720, 656, 368, 3, 0.8, 0.15),    # 0.8, 0.65, 0.5 (the last one: 1280 / 336, 720 / 192 — table path)
    (1920, 1080, 1280, 720, 656, 368, 2, 0.65, 0.25),  # 0.65, 0.4
    (720, 1280, 720, 1280, 368, 656, 2, 0.8, 0.15),    # portrait frame, display and net
    (1920, 1080, 1920, 1080, 1312, 736, 2, 0.8, 0.15), # large net (low-res 92 x 164)
    (1312, 736, 1312, 736, 656, 368, 2, 0.5, 0.25),    # display = 4x / 8x the target: resizeAreaFast_ at start_scale 0.5
    # --resolution smaller than a pyramid level: cv::resize(INTER_AREA) enlarges with its bilinear kernel and area-mode coefficients
    (640, 360, 320, 180, 656, 368, 2, 1.0, 0.3),       # both axes enlarged at both levels (656x368, 464x272 from 320x180)
    (500, 375, 800, 200, 320, 240, 2, 1.0, 0.4),       # mixed: x shrinks (800 -> 320 / 192), y grows (200 -> 240) at level 0, shrinks at level 1 (144)
    (333, 201, 123, 77, 160, 96, 1, 1.0, 0.3),         # odd sizes
]


def _frame(w, h, seed):
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, size=(h // 8 + 2, w // 8 + 2, 3)).astype(np.float32)
    img = np.kron(base, np.ones((8, 8, 1), np.float32))[:h, :w]                   # blocks: edges + flat areas
    img = img * 0.7 + rs.randint(0, 77, size=(h, w, 3))                            # + noise: every rounding case occurs
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("geom", GEOMS)
def test_host_preprocess_equals_independent_opencv_restatement(geom):
    import caffe_rtpose_amd as r
    import _cvref
    import _oracle as orc
    fw, fh, dw, dh, W, H, N, start, gap = geom
    img = _frame(fw, fh, seed=fw + fh)
    x, disp, fs = r.preprocess_frame(img, dw, dh, W, H, N, start, gap)
    want_x, want_disp, want_fs = _cvref.producer_frame(img, dw, dh, W, H, N, start, gap, orc.process_and_pad_image)
    assert fs == want_fs
    assert np.array_equal(disp, want_disp), f"display warp differs in {(disp != want_disp).mean():.2%} of the bytes"
    assert np.array_equal(x, want_x)


def test_cvref_table_and_primitives_self_checks():
    """Properties of the restatement itself: every 2-D weight set sums to 1 << 15 (at phase (0,0) the weight 1.0 saturates
    to 32767 and initInterTab2D's correction puts the missing 1 on entry [2][2]: still the identity on u8 data); an
    integer down-scale by 2 is the rounded-half-up block mean; identity resize is a copy."""
    import _cvref
    tab = _cvref.bicubic_tab_i()
    assert tab.shape == (32, 32, 4, 4) and np.all(tab.sum(axis=(2, 3)) == 32768)
    assert tab[0, 0, 1, 1] == 32767 and tab[0, 0, 2, 2] == 1 and np.count_nonzero(tab[0, 0]) == 2
    img = _frame(64, 48, 5)
    assert np.array_equal(_cvref.warp_affine_scale_cubic(img, 1.0, 64, 48), img)
    half = _cvref.resize_area(img, 32, 24)
    blk = img.astype(np.int64).reshape(24, 2, 32, 2, 3).sum(axis=(1, 3))
    assert np.array_equal(half, ((blk + 2) >> 2).astype(np.uint8))
    assert np.array_equal(_cvref.resize_area(img, 64, 48), img)


@pytest.mark.gpu
def test_cpp_net_api_mirror_runs_on_the_gpu(tmp_path):
    """INTEGRATION.md path A executed, not just linked: a C++ program written against csrc/host_api.h (the Net API names
    rtpose.cpp uses: Net, CopyTrainedLayersFrom, blobs()[0]->Reshape, layer_by_name("nms"/"resize"), ForwardFrom(0),
    blob_by_name("resized_map"/"joints")) returns the blobs the ctypes path returns for the same frame."""
    import caffe_rtpose_amd as r
    W, H, N = 160, 96, 2
    proto, model = tmp_path / "net.prototxt", tmp_path / "net.caffemodel"
    r.write_builtin_prototxt(r.MODEL_COCO_18, proto)
    r.write_synthetic_caffemodel(r.MODEL_COCO_18, 7, model)
    x = r.preprocess_frame(r.synth_frame(320, 240, 3, seed=8), 320, 240, W, H, N, 1.0, 0.25)[0]
    x.tofile(tmp_path / "in.f32")
    src = tmp_path / "use.cpp"
    src.write_text('''
#include <cstdio>
#include <vector>
#include "caffe_rtpose_amd/csrc/host_api.h"
int main(int argc, char** argv) {
  rtpose::Net net(argv[1], rtpose::TEST, 0);
  net.CopyTrainedLayersFrom(argv[2]);
  net.set_display_resolution(320, 240);
  net.blobs()[0]->Reshape({%d, 3, %d, %d});
  auto resize = net.layer_by_name<rtpose::ImResizeLayer>("resize");
  resize->SetStartScale(1.f); resize->SetScaleGap(0.25f);
  net.Reshape();
  auto nms = net.layer_by_name<rtpose::NmsLayer>("nms");
  nms->SetThreshold(0.05f);
  FILE* f = fopen(argv[3], "rb");
  auto in = net.blobs()[0];
  if (!f || fread(in->mutable_cpu_data(), sizeof(float), in->data_.size(), f) != in->data_.size()) return 3;
  fclose(f);
  net.ForwardFrom(0);
  auto res = net.blob_by_name("resized_map");
  auto jo = net.blob_by_name("joints");
  FILE* o = fopen(argv[4], "wb");
  const int hdr[4] = {nms->GetMaxPeaks(), nms->GetNumParts(), (int)res->data_.size(), (int)jo->data_.size()};
  fwrite(hdr, sizeof(int), 4, o);
  fwrite(res->cpu_data(), sizeof(float), res->data_.size(), o);
  fwrite(jo->cpu_data(), sizeof(float), jo->data_.size(), o);
  fclose(o);
  return 0;
}
''' % (N, H, W))
    exe = tmp_path / "use"
    subprocess.check_call(["g++", "-std=c++17", "-I", ROOT, "-o", str(exe), str(src), "-L", os.path.join(ROOT, "caffe_rtpose_amd"),
                           "-lrtpose_mi355x", "-Wl,-rpath," + os.path.join(ROOT, "caffe_rtpose_amd")])
    p = subprocess.run([str(exe), str(proto), str(model), str(tmp_path / "in.f32"), str(tmp_path / "out.bin")], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    raw = open(tmp_path / "out.bin", "rb").read()
    hdr = np.frombuffer(raw[:16], np.int32)
    assert tuple(hdr[:2]) == (64, 18)
    res = np.frombuffer(raw[16:16 + 4 * hdr[2]], np.float32).reshape(57, H, W)
    peaks = np.frombuffer(raw[16 + 4 * hdr[2]:], np.float32).reshape(18, 65, 3)
    e = r.Engine(r.Config(proto_path=str(proto), weights_path=str(model), net_w=W, net_h=H, num_scales=N, scale_gap=0.25, disp_w=320, disp_h=240,
                          frames_in_flight=1))
    d = e.forward_debug(x)
    assert np.array_equal(res, d["resized"]) and np.array_equal(peaks, d["peaks"])
    assert peaks[:, 0, 0].sum() > 0
    e.close()
