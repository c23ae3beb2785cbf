"""image / video decoding without OpenCV (csrc/codecs.cpp), CPU only.

JPEG is pinned bit-for-bit against Pillow's (libjpeg-turbo) decode of the same files — the library
cv::imread runs — through the fixtures made by tools/make_codec_fixtures.py; PNG is lossless and
pinned against the source arrays.  Y4M colour conversion is PARITY UNPINNED (stated in codecs.cpp)."""
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "codecs")


def _cases(ext):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*." + ext)) if os.path.exists(p[:-4] + ".npy"))


@pytest.mark.parametrize("name", _cases("jpg"))
def test_jpeg_decode_bit_exact_vs_libjpeg(name):
    import caffe_rtpose_amd as r
    want = np.load(os.path.join(GOLD, name + ".npy"))
    data = open(os.path.join(GOLD, name + ".jpg"), "rb").read()
    got = r.decode_image(data)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max diff {np.abs(got.astype(int) - want.astype(int)).max()}, {int((got != want).sum())} values"
    # and through the file loader the CLI uses
    assert np.array_equal(r.load_image(os.path.join(GOLD, name + ".jpg")), want)


@pytest.mark.parametrize("name", _cases("png"))
def test_png_decode_exact(name):
    import caffe_rtpose_amd as r
    want = np.load(os.path.join(GOLD, name + ".npy"))
    got = r.load_image(os.path.join(GOLD, name + ".png"))
    assert got.shape == want.shape and np.array_equal(got, want)


def test_unsupported_and_corrupt_files_fail_with_a_message(tmp_path):
    import caffe_rtpose_amd as r
    good = open(os.path.join(GOLD, "j420_q75.jpg"), "rb").read()
    arith = good.replace(b"\xff\xc0", b"\xff\xc9", 1)   # SOF9: arithmetic coding
    with pytest.raises(r.RtpError) as ei:
        r.decode_image(arith)
    assert "not supported" in str(ei.value)
    with pytest.raises(r.RtpError):
        r.decode_image(good[: len(good) // 8])          # cut inside the headers
    with pytest.raises(r.RtpError):
        r.decode_image(b"\x89PNG\r\n\x1a\n" + b"\0" * 40)
    with pytest.raises(r.RtpError):
        r.decode_image(b"GIF89a" + b"\0" * 64)
    # a truncated scan still decodes (libjpeg pads with zeros too): same size, no crash
    assert r.decode_image(good[: len(good) - 40]).shape == np.load(os.path.join(GOLD, "j420_q75.npy")).shape
    png = open(os.path.join(GOLD, "p_rgb8.png"), "rb").read()
    with pytest.raises(r.RtpError):
        r.decode_image(png[: len(png) - 30])


def test_y4m_and_mjpeg_readers(tmp_path):
    import caffe_rtpose_amd as r
    W, H, n = 20, 12, 3
    rs = np.random.RandomState(3)
    frames = []
    p = tmp_path / "clip.y4m"
    with open(p, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F25:1 Ip A1:1 C420jpeg\n" % (W, H))
        for _ in range(n):
            y = rs.randint(16, 236, (H, W)).astype(np.uint8)
            u = rs.randint(16, 241, (H // 2, W // 2)).astype(np.uint8)
            v = rs.randint(16, 241, (H // 2, W // 2)).astype(np.uint8)
            f.write(b"FRAME\n" + y.tobytes() + u.tobytes() + v.tobytes())
            frames.append((y, u, v))
    vid = r.Video(p)
    assert (vid.w, vid.h_, vid.nframes) == (W, H, n)
    for y, u, v in frames:
        got = vid.read()
        c = 298 * (y.astype(np.int64) - 16)
        d = np.repeat(np.repeat(u, 2, 0), 2, 1).astype(np.int64) - 128
        e = np.repeat(np.repeat(v, 2, 0), 2, 1).astype(np.int64) - 128
        want = np.stack([np.clip((c + 516 * d + 128) >> 8, 0, 255), np.clip((c - 100 * d - 208 * e + 128) >> 8, 0, 255),
                         np.clip((c + 409 * e + 128) >> 8, 0, 255)], -1).astype(np.uint8)
        assert np.array_equal(got, want)
    assert vid.read() is None
    vid.close()
    # raw MJPEG = concatenated JPEG files
    a = open(os.path.join(GOLD, "j420_q75.jpg"), "rb").read()
    b = open(os.path.join(GOLD, "j444_q90.jpg"), "rb").read()
    m = tmp_path / "clip.mjpeg"
    m.write_bytes(a + b + a)
    vid = r.Video(m)
    assert vid.nframes == 3
    assert np.array_equal(vid.read(), np.load(os.path.join(GOLD, "j420_q75.npy")))
    assert np.array_equal(vid.read(), np.load(os.path.join(GOLD, "j444_q90.npy")))
    assert np.array_equal(vid.read(), np.load(os.path.join(GOLD, "j420_q75.npy")))
    assert vid.read() is None
    vid.close()
    with pytest.raises(r.RtpError):
        r.Video(os.path.join(GOLD, "p_rgb8.png"))


@pytest.mark.parametrize("ref", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "enc_*.jpgref"))))
def test_jpeg_encoder_byte_identical_to_libjpeg(ref):
    """cv::imwrite(.jpg, quality) = libjpeg defaults: the file must equal Pillow's (libjpeg-turbo) byte for byte."""
    import caffe_rtpose_amd as r
    src = np.load(os.path.join(GOLD, ref[:-7] + ".npy"))
    q = int(ref[:-7].rsplit("_q", 1)[1])
    want = open(os.path.join(GOLD, ref), "rb").read()
    got = r.encode_jpeg(src, q)
    assert got == want
    # and our own decoder reads it back to what libjpeg would: round trip stays close to the source
    dec = r.decode_image(got)
    assert dec.shape == src.shape
    if q >= 90 and src.shape[0] > 8:
        assert np.abs(dec.astype(int) - src.astype(int)).mean() < 20.0   # noisy source + 4:2:0 chroma


def test_decoders_survive_mutated_files():
    """Untrusted input: corrupted / truncated / spliced files must come back as an image or an error —
    never a crash or a hang (tools/fuzz_codecs.cpp is the ASan/UBSan version of this loop)."""
    import random
    import caffe_rtpose_amd as r
    files = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(GOLD, "*.jpg")) + glob.glob(os.path.join(GOLD, "*.png")))]
    rnd = random.Random(1234)
    decoded = rejected = 0
    for _ in range(4000):
        d = bytearray(rnd.choice(files))
        mode = rnd.random()
        if mode < 0.5:
            for _ in range(rnd.randint(1, 8)):
                d[rnd.randrange(len(d))] = rnd.randrange(256)
        elif mode < 0.7:
            d = d[: rnd.randrange(1, len(d))]
        elif mode < 0.85:
            i = rnd.randrange(len(d))
            d[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 16)))
        else:
            i = rnd.randrange(len(d) - 2)
            d[i] = 0xFF
            d[i + 1] = rnd.choice([0xC0, 0xC2, 0xC4, 0xDA, 0xDB, 0xDD, 0xD9, 0xD0, 0xC9])
        try:
            img = r.decode_image(bytes(d))
            assert img.ndim == 3 and img.shape[2] == 3
            decoded += 1
        except r.RtpError:
            rejected += 1
    assert decoded > 500 and rejected > 500
    with pytest.raises(r.RtpError):   # absurd header: refused before anything is allocated
        r.decode_image(b"\x89PNG\r\n\x1a\n\x00\x00\x00\rIHDR" + (70000).to_bytes(4, "big") + (70000).to_bytes(4, "big") + b"\x08\x02\x00\x00\x00" + b"\0" * 12)
