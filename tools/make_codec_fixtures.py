#!/usr/bin/env python
"""Golden vectors for csrc/codecs.cpp, made in the BUILD container with Pillow (bundled libjpeg-turbo /
libpng — the decoders cv::imread would run): small JPEG/PNG files plus the pixels they decode to.
    python tools/make_codec_fixtures.py          -> tests/golden/codecs/*.{jpg,png,npy}
JPEG expectations are PIL's decode (libjpeg default path: islow IDCT, fancy up-sampling);
PNG expectations are the source arrays mapped the way libpng does under cv::IMREAD_COLOR."""
import io
import os
import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "codecs")
os.makedirs(OUT, exist_ok=True)
rs = np.random.RandomState(7)


def scene(w, h):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 120 * np.sin(xx / 7.0 + yy / 13.0), 127 + 120 * np.cos(xx / 5.0 - yy / 9.0), (xx * 3 + yy * 5) % 256], -1)
    img += rs.randn(h, w, 3) * 12
    img[h // 3:h // 3 + 6, w // 4:w // 4 + 9] = (255, 0, 0)   # hard edges: chroma up-sampling matters
    img[h // 2:h // 2 + 5, w // 2:w // 2 + 7] = (0, 255, 255)
    return np.clip(img, 0, 255).astype(np.uint8)


def save_jpeg(name, rgb, **kw):
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, "JPEG", **kw)
    data = buf.getvalue()
    dec = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    open(os.path.join(OUT, name + ".jpg"), "wb").write(data)
    np.save(os.path.join(OUT, name + ".npy"), dec[:, :, ::-1].copy())  # BGR
    return data


a = scene(67, 45)
save_jpeg("j444_q90", a, quality=90, subsampling=0)
save_jpeg("j422_q85", a, quality=85, subsampling=1)
save_jpeg("j420_q75", a, quality=75, subsampling=2)
save_jpeg("j420_q30_opt", scene(50, 33), quality=30, subsampling=2, optimize=True)
save_jpeg("j420_rst", scene(40, 40), quality=80, subsampling=2, restart_marker_blocks=2)
save_jpeg("j420_tiny", scene(5, 3), quality=80, subsampling=2)       # down-sampled width <= 2: no fancy up-sampling
save_jpeg("j422_w3", scene(3, 9), quality=80, subsampling=1)
save_jpeg("j420_16x16", scene(16, 16), quality=95, subsampling=2)
buf = io.BytesIO(); Image.fromarray(a[:, :, 0]).save(buf, "JPEG", quality=88); d = buf.getvalue()
open(os.path.join(OUT, "jgray_q88.jpg"), "wb").write(d)
np.save(os.path.join(OUT, "jgray_q88.npy"), np.repeat(np.asarray(Image.open(io.BytesIO(d)).convert("L"))[:, :, None], 3, 2))
# progressive (SOF2): DC/AC first and refinement scans, EOB runs
save_jpeg("jprog444_q80", a, quality=80, subsampling=0, progressive=True)
save_jpeg("jprog420_q75", a, quality=75, subsampling=2, progressive=True)
save_jpeg("jprog422_q92", scene(50, 33), quality=92, subsampling=1, progressive=True)
save_jpeg("jprog420_tiny", scene(5, 3), quality=80, subsampling=2, progressive=True)
save_jpeg("jprog420_rst", scene(40, 40), quality=80, subsampling=2, progressive=True, restart_marker_blocks=3)
buf = io.BytesIO(); Image.fromarray(a[:, :, 1]).save(buf, "JPEG", quality=85, progressive=True); d = buf.getvalue()
open(os.path.join(OUT, "jproggray_q85.jpg"), "wb").write(d)
np.save(os.path.join(OUT, "jproggray_q85.npy"), np.repeat(np.asarray(Image.open(io.BytesIO(d)).convert("L"))[:, :, None], 3, 2))

# ---- PNG ----
def save_png(name, im, expect_bgr, **kw):
    im.save(os.path.join(OUT, name + ".png"), "PNG", **kw)
    np.save(os.path.join(OUT, name + ".npy"), np.ascontiguousarray(expect_bgr, np.uint8))


b = scene(37, 29)
save_png("p_rgb8", Image.fromarray(b), b[:, :, ::-1])
alpha = rs.randint(0, 256, b.shape[:2]).astype(np.uint8)
save_png("p_rgba8", Image.fromarray(np.dstack([b, alpha])), b[:, :, ::-1])              # alpha stripped, not blended
gray = b[:, :, 1]
save_png("p_gray8", Image.fromarray(gray), np.repeat(gray[:, :, None], 3, 2))
pal = Image.fromarray(b).quantize(colors=23)
save_png("p_pal8", pal, np.asarray(pal.convert("RGB"))[:, :, ::-1])
bits1 = (gray > 128)
save_png("p_gray1", Image.fromarray(bits1), np.repeat((bits1 * 255).astype(np.uint8)[:, :, None], 3, 2))
g16 = (rs.randint(0, 65536, gray.shape)).astype(np.uint16)
save_png("p_gray16", Image.fromarray(g16), np.repeat((g16 >> 8).astype(np.uint8)[:, :, None], 3, 2))   # strip_16: high byte
# interlaced RGB, written by hand (Pillow cannot): Adam7 passes through zlib
import struct, zlib
def chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
c = scene(11, 9)
raw = b""
for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
    sub = c[y0::dy, x0::dx]
    if sub.size == 0:
        continue
    for row in sub:
        raw += b"\x00" + row.tobytes()
png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 11, 9, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
open(os.path.join(OUT, "p_rgb8_adam7.png"), "wb").write(png)
np.save(os.path.join(OUT, "p_rgb8_adam7.npy"), c[:, :, ::-1].copy())
# ---- JPEG encoder: source pixels (BGR) + the bytes Pillow/libjpeg-turbo writes (= cv::imwrite's defaults: 4:2:0, std tables)
for w, h, q in [(67, 45, 98), (33, 17, 98), (16, 16, 75), (1, 1, 98), (8, 24, 50), (17, 33, 30), (40, 9, 98)]:
    s = scene(w, h)
    buf = io.BytesIO()
    Image.fromarray(s).save(buf, "JPEG", quality=q)
    np.save(os.path.join(OUT, f"enc_{w}x{h}_q{q}.npy"), s[:, :, ::-1].copy())
    open(os.path.join(OUT, f"enc_{w}x{h}_q{q}.jpgref"), "wb").write(buf.getvalue())
print(sorted(os.listdir(OUT)), sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)), "bytes")
