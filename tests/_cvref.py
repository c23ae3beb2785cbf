"""Independent numpy restatement of the two OpenCV primitives on the producer path (row a1; rtpose.cpp:322-368):
    cv::warpAffine(src, dst, M = diag(s, s), dsize, INTER_CUBIC, BORDER_CONSTANT, 0)     8UC3
    cv::resize(src, dst, dsize, 0, 0, INTER_AREA)                                        8UC3, shrinking
TEST INFRASTRUCTURE.  OpenCV is a third-party dependency that is absent from /root/reference and from this image
(and unpinned by the reference: Makefile:197-202 accepts 2.4 or 3.x), so this file restates the published algorithm of
modules/imgproc/src/imgwarp.cpp (the same in 2.4.x and 3.x) from the ground up, table-driven and vectorised, sharing
no code with caffe_rtpose_amd/csrc/preprocess.cpp or preproc.hip:

warpAffine  * M is inverted as warpAffine does (D = 1/(M0*M4 - M1*M3); A11 = M4*D ...), coordinates in fixed point:
              AB_BITS = 10, INTER_BITS = 5: X = (cvRound(A11*x*1024) + cvRound(b1*1024) + 16) >> 5; sx = X >> 5, fx = X & 31
            * remapBicubic with the FIXED-POINT 2-D table of initInterTab2D: for every (fy, fx) phase pair the 16 weights
              are saturate_cast<short>(wy[k1]*wx[k2]*32768) of the float 1-D cubic (A = -0.75) weights, and when they do
              not sum to 32768 the difference goes to the largest (sum too small) / smallest (sum too large) of the four
              entries k1, k2 in {2, 3}; pixel = saturate_cast<uchar>((sum of 16 products + 2^14) >> 15); taps outside
              the image contribute 0 (BORDER_CONSTANT, value 0).
resize AREA * scale = 1/(dsize/ssize) per axis (as resize() computes it); integer scales on both axes take
              resizeAreaFast_ (2x2: (a+b+c+d+2)>>2; else saturate_cast<uchar>(int_sum * (1.f/area)));
            * otherwise computeResizeAreaTab + resizeArea_: per source row buf[dx] += S*alpha in table order (float),
              sum[dx] += beta*buf[dx] in row order (float), saturate_cast<uchar> = round half to even.
"""
import numpy as np


def _cubic_1d():
    A = np.float32(-0.75)
    tab = np.zeros((32, 4), np.float32)
    scale = np.float32(1.0) / np.float32(32)
    one = np.float32(1)
    for i in range(32):
        x = np.float32(i) * scale
        c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
        c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
        c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
        c3 = one - c0 - c1 - c2
        tab[i] = (c0, c1, c2, c3)
    return tab


def bicubic_tab_i():
    """BicubicTab_i[fy][fx][k1][k2] (int16) as initInterTab2D(INTER_CUBIC, fixpt) builds it."""
    t1 = _cubic_1d()
    out = np.zeros((32, 32, 4, 4), np.int64)
    for i in range(32):
        for j in range(32):
            v = (t1[i][:, None] * t1[j][None, :]).astype(np.float32)            # float product vy*vx
            it = np.clip(np.rint((v * np.float32(32768)).astype(np.float32)), -32768, 32767).astype(np.int64)   # saturate_cast<short>(cvRound)
            diff = int(it.sum()) - 32768
            if diff:
                Mk = mk = (2, 2)
                for k1 in (2, 3):
                    for k2 in (2, 3):
                        if it[k1, k2] < it[mk]:
                            mk = (k1, k2)
                        elif it[k1, k2] > it[Mk]:
                            Mk = (k1, k2)
                if diff < 0:
                    it[Mk] -= diff
                else:
                    it[mk] -= diff
            out[i, j] = it
    assert out.min() >= -32768 and out.max() <= 32767
    return out


_TAB = None


def warp_affine_scale_cubic(img, s, dw, dh):
    """cv::warpAffine(img, M = [[s,0,0],[0,s,0]], (dw, dh), INTER_CUBIC, BORDER_CONSTANT, 0) for an HxWx3 uint8 image."""
    global _TAB
    if _TAB is None:
        _TAB = bicubic_tab_i()
    sh, sw, _ = img.shape
    M0, M1, M2, M3, M4, M5 = float(s), 0.0, 0.0, 0.0, float(s), 0.0
    D = M0 * M4 - M1 * M3
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M4 * D, M0 * D
    M0, M1, M3, M4 = A11, M1 * -D, M3 * -D, A22
    b1 = -M0 * M2 - M1 * M5
    b2 = -M3 * M2 - M4 * M5
    M2, M5 = b1, b2
    AB = 1024
    xs = np.arange(dw, dtype=np.float64)
    ys = np.arange(dh, dtype=np.float64)
    adelta = np.rint(M0 * xs * AB).astype(np.int64)
    bdelta = np.rint(M3 * xs * AB).astype(np.int64)
    X0 = np.rint((M1 * ys + M2) * AB).astype(np.int64) + 16
    Y0 = np.rint((M4 * ys + M5) * AB).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx = (X >> 5) - 1
    sy = (Y >> 5) - 1
    w = _TAB[Y & 31, X & 31]                          # [dh][dw][4][4]
    src = np.zeros((sh + 8, sw + 8, 3), np.int64)     # zero border = BORDER_CONSTANT 0 for every tap outside
    src[4:4 + sh, 4:4 + sw] = img
    acc = np.zeros((dh, dw, 3), np.int64)
    sxc = np.clip(sx, -4, sw) + 4                     # patches entirely outside read zeros either way
    syc = np.clip(sy, -4, sh) + 4
    for k1 in range(4):
        for k2 in range(4):
            acc += src[syc + k1, sxc + k2] * w[:, :, k1, k2][:, :, None]
    out = (acc + (1 << 14)) >> 15
    return np.clip(out, 0, 255).astype(np.uint8)


def _area_tab(ssize, dsize, scale):
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _linear_area_tab(ssize, dsize):
    """cv::resize's coefficient loop in `area_mode` (INTER_AREA asked for, but an axis is enlarged: "true area interpolation is only
    implemented for scale_x >= 1 && scale_y >= 1; in other cases it is emulated using some variant of bilinear interpolation"):
    sx = cvFloor(dx * scale); fx = (float)((dx + 1) - (sx + 1) * inv_scale); fx = fx <= 0 ? 0 : fx - cvFloor(fx); the source index is
    clamped at both ends with fx = 0; 11-bit fixed-point weights saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE).
    Returns (s0, s1, a0, a1) int arrays of length dsize (s1 = the right / lower neighbour, clipped)."""
    inv_scale = dsize / float(ssize)
    scale = 1.0 / inv_scale
    s0 = np.zeros(dsize, np.int64); s1 = np.zeros(dsize, np.int64); a0 = np.zeros(dsize, np.int64); a1 = np.zeros(dsize, np.int64)
    for d in range(dsize):
        sx = int(np.floor(d * scale))
        fx = np.float32((d + 1) - (sx + 1) * inv_scale)
        fx = np.float32(0) if fx <= 0 else np.float32(fx - np.floor(fx))
        if sx < 0:
            fx, sx = np.float32(0), 0
        if sx >= ssize - 1:
            fx, sx = np.float32(0), ssize - 1
        c0, c1 = np.float32(np.float32(1) - fx), fx
        a0[d] = int(np.clip(np.rint(np.float32(c0 * np.float32(2048))), -32768, 32767))
        a1[d] = int(np.clip(np.rint(np.float32(c1 * np.float32(2048))), -32768, 32767))
        s0[d], s1[d] = sx, min(sx + 1, ssize - 1)
    return s0, s1, a0, a1


def resize_area_enlarging(img, dw, dh):
    """cv::resize(img, (dw, dh), 0, 0, INTER_AREA) where at least one axis is enlarged, 8UC3: resizeGeneric_ with HResizeLinear (int
    rows: S[s0] a0 + S[s1] a1, weights scaled by 2048) and the u8 specialisation of VResizeLinear:
    dst = (((b0 * (row0 >> 4)) >> 16) + ((b1 * (row1 >> 4)) >> 16) + 2) >> 2."""
    sh, sw, _ = img.shape
    x0, x1, a0, a1 = _linear_area_tab(sw, dw)
    y0, y1, b0, b1 = _linear_area_tab(sh, dh)
    S = img.astype(np.int64)
    rows = S[:, x0] * a0[None, :, None] + S[:, x1] * a1[None, :, None]            # [sh][dw][3]
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return (out & 0xff).astype(np.uint8)                                           # uchar(...) cast


def resize_area(img, dw, dh):
    """cv::resize(img, (dw, dh), 0, 0, INTER_AREA) for an HxWx3 uint8 image."""
    sh, sw, _ = img.shape
    if (dw, dh) == (sw, sh):
        return img.copy()
    if dw > sw or dh > sh:
        return resize_area_enlarging(img, dw, dh)
    scale_x = 1.0 / (dw / float(sw))
    scale_y = 1.0 / (dh / float(sh))
    ix, iy = int(np.rint(scale_x)), int(np.rint(scale_y))
    eps = np.finfo(np.float64).eps
    if abs(scale_x - ix) < eps and abs(scale_y - iy) < eps:     # resizeAreaFast_
        blk = img[:dh * iy, :dw * ix].astype(np.int64).reshape(dh, iy, dw, ix, 3)
        ssum = blk.sum(axis=(1, 3))
        if ix == 2 and iy == 2:
            return ((ssum + 2) >> 2).astype(np.uint8)
        v = (ssum.astype(np.float32) * (np.float32(1.0) / np.float32(ix * iy))).astype(np.float32)
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)
    xt = _area_tab(sw, dw, scale_x)
    yt = _area_tab(sh, dh, scale_y)
    # x pass for every source row at once: buf[row][dx] += S[row][sx] * alpha, in table order (float32 adds)
    S = img.astype(np.float32)
    buf = np.zeros((sh, dw, 3), np.float32)
    for dx, sx, a in xt:
        buf[:, dx] = (buf[:, dx] + (S[:, sx] * a).astype(np.float32)).astype(np.float32)
    out = np.zeros((dh, dw, 3), np.uint8)
    sums = np.zeros((dh, dw, 3), np.float32)
    for dy, sy, b in yt:   # sum[dy] += beta * buf[sy], in row order
        sums[dy] = (sums[dy] + (b * buf[sy]).astype(np.float32)).astype(np.float32)
    out[:] = np.clip(np.rint(sums), 0, 255).astype(np.uint8)
    return out


def fit_scale(ow, oh, disp_w, disp_h):
    """rtpose.cpp:324-329"""
    if ow / float(oh) > disp_w / float(disp_h):
        return disp_w / float(ow)
    return disp_h / float(oh)


def producer_frame(img, disp_w, disp_h, net_w, net_h, num_scales, start_scale, scale_gap, pad):
    """The producer's per-frame work (rtpose.cpp:322-368) with `pad` = process_and_pad_image (oracle or _ref)."""
    s = fit_scale(img.shape[1], img.shape[0], disp_w, disp_h)
    disp = warp_affine_scale_cubic(img, s, disp_w, disp_h)
    outs = []
    for i in range(num_scales):
        sc = np.float32(float(start_scale) - i * float(scale_gap))             # float scale = START_SCALE - i*SCALE_GAP (double flags, rtpose.cpp:360)
        # target_width = 16 * ceil(NET_RESOLUTION_WIDTH * scale / 16): int * float is FLOAT arithmetic — 320 * 0.6f is exactly 192.0f and
        # the level is 192 wide; in double it would be 192.0000076 -> 208 (found by the round-5 geometry with scale_gap 0.4)
        tw = int(16 * np.ceil(np.float32(np.float32(net_w) * sc) / np.float32(16)))
        th = int(16 * np.ceil(np.float32(np.float32(net_h) * sc) / np.float32(16)))
        outs.append(pad(resize_area(disp, tw, th), net_w, net_h, 1))
    return np.stack(outs), disp, np.float32(s)
