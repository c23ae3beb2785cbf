"""ctypes binding of oracle/librtpose_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "librtpose_oracle.so")
_lib = None

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(fp)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _lib = C.CDLL(_SO)
        _lib.orc_net_create.restype = C.c_void_p
        _lib.orc_write_json.restype = C.c_long
    return _lib


def conv2d(x, w, b, pad, stride=1):
    N, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    out = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().orc_conv2d(_f(x), N, Cin, H, W, _f(w), _f(b) if b is not None else None, Cout, k, pad, stride, _f(out))
    return out


def conv2d_naive(x, w, b, pad_h, pad_w, stride_h, stride_w):
    N, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    Ho = (H + 2 * pad_h - kh) // stride_h + 1
    Wo = (W + 2 * pad_w - kw) // stride_w + 1
    out = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().orc_conv2d_naive(_f(x), N, Cin, H, W, _f(w), _f(b) if b is not None else None, Cout, kh, kw, pad_h, pad_w, stride_h, stride_w, _f(out))
    return out


def relu(x, slope=0.0):
    y = np.ascontiguousarray(x, np.float32).copy()
    lib().orc_relu(_f(y), C.c_long(y.size), C.c_float(slope))
    return y


def maxpool(x, k=2, stride=2, pad=0):
    N, Cc, H, W = x.shape
    ho, wo = C.c_int(), C.c_int()
    lib().orc_maxpool_shape(H, W, k, stride, pad, C.byref(ho), C.byref(wo))
    out = np.empty((N, Cc, ho.value, wo.value), np.float32)
    lib().orc_maxpool(_f(x), N, Cc, H, W, k, stride, pad, _f(out))
    return out


def concat2(a, b):
    N, Ca, H, W = a.shape
    Cb = b.shape[1]
    out = np.empty((N, Ca + Cb, H, W), np.float32)
    lib().orc_concat2(_f(a), Ca, _f(b), Cb, N, C.c_long(H * W), _f(out))
    return out


def imresize(src, tw, th, start_scale=1.0, scale_gap=0.3):
    num, Cc, h, w = src.shape
    dst = np.empty((1, Cc, th, tw), np.float32)
    lib().orc_imresize(_f(src), num, Cc, h, w, tw, th, C.c_float(start_scale), C.c_float(scale_gap), _f(dst))
    return dst


def nms(resized, num_parts, max_peaks, threshold, peaks_init=None):
    """resized: [C][H][W] (C > num_parts).  Returns peaks [num_parts][max_peaks+1][3]."""
    Cc, H, W = resized.shape[-3:]
    r = np.ascontiguousarray(resized.reshape(Cc, H, W), np.float32)
    peaks = np.zeros((num_parts, max_peaks + 1, 3), np.float32) if peaks_init is None else peaks_init.copy()
    lib().orc_nms(_f(r), Cc, H, W, num_parts, max_peaks, C.c_float(threshold), _f(peaks))
    return peaks


def default_thresholds(model):
    nms_thr, inter_thr, sc = C.c_float(), C.c_float(), C.c_float()
    above, cnt = C.c_int(), C.c_int()
    lib().orc_default_thresholds(model, C.byref(nms_thr), C.byref(inter_thr), C.byref(above), C.byref(cnt), C.byref(sc))
    return dict(nms_threshold=nms_thr.value, inter_threshold=inter_thr.value, inter_min_above=above.value,
                min_subset_cnt=cnt.value, min_subset_score=sc.value)


def connect(model, resized, peaks, max_peaks, net_w, net_h, disp_w, disp_h, thr=None, max_people=96):
    thr = thr or default_thresholds(model)
    num_parts = 18 if model == 0 else 15
    r = np.ascontiguousarray(resized, np.float32)
    p = np.ascontiguousarray(peaks, np.float32)
    joints = np.zeros((max_people, num_parts, 3), np.float32)
    cnt = lib().orc_connect(model, _f(r), _f(p), max_peaks, net_w, net_h, disp_w, disp_h,
                            C.c_float(thr["inter_threshold"]), thr["inter_min_above"], thr["min_subset_cnt"],
                            C.c_float(thr["min_subset_score"]), max_people, _f(joints))
    assert cnt >= 0, f"oracle connect failed {cnt}"
    return cnt, joints


def connect_trace(model, resized, peaks, max_peaks, net_w, net_h, disp_w, disp_h, thr=None, max_people=96):
    """connect() plus the decision trace of the same run: (count, joints, cand, conn, rows) with
    cand [n][10] = limb, i, j, accepted, sum/count, count, thr_margin, round_margin, norm_vec, near_thr  (EVERY peak pair of every limb),
    conn [n][4]  = limb, i, j, score  (greedy picks in pick order),
    rows [n][num_parts + 3] = subset rows: peaks offsets per part (0 = absent), count, score, kept."""
    thr = thr or default_thresholds(model)
    num_parts = 18 if model == 0 else 15
    r = np.ascontiguousarray(resized, np.float32)
    p = np.ascontiguousarray(peaks, np.float32)
    joints = np.zeros((max_people, num_parts, 3), np.float32)
    nl = 19 if model == 0 else 14
    cap_c, cap_k, cap_r = nl * max_peaks * max_peaks, nl * max_peaks, 2 * nl * max_peaks
    cand = np.zeros((cap_c, 10), np.float64)
    conn = np.zeros((cap_k, 4), np.float64)
    rows = np.zeros((cap_r, num_parts + 3), np.float64)
    nc, nk, nr = C.c_long(), C.c_long(), C.c_long()
    dp = C.POINTER(C.c_double)
    cnt = lib().orc_connect_trace(model, _f(r), _f(p), max_peaks, net_w, net_h, disp_w, disp_h,
                                  C.c_float(thr["inter_threshold"]), thr["inter_min_above"], thr["min_subset_cnt"],
                                  C.c_float(thr["min_subset_score"]), max_people, _f(joints),
                                  cand.ctypes.data_as(dp), C.c_long(cap_c), C.byref(nc), conn.ctypes.data_as(dp), C.c_long(cap_k), C.byref(nk),
                                  rows.ctypes.data_as(dp), C.c_long(cap_r), C.byref(nr))
    assert cnt >= 0, f"oracle connect failed {cnt}"
    assert nc.value <= cap_c and nk.value <= cap_k and nr.value <= cap_r
    return cnt, joints, cand[:nc.value], conn[:nk.value], rows[:nr.value]


def render_pose(model, bgr, joints, num_people, googly=0):
    img = np.ascontiguousarray(bgr, np.uint8)
    j = np.ascontiguousarray(joints, np.float32).reshape(-1)
    if j.size == 0:
        j = np.zeros(3, np.float32)
    out = np.empty_like(img)
    rc = lib().orc_render_pose(model, img.ctypes.data_as(C.POINTER(C.c_ubyte)), img.shape[1], img.shape[0], _f(j), num_people, googly,
                               out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    assert rc == 0
    return out


def render_view(model, bgr, maps, part_to_show):
    """the --part_to_show views of render() (heat map / all parts / PAFs) over a u8 BGR HWC image; maps [C][net_h][net_w]"""
    img = np.ascontiguousarray(bgr, np.uint8)
    m = np.ascontiguousarray(maps, np.float32)
    out = np.empty_like(img)
    rc = lib().orc_render_view(model, img.ctypes.data_as(C.POINTER(C.c_ubyte)), img.shape[1], img.shape[0], m.shape[2], m.shape[1], _f(m),
                               int(part_to_show), out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    assert rc == 0
    return out


def write_json(joints, num_people, num_parts, frame_scale):
    buf = C.create_string_buffer(1 << 20)
    j = np.ascontiguousarray(joints, np.float32)
    n = lib().orc_write_json(buf, C.c_long(len(buf)), _f(j), num_people, num_parts, C.c_float(frame_scale))
    assert n >= 0
    return buf.raw[:n]


def process_and_pad_image(img_u8, tw, th, normalize):
    oh, ow, _ = img_u8.shape
    out = np.empty((3, th, tw), np.float32)
    img = np.ascontiguousarray(img_u8, np.uint8)
    rc = lib().orc_process_and_pad_image(_f(out), img.ctypes.data_as(C.POINTER(C.c_ubyte)), ow, oh, tw, th, int(normalize))
    assert rc == 0
    return out


def model_tables(model):
    npart, nlimb = C.c_int(), C.c_int()
    limb = (C.c_int * 38)()
    mp = (C.c_int * 38)()
    assert lib().orc_model_tables(model, C.byref(npart), C.byref(nlimb), limb, mp) == 0
    n = nlimb.value * 2
    return npart.value, nlimb.value, list(limb)[:n], list(mp)[:n]


class Net:
    """The linevec conv stack (conv1_1 .. concat_stage7) on the CPU oracle."""

    def __init__(self, model):
        self.h = C.c_void_p(lib().orc_net_create(model))
        assert self.h
        self.model = model
        self.convs = []
        name = C.create_string_buffer(64)
        cin, cout, k = C.c_int(), C.c_int(), C.c_int()
        for i in range(lib().orc_net_num_convs(self.h)):
            lib().orc_net_conv_info(self.h, i, name, 64, C.byref(cin), C.byref(cout), C.byref(k))
            self.convs.append((name.value.decode(), cin.value, cout.value, k.value))

    def set_weights(self, i, w, b):
        w = np.ascontiguousarray(w, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        _, cin, cout, k = self.convs[i]
        assert w.size == cout * cin * k * k and b.size == cout
        assert lib().orc_net_set_weights(self.h, i, _f(w), _f(b)) == 0

    def forward(self, x, stop_after=None, keep_all=False):
        x = np.ascontiguousarray(x, np.float32)
        N, c3, H, W = x.shape
        rc = lib().orc_net_forward(self.h, _f(x), N, H, W, (stop_after or "").encode(), int(keep_all))
        assert rc == 0, rc
        return self.blob(stop_after or "concat_stage7")

    def blob(self, name):
        n, c, h, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = lib().orc_net_blob_shape(self.h, name.encode(), C.byref(n), C.byref(c), C.byref(h), C.byref(w))
        assert rc == 0, f"no blob {name}"
        out = np.empty((n.value, c.value, h.value, w.value), np.float32)
        lib().orc_net_blob(self.h, name.encode(), _f(out))
        return out

    def __del__(self):
        try:
            lib().orc_net_destroy(self.h)
        except Exception:
            pass


def num_threads():
    return lib().orc_num_threads()
