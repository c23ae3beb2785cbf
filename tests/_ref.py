"""ctypes binding of oracle/_ref/libref.so — the REFERENCE'S OWN code (connectLimbs*, process_and_pad_image, the JSON
block, modelDescriptorFactory, and the ImResize / NMS CUDA kernels run as host C++), built by oracle/ref_recipe/build_ref.sh
from /root/reference.  TEST INFRASTRUCTURE ONLY: it pins oracle/rtpose_oracle.cpp (tests/test_ref_pin.py) and produces
tests/golden/ref_pin.npz (tools/make_ref_golden.py).  Nothing in the product may load it."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")
_lib = None
fp = C.POINTER(C.c_float)


def available():
    return os.path.exists(SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(SO)
        _lib.ref_last_error.restype = C.c_char_p
    return _lib


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(fp)


def _chk(rc):
    if rc < 0:
        raise RuntimeError("reference CHECK failed: " + lib().ref_last_error().decode())
    return rc


def model_tables(model):
    npart, nlimb = C.c_int(), C.c_int()
    limb, mp = (C.c_int * 64)(), (C.c_int * 64)()
    _chk(lib().ref_model_tables(model, C.byref(npart), C.byref(nlimb), limb, mp))
    n = nlimb.value * 2
    return npart.value, nlimb.value, list(limb)[:n], list(mp)[:n]


def process_and_pad_image(img_u8, tw, th, normalize):
    oh, ow, _ = img_u8.shape
    img = np.ascontiguousarray(img_u8, np.uint8)
    out = np.full((3, th, tw), np.nan, np.float32)
    _chk(lib().ref_process_and_pad_image(_f(out), img.ctypes.data_as(C.POINTER(C.c_ubyte)), ow, oh, tw, th, int(normalize)))
    return out


def imresize(src, tw, th, start_scale=1.0, scale_gap=0.3):
    num, Cc, h, w = src.shape
    src = np.ascontiguousarray(src, np.float32)
    dst = np.full((1, Cc, th, tw), np.nan, np.float32)
    _chk(lib().ref_imresize(_f(src), num, Cc, h, w, tw, th, C.c_float(start_scale), C.c_float(scale_gap), _f(dst)))
    return dst


def nms(resized, num_parts, max_peaks, threshold, peaks_init=None):
    Cc, H, W = resized.shape[-3:]
    r = np.ascontiguousarray(resized.reshape(Cc, H, W), np.float32)
    peaks = np.zeros((num_parts, max_peaks + 1, 3), np.float32) if peaks_init is None else np.ascontiguousarray(peaks_init, np.float32).copy()
    _chk(lib().ref_nms(_f(r), H, W, num_parts, max_peaks, C.c_float(threshold), _f(peaks)))
    return peaks


def connect(model, resized, peaks, max_peaks, net_w, net_h, disp_w, disp_h, thr, max_people=96):
    num_parts = 18 if model == 0 else 15
    r = np.ascontiguousarray(resized, np.float32)
    p = np.ascontiguousarray(peaks, np.float32)
    joints = np.zeros((max_people, num_parts, 3), np.float32)
    cnt = _chk(lib().ref_connect(model, _f(r), _f(p), max_peaks, net_w, net_h, disp_w, disp_h, C.c_float(thr["inter_threshold"]),
                                 thr["inter_min_above"], thr["min_subset_cnt"], C.c_float(thr["min_subset_score"]), _f(joints)))
    return cnt, joints


def render(model, bgr, joints, num_people, net_w, net_h, part_to_show=0, googly=0, heatmaps=None):
    """render() of rtpose.cpp:270-299 on one u8 BGR HWC display image (renderFunctions.cu kernels on the host)."""
    img = np.ascontiguousarray(bgr, np.uint8)
    j = np.ascontiguousarray(joints, np.float32).reshape(-1)
    if j.size == 0:
        j = np.zeros(3, np.float32)
    hm = None if heatmaps is None else np.ascontiguousarray(heatmaps, np.float32)
    out = np.empty_like(img)
    _chk(lib().ref_render(model, img.ctypes.data_as(C.POINTER(C.c_ubyte)), img.shape[1], img.shape[0], net_w, net_h,
                          _f(hm) if hm is not None else None, _f(j), int(num_people), int(part_to_show), int(googly),
                          out.ctypes.data_as(C.POINTER(C.c_ubyte))))
    return out


def write_json(tmpdir, joints, num_people, model, frame_scale, frame_number=7, image_path=None):
    j = np.ascontiguousarray(joints, np.float32).reshape(-1)
    if j.size == 0:
        j = np.zeros(3, np.float32)
    _chk(lib().ref_write_json(str(tmpdir).encode(), image_path.encode() if image_path else None, frame_number, model, _f(j), num_people,
                              C.c_float(frame_scale)))
    if image_path:
        stem = os.path.splitext(os.path.basename(image_path))[0]
        name = f"{stem}.json"
    else:
        name = f"frame{frame_number:06d}.json"
    return name, open(os.path.join(str(tmpdir), name), "rb").read()


# ---- the reference's own convolution / pooling code (oracle/ref_recipe/ref_conv_shim.cpp) ---------------------------------------
def _conv_call(fn, x, w, b, pad, stride):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    N, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = np.full((N, Cout, Ho, Wo), np.nan, np.float32)
    rc = fn(_f(x), N, Cin, H, W, _f(w), _f(np.ascontiguousarray(b, np.float32)) if b is not None else None, Cout, kh, kw, pad, pad, stride, stride, _f(out))
    if rc < 0:
        lib().ref_conv_last_error.restype = C.c_char_p
        raise RuntimeError("reference CHECK failed: " + lib().ref_conv_last_error().decode())
    return out


def caffe_conv(x, w, b, pad, stride=1):
    """caffe_conv of src/caffe/test/test_convolution_layer.cpp:21-139: the naive loop the reference's own tests use as ground truth."""
    return _conv_call(lib().ref_caffe_conv, x, w, b, pad, stride)


def im2col_conv(x, w, b, pad, stride=1):
    """im2col_cpu (src/caffe/util/im2col.cpp:19-55, the reference's) + the GEMM / bias shapes of base_conv_layer.cpp:257-280."""
    return _conv_call(lib().ref_im2col_conv, x, w, b, pad, stride)


def im2col(x, k, pad, stride=1):
    x = np.ascontiguousarray(x, np.float32)
    Cin, H, W = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    col = np.full((Cin * k * k, Ho * Wo), np.nan, np.float32)
    lib().ref_im2col(_f(x), Cin, H, W, k, k, pad, pad, stride, stride, _f(col))
    return col


def maxpool(x, k=2, stride=2, pad=0):
    """The MAX branch of PoolingLayer::Forward_cpu (pooling_layer.cpp:149-186) with Reshape's output size (:90-107)."""
    x = np.ascontiguousarray(x, np.float32)
    N, Cc, H, W = x.shape
    ho, wo = C.c_int(), C.c_int()
    assert lib().ref_maxpool(_f(x), N, Cc, H, W, k, stride, pad, None, C.byref(ho), C.byref(wo)) == 0
    out = np.full((N, Cc, ho.value, wo.value), np.nan, np.float32)
    assert lib().ref_maxpool(_f(x), N, Cc, H, W, k, stride, pad, _f(out), None, None) == 0
    return out
