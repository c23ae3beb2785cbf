"""ctypes mirror of include/rtpose_mi355x.h (numpy in / numpy out).  Plumbing only."""
import ctypes as C

import numpy as np

from ._lib import lib, rtp_config, fp, ip

MODEL_COCO_18, MODEL_MPI_15 = 0, 1
PREC_FP16, PREC_FP32, PREC_MIXED, PREC_F16X3 = 0, 1, 2, 3
EXEC_GRAPH, EXEC_EAGER = 0, 1
MAX_PEOPLE = 96


# error codes of include/rtpose_mi355x.h
RTP_OK, RTP_EINVAL, RTP_ENOMEM, RTP_ENODEV, RTP_EIO, RTP_EAGAIN, RTP_EHIP, RTP_ERANGE = 0, -22, -12, -19, -5, -11, -70, -34


class RtpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtp error {code}: {msg}")
        self.code = code


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(fp)


class Config:
    def __init__(self, **kw):
        self.c = rtp_config()
        lib.rtp_config_default(C.byref(self.c))
        self._keep = []
        for k, v in kw.items():
            self.set(k, v)

    def set(self, k, v):
        if k in ("proto_path", "weights_path", "split_layers"):
            v = None if v is None else str(v).encode()
            self._keep.append(v)
        setattr(self.c, k, v)
        return self


def model_tables(model):
    npart, nlimb = C.c_int(), C.c_int()
    limb = (C.c_int * 38)()
    mp = (C.c_int * 38)()
    rc = lib.rtp_model_tables(model, C.byref(npart), C.byref(nlimb), limb, mp)
    if rc:
        raise RtpError(rc, "bad model")
    n = nlimb.value * 2
    return npart.value, nlimb.value, list(limb)[:n], list(mp)[:n]


def default_thresholds(model):
    a, b, e = C.c_float(), C.c_float(), C.c_float()
    c, d = C.c_int(), C.c_int()
    rc = lib.rtp_default_thresholds(model, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e))
    if rc:
        raise RtpError(rc, "bad model")
    return dict(nms_threshold=a.value, inter_threshold=b.value, inter_min_above=c.value, min_subset_cnt=d.value,
                min_subset_score=e.value)


def process_and_pad_image(img_u8, tw, th, normalize):
    oh, ow, _ = img_u8.shape
    out = np.empty((3, th, tw), np.float32)
    img = np.ascontiguousarray(img_u8, np.uint8)
    rc = lib.rtp_process_and_pad_image(_f(out), img.ctypes.data_as(C.POINTER(C.c_ubyte)), ow, oh, tw, th, int(normalize))
    if rc:
        raise RtpError(rc, "Image too big for target size.")
    return out


def format_json(joints, num_people, num_parts, frame_scale):
    buf = C.create_string_buffer(1 << 20)
    j = np.ascontiguousarray(joints, np.float32).reshape(-1)
    if j.size == 0:
        j = np.zeros(1, np.float32)
    n = lib.rtp_format_json(buf, len(buf), _f(j), num_people, num_parts, C.c_float(frame_scale))
    if n < 0:
        raise RtpError(n, "format_json")
    return buf.raw[:n]


def _u8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def resize_area(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((dh, dw, 3), np.uint8)
    rc = lib.rtp_resize_area(_u8(img), img.shape[1], img.shape[0], _u8(out), dw, dh)
    if rc:
        raise RtpError(rc, "resize_area")
    return out


def warp_display(img, disp_w, disp_h):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((disp_h, disp_w, 3), np.uint8)
    s = C.c_double()
    rc = lib.rtp_warp_display(_u8(img), img.shape[1], img.shape[0], _u8(out), disp_w, disp_h, C.byref(s))
    if rc:
        raise RtpError(rc, "warp_display")
    return out, s.value


def preprocess_frame(img, disp_w, disp_h, net_w, net_h, num_scales=1, start_scale=1.0, scale_gap=0.3):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((num_scales, 3, net_h, net_w), np.float32)
    disp = np.empty((disp_h, disp_w, 3), np.uint8)
    fs = C.c_float()
    rc = lib.rtp_preprocess_frame(_u8(img), img.shape[1], img.shape[0], disp_w, disp_h, net_w, net_h, num_scales, start_scale, scale_gap,
                                  _f(out), _u8(disp), C.byref(fs))
    if rc:
        raise RtpError(rc, "preprocess_frame")
    return out, disp, fs.value


def synth_frame(w, h, index, seed=2):
    out = np.empty((h, w, 3), np.uint8)
    rc = lib.rtp_synth_frame(_u8(out), w, h, index, seed)
    if rc:
        raise RtpError(rc, "synth_frame")
    return out


def load_image(path):
    w, h = C.c_int(), C.c_int()
    rc = lib.rtp_load_image(str(path).encode(), None, 0, C.byref(w), C.byref(h))
    if rc:
        raise RtpError(rc, f"cannot decode {path}")
    out = np.empty((h.value, w.value, 3), np.uint8)
    rc = lib.rtp_load_image(str(path).encode(), _u8(out), out.size, C.byref(w), C.byref(h))
    if rc:
        raise RtpError(rc, f"cannot decode {path}")
    return out


def prototxt_summary(path):
    nl, nc, npart, mp, hc = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    thr = C.c_float()
    rc = lib.rtp_prototxt_summary(str(path).encode(), C.byref(nl), C.byref(nc), C.byref(npart), C.byref(mp), C.byref(thr), C.byref(hc))
    if rc:
        raise RtpError(rc, lib.rtp_last_error(None).decode())
    return dict(num_layers=nl.value, num_conv=nc.value, num_parts=npart.value, max_peaks=mp.value,
                nms_threshold=thr.value, heat_channels=hc.value)


def synth_weights(seed, name, cout, cin, k):
    w = np.empty((cout, cin, k, k), np.float32)
    b = np.empty((cout,), np.float32)
    rc = lib.rtp_synth_weights(seed, name.encode(), cout, cin, k, _f(w), _f(b))
    if rc:
        raise RtpError(rc, "synth_weights")
    return w, b


def write_synthetic_caffemodel(model, seed, path):
    rc = lib.rtp_write_synthetic_caffemodel(model, seed, str(path).encode())
    if rc:
        raise RtpError(rc, lib.rtp_last_error(None).decode())


def write_builtin_prototxt(model, path):
    rc = lib.rtp_write_builtin_prototxt(model, str(path).encode())
    if rc:
        raise RtpError(rc, lib.rtp_last_error(None).decode())


def read_caffemodel_layers(path):
    n = lib.rtp_caffemodel_layer(str(path).encode(), -1, None, 0, None, None, None, None)
    if n < 0:
        raise RtpError(n, lib.rtp_last_error(None).decode())
    out = []
    name = C.create_string_buffer(128)
    nb = C.c_int()
    c0, c1 = C.c_long(), C.c_long()
    head = np.zeros(8, np.float32)
    for i in range(n):
        lib.rtp_caffemodel_layer(str(path).encode(), i, name, 128, C.byref(nb), C.byref(c0), C.byref(c1), _f(head))
        out.append(dict(name=name.value.decode(), num_blobs=nb.value, count0=c0.value, count1=c1.value, head0=head.copy()))
    return out


def decode_image(data):
    """cv::imread(IMREAD_COLOR) of an encoded PNG / JPEG byte string -> BGR HWC u8."""
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    w, h = C.c_int(), C.c_int()
    rc = lib.rtp_decode_image(buf, len(data), None, 0, C.byref(w), C.byref(h))
    if rc:
        raise RtpError(rc, lib.rtp_codec_last_error().decode())
    out = np.empty((h.value, w.value, 3), np.uint8)
    rc = lib.rtp_decode_image(buf, len(data), _u8(out), out.size, C.byref(w), C.byref(h))
    if rc:
        raise RtpError(rc, lib.rtp_codec_last_error().decode())
    return out


def encode_jpeg(bgr, quality=98):
    """cv::imwrite(.jpg) of a BGR HWC u8 image -> bytes."""
    img = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty(img.size * 2 + 4096, np.uint8)   # generous: one pass instead of a size query + encode
    n = lib.rtp_encode_jpeg(_u8(img), img.shape[1], img.shape[0], quality, _u8(out), out.size)
    if n < 0:
        raise RtpError(n, lib.rtp_codec_last_error().decode())
    return out[:n].tobytes()


class Video:
    """cv::VideoCapture for Y4M / raw MJPEG files."""

    def __init__(self, path):
        self.h = C.c_void_p()
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        rc = lib.rtp_video_open(str(path).encode(), C.byref(self.h), C.byref(w), C.byref(h), C.byref(n))
        if rc:
            self.h = C.c_void_p()
            raise RtpError(rc, lib.rtp_codec_last_error().decode())
        self.w, self.h_, self.nframes = w.value, h.value, n.value

    def read(self):
        out = np.empty((self.h_, self.w, 3), np.uint8)
        rc = lib.rtp_video_read(self.h, _u8(out), out.size)
        if rc == -11:
            return None
        if rc:
            raise RtpError(rc, lib.rtp_codec_last_error().decode())
        return out

    def close(self):
        if self.h:
            lib.rtp_video_close(self.h)
            self.h = C.c_void_p()


def device_local_cpus(device_id):
    """rtp_device_local_cpus: the CPUs next to a GPU's PCI function as a list of ints ([] where the platform does not say)."""
    buf = C.create_string_buffer(1024)
    if lib.rtp_device_local_cpus(int(device_id), buf, len(buf)) <= 0:
        return []
    cpus = []
    for part in buf.value.decode().split(","):
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


def plan_summary(cfg):
    buf = C.create_string_buffer(1 << 20)
    n = lib.rtp_plan_summary(C.byref(cfg.c), buf, len(buf))
    if n < 0:
        raise RtpError(n, lib.rtp_last_error(None).decode())
    return buf.raw[:n].decode()


class Engine:
    """One engine per GPU worker, like one caffe::Net per processFrame thread (rtpose.cpp:1463-1472)."""

    def __init__(self, cfg=None, **kw):
        self.cfg = cfg or Config(**kw)
        self.h = C.c_void_p()
        rc = lib.rtp_engine_create(C.byref(self.cfg.c), C.byref(self.h))
        if rc:
            self.h = C.c_void_p()
            raise RtpError(rc, lib.rtp_last_error(None).decode())
        a, b, c, d, e = (C.c_int() for _ in range(5))
        lib.rtp_engine_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e))
        self.num_parts, self.max_peaks, self.heat_channels, self.low_w, self.low_h = a.value, b.value, c.value, d.value, e.value
        self.N = self.cfg.c.num_scales
        self.net_w, self.net_h = self.cfg.c.net_w, self.cfg.c.net_h

    def _chk(self, rc):
        if rc:
            raise RtpError(rc, lib.rtp_last_error(self.h).decode())

    def close(self):
        if self.h:
            lib.rtp_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- hot loop
    def submit(self, x, tag=0):
        x = np.ascontiguousarray(x, np.float32)
        assert x.size == self.N * 3 * self.net_h * self.net_w
        self._chk(lib.rtp_submit(self.h, _f(x), tag))

    def submit_device(self, dptr, tag=0):
        self._chk(lib.rtp_submit_device(self.h, C.c_void_p(dptr), tag))

    def submit_frame(self, img_u8, tag=0):
        img = np.ascontiguousarray(img_u8, np.uint8)
        fs = C.c_float()
        self._chk(lib.rtp_submit_frame(self.h, img.ctypes.data_as(C.POINTER(C.c_ubyte)), img.shape[1], img.shape[0], tag, C.byref(fs)))
        return fs.value

    def debug_preprocess(self, img_u8):
        img = np.ascontiguousarray(img_u8, np.uint8)
        x = np.empty((self.N, 3, self.net_h, self.net_w), np.float32)
        disp = np.empty((self.cfg.c.disp_h, self.cfg.c.disp_w, 3), np.uint8)
        fs = C.c_float()
        self._chk(lib.rtp_debug_preprocess(self.h, img.ctypes.data_as(C.POINTER(C.c_ubyte)), img.shape[1], img.shape[0], _f(x),
                                           disp.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(fs)))
        return x, disp, fs.value

    def profile_steps(self, iters=20):
        ms = (C.c_float * 256)()
        gf = (C.c_double * 256)()
        n = lib.rtp_profile_steps(self.h, iters, ms, gf, 256)
        if n < 0:
            self._chk(n)
        return list(ms)[:n], list(gf)[:n]

    def post_from_lowres(self, lowres, peaks_in=None):
        lowres = np.ascontiguousarray(lowres, np.float32)
        peaks = np.zeros((self.num_parts, self.max_peaks + 1, 3), np.float32) if peaks_in is None else np.ascontiguousarray(peaks_in, np.float32).copy()
        joints = np.zeros((MAX_PEOPLE, self.num_parts, 3), np.float32)
        n = C.c_int()
        self._chk(lib.rtp_post_from_lowres(self.h, _f(lowres), _f(peaks), _f(joints), C.byref(n)))
        return peaks, joints[: n.value].copy(), n.value

    def render(self, display_bgr, joints, num_people, part_to_show=0, googly=0, resized=None):
        """render() of rtpose.cpp:270-299 on caller data: pose overlay (part_to_show 0) or a heat-map / PAF view of `resized`."""
        img = np.ascontiguousarray(display_bgr, np.uint8)
        assert img.shape == (self.cfg.c.disp_h, self.cfg.c.disp_w, 3), img.shape
        j = np.ascontiguousarray(joints, np.float32).reshape(-1)
        if j.size == 0:
            j = np.zeros(3, np.float32)
        res = None if resized is None else np.ascontiguousarray(resized, np.float32)
        out = np.empty_like(img)
        self._chk(lib.rtp_render(self.h, _u8(img), _f(j), int(num_people), int(part_to_show), int(googly), _f(res) if res is not None else None, _u8(out)))
        return out

    def collect_rendered(self):
        tag = C.c_uint64()
        n = C.c_int()
        joints = np.zeros((MAX_PEOPLE, self.num_parts, 3), np.float32)
        img = np.empty((self.cfg.c.disp_h, self.cfg.c.disp_w, 3), np.uint8)
        self._chk(lib.rtp_collect_rendered(self.h, C.byref(tag), _f(joints), C.byref(n), _u8(img)))
        return tag.value, n.value, joints[: n.value].copy(), img

    def flush(self):
        self._chk(lib.rtp_flush(self.h))

    def collect(self):
        tag = C.c_uint64()
        n = C.c_int()
        joints = np.zeros((MAX_PEOPLE, self.num_parts, 3), np.float32)
        self._chk(lib.rtp_collect(self.h, C.byref(tag), _f(joints), C.byref(n)))
        return tag.value, n.value, joints[: n.value].copy()

    def in_flight(self):
        return lib.rtp_in_flight(self.h)

    # ---- thresholds / scales
    def set_thresholds(self, nms_threshold, inter_threshold, inter_min_above, min_subset_cnt, min_subset_score):
        self._chk(lib.rtp_set_thresholds(self.h, nms_threshold, inter_threshold, inter_min_above, min_subset_cnt, min_subset_score))

    def get_thresholds(self):
        a, b, e = C.c_float(), C.c_float(), C.c_float()
        c, d = C.c_int(), C.c_int()
        self._chk(lib.rtp_get_thresholds(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)))
        return dict(nms_threshold=a.value, inter_threshold=b.value, inter_min_above=c.value, min_subset_cnt=d.value,
                    min_subset_score=e.value)

    def set_scales(self, start_scale, scale_gap):
        self._chk(lib.rtp_set_scales(self.h, start_scale, scale_gap))

    # ---- parity taps
    def forward_heatmaps(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty((self.N, self.heat_channels, self.low_h, self.low_w), np.float32)
        self._chk(lib.rtp_forward_heatmaps(self.h, _f(x), _f(out)))
        return out

    def resize(self, lowres):
        lowres = np.ascontiguousarray(lowres, np.float32)
        assert lowres.shape == (self.N, self.heat_channels, self.low_h, self.low_w)
        out = np.empty((self.heat_channels, self.net_h, self.net_w), np.float32)
        self._chk(lib.rtp_resize(self.h, _f(lowres), _f(out)))
        return out

    def nms(self, resized, peaks_init=None):
        resized = np.ascontiguousarray(resized, np.float32).reshape(self.heat_channels, self.net_h, self.net_w)
        peaks = np.zeros((self.num_parts, self.max_peaks + 1, 3), np.float32) if peaks_init is None else np.ascontiguousarray(peaks_init, np.float32).copy()
        self._chk(lib.rtp_nms(self.h, _f(resized), _f(peaks)))
        return peaks

    def connect(self, resized, peaks):
        resized = np.ascontiguousarray(resized, np.float32).reshape(self.heat_channels, self.net_h, self.net_w)
        peaks = np.ascontiguousarray(peaks, np.float32)
        joints = np.zeros((MAX_PEOPLE, self.num_parts, 3), np.float32)
        n = C.c_int()
        self._chk(lib.rtp_connect(self.h, _f(resized), _f(peaks), _f(joints), C.byref(n)))
        return n.value, joints

    def forward_debug(self, x):
        x = np.ascontiguousarray(x, np.float32)
        lowres = np.empty((self.N, self.heat_channels, self.low_h, self.low_w), np.float32)
        resized = np.empty((self.heat_channels, self.net_h, self.net_w), np.float32)
        peaks = np.empty((self.num_parts, self.max_peaks + 1, 3), np.float32)
        joints = np.zeros((MAX_PEOPLE, self.num_parts, 3), np.float32)
        n = C.c_int()
        self._chk(lib.rtp_forward_debug(self.h, _f(x), _f(lowres), _f(resized), _f(peaks), _f(joints), C.byref(n)))
        return dict(lowres=lowres, resized=resized, peaks=peaks, joints=joints, num_people=n.value)

    def get_blob(self, name):
        shape = (C.c_int * 4)()
        self._chk(lib.rtp_get_blob(self.h, name.encode(), None, 0, shape))
        out = np.empty(tuple(shape), np.float32)
        self._chk(lib.rtp_get_blob(self.h, name.encode(), _f(out), out.size, shape))
        return out

    def connect_stats(self):
        nl = 19 if self.num_parts == 18 else 14
        a = (C.c_int * nl)()
        b = (C.c_int * nl)()
        self._chk(lib.rtp_debug_connect_stats(self.h, a, b))
        return [v & ((1 << 30) - 1) for v in a], list(b), [v >> 30 for v in a]

    # ---- weights
    def conv_layers(self):
        res = []
        name = C.create_string_buffer(64)
        cin, cout, k = C.c_int(), C.c_int(), C.c_int()
        for i in range(lib.rtp_num_conv_layers(self.h)):
            lib.rtp_conv_layer_info(self.h, i, name, 64, C.byref(cin), C.byref(cout), C.byref(k))
            res.append((name.value.decode(), cin.value, cout.value, k.value))
        return res

    def get_conv_weights(self, i):
        _, cin, cout, k = self.conv_layers()[i]
        w = np.empty((cout, cin, k, k), np.float32)
        b = np.empty((cout,), np.float32)
        self._chk(lib.rtp_get_conv_weights(self.h, i, _f(w), _f(b)))
        return w, b

    def set_conv_weights(self, i, w, b):
        w = np.ascontiguousarray(w, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        self._chk(lib.rtp_set_conv_weights(self.h, i, _f(w), _f(b)))

    def save_caffemodel(self, path):
        self._chk(lib.rtp_save_caffemodel(self.h, str(path).encode()))

    def save_prototxt(self, path):
        self._chk(lib.rtp_save_prototxt(self.h, str(path).encode()))

    # ---- load-time precision calibration
    def calibrate_precision(self, frames=None, nframes=2, target=0.7e-3):
        """rtp_calibrate_precision: returns (rules, err_before, err_after); frames = [n][N][3][H][W] float32 or None (synthetic)."""
        buf = C.create_string_buffer(4096)
        a, b = C.c_float(), C.c_float()
        if frames is not None:
            frames = np.ascontiguousarray(frames, np.float32)
            nframes = frames.size // (self.N * 3 * self.net_h * self.net_w)
        self._chk(lib.rtp_calibrate_precision(self.h, _f(frames) if frames is not None else None, int(nframes), float(target), buf, len(buf), C.byref(a), C.byref(b)))
        return buf.value.decode(), a.value, b.value

    def calibration_report(self):
        return lib.rtp_calibration_report(self.h).decode()

    def split_layers(self):
        buf = C.create_string_buffer(1 << 16)   # (rtp_get_split_layers returns RTP_ERANGE instead of truncating)
        p = C.c_int()
        self._chk(lib.rtp_get_split_layers(self.h, buf, len(buf), C.byref(p)))
        return buf.value.decode(), p.value

    # ---- caller-owned device buffers (the engine's HIP runtime; bench.py keeps torch out of the data path)
    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(lib.rtp_device_alloc(self.h, int(nbytes), C.byref(p)))
        return p.value

    def device_free(self, dptr):
        self._chk(lib.rtp_device_free(self.h, C.c_void_p(dptr)))

    def device_upload(self, dptr, arr):
        a = np.ascontiguousarray(arr)
        self._chk(lib.rtp_device_upload(self.h, C.c_void_p(dptr), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def device_frame(self, x):
        """A net input resident in HBM (for submit_device): allocates on the engine's device and uploads x."""
        x = np.ascontiguousarray(x, np.float32)
        p = self.device_alloc(x.nbytes)
        self.device_upload(p, x)
        return p

    def synchronize(self):
        self._chk(lib.rtp_device_synchronize(self.h))

    # ---- one-time weight distribution between replicas
    def weight_blob_bytes(self):
        """Length of this plan's weight blob without exporting one (no D2H copy of the arena)."""
        n = lib.rtp_weight_blob_bytes(self.h)
        if n < 0:
            raise RtpError(n, "rtp_weight_blob_bytes")
        return int(n)

    def weight_blob(self):
        n = lib.rtp_weight_blob_bytes(self.h)
        buf = np.empty(n, np.uint8)
        self._chk(lib.rtp_weight_blob_export(self.h, buf.ctypes.data_as(C.c_void_p), n))
        return buf

    def load_weight_blob(self, buf):
        b = np.ascontiguousarray(buf, np.uint8)
        self._chk(lib.rtp_weight_blob_import(self.h, b.ctypes.data_as(C.c_void_p), b.nbytes))

    def copy_weights_from(self, other):
        self._chk(lib.rtp_copy_weights_from(self.h, other.h))

    # ---- diagnostics
    def last_stage_ms(self):
        ms = (C.c_float * 5)()
        lib.rtp_last_stage_ms(self.h, ms)
        return dict(conv=ms[0], resize=ms[1], nms=ms[2], connect=ms[3], total=ms[4])

    def kernel_timing(self, enable=-1):
        tot, fl = C.c_double(), C.c_double()
        n = C.c_long()
        self._chk(lib.rtp_kernel_timing(self.h, enable, C.byref(tot), C.byref(n), C.byref(fl)))
        return tot.value, n.value, fl.value

    def kernel_timing_by_passes(self):
        ms = (C.c_double * 4)()
        n = (C.c_long * 4)()
        self._chk(lib.rtp_kernel_timing_by_passes(self.h, ms, n))
        return {p: (ms[p], n[p]) for p in (1, 2, 3) if n[p]}

    def kernel_timing_steps(self):
        """[(ms, launches)] per plan step of a timing pass switched on with kernel_timing(3) (harvest with kernel_timing(-1) first)."""
        cap = 512
        ms = (C.c_double * cap)()
        n = (C.c_long * cap)()
        k = lib.rtp_kernel_timing_steps(self.h, ms, n, cap)
        if k < 0:
            raise RtpError(k, "rtp_kernel_timing_steps")
        return [(ms[i], n[i]) for i in range(min(k, cap))]

    def busy_probe(self, enable=-1):
        """rtp_busy_probe: returns the spans recorded so far as an [n][3] float32 array {kind, start_ms, end_ms}."""
        n = lib.rtp_busy_probe(self.h, enable, None, 0)
        if n < 0:
            raise RtpError(n, lib.rtp_last_error(self.h).decode())
        out = np.zeros((n, 3), np.float32)
        if n:
            lib.rtp_busy_probe(self.h, -1, _f(out), n)
        return out

    def probe_dropped(self):
        """rtp_probe_dropped: {timing_pairs, busy_spans, busy_graph_frames} the probes could not record since creation."""
        out = (C.c_long * 3)()
        self._chk(lib.rtp_probe_dropped(self.h, out))
        return {"timing_pairs": out[0], "busy_spans": out[1], "busy_graph_frames": out[2]}

    def stamp_probe(self, enable=-1):
        """rtp_stamp_probe: kernel residency spans as an [n][3] float32 array {slot, start_us, end_us} (device-side wall-clock stamps)."""
        n = lib.rtp_stamp_probe(self.h, enable, None, 0)
        if n < 0:
            raise RtpError(n, lib.rtp_last_error(self.h).decode())
        out = np.zeros((n, 3), np.float32)
        if n:
            lib.rtp_stamp_probe(self.h, -1, _f(out), n)
        return out

    def bench_dominant_conv(self, iters=50):
        ms = C.c_float()
        fl = C.c_double()
        self._chk(lib.rtp_bench_dominant_conv(self.h, iters, C.byref(ms), C.byref(fl)))
        return ms.value, fl.value
