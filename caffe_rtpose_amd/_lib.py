"""Loads librtpose_mi355x.so.  No fallback of any kind."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTP_LIB") or os.path.join(_HERE, "librtpose_mi355x.so")  # RTP_LIB: kernel experiments with an alternative build

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP engine first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C caffe_rtpose_amd/csrc). "
        "There is no CPU/PyTorch fallback for this path.")

lib = C.CDLL(LIB_PATH)


class rtp_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint),
        ("device_id", C.c_int),
        ("model", C.c_int),
        ("proto_path", C.c_char_p),
        ("weights_path", C.c_char_p),
        ("synthetic_seed", C.c_uint64),
        ("net_w", C.c_int),
        ("net_h", C.c_int),
        ("num_scales", C.c_int),
        ("start_scale", C.c_float),
        ("scale_gap", C.c_float),
        ("disp_w", C.c_int),
        ("disp_h", C.c_int),
        ("precision", C.c_int),
        ("frames_in_flight", C.c_int),
        ("batch_frames", C.c_int),
        ("render", C.c_int),
        ("exec_mode", C.c_int),
        ("split_layers", C.c_char_p),
        ("keep_blobs", C.c_int),
        ("calibrate_frames", C.c_int),
        ("calibrate_target", C.c_float),
        ("defer_weights", C.c_int),
    ]


fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)
vp = C.c_void_p

# every symbol include/rtpose_mi355x.h declares, with its signature
SIGNATURES = {
    "rtp_config_default": (C.c_int, [C.POINTER(rtp_config)]),
    "rtp_engine_create": (C.c_int, [C.POINTER(rtp_config), C.POINTER(vp)]),
    "rtp_engine_destroy": (None, [vp]),
    "rtp_engine_info": (C.c_int, [vp, ip, ip, ip, ip, ip]),
    "rtp_set_thresholds": (C.c_int, [vp, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float]),
    "rtp_get_thresholds": (C.c_int, [vp, fp, fp, ip, ip, fp]),
    "rtp_set_scales": (C.c_int, [vp, C.c_float, C.c_float]),
    "rtp_debug_f32_to_e4m3": (C.c_int, [fp, C.POINTER(C.c_ubyte), C.c_int]),
    "rtp_kernel_timing_by_passes": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "rtp_kernel_timing_steps": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_long), C.c_int]),
    "rtp_busy_probe": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "rtp_probe_dropped": (C.c_int, [vp, C.POINTER(C.c_long)]),
    "rtp_stamp_probe": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "rtp_submit": (C.c_int, [vp, fp, C.c_uint64]),
    "rtp_submit_device": (C.c_int, [vp, vp, C.c_uint64]),
    "rtp_submit_frame": (C.c_int, [vp, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_uint64, fp]),
    "rtp_debug_preprocess": (C.c_int, [vp, C.POINTER(C.c_ubyte), C.c_int, C.c_int, fp, C.POINTER(C.c_ubyte), fp]),
    "rtp_flush": (C.c_int, [vp]),
    "rtp_collect_rendered": (C.c_int, [vp, C.POINTER(C.c_uint64), fp, ip, C.POINTER(C.c_ubyte)]),
    "rtp_render": (C.c_int, [vp, C.POINTER(C.c_ubyte), fp, C.c_int, C.c_int, C.c_int, fp, C.POINTER(C.c_ubyte)]),
    "rtp_decode_image": (C.c_int, [C.POINTER(C.c_ubyte), C.c_size_t, C.POINTER(C.c_ubyte), C.c_size_t, ip, ip]),
    "rtp_codec_last_error": (C.c_char_p, []),
    "rtp_encode_jpeg": (C.c_long, [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_size_t]),
    "rtp_video_open": (C.c_int, [C.c_char_p, C.POINTER(vp), ip, ip, ip]),
    "rtp_video_read": (C.c_int, [vp, C.POINTER(C.c_ubyte), C.c_size_t]),
    "rtp_video_close": (None, [vp]),
    "rtp_post_from_lowres": (C.c_int, [vp, fp, fp, fp, ip]),
    "rtp_profile_steps": (C.c_int, [vp, C.c_int, fp, C.POINTER(C.c_double), C.c_int]),
    "rtp_collect": (C.c_int, [vp, C.POINTER(C.c_uint64), fp, ip]),
    "rtp_in_flight": (C.c_int, [vp]),
    "rtp_forward_heatmaps": (C.c_int, [vp, fp, fp]),
    "rtp_resize": (C.c_int, [vp, fp, fp]),
    "rtp_nms": (C.c_int, [vp, fp, fp]),
    "rtp_connect": (C.c_int, [vp, fp, fp, fp, ip]),
    "rtp_forward_debug": (C.c_int, [vp, fp, fp, fp, fp, fp, ip]),
    "rtp_get_blob": (C.c_int, [vp, C.c_char_p, fp, C.c_size_t, ip]),
    "rtp_num_conv_layers": (C.c_int, [vp]),
    "rtp_conv_layer_info": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_int, ip, ip, ip]),
    "rtp_get_conv_weights": (C.c_int, [vp, C.c_int, fp, fp]),
    "rtp_set_conv_weights": (C.c_int, [vp, C.c_int, fp, fp]),
    "rtp_save_caffemodel": (C.c_int, [vp, C.c_char_p]),
    "rtp_save_prototxt": (C.c_int, [vp, C.c_char_p]),
    "rtp_model_tables": (C.c_int, [C.c_int, ip, ip, ip, ip]),
    "rtp_default_thresholds": (C.c_int, [C.c_int, fp, fp, ip, ip, fp]),
    "rtp_process_and_pad_image": (C.c_int, [fp, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rtp_format_json": (C.c_long, [C.c_char_p, C.c_size_t, fp, C.c_int, C.c_int, C.c_float]),
    "rtp_display_fit_scale": (C.c_double, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "rtp_resize_area": (C.c_int, [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int]),
    "rtp_warp_display": (C.c_int, [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "rtp_preprocess_frame": (C.c_int, [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, fp, C.POINTER(C.c_ubyte), fp]),
    "rtp_load_image": (C.c_int, [C.c_char_p, C.POINTER(C.c_ubyte), C.c_size_t, ip, ip]),
    "rtp_synth_frame": (C.c_int, [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.c_uint64]),
    "rtp_prototxt_summary": (C.c_int, [C.c_char_p, ip, ip, ip, ip, fp, ip]),
    "rtp_last_error": (C.c_char_p, [vp]),
    "rtp_version": (C.c_char_p, []),
    "rtp_last_stage_ms": (C.c_int, [vp, fp]),
    "rtp_debug_connect_stats": (C.c_int, [vp, ip, ip]),
    "rtp_synth_weights": (C.c_int, [C.c_uint64, C.c_char_p, C.c_int, C.c_int, C.c_int, fp, fp]),
    "rtp_write_synthetic_caffemodel": (C.c_int, [C.c_int, C.c_uint64, C.c_char_p]),
    "rtp_write_builtin_prototxt": (C.c_int, [C.c_int, C.c_char_p]),
    "rtp_caffemodel_layer": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, ip, C.POINTER(C.c_long), C.POINTER(C.c_long), fp]),
    "rtp_plan_summary": (C.c_long, [C.POINTER(rtp_config), C.c_char_p, C.c_size_t]),
    "rtp_kernel_timing": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]),
    "rtp_bench_dominant_conv": (C.c_int, [vp, C.c_int, fp, C.POINTER(C.c_double)]),
    "rtp_calibrate_precision": (C.c_int, [vp, fp, C.c_int, C.c_float, C.c_char_p, C.c_size_t, fp, fp]),
    "rtp_calibration_report": (C.c_char_p, [vp]),
    "rtp_get_split_layers": (C.c_int, [vp, C.c_char_p, C.c_size_t, ip]),
    "rtp_device_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "rtp_device_free": (C.c_int, [vp, vp]),
    "rtp_device_upload": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "rtp_device_synchronize": (C.c_int, [vp]),
    "rtp_weight_blob_bytes": (C.c_long, [vp]),
    "rtp_weight_blob_export": (C.c_int, [vp, vp, C.c_size_t]),
    "rtp_weight_blob_import": (C.c_int, [vp, vp, C.c_size_t]),
    "rtp_copy_weights_from": (C.c_int, [vp, vp]),
    "rtp_device_local_cpus": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args
