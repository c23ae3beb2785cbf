"""caffe_rtpose_amd — MI355X-native realtime multi-person pose engine (hot path of caffe_rtpose).

The product is `librtpose_mi355x.so` (hand-written HIP for gfx950 behind the C-ABI declared in
include/rtpose_mi355x.h) plus the C++ host `rtpose.bin`.  This Python package is only a thin
ctypes mirror of that ABI for tests and bench.py; it contains NO compute and NO CPU fallback:
importing it fails loudly if the HIP library has not been built.
"""
from ._lib import lib, LIB_PATH  # noqa: F401  (raises ImportError when the .so is missing)
from .engine import (Config, Engine, RtpError, model_tables, default_thresholds, process_and_pad_image,
                     format_json, prototxt_summary, plan_summary, device_local_cpus, synth_weights,
                     write_synthetic_caffemodel, read_caffemodel_layers, write_builtin_prototxt, resize_area, warp_display, preprocess_frame, synth_frame, load_image, decode_image, encode_jpeg, Video, MODEL_COCO_18, MODEL_MPI_15, PREC_FP16, PREC_FP32, PREC_MIXED, PREC_F16X3, EXEC_GRAPH, EXEC_EAGER,
                     MAX_PEOPLE, RTP_OK, RTP_EINVAL, RTP_ENOMEM, RTP_ENODEV, RTP_EIO, RTP_EAGAIN, RTP_EHIP, RTP_ERANGE)  # noqa: F401
