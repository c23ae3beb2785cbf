/* rtpose_mi355x.h — C-ABI of the MI355X-native realtime multi-person pose engine.
 *
 * This is the drop-in boundary for the ONE hot path of CMU's caffe_rtpose:
 *   float NCHW net input -> VGG-19+CPM conv stack -> ImResize -> Nms -> connectLimbs* -> joints
 * Every entry point names the reference interface it replaces (file:line relative to the
 * reference tree).  The reference consumes that path through the Caffe Net API
 * (examples/rtpose/rtpose.cpp:183-207, 1127-1166); a maintainer keeps rtpose.cpp's thread
 * structure and swaps those calls for the ones below.
 *
 * Conventions
 *   - plain C types only; caller owns every buffer it passes; all functions return 0 on
 *     success or a negative RTP_E* code and NEVER abort (the reference's glog CHECK /
 *     CUDA_CHECK abort the process, rtpose.cpp:208,247; common.hpp CUDA_CHECK).
 *   - an engine is single-owner and not thread-safe: one engine per GPU worker thread,
 *     exactly like one caffe::Net per processFrame thread (rtpose.cpp:1463-1472).
 *   - host buffers may be pageable or pinned; "_device" variants take HIP device pointers.
 *   - there is NO CPU fallback: without a gfx950 device rtp_engine_create fails with
 *     RTP_ENODEV.
 */
#ifndef RTPOSE_MI355X_H_
#define RTPOSE_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTP_OK 0
#define RTP_EINVAL (-22)   /* bad argument / bad configuration                       */
#define RTP_ENOMEM (-12)   /* host or device allocation failed                       */
#define RTP_ENODEV (-19)   /* no usable gfx950 device                                */
#define RTP_EIO (-5)       /* file could not be read / parsed                        */
#define RTP_EAGAIN (-11)   /* submit queue full / nothing to collect                 */
#define RTP_EHIP (-70)     /* a HIP runtime call failed (see rtp_last_error)         */
#define RTP_ERANGE (-34)   /* out-of-contract data (e.g. sample coordinate < 0)      */

#define RTP_MODEL_COCO_18 0 /* ModelDescriptorFactory::Type::COCO_18 (modelDescriptorFactory.cpp:30) */
#define RTP_MODEL_MPI_15 1  /* ModelDescriptorFactory::Type::MPI_15  (modelDescriptorFactory.cpp:6)  */

#define RTP_PREC_FP16 0 /* fp16 storage, MFMA f16 with fp32 accumulate: fastest, 2-2.6x OUTSIDE +-1e-3 */
#define RTP_PREC_FP32 1 /* fp32 storage, exact-f32 MFMA (parity path, 1/16 the MFMA rate)   */
#define RTP_PREC_MIXED 2 /* fp16 MFMA; the layers named by split_layers (default: the set whose fp16 rounding   *
                          * dominates the final-map error) also multiply the rounding errors of both operands:  *
                          * a_hi*W_hi + a_lo*W_hi + a_hi*W_lo into one fp32 accumulator.  On the 3x3 / 7x7      *
                          * layers the two corrections are ONE extra K chunk of MX-scaled fp8 MFMA (2 pass-     *
                          * times in all), on the 1x1 layers two more fp16 passes.  Reference arithmetic is     *
                          * fp32 throughout (base_conv_layer.cpp:257-280); this is the fastest mode that meets  *
                          * +-1e-3 on the maps.  rtp_calibrate_precision checks / widens the set on the loaded  *
                          * weights.                                                                             */
#define RTP_PREC_F16X3 3 /* every layer split: fp32-class accuracy at about 1/3 of the fp16 rate                 */

#define RTP_EXEC_GRAPH 0 /* default: the conv stack of one batch (~50 launches) is captured ONCE per     *
                          * (context, frames in the batch) into a hipGraph and replayed with one       *
                          * hipGraphLaunch; each frame's short post-processing chain + D2H is launched  *
                          * eagerly on the frame's own stream (rtp_collect waits for its frame only)    */
#define RTP_EXEC_EAGER 1 /* one HIP launch per kernel (diagnostics; per-stage event timing)            */

#define RTP_MAX_PEOPLE 96    /* RENDER_MAX_PEOPLE, include/rtpose/renderFunctions.h:6 */
#define RTP_MAX_NUM_PARTS 70 /* MAX_NUM_PARTS, rtpose.cpp:91                          */

typedef struct rtp_engine rtp_engine;

/* Replaces: flags + warmup() state (rtpose.cpp:50-72, 173-237). */
typedef struct rtp_config {
  unsigned int struct_size;/* sizeof(rtp_config) of the header rtp_config_default was COMPILED against (it writes  *
                            * it).  rtp_engine_create / rtp_plan_summary return RTP_EINVAL for any other value: a    *
                            * caller built against another version of this header (fields were appended in rounds     *
                            * 2-5) or one that skipped rtp_config_default is refused instead of being misread.        */
  int device_id;           /* --start_device + worker index (rtpose.cpp:1466)                     */
  int model;               /* RTP_MODEL_*; ignored when proto_path is given                        */
  const char* proto_path;  /* --caffeproto deploy prototxt, or NULL = built-in linevec net         */
  const char* weights_path;/* --caffemodel binary NetParameter, or NULL = synthetic (seed below)   */
  uint64_t synthetic_seed; /* deterministic He-scaled weights when weights_path == NULL            */
  int net_w, net_h;        /* --net_resolution, multiples of 16 (rtpose.cpp:65)                    */
  int num_scales;          /* --num_scales = blob N (rtpose.cpp:188, 1719)                          */
  float start_scale;       /* --start_scale -> ImResizeLayer::SetStartScale (rtpose.cpp:201)       */
  float scale_gap;         /* --scale_gap   -> ImResizeLayer::SetScaleGap   (rtpose.cpp:202)       */
  int disp_w, disp_h;      /* --resolution: joints are returned in display coordinates (:1061)     */
  int precision;           /* RTP_PREC_*                                                            */
  int frames_in_flight;    /* >=1: frames that may be submitted before a collect is needed         */
  int batch_frames;        /* >=1: frames whose conv stacks share one launch sequence (0 = 1).     *
                            * The reference runs one frame (num_scales images) per Forward; with   *
                            * B > 1 the engine stages B submitted frames and runs the stack over   *
                            * B*num_scales images (bigger tiles fill the 256 CUs), then post-      *
                            * processes each frame on its own stream.  Per-frame results do not    *
                            * depend on B.  ceil(frames_in_flight / B) batches are in flight.       */
  int render;              /* 0: off.  1 + FLAGS_part_to_show: rtp_collect_rendered also returns the  *
                            * display image as render() (rtpose.cpp:270-299) draws it: 1 = the pose    *
                            * overlay (render_pose_*, renderFunctions.cu), 2.. = heat-map / PAF view   *
                            * part_to_show = render - 1 (that frame's resized map is then materialised) */
  int exec_mode;           /* RTP_EXEC_*: how a batch's ~45 launches reach the GPU.  Replaces the  *
                            * reference's per-call layer walk (net.cpp:544-556 ForwardFromTo).       */
  const char* split_layers;/* RTP_PREC_MIXED only; NULL = default set.  Comma-separated rules, each *
                            * "<pattern>[:w|:a|:x]": pattern = layer-name prefix, "*text" = name      *
                            * contains text, "@1x1" = every 1x1 layer, "@all"; ":w" splits only the  *
                            * weights, ":a" only the input activations, none = both; ":x" = both,    *
                            * with the corrections as fp16 passes instead of the fp8 chunk (three     *
                            * pass-times, no e4m3 range limits: what the calibration switches a      *
                            * group to when the fp8 corrections are not accurate enough).             */
  int keep_blobs;          /* 0 (default): blobs that only feed a fused consumer are not written (the     *
                            * convolutions in front of the three pooling layers pool in their epilogue):    *
                            * rtp_get_blob("conv1_2" / "conv2_2" / "conv3_4") then returns RTP_EINVAL —      *
                            * UNLIKE Net::blob_by_name, which has every blob.  1: every blob of the graph    *
                            * stays tappable, pooling runs as its own launch; same values either way.        */
  int calibrate_frames;    /* > 0 (RTP_PREC_MIXED only): rtp_engine_create ends with                         *
                            * rtp_calibrate_precision(e, NULL, calibrate_frames, calibrate_target, ...) on    *
                            * synthetic frames, i.e. the split set is checked — and widened if necessary —    *
                            * on the weights that were just loaded (net.cpp:750-803).                          *
                            * 0 (default): that check runs with ONE frame when weights_path != NULL (weights   *
                            * from a file: the default set was chosen on synthetic ones; ~2-3 s, a line on     *
                            * stderr if the set had to change) and not at all for synthetic weights.           *
                            * -1: never (the split set stays exactly rtp_config.split_layers / the default).   */
  float calibrate_target;  /* max |mixed - f16x3| / max |map| the calibration accepts (<= 0: 0.7e-3)         */
  int defer_weights;       /* 1: a RECEIVING replica of a one-time weight distribution: the engine is built      *
                            * (plan, arena, contexts) but no weights are read, generated, packed or uploaded —    *
                            * the arena stays zero and every entry that would run the net returns RTP_EINVAL      *
                            * until rtp_weight_blob_import / rtp_copy_weights_from has delivered another          *
                            * engine's packed weights (same plan: see there); the launch graphs are captured      *
                            * then.  Saves the N-1 reads + packs of N replicas (rtpose.bin --share_weights,       *
                            * bench.py --broadcast_weights).  No calibration runs on such an engine: create it    *
                            * with the split set the source engine ended up with (rtp_get_split_layers).          */
} rtp_config;

/* Fill cfg with the reference's flag defaults (rtpose.cpp:50-72): COCO, 656x368, 1 scale,
 * start_scale 1, scale_gap 0.3, 1280x720, RTP_PREC_MIXED, 2 frames in flight, graph replay — and
 * struct_size.  Every rtp_config must start here. */
int rtp_config_default(rtp_config* cfg);

/* Replaces: new Net<float>(proto, TEST) + CopyTrainedLayersFrom + Reshape + dry run
 * (rtpose.cpp:183-191, 233; net.cpp:49, 750-803). */
int rtp_engine_create(const rtp_config* cfg, rtp_engine** out);
void rtp_engine_destroy(rtp_engine* e);

/* Replaces: NmsLayer::GetNumParts/GetMaxPeaks (nms_layer.hpp:11-44, rtpose.cpp:194-207) and
 * the resized_map blob shape.  heat_channels = channels of concat_stage7 (57 COCO / 44 MPI). */
int rtp_engine_info(const rtp_engine* e, int* num_parts, int* max_peaks, int* heat_channels,
                    int* low_w, int* low_h);

/* Replaces: the global.* thresholds written by warmup() and the UI thread
 * (rtpose.cpp:212-226) and NmsLayer::SetThreshold (rtpose.cpp:1145). */
int rtp_set_thresholds(rtp_engine* e, float nms_threshold, float connect_inter_threshold,
                       int connect_inter_min_above_threshold, int connect_min_subset_cnt,
                       float connect_min_subset_score);
int rtp_get_thresholds(const rtp_engine* e, float* nms_threshold, float* connect_inter_threshold,
                       int* connect_inter_min_above_threshold, int* connect_min_subset_cnt,
                       float* connect_min_subset_score);

/* Replaces: ImResizeLayer::SetStartScale/SetScaleGap (imresize_layer.hpp:11-45).  Every level s = start_scale - i * scale_gap
 * (i < num_scales) must be positive and its 16-aligned size 16 * ceil(net * s / 16) must fit the net input — what the producer CHECKs
 * (rtpose.cpp:363-364) — else RTP_EINVAL and the engine keeps its previous scales; the same test guards rtp_engine_create and
 * rtp_preprocess_frame.  start_scale < 1 is in contract: ImResize then crops every scale incl. the first (imresize_layer.cu:110-113). */
int rtp_set_scales(rtp_engine* e, float start_scale, float scale_gap);

/* ---- the per-frame hot loop (rtpose.cpp:1127-1166) ---------------------------------- */

/* Replaces: cudaMemcpy H2D into blobs()[0]->mutable_gpu_data() + ForwardFrom(0) + the
 * heatmap/peaks read-back + connectLimbs*().  Asynchronous: returns once the frame is
 * enqueued on one of the engine's frame contexts.  nchw_input: num_scales x 3 x net_h x net_w
 * floats, the output of process_and_pad_image (rtpose.cpp:239-269).  RTP_EAGAIN when all
 * frames_in_flight contexts are busy (collect first). */
int rtp_submit(rtp_engine* e, const float* nchw_input_host, uint64_t tag);
int rtp_submit_device(rtp_engine* e, const float* nchw_input_device, uint64_t tag);

/* Same frame path, but from the DECODED frame: u8 BGR HWC (any size) -> display-fit cubic warp ->
 * INTER_AREA scale pyramid -> u8/256-0.5 -> centre pad, all on the device (what the producer thread
 * does with OpenCV in rtpose.cpp:322-368), bit-identical to rtp_preprocess_frame.  *frame_scale
 * receives Frame::scale (for rtp_format_json). */
int rtp_submit_frame(rtp_engine* e, const unsigned char* bgr_host, int w, int h, uint64_t tag, float* frame_scale);
/* Parity tap for the device pre-processing alone. */
int rtp_debug_preprocess(rtp_engine* e, const unsigned char* bgr_host, int w, int h, float* net_input_host,
                         unsigned char* display_bgr_host, float* frame_scale);

/* rtp_collect + the display-resolution u8 BGR frame with the pose overlay: the image the reference
 * passes to cv::imwrite under --write_frames (rtpose.cpp:1179-1199 render, :1286-1293 float -> u8),
 * without the cv::putText overlays (= --no_text).  Needs rtp_config.render = 1 and rtp_submit_frame. */
int rtp_collect_rendered(rtp_engine* e, uint64_t* tag, float* joints_host, int* num_people, unsigned char* display_bgr_host);

/* Launch a partially filled batch now (batch_frames > 1: end of stream or a latency-sensitive
 * caller).  rtp_collect does this by itself when the oldest frame sits in an unlaunched batch. */
int rtp_flush(rtp_engine* e);

/* Blocks until the OLDEST submitted frame is finished; returns its tag, the number of people
 * (<= RTP_MAX_PEOPLE) and joints[num_people][num_parts][3] = (x, y, score) in display
 * coordinates — the array connectLimbs*() fills (rtpose.cpp:1051-1073).  joints must hold
 * RTP_MAX_PEOPLE*num_parts*3 floats.  RTP_EAGAIN if nothing is in flight. */
int rtp_collect(rtp_engine* e, uint64_t* tag, float* joints, int* num_people);

/* Number of frames submitted and not yet collected. */
int rtp_in_flight(const rtp_engine* e);

/* ---- parity taps (the Net::blob_by_name surface rtpose.cpp uses, :1093-1094, 1149-1150) - */

/* Run the conv stack only; lowres receives concat_stage7 as num_scales x C x h x w fp32. */
int rtp_forward_heatmaps(rtp_engine* e, const float* nchw_input_host, float* lowres_host);
/* ImResizeLayer::Forward_gpu on caller data: lowres (num_scales x C x h x w) -> resized (C x net_h x net_w). */
int rtp_resize(rtp_engine* e, const float* lowres_host, float* resized_host);
/* The PRODUCTION post-processing as a parity tap: rtp_submit/rtp_collect never materialise the
 * 55 MB resized map — NMS builds the resized rows of each 8-row strip in LDS and the PAF samples /
 * centroid windows are evaluated from the low-res maps on demand, with the arithmetic of
 * imresize_layer.cu, so the results equal rtp_resize -> rtp_nms -> rtp_connect bit for bit.
 * lowres: [num_scales][heat_channels][low_h][low_w]; peaks in/out like rtp_nms (may be NULL). */
int rtp_post_from_lowres(rtp_engine* e, const float* lowres_host, float* peaks_host, float* joints_host, int* num_people);

/* render() of rtpose.cpp:270-299 on caller data (parity tap of the renderer, and the --part_to_show views of --write_frames):
 * display_bgr / out_bgr = u8 BGR HWC at cfg.disp_w x cfg.disp_h (what the producer's canvas holds, rtpose.cpp:349, and what the
 * post-processing thread makes of the rendered canvas, :1287-1296); joints [num_people][num_parts][3] in display coordinates.
 * part_to_show = FLAGS_part_to_show: 0 = pose overlay (render_pose_coco_parts / render_pose_29parts; googly = the GUI's
 * googly-eyes toggle, COCO only), 1..parts = one heat map, parts+1 (COCO) = all part maps, above = PAF views
 * (render_coco_aff).  resized_host (heat_channels x net_h x net_w, rtp_resize's output) is only read when part_to_show != 0.
 * RTP_EINVAL for a part_to_show outside the model's maps (the reference would read past its blob). */
int rtp_render(rtp_engine* e, const unsigned char* display_bgr, const float* joints, int num_people, int part_to_show, int googly,
               const float* resized_host, unsigned char* out_bgr);

/* NmsLayer::Forward_gpu on caller data: resized (C x H x W) -> peaks (num_parts x (max_peaks+1) x 3).
 * peaks is IN/OUT: slots the kernel does not write keep the caller's contents. */
int rtp_nms(rtp_engine* e, const float* resized_host, float* peaks_host);
/* connectLimbs / connectLimbsCOCO on caller data -> joints, *num_people. */
int rtp_connect(rtp_engine* e, const float* resized_host, const float* peaks_host, float* joints,
                int* num_people);
/* Full synchronous frame with every intermediate returned (any pointer may be NULL). */
int rtp_forward_debug(rtp_engine* e, const float* nchw_input_host, float* lowres_host,
                      float* resized_host, float* peaks_host, float* joints, int* num_people);
/* Net::blob_by_name(name)->cpu_data() for any conv-stack blob of the LAST synchronous forward
 * (rtp_forward_heatmaps / rtp_forward_debug), as N x C x H x W fp32.  shape[4] out. */
int rtp_get_blob(rtp_engine* e, const char* name, float* out_host, size_t out_capacity_floats,
                 int shape[4]);

/* ---- weights / graph (net.cpp:750-803, caffe.proto:6-22,64-95,310-330) ------------------ */
int rtp_num_conv_layers(const rtp_engine* e);
int rtp_conv_layer_info(const rtp_engine* e, int i, char* name, int name_len, int* cin, int* cout, int* k);
/* Weights in Caffe blob order: w[cout][cin][k][k], b[cout] (base_conv_layer.cpp:135-175). */
int rtp_get_conv_weights(const rtp_engine* e, int i, float* w, float* b);
int rtp_set_conv_weights(rtp_engine* e, int i, const float* w, const float* b);
/* Serialise the current weights as a binary caffe NetParameter (.caffemodel). */
int rtp_save_caffemodel(const rtp_engine* e, const char* path);
/* Emit the engine's layer graph as deploy-prototxt text. */
int rtp_save_prototxt(const rtp_engine* e, const char* path);

/* Load-time precision calibration (where trained weights arrive: CopyTrainedLayersFrom, net.cpp:750-803).  The default
 * split set of RTP_PREC_MIXED was chosen on synthetic weights; this measures it on the LOADED weights: nframes sample frames
 * (frames_host: nframes x num_scales x 3 x net_h x net_w floats as rtp_submit takes them, or NULL = seeded synthetic frames)
 * run through RTP_PREC_F16X3 (every layer split: 1e-5-class reference) and through the current mixed plan;
 * err = max |mixed - f16x3| / max |f16x3| over the final maps.  While err > target (<= 0: 0.7e-3) the layer group whose
 * promotion lowers the error most joins the split set and the engine is re-planned (arena, packed weights, graphs); if every
 * group is in and the target is still missed, groups switch their corrections from the fp8 chunk to fp16 passes (":x": e4m3
 * operands saturate beyond +-112 in the activations), best first; as a last resort the engine switches to RTP_PREC_F16X3.  rules_out (may be NULL) receives the
 * final rule list ("@f16x3" after the fallback), err_before / err_after the measured errors.  Idle engine only; seconds. */
int rtp_calibrate_precision(rtp_engine* e, const float* frames_host, int nframes, float target, char* rules_out, size_t rules_len,
                            float* err_before, float* err_after);
const char* rtp_calibration_report(const rtp_engine* e); /* what the last calibration tried and chose, as text */
/* The rule list the engine runs (after calibration: the widened one) and its precision mode.  RTP_ERANGE (and an empty buf) if the list does
 * not fit buflen bytes incl. the NUL: a truncated list would describe another plan. */
int rtp_get_split_layers(const rtp_engine* e, char* buf, size_t buflen, int* precision);

/* Caller-owned device buffers on the engine's device (replaces blobs()[0]->mutable_gpu_data() as a caller-filled H2D target,
 * rtpose.cpp:1131): what rtp_submit_device reads.  Freed by rtp_device_free or with the engine. */
int rtp_device_alloc(rtp_engine* e, size_t bytes, void** dptr);
int rtp_device_free(rtp_engine* e, void* dptr);
int rtp_device_upload(rtp_engine* e, void* dst_device, const void* src_host, size_t bytes);
int rtp_device_synchronize(rtp_engine* e);

/* The CPUs local to a device ("0-31,128-159": sysfs local_cpulist of its PCI function) for the worker thread that feeds it; returns
 * the string length, 0 when the platform does not expose it. */
int rtp_device_local_cpus(int device_id, char* buf, size_t buflen);

/* One-time weight distribution between replicas (optional; the reference reads the .caffemodel once per GPU thread,
 * rtpose.cpp:183-184).  Blob = the PACKED weight arena + the Caffe-layout floats of an idle engine; the receiver must have
 * the same plan (model, resolution, batch, precision, split set).  rtp_copy_weights_from: device to device (hipMemcpyPeer,
 * xGMI) between two engines of one process. */
long rtp_weight_blob_bytes(const rtp_engine* e);
int rtp_weight_blob_export(rtp_engine* e, void* host, size_t capacity);
int rtp_weight_blob_import(rtp_engine* e, const void* host, size_t bytes);
int rtp_copy_weights_from(rtp_engine* dst, rtp_engine* src);

/* ---- host-side pieces of the path (no GPU needed) ---------------------------------------- */
/* ModelDescriptor tables (modelDescriptorFactory.cpp:25-26,52-53). limb_seq/map_idx: 2*num_limbs ints. */
int rtp_model_tables(int model, int* num_parts, int* num_limbs, int* limb_seq, int* map_idx);
/* Default thresholds chosen by warmup() (rtpose.cpp:212-226). */
int rtp_default_thresholds(int model, float* nms_threshold, float* connect_inter_threshold,
                           int* connect_inter_min_above_threshold, int* connect_min_subset_cnt,
                           float* connect_min_subset_score);
/* process_and_pad_image (rtpose.cpp:239-269): uint8 BGR HWC -> float planar, centre zero-pad. */
int rtp_process_and_pad_image(float* target, const unsigned char* bgr, int ow, int oh, int tw, int th,
                              int normalize);
/* JSON body exactly as displayFrame writes it (rtpose.cpp:1394-1415). Returns bytes written
 * (excluding the NUL) or RTP_ERANGE if buf is too small. frame_scale = Frame::scale. */
long rtp_format_json(char* buf, size_t buflen, const float* joints, int num_people, int num_parts,
                     float frame_scale);
/* Producer-side pre-processing (row a1; rtpose.cpp:322-368, 474-518).  The OpenCV primitives are
 * restated from OpenCV's published algorithms (OpenCV is absent here and unpinned by the
 * reference): rtp_warp_display = warpAffine(M = diag(fit scale), INTER_CUBIC, BORDER_CONSTANT 0),
 * rtp_resize_area = cv::resize(INTER_AREA); rtp_preprocess_frame = the whole producer step ->
 * net input (num_scales x 3 x net_h x net_w) + Frame::scale. */
double rtp_display_fit_scale(int ow, int oh, int disp_w, int disp_h);
int rtp_resize_area(const unsigned char* bgr, int sw, int sh, unsigned char* out, int dw, int dh);
int rtp_warp_display(const unsigned char* bgr, int sw, int sh, unsigned char* out, int disp_w, int disp_h,
                     double* scale_out);
int rtp_preprocess_frame(const unsigned char* bgr, int w, int h, int disp_w, int disp_h, int net_w, int net_h,
                         int num_scales, double start_scale, double scale_gap, float* net_input,
                         unsigned char* display_bgr, float* frame_scale);
/* cv::imread(path, IMREAD_COLOR) (rtpose.cpp:323) without OpenCV: baseline and progressive JPEG
 * (libjpeg's default decode arithmetic, bit-exact), PNG (zlib), binary PPM (P6), 24-bit BMP -> BGR HWC.  out_bgr may be
 * NULL to query the size.  rtp_decode_image: the same for an encoded PNG/JPEG byte string.
 * rtp_synth_frame: frame `index` of the procedural test video. */
int rtp_load_image(const char* path, unsigned char* out_bgr, size_t capacity, int* w, int* h);
int rtp_decode_image(const unsigned char* bytes, size_t n, unsigned char* out_bgr, size_t capacity, int* w, int* h);
const char* rtp_codec_last_error(void);
/* cv::imwrite(name, img, {CV_IMWRITE_JPEG_QUALITY, quality}) (rtpose.cpp:1367-1381): libjpeg's default
 * baseline 4:2:0 encoder restated, byte-identical files.  Returns the size in bytes (out may be NULL
 * to query it) or a negative RTP_E* code. */
long rtp_encode_jpeg(const unsigned char* bgr, int w, int h, int quality, unsigned char* out, size_t capacity);
/* cv::VideoCapture(path) (rtpose.cpp:402-411, 431) for the container-less formats decodable here:
 * Y4M (YUV4MPEG2, 8 bit) and raw MJPEG streams.  nframes may be NULL; rtp_video_read returns
 * RTP_EAGAIN at the end of the stream. */
typedef struct rtp_video rtp_video;
int rtp_video_open(const char* path, rtp_video** v, int* w, int* h, int* nframes);
int rtp_video_read(rtp_video* v, unsigned char* out_bgr, size_t capacity);
void rtp_video_close(rtp_video* v);
int rtp_synth_frame(unsigned char* out_bgr, int w, int h, int index, uint64_t seed);

/* Parse a deploy prototxt and report the graph it describes (for tests / tools). */
int rtp_prototxt_summary(const char* path, int* num_layers, int* num_conv, int* num_parts,
                         int* max_peaks, float* nms_threshold, int* heat_channels);

/* ---- diagnostics ------------------------------------------------------------------------- */
const char* rtp_last_error(const rtp_engine* e); /* never NULL; also valid for e == NULL (create errors) */
const char* rtp_version(void);
/* Per-stage device time of the last collected frame, ms: [0]=conv stack [1]=resize [2]=nms
 * [3]=connect [4]=total (the reference's "CNN time / Connect time" VLOGs, rtpose.cpp:1147,1168). */
int rtp_last_stage_ms(const rtp_engine* e, float ms[5]);
/* Diagnostics: each plan step alone on the chip (back-to-back launches at the full batch).
 * Returns the number of steps written (same order as rtp_plan_summary's "step" lines). */
int rtp_profile_steps(rtp_engine* e, int iters, float* ms_per_step, double* gflop_per_step, int cap);

/* Time `iters` launches of the dominant conv kernel (Mconv 7x7 128->128 pair of stage 2) with
 * HIP events on the engine's stream; returns avg ms per launch and the algorithmic FLOPs of one
 * launch.  Used by bench.py for the roofline line. */
int rtp_bench_dominant_conv(rtp_engine* e, int iters, float* avg_ms, double* flops_per_launch);
/* In-situ timing of the dominant kernel class (every paired 7x7 128->128 launch of every frame):
 * enable = 1/0 switches it on/off (resetting the totals on a change), 2 = on AND reset, enable < 0 only
 * reads; the mode can only change on an idle engine.  While on, batches are launched eagerly (no graph
 * replay) and each such launch of a FULL batch sits between a HIP event pair recorded on the stream it
 * runs on (events without the system-scope fence: time stamps only); the totals are the pairs' elapsed times.  Submit one batch at a time (collect it before the
 * next) and a pair brackets its launch alone on the chip — what a profiler's kernel trace reports.
 * Call with an idle engine to harvest.  (Round 2 compared wall-clock stamps of workgroups on different
 * XCDs; their clocks are not synchronised on every box.) */
int rtp_kernel_timing(rtp_engine* e, int enable, double* total_ms, long* launches, double* flops_per_launch);
/* The same totals split by the MFMA pass-times of the launch: 1 = plain fp16 layer, 2 = fp16 + one fp8 error-compensation
 * chunk per channel group (RTP_PREC_MIXED on 3x3 / 7x7 layers), 3 = three fp16 passes (RTP_PREC_F16X3); index 0 is unused. */
int rtp_kernel_timing_by_passes(const rtp_engine* e, double ms[4], long launches[4]);
/* enable = 3 in rtp_kernel_timing: on AND reset with an event pair around EVERY step of the plan (all convolution launches, the
 * branch-tail launches, conv1_1), not only the dominant class — its totals keep their meaning.  rtp_kernel_timing_steps then returns,
 * per plan step i (the order of rtp_plan_summary's "step" lines), the summed milliseconds and the number of launches timed; return
 * value = number of steps (fill at most `cap`).  bench.py groups them into kernel classes (`roofline.classes`). */
int rtp_kernel_timing_steps(const rtp_engine* e, double* ms, long* launches, int cap);
/* An UNPROFILED account of when the engine had work on the GPU (rocprofv3's timeline carries the tracer's own overhead).  enable = 1:
 * on + reset (idle engine only), 0: off, -1: read.  While on, rtp_collect reads — from events the per-frame path records anyway —
 * {kind, start_ms, end_ms} triples since the switch-on: kind 0 = one batch on its conv stream, from the staging of its first input to
 * the end of its conv stack; kind 1 = one frame's post-processing chain incl. the D2H of its joints.  Returns the number of triples
 * (at most `cap` are copied to `spans`, which may be NULL). */
int rtp_busy_probe(rtp_engine* e, int enable, float* spans, int cap);
/* What the probes above could not record since the engine was created: out[0] = launches rtp_kernel_timing wanted to time while its table
 * of event pairs (4096 between two harvests) was full, out[1] = frames whose rtp_busy_probe spans fell beyond its 65536-triple cap,
 * out[2] = frames the busy probe skipped because their batch ran as one whole-batch graph replay (no per-frame events there).  A
 * measurement that reads non-zero counts here is truncated and must say so (bench.py refuses to print a roofline from it). */
int rtp_probe_dropped(const rtp_engine* e, long out[3]);
/* Kernel RESIDENCY without a profiler: while on, thread 0 of every workgroup of every kernel of the per-frame path (device pre-processing,
 * conv stack, ImResize+Nms, connect) folds the chip's 100 MHz wall clock into a per-launch slot — first workgroup start, last workgroup
 * end.  enable = 1: on + reset (idle engine only; the batch graphs are re-captured with the slots), 0: off, -1: harvest + read.  spans:
 * {slot, start_us, end_us} triples; slot < 64 = plan step (rtp_plan_summary order), 64 + 8 j + k = frame j's strip / write / pairs / match /
 * assemble kernel, 200 + 2 j + k = frame j's warp / area-pad kernel.  1 - union(spans) / wall = the share of the time with NO kernel on the
 * chip (bench.py `idle_frac_kernel_stamps`).  No reference counterpart (measurement only).  Returns the number of triples. */
int rtp_stamp_probe(rtp_engine* e, int enable, float* spans, int cap);
/* Host float -> OCP e4m3 (round to nearest even, clamped to +-448): how the fp8 weight copies of split layers are made. */
int rtp_debug_f32_to_e4m3(const float* in, unsigned char* out, int n);

/* Survivors of the PAF test (temp.size(), rtpose.cpp:950) and accepted connections
 * (connection_k.size(), :980) per limb for the last synchronous frame; arrays of num_limbs ints. */
int rtp_debug_connect_stats(rtp_engine* e, int* cand_count, int* conn_count);

/* Host-only weight utilities.  rtp_synth_weights: the deterministic generator behind
 * synthetic_seed (w[cout][cin][k][k], b[cout]).  rtp_write_synthetic_caffemodel: the same weights
 * for every conv of the built-in linevec net as a binary NetParameter.  rtp_caffemodel_layer:
 * read a .caffemodel (new `layer` or V1 `layers`) — index < 0 returns the layer count. */
int rtp_synth_weights(uint64_t seed, const char* layer_name, int cout, int cin, int k, float* w, float* b);
int rtp_write_synthetic_caffemodel(int model, uint64_t seed, const char* path);
/* The built-in linevec graph (what proto_path == NULL uses) as deploy-prototxt text. */
int rtp_write_builtin_prototxt(int model, const char* path);
int rtp_caffemodel_layer(const char* path, int index, char* name, int name_len, int* num_blobs,
                         long* count0, long* count1, float* head0);

/* Build the execution plan for cfg without touching a device and describe it as text (tile
 * configuration per layer, L1/L2 pairing, arena sizes).  Returns bytes written or a negative code. */
long rtp_plan_summary(const rtp_config* cfg, char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* RTPOSE_MI355X_H_ */
