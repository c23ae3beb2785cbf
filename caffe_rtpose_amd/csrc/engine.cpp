// engine.cpp — static execution plan + frame contexts behind the C-ABI of include/rtpose_mi355x.h.
//
// Replaces the Caffe net runtime on the rtpose path (net.cpp:49 Net::Init, :544-556
// ForwardFromTo, syncedmem.cpp:25-77) with a plan that is compiled ONCE per
// (graph, resolution, num_scales, precision):
//   * every blob gets a halo'd NHWC tensor in a per-context arena (kernels.h), zeroed once;
//   * ReLU is folded into the producing convolution, Concat is eliminated by letting producers
//     write straight into channel slices of the consumer's tensor (multi-destination epilogue),
//     Split is a pointer share;
//   * the independent L1/L2 branch convolutions of a stage are paired into one launch;
//   * weights are re-laid out per layer for the kernel's [tap][chunk][cout][k] staging order;
//   * frames_in_flight contexts (stream + arena) let consecutive frames overlap, which is what
//     fills 256 CUs when one frame's layers are too small to.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtpose_mi355x.h"
#include "kernels.h"
#include "netdef.h"

using namespace rtp;

void rtp_internal_cubic_tab2d(short* tab);
double rtp_internal_warp_inverse_scale(double s);
bool rtp_internal_area_fast(int sw, int sh, int dw, int dh, int* ix, int* iy);
int rtp_internal_area_table(int ssize, int dsize, std::vector<int>* start, std::vector<int>* si, std::vector<float>* alpha);
void rtp_internal_linear_area_table(int ssize, int dsize, std::vector<int>* tab);
extern "C" double rtp_display_fit_scale(int ow, int oh, int disp_w, int disp_h);
extern "C" int rtp_preprocess_frame(const unsigned char* bgr, int w, int h, int disp_w, int disp_h, int net_w, int net_h, int num_scales,
                                    double start_scale, double scale_gap, float* net_input, unsigned char* display_bgr, float* frame_scale);

namespace {

thread_local std::string g_create_error = "";

// HIP refuses every synchronous legacy-stream operation (hipMemset, hipMemcpy, hipDeviceSynchronize, ...) of ANY thread while ANY
// stream of the process is capturing a graph ("operation would make the legacy stream depend on a capturing blocking stream"), whatever
// the capture mode.  One engine per worker thread (rtpose.cpp:1463-1472) means engines are created, captured and torn down concurrently:
// every capture and every such synchronous operation of this library happens under this one process-wide lock.  The per-frame path
// (async copies, launches, event waits on the engine's own streams) never takes it.  Found by rtpose.bin --num_gpu 4 --devices 0,0,0,0.
std::recursive_mutex g_sync_mutex;
#define SYNC_GUARD std::lock_guard<std::recursive_mutex> sync_guard_(g_sync_mutex)

struct Tensor {
  std::string name;
  int C = 0, Cp = 0, level = 0;
  size_t offset = 0;        // byte offset of padded pixel 0 of image 0 inside a context arena
  std::vector<int> chmap;   // reference channel -> internal channel
  // split precision: [0, Cp) hi; need_lo: [Cp, 2Cp) lo = T(v - float(T(v))) (consumers running three fp16 passes);
  // need_q: a further Cp elements = 2*Cp bytes of fp8 compensation operands (consumers running the fp8 passes, ConvDst::q_off)
  bool need_lo = false, need_q = false;
  bool written = true;      // false: the blob was fused away (a convolution pools in its epilogue and writes only the pooled tensor)
  int stride() const { return Cp * (1 + (need_lo ? 1 : 0) + (need_q ? 1 : 0)); }  // channels (elements) per pixel in memory
  int lo_off() const { return need_lo ? Cp : 0; }
  int q_off() const { return need_q ? Cp * (need_lo ? 2 : 1) : 0; }
};

struct ConvOp {
  std::string name;
  int widx = 0;             // index into engine weights
  int in_tensor = -1;
  int k = 1, k_eff = 1;     // k_eff = 1 for the im2col-packed first layer
  int cin = 0, cout = 0;
  bool relu = false, first = false;
  std::vector<std::pair<int, int>> dsts;  // (tensor, channel offset)
  bool to_lowres = false;
  int lowres_coff = 0;
  int level = 0;
  int Cin_p = 0, rowb = 128, nchunk = 1, CoutP = 0, cfg = 0;
  int impl = 0;             // 0 = register-staged kernel (conv_igemm.hip), 1 = LDS-DMA ring (conv_ring.hip)
  bool direct_first = false;  // conv1_1 straight from the NCHW input (conv_first.hip): no im2col tensor, no pack step
  // split precision (RTP_PREC_MIXED / F16X3): the K loop runs the passes [a_hi x W_hi] [a_lo x W_hi] [a_hi x W_lo]
  bool split_a = false, split_w = false;
  bool no_h8 = false;       // rule suffix ":x": the corrections of this layer run as fp16 passes even where the fp8 chunk exists (no e4m3 range limits)
  int ncp = 1;              // chunks of ONE pass (nchunk = ncp * passes)
  bool h8 = false;          // the two correction passes run as ONE fp8 chunk per channel group (MX-scaled MFMA, 2x the fp16 rate)
  int wq_exp = 0;           // h8: fp8(W * 2^wq_exp), fp8(W_lo * 2^(wq_exp + 11))
  int passes() const { return h8 ? 2 : 1 + (split_a ? 1 : 0) + (split_w ? 1 : 0); }  // in units of one fp16 pass of MFMA time
  int wrap_at() const { return h8 ? 0 : (split_w ? (split_a ? 2 * ncp : ncp) : 0); }
  int last_phys() const { return h8 ? ncp - 1 : (split_w ? ncp - 1 : (split_a ? 2 * ncp - 1 : ncp - 1)); }
  int pool = -1;            // >= 0: this convolution's only consumer is pooling layer `pool`; it pools in its epilogue (conv_ring.hip POOL)
  int fused = 0;            // 1 / 2: first / second 1x1 of a conv_pw2 step (weights packed for that kernel)
  int fused_chunks = 0;     // middle channels / 128
  size_t w_off = 0, b_off = 0, w_bytes = 0;
};

struct Step {
  int type;  // 0 pack, 1 conv, 2 pool, 3 two chained 1x1 convolutions in one launch (conv_pw2.hip), 4 input convolution from the NCHW image (conv_first.hip)
  int a = -1, b = -1;    // conv (a) [+ the other branch's conv (b)]; pool index for type 2
  int a2 = -1, b2 = -1;  // type 3: the second 1x1 of each branch
};

struct PoolOp { int in_tensor, out_tensor, C; };

// One frame's post-processing state (a batch context carries batch_frames of them)
struct Slot {
  hipStream_t stream = nullptr;  // resize -> nms -> connect -> D2H of this frame (slot 0 shares the context's stream)
  bool own_stream = false, preset_stream = false;
  float* resized = nullptr;
  float* peaks = nullptr;
  int* strip_count = nullptr;
  int* strip_list = nullptr;
  float* cand_score = nullptr;
  int* cand_ij = nullptr;
  int* cand_count = nullptr;
  int* cand_blk = nullptr;
  int* conn = nullptr;
  float* conn_score = nullptr;
  int* conn_count = nullptr;
  int* tickets = nullptr;       // connect_chain_kernel's tickets (num_limbs + 1 zeros)
  float* joints = nullptr;
  int* num_people = nullptr;
  float* host_out = nullptr;  // pinned: [1 int as float slot][joints]
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // post start, resize, nms, connect, results on host
  // deferred pre-processing (rtp_engine::prep_defer): the frame's H2D copy runs on the engine's copy stream; its warp / area kernels are
  // enqueued on the batch's conv stream once the copy has COMPLETED (polled at the next API call; forced, with a stream wait, at launch)
  hipEvent_t ev_copy = nullptr;
  bool copy_pending = false;
  int pend_w = 0, pend_h = 0;
  unsigned char* frame_dev = nullptr;   // raw u8 frame (device) for rtp_submit_frame
  unsigned char* frame_host = nullptr;  // pinned staging of the raw frame
  size_t frame_cap = 0;
  unsigned char* disp_dev = nullptr;    // display-resolution u8 image
  const unsigned char* disp_cur = nullptr;  // this frame's display image: disp_dev, or frame_dev itself when the frame already HAS the display size (fit scale 1: the
                                            // cubic warp samples at integer positions with weights (0, 1, 0, 0) = a copy; skipped, bit for bit the same image)
  bool has_disp = false;                // disp_dev holds this frame's display image (rtp_submit_frame, device path)
  unsigned char* render_dev = nullptr;  // cfg.render: display image with the pose overlay
  unsigned char* render_host = nullptr; // pinned copy of it
  float* render_tab = nullptr;
  uint64_t tag = 0;
  bool busy = false;
};

// One batch in flight: the conv stack runs once over filled*num_scales images
struct Ctx {
  hipStream_t stream = nullptr, spare_stream = nullptr;
  std::vector<hipStream_t> pad_streams;   // never used: they only take hardware-queue slots in the runtime's round-robin (alloc_ctx)
  unsigned char* arena = nullptr;
  float* input = nullptr;     // device NCHW fp32, batch_frames * num_scales images
  float* host_in = nullptr;   // pinned staging
  float* lowres = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};  // around the conv stack
  // Input staging (H2D of a frame, device pre-processing, or the copy of a resident input) runs on its own stream, so that a frame's copy
  // is under way while earlier batches compute and the conv queue never holds a barrier that waits for PCIe; ev_in = staged input complete
  hipStream_t in_stream = nullptr;
  hipEvent_t ev_in = nullptr;
  bool in_pending = false;
  std::vector<Slot> slot;
  int filled = 0;        // frames staged in the open batch
  bool launched = false;
  // RTP_EXEC_GRAPH: the whole batch (conv stack, every frame's post-processing chain, D2H) captured once
  // per (timed?, frames in the batch) and replayed; gev = {before, after} the replay on `stream`
  hipGraphExec_t gexec[17] = {};
  hipEvent_t gev[2] = {nullptr, nullptr};
  hipEvent_t ev_stage0 = nullptr;         // rtp_busy_probe: recorded in front of the batch's first input staging
  bool stage0_set = false;
  bool graph_run = false;                 // the batch in flight was ONE graph replay incl. post-processing (collect waits on gev[1])
  bool graph_conv = false;                // the conv stack was a replay, the post-processing chains were launched eagerly
  unsigned long long* stamps = nullptr;   // rtp_stamp_probe: STAMP_SLOTS x {~first workgroup start, last workgroup end} of this context's launches (kernels.h KStamp)
  bool stamps_dirty = false;
};

}  // namespace

struct rtp_engine {
  rtp_config cfg;
  std::string proto_path, weights_path;
  NetDef net;
  int model = 0, prec = 0, elem = 2;
  int num_parts = 18, max_peaks = 64, heat_channels = 57, num_limbs = 19;
  int low_w = 0, low_h = 0;
  float nms_threshold = 0.05f, inter_threshold = 0.05f, min_subset_score = 0.4f;
  int inter_min_above = 9, min_subset_cnt = 3;
  float start_scale = 1.f, scale_gap = 0.3f;
  bool weights_pending = false;   // rtp_config.defer_weights: no weights yet (zero arena, no graphs): the net must not run until an import / peer copy
  bool calib_fell_back = false;   // rtp_calibrate_precision ended in RTP_PREC_F16X3 (its last resort)
  bool broken = false;   // a re-plan failed and the previous plan could not be restored: no contexts; every entry point fails, destroy works
  int N = 1;    // images per frame (num_scales)
  int B = 1;    // frames per batch (cfg.batch_frames)
  int NI = 1;   // images per conv launch at a full batch = N * B
  int open_ctx = -1;  // context whose batch is being filled
  Geom geom[8];
  int nlevels = 0;
  std::vector<Tensor> tensors;
  std::map<std::string, int> blob_tensor;                 // blob name -> tensor (NHWC blobs)
  std::map<std::string, std::pair<int, int>> blob_dims;   // blob name -> (C, level)
  std::string lowres_blob;
  std::vector<ConvOp> convs;
  std::vector<Step> steps;
  std::vector<PoolOp> pools;
  std::vector<std::vector<float>> w_ref, b_ref;           // Caffe layout per conv
  size_t arena_bytes = 0, weights_bytes = 0;
  unsigned char* dweights = nullptr;
  int* dchmap = nullptr;  // scratch for export
  std::vector<Ctx> ctx;
  std::deque<int> fifo;
  int nstrips = 0, strip_rows = 8, max_rows = 0;
  float last_ms[5] = {0, 0, 0, 0, 0};
  std::string err;
  int dominant_step = -1;
  bool time_dominant = false;
  double dom_ms_total = 0;
  long dom_launches = 0;
  double dom_ms_pass[4] = {0, 0, 0, 0};   // the same, by MFMA passes of the launch (split-precision layers run 2-3)
  long dom_n_pass[4] = {0, 0, 0, 0};
  // rtp_kernel_timing: one HIP event pair around every dominant-class launch (recorded on the launch's own stream; batches are
  // launched eagerly while it is on).  Nothing here compares clocks of different XCDs.
  std::vector<hipEvent_t> tev;            // 2 * pairs
  std::vector<unsigned char> tev_pass;    // MFMA passes (1..3) of the launch behind pair i
  std::vector<short> tev_step;            // plan step of the launch behind pair i
  // rtp_busy_probe: when on, every batch's stream-busy spans (first input staging .. end of its conv stack on the batch's stream; start ..
  // end of each frame's post-processing chain incl. the D2H of the joints on the frame's stream) are read back at collect time as
  // milliseconds since `busy_base` — an UNPROFILED account of when the engine had work on the GPU (bench.py: gpu_busy)
  bool busy_probe = false;
  hipEvent_t busy_base = nullptr;
  std::vector<float> busy_spans;          // [n][3]: kind (0 conv stream, 1 post chain), start ms, end ms
  // rtp_stamp_probe: device-side residency stamps of every kernel of the per-frame path (no profiler, no events)
  bool stamp_probe = false;
  unsigned long long stamp_base = 0;      // wall-clock ticks (100 MHz) of the first harvested start
  std::vector<float> stamp_spans;         // [n][3]: slot id (see stamp_slot), start us, end us relative to stamp_base
  bool time_all = false;                  // rtp_kernel_timing(3): an event pair around EVERY step of a full batch, not only the dominant class
  std::vector<double> step_ms;            // per plan step: event-timed milliseconds / launches of the timing pass (rtp_kernel_timing_steps)
  std::vector<long> step_n;
  int tev_next = 0;
  long probe_dropped[3] = {0, 0, 0};      // rtp_probe_dropped: timing pairs not recorded (table full), busy spans dropped (cap), graph-replayed batches the busy probe skipped
  static const int TEV_PAIRS = 4096;
  // device-side pre-processing (row a1)
  const short* warp_tab_dev = nullptr;  // inside prep_tables
  std::vector<AreaScale> area_scales;   // device pointers inside prep_tables
  unsigned char* prep_tables = nullptr;
  bool gpu_prep_ok = false;
  bool use_graph = true;
  bool split_fp8 = true;    // RTP_SPLIT_FP8=0: split layers run three fp16 passes everywhere
  bool graph_post = false;  // RTP_GRAPH_POST=1: also capture the per-frame post-processing chains + D2H into the batch graph
  // 0 (default) = a batch's inputs are staged on its conv stream; 1 = on a stream of their own (created after all others), 2 = ... of high
  // priority.  Measured (profiles/r04_input_staging.txt, host u8 frames, 7 in flight): 0: 1045 frames/s, 1: 880-890, 2: 880-990 — the
  // runtime multiplexes streams onto 4 hardware queues, and a staging stream's barrier (kernel behind a PCIe copy) then blocks whichever
  // conv / post-processing stream shares its queue.  Experiments build only (RTP_IN_STREAM).
  // 1 = the connect chain (pairs -> match -> assemble) as ONE launch (connect_chain_kernel: tickets; experiments build, RTP_CHAIN_CONNECT=1).
  // Bit-identical (GPU test) and no faster: 67.5 vs 67.7 us for one planted person, 97.5 vs 96.5 us on noise maps, 1033 vs 1039 frames/s —
  // on a chip of 8 XCDs the device-scope release / acquire between the phases is an L2 write-back / invalidate, i.e. what a kernel boundary
  // costs (with a fence per THREAD instead of per workgroup the first version took 24 us more).  profiles/r04_experiments.txt.
  int chain_connect = 0;
  int in_stream_mode = 0;
  // 1 = deferred pre-processing (experiments build, RTP_PREP_DEFER): a frame's copy goes to a copy-only stream, its kernels are enqueued on
  // the conv stream once the copy has COMPLETED (polled at the next API call), a full batch whose last frame was just copied is launched at
  // the next call — so that no kernel of a compute queue sits behind a barrier that waits for PCIe.  Bit-identical (GPU test), and not
  // adopted: what it gains depends on WHEN the caller makes its next call.  tools/bench_input.py: COCO +0.7 %, MPI (batches of 5) +5 %;
  // inside bench.py's loop (a collect right behind the submit that completes a batch): COCO -5 %, MPI +-0; waiting for the copy on the
  // host instead: -23 % (hipEventSynchronize ~0.3 ms per call).  profiles/r04_input_staging.txt.
  int prep_defer = 0;
  hipStream_t copy_stream = nullptr;   // H2D copies only (created after every other stream)
  std::deque<int> pending_launch;      // full batches whose last frame's copy was still in flight when it was committed (launched by pump())
  int mode = 0;  // rtp_config.precision (RTP_PREC_*); `prec` below selects the kernels' element type (0 fp16, 1 fp32)
  std::string split_rules;
  int nctx_full = 1;            // batch contexts of the configured pipeline (a calibration trial runs with one)
  std::string calib_report;     // what rtp_calibrate_precision last did (rtp_calibration_report)
  std::vector<void*> user_bufs; // rtp_device_alloc allocations still alive
};

namespace {

// Flags of the engine's events.  sync = the event is waited for (by the host or by another stream) after work whose results the waiter reads;
// the others only take time stamps between launches.  Experiments build: RTP_EV_NOFENCE=1 creates the time-stamp events of the per-frame path
// without the system-scope release fence a default event record carries, 2 = every event (measured: no change in the pipelined frame rate;
// the event pairs of rtp_kernel_timing are always created without it).
unsigned event_flags(bool sync) {
  static const char* nf = RTP_EXP_ENV("RTP_EV_NOFENCE");
  const int lvl = nf ? atoi(nf) : 0;
  return (lvl >= 2 || (lvl == 1 && !sync)) ? hipEventDisableSystemFence : hipEventDefault;
}

int fail(rtp_engine* e, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
  else g_create_error = buf;
  return code;
}

#define HIPCHK(e, call)                                                                             \
  do {                                                                                              \
    hipError_t _s = (call);                                                                         \
    if (_s != hipSuccess)                                                                           \
      return fail((e), RTP_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_s), __FILE__, __LINE__); \
  } while (0)

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }
inline size_t round_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

const int GUARD_PIX = 192;  // pixels of slack before/after each tensor (strip over-read of the last tile)

// ---- split-precision policy ------------------------------------------------------------------
// Default RTP_PREC_MIXED set, from tools/sim_precision.py (error of the final maps vs an fp32 run, per layer and per
// rounded operand): the fp16 rounding of weights and activations contributes about equally in every layer, and the
// final-map error is dominated by the trunk from conv2 on, the last two refinement stages and (cheaply fixed) all 1x1
// layers; the first refinement stages are attenuated by each later stage's re-injection of conv4_4_CPM.
// Measured (tests/test_precision.py; tools/sim_precision.py reproduces the rms to 3 digits): final maps normalised to max 1,
//   fp16 everywhere                          rms 3.5e-4   max 2.0-2.6e-3
//   conv2-4, stages 5-6, 1x1 (2.08x MFMA)    rms 1.36e-4  max 0.8-1.03e-3   <- no margin on the +-1e-3 tolerance
//   + stage 4 (this default, 2.35x MFMA)     rms 1.03e-4  max <= 0.8e-3
//   every layer (RTP_PREC_F16X3, 3x MFMA)    rms 1.6e-6   max 1e-5
const char* kDefaultSplit = "conv2_,conv3_,conv4_,*_stage4_,*_stage5_,*_stage6_,@1x1";
void layer_split(const rtp_engine* e, const ConvOp& c, bool* w, bool* a, bool* x = nullptr) {
  *w = *a = false;
  if (x) *x = false;
  if (e->prec != 0) return;
  if (e->mode == RTP_PREC_F16X3) { *w = *a = true; return; }
  if (e->mode != RTP_PREC_MIXED) return;
  const std::string& rules = e->split_rules;
  size_t pos = 0;
  while (pos <= rules.size()) {
    size_t c2 = rules.find(',', pos);
    std::string tok = rules.substr(pos, c2 == std::string::npos ? std::string::npos : c2 - pos);
    pos = c2 == std::string::npos ? rules.size() + 1 : c2 + 1;
    if (tok.empty()) continue;
    bool tw = true, ta = true, tx = false;
    if (tok.size() > 2 && tok[tok.size() - 2] == ':') {
      const char k = tok.back();
      tok.resize(tok.size() - 2);
      if (k == 'w') ta = false;
      else if (k == 'a') tw = false;
      else if (k == 'x') tx = true;   // both operands, corrections as two more fp16 passes (what RTP_PREC_F16X3 runs everywhere)
    }
    bool hit;
    if (tok == "@all") hit = true;
    else if (tok == "@1x1") hit = c.k == 1;
    else if (tok[0] == '*') hit = c.name.find(tok.substr(1)) != std::string::npos;
    else hit = c.name.compare(0, tok.size(), tok) == 0;
    if (hit) { *w = *w || tw; *a = *a || ta; if (x) *x = *x || tx; }
  }
}

// ---- plan ---------------------------------------------------------------------------------
int build_plan(rtp_engine* e) {
  const NetDef& net = e->net;
  const int CALIGN = 128 / e->elem;
  e->blob_dims.clear();
  if (net.inputs.empty()) return fail(e, RTP_EINVAL, "prototxt declares no input blob");
  const std::string in_name = net.inputs[0];
  e->blob_dims[in_name] = {3, 0};
  struct ConcatInfo { std::vector<std::string> inputs; };
  std::map<std::string, ConcatInfo> concats;
  std::map<std::string, int> producer_conv;  // blob -> conv index
  std::vector<int> level_halo(8, 0);
  int max_level = 0;
  bool have_resize = false, have_nms = false;
  struct PoolTmp { std::string in, out; };
  std::vector<std::pair<int, int>> order;  // (kind 1 conv / 2 pool, index)
  std::vector<PoolTmp> pools;

  for (size_t li = 0; li < net.layers.size(); ++li) {
    const LayerDef& L = net.layers[li];
    if (L.type == "Convolution") {
      if (L.bottoms.size() != 1 || L.tops.size() != 1) return fail(e, RTP_EINVAL, "layer %s: expected 1 bottom/1 top", L.name.c_str());
      auto it = e->blob_dims.find(L.bottoms[0]);
      if (it == e->blob_dims.end()) return fail(e, RTP_EINVAL, "layer %s: unknown bottom %s", L.name.c_str(), L.bottoms[0].c_str());
      if (L.stride != 1 || !(L.kernel == 1 || L.kernel == 3 || L.kernel == 7) || L.pad != (L.kernel - 1) / 2 || !L.bias_term)
        return fail(e, RTP_EINVAL, "layer %s: only stride-1 'same' convolutions with k in {1,3,7} and a bias are on the linevec path", L.name.c_str());
      ConvOp c;
      c.name = L.name; c.k = L.kernel; c.k_eff = L.kernel; c.cin = it->second.first; c.cout = L.num_output;
      c.level = it->second.second;
      c.first = (L.bottoms[0] == in_name);
      if (c.first && !(c.cin == 3 && c.k == 3)) return fail(e, RTP_EINVAL, "layer %s: the input convolution must be 3x3 on 3 channels", L.name.c_str());
      if (c.first) c.k_eff = 1;
      c.widx = (int)e->convs.size();
      level_halo[c.level] = std::max(level_halo[c.level], c.k_eff / 2);
      e->blob_dims[L.tops[0]] = {c.cout, c.level};
      producer_conv[L.tops[0]] = (int)e->convs.size();
      order.push_back({1, (int)e->convs.size()});
      e->convs.push_back(c);
    } else if (L.type == "ReLU") {
      if (L.bottoms.size() != 1 || L.tops.size() != 1 || L.bottoms[0] != L.tops[0] || !producer_conv.count(L.bottoms[0]))
        return fail(e, RTP_EINVAL, "layer %s: ReLU must be in-place on a convolution output", L.name.c_str());
      if (L.negative_slope != 0.f) return fail(e, RTP_EINVAL, "layer %s: negative_slope != 0 unsupported", L.name.c_str());
      e->convs[producer_conv[L.bottoms[0]]].relu = true;
    } else if (L.type == "Pooling") {
      auto it = e->blob_dims.find(L.bottoms.empty() ? "" : L.bottoms[0]);
      if (it == e->blob_dims.end()) return fail(e, RTP_EINVAL, "layer %s: unknown bottom", L.name.c_str());
      if (L.pool_method != "MAX" || L.pool_kernel != 2 || L.pool_stride != 2 || L.pool_pad != 0)
        return fail(e, RTP_EINVAL, "layer %s: only MAX 2x2 stride 2 pooling is on the linevec path", L.name.c_str());
      e->blob_dims[L.tops[0]] = {it->second.first, it->second.second + 1};
      max_level = std::max(max_level, it->second.second + 1);
      pools.push_back({L.bottoms[0], L.tops[0]});
      order.push_back({2, (int)pools.size() - 1});
    } else if (L.type == "Concat") {
      if (L.axis != 1) return fail(e, RTP_EINVAL, "layer %s: only channel concat", L.name.c_str());
      int C = 0, lvl = -1;
      for (auto& b : L.bottoms) {
        auto it = e->blob_dims.find(b);
        if (it == e->blob_dims.end()) return fail(e, RTP_EINVAL, "layer %s: unknown bottom %s", L.name.c_str(), b.c_str());
        if (!producer_conv.count(b)) return fail(e, RTP_EINVAL, "layer %s: concat inputs must be convolution outputs", L.name.c_str());
        if (lvl >= 0 && lvl != it->second.second) return fail(e, RTP_EINVAL, "layer %s: concat inputs at different resolutions", L.name.c_str());
        lvl = it->second.second;
        C += it->second.first;
      }
      e->blob_dims[L.tops[0]] = {C, lvl};
      concats[L.tops[0]] = ConcatInfo{L.bottoms};
    } else if (L.type == "ImResize") {
      if (!e->blob_dims.count(L.bottoms[0])) return fail(e, RTP_EINVAL, "resize: unknown bottom");
      if (L.factor != 8.f) return fail(e, RTP_EINVAL, "resize: only factor 8 (the net's total stride) is supported");
      e->lowres_blob = L.bottoms[0];
      have_resize = true;
    } else if (L.type == "Nms") {
      e->num_parts = L.num_parts;
      e->max_peaks = L.max_peaks;
      have_nms = true;
    } else if (L.type == "Split") {
      // pointer share: alias tops to the bottom
      for (auto& t : L.tops) e->blob_dims[t] = e->blob_dims[L.bottoms[0]];
      return fail(e, RTP_EINVAL, "layer %s: explicit Split layers are not expected in a deploy prototxt", L.name.c_str());
    } else {
      return fail(e, RTP_EINVAL, "layer %s: type %s is not on the linevec hot path", L.name.c_str(), L.type.c_str());
    }
  }
  if (!have_resize || !have_nms) return fail(e, RTP_EINVAL, "graph must end in ImResize + Nms layers");
  if (e->num_parts == 18) e->model = RTP_MODEL_COCO_18;
  else if (e->num_parts == 15) e->model = RTP_MODEL_MPI_15;
  else return fail(e, RTP_EINVAL, "Unknown number of parts (%d)! Couldn't set model", e->num_parts);  // rtpose.cpp:227
  e->num_limbs = e->model == 0 ? 19 : 14;
  if (e->max_peaks < 1 || e->max_peaks > 127) return fail(e, RTP_EINVAL, "max_peaks %d out of range [1,127]", e->max_peaks);
  e->heat_channels = e->blob_dims[e->lowres_blob].first;
  if (e->blob_dims[e->lowres_blob].second != 3) return fail(e, RTP_EINVAL, "resize input must be at 1/8 resolution");
  {
    const int need = (e->model == 0 ? 57 : 44);
    if (e->heat_channels != need) return fail(e, RTP_EINVAL, "resize input has %d channels, model needs %d", e->heat_channels, need);
  }

  // geometry
  e->nlevels = max_level + 1;
  if ((e->cfg.net_w % 16) || (e->cfg.net_h % 16) || e->cfg.net_w < 16 || e->cfg.net_h < 16)
    return fail(e, RTP_EINVAL, "net_resolution %dx%d must be positive multiples of 16", e->cfg.net_w, e->cfg.net_h);
  for (int l = 0; l < e->nlevels; ++l) {
    Geom g;
    g.N = e->NI; g.H = e->cfg.net_h >> l; g.W = e->cfg.net_w >> l; g.halo = level_halo[l];
    // ONE zero gap of `halo` pixels between consecutive rows serves as the right halo of row y and the left halo of row y+1
    // (flat addressing: pixel p's tap (r,s) is p + (r-pad)*Wp + (s-pad), so x+pad past the row end lands in the gap and x-pad
    // before the row start lands in the previous row's gap).  Wp = W + halo instead of W + 2*halo: 3.4 % fewer GEMM rows at
    // 1/8 resolution (85 instead of 88 per row) and 31 instead of 32 M-tiles of 128 per 46x82 image — a launch of the paired
    // 7x7 layers at batch_frames = 2 is 248 workgroups, not 256: it no longer needs EVERY CU at once.
    // The last pixel's far corner tap reads 2 pixels past Hp*Wp: the next image's top halo / the tensor's zero guard.
    static const char* sh = RTP_EXP_ENV("RTP_HALO_SHARED");  // experiments: 0 = a halo on both sides of every row
    const bool shared = !(sh && sh[0] == '0');
    g.Hp = g.H + 2 * g.halo; g.Wp = g.W + (shared ? 1 : 2) * g.halo; g.img_pix = (long)g.Hp * g.Wp;
    e->geom[l] = g;
  }
  e->low_w = e->cfg.net_w / 8;
  e->low_h = e->cfg.net_h / 8;

  // tensors
  auto new_tensor = [&](const std::string& name, int C, int level) {
    Tensor t;
    t.name = name; t.C = C; t.level = level;
    t.Cp = round_up(C, CALIGN);
    t.chmap.resize(C);
    for (int i = 0; i < C; ++i) t.chmap[i] = i;
    e->tensors.push_back(t);
    return (int)e->tensors.size() - 1;
  };
  // packed im2col input
  int packed_tensor = -1;
  {
    Tensor t;
    t.name = "__im2col_input"; t.C = 27; t.level = 0; t.Cp = 32;
    t.chmap.resize(27);
    for (int i = 0; i < 27; ++i) t.chmap[i] = i;
    e->tensors.push_back(t);
    packed_tensor = 0;
  }
  for (auto& c : e->convs) e->blob_tensor[c.name] = new_tensor(c.name, c.cout, c.level);
  for (auto& p : pools) e->blob_tensor[p.out] = new_tensor(p.out, e->blob_dims[p.out].first, e->blob_dims[p.out].second);
  // concat tensors (those read by convolutions); aligned inputs first
  std::map<std::string, std::vector<std::pair<std::string, int>>> concat_slices;  // concat -> (input, internal offset)
  for (auto& kv : concats) {
    if (kv.first == e->lowres_blob) continue;
    const auto& ins = kv.second.inputs;
    std::vector<int> ord(ins.size());
    for (size_t i = 0; i < ins.size(); ++i) ord[i] = (int)i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
      const bool ua = (e->blob_dims[ins[a]].first % 8) != 0, ub = (e->blob_dims[ins[b]].first % 8) != 0;
      return (int)ua < (int)ub;
    });
    std::vector<int> internal_off(ins.size());
    int off = 0;
    for (int i : ord) { internal_off[i] = off; off += e->blob_dims[ins[i]].first; }
    {  // every slice on an 8-channel boundary where the pad channels of the tensor pay for it (concat_stageK: conv4_4_CPM at 0, L1 at 128, L2 at
       // 168 instead of 166, 192 channels either way): the producers' epilogues then write the slice with 16-byte stores instead of one 2-byte
       // store + two fp8 byte stores per channel (conv_common.h conv_store_dst) — the branch tails' epilogue was 4.1 us of an 11 us workgroup
       // for that reason.  The skipped channels are pad channels like the tail's: zero activations, zero weights (chmap never points at them).
      std::vector<int> aligned(ins.size());
      int a = 0;
      for (int i : ord) { a = round_up(a, 8); aligned[i] = a; a += e->blob_dims[ins[i]].first; }
      if (round_up(a, CALIGN) == round_up(off, CALIGN)) internal_off = aligned;
    }
    const int tid = new_tensor(kv.first, e->blob_dims[kv.first].first, e->blob_dims[kv.first].second);
    e->blob_tensor[kv.first] = tid;
    int refc = 0;
    for (size_t i = 0; i < ins.size(); ++i) {
      const int C = e->blob_dims[ins[i]].first;
      for (int c = 0; c < C; ++c) e->tensors[tid].chmap[refc + c] = internal_off[i] + c;
      refc += C;
      concat_slices[kv.first].push_back({ins[i], internal_off[i]});
    }
  }
  e->pools.clear();
  for (auto& p : pools) {
    PoolOp po;
    po.in_tensor = e->blob_tensor.at(p.in);
    po.out_tensor = e->blob_tensor.at(p.out);
    po.C = e->tensors[po.out_tensor].C;
    e->pools.push_back(po);
  }
  // conv inputs / destinations
  for (auto& c : e->convs) {
    const LayerDef* L = nullptr;
    for (auto& l : net.layers) if (l.type == "Convolution" && l.name == c.name) L = &l;
    if (c.first) c.in_tensor = packed_tensor;
    else {
      auto it = e->blob_tensor.find(L->bottoms[0]);
      if (it == e->blob_tensor.end()) return fail(e, RTP_EINVAL, "layer %s: bottom %s has no tensor", c.name.c_str(), L->bottoms[0].c_str());
      c.in_tensor = it->second;
    }
    c.dsts.push_back({e->blob_tensor[c.name], 0});
    for (auto& cs : concat_slices)
      for (auto& sl : cs.second)
        if (sl.first == c.name) c.dsts.push_back({e->blob_tensor[cs.first], sl.second});
    if ((int)c.dsts.size() > RTP_MAX_DST) return fail(e, RTP_EINVAL, "layer %s feeds %d tensors (max %d)", c.name.c_str(), (int)c.dsts.size(), RTP_MAX_DST);
    // low-res output (the blob ImResize reads), reference channel order
    if (c.name == e->lowres_blob) { c.to_lowres = true; c.lowres_coff = 0; }
    else if (concats.count(e->lowres_blob)) {
      int off = 0;
      for (auto& in : concats[e->lowres_blob].inputs) {
        if (in == c.name) { c.to_lowres = true; c.lowres_coff = off; }
        off += e->blob_dims[in].first;
      }
    }
    const Tensor& ti = e->tensors[c.in_tensor];
    c.Cin_p = ti.Cp;
    c.rowb = (c.Cin_p * e->elem >= 128) ? 128 : 64;
    if ((c.Cin_p * e->elem) % c.rowb) return fail(e, RTP_EINVAL, "layer %s: internal channel padding error", c.name.c_str());
    c.nchunk = c.Cin_p * e->elem / c.rowb;
  }

  // split precision: which layers, then which tensors must carry a lo block
  for (auto& c : e->convs) {
    layer_split(e, c, &c.split_w, &c.split_a, &c.no_h8);
    if (c.first) c.split_a = false;  // the image (u8/256 - 0.5) is exact in fp16: its lo part is zero
  }

  // steps + pairing + tile configuration
  e->steps.clear();
  e->steps.push_back({0, -1, -1});
  for (size_t oi = 0; oi < order.size(); ++oi) {
    if (order[oi].first == 2) { e->steps.push_back({2, order[oi].second, -1}); continue; }
    const int a = order[oi].second;
    int b = -1;
    if (oi + 1 < order.size() && order[oi + 1].first == 1) {
      const int cand = order[oi + 1].second;
      const ConvOp& A = e->convs[a];
      const ConvOp& B = e->convs[cand];
      bool dep = false;
      for (auto& d : A.dsts) if (d.first == B.in_tensor) dep = true;
      if (!dep && A.split_a == B.split_a && A.split_w == B.split_w && A.k_eff == B.k_eff && A.Cin_p == B.Cin_p && A.level == B.level && A.relu == B.relu && A.rowb == B.rowb &&
          round_up(A.cout, 64) == round_up(B.cout, 64) && e->tensors[A.in_tensor].Cp == e->tensors[B.in_tensor].Cp)
        b = cand;
    }
    e->steps.push_back({1, a, b});
    if (b >= 0) ++oi;
  }
  for (size_t si_ = 0; si_ < e->steps.size(); ++si_) {
    Step& s = e->steps[si_];
    if (s.type != 1) continue;
    ConvOp& A = e->convs[s.a];
    const Geom& g = e->geom[A.level];
    // a pooling layer follows and reads only this blob: tiles the POOL kernel exists for save its launch (fusion pass below)
    const bool pool_next = s.b < 0 && si_ + 1 < e->steps.size() && e->steps[si_ + 1].type == 2 && A.dsts.size() == 1 &&
                           pools[e->steps[si_ + 1].a].in == A.name && (g.H % 2) == 0 && (g.W % 2) == 0 && g.W >= 128 && !e->cfg.keep_blobs;
    const int nprob = s.b >= 0 ? 2 : 1;
    const int maxcout = std::max(A.cout, s.b >= 0 ? e->convs[s.b].cout : 0);
    const long M = (long)g.H * g.Wp;
    // Tile choice by a time model of the ring kernel (cycles; the constants are measured):
    //   one K step (tap x chunk) of a workgroup = max(MFMA time, L2->LDS time) + barrier:
    //     MFMA:  BM*BN*channels_per_chunk / (4 consumer waves * 32*32*16) instructions per wave at ~43 cycles on real operands
    //     DMA:   the weight tile (BN rows) + 1/k of the (BM+k-1)-pixel strip, at ~56 B/clk/CU
    //   a workgroup = steps * that + ~4500 cycles of prologue / epilogue; a launch = workgroups / 256 CUs rounds, where a partial
    //   round of fill f costs 0.5 + 0.5 f of a full one (fewer busy CUs clock higher and wait less for L2: 372 workgroups of the
    //   dominant shape take 1.70x the time of 248, not 2x), + ~6000 cycles of dispatch per launch.  (Two co-resident workgroups of
    //   the small-LDS 64x64 kernel share one matrix pipe: no credit for them.)
    // Replaces round 2's "fewest bytes among the tiles with >= 224 workgroups", which left plans whose M does not fill the chip
    // (MPI 46x62 maps at batch_frames 2: 192 workgroups of 128x64) on half-size tiles in two rounds.
    const bool ring_ok = !A.first && (A.k_eff == 3 || A.k_eff == 7);
    const int row_bytes_all = A.Cin_p * e->elem;
    std::vector<int> cands;
    if (A.rowb == 64) cands = {CFG_128x64};
    else if (maxcout <= 32 && ring_ok && row_bytes_all % 256 == 0) cands = {CFG_128x32, CFG_128x64, CFG_64x64};
    else if (maxcout <= 64) cands = {CFG_128x64, CFG_64x64};
    else if (ring_ok && row_bytes_all % 256 == 0) {
      cands = {CFG_128x128, CFG_64x128, CFG_128x64, CFG_64x64, CFG_128x32};
      // CFG_256x64 (round 5): twice the pixels per workgroup for the 7x7 layers.  Alone at batches of 4 it is 12-14 % faster per image than the
      // 128x64 tile at batches of 2; in the PIPELINE it changes nothing (B = 4: 1012 vs 1006 frames/s against 128x128 tiles, MPI B = 5: 1209 vs
      // 1207; profiles/r05_experiments.txt), so the production plans keep round 4's measured tiles and the candidate exists in the experiments
      // build only (RTP_TILE_256=1: let the time model choose it; RTP_DOM_256 forces it).
      static const char* t256 = RTP_EXP_ENV("RTP_TILE_256");
      if (e->prec == 0 && A.k_eff == 7 && maxcout >= 64 && ((t256 && t256[0] == '1') || RTP_EXP_ENV("RTP_DOM_256"))) cands.push_back(CFG_256x64);   // (chosen by the model where 128-pixel tiles would need two rounds: batches of >= 4 images)
    }
    else if (ring_ok) cands = {CFG_128x128, CFG_64x128, CFG_128x64, CFG_64x64};
    else cands = {CFG_128x128, CFG_64x128, CFG_64x64};
    int best = cands.back();
    double best_t = 1e300, best_bytes = 1e300;
    static const char* tm = RTP_EXP_ENV("RTP_TILE_RULE");  // experiments: "r2" = round 2's rule
    const bool rule_r2 = tm && !strcmp(tm, "r2");
    long best_wg = -1;
    bool chosen = false;
    const int passes = (A.split_a || A.split_w) ? ((e->split_fp8 && e->mode == RTP_PREC_MIXED && A.split_a && A.split_w && !A.no_h8 && ring_ok) ? 2 : 1 + (A.split_a ? 1 : 0) + (A.split_w ? 1 : 0)) : 1;
    for (int cf : cands) {
      const ConvCfgInfo ci = conv_cfg_info(cf);
      const long wg = ((M + ci.BM - 1) / ci.BM) * e->NI * (round_up(maxcout, ci.BN) / ci.BN) * nprob;
      const double bytes = (double)wg * (ci.BN + (double)(ci.BM + A.k_eff - 1) / A.k_eff);
      if (rule_r2) {
        if (wg >= 224) { if (!chosen || bytes < best_bytes) { best = cf; best_bytes = bytes; chosen = true; } }
        else if (!chosen && wg > best_wg) { best = cf; best_wg = wg; }
        continue;
      }
      const int chb = (ring_ok && (cf == CFG_64x64 || cf == CFG_128x64 || cf == CFG_128x32) && row_bytes_all % 256 == 0) ? 256 : std::min(128, row_bytes_all);
      const double chc = (double)chb / e->elem;                                     // channels per chunk
      const double t_mfma = (double)ci.BM * ci.BN * chc / (4.0 * 32 * 32 * 16) * (e->prec ? 4 * 43.0 : 43.0);
      const double t_dma = ((double)ci.BN * chb + (double)(ci.BM + A.k_eff - 1) * chb / A.k_eff) / 56.0;
      const double steps = (double)A.k_eff * A.k_eff * (row_bytes_all / (double)chb) * passes;
      const double t_wg = steps * (std::max(t_mfma, t_dma) + 60.0) + 4500.0;
      const double full = std::floor((double)wg / 256.0), frac = (double)wg / 256.0 - full;
      double t = (full + (frac > 0 ? 0.5 + 0.5 * frac : 0.0)) * t_wg + 6000.0;
      if (pool_next && !(e->prec == 0 && A.k_eff == 3 && chb == 128 && (cf == CFG_128x64 || cf == CFG_128x128)))  // the stand-alone pooling launch: ~5 B per cycle and CU
        t += 8000.0 + (double)e->NI * g.H * g.W * e->tensors[A.dsts[0].first].stride() * e->elem * 1.25 / (256.0 * 9.0);
      if (t < best_t * 0.98 || (t < best_t * 1.02 && bytes < best_bytes)) { best = cf; best_t = std::min(t, best_t); best_bytes = bytes; }
    }
    // Half-chip launches.  The runtime's hardware queues run two conv stacks at a time, so two launches of <= 128
    // workgroups share the chip: each workgroup moves half the weight bytes per MFMA and one stack's launch gaps are covered by the
    // other's kernel.  Measured in the pipeline, same box (profiles/r03_tile_model.txt): 128x128 tiles (124 workgroups at batch_frames 2)
    // for every k x k layer at 1/8 resolution +2.7 % frames/s, for all of them except the dominant shape +1.5..3 %.  A launch alone then
    // fills half the chip, which is what a per-launch roofline reports (0.16 instead of 0.23 for the dominant 7x7 128->128 pair).
    // RTP_HALF_CHIP: 1 (default) = where a 128x128 tile gives 100..128 workgroups and the model's choice 129..256, except the dominant
    // shape (whose per-launch efficiency is the figure this path is judged on); 2 = the dominant shape too; 0 = the model alone.
    if (!rule_r2 && ring_ok) {
      static const char* hc = RTP_EXP_ENV("RTP_HALF_CHIP");
      const int mode = hc ? atoi(hc) : 1;
      const ConvCfgInfo cb = conv_cfg_info(best);
      const long wg_best = ((M + cb.BM - 1) / cb.BM) * e->NI * (round_up(maxcout, cb.BN) / cb.BN) * nprob;
      const long wg_128 = ((M + 127) / 128) * e->NI * (round_up(maxcout, 128) / 128) * nprob;
      const bool has128 = std::find(cands.begin(), cands.end(), (int)CFG_128x128) != cands.end();
      const bool dominant = A.k_eff == 7 && A.cin == 128;
      if (mode > 0 && has128 && best != CFG_128x128 && wg_best > 128 && wg_best <= 256 && wg_128 >= 100 && wg_128 <= 128 && (mode >= 2 || !dominant))
        best = CFG_128x128;
    }
    {
      static const char* d256 = RTP_EXP_ENV("RTP_DOM_256");   // experiments: 1 = 256x64 tiles for the 7x7 layers whatever the batch (half-chip launches of double-size workgroups at batch_frames 2); 2 = the dominant shape only
      if (d256 && ring_ok && A.k_eff == 7 && std::find(cands.begin(), cands.end(), (int)CFG_256x64) != cands.end() && (d256[0] == '1' || (d256[0] == '2' && A.cin == 128))) best = CFG_256x64;
    }
    if (const char* ov = RTP_EXP_ENV("RTP_TILE_OVERRIDE")) {  // experiments: "conv2_1=3,conv3_1=3" forces tile ids (kernels.h ConvCfg) per layer
      const std::string key = A.name + "=";
      for (const char* hit = strstr(ov, key.c_str()); hit; hit = strstr(hit + 1, key.c_str())) {  // "Mconv2_1=.." also contains "conv2_1=": take the entry that starts at a boundary
        if (!(hit == ov || hit[-1] == ',')) continue;
        const int v = atoi(hit + key.size());
        if (std::find(cands.begin(), cands.end(), v) == cands.end())
          return fail(e, RTP_EINVAL, "RTP_TILE_OVERRIDE: tile id %d is not a candidate for layer %s", v, A.name.c_str());
        best = v;
        break;
      }
    }
    {
      static const char* fc = RTP_EXP_ENV("RTP_FORCE_CFG");  // experiments only: force a tile for the k x k layers at 1/8 resolution
      static const char* kd = RTP_EXP_ENV("RTP_FORCE_CFG_KEEP_DOM");  // 1: ... except the dominant shape (7x7, 128 input channels)
      if (fc && ring_ok && A.level == 3 && maxcout > 64 && !(kd && kd[0] == '1' && A.k_eff == 7 && A.cin == 128)) {
        const int v = atoi(fc);
        if (std::find(cands.begin(), cands.end(), v) == cands.end())
          return fail(e, RTP_EINVAL, "RTP_FORCE_CFG: tile id %d is not a candidate for layer %s", v, A.name.c_str());
        best = v;
      }
    }
    const ConvCfgInfo ci = conv_cfg_info(best);
    const char* force = RTP_EXP_ENV("RTP_CONV_IMPL");
    const bool allow_ring = !(force && !strcmp(force, "v1"));
    for (int idx : {s.a, s.b}) {
      if (idx < 0) continue;
      ConvOp& c = e->convs[idx];
      c.cfg = best;
      c.CoutP = round_up(maxcout, ci.BN);
      c.impl = 0;
      if (allow_ring && !c.first && (c.k_eff == 3 || c.k_eff == 7)) {
        const int row_bytes = c.Cin_p * e->elem;
        static const char* f128 = RTP_EXP_ENV("RTP_RING_CHB128");
        int chb = ((best == CFG_64x64 || best == CFG_128x64 || best == CFG_128x32) && row_bytes % 256 == 0 && !(f128 && f128[0] == '1')) ? 256 : 128;
        if (best == CFG_128x32 && chb != 256) { best = CFG_64x64; c.cfg = best; c.CoutP = round_up(maxcout, 64); chb = 128; }
        if (row_bytes % chb == 0) {
          c.impl = 1;
          c.rowb = chb;
          c.nchunk = row_bytes / chb;
        }
      }
    }
  }

  // fp8 compensation where the kernel supports it: ring kernels whose waves own >= 64 bytes of K per chunk
  for (auto& c : e->convs) {
    // k-split of the kernel that would run it (conv_ring.hip; q layers on the 64x64 tile with 128-byte chunks get a 2-way split)
    const int ksplit = c.cfg == CFG_128x128 ? 1 : (c.cfg == CFG_64x64 ? (c.rowb == 128 ? 2 : 4) : 2);
    const int gpw = (c.rowb / 32) / ksplit;
    c.h8 = e->split_fp8 && e->mode == RTP_PREC_MIXED && e->prec == 0 && c.impl == 1 && c.split_a && c.split_w && !c.no_h8 && gpw >= 2 && gpw % 2 == 0;
  }
  for (auto& s : e->steps)  // both branches of a pair run the same kernel
    if (s.type == 1 && s.b >= 0 && e->convs[s.a].h8 != e->convs[s.b].h8)
      for (int idx : {s.a, s.b}) e->convs[idx].h8 = false;
  for (auto& c : e->convs)  // which operand blocks the input tensors must carry, from the FINAL flags
    if (c.split_a) { if (c.h8) e->tensors[c.in_tensor].need_q = true; else e->tensors[c.in_tensor].need_lo = true; }
  for (size_t pi = e->pools.size(); pi-- > 0;) {  // a pool output with lo / q parts needs them in its input
    if (e->tensors[e->pools[pi].out_tensor].need_lo) e->tensors[e->pools[pi].in_tensor].need_lo = true;
    if (e->tensors[e->pools[pi].out_tensor].need_q) e->tensors[e->pools[pi].in_tensor].need_q = true;
  }
  // 2x2 max pooling inside the producing convolution's epilogue: the pooling layer's input blob has no other consumer, the layer
  // runs on a ring kernel with 128-pixel tiles of 128-byte chunks (the trunk's conv1_2 / conv2_2 / conv3_4), even resolution.
  // The un-pooled blob is then never written (rtp_config.keep_blobs = 1 keeps every blob tappable and pools in its own launch).
  {
    static const char* fp = RTP_EXP_ENV("RTP_FUSE_POOL");  // experiments: 0 = stand-alone pooling launches
    for (size_t si = 1; si < e->steps.size() && !e->cfg.keep_blobs && !(fp && fp[0] == '0'); ++si) {
      if (e->steps[si].type != 2 || e->steps[si - 1].type != 1 || e->steps[si - 1].b >= 0) continue;
      const int pi = e->steps[si].a;
      ConvOp& A = e->convs[e->steps[si - 1].a];
      const PoolOp& po = e->pools[pi];
      const Geom& g = e->geom[A.level];
      bool ok = e->prec == 0 && A.impl == 1 && A.k_eff == 3 && A.rowb == 128 && (A.cfg == CFG_128x64 || A.cfg == CFG_128x128) && !A.to_lowres &&
                A.dsts.size() == 1 && A.dsts[0].first == po.in_tensor && (g.H % 2) == 0 && (g.W % 2) == 0 && g.W >= 128 /* one wrap per tile at most */ && A.level + 1 < e->nlevels;
      for (auto& c : e->convs) if (c.in_tensor == po.in_tensor) ok = false;  // somebody convolves the un-pooled blob
      if (!ok) continue;
      A.pool = pi;
      A.dsts[0] = {po.out_tensor, 0};
      e->tensors[po.in_tensor].written = false;
      e->steps.erase(e->steps.begin() + (long)si);
      --si;
    }
  }
  {  // the input convolution without the im2col tensor: fp16 storage, 64 channels, one plain destination
    static const char* fd = RTP_EXP_ENV("RTP_FIRST_DIRECT");  // experiments: 0 = the pack + 1x1 route
    for (size_t si = 0; si + 1 < e->steps.size() && !(fd && fd[0] == '0'); ++si) {
      const Step& s1 = e->steps[si];
      if (s1.type != 1 || s1.b >= 0 || !e->convs[s1.a].first) continue;
      ConvOp& c = e->convs[s1.a];
      const Tensor& to = e->tensors[c.dsts[0].first];
      if (e->prec != 0 || c.cout != 64 || c.split_w || c.dsts.size() != 1 || c.to_lowres || to.need_lo || to.need_q || e->steps[0].type != 0) break;
      if (((size_t)3 * (e->geom[0].W + 2) * 3 + 8) * 2 > 64 * 1024) break;
      c.direct_first = true;
      e->steps[0] = Step{4, s1.a, -1};
      e->steps.erase(e->steps.begin() + (long)si);
      break;
    }
  }
  for (auto& c : e->convs) {  // K chunks: one pass = ncp chunks of rowb bytes; split layers run 2-3 passes (h8: hi chunks + q chunks)
    c.ncp = c.nchunk;
    c.nchunk = c.ncp * c.passes();
  }
  // branch tails: 1x1 (ReLU) -> 1x1 with nobody else reading the middle blob become ONE launch (conv_pw2.hip)
  {
    static const char* nf = RTP_EXP_ENV("RTP_FUSE_1X1");
    const bool allow = e->prec == 0 && !(nf && nf[0] == '0');
    for (size_t si = 0; allow && si + 1 < e->steps.size(); ++si) {
      Step& s1 = e->steps[si];
      const Step& s2 = e->steps[si + 1];
      if (s1.type != 1 || s2.type != 1 || (s1.b >= 0) != (s2.b >= 0)) continue;
      auto chain = [&](int ia, int ic) {
        const ConvOp& A = e->convs[ia];
        const ConvOp& C = e->convs[ic];
        return A.k == 1 && C.k == 1 && !A.first && A.Cin_p == 128 && A.cout % 128 == 0 && A.cout <= 512 /* conv_pw2.hip PW_MAXMID */ && C.cin == A.cout && A.dsts.size() == 1 &&
               C.in_tensor == A.dsts[0].first && C.cout <= 64 && e->tensors[A.dsts[0].first].C == A.cout;
      };
      if (!chain(s1.a, s2.a) || (s1.b >= 0 && !chain(s1.b, s2.b))) continue;
      if (s1.b >= 0 && (e->convs[s1.a].cout != e->convs[s1.b].cout)) continue;
      s1.type = 3; s1.a2 = s2.a; s1.b2 = s2.b;
      // the middle blob (Mconv6_stageK / conv5_4_CPM) lives in LDS between the two GEMMs; nobody reads it from memory: it is written only
      // when every blob must stay tappable (keep_blobs) — 32-128 KB of stores per workgroup and ~0.9 us of its ~11 us otherwise
      for (int idx : {s1.a, s1.b}) {
        if (idx < 0 || e->cfg.keep_blobs) continue;
        const int mid = e->convs[idx].dsts[0].first;
        bool read_elsewhere = false;
        for (auto& c2 : e->convs) if (c2.in_tensor == mid && &c2 != &e->convs[idx == s1.a ? s1.a2 : s1.b2]) read_elsewhere = true;
        for (auto& po : e->pools) if (po.in_tensor == mid) read_elsewhere = true;
        if (!read_elsewhere) e->tensors[mid].written = false;
      }
      for (int idx : {s1.a, s1.b}) if (idx >= 0) { ConvOp& A = e->convs[idx]; A.fused = 1; A.fused_chunks = A.cout / 128; A.CoutP = A.cout; }
      for (int idx : {s1.a2, s1.b2}) if (idx >= 0) { ConvOp& C = e->convs[idx]; C.fused = 2; C.fused_chunks = C.cin / 128; C.CoutP = 64; }
      e->steps.erase(e->steps.begin() + si + 1);
    }
  }
  // arena layout
  size_t off = 0;
  for (auto& t : e->tensors) {
    if (!t.written) { t.offset = 0; continue; }  // fused away (its convolution pools in the epilogue): never read, never written, no space
    const Geom& g = e->geom[t.level];
    const size_t pix_bytes = (size_t)t.stride() * e->elem;
    off = round_up_sz(off, 256);
    off += GUARD_PIX * pix_bytes;
    off = round_up_sz(off, 256);
    t.offset = off;
    off += (size_t)e->NI * g.img_pix * pix_bytes + GUARD_PIX * pix_bytes;
  }
  e->arena_bytes = round_up_sz(off, 256) + (4u << 20);  // tail pad: the ring kernel's dummy prefetches read past the last strip
  // weight arena
  size_t woff = 0;
  for (auto& c : e->convs) {
    c.w_bytes = (size_t)c.k_eff * c.k_eff * c.nchunk * c.CoutP * c.rowb;
    if (c.fused == 1) c.w_bytes = (size_t)c.fused_chunks * (c.split_w ? 2 : 1) * 128 * 256;
    if (c.fused == 2) c.w_bytes = (size_t)c.fused_chunks * (c.split_w ? 2 : 1) * 64 * 256;
    if (c.direct_first) c.w_bytes = 2 * 2 * 64 * 16;
    woff = round_up_sz(woff, 256);
    c.w_off = woff;
    woff += c.w_bytes;
    woff = round_up_sz(woff, 256);
    c.b_off = woff;
    woff += (size_t)c.CoutP * sizeof(float);
  }
  e->weights_bytes = round_up_sz(woff, 256) + (1u << 20);  // tail pad: dummy weight-tile prefetches of the last layer
  // dominant conv step for the roofline probe: the first paired 7x7 step whose input is not a concat
  e->dominant_step = -1;
  for (size_t si = 0; si < e->steps.size(); ++si) {
    const Step& s = e->steps[si];
    static const char* dq = RTP_EXP_ENV("RTP_DOMINANT_Q");  // profiling: 1 = probe the first fp8-compensated launch of that shape instead (stage 4)
    if (s.type == 1 && e->convs[s.a].k == 7 && e->convs[s.a].cin == 128 && (!(dq && dq[0] == '1') || e->convs[s.a].h8)) { e->dominant_step = (int)si; break; }
  }
  e->strip_rows = e->N > 1 ? 16 : 8;  // several scales: the row interpolations of a strip are the larger share, taller strips amortise them (+3 % frames/s at 3 scales)
  if (const char* sr = RTP_EXP_ENV("RTP_NMS_STRIP_ROWS")) { const int v = atoi(sr); if (v >= 2 && v <= 16) e->strip_rows = v; }  // experiments
  // the strip kernel keeps (strip_rows + 2 + NMSF_TROWS) rows of W floats + a W x 8-byte column table in LDS (postproc.hip, 150 KiB cap)
  while (e->strip_rows > 2 && ((size_t)(e->strip_rows + 2 + 8 /* NMSF_TROWS */) * e->cfg.net_w * 4 + (size_t)e->cfg.net_w * 8) > 150 * 1024) e->strip_rows /= 2;
  if (((size_t)(e->strip_rows + 2 + 8 /* NMSF_TROWS */) * e->cfg.net_w * 4 + (size_t)e->cfg.net_w * 8) > 150 * 1024)
    return fail(e, RTP_EINVAL, "net_resolution width %d is too large for the fused ImResize+Nms strip kernel", e->cfg.net_w);
  e->nstrips = (e->cfg.net_h + e->strip_rows - 1) / e->strip_rows;
  e->max_rows = e->num_limbs * e->max_peaks;
  {
    // connect kernels: sort keys hold 7-bit peak ordinals; the subset table is int16 in LDS
    const size_t lds2 = (size_t)e->max_rows * (sizeof(double) + sizeof(short) + sizeof(short) * e->num_parts);
    if (e->max_peaks > 127 || lds2 > 150 * 1024) return fail(e, RTP_EINVAL, "max_peaks %d out of range [1,127]", e->max_peaks);
  }
  return RTP_OK;
}

// ---- weight packing -------------------------------------------------------------------------
// float -> OCP e4m3 (round to nearest even, subnormals kept, clamped to +-448): what v_cvt_pk_fp8_f32 does after the clamp
unsigned char f32_to_e4m3(float x) {
  const unsigned sign = std::signbit(x) ? 0x80u : 0u;
  float a = std::fabs(x);
  if (a != a) return 0x7f;
  if (a >= 448.f) return (unsigned char)(sign | 0x7e);
  if (a >= 0.015625f) {  // normal: 2^-6 and up
    int e;
    (void)std::frexp(a, &e);                       // a in [2^(e-1), 2^e)
    const float scaled = std::ldexp(a, -(e - 1));  // [1, 2)
    int M = (int)std::nearbyint((scaled - 1.f) * 8.f);
    int E = e - 1 + 7;
    if (M == 8) { M = 0; ++E; }
    if (E > 15 || (E == 15 && M == 7)) return (unsigned char)(sign | 0x7e);
    return (unsigned char)(sign | (unsigned)(E << 3) | (unsigned)M);
  }
  const int M = (int)std::nearbyint(std::ldexp(a, 9));  // subnormal: M * 2^-9, M = 8 is the smallest normal
  return (unsigned char)(sign | (unsigned)(M == 8 ? 8 : M));
}

template <typename T>
void pack_conv(const rtp_engine* e, const ConvOp& c, const std::vector<float>& w, const std::vector<float>& b,
               std::vector<unsigned char>* out_w, std::vector<float>* out_b) {
  const Tensor& ti = e->tensors[c.in_tensor];
  const int per_chunk = c.rowb / (int)sizeof(T);
  const int taps = c.k_eff * c.k_eff;
  out_w->assign(c.w_bytes, 0);
  T* pw = (T*)out_w->data();
  // element index of internal channel kk of output row n inside its rowb-byte row (ring kernels
  // XOR the 16-byte chunk index with a function of the row, see conv_ring.hip)
  auto kpos = [&](int n, int kk) {
    if (c.impl == 0) return kk;
    const int per16 = 16 / (int)sizeof(T);
    const int c16 = kk / per16, within = kk % per16;
    return ((c16 ^ conv_ring_swz(c.rowb, n)) * per16) + within;
  };
  if (c.direct_first) {  // conv_first.hip: A-operand fragments [tile][K-step][lane][8 taps], tap k = (r*3 + s)*3 + cc
    for (int t = 0; t < 2; ++t)
      for (int j = 0; j < 2; ++j)
        for (int l = 0; l < 64; ++l) {
          const int n = t * 32 + conv_first_channel_of_row(l & 31);
          for (int e2 = 0; e2 < 8; ++e2) {
            const int k = j * 16 + (l >> 5) * 8 + e2;
            float wv = 0.f;
            if (k < 27) { const int r = k / 9, s2 = (k % 9) / 3, cc = k % 3; wv = w[((size_t)(n * 3 + cc) * 3 + r) * 3 + s2]; }
            pw[((size_t)(t * 2 + j) * 64 + l) * 8 + e2] = (T)wv;
          }
        }
    out_b->assign(c.CoutP, 0.f);
    for (int n = 0; n < c.cout; ++n) (*out_b)[n] = b[n];
    return;
  }
  if (c.fused) {  // conv_pw2.hip: [chunk][part (hi, lo)][rows][128 k], rows = 128 middle channels (first) / 64 outputs (second)
    const int rows = c.fused == 1 ? 128 : 64;
    const int parts = c.split_w ? 2 : 1;
    for (int n = 0; n < c.cout; ++n)
      for (int cr = 0; cr < c.cin; ++cr) {
        const int ci = ti.chmap[cr];  // internal channel of the input tensor (identity for the middle blob)
        const int chunk = c.fused == 1 ? n / 128 : ci / 128;
        const int row = c.fused == 1 ? n % 128 : n;
        const int k = ci % 128;
        const float wv = w[(size_t)n * c.cin + cr];
        const T hi = (T)wv;
        pw[(((size_t)chunk * parts + 0) * rows + row) * 128 + k] = hi;
        if (c.split_w) pw[(((size_t)chunk * parts + 1) * rows + row) * 128 + k] = (T)(wv - (float)hi);
      }
    out_b->assign(c.CoutP, 0.f);
    for (int n = 0; n < c.cout; ++n) (*out_b)[n] = b[n];
    return;
  }
  // internal channel -> reference index
  if (c.first) {
    // internal channel j = (r*3+s)*3 + cc  <->  W[n][cc][r][s]
    for (int n = 0; n < c.cout; ++n)
      for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s)
          for (int cc = 0; cc < 3; ++cc) {
            const int j = (r * 3 + s) * 3 + cc;
            const int chunk = j / per_chunk, kk = j % per_chunk;
            const float wv = w[((size_t)(n * 3 + cc) * 3 + r) * 3 + s];
            const T hi = (T)wv;
            pw[((size_t)(0 * c.nchunk + chunk) * c.CoutP + n) * per_chunk + kpos(n, kk)] = hi;
            if (c.split_w) pw[((size_t)(0 * c.nchunk + c.ncp + chunk) * c.CoutP + n) * per_chunk + kpos(n, kk)] = (T)(wv - (float)hi);
          }
  } else {
    for (int n = 0; n < c.cout; ++n)
      for (int cr = 0; cr < c.cin; ++cr) {
        const int ci = ti.chmap[cr];
        const int chunk = ci / per_chunk, kk = ci % per_chunk;
        for (int r = 0; r < c.k; ++r)
          for (int s = 0; s < c.k; ++s) {
            // register-staged kernel: [tap][chunk]; ring kernel: step order [r][chunk][s]
            const int tap = r * c.k + s;
            const float wv = w[((size_t)(n * c.cin + cr) * c.k + r) * c.k + s];
            const T hi = (T)wv;
            const T lo = (T)(wv - (float)hi);
            if (c.h8) {  // [hi chunks: fp16 W_hi] [q chunks: per 64-channel group 64 B fp8(W_hi * 2^wq) | 64 B fp8(W_lo * 2^(wq+11))]
              const size_t tile0 = (size_t)(r * c.nchunk + chunk) * c.k + s;
              pw[(tile0 * c.CoutP + n) * per_chunk + kpos(n, kk)] = hi;
              const size_t tileq = (size_t)(r * c.nchunk + c.ncp + chunk) * c.k + s;
              unsigned char* qrow = out_w->data() + (tileq * c.CoutP + n) * c.rowb;
              const int g = kk / 64, pos = kk % 64;
              auto qpos = [&](int byte) { return ((byte / 16) ^ conv_ring_swz(c.rowb, n)) * 16 + byte % 16; };
              qrow[qpos(g * 128 + pos)] = f32_to_e4m3(std::ldexp((float)hi, c.wq_exp));
              qrow[qpos(g * 128 + 64 + pos)] = f32_to_e4m3(std::ldexp(wv - (float)hi, c.wq_exp + 11));
              continue;
            }
            // passes of a split layer are further chunks of the K loop: [a_hi x W_hi] [a_lo x W_hi] [a_hi x W_lo]
            int vbase = 0;
            for (int pass = 0; pass < 3; ++pass) {
              if ((pass == 1 && !c.split_a) || (pass == 2 && !c.split_w)) continue;
              const int vchunk = vbase + chunk;
              vbase += c.ncp;
              const size_t tile = c.impl == 1 ? ((size_t)(r * c.nchunk + vchunk) * c.k + s) : ((size_t)tap * c.nchunk + vchunk);
              pw[(tile * c.CoutP + n) * per_chunk + kpos(n, kk)] = pass == 2 ? lo : hi;
            }
          }
      }
  }
  out_b->assign(c.CoutP, 0.f);
  for (int n = 0; n < c.cout; ++n) (*out_b)[n] = b[n];
}

// h8 layers: one power-of-two scale for the fp8 weight copies, shared by the two branches of a paired launch
void compute_wq_exp(rtp_engine* e) {
  for (auto& s : e->steps) {
    if (s.type != 1 || !e->convs[s.a].h8) continue;
    float mx = 0.f;
    for (int idx : {s.a, s.b}) if (idx >= 0) for (float v : e->w_ref[idx]) mx = std::max(mx, std::fabs(v));
    int ex = mx > 0.f ? (int)std::floor(std::log2(448.0 / mx)) : 0;
    ex = std::max(-20, std::min(ex, 40));
    for (int idx : {s.a, s.b}) if (idx >= 0) e->convs[idx].wq_exp = ex;
  }
}

int upload_conv_weights(rtp_engine* e, int i) {
  const ConvOp& c = e->convs[i];
  std::vector<unsigned char> pw;
  std::vector<float> pb;
  if (e->prec == 0) pack_conv<_Float16>(e, c, e->w_ref[i], e->b_ref[i], &pw, &pb);
  else pack_conv<float>(e, c, e->w_ref[i], e->b_ref[i], &pw, &pb);
  SYNC_GUARD;   // (the packing above, the expensive part, runs in parallel across engines)
  HIPCHK(e, hipMemcpy(e->dweights + c.w_off, pw.data(), pw.size(), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(e->dweights + c.b_off, pb.data(), pb.size() * sizeof(float), hipMemcpyHostToDevice));
  return RTP_OK;
}

// Every layer of the plan: packed on a few host threads (52 M weights through the kernels' staging order, swizzle and fp8 conversions: the
// longest part of creating or re-planning an engine when it runs on one thread), then uploaded into ONE host image of the arena with a
// single copy.
int upload_all_weights(rtp_engine* e) {
  std::vector<unsigned char> image(e->weights_bytes, 0);
  const int n = (int)e->convs.size();
  const int nthr = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, n}));
  std::atomic<int> next{0};
  auto work = [&]() {
    std::vector<unsigned char> pw;
    std::vector<float> pb;
    for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
      const ConvOp& c = e->convs[i];
      if (e->prec == 0) pack_conv<_Float16>(e, c, e->w_ref[i], e->b_ref[i], &pw, &pb);
      else pack_conv<float>(e, c, e->w_ref[i], e->b_ref[i], &pw, &pb);
      memcpy(image.data() + c.w_off, pw.data(), pw.size());                      // disjoint ranges of the image
      memcpy(image.data() + c.b_off, pb.data(), pb.size() * sizeof(float));
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nthr; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  SYNC_GUARD;
  HIPCHK(e, hipMemcpy(e->dweights, image.data(), e->weights_bytes, hipMemcpyHostToDevice));
  return RTP_OK;
}

// ---- launches -------------------------------------------------------------------------------
// Residency-stamp slots of one batch context: [0, 64) plan steps (conv stack), 64 + 8 j + {0 strip, 1 write, 2 pairs, 3 match, 4 assemble}
// = frame j's post-processing chain, 200 + 2 j + {0 warp, 1 area/pad} = frame j's device pre-processing.
constexpr int STAMP_SLOTS = 256;
unsigned long long* stamp_slot(const rtp_engine* e, const Ctx& cx, int idx) {
  return (e->stamp_probe && cx.stamps && idx >= 0 && idx < STAMP_SLOTS) ? cx.stamps + 2 * (size_t)idx : nullptr;
}
// the stamps of the batch that last ran on `cx` (complete: the context is idle) -> e->stamp_spans; slots zeroed for the next batch
int stamp_harvest(rtp_engine* e, Ctx& cx) {
  if (!cx.stamps || !cx.stamps_dirty) return RTP_OK;
  std::vector<unsigned long long> h(2 * STAMP_SLOTS);
  HIPCHK(e, hipMemcpy(h.data(), cx.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(e, hipMemset(cx.stamps, 0, h.size() * sizeof(unsigned long long)));
  cx.stamps_dirty = false;
  for (int i = 0; i < STAMP_SLOTS; ++i) {
    if (!h[2 * i + 1] || !h[2 * i]) continue;
    const unsigned long long t0 = ~h[2 * i], t1 = h[2 * i + 1];
    if (!e->stamp_base) e->stamp_base = t0;
    if (e->stamp_spans.size() < 3 * (size_t)(1 << 20)) {
      e->stamp_spans.push_back((float)i);
      e->stamp_spans.push_back((float)((double)((long long)(t0 - e->stamp_base)) * 0.01));
      e->stamp_spans.push_back((float)((double)((long long)(t1 - e->stamp_base)) * 0.01));
    }
  }
  return RTP_OK;
}

void fill_problem(const rtp_engine* e, const Ctx& cx, const ConvOp& c, ConvProblem* pr) {
  memset(pr, 0, sizeof(*pr));
  pr->in = cx.arena + e->tensors[c.in_tensor].offset;
  pr->w = e->dweights + c.w_off;
  pr->bias = (const float*)(e->dweights + c.b_off);
  pr->ndst = (int)c.dsts.size();
  for (int d = 0; d < pr->ndst; ++d) {
    const Tensor& t = e->tensors[c.dsts[d].first];
    pr->dst[d].base = cx.arena + t.offset;
    pr->dst[d].cstride = t.stride();
    pr->dst[d].coff = c.dsts[d].second;
    pr->dst[d].lo_off = t.lo_off();
    pr->dst[d].q_off = t.q_off();
  }
  if (c.to_lowres) {
    pr->out_nchw = cx.lowres;
    pr->out_C = e->heat_channels;
    pr->out_coff = c.lowres_coff;
  }
  pr->Cout = c.cout;
}

unsigned long long* g_clkprobe = nullptr;  // diagnostics (rtp_bench_dominant_conv under RTP_CLKPROBE)
int launch_conv_step(rtp_engine* e, Ctx& cx, const Step& s, int nimg) {
  const ConvOp& A = e->convs[s.a];
  const Geom& g = e->geom[A.level];
  ConvParams P;
  memset(&P, 0, sizeof P);
  fill_problem(e, cx, A, &P.prob[0]);
  if (s.b >= 0) fill_problem(e, cx, e->convs[s.b], &P.prob[1]);
  P.H = g.H; P.W = g.W; P.Wp = g.Wp; P.halo = g.halo; P.img_pix = g.img_pix;
  P.in_cstride = e->tensors[A.in_tensor].stride();
  P.nchunk = A.nchunk;
  P.wrap_at = A.wrap_at();
  P.last_phys = A.last_phys();
  {
    const Tensor& tin = e->tensors[A.in_tensor];
    const int qb = tin.q_off() * e->elem;  // byte offset of the input's q block inside a pixel
    P.q_from = A.h8 ? A.ncp : 0;
    P.jump_delta = A.h8 ? qb - (A.ncp - 1) * A.rowb : 0;
    P.row_back = A.h8 ? qb + (A.ncp - 1) * A.rowb : A.last_phys() * A.rowb;
    P.wq_exp = A.wq_exp;
  }
  P.CoutP = A.CoutP;
  const ConvCfgInfo ci = conv_cfg_info(A.cfg);
  P.tiles_per_img = (int)(((long)g.H * g.Wp + ci.BM - 1) / ci.BM);
  if (A.pool >= 0) {  // tiles of 2 image rows x BM/2 pixels, walked with an even pitch; the epilogue writes the next level's tensor
    const Geom& go = e->geom[A.level + 1];
    P.pool = 1;
    P.pool_wq = (g.W + A.k_eff / 2 + 1) & ~1;
    P.pool_Wp = go.Wp; P.pool_halo = go.halo; P.pool_img_pix = go.img_pix;
    P.tiles_per_img = (int)(((long)(g.H / 2) * P.pool_wq + ci.BM / 2 - 1) / (ci.BM / 2));
  }
  P.relu = A.relu ? 1 : 0;
  P.clkprobe = g_clkprobe;
  P.stamp = stamp_slot(e, cx, (int)(&s - e->steps.data()));
  P.nimg = nimg;
  {
    static const char* rot = RTP_EXP_ENV("RTP_CONV_ROTATE");
    P.rotate = (rot && rot[0] == '0') ? 0 : 1;
    static const char* xm = RTP_EXP_ENV("RTP_CONV_XCDMAP");
    P.xcdmap = xm ? atoi(xm) : 1;
    static const char* sb = RTP_EXP_ENV("RTP_RING_SB");
    P.ring_sb = sb ? atoi(sb) : 6;
    static const char* sp = RTP_EXP_ENV("RTP_RING_SPEC");
    P.spec = (sp && sp[0] == '0') ? 0 : 1;  // wave-specialised ring kernels (default); 0 = every wave does both
    static const char* rv = RTP_EXP_ENV("RTP_RING_VAR");
    P.variant = rv ? atoi(rv) : 0;
    static const char* ed = RTP_EXP_ENV("RTP_EPI_DIAG");   // experiments (timing only): 1 = epilogues do not store, 2 = no epilogue
    P.diag = ed ? atoi(ed) : 0;
    static const char* il = RTP_EXP_ENV("RTP_RING_ILV");
    // interleaved A-fragment rows (conv_ring.hip ILV; bit-identical): default for the fp8-compensated launches, whose plain variant
    // spills 6 registers (44.6 vs 45.3 us on the dominant shape); the plain fp16 launches are faster without.  "1" = all, "0" = none
    P.ilv = il ? (il[0] == '1' || (il[0] == 'q' && A.h8) || (il[0] == '7' && A.h8 && A.k_eff == 7)) : (A.h8 ? 1 : 0);
  }
  if (A.impl == 1) HIPCHK(e, launch_conv_ring(e->prec, A.cfg, A.k_eff, A.rowb, P, s.b >= 0 ? 2 : 1, nimg, cx.stream));
  else HIPCHK(e, launch_conv(e->prec, A.cfg, A.k_eff, A.rowb, P, s.b >= 0 ? 2 : 1, nimg, cx.stream));
  return RTP_OK;
}

int launch_pw2_step(rtp_engine* e, Ctx& cx, const Step& s, int nimg) {
  const ConvOp& A = e->convs[s.a];
  const ConvOp& C = e->convs[s.a2];
  const Geom& g = e->geom[C.level];
  Pw2Params Q;
  memset(&Q, 0, sizeof Q);
  ConvParams& P = Q.P2;
  fill_problem(e, cx, C, &P.prob[0]);
  if (s.b2 >= 0) fill_problem(e, cx, e->convs[s.b2], &P.prob[1]);
  P.H = g.H; P.W = g.W; P.Wp = g.Wp; P.halo = g.halo; P.img_pix = g.img_pix;
  P.CoutP = 64;
  P.tiles_per_img = (int)(((long)g.H * g.Wp + 63) / 64);
  P.relu = C.relu ? 1 : 0;
  P.stamp = stamp_slot(e, cx, (int)(&s - e->steps.data()));
  P.nimg = nimg;
  const int firsts[2] = {s.a, s.b};
  for (int q = 0; q < 2; ++q) {
    if (firsts[q] < 0) continue;
    const ConvOp& F = e->convs[firsts[q]];
    const Tensor& ti = e->tensors[F.in_tensor];
    const Tensor& tm = e->tensors[F.dsts[0].first];
    Q.x_in[q] = cx.arena + ti.offset;
    Q.x_cstride = ti.stride();
    Q.x_lo_off = F.split_a ? ti.lo_off() : 0;
    Q.w1[q] = e->dweights + F.w_off;
    Q.b1[q] = (const float*)(e->dweights + F.b_off);
    Q.mid[q].base = tm.written ? cx.arena + tm.offset : nullptr;   // (not materialised unless keep_blobs: the kernel skips the store)
    Q.mid[q].cstride = tm.stride();
    Q.mid[q].coff = 0;
    Q.mid[q].lo_off = tm.lo_off();
  }
  Q.c1_chunks = A.fused_chunks;
  Q.relu1 = A.relu ? 1 : 0;
  Q.split_w1 = A.split_w ? 1 : 0;
  Q.split_w2 = C.split_w ? 1 : 0;
  Q.h_lo = C.split_a ? 1 : 0;
  static const char* probe = RTP_EXP_ENV("RTP_PW2_PROBE");  // diagnostics (eager mode only): phase stamps of workgroup 0
  static int probed = 0;
  hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
  if (probe) (void)hipStreamIsCapturing(cx.stream, &cst);
  if (probe && cst == hipStreamCaptureStatusNone && probed < 16) {
    unsigned long long* d = nullptr;
    HIPCHK(e, hipMalloc((void**)&d, 32 * 8));
    HIPCHK(e, hipMemset(d, 0, 32 * 8));
    P.clkprobe = d;
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(e, launch_conv_pw2(Q, s.b >= 0 ? 2 : 1, nimg, cx.stream));
    HIPCHK(e, hipStreamSynchronize(cx.stream));
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long h[32];
    HIPCHK(e, hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    fprintf(stderr, "pw2 probe %s chunks %d splits %d%d%d%d (launch+sync %.1f us): ", A.name.c_str(), Q.c1_chunks, Q.x_lo_off ? 1 : 0, Q.split_w1, Q.h_lo, Q.split_w2, host_us);
    for (unsigned i = 2; i <= h[0] && i < 32; ++i) fprintf(stderr, "%.2f ", (double)(h[i] - h[i - 1]) / 100.0);
    fprintf(stderr, "us; total %.2f us\n", (double)(h[h[0]] - h[1]) / 100.0);
    ++probed;
    return RTP_OK;
  }
  HIPCHK(e, launch_conv_pw2(Q, s.b >= 0 ? 2 : 1, nimg, cx.stream));
  return RTP_OK;
}

}  // namespace

namespace {
bool is_dominant_class(const rtp_engine* e, const Step& s) {
  if (s.type != 1 || e->dominant_step < 0) return false;
  const ConvOp& a = e->convs[s.a];
  const ConvOp& d = e->convs[e->steps[e->dominant_step].a];
  // every launch of the dominant kernel SYMBOL (what a profiler aggregates), whatever its number of MFMA passes
  return a.k == d.k && a.cin == d.cin && a.cout == d.cout && (s.b >= 0) == (e->steps[e->dominant_step].b >= 0);
}

int launch_first_step(rtp_engine* e, Ctx& cx, const Step& s, const float* input_dev, int nimg) {
  const ConvOp& c = e->convs[s.a];
  const Tensor& to = e->tensors[c.dsts[0].first];
  FirstParams Q;
  Q.in = input_dev;
  Q.g = e->geom[0];
  Q.g.N = nimg;
  Q.wfrag = (const uint4*)(e->dweights + c.w_off);
  Q.bias = (const float*)(e->dweights + c.b_off);
  Q.out = (_Float16*)(cx.arena + to.offset);
  Q.Cp = to.stride();
  Q.relu = c.relu ? 1 : 0;
  Q.stamp = stamp_slot(e, cx, (int)(&s - e->steps.data()));
  HIPCHK(e, launch_conv_first(Q, cx.stream));
  return RTP_OK;
}

int run_frame_stack(rtp_engine* e, Ctx& cx, const float* input_dev, int nimg, bool cap = false) {
  const std::vector<PoolOp>& pools = e->pools;
  auto geom_n = [&](int level) { Geom g = e->geom[level]; g.N = nimg; return g; };
  for (size_t si = 0; si < e->steps.size(); ++si) {
    const Step& s = e->steps[si];
    // timing pass: an event pair on this stream around the launch (full batches only: the FLOP count reported is the full batch's);
    // the dominant class only (rtp_kernel_timing 1 / 2) or every step of the plan (3: bench.py's roofline.classes)
    const bool want_timed = e->time_dominant && !cap && nimg == e->NI && (e->time_all || is_dominant_class(e, s));
    const bool timed = want_timed && e->tev_next < (int)e->tev.size() / 2;
    if (want_timed && !timed) e->probe_dropped[0]++;
    if (timed) HIPCHK(e, hipEventRecord(e->tev[2 * (size_t)e->tev_next], cx.stream));
    struct Close {   // second event + bookkeeping on every way out of the step's branch
      rtp_engine* e; Ctx& cx; const Step& s; size_t si; bool timed;
      ~Close() {
        if (!timed) return;
        (void)hipEventRecord(e->tev[2 * (size_t)e->tev_next + 1], cx.stream);
        e->tev_step[e->tev_next] = (short)si;
        e->tev_pass[e->tev_next++] = (unsigned char)(s.type == 1 ? e->convs[s.a].passes() : 1);
      }
    } close_{e, cx, s, si, timed};
    if (s.type == 0) {
      const Tensor& t = e->tensors[0];
      HIPCHK(e, launch_pack_input(e->prec, input_dev, cx.arena + t.offset, geom_n(0), t.stride(), cx.stream));
    } else if (s.type == 1) {
      const int rc = launch_conv_step(e, cx, s, nimg);
      if (rc) return rc;
    } else if (s.type == 3) {
      const int rc = launch_pw2_step(e, cx, s, nimg);
      if (rc) return rc;
    } else if (s.type == 4) {
      const int rc = launch_first_step(e, cx, s, input_dev, nimg);
      if (rc) return rc;
    } else {
      const PoolOp& p = pools[s.a];
      const Tensor& ti = e->tensors[p.in_tensor];
      const Tensor& to = e->tensors[p.out_tensor];
      HIPCHK(e, launch_maxpool(e->prec, cx.arena + ti.offset, geom_n(ti.level), ti.stride(), cx.arena + to.offset, geom_n(to.level), to.stride(),
                               round_up(p.C, 16 / e->elem), ti.lo_off(), to.lo_off(), ti.q_off(), to.q_off(), cx.stream));
    }
  }
  return RTP_OK;
}

ResizeParams resize_params(rtp_engine* e, Ctx& cx, int sj) {
  Slot& sl = cx.slot[sj];
  ResizeParams rp;
  rp.src = cx.lowres + (size_t)sj * e->N * e->heat_channels * e->low_h * e->low_w;
  rp.dst = sl.resized; rp.num = e->N; rp.C = e->heat_channels;
  rp.h = e->low_h; rp.w = e->low_w; rp.tw = e->cfg.net_w; rp.th = e->cfg.net_h;
  rp.start_scale = e->start_scale; rp.scale_gap = e->scale_gap;
  return rp;
}
int run_resize(rtp_engine* e, Ctx& cx, int sj = 0) {
  HIPCHK(e, launch_resize(resize_params(e, cx, sj), cx.slot[sj].stream));
  return RTP_OK;
}
NmsParams nms_params(rtp_engine* e, Ctx& cx, int sj) {
  Slot& sl = cx.slot[sj];
  NmsParams np;
  np.src = sl.resized; np.peaks = sl.peaks; np.strip_count = sl.strip_count; np.strip_list = sl.strip_list;
  np.src_planes = e->heat_channels; np.H = e->cfg.net_h; np.W = e->cfg.net_w; np.num_parts = e->num_parts;
  np.max_peaks = e->max_peaks; np.nstrips = e->nstrips; np.strip_rows = e->strip_rows; np.threshold = e->nms_threshold;
  np.probe = nullptr;
  np.clear_flag = nullptr;
  np.stamp = stamp_slot(e, cx, 64 + 8 * sj);
  return np;
}
int run_nms(rtp_engine* e, Ctx& cx, int sj = 0) {
  HIPCHK(e, launch_nms(nms_params(e, cx, sj), cx.slot[sj].stream));
  return RTP_OK;
}
ConnectParams connect_params(rtp_engine* e, Ctx& cx, int sj) {
  Slot& sl = cx.slot[sj];
  ConnectParams cp;
  memset(&cp, 0, sizeof cp);
  cp.heat = sl.resized; cp.peaks = sl.peaks; cp.joints = sl.joints; cp.num_people = sl.num_people;
  cp.cand_score = sl.cand_score; cp.cand_ij = sl.cand_ij; cp.cand_count = sl.cand_count; cp.cand_blk = sl.cand_blk;
  cp.conn = sl.conn; cp.conn_score = sl.conn_score; cp.conn_count = sl.conn_count;
  cp.max_rows = e->max_rows; cp.model = e->model; cp.num_parts = e->num_parts; cp.num_limbs = e->num_limbs;
  cp.max_peaks = e->max_peaks; cp.net_w = e->cfg.net_w; cp.net_h = e->cfg.net_h; cp.disp_w = e->cfg.disp_w; cp.disp_h = e->cfg.disp_h;
  cp.inter_threshold = e->inter_threshold; cp.inter_min_above = e->inter_min_above; cp.min_subset_cnt = e->min_subset_cnt;
  cp.min_subset_score = e->min_subset_score; cp.max_people = RTP_MAX_PEOPLE;
  cp.stamp = stamp_slot(e, cx, 64 + 8 * sj + 2);
  {
    static const char* pf = RTP_EXP_ENV("RTP_PAIRS_FULL");
    cp.pairs_full = (pf && pf[0] == '1') ? 1 : 0;
  }
  return cp;
}
int run_connect(rtp_engine* e, Ctx& cx, int sj = 0) {
  HIPCHK(e, launch_connect(connect_params(e, cx, sj), cx.slot[sj].stream));
  return RTP_OK;
}
// production post-processing: peaks and PAF samples straight from the low-res maps; the 55 MB
// resized map is never written (the taps / rtp_forward_debug still materialise it)
int run_post_fused(rtp_engine* e, Ctx& cx, int sj, hipEvent_t ev_nms) {
  Slot& sl = cx.slot[sj];
  const ResizeParams rp = resize_params(e, cx, sj);
  NmsParams np = nms_params(e, cx, sj);
  np.clear_flag = sl.num_people;          // (one launch less in the chain: the 4-byte fill in front of the connect kernels)
  HIPCHK(e, launch_nms_fused(np, rp, sl.stream));
  HIPCHK(e, hipEventRecord(ev_nms, sl.stream));
  ConnectParams cp = connect_params(e, cx, sj);
  cp.counter_cleared = 1;
  cp.tickets = e->chain_connect ? sl.tickets : nullptr;
  HIPCHK(e, launch_connect_fused(cp, rp, sl.stream));
  return RTP_OK;
}

// one batch on one context: conv stack over nframes*num_scales images, then per frame (on the
// frame slot's stream) resize -> nms -> connect -> D2H of the joints
int launch_batch_body(rtp_engine* e, Ctx& cx, int nframes, const float* input_dev, bool materialize, bool cap, int part = 3) {
  int rc;
  if (part & 1) {
    HIPCHK(e, hipEventRecord(cx.ev[0], cx.stream));
    if ((rc = run_frame_stack(e, cx, input_dev, nframes * e->N, cap))) return rc;
  }
  if (!(part & 2)) return RTP_OK;
  HIPCHK(e, hipEventRecord(cx.ev[1], cx.stream));
  const size_t jbytes = (size_t)RTP_MAX_PEOPLE * e->num_parts * 3 * sizeof(float);
  for (int j = 0; j < nframes; ++j) {
    Slot& sl = cx.slot[j];
    if (sl.stream != cx.stream) HIPCHK(e, hipStreamWaitEvent(sl.stream, cx.ev[1], 0));
    static const char* diag = RTP_EXP_ENV("RTP_DIAG_SKIP_POST");  // diagnosis only: 1 = no connect, 2 = no post-processing at all (both through the
    const int skip = diag ? atoi(diag) : 0;                  // materialised map); 3..6 = production kernels: 3 nms only, 4 + pairs, 5 + match, 6 all
    static const char* unf = RTP_EXP_ENV("RTP_POST_UNFUSED");  // experiments: production path through the materialised map
    HIPCHK(e, hipEventRecord(sl.ev[0], sl.stream));
    if (skip >= 3) {
      const ResizeParams rp = resize_params(e, cx, j);
      HIPCHK(e, hipEventRecord(sl.ev[1], sl.stream));
      HIPCHK(e, launch_nms_fused(nms_params(e, cx, j), rp, sl.stream));
      HIPCHK(e, hipEventRecord(sl.ev[2], sl.stream));
      if (skip >= 4) {
        ConnectParams cp = connect_params(e, cx, j);
        cp.diag_stages = skip - 3;   // 1 pairs, 2 + match, 3 + assemble
        HIPCHK(e, launch_connect_fused(cp, rp, sl.stream));
      }
    } else if (materialize || skip || (unf && unf[0] == '1')) {
      if (skip < 2 && (rc = run_resize(e, cx, j))) return rc;
      HIPCHK(e, hipEventRecord(sl.ev[1], sl.stream));
      if (skip < 2 && (rc = run_nms(e, cx, j))) return rc;
      HIPCHK(e, hipEventRecord(sl.ev[2], sl.stream));
      if (skip < 1 && (rc = run_connect(e, cx, j))) return rc;
    } else {
      HIPCHK(e, hipEventRecord(sl.ev[1], sl.stream));  // no resize stage
      if ((rc = run_post_fused(e, cx, j, sl.ev[2]))) return rc;
    }
    HIPCHK(e, hipEventRecord(sl.ev[3], sl.stream));
    if (e->cfg.render && sl.has_disp) {  // pose overlay on the display image (renderFunctions.cu, part_to_show == 0)
      const size_t dbytes = (size_t)e->cfg.disp_w * e->cfg.disp_h * 3;
      if (!sl.render_dev) {
        SYNC_GUARD;
        HIPCHK(e, hipMalloc((void**)&sl.render_dev, dbytes));
        HIPCHK(e, hipHostMalloc((void**)&sl.render_host, dbytes, hipHostMallocDefault));
        HIPCHK(e, hipMalloc((void**)&sl.render_tab, render_tab_floats(RTP_MAX_PEOPLE) * sizeof(float)));
      }
      if (e->cfg.render == 1) {
        RenderParams rp;
        rp.src = sl.disp_cur; rp.dst = sl.render_dev; rp.w = e->cfg.disp_w; rp.h = e->cfg.disp_h;
        rp.poses = sl.joints; rp.num_people = sl.num_people; rp.tab = sl.render_tab;
        rp.model = e->model; rp.googly = 0; rp.max_people = RTP_MAX_PEOPLE;
        HIPCHK(e, launch_render(rp, sl.stream));
      } else {  // --part_to_show view: needs the frame's net-resolution maps, which the production path never builds
        if (!(materialize || skip || (unf && unf[0] == '1')) && (rc = run_resize(e, cx, j))) return rc;
        RenderViewParams vp;
        vp.src = sl.disp_cur; vp.dst = sl.render_dev; vp.w = e->cfg.disp_w; vp.h = e->cfg.disp_h;
        vp.maps = sl.resized; vp.net_w = e->cfg.net_w; vp.net_h = e->cfg.net_h;
        vp.model = e->model; vp.part_to_show = e->cfg.render - 1;
        HIPCHK(e, launch_render_view(vp, sl.stream));
      }
      HIPCHK(e, hipMemcpyAsync(sl.render_host, sl.render_dev, dbytes, hipMemcpyDeviceToHost, sl.stream));
    }
    HIPCHK(e, hipMemcpyAsync(sl.host_out + 4, sl.joints, jbytes, hipMemcpyDeviceToHost, sl.stream));
    HIPCHK(e, hipMemcpyAsync(sl.host_out, sl.num_people, sizeof(int), hipMemcpyDeviceToHost, sl.stream));
    HIPCHK(e, hipEventRecord(sl.ev[4], sl.stream));
    if (cap && sl.stream != cx.stream) HIPCHK(e, hipStreamWaitEvent(cx.stream, sl.ev[4], 0));  // join the capture
  }
  return RTP_OK;
}

// Static launch plan as a hipGraph: captured the first time a batch of `nframes` frames is launched on
// this context (stream capture of exactly the eager sequence; the frame slots' streams fork from and
// join the context's stream), then ONE hipGraphLaunch per batch.  Replaces net.cpp:544-556.
int capture_batch(rtp_engine* e, Ctx& cx, int nframes, hipGraphExec_t* out) {
  SYNC_GUARD;
  hipGraph_t g = nullptr;
  HIPCHK(e, hipStreamBeginCapture(cx.stream, hipStreamCaptureModeThreadLocal));
  const int rc = launch_batch_body(e, cx, nframes, cx.input, false, true, e->graph_post ? 3 : 1);
  const hipError_t s = hipStreamEndCapture(cx.stream, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (s != hipSuccess || !g) return fail(e, RTP_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(s));
  const hipError_t si = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (si != hipSuccess) return fail(e, RTP_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(si));
  return RTP_OK;
}

int flush_prep(rtp_engine* e, Ctx& cx, bool force);
int launch_batch(rtp_engine* e, Ctx& cx, int nframes, const float* input_dev, bool materialize = false) {
  int rc;
  if (e->prep_defer && (rc = flush_prep(e, cx, true))) return rc;
  if (cx.in_pending) {   // the batch's inputs were staged on the staging stream: the conv stream starts when the last of them is complete
    if (cx.in_stream != cx.stream) {
      HIPCHK(e, hipEventRecord(cx.ev_in, cx.in_stream));
      HIPCHK(e, hipStreamWaitEvent(cx.stream, cx.ev_in, 0));
    }
    cx.in_pending = false;
  }
  static const char* diag = RTP_EXP_ENV("RTP_DIAG_SKIP_POST");
  static const char* unf = RTP_EXP_ENV("RTP_POST_UNFUSED");
  // (a timing pass launches eagerly: its event pairs sit between the launches)
  const bool graph = e->use_graph && !e->time_dominant && !materialize && input_dev == cx.input && !e->cfg.render && !diag && !unf && nframes <= 16;
  cx.graph_run = graph;
  cx.graph_conv = false;
  if (!graph) {
    if ((rc = launch_batch_body(e, cx, nframes, input_dev, materialize, false))) return rc;
    cx.launched = true;
    return RTP_OK;
  }
  if (!cx.gexec[nframes] && (rc = capture_batch(e, cx, nframes, &cx.gexec[nframes]))) return rc;
  HIPCHK(e, hipEventRecord(cx.gev[0], cx.stream));
  HIPCHK(e, hipGraphLaunch(cx.gexec[nframes], cx.stream));
  if (!e->graph_post) {  // the conv stack is one replay; every frame's short post-processing chain is launched eagerly on its slot's stream
    if ((rc = launch_batch_body(e, cx, nframes, cx.input, false, false, 2))) return rc;
    cx.graph_run = false;            // collect waits for the frame's own event, not for the whole batch
    cx.graph_conv = true;
  }
  HIPCHK(e, hipEventRecord(cx.gev[1], cx.stream));
  cx.launched = true;
  return RTP_OK;
}
int enqueue_frame(rtp_engine* e, Ctx& cx, const float* input_dev, bool materialize = false) { return launch_batch(e, cx, 1, input_dev, materialize); }

// Streams of the batch contexts (post = false) and of the per-frame post-processing chains (post = true).
// Experiments build only: RTP_CONV_PRIO / RTP_POST_PRIO = a stream priority (hipDeviceGetStreamPriorityRange: lower number = higher
// priority); RTP_POST_CUS = n: the post-processing streams may only use n CUs (hipExtStreamCreateWithCUMask; the KFD interleaves the
// mask bits over the XCDs, so the low 8 bits are one CU in each of the 8 XCDs).  Both measured worse than plain streams.
// ---- hardware queues ------------------------------------------------------------------------------------------------------------------
// A hardware queue runs the kernels of ITS streams one after the other, and the HIP runtime attaches every new stream to the queue with the
// fewest streams (tools/hwq_probe.hip: the first Q streams take queues 0..Q-1, later ones Q-1, Q-2, .., 0, Q-1, ..; Q = GPU_MAX_HW_QUEUES,
// default 4, read once at the process's first HIP call).  Rounds 1-5 gave every frame of a batch beyond the first a post-processing stream of
// its own; with batches of 2 and five contexts that is ten streams on four queues, and every conv stack shares its queue with another context's
// stack or chain (head-of-line blocking: a kernel waits behind a kernel of ANOTHER batch although its own inputs are ready).
// Round 6: when the queues suffice (contexts <= Q), a context has ONE stream — staging, conv stack and the chains of all its frames — and
// therefore one hardware queue to itself: COCO batches of 2, seven in flight 1071 -> 1203 frames/s on 8 queues (+12 %), MPI batches of 5
// 1328 -> 1570-1600 (three contexts: the default 4 queues suffice); with too few queues the old arrangement is the better one (1030 against
// 1068 on 4 queues at batches of 2) and stays.  The host programs raise the count (bench.py --hw_queues, rtpose.bin: GPU_MAX_HW_QUEUES=8 before
// their first HIP call unless the environment has it; a library cannot: the runtime reads it once).  profiles/r06_experiments.txt section 6.
int hw_queue_count() {
  const char* v = getenv("GPU_MAX_HW_QUEUES");
  const int q = v ? atoi(v) : 4;
  return q >= 1 && q <= 64 ? q : 4;
}
hipError_t make_stream(hipStream_t* s, bool post) {
#ifdef RTP_EXPERIMENTS
  if (post) {
    const char* c = getenv("RTP_POST_CUS");
    const int n = c ? atoi(c) : 0;
    if (n > 0 && n < 256) {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int b = 0; b < n; ++b) mask[b >> 5] |= 1u << (b & 31);
      return hipExtStreamCreateWithCUMask(s, 8, mask);
    }
  }
  const char* v = getenv(post ? "RTP_POST_PRIO" : "RTP_CONV_PRIO");
  if (v && v[0]) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    int pr = atoi(v);
    pr = std::max(greatest, std::min(least, pr));
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, pr);
  }
#endif
  (void)post;
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

int alloc_slot(rtp_engine* e, Ctx& cx, Slot& sl, bool share_stream) {
  if (sl.preset_stream) {}  // RTP_STREAM_PLAN: alloc_ctx chose it
  else if (share_stream) sl.stream = cx.stream;
  else { HIPCHK(e, make_stream(&sl.stream, true)); sl.own_stream = true; }
  const size_t res_floats = (size_t)e->heat_channels * e->cfg.net_h * e->cfg.net_w;
  HIPCHK(e, hipMalloc((void**)&sl.resized, res_floats * sizeof(float)));
  const size_t peak_floats = (size_t)e->num_parts * (e->max_peaks + 1) * 3;
  HIPCHK(e, hipMalloc((void**)&sl.peaks, peak_floats * sizeof(float)));
  HIPCHK(e, hipMemset(sl.peaks, 0, peak_floats * sizeof(float)));
  HIPCHK(e, hipMalloc((void**)&sl.strip_count, (size_t)e->num_parts * e->nstrips * sizeof(int)));
  HIPCHK(e, hipMalloc((void**)&sl.strip_list, (size_t)e->num_parts * e->nstrips * e->max_peaks * sizeof(int)));
  const size_t pairs = (size_t)e->num_limbs * e->max_peaks * e->max_peaks;
  HIPCHK(e, hipMalloc((void**)&sl.cand_score, pairs * sizeof(float)));
  HIPCHK(e, hipMalloc((void**)&sl.cand_ij, pairs * sizeof(int)));
  HIPCHK(e, hipMalloc((void**)&sl.cand_count, e->num_limbs * sizeof(int)));
  HIPCHK(e, hipMalloc((void**)&sl.cand_blk, (size_t)e->num_limbs * ((e->max_peaks * e->max_peaks + 255) / 256) * sizeof(int)));
  HIPCHK(e, hipMalloc((void**)&sl.conn, (size_t)e->num_limbs * e->max_peaks * 2 * sizeof(int)));
  HIPCHK(e, hipMalloc((void**)&sl.conn_score, (size_t)e->num_limbs * e->max_peaks * sizeof(float)));
  HIPCHK(e, hipMalloc((void**)&sl.conn_count, e->num_limbs * sizeof(int)));
  HIPCHK(e, hipMalloc((void**)&sl.tickets, (e->num_limbs + 1) * sizeof(int)));
  HIPCHK(e, hipMemset(sl.tickets, 0, (e->num_limbs + 1) * sizeof(int)));
  const size_t jfloats = (size_t)RTP_MAX_PEOPLE * e->num_parts * 3;
  HIPCHK(e, hipMalloc((void**)&sl.joints, jfloats * sizeof(float)));
  HIPCHK(e, hipMemset(sl.joints, 0, jfloats * sizeof(float)));
  HIPCHK(e, hipMalloc((void**)&sl.num_people, sizeof(int)));
  HIPCHK(e, hipHostMalloc((void**)&sl.host_out, (jfloats + 4) * sizeof(float), hipHostMallocDefault));
  for (int i = 0; i < 5; ++i) HIPCHK(e, hipEventCreateWithFlags(&sl.ev[i], event_flags(i == 4)));
  // Staging buffers of rtp_submit_frame for frames up to the display size (a video at --resolution, BASELINE configs[1]) and the renderer's
  // buffers are allocated HERE, under the creation lock: the per-frame path then never allocates (and never takes g_sync_mutex) unless a
  // frame is larger than the display image.
  {
    const size_t fbytes = (size_t)e->cfg.disp_w * e->cfg.disp_h * 3;
    HIPCHK(e, hipMalloc((void**)&sl.frame_dev, fbytes));
    HIPCHK(e, hipHostMalloc((void**)&sl.frame_host, fbytes, hipHostMallocDefault));
    sl.frame_cap = fbytes;
    HIPCHK(e, hipMalloc((void**)&sl.disp_dev, fbytes));
    if (e->cfg.render) {
      HIPCHK(e, hipMalloc((void**)&sl.render_dev, fbytes));
      HIPCHK(e, hipHostMalloc((void**)&sl.render_host, fbytes, hipHostMallocDefault));
      HIPCHK(e, hipMalloc((void**)&sl.render_tab, render_tab_floats(RTP_MAX_PEOPLE) * sizeof(float)));
    }
  }
  return RTP_OK;
}

int alloc_ctx(rtp_engine* e, Ctx& cx) {
  // RTP_STREAM_PLAN=1 (experiment): the runtime hands its 4 hardware queues to streams round-robin in creation order.  Create four
  // streams per context so that conv stacks alternate between queues 0 / 1 and BOTH post-processing chains of a batch sit on queues
  // 2 / 3: no post chain ever stands in front of another context's conv stack.  (Default: conv stream, then the extra slots' streams:
  // with batches of 2 the conv stacks share queues 0 / 2 with frame 0's chains, frame 1's chains use queues 1 / 3.)
  static const char* plan = RTP_EXP_ENV("RTP_STREAM_PLAN");
  const bool planned = plan && plan[0] == '1' && e->B == 2;
  std::vector<hipStream_t> planned_streams;
  if (planned) {
    const int ci = (int)(&cx - &e->ctx[0]);
    hipStream_t q[4];
    for (int i = 0; i < 4; ++i) HIPCHK(e, hipStreamCreateWithFlags(&q[i], hipStreamNonBlocking));
    cx.stream = q[ci & 1];
    cx.spare_stream = q[(ci & 1) ^ 1];
    planned_streams = {q[2], q[3]};
  } else
#ifdef RTP_EXPERIMENTS
  if (const char* cm = getenv("RTP_CTX_CUMASK")) {   // experiments: 2 = contexts alternate between two halves of the chip (16 CUs of every XCD each), 3 = three thirds
    const int parts = atoi(cm), ci = (int)(&cx - &e->ctx[0]);
    if (parts >= 2 && parts <= 4) {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const int per = 32 / parts, lo = (ci % parts) * per;           // CU index inside an XCD = bit / 8 (the KFD interleaves the mask bits over the XCDs)
      for (int b = 0; b < 256; ++b) if ((b >> 3) >= lo && (b >> 3) < lo + per) mask[b >> 5] |= 1u << (b & 31);
      HIPCHK(e, hipExtStreamCreateWithCUMask(&cx.stream, 8, mask));
    } else HIPCHK(e, make_stream(&cx.stream, false));
  } else
#endif
  HIPCHK(e, make_stream(&cx.stream, false));
  HIPCHK(e, hipMalloc((void**)&cx.arena, e->arena_bytes));
  HIPCHK(e, hipMemset(cx.arena, 0, e->arena_bytes));
  const size_t in_floats = (size_t)e->NI * 3 * e->cfg.net_h * e->cfg.net_w;
  HIPCHK(e, hipMalloc((void**)&cx.input, in_floats * sizeof(float)));
  HIPCHK(e, hipHostMalloc((void**)&cx.host_in, in_floats * sizeof(float), hipHostMallocDefault));
  const size_t low_floats = (size_t)e->NI * e->heat_channels * e->low_h * e->low_w;
  HIPCHK(e, hipMalloc((void**)&cx.lowres, low_floats * sizeof(float)));
  HIPCHK(e, hipMemset(cx.lowres, 0, low_floats * sizeof(float)));
  for (int i = 0; i < 2; ++i) HIPCHK(e, hipEventCreateWithFlags(&cx.ev[i], event_flags(i == 1)));
  for (int i = 0; i < 2; ++i) HIPCHK(e, hipEventCreateWithFlags(&cx.gev[i], event_flags(i == 1)));
  HIPCHK(e, hipMalloc((void**)&cx.stamps, 2 * STAMP_SLOTS * sizeof(unsigned long long)));
  HIPCHK(e, hipMemset(cx.stamps, 0, 2 * STAMP_SLOTS * sizeof(unsigned long long)));
  cx.slot.resize(e->B);
  for (int j = 0; j < e->B; ++j) {
    int rc;
    static const char* own0 = RTP_EXP_ENV("RTP_POST_OWN0");  // experiments: 1 = frame 0's post-processing chain also gets its own stream
    if (planned) {
      cx.slot[j].stream = planned_streams[j];
      cx.slot[j].own_stream = true;
      cx.slot[j].preset_stream = true;
    }
    static const char* shall = RTP_EXP_ENV("RTP_POST_SHARE_ALL");  // experiments: 1 / 0 = every frame's chain on the batch's conv stream (one stream per context) / round 5's streams
    const bool one_stream = shall ? shall[0] == '1' : (int)e->ctx.size() <= hw_queue_count();   // one hardware queue per context when the queues suffice (see "hardware queues" above)
    if (!one_stream && !shall && j == 1) {   // once per process: the setting a host program can make and a library cannot
      static std::atomic<bool> hinted{false};
      if (!hinted.exchange(true))
        fprintf(stderr, "rtpose-mi355x: %zu batch contexts on %d HIP hardware queues: with GPU_MAX_HW_QUEUES >= %zu in the environment before the process's first "
                        "HIP call every context gets a queue to itself (+10 %% frames/s at batches of 2; INTEGRATION.md)\n", e->ctx.size(), hw_queue_count(), e->ctx.size());
    }
    if ((rc = alloc_slot(e, cx, cx.slot[j], (j == 0 && !(own0 && own0[0] == '1')) || one_stream))) return rc;
  }
  {
    // HIP deals its hardware queues to streams round-robin in creation order (4 queues).  A context owns batch_frames streams (its conv
    // stream, which also carries frame 0's chain, + one per further frame): with batch_frames = 4 (8, ..) EVERY context's conv stream would
    // land on the same hardware queue and the conv stacks of different batches would run strictly one after the other — measured
    // (profiles/r05_experiments.txt): batches of 4 pipelined no faster than one batch at a time.  Two unused streams per context restore the
    // arrangement batches of 2 have by construction: conv stacks alternate between two queues.
    static const char* pad_env = RTP_EXP_ENV("RTP_STREAM_PAD");
    const int pad = pad_env ? atoi(pad_env) : ((e->B % 4) == 0 && !planned ? 2 : 0);
    for (int i = 0; i < pad; ++i) {
      hipStream_t d = nullptr;
      HIPCHK(e, hipStreamCreateWithFlags(&d, hipStreamNonBlocking));
      cx.pad_streams.push_back(d);
    }
  }
  {
    static const char* sh1 = RTP_EXP_ENV("RTP_POST_SHARE1");  // experiments: 1 = frame 0's chain runs on frame 1's stream (behind nothing of the conv queues)
    if (sh1 && sh1[0] == '1' && e->B >= 2 && !planned && cx.slot[0].stream == cx.stream) cx.slot[0].stream = cx.slot[1].stream;
  }
  return RTP_OK;
}

void free_ctx(Ctx& cx) {
  if (cx.stream) (void)hipStreamSynchronize(cx.stream);
  if (cx.in_stream && cx.in_stream != cx.stream) { (void)hipStreamSynchronize(cx.in_stream); (void)hipStreamDestroy(cx.in_stream); }
  if (cx.ev_in) (void)hipEventDestroy(cx.ev_in);
  for (Slot& sl : cx.slot) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    void* dptrs[] = {sl.resized, sl.peaks, sl.strip_count, sl.strip_list, sl.cand_score, sl.cand_ij, sl.cand_count, sl.cand_blk, sl.conn, sl.conn_score,
                     sl.conn_count, sl.tickets, sl.joints, sl.num_people, sl.frame_dev, sl.disp_dev, sl.render_dev, sl.render_tab};
    for (void* p : dptrs) if (p) (void)hipFree(p);
    if (sl.frame_host) (void)hipHostFree(sl.frame_host);
    if (sl.render_host) (void)hipHostFree(sl.render_host);
    if (sl.host_out) (void)hipHostFree(sl.host_out);
    for (int i = 0; i < 5; ++i) if (sl.ev[i]) (void)hipEventDestroy(sl.ev[i]);
    if (sl.ev_copy) (void)hipEventDestroy(sl.ev_copy);
    if (sl.own_stream && sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  for (hipGraphExec_t& g : cx.gexec) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
  void* dptrs[] = {cx.arena, cx.input, cx.lowres, cx.stamps};
  for (void* p : dptrs) if (p) (void)hipFree(p);
  if (cx.host_in) (void)hipHostFree(cx.host_in);
  for (int i = 0; i < 2; ++i) if (cx.ev[i]) (void)hipEventDestroy(cx.ev[i]);
  for (int i = 0; i < 2; ++i) if (cx.gev[i]) (void)hipEventDestroy(cx.gev[i]);
  if (cx.ev_stage0) { (void)hipEventDestroy(cx.ev_stage0); cx.ev_stage0 = nullptr; }
  if (cx.stream) (void)hipStreamDestroy(cx.stream);
  if (cx.spare_stream) (void)hipStreamDestroy(cx.spare_stream);
  for (hipStream_t d : cx.pad_streams) if (d) (void)hipStreamDestroy(d);
  cx = Ctx();
}

int use_device(rtp_engine* e) {
  if (e->broken) return fail(e, RTP_EHIP, "this engine lost its plan in a failed re-plan and can only be destroyed");
  HIPCHK(e, hipSetDevice(e->cfg.device_id));
  return RTP_OK;
}

// A captured batch graph bakes every by-value kernel argument (ConvParams::wq_exp of the fp8-compensated layers; with
// RTP_GRAPH_POST=1 also the thresholds and scales of the post-processing chains).  Whoever changes one of those drops the
// graphs; launch_batch re-captures on the next batch.  Needs an idle engine.
int invalidate_graphs(rtp_engine* e) {
  SYNC_GUARD;
  for (Ctx& cx : e->ctx) {
    if (cx.stream) HIPCHK(e, hipStreamSynchronize(cx.stream));
    for (hipGraphExec_t& g : cx.gexec) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
  }
  return RTP_OK;
}

// INTER_AREA tables of every pyramid level (display resolution -> 16*ceil(net*s/16)) + cubic table
int build_prep_tables(rtp_engine* e) {
  SYNC_GUARD;
  e->gpu_prep_ok = false;
  struct Host { int tw, th, identity, fx = 0, fy = 0, linear = 0; std::vector<int> xs, xsi, ys, ysi, lx, ly; std::vector<float> xa, ya; };
  std::vector<Host> hs(e->N);
  size_t bytes = 0;
  for (int i = 0; i < e->N; ++i) {
    const float scale = (float)((double)e->start_scale - i * (double)e->scale_gap);
    Host& h = hs[i];
    h.tw = (int)(16 * std::ceil(e->cfg.net_w * scale / 16));
    h.th = (int)(16 * std::ceil(e->cfg.net_h * scale / 16));
    h.identity = (h.tw == e->cfg.disp_w && h.th == e->cfg.disp_h) ? 1 : 0;
    if (h.tw > e->cfg.disp_w || h.th > e->cfg.disp_h) {   // an axis is enlarged (--resolution smaller than the level): cv::resize's bilinear kernel with area-mode coefficients
      h.linear = 1;
      rtp_internal_linear_area_table(e->cfg.disp_w, h.tw, &h.lx);
      rtp_internal_linear_area_table(e->cfg.disp_h, h.th, &h.ly);
    } else if (!h.identity && rtp_internal_area_fast(e->cfg.disp_w, e->cfg.disp_h, h.tw, h.th, &h.fx, &h.fy)) {
      // integer scale on both axes: block sums, no tables
    } else if (!h.identity) {
      h.fx = h.fy = 0;
      if (rtp_internal_area_table(e->cfg.disp_w, h.tw, &h.xs, &h.xsi, &h.xa)) return RTP_OK;
      if (rtp_internal_area_table(e->cfg.disp_h, h.th, &h.ys, &h.ysi, &h.ya)) return RTP_OK;
    }
    bytes += 256 * 8 + (h.xs.size() + h.xsi.size() + h.ys.size() + h.ysi.size() + h.lx.size() + h.ly.size()) * sizeof(int) + (h.xa.size() + h.ya.size()) * sizeof(float);
  }
  bytes += 32 * 32 * 16 * sizeof(short) + 256;
  std::vector<unsigned char> blob(bytes + 256, 0);
  if (e->prep_tables) { (void)hipFree(e->prep_tables); e->prep_tables = nullptr; }
  HIPCHK(e, hipMalloc((void**)&e->prep_tables, blob.size()));
  size_t off = 0;
  auto put = [&](const void* p, size_t n) { off = round_up_sz(off, 256); const size_t o = off; if (n) memcpy(blob.data() + o, p, n); off += n; return o; };
  e->area_scales.assign(e->N, AreaScale{});
  for (int i = 0; i < e->N; ++i) {
    Host& h = hs[i];
    AreaScale& a = e->area_scales[i];
    a.tw = h.tw; a.th = h.th; a.identity = h.identity;
    a.fast_x = h.identity ? 0 : h.fx; a.fast_y = h.identity ? 0 : h.fy;
    a.xstart = (const int*)(e->prep_tables + put(h.xs.data(), h.xs.size() * sizeof(int)));
    a.xsi = (const int*)(e->prep_tables + put(h.xsi.data(), h.xsi.size() * sizeof(int)));
    a.xalpha = (const float*)(e->prep_tables + put(h.xa.data(), h.xa.size() * sizeof(float)));
    a.ystart = (const int*)(e->prep_tables + put(h.ys.data(), h.ys.size() * sizeof(int)));
    a.ysi = (const int*)(e->prep_tables + put(h.ysi.data(), h.ysi.size() * sizeof(int)));
    a.yalpha = (const float*)(e->prep_tables + put(h.ya.data(), h.ya.size() * sizeof(float)));
    a.linear = h.linear;
    a.lx = (const int*)(e->prep_tables + put(h.lx.data(), h.lx.size() * sizeof(int)));
    a.ly = (const int*)(e->prep_tables + put(h.ly.data(), h.ly.size() * sizeof(int)));
  }
  {
    std::vector<short> t2(32 * 32 * 16);
    rtp_internal_cubic_tab2d(t2.data());
    e->warp_tab_dev = (const short*)(e->prep_tables + put(t2.data(), t2.size() * sizeof(short)));
  }
  HIPCHK(e, hipMemcpy(e->prep_tables, blob.data(), blob.size(), hipMemcpyHostToDevice));
  e->gpu_prep_ok = true;
  return RTP_OK;
}

// raw u8 BGR frame (host) -> frame slot sj of cx.input on the device
int enqueue_preprocess(rtp_engine* e, Ctx& cx, int sj, const unsigned char* bgr, int w, int h, float* frame_scale) {
  Slot& sl = cx.slot[sj];
  const size_t fbytes = (size_t)w * h * 3;
  if (fbytes > sl.frame_cap) {   // first frame of this slot (or a larger one): rare, and hipFree synchronises the device
    SYNC_GUARD;
    HIPCHK(e, hipStreamSynchronize(cx.in_stream));
    if (sl.frame_dev) (void)hipFree(sl.frame_dev);
    if (sl.frame_host) (void)hipHostFree(sl.frame_host);
    sl.frame_dev = nullptr; sl.frame_host = nullptr; sl.frame_cap = 0;
    HIPCHK(e, hipMalloc((void**)&sl.frame_dev, fbytes));
    HIPCHK(e, hipHostMalloc((void**)&sl.frame_host, fbytes, hipHostMallocDefault));
    sl.frame_cap = fbytes;
  }
  if (!sl.disp_dev) { SYNC_GUARD; HIPCHK(e, hipMalloc((void**)&sl.disp_dev, (size_t)e->cfg.disp_w * e->cfg.disp_h * 3)); }
  const double s = rtp_display_fit_scale(w, h, e->cfg.disp_w, e->cfg.disp_h);
  if (frame_scale) *frame_scale = (float)s;
  memcpy(sl.frame_host, bgr, fbytes);
  if (e->prep_defer) {   // copy now, kernels when the copy is done (flush_prep)
    if (!sl.ev_copy) HIPCHK(e, hipEventCreateWithFlags(&sl.ev_copy, hipEventDisableTiming));
    HIPCHK(e, hipMemcpyAsync(sl.frame_dev, sl.frame_host, fbytes, hipMemcpyHostToDevice, e->copy_stream));
    HIPCHK(e, hipEventRecord(sl.ev_copy, e->copy_stream));
    sl.copy_pending = true;
    sl.pend_w = w; sl.pend_h = h;
    return RTP_OK;
  }
  float* dst = cx.input + (size_t)sj * e->N * 3 * e->cfg.net_h * e->cfg.net_w;
  HIPCHK(e, hipMemcpyAsync(sl.frame_dev, sl.frame_host, fbytes, hipMemcpyHostToDevice, cx.in_stream));
  if (w == e->cfg.disp_w && h == e->cfg.disp_h) sl.disp_cur = sl.frame_dev;   // identity warp (20 us per 720p frame saved)
  else {
    sl.disp_cur = sl.disp_dev;
    HIPCHK(e, launch_warp(stamp_slot(e, cx, 200 + 2 * sj), sl.frame_dev, w, h, rtp_internal_warp_inverse_scale(s), e->warp_tab_dev, sl.disp_dev, e->cfg.disp_w, e->cfg.disp_h, cx.in_stream));
  }
  HIPCHK(e, launch_area_pad(stamp_slot(e, cx, 200 + 2 * sj + 1), sl.disp_cur, e->cfg.disp_w, e->cfg.disp_h, e->area_scales.data(), e->N, dst, e->cfg.net_w, e->cfg.net_h, cx.in_stream));
  cx.in_pending = true;
  return RTP_OK;
}

// Deferred pre-processing: enqueue the warp / area kernels of every frame of `cx` whose H2D copy has completed (force: of every
// pending frame, behind a stream wait on its copy — the batch is about to be launched).
int flush_prep(rtp_engine* e, Ctx& cx, bool force) {
  for (size_t sj = 0; sj < cx.slot.size(); ++sj) {
    Slot& sl = cx.slot[sj];
    if (!sl.copy_pending) continue;
    if (force) HIPCHK(e, hipStreamWaitEvent(cx.stream, sl.ev_copy, 0));
    else if (hipEventQuery(sl.ev_copy) != hipSuccess) continue;
    if (sl.pend_w == 0) { sl.copy_pending = false; continue; }   // rtp_submit: the copy WAS the staging (net input already pre-processed)
    const double s = rtp_display_fit_scale(sl.pend_w, sl.pend_h, e->cfg.disp_w, e->cfg.disp_h);
    float* dst = cx.input + sj * (size_t)e->N * 3 * e->cfg.net_h * e->cfg.net_w;
    if (sl.pend_w == e->cfg.disp_w && sl.pend_h == e->cfg.disp_h) sl.disp_cur = sl.frame_dev;
    else {
      sl.disp_cur = sl.disp_dev;
      HIPCHK(e, launch_warp(stamp_slot(e, cx, 200 + 2 * (int)sj), sl.frame_dev, sl.pend_w, sl.pend_h, rtp_internal_warp_inverse_scale(s), e->warp_tab_dev, sl.disp_dev, e->cfg.disp_w, e->cfg.disp_h, cx.stream));
    }
    HIPCHK(e, launch_area_pad(stamp_slot(e, cx, 200 + 2 * (int)sj + 1), sl.disp_cur, e->cfg.disp_w, e->cfg.disp_h, e->area_scales.data(), e->N, dst, e->cfg.net_w, e->cfg.net_h, cx.stream));
    sl.copy_pending = false;
  }
  return RTP_OK;
}

// Device state of the current plan: weight arena (packed + uploaded from w_ref / b_ref), `nctx` batch contexts, the dry run
// warmup() does (rtpose.cpp:233; it also sets the kernels' LDS attributes), and — `capture` — the full-batch graph of every context.
int materialize_plan(rtp_engine* e, int nctx, bool capture) {
  int rc;
  if ((rc = use_device(e))) return rc;
  {
    SYNC_GUARD;
    hipError_t s = hipMalloc((void**)&e->dweights, e->weights_bytes);
    if (s != hipSuccess) { e->dweights = nullptr; return fail(e, RTP_ENOMEM, "hipMalloc(%zu) for weights failed: %s", e->weights_bytes, hipGetErrorString(s)); }
    s = hipMemset(e->dweights, 0, e->weights_bytes);
    if (s != hipSuccess) return fail(e, RTP_EHIP, "hipMemset failed: %s", hipGetErrorString(s));
    if (!e->dchmap) {
      s = hipMalloc((void**)&e->dchmap, 4096 * sizeof(int));
      if (s != hipSuccess) return fail(e, RTP_ENOMEM, "hipMalloc failed");
    }
  }
  if (!e->weights_pending) {
    compute_wq_exp(e);
    if ((rc = upload_all_weights(e))) return rc;
  }
  SYNC_GUARD;   // contexts (hipMemset of the arenas), the dry run's synchronisation, the graph captures
  e->ctx.resize(nctx);
  for (auto& c : e->ctx)
    if ((rc = alloc_ctx(e, c))) return rc;
  // staging streams LAST: the runtime deals its hardware queues to streams round-robin in creation order, and the arrangement of the
  // conv / post-processing streams is the measured optimum
  for (auto& c : e->ctx) {
    if (e->in_stream_mode == 0) c.in_stream = c.stream;
    else if (e->in_stream_mode == 2) {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      HIPCHK(e, hipStreamCreateWithPriority(&c.in_stream, hipStreamNonBlocking, greatest));
    } else HIPCHK(e, hipStreamCreateWithFlags(&c.in_stream, hipStreamNonBlocking));
    HIPCHK(e, hipEventCreateWithFlags(&c.ev_in, hipEventDisableTiming));
  }
  if (e->prep_defer && !e->copy_stream) HIPCHK(e, hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  {
    Ctx& cx = e->ctx[0];
    const size_t in_floats = (size_t)e->N * 3 * e->cfg.net_h * e->cfg.net_w;
    HIPCHK(e, hipMemsetAsync(cx.input, 0, in_floats * sizeof(float), cx.stream));
    if ((rc = launch_batch_body(e, cx, 1, cx.input, false, false))) return rc;  // eager: also sets the kernels' LDS attributes
    hipError_t s = hipStreamSynchronize(cx.stream);
    if (s != hipSuccess) return fail(e, RTP_EHIP, "dry run failed: %s", hipGetErrorString(s));
    for (Slot& sl : cx.slot) {
      s = hipStreamSynchronize(sl.stream);
      if (s != hipSuccess) return fail(e, RTP_EHIP, "dry run failed: %s", hipGetErrorString(s));
    }
  }
  if (capture && e->use_graph && !e->cfg.render)  // capture the plan of every context now — the full batch AND the partial batches a flush /
    for (auto& c : e->ctx)                        // collect can launch — so that the per-frame path never captures (never takes g_sync_mutex)
      for (int nf = e->B; nf >= 1; --nf)
        if ((rc = capture_batch(e, c, nf, &c.gexec[nf]))) return rc;
  return RTP_OK;
}

// Drop the plan and everything on the device that was laid out for it; the reference weights (w_ref / b_ref), thresholds, scales and
// the pre-processing tables stay.
void drop_plan(rtp_engine* e) {
  SYNC_GUARD;
  (void)hipSetDevice(e->cfg.device_id);
  (void)hipDeviceSynchronize();
  for (auto& c : e->ctx) free_ctx(c);
  e->ctx.clear();
  if (e->dweights) { (void)hipFree(e->dweights); e->dweights = nullptr; }
  e->tensors.clear(); e->blob_tensor.clear(); e->blob_dims.clear(); e->convs.clear(); e->steps.clear(); e->pools.clear();
  e->open_ctx = -1;
  e->fifo.clear();
  e->pending_launch.clear();
}

// Re-plan an idle engine for another precision mode / split set (load-time calibration).  light = one context, no graph capture.
int replan(rtp_engine* e, int mode, const std::string& rules, bool light) {
  drop_plan(e);
  e->mode = mode;
  e->prec = mode == RTP_PREC_FP32 ? 1 : 0;
  e->elem = e->prec ? 4 : 2;
  e->split_rules = rules;
  int rc = build_plan(e);
  if (rc) return rc;
  return materialize_plan(e, light ? 1 : e->nctx_full, !light);
}

int busy_mark_stage0(rtp_engine* e, Ctx& cx) {
  if (!cx.ev_stage0) HIPCHK(e, hipEventCreateWithFlags(&cx.ev_stage0, event_flags(false)));
  HIPCHK(e, hipEventRecord(cx.ev_stage0, cx.in_stream));
  cx.stage0_set = true;
  return RTP_OK;
}

// rtp_config.defer_weights: entries that would run the net on an all-zero arena refuse
int need_weights(rtp_engine* e) {
  if (e->weights_pending) return fail(e, RTP_EINVAL, "this engine was created with defer_weights = 1 and has no weights yet: rtp_weight_blob_import or rtp_copy_weights_from first");
  return RTP_OK;
}
// ... and once the weights are there, the launch graphs of every context are captured (what materialize_plan does for other engines)
int weights_delivered(rtp_engine* e) {
  if (!e->weights_pending) return RTP_OK;
  e->weights_pending = false;
  int rc;
  if ((rc = invalidate_graphs(e))) return rc;
  if (e->use_graph && !e->cfg.render)
    for (auto& c : e->ctx)
      for (int nf = e->B; nf >= 1; --nf)
        if ((rc = capture_batch(e, c, nf, &c.gexec[nf]))) return rc;
  return RTP_OK;
}

// ---- batching: frames are staged into the open context; a full batch is launched at once ----------
int open_slot(rtp_engine* e, int* ci, int* sj) {
  if (e->open_ctx < 0) {
    for (size_t i = 0; i < e->ctx.size() && e->open_ctx < 0; ++i) {
      Ctx& c = e->ctx[i];
      bool idle = !c.launched && c.filled == 0;
      for (Slot& s : c.slot) idle = idle && !s.busy;
      if (idle) e->open_ctx = (int)i;
    }
    if (e->open_ctx < 0) return fail(e, RTP_EAGAIN, "all %zu batch contexts (%d frames each) are busy", e->ctx.size(), e->B);
    if (e->stamp_probe) {   // the context's previous batch is complete: read its kernels' residency stamps before anything of the new batch is launched
      Ctx& c = e->ctx[e->open_ctx];
      const int rc = stamp_harvest(e, c);
      if (rc) return rc;
      c.stamps_dirty = true;
    }
  }
  *ci = e->open_ctx;
  *sj = e->ctx[e->open_ctx].filled;
  return RTP_OK;
}
int launch_open(rtp_engine* e) {
  if (e->open_ctx < 0) return RTP_OK;
  Ctx& cx = e->ctx[e->open_ctx];
  e->open_ctx = -1;
  if (cx.filled == 0) return RTP_OK;
  return launch_batch(e, cx, cx.filled, cx.input);
}
// Deferred pre-processing: launch the full batches that were waiting for their last frame's copy, oldest first.
//   PUMP_POLL   launch what is ready (hipEventQuery), leave the rest for the next call            — rtp_submit*
//   PUMP_HOST   wait for the copies ON THE HOST (<= one PCIe copy, ~55 us), then launch             — rtp_collect, which is about to block for a
//               whole frame anyway: a batch must not sit un-launched behind that wait
//   PUMP_STREAM launch now, the conv stream waits for the copies                                   — somebody needs this very batch (rtp_flush)
enum { PUMP_POLL = 0, PUMP_HOST = 1, PUMP_STREAM = 2 };
int pump(rtp_engine* e, int mode) {
  while (!e->pending_launch.empty()) {
    Ctx& cx = e->ctx[e->pending_launch.front()];
    if (mode != PUMP_STREAM)
      for (Slot& sl : cx.slot) {
        if (!sl.copy_pending) continue;
        if (mode == PUMP_HOST) HIPCHK(e, hipEventSynchronize(sl.ev_copy));
        else if (hipEventQuery(sl.ev_copy) != hipSuccess) return RTP_OK;   // not yet: the next API call looks again
      }
    e->pending_launch.pop_front();
    const int rc = launch_batch(e, cx, cx.filled, cx.input);   // (flush_prep inside: every copy is done, or waited for on the stream)
    if (rc) return rc;
  }
  return RTP_OK;
}
int commit_slot(rtp_engine* e, int ci, int sj, uint64_t tag) {
  Ctx& cx = e->ctx[ci];
  cx.slot[sj].tag = tag;
  cx.slot[sj].busy = true;
  cx.filled = sj + 1;
  e->fifo.push_back(ci * 64 + sj);
  if (cx.filled == e->B) {
    if (e->prep_defer && e->B > 1 && cx.slot[sj].copy_pending) {   // the frame that completes the batch was copied a moment ago: launch at the next call
      e->open_ctx = -1;
      e->pending_launch.push_back(ci);
      return RTP_OK;
    }
    if (e->prep_defer) { const int rc = pump(e, PUMP_STREAM); if (rc) return rc; }   // keep the launch order
    return launch_open(e);
  }
  return RTP_OK;
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

int rtp_config_default(rtp_config* cfg) {
  if (!cfg) return RTP_EINVAL;
  memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = (unsigned)sizeof *cfg;
  cfg->device_id = 0;
  cfg->model = RTP_MODEL_COCO_18;
  cfg->proto_path = nullptr;
  cfg->weights_path = nullptr;
  cfg->synthetic_seed = 1;
  cfg->net_w = 656; cfg->net_h = 368;
  cfg->num_scales = 1;
  cfg->start_scale = 1.f; cfg->scale_gap = 0.3f;
  cfg->disp_w = 1280; cfg->disp_h = 720;
  cfg->precision = RTP_PREC_MIXED;  // the fastest mode inside the north-star tolerance
  cfg->frames_in_flight = 2;
  cfg->batch_frames = 1;
  cfg->exec_mode = RTP_EXEC_GRAPH;
  cfg->calibrate_frames = 0;
  cfg->calibrate_target = 0.7e-3f;
  cfg->defer_weights = 0;
  return RTP_OK;
}

const char* rtp_version(void) { return "rtpose-mi355x 0.1 (gfx950)"; }

const char* rtp_last_error(const rtp_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

void rtp_engine_destroy(rtp_engine* e) {
  if (!e) return;
  SYNC_GUARD;
  (void)hipSetDevice(e->cfg.device_id);
  for (auto& c : e->ctx) free_ctx(c);
  if (e->dweights) (void)hipFree(e->dweights);
  if (e->dchmap) (void)hipFree(e->dchmap);
  for (hipEvent_t ev : e->tev) if (ev) (void)hipEventDestroy(ev);
  if (e->busy_base) (void)hipEventDestroy(e->busy_base);
  if (e->prep_tables) (void)hipFree(e->prep_tables);
  if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
  for (void* p : e->user_bufs) if (p) (void)hipFree(p);
  delete e;
}

// CHECK_LE(target_width, NET_RESOLUTION_WIDTH) / CHECK_LE(target_height, ..) (rtpose.cpp:363-364, 513-514): every scale
// s = start_scale - i * scale_gap (the producer's own arithmetic: double, then float, :360) must be positive and its 16-aligned target
// must fit the net input; NaN / inf never do.  The same test guards ImResize's crops (imresize_layer.cu:110-113: padw >= 0, ow >= 1).
static bool scales_fit(int net_w, int net_h, int num_scales, float start_scale, float scale_gap, int* bad, float* bad_s) {
  for (int i = 0; i < num_scales; ++i) {
    const float s = (float)((double)start_scale - i * (double)scale_gap);
    const bool ok = s > 0.f && 16 * std::ceil(net_w * s / 16) <= net_w && 16 * std::ceil(net_h * s / 16) <= net_h &&
                    16 * std::ceil(net_w * s / 16) >= 16 && 16 * std::ceil(net_h * s / 16) >= 16;
    if (!ok) { if (bad) *bad = i; if (bad_s) *bad_s = s; return false; }
  }
  return true;
}

static int engine_create_impl(const rtp_config* cfg, rtp_engine** out) {
  if (!cfg || !out) return fail(nullptr, RTP_EINVAL, "null argument");
  *out = nullptr;
  if (cfg->num_scales < 1 || cfg->num_scales > 16) return fail(nullptr, RTP_EINVAL, "num_scales %d out of range", cfg->num_scales);
  if (cfg->calibrate_frames < -1 || cfg->calibrate_frames > 64) return fail(nullptr, RTP_EINVAL, "calibrate_frames %d out of range [-1, 64]", cfg->calibrate_frames);
  if (cfg->frames_in_flight < 1 || cfg->frames_in_flight > 64) return fail(nullptr, RTP_EINVAL, "frames_in_flight %d out of range", cfg->frames_in_flight);
  if (cfg->batch_frames < 0 || cfg->batch_frames > 16) return fail(nullptr, RTP_EINVAL, "batch_frames %d out of range", cfg->batch_frames);
  {  // render = 1 + part_to_show: the view must stay inside the model's maps (44 MPI: parts + background + 28 PAFs; COCO: 18 parts, "all", 20 PAF views)
    const int part = cfg->render - 1;
    const bool mpi = cfg->model == RTP_MODEL_MPI_15;
    if (cfg->render < 0 || (cfg->render > 1 && (mpi ? part > 44 : part > 39)))
      return fail(nullptr, RTP_EINVAL, "render %d: part_to_show %d is outside the model's maps", cfg->render, part);
  }
  if (cfg->precision < RTP_PREC_FP16 || cfg->precision > RTP_PREC_F16X3) return fail(nullptr, RTP_EINVAL, "unknown precision %d", cfg->precision);
  if (cfg->exec_mode != RTP_EXEC_GRAPH && cfg->exec_mode != RTP_EXEC_EAGER) return fail(nullptr, RTP_EINVAL, "unknown exec_mode %d", cfg->exec_mode);
  if (cfg->disp_w < 1 || cfg->disp_h < 1) return fail(nullptr, RTP_EINVAL, "bad display resolution");
  if ((cfg->net_w % 16) || (cfg->net_h % 16) || cfg->net_w < 16 || cfg->net_h < 16)
    return fail(nullptr, RTP_EINVAL, "net_resolution %dx%d must be positive multiples of 16", cfg->net_w, cfg->net_h);
  {
    int bad = -1;
    float s = 0.f;
    if (!scales_fit(cfg->net_w, cfg->net_h, cfg->num_scales, cfg->start_scale, cfg->scale_gap, &bad, &s))
      return fail(nullptr, RTP_EINVAL, "scale %d (%.3f) does not fit the net resolution", bad, s);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, RTP_ENODEV, "no HIP device visible: this engine has no CPU fallback");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, RTP_ENODEV, "device %d not present (%d devices)", cfg->device_id, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess) return fail(nullptr, RTP_ENODEV, "cannot query device %d", cfg->device_id);
  if (!strstr(prop.gcnArchName, "gfx950")) return fail(nullptr, RTP_ENODEV, "device %d is %s; this engine is built for gfx950 (MI355X) only", cfg->device_id, prop.gcnArchName);

  rtp_engine* e = new rtp_engine();
  e->cfg = *cfg;
  if (cfg->proto_path) { e->proto_path = cfg->proto_path; e->cfg.proto_path = e->proto_path.c_str(); }
  if (cfg->weights_path) { e->weights_path = cfg->weights_path; e->cfg.weights_path = e->weights_path.c_str(); }
  e->mode = cfg->precision;
  e->prec = cfg->precision == RTP_PREC_FP32 ? 1 : 0;
  e->elem = e->prec ? 4 : 2;
  {
    const char* sr = RTP_EXP_ENV("RTP_SPLIT_LAYERS");  // experiments: override the split set of RTP_PREC_MIXED
    e->split_rules = sr ? sr : (cfg->split_layers ? cfg->split_layers : kDefaultSplit);
    e->cfg.split_layers = nullptr;
  }
  e->N = cfg->num_scales;
  e->B = cfg->batch_frames < 1 ? 1 : cfg->batch_frames;
  e->NI = e->N * e->B;
  e->start_scale = cfg->start_scale;
  e->scale_gap = cfg->scale_gap;
  {
    const char* eg = getenv("RTP_EXEC");  // the ONE environment variable the production library reads: "eager" / "graph" override rtp_config.exec_mode (same results either way)
    e->use_graph = cfg->exec_mode == RTP_EXEC_GRAPH;
    if (eg && !strcmp(eg, "eager")) e->use_graph = false;
    if (eg && !strcmp(eg, "graph")) e->use_graph = true;
    const char* f8 = RTP_EXP_ENV("RTP_SPLIT_FP8");
    e->split_fp8 = !(f8 && f8[0] == '0');
    const char* gp = RTP_EXP_ENV("RTP_GRAPH_POST");
    e->graph_post = gp && gp[0] == '1';
    const char* im = RTP_EXP_ENV("RTP_IN_STREAM");   // experiments: 0 = stage inputs on the conv stream, 1 = own stream, 2 = own high-priority stream
    if (im) e->in_stream_mode = atoi(im);
    const char* cc = RTP_EXP_ENV("RTP_CHAIN_CONNECT");
    if (cc) e->chain_connect = atoi(cc);
    const char* pd = RTP_EXP_ENV("RTP_PREP_DEFER");   // experiments: 0 / 1 = pre-processing kernels right behind the copy / once the copy has completed
    if (pd) e->prep_defer = atoi(pd);
    if (e->B == 1) e->prep_defer = 0;   // a batch of one frame is launched by the submit that stages it: nothing to defer, the copy stays on the conv stream
  }
  auto bail = [&](int rc) { g_create_error = e->err; rtp_engine_destroy(e); return rc; };

  if (!e->proto_path.empty()) {
    std::ifstream f(e->proto_path);
    if (!f) return bail(fail(e, RTP_EIO, "cannot open prototxt %s", e->proto_path.c_str()));
    std::stringstream ss;
    ss << f.rdbuf();
    std::string perr;
    if (!parse_prototxt(ss.str(), &e->net, &perr)) return bail(fail(e, RTP_EIO, "prototxt %s: %s", e->proto_path.c_str(), perr.c_str()));
  } else {
    if (cfg->model != RTP_MODEL_COCO_18 && cfg->model != RTP_MODEL_MPI_15) return bail(fail(e, RTP_EINVAL, "unknown model %d", cfg->model));
    e->net = build_linevec(cfg->model);
  }
  int rc = build_plan(e);
  if (rc) return bail(rc);
  // thresholds as warmup() sets them (rtpose.cpp:212-226)
  rtp_default_thresholds(e->model, &e->nms_threshold, &e->inter_threshold, &e->inter_min_above, &e->min_subset_cnt, &e->min_subset_score);

  // weights
  e->w_ref.resize(e->convs.size());
  e->b_ref.resize(e->convs.size());
  e->weights_pending = cfg->defer_weights != 0;
  if (e->weights_pending) {   // a receiving replica: sizes only (rtp_get_conv_weights / rtp_weight_blob_bytes need them), contents arrive with the blob
    for (size_t i = 0; i < e->convs.size(); ++i) {
      const ConvOp& c = e->convs[i];
      e->w_ref[i].assign((size_t)c.cout * c.cin * c.k * c.k, 0.f);
      e->b_ref[i].assign((size_t)c.cout, 0.f);
    }
  } else if (!e->weights_path.empty()) {
    std::vector<LayerWeights> lw;
    std::string werr;
    if (!read_caffemodel(e->weights_path, &lw, &werr)) return bail(fail(e, RTP_EIO, "%s", werr.c_str()));
    // CopyTrainedLayersFrom: match by layer name; unknown source layers ignored; shapes must match (net.cpp:750-786)
    for (size_t i = 0; i < e->convs.size(); ++i) {
      const ConvOp& c = e->convs[i];
      const LayerWeights* src = nullptr;
      for (auto& L : lw) if (L.name == c.name) src = &L;
      if (!src) return bail(fail(e, RTP_EIO, "layer %s has no weights in %s", c.name.c_str(), e->weights_path.c_str()));
      if (src->blobs.size() != 2 || src->blobs[0].data.size() != (size_t)c.cout * c.cin * c.k * c.k || src->blobs[1].data.size() != (size_t)c.cout)
        return bail(fail(e, RTP_EIO, "Cannot copy param of layer '%s': shape mismatch", c.name.c_str()));
      e->w_ref[i] = src->blobs[0].data;
      e->b_ref[i] = src->blobs[1].data;
    }
  } else {
    for (size_t i = 0; i < e->convs.size(); ++i) {
      const ConvOp& c = e->convs[i];
      synth_conv_weights(cfg->synthetic_seed, c.name, c.cout, c.cin, c.k, &e->w_ref[i], &e->b_ref[i]);
    }
  }

  e->nctx_full = (cfg->frames_in_flight + e->B - 1) / e->B + (e->B > 1 ? 1 : 0);  // batches in flight (+1 being filled)
  if ((rc = materialize_plan(e, e->nctx_full, !e->weights_pending))) return bail(rc);   // (a receiving replica captures when its weights arrive: wq_exp is baked into the graphs)
  if ((rc = build_prep_tables(e))) return bail(rc);
  // Load-time precision calibration (net.cpp:750-803 is where real weights arrive).  The default split set was chosen on synthetic
  // weights; weights that come from a FILE are checked by default (one synthetic frame: mixed vs F16X3, and the set is widened if the
  // check fails) — two of four unseen weight families are 4-7x outside the tolerance with the default set.
  // calibrate_frames = -1 opts out; > 0 asks for that many frames (also for synthetic weights).
  int calib = cfg->calibrate_frames;
  if (calib == 0 && !e->weights_path.empty()) calib = 1;
  if (calib > 0 && e->mode == RTP_PREC_MIXED && !e->weights_pending) {
    float before = 0.f, after = 0.f;
    rc = rtp_calibrate_precision(e, nullptr, calib, cfg->calibrate_target, nullptr, 0, &before, &after);
    if (rc && (cfg->calibrate_frames > 0 || e->broken)) return bail(rc);   // asked for explicitly, or the engine lost its plan: a create failure
    if (rc) {  // the UNREQUESTED default check could not run (reference maps zero / not finite on the synthetic frame, no memory for the trial
               // plans next to N other processes, ...): rtp_calibrate_precision restored the plan it found — keep it and say so
      fprintf(stderr, "rtpose-mi355x: the default precision check on the weights of %s could not run (%s); the engine keeps the default split set "
                      "(rtp_config.calibrate_frames > 0 makes this an error, -1 skips the check)\n", e->weights_path.c_str(), e->err.c_str());
      e->err.clear();
    } else if (cfg->calibrate_frames == 0 && (e->split_rules != (cfg->split_layers ? std::string(cfg->split_layers) : std::string(kDefaultSplit)) || e->mode != RTP_PREC_MIXED))
      fprintf(stderr, "rtpose-mi355x: the default mixed-precision split set measured %.2e of the map maximum on the weights of %s (tolerance 1e-3, target %.1e): "
                      "now %s \"%s\" at %.2e (rtp_config.calibrate_frames = -1 keeps the default set)\n", before, e->weights_path.c_str(),
              cfg->calibrate_target > 0 ? cfg->calibrate_target : 0.7e-3f, e->mode == RTP_PREC_MIXED ? "mixed" : "f16x3", e->split_rules.c_str(), after);
  }
  *out = e;
  return RTP_OK;
}

int rtp_engine_info(const rtp_engine* e, int* num_parts, int* max_peaks, int* heat_channels, int* low_w, int* low_h) {
  if (!e) return RTP_EINVAL;
  if (num_parts) *num_parts = e->num_parts;
  if (max_peaks) *max_peaks = e->max_peaks;
  if (heat_channels) *heat_channels = e->heat_channels;
  if (low_w) *low_w = e->low_w;
  if (low_h) *low_h = e->low_h;
  return RTP_OK;
}

static int need_idle(rtp_engine* e);
int rtp_set_thresholds(rtp_engine* e, float nms_threshold, float connect_inter_threshold, int connect_inter_min_above_threshold,
                       int connect_min_subset_cnt, float connect_min_subset_score) {
  SYNC_GUARD;
  if (!e) return RTP_EINVAL;
  if (e->graph_post && !e->ctx.empty()) {  // the post-processing chains live inside the batch graphs: their arguments are baked
    int rc;
    if ((rc = need_idle(e))) return rc;
    if ((rc = invalidate_graphs(e))) return rc;
  }
  e->nms_threshold = nms_threshold;
  e->inter_threshold = connect_inter_threshold;
  e->inter_min_above = connect_inter_min_above_threshold;
  e->min_subset_cnt = connect_min_subset_cnt;
  e->min_subset_score = connect_min_subset_score;
  return RTP_OK;
}
int rtp_get_thresholds(const rtp_engine* e, float* a, float* b, int* c, int* d, float* f) {
  if (!e) return RTP_EINVAL;
  if (a) *a = e->nms_threshold;
  if (b) *b = e->inter_threshold;
  if (c) *c = e->inter_min_above;
  if (d) *d = e->min_subset_cnt;
  if (f) *f = e->min_subset_score;
  return RTP_OK;
}
int rtp_set_scales(rtp_engine* e, float start_scale, float scale_gap) {
  SYNC_GUARD;
  if (!e) return RTP_EINVAL;
  {
    int bad = -1;
    float s = 0.f;
    if (!scales_fit(e->cfg.net_w, e->cfg.net_h, e->N, start_scale, scale_gap, &bad, &s))   // nothing is changed by a refused call
      return fail(e, RTP_EINVAL, "rtp_set_scales: scale %d (%.3f) does not fit the net resolution (rtpose.cpp:363)", bad, s);
  }
  e->start_scale = start_scale;
  e->scale_gap = scale_gap;
  e->cfg.start_scale = start_scale;
  e->cfg.scale_gap = scale_gap;
  if (!e->ctx.empty()) {
    int rc;
    if ((rc = use_device(e))) return rc;
    HIPCHK(e, hipDeviceSynchronize());
    if ((rc = build_prep_tables(e))) return rc;
    if (e->graph_post && (rc = invalidate_graphs(e))) return rc;
  }
  return RTP_OK;
}

int rtp_submit_device(rtp_engine* e, const float* d_in, uint64_t tag) {
  if (!e || !d_in) return RTP_EINVAL;
  int rc, ci, sj;
  if ((rc = use_device(e))) return rc;
  if ((rc = need_weights(e))) return rc;
  if (e->prep_defer && (rc = pump(e, PUMP_POLL))) return rc;
  if ((rc = open_slot(e, &ci, &sj))) return rc;
  Ctx& cx = e->ctx[ci];
  if (e->busy_probe && sj == 0 && (rc = busy_mark_stage0(e, cx))) return rc;
  const size_t bytes = (size_t)e->N * 3 * e->cfg.net_h * e->cfg.net_w * sizeof(float);
  if (e->B == 1 && !e->use_graph) {  // eager: no staging copy, the conv stack reads the caller's tensor
    cx.slot[0].tag = tag;
    cx.slot[0].busy = true;
    cx.slot[0].has_disp = false;
    cx.filled = 1;
    e->fifo.push_back(ci * 64);
    e->open_ctx = -1;
    return launch_batch(e, cx, 1, d_in);
  }
  HIPCHK(e, hipMemcpyAsync((char*)cx.input + sj * bytes, d_in, bytes, hipMemcpyDeviceToDevice, cx.in_stream));
  cx.in_pending = true;
  cx.slot[sj].has_disp = false;
  return commit_slot(e, ci, sj, tag);
}

int rtp_submit(rtp_engine* e, const float* h_in, uint64_t tag) {
  if (!e || !h_in) return RTP_EINVAL;
  int rc, ci, sj;
  if ((rc = use_device(e))) return rc;
  if ((rc = need_weights(e))) return rc;
  if (e->prep_defer && (rc = pump(e, PUMP_POLL))) return rc;
  if ((rc = open_slot(e, &ci, &sj))) return rc;
  Ctx& cx = e->ctx[ci];
  if (e->busy_probe && sj == 0 && (rc = busy_mark_stage0(e, cx))) return rc;
  const size_t bytes = (size_t)e->N * 3 * e->cfg.net_h * e->cfg.net_w * sizeof(float);
  memcpy((char*)cx.host_in + sj * bytes, h_in, bytes);
  if (e->prep_defer) {   // the copy on the copy-only stream; the conv stream learns about it when it has completed (flush_prep / pump)
    Slot& sl = cx.slot[sj];
    if (!sl.ev_copy) HIPCHK(e, hipEventCreateWithFlags(&sl.ev_copy, hipEventDisableTiming));
    HIPCHK(e, hipMemcpyAsync((char*)cx.input + sj * bytes, (char*)cx.host_in + sj * bytes, bytes, hipMemcpyHostToDevice, e->copy_stream));
    HIPCHK(e, hipEventRecord(sl.ev_copy, e->copy_stream));
    sl.copy_pending = true;
    sl.pend_w = sl.pend_h = 0;
  } else {
    HIPCHK(e, hipMemcpyAsync((char*)cx.input + sj * bytes, (char*)cx.host_in + sj * bytes, bytes, hipMemcpyHostToDevice, cx.in_stream));
    cx.in_pending = true;
  }
  cx.slot[sj].has_disp = false;
  return commit_slot(e, ci, sj, tag);
}

// Replaces the producer's per-frame OpenCV work + H2D (rtpose.cpp:322-368, 1131-1133): raw u8 BGR frame
// in, pre-processing on the device (preproc.hip), then the same frame path as rtp_submit.
int rtp_submit_frame(rtp_engine* e, const unsigned char* bgr, int w, int h, uint64_t tag, float* frame_scale) {
  if (!e || !bgr || w < 1 || h < 1) return RTP_EINVAL;
  int rc, ci, sj;
  if ((rc = use_device(e))) return rc;
  if ((rc = need_weights(e))) return rc;
  if (e->prep_defer && (rc = pump(e, PUMP_POLL))) return rc;
  if ((rc = open_slot(e, &ci, &sj))) return rc;
  Ctx& cx = e->ctx[ci];
  if (e->busy_probe && sj == 0 && (rc = busy_mark_stage0(e, cx))) return rc;
  if (e->prep_defer && (rc = flush_prep(e, cx, false))) return rc;   // an earlier frame of this batch whose copy is done by now
  cx.slot[sj].has_disp = e->gpu_prep_ok;
  if (e->gpu_prep_ok) {
    if ((rc = enqueue_preprocess(e, cx, sj, bgr, w, h, frame_scale))) return rc;
  } else {  // a pyramid level would have to be enlarged: host restatement (linear fallback) + H2D
    const size_t bytes = (size_t)e->N * 3 * e->cfg.net_h * e->cfg.net_w * sizeof(float);
    float* hin = (float*)((char*)cx.host_in + sj * bytes);
    rc = rtp_preprocess_frame(bgr, w, h, e->cfg.disp_w, e->cfg.disp_h, e->cfg.net_w, e->cfg.net_h, e->N, e->start_scale, e->scale_gap,
                              hin, nullptr, frame_scale);
    if (rc) return fail(e, rc, "pre-processing failed (a scale does not fit the net resolution?)");
    HIPCHK(e, hipMemcpyAsync((char*)cx.input + sj * bytes, hin, bytes, hipMemcpyHostToDevice, cx.in_stream));
    cx.in_pending = true;
  }
  return commit_slot(e, ci, sj, tag);
}

// Launch a partially filled batch now (end of stream, or a latency-sensitive caller).
int rtp_flush(rtp_engine* e) {
  if (!e) return RTP_EINVAL;
  int rc;
  if ((rc = use_device(e))) return rc;
  if (e->prep_defer && (rc = pump(e, PUMP_STREAM))) return rc;
  return launch_open(e);
}

static int need_idle(rtp_engine* e);
// Parity tap: the device pre-processing alone (net input and display image back on the host).
int rtp_debug_preprocess(rtp_engine* e, const unsigned char* bgr, int w, int h, float* net_input, unsigned char* display_bgr, float* frame_scale) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (!bgr || w < 1 || h < 1) return RTP_EINVAL;
  if (!e->gpu_prep_ok) return fail(e, RTP_EINVAL, "device pre-processing unavailable for this configuration (a level would be enlarged)");
  Ctx& cx = e->ctx[0];
  if ((rc = enqueue_preprocess(e, cx, 0, bgr, w, h, frame_scale))) return rc;
  if (e->prep_defer && (rc = flush_prep(e, cx, true))) return rc;
  cx.in_pending = false;
  HIPCHK(e, hipStreamSynchronize(e->prep_defer ? cx.stream : cx.in_stream));
  if (net_input) HIPCHK(e, hipMemcpy(net_input, cx.input, (size_t)e->N * 3 * e->cfg.net_h * e->cfg.net_w * sizeof(float), hipMemcpyDeviceToHost));
  if (display_bgr) HIPCHK(e, hipMemcpy(display_bgr, cx.slot[0].disp_cur, (size_t)e->cfg.disp_w * e->cfg.disp_h * 3, hipMemcpyDeviceToHost));
  return RTP_OK;
}

int rtp_in_flight(const rtp_engine* e) { return e ? (int)e->fifo.size() : 0; }

static void stage_ms(rtp_engine* e, Ctx& cx, Slot& sl) {
  if (cx.graph_run) {  // events recorded inside a capture carry no time: only the whole replay is bracketed
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, cx.gev[0], cx.gev[1]);
    for (int i = 0; i < 4; ++i) e->last_ms[i] = 0.f;
    e->last_ms[4] = ms;
    return;
  }
  hipEvent_t start = cx.graph_conv ? cx.gev[0] : cx.ev[0];  // the conv stack's own start event lives inside the replay
  hipEvent_t a[5] = {start, sl.ev[0], sl.ev[1], sl.ev[2], start};
  hipEvent_t b[5] = {cx.ev[1], sl.ev[1], sl.ev[2], sl.ev[3], sl.ev[4]};
  for (int i = 0; i < 5; ++i) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a[i], b[i]);
    e->last_ms[i] = ms;
  }
}

static int collect_impl(rtp_engine* e, uint64_t* tag, float* joints, int* num_people, unsigned char* rendered, bool want_render);
int rtp_collect(rtp_engine* e, uint64_t* tag, float* joints, int* num_people) { return collect_impl(e, tag, joints, num_people, nullptr, false); }
// rtp_collect + the display-resolution frame with the pose overlay (the image the reference hands to
// cv::imwrite under --write_frames --no_text, rtpose.cpp:1179-1199, 1286-1293).  Needs
// rtp_config.render = 1 and frames submitted with rtp_submit_frame.
int rtp_collect_rendered(rtp_engine* e, uint64_t* tag, float* joints, int* num_people, unsigned char* display_bgr) {
  return collect_impl(e, tag, joints, num_people, display_bgr, true);
}
static int collect_impl(rtp_engine* e, uint64_t* tag, float* joints, int* num_people, unsigned char* rendered, bool want_render) {
  if (!e) return RTP_EINVAL;
  if (e->fifo.empty()) return fail(e, RTP_EAGAIN, "nothing in flight");
  int rc;
  if ((rc = use_device(e))) return rc;
  const int ci = e->fifo.front() / 64, sj = e->fifo.front() % 64;
  Ctx& cx = e->ctx[ci];
  Slot& sl = cx.slot[sj];
  if (e->prep_defer) {
    if ((rc = pump(e, PUMP_STREAM))) return rc;   // nothing stays un-launched while this call blocks for a frame (a host-side wait for the copy
                                                  // was measured: hipEventSynchronize costs ~0.3 ms per call here, 1010 -> 777 frames/s)
    if (e->open_ctx >= 0 && e->open_ctx != ci && (rc = flush_prep(e, e->ctx[e->open_ctx], false))) return rc;
  }
  if (!cx.launched && (rc = launch_open(e))) return rc;  // the oldest frame sits in a partial batch
  HIPCHK(e, hipEventSynchronize(cx.graph_run ? cx.gev[1] : sl.ev[4]));
  e->fifo.pop_front();
  sl.busy = false;
  bool any = false;
  for (Slot& s : cx.slot) any = any || s.busy;
  if (!any) { cx.launched = false; cx.filled = 0; }
  int n;
  memcpy(&n, sl.host_out, sizeof(int));
  if (tag) *tag = sl.tag;
  stage_ms(e, cx, sl);
  if (e->busy_probe && e->busy_base && cx.graph_run) e->probe_dropped[2]++;
  if (e->busy_probe && e->busy_base && !cx.graph_run && e->busy_spans.size() >= 3 * 65536) e->probe_dropped[1]++;
  if (e->busy_probe && e->busy_base && !cx.graph_run && e->busy_spans.size() < 3 * 65536) {
    float t0 = 0.f, t1 = 0.f;
    if (sj == 0 && cx.stage0_set && hipEventElapsedTime(&t0, e->busy_base, cx.ev_stage0) == hipSuccess &&
        hipEventElapsedTime(&t1, e->busy_base, cx.ev[1]) == hipSuccess) { e->busy_spans.push_back(0.f); e->busy_spans.push_back(t0); e->busy_spans.push_back(t1); }
    if (hipEventElapsedTime(&t0, e->busy_base, sl.ev[0]) == hipSuccess && hipEventElapsedTime(&t1, e->busy_base, sl.ev[4]) == hipSuccess) {
      e->busy_spans.push_back(1.f); e->busy_spans.push_back(t0); e->busy_spans.push_back(t1);
    }
  }
  if (n < 0) {
    if (num_people) *num_people = 0;
    return fail(e, RTP_ERANGE, "connect: a PAF sample coordinate fell outside the net resolution (the reference CHECK-fails here, rtpose.cpp:928)");
  }
  if (n > RTP_MAX_PEOPLE) n = RTP_MAX_PEOPLE;
  if (num_people) *num_people = n;
  if (joints) memcpy(joints, sl.host_out + 4, (size_t)n * e->num_parts * 3 * sizeof(float));
  if (want_render) {
    if (!e->cfg.render) return fail(e, RTP_EINVAL, "rtp_collect_rendered needs rtp_config.render = 1");
    if (!sl.has_disp || !sl.render_host) return fail(e, RTP_EINVAL, "no display image for this frame: submit it with rtp_submit_frame (device pre-processing)");
    if (rendered) memcpy(rendered, sl.render_host, (size_t)e->cfg.disp_w * e->cfg.disp_h * 3);
  }
  return RTP_OK;
}

int rtp_last_stage_ms(const rtp_engine* e, float ms[5]) {
  if (!e || !ms) return RTP_EINVAL;
  memcpy(ms, e->last_ms, sizeof e->last_ms);
  return RTP_OK;
}

// ---- synchronous taps (context 0; require an idle engine) ---------------------------------------
static int need_idle(rtp_engine* e) {
  if (!e) return RTP_EINVAL;
  if (!e->fifo.empty()) return fail(e, RTP_EAGAIN, "parity taps need an idle engine (collect %zu frames first)", e->fifo.size());
  return use_device(e);
}

int rtp_forward_debug(rtp_engine* e, const float* h_in, float* lowres, float* resized, float* peaks, float* joints, int* num_people) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if ((rc = need_weights(e))) return rc;
  if (!h_in) return RTP_EINVAL;
  Ctx& cx = e->ctx[0];
  const size_t bytes = (size_t)e->N * 3 * e->cfg.net_h * e->cfg.net_w * sizeof(float);
  HIPCHK(e, hipMemcpy(cx.input, h_in, bytes, hipMemcpyHostToDevice));
  if ((rc = enqueue_frame(e, cx, cx.input, true))) return rc;  // taps: materialised map
  HIPCHK(e, hipStreamSynchronize(cx.stream));
  if (lowres) HIPCHK(e, hipMemcpy(lowres, cx.lowres, (size_t)e->N * e->heat_channels * e->low_h * e->low_w * sizeof(float), hipMemcpyDeviceToHost));
  Slot& sl = cx.slot[0];
  cx.launched = false;
  if (resized) HIPCHK(e, hipMemcpy(resized, sl.resized, (size_t)e->heat_channels * e->cfg.net_h * e->cfg.net_w * sizeof(float), hipMemcpyDeviceToHost));
  if (peaks) HIPCHK(e, hipMemcpy(peaks, sl.peaks, (size_t)e->num_parts * (e->max_peaks + 1) * 3 * sizeof(float), hipMemcpyDeviceToHost));
  int n;
  memcpy(&n, sl.host_out, sizeof(int));
  stage_ms(e, cx, sl);
  if (n < 0) { if (num_people) *num_people = 0; return fail(e, RTP_ERANGE, "connect: PAF sample coordinate out of range"); }
  if (num_people) *num_people = n;
  if (joints) memcpy(joints, sl.host_out + 4, (size_t)n * e->num_parts * 3 * sizeof(float));
  return RTP_OK;
}

int rtp_forward_heatmaps(rtp_engine* e, const float* h_in, float* lowres) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if ((rc = need_weights(e))) return rc;
  if (!h_in || !lowres) return RTP_EINVAL;
  Ctx& cx = e->ctx[0];
  const size_t bytes = (size_t)e->N * 3 * e->cfg.net_h * e->cfg.net_w * sizeof(float);
  HIPCHK(e, hipMemcpy(cx.input, h_in, bytes, hipMemcpyHostToDevice));
  if ((rc = run_frame_stack(e, cx, cx.input, e->N))) return rc;
  HIPCHK(e, hipStreamSynchronize(cx.stream));
  HIPCHK(e, hipMemcpy(lowres, cx.lowres, (size_t)e->N * e->heat_channels * e->low_h * e->low_w * sizeof(float), hipMemcpyDeviceToHost));
  return RTP_OK;
}

int rtp_resize(rtp_engine* e, const float* lowres, float* resized) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (!lowres || !resized) return RTP_EINVAL;
  Ctx& cx = e->ctx[0];
  HIPCHK(e, hipMemcpy(cx.lowres, lowres, (size_t)e->N * e->heat_channels * e->low_h * e->low_w * sizeof(float), hipMemcpyHostToDevice));
  if ((rc = run_resize(e, cx))) return rc;
  HIPCHK(e, hipStreamSynchronize(cx.stream));
  HIPCHK(e, hipMemcpy(resized, cx.slot[0].resized, (size_t)e->heat_channels * e->cfg.net_h * e->cfg.net_w * sizeof(float), hipMemcpyDeviceToHost));
  return RTP_OK;
}

// Parity tap for the PRODUCTION post-processing (no resized map): low-res maps in, peaks and joints out.
int rtp_post_from_lowres(rtp_engine* e, const float* lowres, float* peaks, float* joints, int* num_people) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (!lowres) return RTP_EINVAL;
  Ctx& cx = e->ctx[0];
  Slot& sl = cx.slot[0];
  const size_t pbytes = (size_t)e->num_parts * (e->max_peaks + 1) * 3 * sizeof(float);
  HIPCHK(e, hipMemcpy(cx.lowres, lowres, (size_t)e->N * e->heat_channels * e->low_h * e->low_w * sizeof(float), hipMemcpyHostToDevice));
  if (peaks) HIPCHK(e, hipMemcpy(sl.peaks, peaks, pbytes, hipMemcpyHostToDevice));  // stale slots stay, like the reference's blob
  if (RTP_EXP_ENV("RTP_NMS_PROBE")) {  // diagnostics: phase stamps of the middle strip workgroup of part 0
    unsigned long long* d = nullptr;
    HIPCHK(e, hipMalloc((void**)&d, 32 * 8));
    HIPCHK(e, hipMemset(d, 0, 32 * 8));
    NmsParams np = nms_params(e, cx, 0);
    np.probe = d;
    HIPCHK(e, launch_nms_fused(np, resize_params(e, cx, 0), sl.stream));
    HIPCHK(e, hipStreamSynchronize(sl.stream));
    unsigned long long h[32];
    HIPCHK(e, hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    fprintf(stderr, "nms strip probe (us):");
    for (unsigned i = 2; i <= h[0] && i < 32; ++i) fprintf(stderr, " %.2f", (double)(h[i] - h[i - 1]) / 100.0);
    fprintf(stderr, "  total %.2f\n", h[0] ? (double)(h[h[0]] - h[1]) / 100.0 : 0.0);
  }
  HIPCHK(e, hipEventRecord(sl.ev[1], sl.stream));
  if ((rc = run_post_fused(e, cx, 0, sl.ev[2]))) return rc;
  HIPCHK(e, hipEventRecord(sl.ev[3], sl.stream));
  HIPCHK(e, hipStreamSynchronize(sl.stream));
  {  // rtp_last_stage_ms: {-, -, nms, connect, both} of this call
    float a = 0.f, b = 0.f;
    (void)hipEventElapsedTime(&a, sl.ev[1], sl.ev[2]);
    (void)hipEventElapsedTime(&b, sl.ev[2], sl.ev[3]);
    e->last_ms[0] = e->last_ms[1] = 0.f; e->last_ms[2] = a; e->last_ms[3] = b; e->last_ms[4] = a + b;
  }
  if (peaks) HIPCHK(e, hipMemcpy(peaks, sl.peaks, pbytes, hipMemcpyDeviceToHost));
  int n = 0;
  HIPCHK(e, hipMemcpy(&n, sl.num_people, sizeof(int), hipMemcpyDeviceToHost));
  if (n < 0) { if (num_people) *num_people = 0; return fail(e, RTP_ERANGE, "connect: PAF sample coordinate out of range"); }
  if (num_people) *num_people = n;
  if (joints && n > 0) HIPCHK(e, hipMemcpy(joints, sl.joints, (size_t)n * e->num_parts * 3 * sizeof(float), hipMemcpyDeviceToHost));
  return RTP_OK;
}

int rtp_nms(rtp_engine* e, const float* resized, float* peaks) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (!resized || !peaks) return RTP_EINVAL;
  Ctx& cx = e->ctx[0];
  const size_t pbytes = (size_t)e->num_parts * (e->max_peaks + 1) * 3 * sizeof(float);
  HIPCHK(e, hipMemcpy(cx.slot[0].resized, resized, (size_t)e->heat_channels * e->cfg.net_h * e->cfg.net_w * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(cx.slot[0].peaks, peaks, pbytes, hipMemcpyHostToDevice));
  if ((rc = run_nms(e, cx))) return rc;
  HIPCHK(e, hipStreamSynchronize(cx.stream));
  HIPCHK(e, hipMemcpy(peaks, cx.slot[0].peaks, pbytes, hipMemcpyDeviceToHost));
  return RTP_OK;
}

int rtp_connect(rtp_engine* e, const float* resized, const float* peaks, float* joints, int* num_people) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (!resized || !peaks) return RTP_EINVAL;
  Ctx& cx = e->ctx[0];
  HIPCHK(e, hipMemcpy(cx.slot[0].resized, resized, (size_t)e->heat_channels * e->cfg.net_h * e->cfg.net_w * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(cx.slot[0].peaks, peaks, (size_t)e->num_parts * (e->max_peaks + 1) * 3 * sizeof(float), hipMemcpyHostToDevice));
  if ((rc = run_connect(e, cx))) return rc;
  HIPCHK(e, hipStreamSynchronize(cx.stream));
  int n = 0;
  HIPCHK(e, hipMemcpy(&n, cx.slot[0].num_people, sizeof(int), hipMemcpyDeviceToHost));
  if (n < 0) { if (num_people) *num_people = 0; return fail(e, RTP_ERANGE, "connect: PAF sample coordinate out of range"); }
  if (num_people) *num_people = n;
  if (joints && n > 0) HIPCHK(e, hipMemcpy(joints, cx.slot[0].joints, (size_t)n * e->num_parts * 3 * sizeof(float), hipMemcpyDeviceToHost));
  return RTP_OK;
}

// render() of rtpose.cpp:270-299 on caller data: the pose overlay or one of the --part_to_show views
int rtp_render(rtp_engine* e, const unsigned char* display_bgr, const float* joints, int num_people, int part_to_show, int googly,
               const float* resized_host, unsigned char* out_bgr) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (!display_bgr || !out_bgr || num_people < 0 || (num_people > 0 && !joints)) return RTP_EINVAL;
  // the last map a view reads: MPI part_to_show - 1; COCO heat maps <= 18, PAF view p reads channels up to 2*(p-20)-2+19+1
  const int last_map = e->model != 0 ? part_to_show - 1 : (part_to_show <= 19 ? 17 : (part_to_show == 20 ? 56 : 2 * (part_to_show - 20) + 18));
  if (part_to_show < 0 || (part_to_show > 0 && (!resized_host || last_map >= e->heat_channels)))
    return fail(e, RTP_EINVAL, "part_to_show %d is outside the model's %d maps", part_to_show, e->heat_channels);
  Ctx& cx = e->ctx[0];
  Slot& sl = cx.slot[0];
  const int w = e->cfg.disp_w, h = e->cfg.disp_h;
  const size_t dbytes = (size_t)w * h * 3;
  if (!sl.render_dev) {
    HIPCHK(e, hipMalloc((void**)&sl.render_dev, dbytes));
    HIPCHK(e, hipHostMalloc((void**)&sl.render_host, dbytes, hipHostMallocDefault));
    HIPCHK(e, hipMalloc((void**)&sl.render_tab, render_tab_floats(RTP_MAX_PEOPLE) * sizeof(float)));
  }
  if (!sl.disp_dev) HIPCHK(e, hipMalloc((void**)&sl.disp_dev, dbytes));
  HIPCHK(e, hipMemcpy(sl.disp_dev, display_bgr, dbytes, hipMemcpyHostToDevice));
  if (part_to_show == 0) {
    const int n = std::min(num_people, (int)RTP_MAX_PEOPLE);
    if (n > 0) HIPCHK(e, hipMemcpy(sl.joints, joints, (size_t)n * e->num_parts * 3 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(sl.num_people, &n, sizeof(int), hipMemcpyHostToDevice));
    RenderParams rp;
    rp.src = sl.disp_dev; rp.dst = sl.render_dev; rp.w = w; rp.h = h;
    rp.poses = sl.joints; rp.num_people = sl.num_people; rp.tab = sl.render_tab;
    rp.model = e->model; rp.googly = googly ? 1 : 0; rp.max_people = RTP_MAX_PEOPLE;
    HIPCHK(e, launch_render(rp, sl.stream));
  } else {
    HIPCHK(e, hipMemcpy(sl.resized, resized_host, (size_t)e->heat_channels * e->cfg.net_h * e->cfg.net_w * sizeof(float), hipMemcpyHostToDevice));
    RenderViewParams vp;
    vp.src = sl.disp_dev; vp.dst = sl.render_dev; vp.w = w; vp.h = h;
    vp.maps = sl.resized; vp.net_w = e->cfg.net_w; vp.net_h = e->cfg.net_h;
    vp.model = e->model; vp.part_to_show = part_to_show;
    HIPCHK(e, launch_render_view(vp, sl.stream));
  }
  HIPCHK(e, hipStreamSynchronize(sl.stream));
  HIPCHK(e, hipMemcpy(out_bgr, sl.render_dev, dbytes, hipMemcpyDeviceToHost));
  return RTP_OK;
}

int rtp_get_blob(rtp_engine* e, const char* name, float* out, size_t cap, int shape[4]) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (!name) return RTP_EINVAL;
  Ctx& cx = e->ctx[0];
  if (e->lowres_blob == name) {
    const size_t n = (size_t)e->N * e->heat_channels * e->low_h * e->low_w;
    if (shape) { shape[0] = e->N; shape[1] = e->heat_channels; shape[2] = e->low_h; shape[3] = e->low_w; }
    if (!out) return RTP_OK;
    if (cap < n) return fail(e, RTP_EINVAL, "blob %s needs %zu floats", name, n);
    HIPCHK(e, hipMemcpy(out, cx.lowres, n * sizeof(float), hipMemcpyDeviceToHost));
    return RTP_OK;
  }
  auto it = e->blob_tensor.find(name);
  if (it == e->blob_tensor.end()) return fail(e, RTP_EINVAL, "Unknown blob name %s", name);  // net.cpp blob_by_name warning
  const Tensor& t = e->tensors[it->second];
  if (!t.written) return fail(e, RTP_EINVAL, "blob %s is not materialised: its convolution pools in the epilogue and writes only the pooled blob, or it is the middle blob of a fused branch tail that lives in LDS (create the engine with keep_blobs = 1 to tap it)", name);
  Geom g = e->geom[t.level];
  g.N = e->N;  // the taps run one frame (slot 0 of the batch)
  const size_t n = (size_t)g.N * t.C * g.H * g.W;
  if (shape) { shape[0] = g.N; shape[1] = t.C; shape[2] = g.H; shape[3] = g.W; }
  if (!out) return RTP_OK;
  if (cap < n) return fail(e, RTP_EINVAL, "blob %s needs %zu floats", name, n);
  if (t.C > 4096) return fail(e, RTP_EINVAL, "blob too wide");
  float* dtmp = nullptr;
  HIPCHK(e, hipMalloc((void**)&dtmp, n * sizeof(float)));
  hipError_t s = hipMemcpy(e->dchmap, t.chmap.data(), t.C * sizeof(int), hipMemcpyHostToDevice);
  if (s == hipSuccess) s = launch_export(e->prec, cx.arena + t.offset, g, t.stride(), e->dchmap, t.C, t.lo_off(), t.q_off(), dtmp, cx.stream);
  if (s == hipSuccess) s = hipStreamSynchronize(cx.stream);
  if (s == hipSuccess) s = hipMemcpy(out, dtmp, n * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(dtmp);
  if (s != hipSuccess) return fail(e, RTP_EHIP, "export failed: %s", hipGetErrorString(s));
  return RTP_OK;
}

// ---- weights / graph ---------------------------------------------------------------------------
int rtp_num_conv_layers(const rtp_engine* e) { return e ? (int)e->convs.size() : RTP_EINVAL; }
int rtp_conv_layer_info(const rtp_engine* e, int i, char* name, int name_len, int* cin, int* cout, int* k) {
  if (!e || i < 0 || i >= (int)e->convs.size()) return RTP_EINVAL;
  const ConvOp& c = e->convs[i];
  if (name && name_len > 0) snprintf(name, name_len, "%s", c.name.c_str());
  if (cin) *cin = c.cin;
  if (cout) *cout = c.cout;
  if (k) *k = c.k;
  return RTP_OK;
}
int rtp_get_conv_weights(const rtp_engine* e, int i, float* w, float* b) {
  if (!e || i < 0 || i >= (int)e->convs.size()) return RTP_EINVAL;
  if (w) memcpy(w, e->w_ref[i].data(), e->w_ref[i].size() * sizeof(float));
  if (b) memcpy(b, e->b_ref[i].data(), e->b_ref[i].size() * sizeof(float));
  return RTP_OK;
}
int rtp_set_conv_weights(rtp_engine* e, int i, const float* w, const float* b) {
  SYNC_GUARD;
  if (!e || i < 0 || i >= (int)e->convs.size() || !w || !b) return RTP_EINVAL;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if ((rc = need_weights(e))) return rc;   // (a defer_weights engine takes its weights as ONE blob / peer copy, not layer by layer)
  e->w_ref[i].assign(w, w + e->w_ref[i].size());
  e->b_ref[i].assign(b, b + e->b_ref[i].size());
  HIPCHK(e, hipDeviceSynchronize());
  if (e->convs[i].h8) {  // the fp8 scale is shared with the other branch of the launch: re-pack both when it moves
    std::vector<int> old(e->convs.size());
    for (size_t j = 0; j < e->convs.size(); ++j) old[j] = e->convs[j].wq_exp;
    compute_wq_exp(e);
    bool moved = false;
    for (size_t j = 0; j < e->convs.size(); ++j) {
      moved = moved || e->convs[j].wq_exp != old[j];
      if ((int)j != i && e->convs[j].wq_exp != old[j] && (rc = upload_conv_weights(e, (int)j))) return rc;
    }
    if (moved && (rc = invalidate_graphs(e))) return rc;  // ConvParams::wq_exp is a by-value argument baked into the captured graphs
  }
  return upload_conv_weights(e, i);
}
int rtp_save_caffemodel(const rtp_engine* e, const char* path) {
  if (!e || !path) return RTP_EINVAL;
  std::vector<LayerWeights> lw;
  for (size_t i = 0; i < e->convs.size(); ++i) {
    const ConvOp& c = e->convs[i];
    LayerWeights L;
    L.name = c.name; L.type = "Convolution";
    BlobData w, b;
    w.shape = {c.cout, c.cin, c.k, c.k};
    w.data = e->w_ref[i];
    b.shape = {c.cout};
    b.data = e->b_ref[i];
    L.blobs = {w, b};
    lw.push_back(L);
  }
  std::string err;
  if (!write_caffemodel(path, e->net.name, lw, &err)) return fail(const_cast<rtp_engine*>(e), RTP_EIO, "%s", err.c_str());
  return RTP_OK;
}
int rtp_save_prototxt(const rtp_engine* e, const char* path) {
  if (!e || !path) return RTP_EINVAL;
  std::ofstream f(path);
  if (!f) return fail(const_cast<rtp_engine*>(e), RTP_EIO, "cannot create %s", path);
  f << emit_prototxt(e->net);
  return f ? RTP_OK : RTP_EIO;
}

int rtp_prototxt_summary(const char* path, int* num_layers, int* num_conv, int* num_parts, int* max_peaks, float* nms_threshold,
                         int* heat_channels) {
  if (!path) return RTP_EINVAL;
  std::ifstream f(path);
  if (!f) return fail(nullptr, RTP_EIO, "cannot open %s", path);
  std::stringstream ss;
  ss << f.rdbuf();
  NetDef n;
  std::string err;
  if (!parse_prototxt(ss.str(), &n, &err)) return fail(nullptr, RTP_EIO, "%s", err.c_str());
  int nc = 0, np = 0, mp = 0, hc = 0;
  float thr = 0;
  std::map<std::string, int> ch;
  for (auto& i : n.inputs) ch[i] = 3;
  for (auto& L : n.layers) {
    if (L.type == "Convolution") { ++nc; ch[L.tops[0]] = L.num_output; }
    else if (L.type == "Pooling") ch[L.tops[0]] = ch[L.bottoms[0]];
    else if (L.type == "Concat") { int c = 0; for (auto& b : L.bottoms) c += ch[b]; ch[L.tops[0]] = c; }
    else if (L.type == "ImResize") hc = ch[L.bottoms[0]];
    else if (L.type == "Nms") { np = L.num_parts; mp = L.max_peaks; thr = L.nms_threshold; }
  }
  if (num_layers) *num_layers = (int)n.layers.size();
  if (num_conv) *num_conv = nc;
  if (num_parts) *num_parts = np;
  if (max_peaks) *max_peaks = mp;
  if (nms_threshold) *nms_threshold = thr;
  if (heat_channels) *heat_channels = hc;
  return RTP_OK;
}

// Diagnostics of the last synchronous connect on context 0: survivors of the PAF test and accepted
// connections per limb (what `temp.size()` / `connection_k.size()` were, rtpose.cpp:950,980).
int rtp_debug_connect_stats(rtp_engine* e, int* cand_count, int* conn_count) {
  int rc;
  if ((rc = need_idle(e))) return rc;
  Ctx& cx = e->ctx[0];
  if (cand_count) HIPCHK(e, hipMemcpy(cand_count, cx.slot[0].cand_count, e->num_limbs * sizeof(int), hipMemcpyDeviceToHost));
  if (conn_count) HIPCHK(e, hipMemcpy(conn_count, cx.slot[0].conn_count, e->num_limbs * sizeof(int), hipMemcpyDeviceToHost));
  return RTP_OK;
}

// ---- host-only weight utilities ------------------------------------------------------------------
int rtp_synth_weights(uint64_t seed, const char* layer_name, int cout, int cin, int k, float* w, float* b) {
  if (!layer_name || !w || !b || cout < 1 || cin < 1 || k < 1) return RTP_EINVAL;
  std::vector<float> vw, vb;
  synth_conv_weights(seed, layer_name, cout, cin, k, &vw, &vb);
  memcpy(w, vw.data(), vw.size() * sizeof(float));
  memcpy(b, vb.data(), vb.size() * sizeof(float));
  return RTP_OK;
}

int rtp_write_synthetic_caffemodel(int model, uint64_t seed, const char* path) {
  if (!path || (model != RTP_MODEL_COCO_18 && model != RTP_MODEL_MPI_15)) return RTP_EINVAL;
  const NetDef net = build_linevec(model);
  std::map<std::string, int> ch;
  ch["image"] = 3;
  std::vector<LayerWeights> lw;
  for (auto& L : net.layers) {
    if (L.type == "Convolution") {
      const int cin = ch[L.bottoms[0]];
      LayerWeights W;
      W.name = L.name; W.type = "Convolution";
      BlobData w, b;
      synth_conv_weights(seed, L.name, L.num_output, cin, L.kernel, &w.data, &b.data);
      w.shape = {L.num_output, cin, L.kernel, L.kernel};
      b.shape = {L.num_output};
      W.blobs = {w, b};
      lw.push_back(W);
      ch[L.tops[0]] = L.num_output;
    } else if (L.type == "Pooling") ch[L.tops[0]] = ch[L.bottoms[0]];
    else if (L.type == "Concat") { int c = 0; for (auto& bn : L.bottoms) c += ch[bn]; ch[L.tops[0]] = c; }
  }
  std::string err;
  if (!write_caffemodel(path, net.name, lw, &err)) return fail(nullptr, RTP_EIO, "%s", err.c_str());
  return RTP_OK;
}

int rtp_write_builtin_prototxt(int model, const char* path) {
  if (!path || (model != RTP_MODEL_COCO_18 && model != RTP_MODEL_MPI_15)) return RTP_EINVAL;
  std::ofstream f(path);
  if (!f) return fail(nullptr, RTP_EIO, "cannot create %s", path);
  f << emit_prototxt(build_linevec(model));
  return f ? RTP_OK : RTP_EIO;
}

int rtp_caffemodel_layer(const char* path, int index, char* name, int name_len, int* num_blobs, long* count0, long* count1,
                         float* head0 /* first min(8,count0) floats of blob 0 */) {
  if (!path) return RTP_EINVAL;
  static thread_local std::string cached_path;
  static thread_local std::vector<LayerWeights> cached;
  if (cached_path != path) {
    std::vector<LayerWeights> lw;
    std::string err;
    if (!read_caffemodel(path, &lw, &err)) return fail(nullptr, RTP_EIO, "%s", err.c_str());
    cached.swap(lw);
    cached_path = path;
  }
  if (index < 0) return (int)cached.size();
  if (index >= (int)cached.size()) return RTP_EINVAL;
  const LayerWeights& L = cached[index];
  if (name && name_len > 0) snprintf(name, name_len, "%s", L.name.c_str());
  if (num_blobs) *num_blobs = (int)L.blobs.size();
  if (count0) *count0 = L.blobs.size() > 0 ? (long)L.blobs[0].data.size() : 0;
  if (count1) *count1 = L.blobs.size() > 1 ? (long)L.blobs[1].data.size() : 0;
  if (head0 && !L.blobs.empty())
    for (size_t i = 0; i < 8 && i < L.blobs[0].data.size(); ++i) head0[i] = L.blobs[0].data[i];
  return RTP_OK;
}

// Build the execution plan for cfg WITHOUT touching a device and describe it as text (tensors,
// per-layer tile configuration, branch pairing, arena sizes).  Host logic only.
static long plan_summary_impl(const rtp_config* cfg, char* buf, size_t buflen) {
  if (!cfg || !buf) return RTP_EINVAL;
  rtp_engine* e = new rtp_engine();
  e->cfg = *cfg;
  if (cfg->precision < RTP_PREC_FP16 || cfg->precision > RTP_PREC_F16X3) { delete e; return fail(nullptr, RTP_EINVAL, "unknown precision %d", cfg->precision); }
  e->mode = cfg->precision;
  e->prec = cfg->precision == RTP_PREC_FP32 ? 1 : 0;
  e->elem = e->prec ? 4 : 2;
  {
    const char* sr = RTP_EXP_ENV("RTP_SPLIT_LAYERS");
    e->split_rules = sr ? sr : (cfg->split_layers ? cfg->split_layers : kDefaultSplit);
    const char* f8 = RTP_EXP_ENV("RTP_SPLIT_FP8");
    e->split_fp8 = !(f8 && f8[0] == '0');
  }
  e->N = cfg->num_scales;
  e->B = cfg->batch_frames < 1 ? 1 : cfg->batch_frames;
  e->NI = e->N * e->B;
  if (cfg->proto_path) {
    std::ifstream f(cfg->proto_path);
    std::stringstream ss;
    std::string perr;
    if (!f) { delete e; return fail(nullptr, RTP_EIO, "cannot open prototxt %s", cfg->proto_path); }
    ss << f.rdbuf();
    if (!parse_prototxt(ss.str(), &e->net, &perr)) { delete e; return fail(nullptr, RTP_EIO, "%s", perr.c_str()); }
  } else {
    if (cfg->model != RTP_MODEL_COCO_18 && cfg->model != RTP_MODEL_MPI_15) { delete e; return fail(nullptr, RTP_EINVAL, "unknown model %d", cfg->model); }
    e->net = build_linevec(cfg->model);
  }
  const int rc = build_plan(e);
  if (rc) { g_create_error = e->err; delete e; return rc; }
  std::ostringstream o;
  o << "model " << e->model << " parts " << e->num_parts << " max_peaks " << e->max_peaks << " heat_channels " << e->heat_channels << "\n";
  for (int l = 0; l < e->nlevels; ++l)
    o << "level " << l << " H " << e->geom[l].H << " W " << e->geom[l].W << " halo " << e->geom[l].halo << "\n";
  o << "arena_bytes " << e->arena_bytes << " weights_bytes " << e->weights_bytes << " tensors " << e->tensors.size() << "\n";
  {  // which streams a batch context gets (alloc_ctx; "hardware queues" above): one for everything when the runtime's hardware queues suffice
    const int nctx = (e->cfg.frames_in_flight + e->B - 1) / e->B + (e->B > 1 ? 1 : 0);
    o << "streams contexts " << nctx << " hw_queues " << hw_queue_count() << " arrangement " << ((nctx <= hw_queue_count() || e->B == 1) ? "one_per_context" : "per_frame_chains") << "\n";
  }
  double gflop = 0, mfma_gflop = 0;
  for (auto& s : e->steps) {
    if (s.type == 0) o << "step pack\n";
    else if (s.type == 4) {
      const ConvOp& c = e->convs[s.a];
      const Geom& g = e->geom[c.level];
      o << "step first " << c.name << " k 3 cin 3 cout " << c.cout << " relu " << c.relu << " passes 1 wgs " << (long)g.H * e->NI << "\n";
      const double gf = 2.0 * c.cout * c.cin * c.k * c.k * (double)g.H * g.W * e->N * 1e-9;
      gflop += gf;
      mfma_gflop += gf;
    } else if (s.type == 2) o << "step pool " << e->tensors[e->pools[s.a].in_tensor].name << " -> " << e->tensors[e->pools[s.a].out_tensor].name << "\n";
    else if (s.type == 3) {
      const ConvOp& A = e->convs[s.a];
      const ConvOp& C = e->convs[s.a2];
      const Geom& g = e->geom[A.level];
      o << "step pw2 " << A.name;
      if (s.b >= 0) o << " + " << e->convs[s.b].name;
      o << " -> " << C.name;
      if (s.b2 >= 0) o << " + " << e->convs[s.b2].name;
      o << " k 1 cin_p " << A.Cin_p << " mid " << A.cout << " cout " << C.cout << " passes " << A.passes() << (A.split_a ? "a" : "") << (A.split_w ? "w" : "") << "/"
        << C.passes() << (C.split_a ? "a" : "") << (C.split_w ? "w" : "") << " tile 64 wgs " << (((long)g.H * g.Wp + 63) / 64) * e->NI * (s.b >= 0 ? 2 : 1)
        << " lowres " << C.to_lowres << "\n";
      for (int idx : {s.a, s.b, s.a2, s.b2}) if (idx >= 0) {
        const ConvOp& c = e->convs[idx];
        const double gf = 2.0 * c.cout * c.cin * (double)g.H * g.W * e->N * 1e-9;
        gflop += gf;
        mfma_gflop += gf * c.passes();
      }
    } else {
      const ConvOp& A = e->convs[s.a];
      const ConvCfgInfo ci = conv_cfg_info(A.cfg);
      const Geom& g = e->geom[A.level];
      const long tiles = A.pool >= 0 ? ((long)(g.H / 2) * ((g.W + A.k_eff / 2 + 1) & ~1) + ci.BM / 2 - 1) / (ci.BM / 2) : ((long)g.H * g.Wp + ci.BM - 1) / ci.BM;
      o << "step conv " << A.name;
      if (A.pool >= 0) o << " +pool";
      if (s.b >= 0) o << " + " << e->convs[s.b].name;
      o << " k " << A.k << " cin_p " << A.Cin_p << " cout " << A.cout << " coutp " << A.CoutP << " relu " << A.relu << " tile " << ci.BM << "x" << ci.BN
        << " rowb " << A.rowb << " passes " << A.passes() << (A.h8 ? "q" : "") << (!A.h8 && A.split_a ? "a" : "") << (!A.h8 && A.split_w ? "w" : "") << " impl " << (A.impl ? "ring" : "reg") << " wgs " << tiles * e->NI * (A.CoutP / ci.BN) * (s.b >= 0 ? 2 : 1) << " dsts " << A.dsts.size() << " lowres " << A.to_lowres << "\n";
      for (int idx : {s.a, s.b}) if (idx >= 0) {
        const ConvOp& c = e->convs[idx];
        const double gf = 2.0 * c.cout * c.cin * c.k * c.k * (double)g.H * g.W * e->N * 1e-9;
        gflop += gf;
        mfma_gflop += gf * c.passes();
      }
    }
  }
  o << "conv_gflop " << gflop << "\n";
  o << "mfma_gflop " << mfma_gflop << "\n";
  delete e;
  const std::string str = o.str();
  if (str.size() + 1 > buflen) return RTP_ERANGE;
  memcpy(buf, str.c_str(), str.size() + 1);
  return (long)str.size();
}

int rtp_kernel_timing(rtp_engine* e, int enable, double* total_ms, long* launches, double* flops_per_launch) {
  SYNC_GUARD;
  if (!e) return RTP_EINVAL;
  int rc;
  if ((rc = use_device(e))) return rc;
  // harvest the event pairs: each brackets ONE dominant-class launch on the stream it ran on
  if (e->tev_next > 0 && e->fifo.empty()) {
    HIPCHK(e, hipDeviceSynchronize());
    for (int i = 0; i < e->tev_next; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e->tev[2 * (size_t)i], e->tev[2 * (size_t)i + 1]) == hipSuccess && ms > 0.f) {
        const int si = e->tev_step[i];
        if (si >= 0 && si < (int)e->steps.size()) {
          if (e->step_ms.size() != e->steps.size()) { e->step_ms.assign(e->steps.size(), 0.0); e->step_n.assign(e->steps.size(), 0); }
          e->step_ms[si] += ms; e->step_n[si]++;
        }
        if (si < 0 || si >= (int)e->steps.size() || is_dominant_class(e, e->steps[si])) {
          e->dom_ms_total += ms; e->dom_launches++;
          e->dom_ms_pass[e->tev_pass[i] & 3] += ms; e->dom_n_pass[e->tev_pass[i] & 3]++;
        }
      }
    }
    e->tev_next = 0;
  }
  if (total_ms) *total_ms = e->dom_ms_total;
  if (launches) *launches = e->dom_launches;
  if (flops_per_launch) {
    double fl = 0;
    if (e->dominant_step >= 0) {
      const Step& s = e->steps[e->dominant_step];
      const Geom& g = e->geom[e->convs[s.a].level];
      for (int idx : {s.a, s.b}) if (idx >= 0) { const ConvOp& c = e->convs[idx]; fl += 2.0 * c.cout * c.cin * c.k * c.k * (double)g.H * g.W * e->NI; }
    }
    *flops_per_launch = fl;
  }
  if (enable >= 0) {
    if ((enable != 0) != e->time_dominant && !e->fifo.empty()) return fail(e, RTP_EAGAIN, "kernel timing can only be switched on an idle engine");
    if ((enable != 0) != e->time_dominant || enable >= 2) {  // 2 = on + reset; 3 = on + reset, every step of the plan
      e->dom_ms_total = 0; e->dom_launches = 0;
      for (int i = 0; i < 4; ++i) { e->dom_ms_pass[i] = 0; e->dom_n_pass[i] = 0; }
      e->step_ms.assign(e->steps.size(), 0.0); e->step_n.assign(e->steps.size(), 0);
    }
    e->time_dominant = enable != 0;
    e->time_all = enable == 3;
    if (e->time_dominant) {
      if (e->tev.empty()) {
        e->tev.assign((size_t)2 * rtp_engine::TEV_PAIRS, nullptr);
        // time stamps only, nothing waits for them: no system-scope release fence in the marker (a default event record writes the L2s back
        // before it stamps: 1.8 us of every pair that do not belong to the launch; profiles/r06_experiments.txt)
        for (hipEvent_t& ev : e->tev) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));
        e->tev_pass.assign(rtp_engine::TEV_PAIRS, 1);
        e->tev_step.assign(rtp_engine::TEV_PAIRS, -1);
      }
      e->tev_next = 0;
    }
  }
  return RTP_OK;
}

// Host float -> OCP e4m3 conversion used for the fp8 weight copies (tests compare it with torch.float8_e4m3fn).
int rtp_debug_f32_to_e4m3(const float* in, unsigned char* out, int n) {
  if (!in || !out || n < 0) return RTP_EINVAL;
  for (int i = 0; i < n; ++i) out[i] = f32_to_e4m3(in[i]);
  return RTP_OK;
}

// Stream-busy account without a profiler.  enable = 1: on + reset (idle engine), 0: off, -1: read only.  spans (may be NULL): up to `cap`
// triples {kind, start_ms, end_ms} since the probe was switched on — kind 0: a batch on its conv stream, from the staging of its first
// input (H2D / pre-processing) to the end of its conv stack; kind 1: one frame's post-processing chain incl. the D2H of its joints.
// Returns the number of triples recorded.  The union of the spans over the wall between the first start and the last end is the share
// of the time the engine had work on the GPU; 1 - that is time in which NO stream of the engine had anything to run.
int rtp_busy_probe(rtp_engine* e, int enable, float* spans, int cap) {
  if (!e) return RTP_EINVAL;
  int rc;
  if ((rc = use_device(e))) return rc;
  if (enable >= 0) {
    if (!e->fifo.empty()) return fail(e, RTP_EAGAIN, "the busy probe can only be switched on an idle engine");
    if (enable == 1) {
      if (!e->busy_base) HIPCHK(e, hipEventCreate(&e->busy_base));
      e->busy_spans.clear();
      for (Ctx& cx : e->ctx) cx.stage0_set = false;
      HIPCHK(e, hipEventRecord(e->busy_base, e->ctx[0].stream));
      HIPCHK(e, hipEventSynchronize(e->busy_base));
    }
    e->busy_probe = enable == 1;
  }
  const int n = (int)(e->busy_spans.size() / 3);
  if (spans) memcpy(spans, e->busy_spans.data(), sizeof(float) * 3 * (size_t)std::min(n, std::max(cap, 0)));
  return n;
}

// Kernel residency without a profiler.  enable = 1: on + reset (idle engine; the batch graphs are re-captured with the
// stamp slots in their launches), 0: off (graphs re-captured without), -1: harvest every idle context and read.  spans (may be NULL): up to
// `cap` triples {slot, start_us, end_us}: for every kernel launch of the per-frame path since the probe was switched on, the wall-clock time
// (100 MHz, one clock for all XCDs) at which its FIRST workgroup started and its LAST workgroup ended, written by the workgroups themselves
// (kernels.h KStamp).  slot: < 64 = plan step (rtp_plan_summary order), 64 + 8 j + k = frame j's strip / write / pairs / match / assemble
// kernel, 200 + 2 j + k = frame j's warp / area-pad kernel.  The union of the spans over the wall = the share of the time in which at least one
// kernel was resident on the chip.  Returns the number of triples.
int rtp_stamp_probe(rtp_engine* e, int enable, float* spans, int cap) {
  if (!e) return RTP_EINVAL;
  int rc;
  if ((rc = use_device(e))) return rc;
  if (enable >= 0) {
    if (!e->fifo.empty()) return fail(e, RTP_EAGAIN, "the stamp probe can only be switched on an idle engine");
    if ((enable == 1) != e->stamp_probe || enable == 1) {
      if ((rc = invalidate_graphs(e))) return rc;   // the slot pointers are launch arguments: baked into the captured graphs
      e->stamp_probe = enable == 1;
      for (Ctx& cx : e->ctx) {
        if (!cx.stamps) continue;
        HIPCHK(e, hipMemset(cx.stamps, 0, 2 * STAMP_SLOTS * sizeof(unsigned long long)));
        cx.stamps_dirty = false;
      }
      if (enable == 1) { e->stamp_spans.clear(); e->stamp_base = 0; }
    }
  } else if (e->fifo.empty()) {
    for (Ctx& cx : e->ctx) if ((rc = stamp_harvest(e, cx))) return rc;
  }
  const int n = (int)(e->stamp_spans.size() / 3);
  if (spans) memcpy(spans, e->stamp_spans.data(), sizeof(float) * 3 * (size_t)std::min(n, std::max(cap, 0)));
  return n;
}

// What the measurement probes could NOT record since engine creation: out[0] = launches rtp_kernel_timing wanted to time after its event
// table (TEV_PAIRS pairs between two harvests) was full, out[1] = frames whose rtp_busy_probe spans were dropped at the 65536-triple cap,
// out[2] = frames the busy probe skipped because their batch was a whole-batch graph replay (no per-frame events exist there).
int rtp_probe_dropped(const rtp_engine* e, long out[3]) {
  if (!e || !out) return RTP_EINVAL;
  for (int i = 0; i < 3; ++i) out[i] = e->probe_dropped[i];
  return RTP_OK;
}

// Per-step totals of a timing pass switched on with rtp_kernel_timing(e, 3, ..): ms[i] / launches[i] for plan step i (the order of
// rtp_plan_summary's "step" lines); returns the number of steps.  Harvest first with rtp_kernel_timing(e, -1, ..) on an idle engine.
int rtp_kernel_timing_steps(const rtp_engine* e, double* ms, long* launches, int cap) {
  if (!e || !ms || !launches || cap < 0) return RTP_EINVAL;
  const int n = (int)e->steps.size();
  for (int i = 0; i < n && i < cap; ++i) {
    ms[i] = i < (int)e->step_ms.size() ? e->step_ms[i] : 0.0;
    launches[i] = i < (int)e->step_n.size() ? e->step_n[i] : 0;
  }
  return n;
}

// Per-pass-count breakdown of rtp_kernel_timing's totals: ms[p], launches[p] for p = 1..3 MFMA passes (index 0 unused).
int rtp_kernel_timing_by_passes(const rtp_engine* e, double ms[4], long launches[4]) {
  if (!e || !ms || !launches) return RTP_EINVAL;
  for (int i = 0; i < 4; ++i) { ms[i] = e->dom_ms_pass[i]; launches[i] = e->dom_n_pass[i]; }
  return RTP_OK;
}

// Diagnostics: every step of the plan alone on the chip, `iters` back-to-back launches at the full
// batch; ms[i] = average launch time of step i, gflop[i] = its convolution work (0 for pack/pool).
int rtp_profile_steps(rtp_engine* e, int iters, float* ms, double* gflop, int cap) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (iters < 1) return RTP_EINVAL;
  Ctx& cx = e->ctx[0];
  auto geom_n = [&](int level) { Geom g = e->geom[level]; g.N = e->NI; return g; };
  int n = 0;
  for (auto& s : e->steps) {
    if (n >= cap) break;
    auto once = [&]() -> int {
      if (s.type == 0) {
        const Tensor& t = e->tensors[0];
        HIPCHK(e, launch_pack_input(e->prec, cx.input, cx.arena + t.offset, geom_n(0), t.stride(), cx.stream));
      } else if (s.type == 1) {
        return launch_conv_step(e, cx, s, e->NI);
      } else if (s.type == 3) {
        return launch_pw2_step(e, cx, s, e->NI);
      } else if (s.type == 4) {
        return launch_first_step(e, cx, s, cx.input, e->NI);
      } else {
        const PoolOp& p = e->pools[s.a];
        const Tensor& ti = e->tensors[p.in_tensor];
        const Tensor& to = e->tensors[p.out_tensor];
        HIPCHK(e, launch_maxpool(e->prec, cx.arena + ti.offset, geom_n(ti.level), ti.stride(), cx.arena + to.offset, geom_n(to.level), to.stride(),
                                 round_up(p.C, 16 / e->elem), ti.lo_off(), to.lo_off(), ti.q_off(), to.q_off(), cx.stream));
      }
      return RTP_OK;
    };
    for (int i = 0; i < 2; ++i) if ((rc = once())) return rc;
    HIPCHK(e, hipEventRecord(cx.ev[0], cx.stream));
    for (int i = 0; i < iters; ++i) if ((rc = once())) return rc;
    HIPCHK(e, hipEventRecord(cx.ev[1], cx.stream));
    HIPCHK(e, hipEventSynchronize(cx.ev[1]));
    float t = 0.f;
    HIPCHK(e, hipEventElapsedTime(&t, cx.ev[0], cx.ev[1]));
    if (ms) ms[n] = t / iters;
    double fl = 0;
    if (s.type == 1 || s.type == 3 || s.type == 4) {
      const Geom& g = e->geom[e->convs[s.a].level];
      for (int idx : {s.a, s.b, s.a2, s.b2}) if (idx >= 0) { const ConvOp& c = e->convs[idx]; fl += 2.0 * c.cout * c.cin * c.k * c.k * (double)g.H * g.W * e->NI; }
    }
    if (gflop) gflop[n] = fl * 1e-9;
    ++n;
  }
  return n;
}

int rtp_bench_dominant_conv(rtp_engine* e, int iters, float* avg_ms, double* flops_per_launch) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (e->dominant_step < 0 || iters < 1) return fail(e, RTP_EINVAL, "no 7x7 128->128 convolution step in this graph");
  Ctx& cx = e->ctx[0];
  const Step& s = e->steps[e->dominant_step];
  const ConvOp& A = e->convs[s.a];
  const Geom& g = e->geom[A.level];
  for (int i = 0; i < 3; ++i) if ((rc = launch_conv_step(e, cx, s, e->NI))) return rc;
  HIPCHK(e, hipEventRecord(cx.ev[0], cx.stream));
  for (int i = 0; i < iters; ++i) if ((rc = launch_conv_step(e, cx, s, e->NI))) return rc;
  HIPCHK(e, hipEventRecord(cx.ev[1], cx.stream));
  HIPCHK(e, hipEventSynchronize(cx.ev[1]));
  if (RTP_EXP_ENV("RTP_CLKPROBE")) {  // diagnostics: effective shader clock while this kernel runs back to back
    unsigned long long* d = nullptr;
    HIPCHK(e, hipMalloc((void**)&d, 32));
    HIPCHK(e, hipMemset(d, 0, 32));
    g_clkprobe = d;
    for (int i = 0; i < 20; ++i) if ((rc = launch_conv_step(e, cx, s, e->NI))) return rc;
    g_clkprobe = nullptr;
    HIPCHK(e, hipStreamSynchronize(cx.stream));
    unsigned long long h[4] = {0, 0, 0, 0};
    HIPCHK(e, hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->cfg.device_id);
    if (h[1]) fprintf(stderr, "clkprobe: %llu shader cycles in %llu wall ticks (%d kHz) -> %.0f MHz, workgroup 0 alive %.2f us\n", h[0], h[1], khz,
                      (double)h[0] / (double)h[1] * khz / 1e3, (double)h[1] / khz * 1e3);
    if (h[1]) fprintf(stderr, "clkprobe: workgroup 0: first strip + tile landed at %.2f us, K loop done at %.2f us, epilogue done at %.2f us\n",
                      (double)h[2] / khz * 1e3, (double)h[3] / khz * 1e3, (double)h[1] / khz * 1e3);
  }
  float ms = 0.f;
  HIPCHK(e, hipEventElapsedTime(&ms, cx.ev[0], cx.ev[1]));
  if (avg_ms) *avg_ms = ms / iters;
  const int nprob = s.b >= 0 ? 2 : 1;
  double fl = 0;
  for (int idx : {s.a, s.b}) if (idx >= 0) { const ConvOp& c = e->convs[idx]; fl += 2.0 * c.cout * c.cin * c.k * c.k * (double)g.H * g.W * e->NI; }
  (void)nprob;
  if (flops_per_launch) *flops_per_launch = fl;
  return RTP_OK;
}

// ---- caller-owned device buffers ------------------------------------------------------------------
// Replaces: blobs()[0]->mutable_gpu_data() as the H2D target a caller fills itself (rtpose.cpp:1131): a device buffer on the ENGINE's
// device from the ENGINE's HIP runtime, for rtp_submit_device.  (A process that also loads another HIP runtime — torch's wheel bundles
// its own — must not hand that runtime's pointers to this library and expect the two to agree about streams; bench.py allocates here.)
int rtp_device_alloc(rtp_engine* e, size_t bytes, void** dptr) {
  SYNC_GUARD;
  if (!e || !dptr || bytes == 0) return RTP_EINVAL;
  int rc;
  if ((rc = use_device(e))) return rc;
  void* p = nullptr;
  const hipError_t s = hipMalloc(&p, bytes);
  if (s != hipSuccess) return fail(e, RTP_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(s));
  e->user_bufs.push_back(p);
  *dptr = p;
  return RTP_OK;
}
int rtp_device_free(rtp_engine* e, void* dptr) {
  SYNC_GUARD;
  if (!e || !dptr) return RTP_EINVAL;
  auto it = std::find(e->user_bufs.begin(), e->user_bufs.end(), dptr);
  if (it == e->user_bufs.end()) return fail(e, RTP_EINVAL, "rtp_device_free: not a buffer of this engine");
  int rc;
  if ((rc = use_device(e))) return rc;
  e->user_bufs.erase(it);
  HIPCHK(e, hipFree(dptr));
  return RTP_OK;
}
int rtp_device_upload(rtp_engine* e, void* dst_device, const void* src_host, size_t bytes) {
  SYNC_GUARD;
  if (!e || !dst_device || !src_host) return RTP_EINVAL;
  int rc;
  if ((rc = use_device(e))) return rc;
  HIPCHK(e, hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
  return RTP_OK;
}
int rtp_device_synchronize(rtp_engine* e) {
  SYNC_GUARD;
  if (!e) return RTP_EINVAL;
  int rc;
  if ((rc = use_device(e))) return rc;
  HIPCHK(e, hipDeviceSynchronize());
  return RTP_OK;
}

// The CPUs next to a device (its PCI function's local_cpulist in sysfs, e.g. "0-31,128-159"): rtpose.bin pins worker g's thread there so
// that the pinned staging buffers it allocates and fills (rtp_submit_frame's copy of the 2.76 MB frame) live on the GPU's NUMA node.
// Returns the length written, 0 when the topology is not exposed (containers), or a negative code.
int rtp_device_local_cpus(int device_id, char* buf, size_t buflen) {
  if (!buf || buflen < 2) return RTP_EINVAL;
  buf[0] = 0;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof bus, device_id) != hipSuccess) return RTP_ENODEV;
  for (char* c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
  const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
  std::ifstream f(path);
  if (!f) return 0;
  std::string line;
  std::getline(f, line);
  while (!line.empty() && isspace((unsigned char)line.back())) line.pop_back();
  if (line.size() + 1 > buflen) return RTP_ERANGE;
  memcpy(buf, line.c_str(), line.size() + 1);
  return (int)line.size();
}

// ---- load-time precision calibration ----------------------------------------------------------------
// The default split set of RTP_PREC_MIXED was chosen on one synthetic weight set.  Trained weights arrive through
// CopyTrainedLayersFrom (net.cpp:750-803) with another spectrum; this measures the set on the weights that are LOADED:
//   reference  = the same frames through RTP_PREC_F16X3 (every layer as hi + lo fp16 pairs: 1e-5 of the map maximum vs fp32),
//   candidate  = the current mixed plan;  err = max |candidate - reference| / max |reference|  over the final maps.
// While err > target, every layer group that is not split yet is tried on top of the current set and the one that lowers the
// error most is promoted; when every group is in and the error still exceeds the target the engine falls back to F16X3.
int rtp_calibrate_precision(rtp_engine* e, const float* frames_host, int nframes, float target, char* rules_out, size_t rules_len,
                            float* err_before, float* err_after) {
  // (no process-wide lock across this function: every re-plan and every tap below takes g_sync_mutex for its own captures / synchronous
  // calls, so other engines of the process — rtpose.bin --num_gpu N --calibrate K — interleave with the trials instead of queueing behind
  // the dozens of seconds a calibration can take)
  int rc;
  if ((rc = need_idle(e))) return rc;
  if ((rc = need_weights(e))) return rc;
  if (e->mode == RTP_PREC_F16X3 && e->calib_fell_back) {   // an earlier calibration ended in the parity-grade mode: nothing is left to adjust
    if (err_before) *err_before = 0.f;
    if (err_after) *err_after = 0.f;
    if (rules_out && rules_len) snprintf(rules_out, rules_len, "@f16x3");
    e->calib_report += "; called again: already RTP_PREC_F16X3 (every layer runs three fp16 passes), nothing to adjust";
    return RTP_OK;
  }
  if (e->mode != RTP_PREC_MIXED) return fail(e, RTP_EINVAL, "rtp_calibrate_precision adjusts the split set of RTP_PREC_MIXED (engine precision is %d)", e->mode);
  if (nframes < 1 || nframes > 64) return fail(e, RTP_EINVAL, "calibration frames %d out of range [1, 64]", nframes);
  if (!(target > 0.f)) target = 0.7e-3f;
  const size_t in_floats = (size_t)e->N * 3 * e->cfg.net_h * e->cfg.net_w;
  const size_t low_floats = (size_t)e->N * e->heat_channels * e->low_h * e->low_w;
  std::vector<float> synth;
  if (!frames_host) {  // what process_and_pad_image makes of a u8 frame: v / 256 - 0.5 (rtpose.cpp:259), seeded
    synth.resize(in_floats * nframes);
    uint64_t st = 0x9E3779B97F4A7C15ull ^ e->cfg.synthetic_seed;
    for (float& v : synth) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = (float)((st >> 56) & 0xff) / 256.0f - 0.5f; }
    frames_host = synth.data();
  }
  std::vector<float> ref(low_floats * nframes), got(low_floats);
  auto run = [&](const float* x, float* out) { return rtp_forward_heatmaps(e, x, out); };
  const std::string base_rules = e->split_rules;
  // a failure half way (a trial plan that does not fit the device, a HIP error) must not leave the caller with a one-context trial engine:
  // put the plan it came with back (best effort) and report the original error
  auto bail = [&](int code) {
    const std::string msg = e->err;
    if (replan(e, RTP_PREC_MIXED, base_rules, false) != RTP_OK) {
      // the restore failed too (e.g. the same out-of-memory condition): the engine has NO plan.  Mark it: every entry point that would
      // touch a context reports RTP_EHIP instead of dereferencing an empty vector; rtp_engine_destroy still works.
      drop_plan(e);
      e->broken = true;
      e->err = msg + " (and the engine's previous plan could not be restored: the engine is unusable, destroy it)";
      return code;
    }
    e->err = msg;
    return code;
  };
  // reference maps
  if ((rc = replan(e, RTP_PREC_F16X3, base_rules, true))) return bail(rc);
  for (int f = 0; f < nframes; ++f)
    if ((rc = run(frames_host + (size_t)f * in_floats, ref.data() + (size_t)f * low_floats))) return bail(rc);
  double norm = 0;
  for (float v : ref) norm = std::max(norm, (double)std::fabs(v));
  if (!(norm > 0) || !std::isfinite(norm)) {
    if (bail(RTP_ERANGE) && e->broken) return RTP_ERANGE;
    return fail(e, RTP_ERANGE, "calibration: the reference maps are %s (weights / frames out of range for fp16 storage?)", norm > 0 ? "not finite" : "all zero");
  }
  auto measure = [&](double* err) -> int {
    double m = 0;
    for (int f = 0; f < nframes; ++f) {
      int r2 = run(frames_host + (size_t)f * in_floats, got.data());
      if (r2) return r2;
      const float* rf = ref.data() + (size_t)f * low_floats;
      for (size_t i = 0; i < low_floats; ++i) {
        const double d = std::fabs((double)got[i] - (double)rf[i]);
        m = (d == d) ? std::max(m, d) : 1e30;   // NaN counts as unbounded
      }
    }
    *err = m / norm;
    return RTP_OK;
  };
  std::ostringstream rep;
  std::string rules = base_rules;
  double err = 0;
  if ((rc = replan(e, RTP_PREC_MIXED, rules, true))) return bail(rc);
  if ((rc = measure(&err))) return bail(rc);
  if (err_before) *err_before = (float)err;
  rep << "target " << target << "; frames " << nframes << "; set \"" << rules << "\" err " << err;
  // the layer groups a rule can name: trunk blocks by their "convN_" prefix, stage 1, the refinement stages (from the MIXED plan's flags)
  std::vector<std::string> all_groups;
  {
    auto add = [&](const std::string& g) { if (std::find(all_groups.begin(), all_groups.end(), g) == all_groups.end()) all_groups.push_back(g); };
    for (auto& c : e->convs) {
      const size_t st = c.name.find("_stage");
      if (st != std::string::npos) { size_t en = st + 6; while (en < c.name.size() && isdigit((unsigned char)c.name[en])) ++en; add("*" + c.name.substr(st, en - st) + "_"); }
      else { const size_t us = c.name.find('_'); add(us == std::string::npos ? c.name : c.name.substr(0, us + 1)); }
    }
  }
  auto in_group = [](const std::string& g, const ConvOp& c) { return g[0] == '*' ? c.name.find(g.substr(1)) != std::string::npos : c.name.compare(0, g.size(), g) == 0; };
  auto open_groups = [&](bool want_h8) {   // groups with a layer that is not fully split yet (stage 1) / that still runs an fp8 chunk (stage 2)
    std::vector<std::string> gs;
    for (auto& g : all_groups) {
      bool open = false;
      for (auto& c : e->convs) if (in_group(g, c) && (want_h8 ? c.h8 : !(c.split_w && (c.split_a || c.first)))) open = true;
      if (open) gs.push_back(g);
    }
    return gs;
  };
  // stage 1: promote un-split groups, the one that lowers the error most first
  std::vector<std::string> groups = open_groups(false);
  while (err > target && !groups.empty()) {
    int best = -1;
    double best_err = 1e300;
    for (size_t g = 0; g < groups.size(); ++g) {
      double eg = 0;
      if ((rc = replan(e, RTP_PREC_MIXED, rules + "," + groups[g], true))) return bail(rc);
      if ((rc = measure(&eg))) return bail(rc);
      rep << "; try +" << groups[g] << " -> " << eg;
      if (eg < best_err) { best_err = eg; best = (int)g; }
    }
    rules += "," + groups[best];
    rep << "; promote " << groups[best];
    groups.erase(groups.begin() + best);
    err = best_err;
  }
  // stage 2: every group is split and the target is still missed — the fp8 correction chunks are the limit (e4m3 operands: activations
  // beyond +-112 or rounding errors beyond their range saturate; 3 mantissa bits).  Groups switch to fp16 correction passes (":x"),
  // ranked by what the switch gains alone, applied cumulatively until the target holds.
  if (err > target) {
    if ((rc = replan(e, RTP_PREC_MIXED, rules, true))) return bail(rc);
    std::vector<std::string> hg = open_groups(true);
    std::vector<std::pair<double, std::string>> gain;
    for (auto& g : hg) {
      double eg = 0;
      if ((rc = replan(e, RTP_PREC_MIXED, rules + "," + g + ":x", true))) return bail(rc);
      if ((rc = measure(&eg))) return bail(rc);
      rep << "; try " << g << ":x -> " << eg;
      gain.push_back({eg, g});
    }
    std::sort(gain.begin(), gain.end());
    for (auto& ge : gain) {
      if (err <= target) break;
      rules += "," + ge.second + ":x";
      if ((rc = replan(e, RTP_PREC_MIXED, rules, true))) return bail(rc);
      if ((rc = measure(&err))) return bail(rc);
      rep << "; switch " << ge.second << ":x -> " << err;
    }
  }
  int final_mode = RTP_PREC_MIXED;
  if (err > target) {   // every layer runs three fp16 passes already where it can: the parity-grade mode for all of them
    final_mode = RTP_PREC_F16X3;
    rep << "; err " << err << " > target with every group switched: falling back to RTP_PREC_F16X3";
    err = 0;
  }
  if ((rc = replan(e, final_mode, final_mode == RTP_PREC_MIXED ? rules : base_rules, false))) return bail(rc);
  e->calib_fell_back = final_mode == RTP_PREC_F16X3;   // (rtp_get_split_layers then reports precision RTP_PREC_F16X3 next to the caller's own rule list: the rules do not apply in that mode)
  if (final_mode == RTP_PREC_MIXED && (rc = measure(&err))) return bail(rc);
  if (err_after) *err_after = (float)err;
  rep << "; final " << (final_mode == RTP_PREC_MIXED ? "mixed" : "f16x3") << " set \"" << rules << "\" err " << err;
  e->calib_report = rep.str();
  if (rules_out && rules_len) snprintf(rules_out, rules_len, "%s", final_mode == RTP_PREC_MIXED ? rules.c_str() : "@f16x3");
  return RTP_OK;
}
const char* rtp_calibration_report(const rtp_engine* e) { return e ? e->calib_report.c_str() : ""; }
// The split set in force (rule list of rtp_config.split_layers syntax) and the precision mode (a calibration may have changed both).
int rtp_get_split_layers(const rtp_engine* e, char* buf, size_t buflen, int* precision) {
  if (!e) return RTP_EINVAL;
  if (precision) *precision = e->mode;
  if (buf && buflen) {
    if (e->split_rules.size() + 1 > buflen) {   // never a silently truncated rule list: the receiver would build another plan
      buf[0] = 0;
      return fail(const_cast<rtp_engine*>(e), RTP_ERANGE, "rtp_get_split_layers: the rule list needs %zu bytes, the buffer has %zu", e->split_rules.size() + 1, buflen);
    }
    snprintf(buf, buflen, "%s", e->split_rules.c_str());
  }
  return RTP_OK;
}

// ---- one-time weight distribution (optional; nothing here is on the per-frame path) -------------------------
// The reference reads the .caffemodel once per GPU thread (rtpose.cpp:183-184: NUM_GPU disk reads + NUM_GPU H2D copies).  The engine
// packs weights into its kernels' staging order on the host, which costs more than the read.  Worker 0 does that once; the others take
// the PACKED arena: rtp_copy_weights_from over xGMI inside one process (rtpose.bin --share_weights), or export -> broadcast -> import
// between processes (bench.py --broadcast_weights: torch.distributed broadcast of the host blob, RCCL when the group is nccl).
// Blob layout: magic, plan hash, nconv, wq_exp[nconv], arena bytes, arena, then per conv the Caffe-layout floats (so that
// rtp_get_conv_weights / rtp_save_caffemodel of the receiver stay truthful).
namespace {
uint64_t plan_hash(const rtp_engine* e) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint64_t v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } };
  mix(e->weights_bytes); mix(e->convs.size()); mix((uint64_t)e->mode);
  mix((uint64_t)e->prec); mix((uint64_t)e->split_fp8);
  for (auto& c : e->convs) {
    mix(c.w_off); mix(c.b_off); mix(c.w_bytes); mix((uint64_t)c.cfg); mix((uint64_t)c.nchunk); mix((uint64_t)c.h8);
    // what decides the CONTENTS of the packed arena at equal sizes: which operand the second pass carries (":w" = W_lo, ":a" = W_hi
    // again), fp16 instead of fp8 corrections (":x"), the chunking, the kernel the weights are packed for
    mix((uint64_t)c.split_w); mix((uint64_t)c.split_a); mix((uint64_t)c.no_h8); mix((uint64_t)c.ncp); mix((uint64_t)c.rowb); mix((uint64_t)c.CoutP);
    mix((uint64_t)c.Cin_p); mix((uint64_t)c.fused); mix((uint64_t)c.fused_chunks); mix((uint64_t)c.direct_first); mix((uint64_t)c.impl); mix((uint64_t)c.k); mix((uint64_t)c.pool);
    for (char ch : c.name) mix((uint64_t)(unsigned char)ch);
  }
  for (char ch : e->split_rules) mix((uint64_t)(unsigned char)ch);
  return h;
}
size_t ref_floats(const rtp_engine* e) {
  size_t n = 0;
  for (size_t i = 0; i < e->convs.size(); ++i) n += e->w_ref[i].size() + e->b_ref[i].size();
  return n;
}
}  // namespace
long rtp_weight_blob_bytes(const rtp_engine* e) {
  if (!e) return RTP_EINVAL;
  return (long)(4 * sizeof(uint64_t) + e->convs.size() * sizeof(int) + e->weights_bytes + ref_floats(e) * sizeof(float));
}
int rtp_weight_blob_export(rtp_engine* e, void* host, size_t capacity) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if ((rc = need_weights(e))) return rc;
  if (!host || (long)capacity < rtp_weight_blob_bytes(e)) return fail(e, RTP_EINVAL, "weight blob needs %ld bytes", rtp_weight_blob_bytes(e));
  unsigned char* p = (unsigned char*)host;
  const uint64_t head[4] = {0x5254505742303031ull /* "RTPWB001" */, plan_hash(e), (uint64_t)e->convs.size(), (uint64_t)e->weights_bytes};
  memcpy(p, head, sizeof head); p += sizeof head;
  for (auto& c : e->convs) { memcpy(p, &c.wq_exp, sizeof(int)); p += sizeof(int); }
  HIPCHK(e, hipMemcpy(p, e->dweights, e->weights_bytes, hipMemcpyDeviceToHost)); p += e->weights_bytes;
  for (size_t i = 0; i < e->convs.size(); ++i) {
    memcpy(p, e->w_ref[i].data(), e->w_ref[i].size() * sizeof(float)); p += e->w_ref[i].size() * sizeof(float);
    memcpy(p, e->b_ref[i].data(), e->b_ref[i].size() * sizeof(float)); p += e->b_ref[i].size() * sizeof(float);
  }
  return RTP_OK;
}
int rtp_weight_blob_import(rtp_engine* e, const void* host, size_t bytes) {
  SYNC_GUARD;
  int rc;
  if ((rc = need_idle(e))) return rc;
  if (!host || (long)bytes < rtp_weight_blob_bytes(e)) return fail(e, RTP_EINVAL, "weight blob too short (%zu bytes, this plan needs %ld)", bytes, rtp_weight_blob_bytes(e));
  const unsigned char* p = (const unsigned char*)host;
  uint64_t head[4];
  memcpy(head, p, sizeof head); p += sizeof head;
  if (head[0] != 0x5254505742303031ull) return fail(e, RTP_EINVAL, "not a weight blob");
  if (head[1] != plan_hash(e) || head[2] != e->convs.size() || head[3] != e->weights_bytes)
    return fail(e, RTP_EINVAL, "weight blob was exported by an engine with another plan (model / resolution / precision / split set must match)");
  bool moved = false;
  for (auto& c : e->convs) { int ex; memcpy(&ex, p, sizeof(int)); p += sizeof(int); moved = moved || ex != c.wq_exp; c.wq_exp = ex; }
  HIPCHK(e, hipDeviceSynchronize());
  HIPCHK(e, hipMemcpy(e->dweights, p, e->weights_bytes, hipMemcpyHostToDevice)); p += e->weights_bytes;
  for (size_t i = 0; i < e->convs.size(); ++i) {
    memcpy(e->w_ref[i].data(), p, e->w_ref[i].size() * sizeof(float)); p += e->w_ref[i].size() * sizeof(float);
    memcpy(e->b_ref[i].data(), p, e->b_ref[i].size() * sizeof(float)); p += e->b_ref[i].size() * sizeof(float);
  }
  if (moved && (rc = invalidate_graphs(e))) return rc;  // ConvParams::wq_exp is baked into the captured graphs
  return weights_delivered(e);   // (defer_weights: the first weights of this engine — capture its graphs now)
}
// dst takes src's packed arena device-to-device (hipMemcpyPeer: xGMI between two GPUs of a node); both engines idle, same plan.
int rtp_copy_weights_from(rtp_engine* dst, rtp_engine* src) {
  SYNC_GUARD;
  if (!dst || !src) return RTP_EINVAL;
  int rc;
  if ((rc = need_idle(dst))) return rc;
  if (!src->fifo.empty()) return fail(dst, RTP_EAGAIN, "rtp_copy_weights_from: the source engine has frames in flight");
  if (src->weights_pending) return fail(dst, RTP_EINVAL, "rtp_copy_weights_from: the source engine has no weights itself (defer_weights)");
  if (plan_hash(dst) != plan_hash(src)) return fail(dst, RTP_EINVAL, "rtp_copy_weights_from: the engines have different plans");
  HIPCHK(dst, hipDeviceSynchronize());
  HIPCHK(dst, hipMemcpyPeer(dst->dweights, dst->cfg.device_id, src->dweights, src->cfg.device_id, dst->weights_bytes));
  bool moved = false;
  for (size_t i = 0; i < dst->convs.size(); ++i) {
    moved = moved || dst->convs[i].wq_exp != src->convs[i].wq_exp;
    dst->convs[i].wq_exp = src->convs[i].wq_exp;
    dst->w_ref[i] = src->w_ref[i];
    dst->b_ref[i] = src->b_ref[i];
  }
  if (moved && (rc = invalidate_graphs(dst))) return rc;
  return weights_delivered(dst);
}

// Nothing may unwind through the C boundary: allocation failures while building a plan come back as codes.
// A caller compiled against another version of the header (or one that never called rtp_config_default) hands over a struct of another
// layout: refuse it instead of reading fields that are not there.
static int config_layout_ok(const rtp_config* cfg) {
  if (cfg && cfg->struct_size != (unsigned)sizeof(rtp_config))
    return fail(nullptr, RTP_EINVAL, "rtp_config.struct_size is %u, this library's rtp_config has %u bytes: fill the struct with rtp_config_default() of the "
                "header this library was built from (include/rtpose_mi355x.h) and recompile the caller", cfg->struct_size, (unsigned)sizeof(rtp_config));
  return RTP_OK;
}

int rtp_engine_create(const rtp_config* cfg, rtp_engine** out) {
  if (int rc = config_layout_ok(cfg)) return rc;
  try {
    return engine_create_impl(cfg, out);
  } catch (const std::bad_alloc&) {
    return fail(nullptr, RTP_ENOMEM, "out of host memory while creating the engine");
  } catch (const std::exception& ex) {
    return fail(nullptr, RTP_EINVAL, "engine creation failed: %s", ex.what());
  }
}
long rtp_plan_summary(const rtp_config* cfg, char* buf, size_t buflen) {
  if (int rc = config_layout_ok(cfg)) return rc;
  try {
    return plan_summary_impl(cfg, buf, buflen);
  } catch (const std::exception& ex) {
    return fail(nullptr, RTP_ENOMEM, "plan summary failed: %s", ex.what());
  }
}

}  // extern "C"
