// rtpose_main.cpp — rtpose.bin: the reference's CLI surface (examples/rtpose/rtpose.cpp:50-72,
// 1674-1780) over the MI355X engine.  Thread structure follows rtcpm() (rtpose.cpp:1459-1547):
//   1 producer (decode/generate -> display-fit warp -> scale pyramid -> net input)   :302/393
//   NUM_GPU workers, one engine each, all pulling from ONE shared queue              :1099-1203
//   1 re-orderer (min-heap on frame.index, window BUFFER_SIZE = 4, skips dropped)    :1214-1273
//   1 writer (JSON, latency log every 30 frames)                                     :1315-1454
// There is no collective: frames are independent.  Differences from the reference,
// all forced by the environment: no display/camera (no GUI stack), image decoding limited to
// JPEG/PNG/PPM/BMP and Y4M / raw MJPEG (codecs.cpp), `--video synthetic:WxH:frames[:seed]` generates frames procedurally, and the
// process exits at end of input also without --write_frames (the reference loops the video forever).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <dirent.h>
#include <fstream>
#include <future>
#include <map>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <queue>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

#include "../../include/rtpose_mi355x.h"

extern "C" {
double rtp_display_fit_scale(int ow, int oh, int disp_w, int disp_h);
int rtp_preprocess_frame(const unsigned char* bgr, int w, int h, int disp_w, int disp_h, int net_w, int net_h, int num_scales,
                         double start_scale, double scale_gap, float* net_input, unsigned char* display_bgr, float* frame_scale);
int rtp_load_image(const char* path, unsigned char* out_bgr, size_t capacity, int* w, int* h);
int rtp_synth_frame(unsigned char* out_bgr, int w, int h, int index, uint64_t seed);
}

namespace {

double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- flags (names and defaults: rtpose.cpp:50-72) --------------------------------------------
struct Flags {
  bool host_preprocess = false;
  bool fullscreen = false, no_frame_drops = false, no_display = false, no_text = false, logtostderr = false;
  int part_to_show = 0, camera = 0, start_frame = 0, start_device = 0, num_gpu = 1, num_scales = 1;
  std::string write_frames, write_json, video, image_dir;
  std::string caffemodel = "model/coco/pose_iter_440000.caffemodel", caffeproto = "model/coco/pose_deploy_linevec.prototxt";
  std::string resolution = "1280x720", net_resolution = "656x368", camera_resolution = "1280x720";
  double start_scale = 1, scale_gap = 0.3;
  // extensions (not in the reference)
  std::string precision = "mixed", model = "";  // --model coco|mpi: use the built-in graph + synthetic weights
  int frames_in_flight = 2, batch_frames = 1;
  unsigned long long synthetic_seed = 1;
  std::string devices;           // "0,0,1": device of worker i (default start_device + i); lets two workers share one GPU
  int test_worker_delay_ms = 0;  // test hook: every worker sleeps this long after a submit (provokes the >0.1 s frame drops)
  // --dry_engine RATE: no GPU.  Every worker is a stand-in that copies the frame like rtp_submit_frame's staging copy and
  // completes frames at RATE frames/s (pipelined, frames_in_flight deep) with --dry_people fake people each: the host side of
  // --num_gpu N (producer -> queue -> workers -> re-orderer -> writer) can be driven at N x the measured per-GPU rate on any box.
  double dry_engine = 0;
  int dry_people = 5;
  int producer_threads = 0;      // frames are generated / decoded ahead by this many threads (0 = hardware threads / 4, clamped to [2, 16])
  bool share_weights = false;    // workers 1.. take worker 0's PACKED weight arena device to device (rtp_copy_weights_from: hipMemcpyPeer over xGMI)
                                 // instead of keeping the copy they packed themselves (the reference: one .caffemodel read per GPU thread, rtpose.cpp:183-184)
  bool pin_workers = true;       // every worker thread runs on the CPUs local to its GPU (rtp_device_local_cpus), so the pinned staging buffers it
                                 // allocates and fills are on that NUMA node; --nopin_workers leaves the placement to the scheduler
  int calibrate = 0;             // K > 0: rtp_calibrate_precision on K frames right after the weights are loaded (mixed precision only)
  int json_writers = -1;         // JSON files are written by this many threads (0 = by the display/writer thread itself, like the reference;
                                 // default: 1 thread per worker — creating a file costs the one display thread ~0.5 ms, 1600 frames/s at most)
};

int parse_flags(int argc, char** argv, Flags& F) {
  std::map<std::string, std::string*> sflags = {{"write_frames", &F.write_frames}, {"write_json", &F.write_json}, {"video", &F.video},
      {"image_dir", &F.image_dir}, {"caffemodel", &F.caffemodel}, {"caffeproto", &F.caffeproto}, {"resolution", &F.resolution},
      {"net_resolution", &F.net_resolution}, {"camera_resolution", &F.camera_resolution}, {"precision", &F.precision}, {"model", &F.model}, {"devices", &F.devices}};
  std::map<std::string, int*> iflags = {{"part_to_show", &F.part_to_show}, {"camera", &F.camera}, {"start_frame", &F.start_frame},
      {"start_device", &F.start_device}, {"num_gpu", &F.num_gpu}, {"num_scales", &F.num_scales}, {"frames_in_flight", &F.frames_in_flight}, {"batch_frames", &F.batch_frames},
      {"test_worker_delay_ms", &F.test_worker_delay_ms}, {"dry_people", &F.dry_people}, {"json_writers", &F.json_writers}, {"producer_threads", &F.producer_threads}, {"calibrate", &F.calibrate}};
  std::map<std::string, double*> dflags = {{"start_scale", &F.start_scale}, {"scale_gap", &F.scale_gap}, {"dry_engine", &F.dry_engine}};
  std::map<std::string, bool*> bflags = {{"fullscreen", &F.fullscreen}, {"no_frame_drops", &F.no_frame_drops}, {"host_preprocess", &F.host_preprocess}, {"no_display", &F.no_display},
      {"no_text", &F.no_text}, {"logtostderr", &F.logtostderr}, {"share_weights", &F.share_weights}, {"pin_workers", &F.pin_workers}};
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--help" || a == "-help" || a == "-h") return 2;
    if (a.size() < 2 || a[0] != '-') { fprintf(stderr, "ERROR: unexpected argument '%s'\n", a.c_str()); return 1; }
    a = a.substr(a[1] == '-' ? 2 : 1);
    std::string val;
    bool has_val = false;
    const size_t eq = a.find('=');
    if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); has_val = true; }
    if (bflags.count(a)) {
      if (has_val) *bflags[a] = (val == "true" || val == "1");
      else *bflags[a] = true;
      continue;
    }
    if (a.rfind("no", 0) == 0 && bflags.count(a.substr(2)) && !has_val) { *bflags[a.substr(2)] = false; continue; }  // gflags --noflag
    if (!sflags.count(a) && !iflags.count(a) && !dflags.count(a) && a != "synthetic_seed") {
      fprintf(stderr, "ERROR: unknown command line flag '%s'\n", a.c_str());
      return 1;
    }
    if (!has_val) {
      if (i + 1 >= argc) { fprintf(stderr, "ERROR: flag '%s' is missing its argument\n", a.c_str()); return 1; }
      val = argv[++i];
    }
    if (sflags.count(a)) *sflags[a] = val;
    else if (iflags.count(a)) *iflags[a] = atoi(val.c_str());
    else if (dflags.count(a)) *dflags[a] = atof(val.c_str());
    else F.synthetic_seed = strtoull(val.c_str(), nullptr, 10);
  }
  return 0;
}

void usage() {
  printf("rtpose.bin (MI355X engine) — flags as examples/rtpose/rtpose.cpp:\n"
         "  --video FILE.y4m|FILE.mjpeg|synthetic:WxH:frames[:seed]   --image_dir DIR (jpg/png/bmp/ppm)   --camera N (unsupported)\n"
         "  --caffeproto FILE --caffemodel FILE   | --model coco|mpi (built-in graph, synthetic weights)\n"
         "  --resolution WxH (1280x720) --net_resolution WxH (656x368) --num_scales N (1) --scale_gap G (0.3) --start_scale S (1)\n"
         "  --num_gpu N (1) --start_device D (0) --no_frame_drops --write_json DIR --write_frames DIR --start_frame N\n"
         "  --no_display --no_text --fullscreen --part_to_show N --logtostderr   [--precision mixed|fp16|f16x3|fp32 --frames_in_flight K --batch_frames B --host_preprocess\n"
         "   --devices d0,d1,.. (device of each worker; the same device may appear twice)\n"
         "   --dry_engine RATE (no GPU: every worker is a stand-in finishing RATE frames/s; measures the host side of --num_gpu N) --dry_people P\n"
         "   --json_writers K (JSON files written by K threads instead of the one display thread) --producer_threads K (decode / generate ahead)\n"
         "   --calibrate K (check / widen the mixed-precision split set on the loaded weights with K sample frames)\n"
         "   --share_weights (workers 1.. copy worker 0's packed weights GPU to GPU) --nopin_workers (no CPU affinity next to each worker's GPU)]\n"
         "  --write_frames draws the pose overlay only: the FPS / people-count text of the reference (cv::putText, rtpose.cpp:1319-1333) is not\n"
         "  drawn, i.e. --no_text is implied.\n");
}

// ---- queues (caffe::BlockingQueue, util/blocking_queue.cpp:26-61) -----------------------------
template <typename T> class BlockingQueue {
 public:
  void push(T v) { { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(v)); } cv_.notify_one(); }
  bool try_pop(T* v) { std::lock_guard<std::mutex> l(m_); if (q_.empty()) return false; *v = std::move(q_.front()); q_.pop_front(); return true; }
  bool pop_wait(T* v, std::atomic<bool>& quit) {
    std::unique_lock<std::mutex> l(m_);
    while (q_.empty()) { if (quit.load()) return false; cv_.wait_for(l, std::chrono::milliseconds(2)); }
    *v = std::move(q_.front()); q_.pop_front();
    return true;
  }
  size_t size() { std::lock_guard<std::mutex> l(m_); return q_.size(); }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<T> q_;
};

// include/caffe/cpm/frame.h:6-34
struct Frame {
  std::vector<float> data;         // net input (only with --host_preprocess)
  std::vector<unsigned char> image;  // decoded u8 BGR frame (default: pre-processing runs on the GPU)
  std::vector<unsigned char> rendered;  // --write_frames: display-resolution frame with the pose overlay
  int img_w = 0, img_h = 0;
  double commit_time = 0, preprocessed_time = 0, gpu_fetched_time = 0, gpu_computed_time = 0, buffer_start_time = 0, buffer_end_time = 0;
  int index = 0, numPeople = 0, video_frame_number = 0;
  float scale = 1.f;
  std::string stem;
  std::vector<float> joints;
};

struct EncodeJob { std::string path; std::vector<unsigned char> bgr; };

struct Global {
  BlockingQueue<Frame> input_queue, output_queue, output_queue_ordered;
  BlockingQueue<EncodeJob> encode_queue;  // --write_frames: JPEG encoding is spread over a few threads (files are independent)
  std::atomic<bool> encode_done{false};
  std::priority_queue<int, std::vector<int>, std::greater<int>> dropped_index;
  std::mutex mutex;
  std::atomic<bool> quit_threads{false};
  std::atomic<int> produced{0}, finished{0}, dropped{0};
  std::vector<int> per_worker;  // frames each worker submitted (dynamic pull from the one shared queue)
  std::atomic<rtp_engine*> engine0{nullptr};  // --share_weights: worker 0's engine once it is up (idle until every worker is ready)
  std::atomic<int> weights_shared{0};
  std::atomic<int> share_done{0};      // receiving workers that are through with worker 0's engine (copied, or failed): worker 0 keeps it alive until all are
  std::atomic<int> workers_ready{0};  // workers start pulling once EVERY engine is up (engine creation takes seconds, short inputs milliseconds)
  std::atomic<bool> producer_done{false};
  double first_commit = 0;                     // steady-state window: first frame committed .. last file written
  std::atomic<double> last_written{0};         // (by the display thread, or by whichever JSON writer finished last)
  BlockingQueue<Frame> json_queue;             // --json_writers K
  std::atomic<bool> json_done{false};
  std::atomic<int> num_parts{18};              // every worker stores the same value (its engine's) while the JSON writers read it
  std::vector<std::string> image_list;
};

Flags F;
Global G;
int DISP_W, DISP_H, NET_W, NET_H;
const int BUFFER_SIZE = 4;  // rtpose.cpp:90

bool mkdir_p(const std::string& d) { struct stat st; if (stat(d.c_str(), &st) == 0) return S_ISDIR(st.st_mode); return mkdir(d.c_str(), 0755) == 0; }

// ---- producer -----------------------------------------------------------------------------------
void producer() {
  int global_counter = 1;
  int sw = 0, sh = 0, nframes = 0;
  unsigned long long seed = 2;
  const bool synthetic = F.video.rfind("synthetic:", 0) == 0;
  rtp_video* vid = nullptr;
  if (synthetic) {
    if (sscanf(F.video.c_str(), "synthetic:%dx%d:%d:%llu", &sw, &sh, &nframes, &seed) < 3) { fprintf(stderr, "bad --video %s\n", F.video.c_str()); G.quit_threads = true; return; }
  } else if (!F.video.empty()) {  // cv::VideoCapture(FLAGS_video), rtpose.cpp:402-411
    if (rtp_video_open(F.video.c_str(), &vid, &sw, &sh, &nframes) != RTP_OK) { fprintf(stderr, "Couldn't open video file %s: %s\n", F.video.c_str(), rtp_codec_last_error()); G.quit_threads = true; G.producer_done = true; return; }
    if (nframes < 0) nframes = 1 << 30;
    std::vector<unsigned char> skip((size_t)sw * sh * 3);
    for (int i = 0; i < F.start_frame; ++i) if (rtp_video_read(vid, skip.data(), skip.size()) != RTP_OK) break;  // CAP_PROP_POS_FRAMES, :411
  } else nframes = (int)G.image_list.size();
  std::vector<unsigned char> img;
  // --image_dir files and synthetic frames are produced a few ahead by a small pool (a 720p JPEG takes ~13 ms on one core, a
  // synthetic frame ~1 ms; 8 GPUs at 1 scale want ~8000 frames/s); the producer still hands the frames over in index order,
  // like the reference's single loop.  Video files are read sequentially (one decoder state).
  struct Decoded { std::vector<unsigned char> bgr; int w = 0, h = 0; std::string err; };
  const int hw = (int)std::thread::hardware_concurrency();
  const int pool_n = (synthetic || vid == nullptr) ? (F.producer_threads > 0 ? std::min(F.producer_threads, 64) : std::max(2, std::min(16, hw / 4))) : 0;
  const int window = 4 * std::max(pool_n, 1);
  std::mutex pm;
  std::condition_variable pcv;
  std::map<int, Decoded> ready;            // frame index -> frame, filled by the pool
  int next_job = F.start_frame, consumed = F.start_frame;
  bool pool_quit = false;
  auto make_frame = [&](int fi) {
    Decoded d;
    if (synthetic) {
      d.w = sw; d.h = sh;
      d.bgr.resize((size_t)sw * sh * 3);
      rtp_synth_frame(d.bgr.data(), sw, sh, fi, seed);
    } else {
      const std::string& path = G.image_list[fi];
      if (rtp_load_image(path.c_str(), nullptr, 0, &d.w, &d.h) != RTP_OK) { d.err = rtp_codec_last_error(); d.w = 0; return d; }
      d.bgr.resize((size_t)d.w * d.h * 3);
      if (rtp_load_image(path.c_str(), d.bgr.data(), d.bgr.size(), &d.w, &d.h) != RTP_OK) { d.err = rtp_codec_last_error(); d.w = 0; }
    }
    return d;
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < pool_n; ++t)
    pool.emplace_back([&]() {
      while (true) {
        int fi;
        {
          std::unique_lock<std::mutex> l(pm);
          pcv.wait(l, [&] { return pool_quit || (next_job < nframes && next_job < consumed + window); });
          if (pool_quit) return;
          fi = next_job++;
        }
        Decoded d = make_frame(fi);
        { std::lock_guard<std::mutex> l(pm); ready[fi] = std::move(d); }
        pcv.notify_all();
      }
    });
  while (G.workers_ready.load() < F.num_gpu && !G.quit_threads) std::this_thread::sleep_for(std::chrono::milliseconds(1));  // no frame ages while the engines start
  for (int fi = F.start_frame; fi < nframes && !G.quit_threads; ++fi) {
    // back-pressure (rtpose.cpp:311, 424-429); the queue bound follows the number of consumers
    while ((int)G.input_queue.size() > 10 + 4 * F.num_gpu && !G.quit_threads) std::this_thread::sleep_for(std::chrono::microseconds(200));
    Frame fr;
    int w = sw, h = sh;
    if (vid) {
      img.resize((size_t)w * h * 3);
      const int rc = rtp_video_read(vid, img.data(), img.size());
      if (rc == RTP_EAGAIN) break;  // end of the stream
      if (rc != RTP_OK) { fprintf(stderr, "video frame %d: %s\n", fi, rtp_codec_last_error()); break; }
      char nm[64]; snprintf(nm, sizeof nm, "frame%06d", fi); fr.stem = nm;
    } else {
      Decoded d;
      {
        std::unique_lock<std::mutex> l(pm);
        pcv.wait(l, [&] { return ready.count(fi) != 0; });
        d = std::move(ready[fi]);
        ready.erase(fi);
        consumed = fi + 1;
      }
      pcv.notify_all();
      if (d.w == 0) { fprintf(stderr, "cannot decode %s: %s\n", G.image_list[fi].c_str(), d.err.c_str()); continue; }
      img.swap(d.bgr);
      w = d.w; h = d.h;
      if (synthetic) { char nm[64]; snprintf(nm, sizeof nm, "frame%06d", fi); fr.stem = nm; }
      else {
        const std::string& path = G.image_list[fi];
        size_t sl = path.find_last_of('/'), dot = path.find_last_of('.');
        fr.stem = path.substr(sl == std::string::npos ? 0 : sl + 1, dot - (sl == std::string::npos ? 0 : sl + 1));
      }
    }
    fr.commit_time = wall();
    if (F.host_preprocess) {
      fr.data.resize((size_t)F.num_scales * 3 * NET_H * NET_W);
      if (rtp_preprocess_frame(img.data(), w, h, DISP_W, DISP_H, NET_W, NET_H, F.num_scales, F.start_scale, F.scale_gap, fr.data.data(), nullptr, &fr.scale) != RTP_OK) {
        fprintf(stderr, "preprocess failed for frame %d\n", fi);
        continue;
      }
    } else {
      fr.image = std::move(img);   // handed over, not copied (2.76 MB per 720p frame)
      img.clear();
      fr.img_w = w;
      fr.img_h = h;
    }
    fr.index = global_counter++;
    fr.video_frame_number = fi;
    fr.preprocessed_time = wall();
    if (G.first_commit == 0) G.first_commit = fr.commit_time;
    G.produced++;
    G.input_queue.push(std::move(fr));
  }
  { std::lock_guard<std::mutex> l(pm); pool_quit = true; }
  pcv.notify_all();
  for (auto& t : pool) t.join();
  if (vid) rtp_video_close(vid);
  G.producer_done = true;
}

// CPU affinity of a worker thread: the CPUs local to its GPU's PCI function ("0-31,128-159"); silently skipped where sysfs does not say
void pin_thread_near_device(int widx, int device) {
  char list[512];
  if (rtp_device_local_cpus(device, list, sizeof list) <= 0) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  int n = 0;
  for (const char* p = list; *p;) {
    char* end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); ++n; }
    if (*p == ',') ++p;
  }
  if (n > 0 && pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0) fprintf(stderr, "worker %d (GPU %d) runs on CPUs %s\n", widx, device, list);
}

// ---- per-GPU worker (processFrame, rtpose.cpp:1079-1203) -----------------------------------------
void worker(int widx, int device, int* status) {
  rtp_config cfg;
  rtp_config_default(&cfg);
  cfg.device_id = device;
  if (!F.model.empty()) cfg.model = (F.model == "mpi") ? RTP_MODEL_MPI_15 : RTP_MODEL_COCO_18;
  else { cfg.proto_path = F.caffeproto.c_str(); cfg.weights_path = F.caffemodel.c_str(); }
  cfg.synthetic_seed = F.synthetic_seed;
  cfg.net_w = NET_W; cfg.net_h = NET_H; cfg.num_scales = F.num_scales;
  cfg.start_scale = (float)F.start_scale; cfg.scale_gap = (float)F.scale_gap;
  cfg.disp_w = DISP_W; cfg.disp_h = DISP_H;
  cfg.precision = F.precision == "fp32" ? RTP_PREC_FP32 : F.precision == "fp16" ? RTP_PREC_FP16 : F.precision == "f16x3" ? RTP_PREC_F16X3 : RTP_PREC_MIXED;
  cfg.frames_in_flight = F.frames_in_flight;
  cfg.batch_frames = F.batch_frames;
  cfg.render = F.write_frames.empty() ? 0 : 1 + F.part_to_show;  // render() of rtpose.cpp:270-299: pose overlay or a --part_to_show view
  cfg.calibrate_frames = F.calibrate;
  rtp_engine* e = nullptr;
  const bool dry = F.dry_engine > 0;
  if (!dry && F.pin_workers && F.num_gpu > 1) pin_thread_near_device(widx, device);
  // --share_weights: ONE read + pack of the model for N replicas.  Worker 0 loads (and, with --calibrate or a model file, calibrates);
  // the others wait for it, are created WITHOUT weights (rtp_config.defer_weights) on worker 0's final split set and take its packed
  // arena device to device before anybody submits a frame (worker 0's engine is idle until every worker is ready).
  std::vector<char> src_rules(1 << 16);   // (a calibrated rule list names layer groups: a few hundred bytes; rtp_get_split_layers refuses a buffer that is too small)
  rtp_engine* src = nullptr;
  struct ShareDone {   // counted on EVERY way out of this function once a receiving worker exists (worker 0 waits for the count before it destroys its engine)
    bool armed = false;
    ~ShareDone() { if (armed) G.share_done++; }
  } share_done_guard;
  if (!dry && F.share_weights && widx != 0) {
    share_done_guard.armed = true;
    while (!(src = G.engine0.load()) && !G.quit_threads) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (!src) { *status = 1; return; }   // worker 0 failed and everybody is quitting: nothing was copied, nothing is counted
    int prec = cfg.precision;
    if (rtp_get_split_layers(src, src_rules.data(), src_rules.size(), &prec) != RTP_OK) { fprintf(stderr, "GPU %d: --share_weights: %s\n", device, rtp_last_error(src)); *status = 1; G.quit_threads = true; return; }
    cfg.precision = prec;
    cfg.split_layers = src_rules.data();
    cfg.calibrate_frames = -1;
    cfg.defer_weights = 1;
  }
  if (!dry && rtp_engine_create(&cfg, &e) != RTP_OK) {
    fprintf(stderr, "GPU %d: %s\n", device, rtp_last_error(nullptr));
    *status = 1;
    G.quit_threads = true;
    return;
  }
  if (!dry && F.share_weights) {
    if (widx == 0) G.engine0 = e;
    else {
      if (rtp_copy_weights_from(e, src) != RTP_OK) {
        fprintf(stderr, "GPU %d: --share_weights: %s\n", device, rtp_last_error(e));
        *status = 1;
        G.quit_threads = true;
        rtp_engine_destroy(e);
        return;
      }
      G.weights_shared++;
    }
    if (widx != 0) { share_done_guard.armed = false; G.share_done++; }   // through with worker 0's engine
  }
  int num_parts = F.model == "mpi" ? 15 : 18;
  if (!dry) rtp_engine_info(e, &num_parts, nullptr, nullptr, nullptr, nullptr);
  G.num_parts.store(num_parts);
  // --dry_engine: what the engine costs the HOST per frame (the staging copy of rtp_submit_frame, the joints copy of
  // rtp_collect) and when a frame completes (RATE frames/s, at least two frame times after its submit)
  std::vector<unsigned char> dry_staging;
  std::deque<double> dry_done;   // completion times of the frames in flight
  double dry_last = 0;
  auto dry_submit = [&](const Frame& fr) {
    const size_t n = F.host_preprocess ? fr.data.size() * sizeof(float) : fr.image.size();
    if (dry_staging.size() < n) dry_staging.resize(n);
    memcpy(dry_staging.data(), F.host_preprocess ? (const void*)fr.data.data() : (const void*)fr.image.data(), n);
    const double now = wall();
    dry_last = std::max(now + 2.0 / F.dry_engine, dry_last + 1.0 / F.dry_engine);
    dry_done.push_back(dry_last);
    return RTP_OK;
  };
  auto dry_collect = [&](Frame& fr, float* joints, int* n) {
    const double t = dry_done.front();
    dry_done.pop_front();
    for (double now = wall(); now < t; now = wall()) std::this_thread::sleep_for(std::chrono::duration<double>(std::min(t - now, 0.0005)));
    const int P = std::max(0, std::min(F.dry_people, (int)RTP_MAX_PEOPLE));
    unsigned long long r = 0x9E3779B97F4A7C15ull * (unsigned long long)(fr.index + 1);
    for (int i = 0; i < P * num_parts; ++i) {
      r ^= r << 13; r ^= r >> 7; r ^= r << 17;
      joints[3 * i] = (float)(r % 1280000) / 1000.f;
      joints[3 * i + 1] = (float)((r >> 20) % 720000) / 1000.f;
      joints[3 * i + 2] = (float)((r >> 40) % 1000) / 1000.f;
    }
    *n = P;
    if (!fr.rendered.empty()) memset(fr.rendered.data(), 40 + (fr.index * 7) % 160, fr.rendered.size());  // --write_frames: what the encoders get instead of rtp_collect_rendered's frame
    return RTP_OK;
  };
  fprintf(stderr, dry ? "dry worker %d is ready (%.0f frames/s)\n" : "GPU %d is ready\n", device, F.dry_engine);
  G.workers_ready++;
  while (G.workers_ready.load() < F.num_gpu && !G.quit_threads) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  std::deque<Frame> inflight;
  std::vector<float> joints((size_t)RTP_MAX_PEOPLE * num_parts * 3);
  auto collect_one = [&]() {
    uint64_t tag;
    int n = 0;
    Frame fr = std::move(inflight.front());
    inflight.pop_front();
    if (!F.write_frames.empty()) fr.rendered.resize((size_t)DISP_W * DISP_H * 3);
    const int rc = dry ? dry_collect(fr, joints.data(), &n)
                       : F.write_frames.empty() ? rtp_collect(e, &tag, joints.data(), &n)
                                                : rtp_collect_rendered(e, &tag, joints.data(), &n, fr.rendered.data());
    if (rc != RTP_OK) { fprintf(stderr, "GPU %d frame %d: %s\n", device, fr.index, rtp_last_error(e)); n = 0; }
    fr.numPeople = n;
    fr.joints.assign(joints.begin(), joints.begin() + (size_t)n * num_parts * 3);
    fr.gpu_computed_time = wall();
    fr.data.clear();
    fr.data.shrink_to_fit();
    G.output_queue.push(std::move(fr));
  };
  while (!G.quit_threads) {
    Frame fr;
    bool got = (int)inflight.size() < F.frames_in_flight && G.input_queue.try_pop(&fr);
    if (got) {
      fr.gpu_fetched_time = wall();
      if (fr.gpu_fetched_time - fr.commit_time > 0.1 && !F.no_frame_drops) {  // rtpose.cpp:1112-1124
        std::lock_guard<std::mutex> l(G.mutex);
        G.dropped_index.push(fr.index);
        G.dropped++;
        continue;
      }
      if (dry && !F.host_preprocess) fr.scale = (float)rtp_display_fit_scale(fr.img_w, fr.img_h, DISP_W, DISP_H);
      const int src = dry ? dry_submit(fr)
                          : F.host_preprocess ? rtp_submit(e, fr.data.data(), (uint64_t)fr.index)
                                              : rtp_submit_frame(e, fr.image.data(), fr.img_w, fr.img_h, (uint64_t)fr.index, &fr.scale);
      fr.image.clear();
      fr.image.shrink_to_fit();
      if (src != RTP_OK) {  // nobody else may be left to drain the queue: stop the producer too
        fprintf(stderr, "GPU %d: %s\n", device, rtp_last_error(e));
        *status = 1;
        G.quit_threads = true;
        break;
      }
      G.per_worker[widx]++;
      inflight.push_back(std::move(fr));
      if (F.test_worker_delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(F.test_worker_delay_ms));
      if ((int)inflight.size() < F.frames_in_flight) continue;
    }
    if (!inflight.empty()) collect_one();
    else if (G.producer_done && G.input_queue.size() == 0) break;
    else std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  while (!inflight.empty()) collect_one();
  if (e && F.share_weights && widx == 0 && !dry)   // a receiver may still be inside rtp_get_split_layers / rtp_copy_weights_from on this engine (error paths set quit_threads early)
    while (G.share_done.load() < F.num_gpu - 1) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  if (e) rtp_engine_destroy(e);
}

// ---- re-orderer (buffer_and_order, rtpose.cpp:1214-1273) ---------------------------------------------
struct FrameCompare { bool operator()(const Frame& a, const Frame& b) const { return a.index > b.index; } };
void reorderer(std::atomic<bool>* workers_done) {
  std::priority_queue<Frame, std::vector<Frame>, FrameCompare> buffer;
  int frame_waited = 1;
  auto skip_dropped = [&]() {
    std::lock_guard<std::mutex> l(G.mutex);
    while (!G.dropped_index.empty() && G.dropped_index.top() == frame_waited) { frame_waited++; G.dropped_index.pop(); }
  };
  auto emit = [&](Frame f) { f.buffer_end_time = wall(); G.output_queue_ordered.push(std::move(f)); };
  while (true) {
    Frame fr;
    if (!G.output_queue.try_pop(&fr)) {
      if (workers_done->load() && G.output_queue.size() == 0) break;
      std::this_thread::sleep_for(std::chrono::microseconds(200));
      continue;
    }
    fr.buffer_start_time = wall();
    skip_dropped();
    if (fr.index == frame_waited) {
      emit(std::move(fr));
      frame_waited++;
      skip_dropped();
      while (!buffer.empty() && buffer.top().index == frame_waited) { emit(buffer.top()); buffer.pop(); frame_waited++; skip_dropped(); }
    } else buffer.push(std::move(fr));
    if ((int)buffer.size() > BUFFER_SIZE) {  // window overflow: force-emit the smallest
      Frame extra = buffer.top();
      buffer.pop();
      frame_waited = extra.index + 1;
      emit(std::move(extra));
      while (!buffer.empty() && buffer.top().index == frame_waited) { emit(buffer.top()); buffer.pop(); frame_waited++; }
    }
  }
  while (!buffer.empty()) { emit(buffer.top()); buffer.pop(); }
}

// ---- JPEG encoders for --write_frames ------------------------------------------------------------------
void encoder() {
  std::vector<unsigned char> jpg((size_t)DISP_W * DISP_H * 4 + 4096);
  while (true) {
    EncodeJob job;
    if (!G.encode_queue.try_pop(&job)) {
      if (G.encode_done.load() && G.encode_queue.size() == 0) break;
      std::this_thread::sleep_for(std::chrono::microseconds(500));
      continue;
    }
    const long n = rtp_encode_jpeg(job.bgr.data(), DISP_W, DISP_H, 98, jpg.data(), jpg.size());
    if (n > 0) {
      std::ofstream fs(job.path, std::ios::binary);
      fs.write((const char*)jpg.data(), n);
    } else fprintf(stderr, "JPEG encode failed for %s\n", job.path.c_str());
  }
}

// the JSON block of displayFrame (rtpose.cpp:1383-1416)
void write_json_file(const Frame& fr, std::vector<char>& buf) {
  char fname[1024];
  if (F.image_dir.empty()) snprintf(fname, sizeof fname, "%s/frame%06d.json", F.write_json.c_str(), fr.video_frame_number);  // :1388
  else snprintf(fname, sizeof fname, "%s/%s.json", F.write_json.c_str(), fr.stem.c_str());                                   // :1390-1393
  const long n = rtp_format_json(buf.data(), buf.size(), fr.joints.data(), fr.numPeople, G.num_parts.load(), fr.scale);
  if (n >= 0) {
    FILE* f = fopen(fname, "wb");
    if (f) { fwrite(buf.data(), 1, (size_t)n, f); fclose(f); }
    else fprintf(stderr, "cannot create %s\n", fname);
  } else fprintf(stderr, "JSON buffer too small for frame %d\n", fr.index);
}
// --json_writers K: the files are independent (named by frame number / image stem), so K threads format and write them;
// ordering is still the re-orderer's (what the display thread shows / logs)
void json_writer() {
  std::vector<char> buf(1 << 20);
  while (true) {
    Frame fr;
    if (!G.json_queue.try_pop(&fr)) {
      if (G.json_done.load() && G.json_queue.size() == 0) break;
      std::this_thread::sleep_for(std::chrono::microseconds(200));
      continue;
    }
    write_json_file(fr, buf);
    const double now = wall();
    double prev = G.last_written.load();
    while (prev < now && !G.last_written.compare_exchange_weak(prev, now)) {}
  }
}

// ---- writer (displayFrame, rtpose.cpp:1315-1454) -------------------------------------------------------
void writer(std::atomic<bool>* reorder_done) {
  int counter = 1;
  double last_time = wall();
  std::vector<char> buf(1 << 20);
  while (true) {
    Frame fr;
    if (!G.output_queue_ordered.try_pop(&fr)) {
      if (reorder_done->load() && G.output_queue_ordered.size() == 0) break;
      std::this_thread::sleep_for(std::chrono::microseconds(200));
      continue;
    }
    double t_commit = fr.commit_time, t_pre = fr.preprocessed_time, t_fetch = fr.gpu_fetched_time, t_done = fr.gpu_computed_time, t_b0 = fr.buffer_start_time, t_b1 = fr.buffer_end_time;
    const int f_index = fr.index, f_np = fr.numPeople;
    if (!F.write_frames.empty() && !fr.rendered.empty()) {  // cv::imwrite(fname, wrap_frame, {JPEG_QUALITY, 98}), rtpose.cpp:1367-1381
      char fname[1024];
      if (F.image_dir.empty()) snprintf(fname, sizeof fname, "%s/frame%06d.jpg", F.write_frames.c_str(), fr.video_frame_number);
      else snprintf(fname, sizeof fname, "%s/%s.jpg", F.write_frames.c_str(), fr.stem.c_str());
      while (G.encode_queue.size() > 32) std::this_thread::sleep_for(std::chrono::milliseconds(1));  // back-pressure on the encoders
      G.encode_queue.push(EncodeJob{fname, std::move(fr.rendered)});
    }
    if (!F.write_json.empty()) {
      if (F.json_writers > 0) {
        while (G.json_queue.size() > 256) std::this_thread::sleep_for(std::chrono::microseconds(200));  // back-pressure on the JSON writers
        G.json_queue.push(std::move(fr));
      } else write_json_file(fr, buf);
    }
    G.finished++;
    {
      const double now = wall();
      double prev = G.last_written.load();
      while (prev < now && !G.last_written.compare_exchange_weak(prev, now)) {}
    }
    counter++;
    if (counter % 30 == 0) {  // rtpose.cpp:1421-1441
      const double now = wall();
      const double fps = 30.0 / (now - last_time);
      last_time = now;
      fprintf(stderr, "# %d, NP %d, Latency %.3f, Preprocess %.3f, QueueA %.3f, GPU %.3f, QueueB %.3f, Buffered %.3f, QueueD %.3f, FPS = %.1f\n",
              f_index, f_np, now - t_commit, t_pre - t_commit, t_fetch - t_pre, t_done - t_fetch, t_b0 - t_done, t_b1 - t_b0, now - t_b1, fps);
    }
  }
}

int read_image_dir() {  // readImageDirIfFlagEnabled, rtpose.cpp:1732-1755
  if (F.image_dir.empty()) return 0;
  DIR* d = opendir(F.image_dir.c_str());
  if (!d) { fprintf(stderr, "Folder %s does not exist.\n", F.image_dir.c_str()); return -1; }
  while (dirent* ent = readdir(d)) {
    const std::string n = ent->d_name;
    const size_t dot = n.find_last_of('.');
    if (dot == std::string::npos) continue;
    const std::string ext = n.substr(dot);
    std::string ext_l = ext;
    for (char& ch : ext_l) ch = (char)tolower((unsigned char)ch);
    if (ext_l == ".jpg" || ext_l == ".jpeg" || ext_l == ".png" || ext_l == ".bmp" || ext_l == ".ppm") G.image_list.push_back(F.image_dir + "/" + n);
  }
  closedir(d);
  std::sort(G.image_list.begin(), G.image_list.end());
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  const int pf = parse_flags(argc, argv, F);
  if (pf == 2) { usage(); return 0; }
  if (pf) return 1;
  if (sscanf(F.resolution.c_str(), "%dx%d", &DISP_W, &DISP_H) != 2) { fprintf(stderr, "Error, resolution format (%s) invalid, should be e.g., 960x540\n", F.resolution.c_str()); return 1; }
  if (sscanf(F.net_resolution.c_str(), "%dx%d", &NET_W, &NET_H) != 2) { fprintf(stderr, "Error, net resolution format (%s) invalid, should be e.g., 656x368 (multiples of 16)\n", F.net_resolution.c_str()); return 1; }
  if (read_image_dir() != 0) return 1;
  if (DISP_W == -1) {  // rtpose.cpp:1676-1693
    int w = 0, h = 0;
    if (!F.image_dir.empty() && !G.image_list.empty() && rtp_load_image(G.image_list[0].c_str(), nullptr, 0, &w, &h) == RTP_OK) { DISP_W = w; DISP_H = h; }
    else if (F.video.rfind("synthetic:", 0) == 0 && sscanf(F.video.c_str(), "synthetic:%dx%d", &w, &h) == 2) { DISP_W = w; DISP_H = h; }
    else if (!F.video.empty() && F.video.rfind("synthetic:", 0) != 0) {
      rtp_video* v = nullptr;
      if (rtp_video_open(F.video.c_str(), &v, &w, &h, nullptr) != RTP_OK) { fprintf(stderr, "Couldn't open video file %s: %s\n", F.video.c_str(), rtp_codec_last_error()); return 1; }
      rtp_video_close(v);
      DISP_W = w; DISP_H = h;
    }
    else { fprintf(stderr, "Invalid resolution without video/images: %dx%d\n", DISP_W, DISP_H); return 1; }
  }
  if (F.video.empty() && F.image_dir.empty()) { fprintf(stderr, "Couldn't open camera %d (no capture stack in this build): use --video or --image_dir\n", F.camera); return 1; }
  if (!F.video.empty() && F.video.rfind("synthetic:", 0) != 0) {
    rtp_video* v = nullptr;
    if (rtp_video_open(F.video.c_str(), &v, nullptr, nullptr, nullptr) != RTP_OK) { fprintf(stderr, "Couldn't open video file %s: %s\n", F.video.c_str(), rtp_codec_last_error()); return 1; }
    rtp_video_close(v);
  }
  for (const std::string* d : {&F.write_frames, &F.write_json})
    if (!d->empty() && !mkdir_p(*d)) { fprintf(stderr, "Could not write to or create directory %s\n", d->c_str()); return 1; }
  if (F.num_gpu < 1) { fprintf(stderr, "--num_gpu must be >= 1\n"); return 1; }
  if (F.precision != "mixed" && F.precision != "fp16" && F.precision != "f16x3" && F.precision != "fp32") { fprintf(stderr, "--precision must be mixed, fp16, f16x3 or fp32\n"); return 1; }
  if (F.frames_in_flight < 1 || F.frames_in_flight > 64) { fprintf(stderr, "--frames_in_flight must be in [1, 64]\n"); return 1; }
  if (F.batch_frames < 1 || F.batch_frames > 16) { fprintf(stderr, "--batch_frames must be in [1, 16]\n"); return 1; }
  // Hardware queues of the HIP runtime (read once, at the process's first HIP call — nothing has touched HIP yet; a value from the environment
  // wins): with at least as many queues as batch contexts the engine gives every context ONE stream and so one queue to itself: +12 % frames/s
  // at batches of 2 against the default 4 queues (engine.cpp "hardware queues").
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  if (F.json_writers < 0) F.json_writers = F.num_gpu;
  if (F.dry_engine < 0 || F.json_writers > 64) { fprintf(stderr, "--dry_engine must be >= 0 and --json_writers in [0, 64]\n"); return 1; }
  std::vector<int> devs;
  for (int g = 0; g < F.num_gpu; ++g) devs.push_back(g + F.start_device);  // rtpose.cpp:1466
  if (!F.devices.empty()) {
    devs.clear();
    size_t pos = 0;
    while (pos <= F.devices.size()) {
      const size_t c = F.devices.find(',', pos);
      const std::string tok = F.devices.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
      if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) { fprintf(stderr, "bad --devices '%s'\n", F.devices.c_str()); return 1; }
      devs.push_back(atoi(tok.c_str()));
      if (c == std::string::npos) break;
      pos = c + 1;
    }
    if ((int)devs.size() != F.num_gpu) { fprintf(stderr, "--devices names %zu devices but --num_gpu is %d\n", devs.size(), F.num_gpu); return 1; }
  }
  G.per_worker.assign(F.num_gpu, 0);
  if (!F.write_frames.empty() && F.host_preprocess) { fprintf(stderr, "--write_frames needs the device pre-processing path (drop --host_preprocess)\n"); return 1; }

  const double t0 = wall();
  std::vector<std::thread> workers;
  std::vector<int> status(F.num_gpu, 0);
  for (int g = 0; g < F.num_gpu; ++g) workers.emplace_back(worker, g, devs[g], &status[g]);
  std::atomic<bool> workers_done{false}, reorder_done{false};
  std::thread prod(producer), reo(reorderer, &workers_done), wr(writer, &reorder_done);
  std::vector<std::thread> encoders;
  if (!F.write_frames.empty()) for (int i = 0; i < 8; ++i) encoders.emplace_back(encoder);
  std::vector<std::thread> jsonw;
  if (!F.write_json.empty()) for (int i = 0; i < F.json_writers; ++i) jsonw.emplace_back(json_writer);
  prod.join();
  for (auto& t : workers) t.join();
  workers_done = true;
  reo.join();
  reorder_done = true;
  wr.join();
  G.encode_done = true;
  for (auto& t : encoders) t.join();
  G.json_done = true;
  for (auto& t : jsonw) t.join();
  int rc = 0;
  for (int s : status) rc |= s;
  const double dt = wall() - t0;
  for (int g = 0; g < F.num_gpu; ++g) fprintf(stderr, "worker %d (GPU %d) processed %d frames\n", g, devs[g], G.per_worker[g]);
  if (F.share_weights) fprintf(stderr, "share_weights: %d worker(s) took worker 0's packed weights\n", G.weights_shared.load());
  const double steady = (G.finished.load() > 1 && G.last_written.load() > G.first_commit) ? G.finished.load() / (G.last_written.load() - G.first_commit) : 0.0;
  fprintf(stderr, "rtcpm %s. Total time: %.3f seconds. frames produced %d, written %d, dropped %d (%.1f FPS incl. init, %.1f FPS first frame committed -> last frame written)\n",
          rc ? "FAILED" : "successfully finished", dt, G.produced.load(), G.finished.load(), G.dropped.load(), G.finished.load() / dt, steady);
  return rc;
}
