// ref_shim.cpp — C entry points around the REFERENCE'S OWN code, compiled where it lies under /root/reference.
// TEST INFRASTRUCTURE (oracle/_ref/libref.so): used only by tests/ to pin oracle/rtpose_oracle.cpp, and by
// tools/make_ref_golden.py to write tests/golden/ref_*.npz for the GPU box (where /root/reference does not exist).
//
// build_ref.sh cuts these line ranges out of the reference tree into oracle/_ref/gen/*.inc at build time (nothing
// from the reference is committed to this repository) and this file #includes them between stub declarations:
//   rtpose.cpp:144-152     struct ColumnCompare
//   rtpose.cpp:239-269     process_and_pad_image
//   rtpose.cpp:549-751     connectLimbs           (MPI_15)
//   rtpose.cpp:808-1076    connectLimbsCOCO       (COCO_18)
//   rtpose.cpp:1383-1416   the --write_json block of displayFrame
//   imresize_layer.cu:8-18, 98-155   cubic_interpolation, imresize_cubic_kernel
//   nms_layer.cu:14-113              nms_register_kernel, writeResultKernel
//   renderFunctions.cu:4-329, 394-975   the render kernels (pose overlay, heat-map and PAF views; MPI and COCO) with their colour maps
// modelDescriptor.cpp / modelDescriptorFactory.cpp are compiled as they are (own translation units).
// The launch sequences of ImResizeLayer::Forward_gpu (imresize_layer.cu:158-186) and NmsLayer::Forward_gpu
// (nms_layer.cu:117-184, thrust::exclusive_scan = a serial exclusive prefix sum) are restated below: they are
// kernel launches, which only exist as <<<>>> syntax in the reference.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "cuda_emul.h"
#include "cvmat_stub.h"
#include "glog_stub.h"
#include "rtpose/modelDescriptor.h"
#include "rtpose/modelDescriptorFactory.h"
#include "rtpose/renderFunctions.h"  // RENDER_MAX_PEOPLE (plain C header)

// ---- the globals of rtpose.cpp the extracted functions read (rtpose.cpp:75-131) ---------------------------------
int DISPLAY_RESOLUTION_WIDTH;
int DISPLAY_RESOLUTION_HEIGHT;
int NET_RESOLUTION_WIDTH;
int NET_RESOLUTION_HEIGHT;
const auto MAX_PEOPLE = RENDER_MAX_PEOPLE;
struct Global {
  std::vector<std::string> image_list;
  float nms_threshold;
  int connect_min_subset_cnt;
  float connect_min_subset_score;
  float connect_inter_threshold;
  int connect_inter_min_above_threshold;
};
Global global;
struct NetCopy { std::unique_ptr<ModelDescriptor> up_model_descriptor; };
std::vector<NetCopy> net_copies;
std::string FLAGS_write_json, FLAGS_image_dir;
struct Frame {  // include/caffe/cpm/frame.h:6-34, the fields the JSON block reads
  int numPeople, video_frame_number;
  float scale;
  std::shared_ptr<float[]> joints;
};
namespace boost { namespace filesystem {
struct path {  // p.stem().string() of rtpose.cpp:1390-1391
  std::string s;
  explicit path(const std::string& v) : s(v) {}
  path stem() const {
    const size_t sl = s.find_last_of('/');
    const std::string f = s.substr(sl == std::string::npos ? 0 : sl + 1);
    const size_t dot = f.find_last_of('.');
    return path(dot == std::string::npos || dot == 0 ? f : f.substr(0, dot));
  }
  std::string string() const { return s; }
};
}}  // namespace boost::filesystem

#include "gen/rtpose_144_152.inc"
#include "gen/rtpose_239_269.inc"
#include "gen/rtpose_549_751.inc"
#include "gen/rtpose_808_1076.inc"

static void ref_json_block(const Frame& frame) {
#include "gen/rtpose_1383_1416.inc"
}

namespace caffe {
#include "gen/imresize_8_18.inc"
#include "gen/imresize_98_155.inc"
#define NUMBER_THREADS_PER_BLOCK 256  /* nms_layer.cu:8 */
#include "gen/nms_14_113.inc"
inline int updiv(int a, int b) { return (a + b - 1) / b; }  // caffe/cpm/util/math_functions.hpp
}  // namespace caffe

// renderFunctions.cu: the kernels as host C++ (one barrier each, pre-barrier part idempotent: two-phase emulation).  The host
// wrappers render_mpi_parts / render_coco_parts / render_coco_aff (:331-389, :978-1080) are <<<>>> launches and are restated in
// ref_render below — INCLUDING their swapped launch arguments (<<<threadsPerBlock, numBlocks>>>: a 32x32 grid of blocks of
// updiv(w,32) x updiv(h,32) threads, which covers the same pixels).
namespace refrender {
#include "gen/render_4_329.inc"
#include "gen/render_394_975.inc"
}  // namespace refrender

namespace {
std::string g_err;
template <typename F> int guarded(F f) {
  try {
    return f();
  } catch (const std::exception& ex) {
    g_err = ex.what();
    return -1;
  }
}
void set_model(int model) {
  if (net_copies.empty()) net_copies.resize(1);
  ModelDescriptorFactory::createModelDescriptor(model == 0 ? ModelDescriptorFactory::Type::COCO_18 : ModelDescriptorFactory::Type::MPI_15,
                                               net_copies[0].up_model_descriptor);
}
}  // namespace

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

// ModelDescriptorFactory::createModelDescriptor (modelDescriptorFactory.cpp:4-61)
int ref_model_tables(int model, int* num_parts, int* num_limbs, int* limb_seq, int* map_idx) {
  return guarded([&] {
    set_model(model);
    ModelDescriptor* md = net_copies[0].up_model_descriptor.get();
    *num_parts = md->get_number_parts();
    *num_limbs = md->number_limb_sequence();
    const auto& l = md->get_limb_sequence();
    const auto& m = md->get_map_idx();
    for (size_t i = 0; i < l.size(); ++i) { limb_seq[i] = l[i]; map_idx[i] = m[i]; }
    return 0;
  });
}

// process_and_pad_image (rtpose.cpp:239-269)
int ref_process_and_pad_image(float* target, const unsigned char* bgr, int ow, int oh, int tw, int th, int normalize) {
  return guarded([&] {
    cv::Mat m;
    m.cols = ow; m.rows = oh; m.data = const_cast<unsigned char*>(bgr);
    process_and_pad_image(target, m, tw, th, normalize != 0);
    return 0;
  });
}

// connectLimbs / connectLimbsCOCO (rtpose.cpp:549-751, 808-1076), called as processFrame does (:1158-1166)
int ref_connect(int model, const float* heatmap, const float* peaks, int max_peaks, int net_w, int net_h, int disp_w, int disp_h,
                float inter_threshold, int inter_min_above, int min_subset_cnt, float min_subset_score, float* joints /* MAX_PEOPLE*num_parts*3 */) {
  return guarded([&] {
    set_model(model);
    NET_RESOLUTION_WIDTH = net_w; NET_RESOLUTION_HEIGHT = net_h;
    DISPLAY_RESOLUTION_WIDTH = disp_w; DISPLAY_RESOLUTION_HEIGHT = disp_h;
    global.connect_inter_threshold = inter_threshold;
    global.connect_inter_min_above_threshold = inter_min_above;
    global.connect_min_subset_cnt = min_subset_cnt;
    global.connect_min_subset_score = min_subset_score;
    std::vector<std::vector<double>> subset;
    std::vector<std::vector<std::vector<double>>> connection;
    ModelDescriptor* md = net_copies[0].up_model_descriptor.get();
    if (md->get_number_parts() == 15) return connectLimbs(subset, connection, heatmap, peaks, max_peaks, joints, md);
    return connectLimbsCOCO(subset, connection, heatmap, peaks, max_peaks, joints, md);
  });
}

// the --write_json block (rtpose.cpp:1383-1416): writes <dir>/frame%06d.json or <dir>/<stem>.json
int ref_write_json(const char* dir, const char* image_path_or_null, int video_frame_number, int model, const float* joints, int num_people, float frame_scale) {
  return guarded([&] {
    set_model(model);
    FLAGS_write_json = dir;
    FLAGS_image_dir = image_path_or_null ? "x" : "";
    global.image_list.assign((size_t)video_frame_number + 1, image_path_or_null ? image_path_or_null : "");
    Frame f;
    f.numPeople = num_people;
    f.video_frame_number = video_frame_number;
    f.scale = frame_scale;
    const int np = net_copies[0].up_model_descriptor->get_number_parts();
    f.joints = std::shared_ptr<float[]>(new float[(size_t)std::max(1, num_people) * np * 3]);
    for (int i = 0; i < num_people * np * 3; ++i) f.joints[i] = joints[i];
    ref_json_block(f);
    return 0;
  });
}

// ImResizeLayer::Forward_gpu (imresize_layer.cu:158-186): one launch of imresize_cubic_kernel per channel
int ref_imresize(const float* src, int num, int channel, int oriSpatialHeight, int oriSpatialWidth, int targetSpatialWidth, int targetSpatialHeight,
                 float start_scale, float scale_gap, float* dst) {
  return guarded([&] {
    const dim3 threadsPerBlock(16, 16);
    const dim3 numBlocks(caffe::updiv(targetSpatialWidth, threadsPerBlock.x), caffe::updiv(targetSpatialHeight, threadsPerBlock.y));
    const int offset_src = oriSpatialHeight * oriSpatialWidth;
    const int offset_dst = targetSpatialWidth * targetSpatialHeight;
    for (int c = 0; c < channel; c++)
      cuda_emul::emu_launch(numBlocks, threadsPerBlock, 1, [&] {
        caffe::imresize_cubic_kernel<float>(src + c * offset_src, dst + c * offset_dst, channel * offset_src, num, scale_gap, start_scale,
                                            oriSpatialWidth, oriSpatialHeight, targetSpatialWidth, targetSpatialHeight);
      });
    return 0;
  });
}

// NmsLayer::Forward_gpu (nms_layer.cu:117-184), num = 1.  `top` is [num_parts][max_peaks+1][3] and is NOT cleared (stale slots stay).
// Height and width must be multiples of 16 (net resolutions are): with partial 16x16 blocks nms_register_kernel's
// `else if` writes outside its row (a race on the GPU).
int ref_nms(const float* bottom, int height, int width, int num_parts, int max_peaks, float threshold, float* top) {
  return guarded([&] {
    if (height % 16 || width % 16) throw std::runtime_error("ref_nms: height and width must be multiples of 16");
    const int offset = height * width;
    const int offset_dst = (max_peaks + 1) * 3;
    std::vector<int> workspace((size_t)num_parts * offset, 0);
    const dim3 threadsPerBlock(16, 16);
    const dim3 numBlocks(caffe::updiv(width, threadsPerBlock.x), caffe::updiv(height, threadsPerBlock.y));
    for (int c = 0; c < num_parts; c++) {
      int* w_pointer1 = workspace.data() + (size_t)c * offset;
      const float* src = bottom + (size_t)c * offset;
      float* dst = top + (size_t)c * offset_dst;
      cuda_emul::emu_launch(numBlocks, threadsPerBlock, 1, [&] { caffe::nms_register_kernel<float>(src, w_pointer1, width, height, threshold); });
      int run = 0;  // thrust::exclusive_scan(dev_ptr, dev_ptr + offset, dev_ptr)
      for (int i = 0; i < offset; ++i) { const int v = w_pointer1[i]; w_pointer1[i] = run; run += v; }
      cuda_emul::emu_launch(dim3(caffe::updiv(offset, 256)), dim3(256), 2,
                            [&] { caffe::writeResultKernel<float>(offset, w_pointer1, src, dst, width, max_peaks); });
    }
    return 0;
  });
}

// render() of rtpose.cpp:270-299 on one frame, between the producer's canvas (process_and_pad_image(.., normalize = 0), :349) and the
// post-processing thread's float -> u8 conversion (:1287-1296).  `heatmaps` = the resized map [C][net_h][net_w] (only read for part != 0).
int ref_render(int model, const unsigned char* in_bgr, int w, int h, int net_w, int net_h, const float* heatmaps, const float* poses,
               int num_people, int part_to_show, int googly, unsigned char* out_bgr) {
  return guarded([&] {
    using namespace refrender;
    std::vector<float> canvas((size_t)w * h * 3);
    {
      cv::Mat m;
      m.cols = w; m.rows = h; m.data = const_cast<unsigned char*>(in_bgr);
      process_and_pad_image(canvas.data(), m, w, h, false);
    }
    std::vector<float> pose_copy(poses, poses + (size_t)std::max(1, num_people) * (model == 0 ? 18 : 15) * 3);
    float* hm = const_cast<float*>(heatmaps);
    const dim3 threadsPerBlock(numThreadsPerBlock_1d, numThreadsPerBlock_1d);
    const dim3 numBlocks(caffe::updiv(w, threadsPerBlock.x), caffe::updiv(h, threadsPerBlock.y));
    const float ratio_to_origin = (float)h / (float)net_h;
    float* cv = canvas.data();
    float* ps = pose_copy.data();
    const int boxsize = 368;
    if (model != 0) {  // render_mpi_parts :331-389
      const float threshold = 0.0;
      if (part_to_show == 0) {
        if (num_people != 0)
          cuda_emul::emu_launch(threadsPerBlock, numBlocks, 2, [&] { render_pose_29parts(cv, w, h, ratio_to_origin, ps, boxsize, num_people, threshold); });
      } else if (part_to_show > 0) {
        cuda_emul::emu_launch(threadsPerBlock, numBlocks, 2, [&] { render_pose_29parts_heatmap(cv, w, h, net_w, net_h, hm, num_people, part_to_show - 1); });
      }
    } else if (part_to_show - 1 <= 18) {  // render_coco_parts :978-1036
      const float threshold = 0.01;
      const int NUM_PARTS = 18;
      if (part_to_show == 0) {
        if (num_people != 0)
          cuda_emul::emu_launch(threadsPerBlock, numBlocks, 2,
                                [&] { render_pose_coco_parts(cv, w, h, ratio_to_origin, ps, boxsize, num_people, threshold, googly != 0); });
      } else if (part_to_show > 0 && part_to_show < 58) {
        if (part_to_show - 1 == NUM_PARTS)
          cuda_emul::emu_launch(threadsPerBlock, numBlocks, 2, [&] { render_pose_coco_heatmap2(cv, w, h, net_w, net_h, hm, num_people, 0); });
        else
          cuda_emul::emu_launch(threadsPerBlock, numBlocks, 2, [&] { render_pose_coco_heatmap(cv, w, h, net_w, net_h, hm, num_people, part_to_show - 1); });
      }
    } else {  // rtpose.cpp:286-297 -> render_coco_aff :1038-1080
      int aff_part = ((part_to_show - 1) - 18 - 1) * 2;
      int num_parts_accum = 1;
      if (aff_part == 0) num_parts_accum = 19;
      else aff_part = aff_part - 2;
      aff_part += 1 + 18;
      cuda_emul::emu_launch(threadsPerBlock, numBlocks, 2, [&] { render_pose_coco_affinity(cv, w, h, net_w, net_h, hm, num_parts_accum, num_people, aff_part); });
    }
    const int offset = w * h;
    for (int c = 0; c < 3; c++)
      for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
          int value = int(canvas[(size_t)c * offset + i * w + j] + 0.5);
          value = value < 0 ? 0 : (value > 255 ? 255 : value);
          out_bgr[3 * (i * w + j) + c] = (unsigned char)(value);
        }
    return 0;
  });
}

}  // extern "C"
