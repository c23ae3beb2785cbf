// host_util.cpp — the host-only pieces of the path behind the C-ABI: model descriptor tables,
// default thresholds, process_and_pad_image and the JSON body.  No GPU needed.
#include <charconv>
#include <cstring>
#include <sstream>
#include <string>

#include "../../include/rtpose_mi355x.h"

namespace {
// ModelDescriptorFactory::createModelDescriptor (src/rtpose/modelDescriptorFactory.cpp:25-26, 52-53)
const int kCocoLimb[38] = {1, 2, 1, 5, 2, 3, 3, 4, 5, 6, 6, 7, 1, 8, 8, 9, 9, 10, 1, 11, 11, 12, 12, 13, 1, 0, 0, 14, 14, 16, 0, 15, 15, 17, 2, 16, 5, 17};
const int kCocoMap[38] = {31, 32, 39, 40, 33, 34, 35, 36, 41, 42, 43, 44, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 47, 48, 49, 50, 53, 54, 51, 52, 55, 56, 37, 38, 45, 46};
const int kMpiLimb[28] = {0, 1, 1, 2, 2, 3, 3, 4, 1, 5, 5, 6, 6, 7, 1, 14, 14, 11, 11, 12, 12, 13, 14, 8, 8, 9, 9, 10};
const int kMpiMap[28] = {16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 38, 39, 40, 41, 42, 43, 32, 33, 34, 35, 36, 37};
}  // namespace

extern "C" {

int rtp_model_tables(int model, int* num_parts, int* num_limbs, int* limb_seq, int* map_idx) {
  if (model == RTP_MODEL_COCO_18) {
    if (num_parts) *num_parts = 18;
    if (num_limbs) *num_limbs = 19;
    if (limb_seq) memcpy(limb_seq, kCocoLimb, sizeof kCocoLimb);
    if (map_idx) memcpy(map_idx, kCocoMap, sizeof kCocoMap);
    return RTP_OK;
  }
  if (model == RTP_MODEL_MPI_15) {
    if (num_parts) *num_parts = 15;
    if (num_limbs) *num_limbs = 14;
    if (limb_seq) memcpy(limb_seq, kMpiLimb, sizeof kMpiLimb);
    if (map_idx) memcpy(map_idx, kMpiMap, sizeof kMpiMap);
    return RTP_OK;
  }
  return RTP_EINVAL;  // "Undefined ModelDescriptor selected." (modelDescriptorFactory.cpp:59)
}

// warmup(), rtpose.cpp:212-226
int rtp_default_thresholds(int model, float* nms_threshold, float* connect_inter_threshold, int* connect_inter_min_above_threshold,
                           int* connect_min_subset_cnt, float* connect_min_subset_score) {
  if (model != RTP_MODEL_COCO_18 && model != RTP_MODEL_MPI_15) return RTP_EINVAL;
  const bool mpi = model == RTP_MODEL_MPI_15;
  if (nms_threshold) *nms_threshold = mpi ? 0.2f : 0.05f;
  if (connect_inter_threshold) *connect_inter_threshold = mpi ? 0.01f : 0.050f;
  if (connect_inter_min_above_threshold) *connect_inter_min_above_threshold = mpi ? 8 : 9;
  if (connect_min_subset_cnt) *connect_min_subset_cnt = 3;
  if (connect_min_subset_score) *connect_min_subset_score = 0.4f;
  return RTP_OK;
}

// process_and_pad_image, rtpose.cpp:239-269
int rtp_process_and_pad_image(float* target, const unsigned char* bgr, int ow, int oh, int tw, int th, int normalize) {
  if (!target || !bgr || ow < 0 || oh < 0 || tw <= 0 || th <= 0) return RTP_EINVAL;
  const int plane = tw * th;
  const int padw = (tw - ow) / 2, padh = (th - oh) / 2;
  if (padw < 0 || padh < 0) return RTP_EINVAL;  // "Image too big for target size."
  for (int c = 0; c < 3; ++c) {
    float* t = target + (size_t)c * plane;
    for (int y = 0; y < th; ++y) {
      const int oy = y - padh;
      const bool row_in = oy >= 0 && oy < oh;
      for (int x = 0; x < tw; ++x) {
        const int ox = x - padw;
        float v = 0.f;
        if (row_in && ox >= 0 && ox < ow) {
          const float p = (float)bgr[((size_t)oy * ow + ox) * 3 + c];
          v = normalize ? p / 256.0f - 0.5f : p;
        }
        t[(size_t)y * tw + x] = v;
      }
    }
  }
  return RTP_OK;
}

// displayFrame's JSON body, rtpose.cpp:1394-1415: std::ofstream at default precision, i.e. every number as printf("%g")
// (floatfield unset, precision 6).  std::to_chars(general, 6) is specified to produce exactly that text and costs a quarter of
// an ostream insertion: 76 people x 54 numbers (the noise-map worst case) took 1.2 ms per frame through std::ostringstream —
// one writer thread would have capped an 8-GPU node at ~800 frames/s.  tests/test_ref_golden.py: byte-identical to the
// reference's own block, edge values included.
long rtp_format_json(char* buf, size_t buflen, const float* joints, int num_people, int num_parts, float frame_scale) {
  if (!buf || (num_people > 0 && !joints) || num_people < 0 || num_parts <= 0) return RTP_EINVAL;
  const double scale = 1.0 / frame_scale;
  char* p = buf;
  char* const end = buf + buflen;
  auto lit = [&](const char* s) { const size_t n = strlen(s); if ((size_t)(end - p) < n + 1) return false; memcpy(p, s, n); p += n; return true; };
  auto num = [&](double v) {
    if (end - p < 40) return false;
    const std::to_chars_result r = std::to_chars(p, end - 1, v, std::chars_format::general, 6);
    if (r.ec != std::errc()) return false;
    p = r.ptr;
    return true;
  };
  bool ok = lit("{\n\"version\":0.1,\n\"bodies\":[\n");
  for (int ip = 0; ok && ip < num_people; ip++) {
    ok = lit("{\n\"joints\":[");
    for (int ij = 0; ok && ij < num_parts; ij++) {
      const float* j = joints + ((size_t)ip * num_parts + ij) * 3;
      ok = num(scale * j[0]) && lit(",") && num(scale * j[1]) && lit(",") && num((double)j[2]) && (ij == num_parts - 1 || lit(","));
    }
    ok = ok && lit("]\n}") && (ip == num_people - 1 || lit(",\n"));
  }
  ok = ok && lit("]\n}\n");
  if (!ok) return RTP_ERANGE;
  *p = 0;
  return (long)(p - buf);
}

}  // extern "C"
