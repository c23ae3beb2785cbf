// host_api.h — C++ mirror of the slice of the Caffe Net API that examples/rtpose/rtpose.cpp
// consumes (rtpose.cpp:183-207, 1093-1094, 1131-1150), implemented over the C-ABI of
// include/rtpose_mi355x.h.  A maintainer who wants to keep rtpose.cpp's source shape can include
// this header instead of caffe/net.hpp + caffe/cpm/layers/*.hpp; same names, same argument meaning.
// Errors throw std::runtime_error (the reference glog-CHECK-aborts in the same places).
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtpose_mi355x.h"

namespace rtpose {

enum Phase { TRAIN = 0, TEST = 1 };

class Net;

// caffe::NmsLayer<float> (include/caffe/cpm/layers/nms_layer.hpp:11-44)
class NmsLayer {
 public:
  explicit NmsLayer(Net* n) : net_(n) {}
  int GetMaxPeaks() const;
  int GetNumParts() const;
  void SetThreshold(float t);
 private:
  Net* net_;
};

// caffe::ImResizeLayer<float> (include/caffe/cpm/layers/imresize_layer.hpp:11-45)
class ImResizeLayer {
 public:
  explicit ImResizeLayer(Net* n) : net_(n) {}
  void SetStartScale(float s);
  void SetScaleGap(float g);
 private:
  Net* net_;
};

// caffe::Blob<float>: host view of "image", "resized_map" or "joints"
class Blob {
 public:
  std::vector<int> shape_;
  std::vector<float> data_;
  const std::vector<int>& shape() const { return shape_; }
  int shape(int i) const { return shape_[i]; }
  float* mutable_cpu_data() { return data_.data(); }
  const float* cpu_data() const { return data_.data(); }
  void Reshape(const std::vector<int>& s) {
    shape_ = s;
    size_t n = 1;
    for (int d : s) n *= (size_t)d;
    data_.assign(n, 0.f);
  }
};

// caffe::Net<float> as rtpose.cpp uses it: construct, CopyTrainedLayersFrom, reshape the input blob,
// grab the "nms"/"resize" layers, fill blobs()[0], ForwardFrom(0), read "resized_map"/"joints".
class Net {
 public:
  Net(const std::string& proto, Phase, int device_id = 0) : proto_(proto), device_(device_id) {
    rtp_config_default(&cfg_);
    cfg_.device_id = device_id;
    image_.reset(new Blob());
    resized_.reset(new Blob());
    joints_.reset(new Blob());
  }
  ~Net() { if (e_) rtp_engine_destroy(e_); }
  void CopyTrainedLayersFrom(const std::string& caffemodel) { weights_ = caffemodel; }
  void set_precision(int p) { cfg_.precision = p; }
  void set_display_resolution(int w, int h) { cfg_.disp_w = w; cfg_.disp_h = h; }
  std::vector<std::shared_ptr<Blob>> blobs() { return {image_}; }
  // rtpose.cpp:188-191: blobs()[0]->Reshape({N,3,H,W}); Reshape();
  void Reshape() {
    const std::vector<int>& s = image_->shape();
    if (s.size() != 4 || s[1] != 3) throw std::runtime_error("input blob must be N x 3 x H x W");
    if (e_) { rtp_engine_destroy(e_); e_ = nullptr; }
    cfg_.num_scales = s[0]; cfg_.net_h = s[2]; cfg_.net_w = s[3];
    cfg_.proto_path = proto_.empty() ? nullptr : proto_.c_str();
    cfg_.weights_path = weights_.empty() ? nullptr : weights_.c_str();
    cfg_.start_scale = start_scale_; cfg_.scale_gap = scale_gap_;
    cfg_.frames_in_flight = 1;
    if (rtp_engine_create(&cfg_, &e_) != RTP_OK) throw std::runtime_error(rtp_last_error(nullptr));
    int hc, lw, lh;
    rtp_engine_info(e_, &num_parts_, &max_peaks_, &hc, &lw, &lh);
    resized_->Reshape({1, hc, cfg_.net_h, cfg_.net_w});
    joints_->Reshape({1, num_parts_, max_peaks_ + 1, 3});
  }
  template <class L> std::shared_ptr<L> layer_by_name(const std::string& name);
  std::shared_ptr<Blob> blob_by_name(const std::string& name) {
    if (name == "resized_map") return resized_;
    if (name == "joints") return joints_;
    if (name == "image") return image_;
    throw std::runtime_error("Unknown blob name " + name);
  }
  // net.cpp:544-560 ForwardFrom(0): whole forward; afterwards "resized_map" and "joints" hold
  // what heatmap_blob->mutable_cpu_data() / joints_blob->mutable_cpu_data() return (rtpose.cpp:1149-1150)
  float ForwardFrom(int) {
    need();
    int n = 0;
    std::vector<float> j((size_t)RTP_MAX_PEOPLE * num_parts_ * 3);
    if (rtp_forward_debug(e_, image_->cpu_data(), nullptr, resized_->mutable_cpu_data(), joints_->mutable_cpu_data(), j.data(), &n) != RTP_OK)
      throw std::runtime_error(rtp_last_error(e_));
    return 0.f;
  }
  rtp_engine* engine() { need(); return e_; }

 private:
  friend class NmsLayer;
  friend class ImResizeLayer;
  void need() { if (!e_) Reshape(); }
  std::string proto_, weights_;
  int device_;
  rtp_config cfg_;
  rtp_engine* e_ = nullptr;
  std::shared_ptr<Blob> image_, resized_, joints_;
  int num_parts_ = 18, max_peaks_ = 64;
  float start_scale_ = 1.f, scale_gap_ = 0.3f;
};

inline int NmsLayer::GetMaxPeaks() const { net_->need(); return net_->max_peaks_; }
inline int NmsLayer::GetNumParts() const { net_->need(); return net_->num_parts_; }
inline void NmsLayer::SetThreshold(float t) {
  net_->need();
  float a, b, e; int c, d;
  rtp_get_thresholds(net_->e_, &a, &b, &c, &d, &e);
  rtp_set_thresholds(net_->e_, t, b, c, d, e);
}
inline void ImResizeLayer::SetStartScale(float s) { net_->start_scale_ = s; if (net_->e_) rtp_set_scales(net_->e_, net_->start_scale_, net_->scale_gap_); }
inline void ImResizeLayer::SetScaleGap(float g) { net_->scale_gap_ = g; if (net_->e_) rtp_set_scales(net_->e_, net_->start_scale_, net_->scale_gap_); }
template <> inline std::shared_ptr<NmsLayer> Net::layer_by_name<NmsLayer>(const std::string& name) {
  if (name != "nms") throw std::runtime_error("Unknown layer name " + name);
  return std::make_shared<NmsLayer>(this);
}
template <> inline std::shared_ptr<ImResizeLayer> Net::layer_by_name<ImResizeLayer>(const std::string& name) {
  if (name != "resize") throw std::runtime_error("Unknown layer name " + name);
  return std::make_shared<ImResizeLayer>(this);
}

}  // namespace rtpose
