// caffe_stub.h — the few members of caffe::Blob / caffe::ConvolutionParameter that the extracted reference code touches
// (caffe_conv of test_convolution_layer.cpp, the MAX loop of PoolingLayer::Forward_cpu).  TEST INFRASTRUCTURE (oracle/_ref/libref.so).
// Semantics follow include/caffe/blob.hpp:52-220 (row-major offsets, CHECKed indices) and the protobuf accessors of
// caffe.proto:560-611 (repeated kernel_size / pad / stride / dilation, optional *_h / *_w, group, bias_term).
#pragma once
#include <vector>

#include "glog_stub.h"

namespace caffe {
template <typename Dtype>
class Blob {
 public:
  Blob() {}
  explicit Blob(const std::vector<int>& shape) { Reshape(shape); }
  void Reshape(const std::vector<int>& shape) {
    shape_ = shape;
    long n = 1;
    for (int d : shape) { CHECK_GE(d, 0); n *= d; }
    data_.assign((size_t)n, Dtype(0));
  }
  void Reshape(int n, int c, int h, int w) { Reshape(std::vector<int>{n, c, h, w}); }
  int num_axes() const { return (int)shape_.size(); }
  int shape(int i) const { CHECK_LT(i, num_axes()); return shape_[i]; }
  const std::vector<int>& shape() const { return shape_; }
  int num() const { return shape(0); }
  int channels() const { return shape(1); }
  int height() const { return shape(2); }
  int width() const { return shape(3); }
  int count() const { return (int)data_.size(); }
  int offset(int n, int c = 0, int h = 0, int w = 0) const { return ((n * channels() + c) * height() + h) * width() + w; }   // blob.hpp:158-170
  int offset(const std::vector<int>& indices) const {                                                                      // blob.hpp:172-185
    CHECK_LE(indices.size(), (size_t)num_axes());
    int off = 0;
    for (int i = 0; i < num_axes(); ++i) {
      off *= shape(i);
      if ((int)indices.size() > i) { CHECK_GE(indices[i], 0); CHECK_LT(indices[i], shape(i)); off += indices[i]; }
    }
    return off;
  }
  Dtype data_at(const std::vector<int>& index) const { return data_[offset(index)]; }
  const Dtype* cpu_data() const { return data_.data(); }
  Dtype* mutable_cpu_data() { return data_.data(); }

 private:
  std::vector<int> shape_;
  std::vector<Dtype> data_;
};

class ConvolutionParameter {
 public:
  int kernel = 1, pad_v = 0, stride_v = 1, dilation_v = 1, group_v = 1;
  int kh = -1, kw = -1, ph = -1, pw = -1, sh = -1, sw = -1;   // -1: the optional field is not set
  bool bias = true;
  bool has_kernel_h() const { return kh >= 0; }
  bool has_kernel_w() const { return kw >= 0; }
  int kernel_h() const { return kh; }
  int kernel_w() const { return kw; }
  int kernel_size(int) const { return kernel; }
  bool has_pad_h() const { return ph >= 0; }
  bool has_pad_w() const { return pw >= 0; }
  int pad_h() const { return ph; }
  int pad_w() const { return pw; }
  int pad_size() const { return 1; }
  int pad(int) const { return pad_v; }
  bool has_stride_h() const { return sh >= 0; }
  bool has_stride_w() const { return sw >= 0; }
  int stride_h() const { return sh; }
  int stride_w() const { return sw; }
  int stride_size() const { return 1; }
  int stride(int) const { return stride_v; }
  int dilation_size() const { return 1; }
  int dilation(int) const { return dilation_v; }
  int group() const { return group_v; }
  bool bias_term() const { return bias; }
};
}  // namespace caffe
