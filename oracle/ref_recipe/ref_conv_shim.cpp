// ref_conv_shim.cpp — C entry points around the reference's OWN convolution / pooling code (second translation unit of
// oracle/_ref/libref.so; TEST INFRASTRUCTURE, see ref_shim.cpp).  build_ref.sh cuts, at build time, into oracle/_ref/gen/:
//   src/caffe/util/im2col.cpp:14-55                   is_a_ge_zero_and_a_lt_b, im2col_cpu
//   src/caffe/test/test_convolution_layer.cpp:21-139  caffe_conv — the naive convolution the reference's own tests trust
//   src/caffe/layers/pooling_layer.cpp:90-107          the pooled output size of PoolingLayer::Reshape (ceil mode + pad clip) and the Reshape of the top blob
//   src/caffe/layers/pooling_layer.cpp:149-186         the MAX branch of PoolingLayer::Forward_cpu (initialisation + main loop)
// The two pooling ranges are statements of member functions: they are included into a function whose LOCALS carry the members' names.
// What cannot come from the reference: the GEMM behind forward_cpu_gemm (base_conv_layer.cpp:257-280 calls cblas_sgemm; no BLAS is
// installed here) — ref_im2col_conv multiplies the reference's im2col buffer with the weight matrix in a plain row-major loop
// (M = Cout, N = H_out * W_out, K = Cin * kh * kw, exactly the call's shapes) and adds the bias as forward_cpu_bias does (:273-280).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "caffe_stub.h"

namespace caffe {
using std::max;
using std::min;
using std::shared_ptr;
using std::vector;

template <typename Dtype>
void caffe_set(const int N, const Dtype alpha, Dtype* Y) {  // math_functions.cpp:56-65
  for (int i = 0; i < N; ++i) Y[i] = alpha;
}

#include "gen/im2col_14_55.inc"
#include "gen/caffe_conv_21_139.inc"

template <typename Dtype>
void pooling_max(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top, int kernel_h_, int kernel_w_, int stride_h_, int stride_w_,
                 int pad_h_, int pad_w_) {
  const int channels_ = bottom[0]->channels(), height_ = bottom[0]->height(), width_ = bottom[0]->width();
  int pooled_height_ = 0, pooled_width_ = 0;
#include "gen/pooling_90_107.inc"
  // the locals Forward_cpu sets up before its switch (pooling_layer.cpp:129-137)
  const Dtype* bottom_data = bottom[0]->cpu_data();
  Dtype* top_data = top[0]->mutable_cpu_data();
  const int top_count = top[0]->count();
  const bool use_top_mask = false;
  std::vector<int> max_idx((size_t)top_count, -1);
  int* mask = max_idx.data();
  Dtype* top_mask = NULL;
#include "gen/pooling_149_186.inc"
}
}  // namespace caffe

static std::string g_conv_err;
extern "C" {
#define REF_API __attribute__((visibility("default")))
REF_API const char* ref_conv_last_error() { return g_conv_err.c_str(); }

// caffe_conv on x [N][Cin][H][W], w [Cout][Cin][kh][kw], b [Cout] or NULL -> out [N][Cout][Ho][Wo] (zero-initialised here, as the
// tests do through Blob construction: caffe_conv accumulates into it)
REF_API int ref_caffe_conv(const float* x, int N, int Cin, int H, int W, const float* w, const float* b, int Cout, int kh, int kw, int pad_h, int pad_w,
                           int stride_h, int stride_w, float* out) {
  try {
    using namespace caffe;
    Blob<float> in(std::vector<int>{N, Cin, H, W});
    memcpy(in.mutable_cpu_data(), x, sizeof(float) * in.count());
    std::vector<std::shared_ptr<Blob<float> > > weights(2);
    weights[0].reset(new Blob<float>(std::vector<int>{Cout, Cin, kh, kw}));
    memcpy(weights[0]->mutable_cpu_data(), w, sizeof(float) * weights[0]->count());
    weights[1].reset(new Blob<float>(std::vector<int>{Cout}));
    if (b) memcpy(weights[1]->mutable_cpu_data(), b, sizeof(float) * Cout);
    ConvolutionParameter p;
    p.kh = kh; p.kw = kw; p.ph = pad_h; p.pw = pad_w; p.sh = stride_h; p.sw = stride_w; p.bias = b != nullptr;
    const int Ho = (H + 2 * pad_h - kh) / stride_h + 1, Wo = (W + 2 * pad_w - kw) / stride_w + 1;   // conv_layer.cpp:8-22
    Blob<float> o(std::vector<int>{N, Cout, Ho, Wo});
    caffe_conv(&in, &p, weights, &o);
    memcpy(out, o.cpu_data(), sizeof(float) * o.count());
    return 0;
  } catch (const std::exception& ex) { g_conv_err = ex.what(); return -1; }
}

// im2col_cpu (the reference's) + the GEMM / bias shapes of forward_cpu_gemm / forward_cpu_bias
REF_API int ref_im2col_conv(const float* x, int N, int Cin, int H, int W, const float* w, const float* b, int Cout, int kh, int kw, int pad_h, int pad_w,
                            int stride_h, int stride_w, float* out) {
  try {
    const int Ho = (H + 2 * pad_h - kh) / stride_h + 1, Wo = (W + 2 * pad_w - kw) / stride_w + 1;
    const long K = (long)Cin * kh * kw, S = (long)Ho * Wo;
    std::vector<float> col((size_t)(K * S));
    for (int n = 0; n < N; ++n) {
      caffe::im2col_cpu<float>(x + (long)n * Cin * H * W, Cin, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, 1, 1, col.data());
      float* o = out + (long)n * Cout * S;
      for (int m = 0; m < Cout; ++m)
        for (long s = 0; s < S; ++s) {
          float acc = 0.f;
          for (long k = 0; k < K; ++k) acc += w[(long)m * K + k] * col[(size_t)(k * S + s)];
          o[m * S + s] = acc;
        }
      if (b)
        for (int m = 0; m < Cout; ++m)
          for (long s = 0; s < S; ++s) o[m * S + s] += b[m] * 1.f;   // bias x bias_multiplier (ones), K = 1
    }
    return 0;
  } catch (const std::exception& ex) { g_conv_err = ex.what(); return -1; }
}

// the im2col buffer itself: [Cin*kh*kw][Ho*Wo] for one image (the K order the oracle's convolution must follow)
REF_API int ref_im2col(const float* x, int Cin, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, float* col) {
  caffe::im2col_cpu<float>(x, Cin, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, 1, 1, col);
  return 0;
}

REF_API int ref_maxpool(const float* x, int N, int C, int H, int W, int k, int stride, int pad, float* out, int* ho, int* wo) {
  try {
    using namespace caffe;
    Blob<float> in(std::vector<int>{N, C, H, W}), o;
    memcpy(in.mutable_cpu_data(), x, sizeof(float) * in.count());
    std::vector<Blob<float>*> bottom{&in}, top{&o};
    pooling_max(bottom, top, k, k, stride, stride, pad, pad);
    if (ho) *ho = o.height();
    if (wo) *wo = o.width();
    if (out) memcpy(out, o.cpu_data(), sizeof(float) * o.count());
    return 0;
  } catch (const std::exception& ex) { g_conv_err = ex.what(); return -1; }
}
}
