// netdef.h — the deploy graph (what Net::Init builds from a prototxt, net.cpp:49) and the
// weight container (what CopyTrainedLayersFrom reads, net.cpp:750-803), without protobuf.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace rtp {

struct LayerDef {
  std::string name, type;  // Convolution | ReLU | Pooling | Concat | ImResize | Nms | Input | Split
  std::vector<std::string> bottoms, tops;
  // ConvolutionParameter
  int num_output = 0, kernel = 0, pad = 0, stride = 1;
  bool bias_term = true;
  // PoolingParameter
  int pool_kernel = 0, pool_stride = 1, pool_pad = 0;
  std::string pool_method = "MAX";
  // ReLUParameter
  float negative_slope = 0.f;
  // ConcatParameter
  int axis = 1;
  // ImResizeParameter (caffe.proto:1478-1484)
  int target_w = 368, target_h = 368;
  float factor = 0.f, start_scale = 1.f, scale_gap = 0.1f;
  // NmsParameter (caffe.proto:1471-1476)
  float nms_threshold = 0.5f;
  int max_peaks = 20, num_parts = 15;
};

struct NetDef {
  std::string name;
  std::vector<std::string> inputs;
  std::vector<int> input_dim;  // legacy header: 4 per input
  std::vector<LayerDef> layers;
};

// The two nets in scope, generated in code (the reference's model/{coco,mpi}/
// pose_deploy_linevec.prototxt describe the same graphs; tests compare them layer by layer).
NetDef build_linevec(int model);
bool parse_prototxt(const std::string& text, NetDef* out, std::string* err);
std::string emit_prototxt(const NetDef& net);

struct BlobData {
  std::vector<int64_t> shape;
  std::vector<float> data;
};
struct LayerWeights {
  std::string name, type;
  std::vector<BlobData> blobs;
};
// Binary caffe NetParameter (caffe.proto:64-95): field 100 `layer` (LayerParameter: name=1,
// type=2, blobs=7) and the deprecated field 2 `layers` (V1LayerParameter: name=4, blobs=6).
// BlobProto (caffe.proto:10-22): shape=7{dim=1 packed}, data=5 packed, legacy num/channels/
// height/width = 1..4.
bool read_caffemodel(const std::string& path, std::vector<LayerWeights>* out, std::string* err);
bool write_caffemodel(const std::string& path, const std::string& net_name, const std::vector<LayerWeights>& layers,
                      std::string* err);

// Deterministic synthetic weights (no trained .caffemodel ships with the reference,
// model/getModels.sh:2,5): w ~ N(0, sqrt(2/(cin*k*k))) via a 12-uniform sum, b ~ U(-0.1, 0.1),
// splitmix64 seeded with seed ^ fnv1a(layer name).  Pure integer + exact float ops: the same
// bits on every machine.
void synth_conv_weights(uint64_t seed, const std::string& layer_name, int cout, int cin, int k, std::vector<float>* w,
                        std::vector<float>* b);

}  // namespace rtp
