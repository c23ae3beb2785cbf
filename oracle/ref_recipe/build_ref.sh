#!/bin/bash
# build_ref.sh — oracle/_ref/libref.so from the reference's OWN sources, compiled where they lie.
# Needs /root/reference (this container only; the GPU box uses the prebuilt .so, which travels with gpurun).
# Nothing from the reference tree is copied into the repository: the cut-out line ranges live under
# oracle/_ref/gen/ only while this script runs.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${RTP_REFERENCE:-/root/reference}"
OUT="$HERE/../_ref"
[ -d "$REF" ] || { echo "build_ref.sh: $REF not present, nothing to build"; exit 0; }
mkdir -p "$OUT/gen"
RT="$REF/examples/rtpose/rtpose.cpp"
IM="$REF/src/caffe/cpm/layers/imresize_layer.cu"
NM="$REF/src/caffe/cpm/layers/nms_layer.cu"
RF="$REF/src/rtpose/renderFunctions.cu"

cut_range() {  # file first last expected-first-line-regex expected-last-line-regex output
  local f="$1" a="$2" b="$3" ra="$4" rb="$5" o="$6"
  sed -n "${a}p" "$f" | grep -Eq "$ra" || { echo "build_ref.sh: $f:$a does not match /$ra/ — the reference moved, fix the ranges"; exit 1; }
  sed -n "${b}p" "$f" | grep -Eq "$rb" || { echo "build_ref.sh: $f:$b does not match /$rb/ — the reference moved, fix the ranges"; exit 1; }
  sed -n "${a},${b}p" "$f" > "$OUT/gen/$o"
}
cut_range "$RT" 144 152 '^struct ColumnCompare' '^};' rtpose_144_152.inc
cut_range "$RT" 239 269 '^void process_and_pad_image' '^}' rtpose_239_269.inc
cut_range "$RT" 549 751 '^int connectLimbs\(' '^}' rtpose_549_751.inc
cut_range "$RT" 808 1076 '^int connectLimbsCOCO\(' '^}' rtpose_808_1076.inc
cut_range "$RT" 1383 1416 'FLAGS_write_json.empty' '^        }' rtpose_1383_1416.inc
cut_range "$IM" 8 18 '^template <typename Dtype>' '^}' imresize_8_18.inc
cut_range "$IM" 98 155 '^template <typename Dtype>' '^}' imresize_98_155.inc
cut_range "$NM" 14 113 '^template <typename Dtype>' '^}' nms_14_113.inc
cut_range "$RF" 4 329 '^#define numThreadsPerBlock_1d 32' '^}' render_4_329.inc       # macros, colour maps, MPI kernels
cut_range "$REF/src/caffe/util/im2col.cpp" 14 55 '^inline bool is_a_ge_zero_and_a_lt_b' '^}' im2col_14_55.inc
cut_range "$REF/src/caffe/test/test_convolution_layer.cpp" 21 139 '^template <typename Dtype>' '^}' caffe_conv_21_139.inc
cut_range "$REF/src/caffe/layers/pooling_layer.cpp" 90 107 'pooled_height_ = static_cast<int>' '^      pooled_width_\);' pooling_90_107.inc
cut_range "$REF/src/caffe/layers/pooling_layer.cpp" 149 186 'caffe_set\(top_count, Dtype\(-FLT_MAX\), top_data\)' '^    }' pooling_149_186.inc
cut_range "$RF" 394 975 '^__global__ void render_pose_coco_parts' '^}' render_394_975.inc  # COCO kernels (pose, heat maps, PAFs)

CXX="${CXX:-g++}"
# -ffp-contract=off and no -ffast-math: the floating point of the C++ source, nothing else
FLAGS="-O2 -std=c++17 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -w -I$HERE -I$OUT -I$REF/include"
$CXX $FLAGS -shared -o "$OUT/libref.so" "$HERE/ref_shim.cpp" "$HERE/ref_conv_shim.cpp" "$REF/src/rtpose/modelDescriptor.cpp" "$REF/src/rtpose/modelDescriptorFactory.cpp"
rm -rf "$OUT/gen"
echo "built $OUT/libref.so"
