// cuda_emul.h — run the reference's CUDA kernels (imresize_layer.cu, nms_layer.cu) as host C++.
// TEST INFRASTRUCTURE.  The kernel TEXT is compiled unchanged by g++; this header supplies
//   * the qualifiers (__global__, __device__, __shared__ = per-OS-thread static: the threads of a block run one
//     after the other on one OS thread),
//   * threadIdx / blockIdx / blockDim / gridDim,
//   * __syncthreads() for kernels with ONE barrier whose pre-barrier part is idempotent (writeResultKernel,
//     nms_layer.cu:50-113: fills `local[]`, barrier, reads it): every block is run in two phases — in phase 0 each
//     thread returns at the barrier, in phase 1 each thread runs from the top again and falls through it,
//   * emu_launch(grid, block, phases, thread_body).
// Host floating point follows the C++ source (g++ -ffp-contract=off, no -ffast-math): same promotions and evaluation
// order as nvcc applies to the same text; nvcc's default FMA contraction (-fmad=true) is NOT reproduced.
#pragma once
#include <cmath>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
using std::isnan;  // CUDA puts ::isnan in the global namespace
inline float __saturatef(float v) { return v != v ? 0.f : (v < 0.f ? 0.f : (v > 1.f ? 1.f : v)); }  // clamp to [0,1], NaN -> 0 (CUDA math API)
struct float2 { float x, y; };
struct float3 { float x, y, z; };
namespace cuda_emul {
inline thread_local dim3 t_idx, b_idx, b_dim, g_dim;
inline thread_local int phase = 1;
}  // namespace cuda_emul
#define threadIdx cuda_emul::t_idx
#define blockIdx cuda_emul::b_idx
#define blockDim cuda_emul::b_dim
#define gridDim cuda_emul::g_dim
#define __global__
#define __device__
#define __host__
#define __shared__ static thread_local
#define __syncthreads()                 \
  do {                                  \
    if (cuda_emul::phase == 0) return;  \
  } while (0)

namespace cuda_emul {
template <typename F>
void emu_launch(dim3 grid, dim3 block, int phases, F body) {
  const long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel for schedule(static)
  for (long b = 0; b < nblocks; ++b) {
    g_dim = grid;
    b_dim = block;
    b_idx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
    for (int ph = (phases == 2 ? 0 : 1); ph <= 1; ++ph) {
      phase = ph;
      for (unsigned tz = 0; tz < block.z; ++tz)
        for (unsigned ty = 0; ty < block.y; ++ty)
          for (unsigned tx = 0; tx < block.x; ++tx) {
            t_idx = dim3(tx, ty, tz);
            body();
          }
    }
  }
}
}  // namespace cuda_emul
