// Sustained MFMA rate on REAL (random) operands vs zeros, by instruction shape: every wave of a full-chip launch issues a dependent-free stream
// of MFMAs over 4 x 4 operand / accumulator combinations (operands change with every instruction, as in a GEMM inner loop).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_rate_probe tools/mfma_rate_probe.hip && tools/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  uint4 ra[4], rb[4];
  for (int i = 0; i < 4; ++i) { ra[i] = src[(i * 64 + lane) % 4096 + blockIdx.x % 7]; rb[i] = src[((i + 4) * 64 + lane) % 4096 + blockIdx.x % 5]; }
  if constexpr (SHAPE == 0) {  // 32x32x16 f16
    f16v acc[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[(i + j) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ra[i]), __builtin_bit_cast(half8, rb[j]), acc[(i + j) & 3], 0, 0, 0);
    float s = 0; for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
  } else if constexpr (SHAPE == 1) {  // 16x16x32 f16
    f4v acc[8] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[(i * 4 + j) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ra[i]), __builtin_bit_cast(half8, rb[j]), acc[(i * 4 + j) & 7], 0, 0, 0);
    float s = 0; for (int i = 0; i < 8; ++i) for (int q = 0; q < 4; ++q) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
  } else {  // 32x32x64 fp8 (scale variant, unit scales)
    f16v acc[4] = {};
    i8v a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = {(int)ra[2*i].x,(int)ra[2*i].y,(int)ra[2*i].z,(int)ra[2*i].w,(int)ra[2*i+1].x,(int)ra[2*i+1].y,(int)ra[2*i+1].z,(int)ra[2*i+1].w};
                                  b[i] = {(int)rb[2*i].x,(int)rb[2*i].y,(int)rb[2*i].z,(int)rb[2*i].w,(int)rb[2*i+1].x,(int)rb[2*i+1].y,(int)rb[2*i+1].z,(int)rb[2*i+1].w}; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i * 2 + j], 0, 0, 0, 127, 0, 127);
    float s = 0; for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
  }
}
template <int SHAPE> static void run(const char* name, const uint4* d, float* out, double flop_per_mfma, int mfma_per_iter) {
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<SHAPE>, dim3(256 * 2), dim3(256), 0, 0, d, out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(256), 0, 0, d, out, iters);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = 256.0 * 4 * iters * mfma_per_iter;
  printf("  %-22s %7.1f TFLOP/s   %5.1f ns per MFMA and SIMD\n", name, n * flop_per_mfma / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * (double)mfma_per_iter));
}
int main() {
  std::vector<unsigned short> h(4200 * 8);
  uint4* d; float* out;
  (void)hipMalloc((void**)&d, h.size() * 2); (void)hipMalloc((void**)&out, 64);
  for (int mode = 0; mode < 3; ++mode) {
    srand(1);
    for (auto& v : h) {
      if (mode == 0) v = 0;
      else if (mode == 1) { float f = ((rand() % 2001) - 1000) / 500.f; _Float16 x = (_Float16)f; __builtin_memcpy(&v, &x, 2); }          // dense random fp16 in [-2, 2]
      else { float f = (rand() & 1) ? 0.f : (rand() % 2001) / 500.f; _Float16 x = (_Float16)f; __builtin_memcpy(&v, &x, 2); }               // post-ReLU like: half zeros, non-negative
    }
    (void)hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("%s\n", mode == 0 ? "zero operands" : mode == 1 ? "dense random fp16 operands" : "half-zero non-negative operands (post-ReLU like)");
    run<0>("32x32x16 f16", d, out, 2.0 * 32 * 32 * 16, 16);
    run<1>("16x16x32 f16", d, out, 2.0 * 16 * 16 * 32, 16);
    run<2>("32x32x64 f8f6f4 (fp8)", d, out, 2.0 * 32 * 32 * 64, 4);
  }
  return 0;
}
