// Does v_mfma_f32_32x32x16_f16 honour SUBNORMAL fp16 inputs?  (The split-precision mode stores lo = T(v - float(T(v)))
// parts, most of which are fp16 subnormals for |v| < 0.25.)  hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/mfma_denorm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val, float b_val) {
  half8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  floatx16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d;
  hipMalloc(&d, 4);
  const float cases[][2] = {{1.0f, 1.0f}, {3e-6f, 1.0f}, {1.0f, 3e-6f}, {5.96e-8f, 1.0f}, {3e-6f, 2e-5f}, {6.1e-5f, 1.0f}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, cs[0], cs[1]);
    float h = 0;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    const double want = 16.0 * (double)(float)(_Float16)cs[0] * (double)(float)(_Float16)cs[1];
    printf("a=%g b=%g  mfma sum over k=16: %.9g  expected %.9g  %s\n", cs[0], cs[1], h, want, (h == (float)want) ? "OK" : "DIFFERENT");
  }
  return 0;
}
