// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 for the fp8 error-compensation passes:
//  (1) operand layout: lane l holds A[row = l&31][k = (l>>5)*32 .. +32) / B[k][col = l&31] as 32 consecutive fp8 bytes?
//  (2) E8M0 scale operands: 2^(e-127) per lane, opsel 0 = byte 0
//  (3) v_cvt_pk_fp8_f32 rounding / saturation / subnormals
//  (4) issue rate vs v_mfma_f32_32x32x16_f16
// hipcc --offload-arch=gfx950 -O2 -o tools/mx_fp8_probe.bin tools/mx_fp8_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

__global__ void k_layout(const unsigned char* A /*[32][64]*/, const unsigned char* B /*[64][32] as [col][k]*/, float* C, int ea, int eb) {
  const int l = threadIdx.x;
  intx8 a, b;
  __builtin_memcpy(&a, A + (l & 31) * 64 + (l >> 5) * 32, 32);
  __builtin_memcpy(&b, B + (l & 31) * 64 + (l >> 5) * 32, 32);
  floatx16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, ea, 0, eb);
  for (int q = 0; q < 16; ++q) C[((q & 3) + 8 * (q >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[q];
}
__global__ void k_cvt(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], 0, false);
}
template <int KIND>
__global__ void k_rate(float* out, unsigned long long* cyc, int iters) {
  floatx16 c[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
  intx8 a8, b8;
  half8_t ah, bh;
  for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i & 1); b8[i] = 0x3c3c3c3c; ah[i] = (_Float16)(1.0f + 0.01f * threadIdx.x); bh[i] = (_Float16)0.5f; }
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[j], 0, 0, 0);
      else if (KIND == 1) c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[j], 0, 0, 0, 115, 0, 120);
      else c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[j], 0, 0, 0, 0, 0, 0);
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static float fp8_e4m3_to_float(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x;
  if (e == 0) x = std::ldexp((float)m, -9);
  else if (e == 15 && m == 7) x = NAN;
  else x = std::ldexp(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main() {
  // (1)+(2)
  std::vector<unsigned char> A(32 * 64), B(32 * 64);
  srand(3);
  for (auto& v : A) v = (unsigned char)(rand() & 0x7f) % 0x58 | ((rand() & 1) << 7);  // finite e4m3 values of both signs, |x| < 16
  for (auto& v : B) v = (unsigned char)(rand() & 0x7f) % 0x58 | ((rand() & 1) << 7);
  unsigned char *dA, *dB;
  float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int trial = 0; trial < 3; ++trial) {
    const int ea = trial == 0 ? 127 : (trial == 1 ? 124 : 115), eb = trial == 2 ? 121 : 127;
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dC, ea, eb);
    std::vector<float> C(32 * 32);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int k = 0; k < 64; ++k) ref += (double)fp8_e4m3_to_float(A[i * 64 + k]) * fp8_e4m3_to_float(B[j * 64 + k]);
        ref *= std::ldexp(1.0, ea - 127) * std::ldexp(1.0, eb - 127);
        maxerr = std::fmax(maxerr, std::fabs(ref - C[i * 32 + j]));
        maxref = std::fmax(maxref, std::fabs(ref));
      }
    printf("layout/scale trial %d (scale_a 2^%d, scale_b 2^%d): max |err| %.3e of max |ref| %.3e  %s\n", trial, ea - 127, eb - 127, maxerr, maxref,
           maxerr <= 1e-5 * maxref ? "OK (assumed layout is right)" : "MISMATCH");
  }
  // (3)
  const float vals[16] = {1.0f, -1.0f, 0.0625f, 1.0625f, 1.1875f, 447.f, 448.f, 460.f, 500.f, 1e6f, 0.001953125f, 0.0009765625f, 0.0029f, -0.3f, 17.5f, 3e-4f};
  float* dv; unsigned* du;
  hipMalloc(&dv, sizeof vals); hipMalloc(&du, 8 * 4);
  hipMemcpy(dv, vals, sizeof vals, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dv, du, 8);
  unsigned u[8];
  hipMemcpy(u, du, sizeof u, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) {
    const unsigned char b = (u[i / 2] >> (8 * (i & 1))) & 0xff;
    printf("cvt_pk_fp8_f32(%g) = 0x%02x = %g\n", vals[i], b, fp8_e4m3_to_float(b));
  }
  // (4)
  float* dout; unsigned long long* dcyc;
  hipMalloc(&dout, 256 * 4 * 1024 * 4); hipMalloc(&dcyc, 8);
  const int iters = 2000;
  for (int kind = 0; kind < 3; ++kind)
    for (int blocks : {1, 1024}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, dout, dcyc, iters);
        else if (kind == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, dout, dcyc, iters);
        else hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, dout, dcyc, iters);
      };
      launch();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long cyc; hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
      const double flop = (kind == 0 ? 32768.0 : 131072.0) * 4 * iters * 4 /*waves*/ * blocks;
      printf("%s, %4d workgroups x 4 waves: %.1f cycles per MFMA per wave, %.0f TFLOP/s\n", kind == 0 ? "f16 32x32x16          " : kind == 1 ? "fp8 32x32x64 scaled   " : "fp8 32x32x64 scale=0  ",
             blocks, (double)cyc / (4.0 * iters), flop / (ms * 1e-3) / 1e12);
    }
  return 0;
}
