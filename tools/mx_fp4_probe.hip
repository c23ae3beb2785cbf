// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with A = MX fp4 (e2m1) and B = MX fp6 (e2m3) on gfx950, and of v_cvt_scalef32_pk_fp4_f32:
//  (1) operand packing: lane l holds row/col (l & 31), K elements [32*(l>>5), +32); fp4 as 16 bytes (which nibble is the lower k?),
//      fp6 as a 24-byte bit stream (LSB first?); cbsz = 4 (A fp4), blgp = 2 (B fp6 e2m3)
//  (2) per-lane E8M0 scales and the opsel byte select
//  (3) rounding / saturation / nibble placement of v_cvt_scalef32_pk_fp4_f32
//  (4) issue rate vs the fp8 form
// hipcc --offload-arch=gfx950 -O2 -o /tmp/mx_fp4_probe tools/mx_fp4_probe.hip && /tmp/mx_fp4_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int OPSEL>
__global__ void k_layout(const unsigned char* A /*[64 lanes][16]*/, const unsigned char* B /*[64 lanes][24]*/, const int* SA, const int* SB, float* C) {
  const int l = threadIdx.x;
  intx8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  __builtin_memcpy(&a, A + l * 16, 16);
  __builtin_memcpy(&b, B + l * 24, 24);
  floatx16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 2, OPSEL, SA[l], OPSEL, SB[l]);
  for (int q = 0; q < 16; ++q) C[((q & 3) + 8 * (q >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[q];
}
__global__ void k_cvt(const float* in, const float* scale, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    unsigned r = 0xAAAAAAAAu;
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, in[2 * i], in[2 * i + 1], scale[i], 0);
    out[2 * i] = r;
    unsigned r2 = 0xAAAAAAAAu;
    r2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r2, in[2 * i], in[2 * i + 1], scale[i], 2);
    out[2 * i + 1] = r2;
  }
}
template <int KIND>
__global__ void k_rate(float* out, unsigned long long* cyc, int iters) {
  floatx16 c[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
  intx8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = 0x23452345 + threadIdx.x * 0x01010101 * (i & 1); b8[i] = 0x3c3c3c3c ^ (threadIdx.x * 0x00110011); }
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[j], 0, 0, 0, 120, 0, 120);
      else c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[j], 4, 2, 0, 120, 0, 120);
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static float dec4(int c) { static const float g[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f}; const float v = g[c & 7]; return (c & 8) ? -v : v; }
static float dec6(int c) {  // e2m3: 1 sign, 2 exponent (bias 1), 3 mantissa
  const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? m * 0.125f : std::ldexp(1.f + m / 8.f, e - 1);
  return s ? -v : v;
}
static int enc4(float x) {  // nearest e2m1, ties to even mantissa, saturating
  const float a = std::fabs(x);
  int c;
  if (a <= 0.25f) c = 0; else if (a < 0.75f) c = 1; else if (a <= 1.25f) c = 2; else if (a < 1.75f) c = 3;
  else if (a <= 2.5f) c = 4; else if (a < 3.5f) c = 5; else if (a <= 5.f) c = 6; else c = 7;
  return c | (x < 0 ? 8 : 0);
}

int main() {
  srand(5);
  std::vector<int> a4(32 * 64), b6(32 * 64);
  for (auto& v : a4) v = rand() & 15;
  for (auto& v : b6) v = rand() & 63;
  std::vector<int> ea(64), eb(64);
  for (int l = 0; l < 64; ++l) { ea[l] = 127 + ((l * 7) % 5) - 2; eb[l] = 127 + ((l * 3) % 4) - 1; }
  unsigned char *dA, *dB; int *dSA, *dSB; float* dC;
  hipMalloc(&dA, 64 * 16); hipMalloc(&dB, 64 * 24); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dC, 32 * 32 * 4);
  for (int nib = 0; nib < 2; ++nib)
    for (int bo = 0; bo < 2; ++bo)
      for (int opsel = 0; opsel < 2; ++opsel) {
        std::vector<unsigned char> A(64 * 16, 0), B(64 * 24, 0);
        for (int l = 0; l < 64; ++l) {
          const int row = l & 31, kh = l >> 5;
          for (int i = 0; i < 32; ++i) {
            const int c = a4[row * 64 + kh * 32 + i];
            const int sh = ((i & 1) ^ nib) * 4;       // nib = 0: even element in the low nibble
            A[l * 16 + i / 2] |= (unsigned char)(c << sh);
            const int c6 = b6[row * 64 + kh * 32 + i];
            for (int bit = 0; bit < 6; ++bit) {
              const int pos = i * 6 + (bo == 0 ? bit : 5 - bit);  // bo = 0: LSB-first bit stream
              if ((c6 >> bit) & 1) B[l * 24 + pos / 8] |= (unsigned char)(1 << (pos % 8));
            }
          }
        }
        std::vector<int> SA(64), SB(64);
        for (int l = 0; l < 64; ++l) {
          SA[l] = opsel == 0 ? (ea[l] | 0x11223300) : ((ea[l] << 8) | 0x11220033);
          SB[l] = opsel == 0 ? (eb[l] | 0x44556600) : ((eb[l] << 8) | 0x44550066);
        }
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
        hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
        if (opsel == 0) hipLaunchKernelGGL(k_layout<0>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
        else hipLaunchKernelGGL(k_layout<1>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
        std::vector<float> C(32 * 32);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k)
              s += (double)dec4(a4[i * 64 + k]) * std::ldexp(1.0, ea[(k / 32) * 32 + i] - 127) * (double)dec6(b6[j * 64 + k]) * std::ldexp(1.0, eb[(k / 32) * 32 + j] - 127);
            maxerr = std::fmax(maxerr, std::fabs(s - C[i * 32 + j]));
            maxref = std::fmax(maxref, std::fabs(s));
          }
        printf("layout: fp4 %s nibble first, fp6 %s-first bit stream, opsel %d: max |err| %.3g (max |ref| %.3g) %s\n", nib ? "HIGH" : "LOW", bo ? "MSB" : "LSB", opsel, maxerr, maxref,
               maxerr < 1e-3 * maxref ? "<== MATCH" : "");
      }
  // (3) conversion
  {
    std::vector<float> in, sc;
    const float vals[] = {0.f, 0.2f, 0.25f, 0.26f, 0.5f, 0.74f, 0.75f, 0.76f, 1.f, 1.24f, 1.25f, 1.26f, 1.5f, 1.74f, 1.75f, 1.76f, 2.f, 2.49f, 2.5f, 2.51f, 3.f, 3.49f, 3.5f, 3.51f, 4.f, 4.99f, 5.f, 5.01f, 6.f, 7.f, 100.f, 1e-9f};
    for (float s : {1.f, 0.25f, 8.f})
      for (size_t i = 0; i + 1 < sizeof(vals) / sizeof(float); i += 2) { in.push_back(vals[i] * s); in.push_back(-vals[i + 1] * s); sc.push_back(s); }
    const int n = (int)sc.size();
    float *din, *dsc; unsigned* dout;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dsc, sc.size() * 4); hipMalloc(&dout, n * 8);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, dsc, dout, n);
    std::vector<unsigned> out(2 * n);
    hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
      const int wa = enc4(in[2 * i] / sc[i]), wb = enc4(in[2 * i + 1] / sc[i]);
      const unsigned r = out[2 * i];
      const int ga = r & 15, gb = (r >> 4) & 15;
      if (ga != wa || gb != wb) { if (bad < 8) printf("cvt: a %g b %g scale %g -> %08x (low %x high %x), expected a %x b %x\n", in[2 * i], in[2 * i + 1], sc[i], r, ga, gb, wa, wb); ++bad; }
    }
    printf("cvt_scalef32_pk_fp4_f32 (x / scale, RNE, saturating; a -> low nibble of byte sel): %d of %d pairs differ; sel 0 word %08x, sel 2 word %08x\n", bad, n, out[0 + 2], out[1 + 2]);
  }
  // (4) rate
  {
    float* dout; unsigned long long* dcyc;
    hipMalloc(&dout, 256 * 4); hipMalloc(&dcyc, 8);
    for (int kind = 0; kind < 2; ++kind) {
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(64), 0, 0, dout, dcyc, 2000);
        else hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(64), 0, 0, dout, dcyc, 2000);
        hipDeviceSynchronize();
      }
      unsigned long long c = 0;
      hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
      printf("rate: %s: %.1f cycles per 32x32x64 MFMA (one wave, 4 accumulators)\n", kind == 0 ? "fp8 x fp8" : "fp4 x fp6", (double)c / (2000.0 * 4));
    }
  }
  return 0;
}
