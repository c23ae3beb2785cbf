// conv_common.h — pieces shared by the convolution kernels: MFMA wrappers and the epilogue.
#pragma once
#include "kernels.h"

namespace rtp {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma;
template <> struct Mma<_Float16> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, floatx16& c) {
    half8_t av = __builtin_bit_cast(half8_t, a);
    half8_t bv = __builtin_bit_cast(half8_t, b);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, floatx16& c) {
    floatx4 av = __builtin_bit_cast(floatx4, a);
    floatx4 bv = __builtin_bit_cast(floatx4, b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], c, 0, 0, 0);
  }
};

// XCD-aware decode of a 1-D grid.  The dispatcher places workgroup `lin` on XCD lin % 8 and each
// XCD has a private 4 MiB L2, so the grid is laid out such that one XCD sees a CONTIGUOUS range
// of the logical index L = ((prob * ntiles + ntile) * mtiles_total + mtile): workgroups on one
// XCD share one weight set (prob, ntile) and neighbouring pixel tiles, whose strips overlap.
// (Placement affects speed only; the mapping is a bijection for any grid size.)
struct BlockCoord { int prob, ntile, img, mtile; };
__device__ __forceinline__ BlockCoord decode_block(const ConvParams& P, int ntiles, int total_m /* tiles_per_img*N */) {
  const int total = gridDim.x;
  const int lin = blockIdx.x;
  const int xcd = lin & 7, j = lin >> 3;
  const int q = total >> 3, r = total & 7;
  const int L = P.xcdmap ? ((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j) : lin;
  BlockCoord b;
  const int mt = L % total_m;
  const int rest = L / total_m;
  b.ntile = rest % ntiles;
  b.prob = rest / ntiles;
  b.img = mt / P.tiles_per_img;
  b.mtile = mt % P.tiles_per_img;
  return b;
}

// Split-K reduction across the KSPLIT wave groups of a workgroup (through LDS, which must be
// dead: callers barrier first) followed by bias + ReLU + convert + store of the interior pixels.
// acc layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& P, const ConvProblem& pr, floatx16 (&acc)[TM][TN], unsigned char* smem,
                                              int kg, int wrem, int wm0, int wn0, int lane, int img, int m0, int n0) {
  const int lrow = lane & 31, lhalf = lane >> 5;
  // ---- intra-workgroup split-K reduction (staging LDS is dead after the last barrier) ----
  if constexpr (KSPLIT > 1) {
    float* red = (float*)smem;
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            red[((((kg - 1) * (WM * WN) + wrem) * (TM * TN) + i * TN + j) * 16 + q) * 64 + lane] = acc[i][j][q];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int k2 = 1; k2 < KSPLIT; ++k2)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            acc[i][j][q] += red[((((k2 - 1) * (WM * WN) + wrem) * (TM * TN) + i * TN + j) * 16 + q) * 64 + lane];
  }

  // ---- epilogue: bias, ReLU, convert, store (interior pixels only; the halo stays zero) ----
  const int Mtot = P.H * P.Wp;
  const long img_pix0 = (long)img * P.img_pix + (long)P.halo * P.Wp;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn0 + j * 32 + lrow;
    const bool col_ok = col < pr.Cout;
    const float bias = col_ok ? pr.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = m0 + wm0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhalf;
        const int y = m / P.Wp;
        const int xp = m - y * P.Wp;
        if (col_ok && m < Mtot && xp >= P.halo && xp < P.halo + P.W) {
          float v = acc[i][j][q] + bias;
          if (P.relu) v = v > 0.f ? v : 0.f;
          const long pix = img_pix0 + m;
          for (int d = 0; d < pr.ndst; ++d)
            ((T*)pr.dst[d].base)[pix * pr.dst[d].cstride + pr.dst[d].coff + col] = (T)v;
          if (pr.out_nchw)
            pr.out_nchw[(((long)img * pr.out_C + pr.out_coff + col) * P.H + y) * P.W + (xp - P.halo)] = v;
        }
      }
    }
  }
}

}  // namespace rtp
