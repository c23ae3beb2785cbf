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

typedef int intx8 __attribute__((ext_vector_type(8)));
// fp8 (OCP e4m3) error-compensation passes of a split-precision layer: 32x32x64 MX-scaled MFMA, 64 cycles per instruction =
// twice the fp16 rate per K.  Operand: lane l holds row (l&31), K bytes [(l>>5)*32, +32) of the 64-byte k-block
// (tools/mx_fp8_probe.hip); sa / sb = E8M0 scales (2^(e-127)) applied by the hardware, so the products land in the same fp32
// accumulators as the fp16 pass.
__device__ __forceinline__ void mma_fp8(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1, floatx16& c, int sa, int sb) {
  intx8 a, b;
  a[0] = (int)a0.x; a[1] = (int)a0.y; a[2] = (int)a0.z; a[3] = (int)a0.w; a[4] = (int)a1.x; a[5] = (int)a1.y; a[6] = (int)a1.z; a[7] = (int)a1.w;
  b[0] = (int)b0.x; b[1] = (int)b0.y; b[2] = (int)b0.z; b[3] = (int)b0.w; b[4] = (int)b1.x; b[5] = (int)b1.y; b[6] = (int)b1.z; b[7] = (int)b1.w;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
}
// the same instruction with other operand formats (cbsz / blgp: 0 = fp8 e4m3, 2 = fp6 e2m3, 4 = fp4 e2m1): experiments
template <int FA, int FB>
__device__ __forceinline__ void mma_q(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1, floatx16& c, int sa, int sb) {
  intx8 a, b;
  a[0] = (int)a0.x; a[1] = (int)a0.y; a[2] = (int)a0.z; a[3] = (int)a0.w; a[4] = (int)a1.x; a[5] = (int)a1.y; a[6] = (int)a1.z; a[7] = (int)a1.w;
  b[0] = (int)b0.x; b[1] = (int)b0.y; b[2] = (int)b0.z; b[3] = (int)b0.w; b[4] = (int)b1.x; b[5] = (int)b1.y; b[6] = (int)b1.z; b[7] = (int)b1.w;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FA, FB, 0, sa, 0, sb);
}
constexpr int Q_LO_EXP = 12, Q_HI_EXP = 2;             // q block: fp8(lo * 2^12), fp8(hi * 2^2)
constexpr int Q_SA_LO = 127 - Q_LO_EXP, Q_SA_HI = 127 - Q_HI_EXP;
// v_cvt_pk_fp8_f32 rounds to nearest even and keeps subnormals, but turns |x| > 464 into NaN: clamp first
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
  unsigned r = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, (int)r, true);
}

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
  int mt;
  if (P.xcdmap == 2) {  // N tiles fastest: the workgroups of one pixel tile sit next to each other on one XCD and share its strips in the L2
    b.ntile = L % ntiles;
    const int rest = L / ntiles;
    mt = rest % total_m;
    b.prob = rest / total_m;
  } else {
    mt = L % total_m;
    const int rest = L / total_m;
    b.ntile = rest % ntiles;
    b.prob = rest / ntiles;
  }
  b.img = mt / P.tiles_per_img;
  b.mtile = mt % P.tiles_per_img;
  return b;
}

// One pixel x 16 channels of finished values (bias and ReLU applied) to ONE destination tensor: the value as T, its rounding error as a
// lo block and / or as fp8 compensation operands where the destination carries them.  `dd` is a by-value copy of the descriptor: the
// epilogues load destination 0's once per thread, in front of their barrier (one batch of scalar loads whose latency hides under the
// accumulator dump), where the per-item walk through the kernel-argument table cost ~8 dependent scalar loads per item
// (~0.7 us per item iteration: 1.6 of the 3.5 us epilogue of the dominant launch, profiles/r06_experiments.txt).
template <typename T>
__device__ __forceinline__ void conv_store_dst(const ConvDst dd, int Cout, long pix, int c0, const float (&v)[16], const T (&out)[16], int diag = 0) {
  constexpr int VEC = 16 / (int)sizeof(T);        // elements per 16-byte store
  const int nvalid = (Cout - c0) < 16 ? (Cout - c0) : 16;
#ifdef RTP_EXPERIMENTS
  if (diag == 3) pix &= 15;   // timing only: every store instruction is issued, the dirty footprint is 16 pixels
#endif
  {
    T* dp = (T*)dd.base + pix * dd.cstride + dd.coff + c0;
    const bool vec_ok = nvalid == 16 && (((size_t)dp) & 15) == 0;
    if (vec_ok) {
#pragma unroll
      for (int u = 0; u < 16 / VEC; ++u) {
        uint4 pk;
        __builtin_memcpy(&pk, &out[u * VEC], 16);
#ifdef RTP_EXPERIMENTS
        if (diag == 4) { typedef unsigned u4v_t __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(__builtin_bit_cast(u4v_t, pk), (u4v_t*)dp + u); } else
#endif
        ((uint4*)dp)[u] = pk;
      }
    } else {   // a partial chunk (the last channels of a 38- / 19-channel branch output) or an unaligned slice: as few stores as the alignment allows
      // (unrolled with constant indices and run-time predicates: a run-time index would put out[] into scratch memory)
      int done = 0;
      if constexpr (sizeof(T) == 2) {
        if ((((size_t)dp) & 3) == 0) {
#pragma unroll
          for (int u = 0; u < 16; u += 2)
            if (u + 2 <= nvalid) { unsigned w; __builtin_memcpy(&w, &out[u], 4); *(unsigned*)(dp + u) = w; }
          done = nvalid & ~1;
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (u >= done && u < nvalid) dp[u] = out[u];
    }
    if (dd.lo_off) {  // split-precision consumer: the rounding error of the stored value, as a second T
      T lo[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) lo[u] = (T)(v[u] - (float)out[u]);
      T* lp = dp + dd.lo_off;
      if (vec_ok && (((size_t)lp) & 15) == 0) {
#pragma unroll
        for (int u = 0; u < 16 / VEC; ++u) {
          uint4 pk;
          __builtin_memcpy(&pk, &lo[u * VEC], 16);
          ((uint4*)lp)[u] = pk;
        }
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (u < nvalid) lp[u] = lo[u];
      }
    }
    if constexpr (sizeof(T) == 2) {
      if (dd.q_off) {  // fp8 compensation operands for a consumer that runs the fp8 passes
        const int ch = dd.coff + c0;
        unsigned char* qrow = (unsigned char*)((T*)dd.base + pix * dd.cstride + dd.q_off);
        float lof[16], hif[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { hif[u] = (float)out[u]; lof[u] = (v[u] - hif[u]) * (float)(1 << Q_LO_EXP); hif[u] *= (float)(1 << Q_HI_EXP); }
        if (nvalid == 16 && (ch & 15) == 0) {
          unsigned char* qb = qrow + (ch >> 6) * 128 + (ch & 63);
          uint4 ql, qh;
          ql.x = pack4_fp8(lof[0], lof[1], lof[2], lof[3]); ql.y = pack4_fp8(lof[4], lof[5], lof[6], lof[7]);
          ql.z = pack4_fp8(lof[8], lof[9], lof[10], lof[11]); ql.w = pack4_fp8(lof[12], lof[13], lof[14], lof[15]);
          qh.x = pack4_fp8(hif[0], hif[1], hif[2], hif[3]); qh.y = pack4_fp8(hif[4], hif[5], hif[6], hif[7]);
          qh.z = pack4_fp8(hif[8], hif[9], hif[10], hif[11]); qh.w = pack4_fp8(hif[12], hif[13], hif[14], hif[15]);
          *(uint4*)qb = ql;
          *(uint4*)(qb + 64) = qh;
        } else if (nvalid == 16 && (ch & 7) == 0 && (ch & 63) <= 48) {   // an 8-aligned slice of a concat tensor (L2 at channel 168): the 16 channels
          unsigned char* qb = qrow + (ch >> 6) * 128 + (ch & 63);       // stay inside one 64-channel q group: 8-byte stores instead of 32 byte stores
          uint2 l0, l1, h0, h1;
          l0.x = pack4_fp8(lof[0], lof[1], lof[2], lof[3]); l0.y = pack4_fp8(lof[4], lof[5], lof[6], lof[7]);
          l1.x = pack4_fp8(lof[8], lof[9], lof[10], lof[11]); l1.y = pack4_fp8(lof[12], lof[13], lof[14], lof[15]);
          h0.x = pack4_fp8(hif[0], hif[1], hif[2], hif[3]); h0.y = pack4_fp8(hif[4], hif[5], hif[6], hif[7]);
          h1.x = pack4_fp8(hif[8], hif[9], hif[10], hif[11]); h1.y = pack4_fp8(hif[12], hif[13], hif[14], hif[15]);
          *(uint2*)qb = l0; *(uint2*)(qb + 8) = l1;
          *(uint2*)(qb + 64) = h0; *(uint2*)(qb + 72) = h1;
        } else {
          int qdone = 0;
          if ((ch & 3) == 0) {   // 4 channels = 4 bytes of one 64-channel q group (a group boundary is a multiple of 4)
#pragma unroll
            for (int u = 0; u < 16; u += 4)
              if (u + 4 <= nvalid) {
                const int cu = ch + u;
                unsigned char* qb = qrow + (cu >> 6) * 128 + (cu & 63);
                *(unsigned*)qb = pack4_fp8(lof[u], lof[u + 1], lof[u + 2], lof[u + 3]);
                *(unsigned*)(qb + 64) = pack4_fp8(hif[u], hif[u + 1], hif[u + 2], hif[u + 3]);
              }
            qdone = nvalid & ~3;
          }
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (u >= qdone && u < nvalid) {
              const int cu = ch + u;
              unsigned char* qb = qrow + (cu >> 6) * 128 + (cu & 63);
              qb[0] = (unsigned char)(pack4_fp8(lof[u], 0.f, 0.f, 0.f) & 0xff);
              qb[64] = (unsigned char)(pack4_fp8(hif[u], 0.f, 0.f, 0.f) & 0xff);
            }
        }
      }
    }
  }
}

// ... to every destination tensor of the problem: destination 0 from the caller's register copy `d0`, any further one (concat slices:
// conv4_4_CPM, the branch tails) through the kernel-argument table.
template <typename T>
__device__ __forceinline__ void conv_store_pixel(const ConvProblem& pr, const ConvDst d0, int ndst, int Cout, long pix, int c0, const float (&v)[16], int diag = 0) {
  T out[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) out[u] = (T)v[u];
  conv_store_dst<T>(d0, Cout, pix, c0, v, out, diag);
  for (int d = 1; d < ndst; ++d) conv_store_dst<T>(pr.dst[d], Cout, pix, c0, v, out, diag);
}

// Position of (row, col) inside the epilogues' fp32 staging image (BN floats per row).  The 16-byte group index of a row is XORed with
// f(row) = ((row >> RS) & 3) | (bit HB of row) << 3:
//   * the dump: one ds_write_b32 puts lanes 0..31 on 32 consecutive columns of one row and lanes 32..63 on the same columns of the row whose
//     bit HB differs (MFMA accumulator layout) — with BN a multiple of 64 both halves would hit the same 32 banks; the flipped bit 3 of the
//     group index sends the second half to the other 32;
//   * the item loop: the 16 lanes of a ds_read_b128 phase read the same 16-byte group of 4 consecutive rows (x 4 chunks): 4-way conflict
//     without the row's low bits in the key, none with them (BN = 64; 2-way remains for BN = 128, where chunks c and c + 4 share banks).
template <int BN, int HB, int RS = 0>
__device__ __forceinline__ int red_key(int row) {
  if constexpr (BN >= 64) return ((row >> RS) & 3) | (((row >> HB) & 1) << 3);
  else return 0;
}
template <int BN, int HB, int RS = 0>
__device__ __forceinline__ int red_pos(int row, int col) { return row * BN + ((((col >> 2) ^ red_key<BN, HB, RS>(row)) << 2) | (col & 3)); }

// Epilogue: every wave dumps its fp32 accumulators into LDS as [kg][row][col] (the staging LDS is
// dead: callers barrier first), then ALL 256 threads own (pixel row, 16-channel chunk) items: sum
// the KSPLIT partials, add bias, ReLU, convert, and store 16 channels with 16-byte vector stores
// to every destination tensor (interior pixels only; the halo stays zero).  One integer division
// per item instead of one per element, no idle waves.
// acc layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// ILV: fragment i of a wave covers the pixels TM*r + i instead of 32*i + r (conv_ring.hip, interleaved fragment rows).
// NT = threads that run the item loop (512 in the wave-specialised ring kernels: the four DMA waves have nothing left to load and take
// half of the items; they call with dump = false, their accumulators are not the tile's).
template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int TM, int TN, bool ILV = false, int NT = 256>
__device__ __forceinline__ void conv_epilogue(const ConvParams& P, const ConvProblem& pr, floatx16 (&acc)[TM][TN], unsigned char* smem,
                                              int kg, int wrem, int wm0, int wn0, int lane, int img, int m0, int n0, bool dump = true) {
  static_assert(KSPLIT * BM * BN * 4 <= 160 * 1024, "partials fit in LDS");
  const int lrow = lane & 31, lhalf = lane >> 5;
  float* red = (float*)smem;
  constexpr int HB = ILV ? (TM == 4 ? 4 : TM == 2 ? 3 : 2) : 2;   // the row bit that lane half 1 sets in the accumulator layout
#ifdef RTP_EXPERIMENTS
  if (P.diag == 2) { if (acc[0][0][0] == 12345.678f) red[threadIdx.x] = acc[TM - 1][TN - 1][15]; return; }
#endif
  if (dump) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int r32 = (q & 3) + 8 * (q >> 2) + 4 * lhalf;
          const int row = ILV ? wm0 + TM * r32 + i : wm0 + i * 32 + r32;
          const int col = wn0 + j * 32 + lrow;
          red[kg * BM * BN + red_pos<BN, HB>(row, col)] = acc[i][j][q];
        }
  }
  constexpr int CHUNKS = BN / 16;                 // 16-channel chunks per pixel row
  constexpr int ITEMS = BM * CHUNKS;
  constexpr int VEC = 16 / (int)sizeof(T);        // elements per 16-byte store
  static_assert(256 % CHUNKS == 0 && NT % 256 == 0, "a thread's items all belong to one 16-channel chunk");
  // every item of a thread is the same 16-channel chunk (item = tid + 256 k, 256 % CHUNKS == 0): its bias is loaded ONCE, before the
  // barrier, so that the global-load latency hides under the accumulator dump instead of opening every iteration of the item loop
  const int chunk = (int)threadIdx.x % CHUNKS;
  const int c0 = n0 + chunk * 16;
  floatx4 bias4[4];
  {
    const floatx4* bsrc = (const floatx4*)(pr.bias + c0);  // bias is padded to CoutP (multiple of 64)
#pragma unroll
    for (int u = 0; u < 4; ++u) bias4[u] = bsrc[u];
  }
  const ConvDst d0 = pr.dst[0];          // destination 0's descriptor, the counts and the flags: scalar loads in ONE batch, before the barrier
  const int ndst = pr.ndst, Cout = pr.Cout, relu = P.relu;
  float* const out_nchw = pr.out_nchw;
  __syncthreads();

  const int Mtot = P.H * P.Wp;
  const long img_pix0 = (long)img * P.img_pix + (long)P.halo * P.Wp;
  for (int item = threadIdx.x; item < ITEMS; item += NT) {
    const int row = item / CHUNKS;
    const int m = m0 + row;
    const int y = m / P.Wp;
    const int xp = m - y * P.Wp;
    if (m >= Mtot || xp < P.halo || xp >= P.halo + P.W || c0 >= Cout) continue;
    float v[16];
    {
      const int key = red_key<BN, HB>(row);
      const float* src = red + row * BN;
#pragma unroll
      for (int u = 0; u < 4; ++u) { const floatx4 t = *(const floatx4*)(src + (((chunk * 4 + u) ^ key) << 2)); v[4 * u] = t[0]; v[4 * u + 1] = t[1]; v[4 * u + 2] = t[2]; v[4 * u + 3] = t[3]; }
#pragma unroll
      for (int k2 = 1; k2 < KSPLIT; ++k2) {
        const float* s2 = red + (k2 * BM + row) * BN;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const floatx4 t = *(const floatx4*)(s2 + (((chunk * 4 + u) ^ key) << 2)); v[4 * u] += t[0]; v[4 * u + 1] += t[1]; v[4 * u + 2] += t[2]; v[4 * u + 3] += t[3]; }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const floatx4 bb = bias4[u];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        float t = v[4 * u + e2] + bb[e2];
        if (relu) t = t > 0.f ? t : 0.f;
        v[4 * u + e2] = t;
      }
    }
    const int nvalid = (Cout - c0) < 16 ? (Cout - c0) : 16;
    const long pix = img_pix0 + m;
#ifdef RTP_EXPERIMENTS
    if (P.diag == 1 && v[3] != 12345.678f) continue;
#endif
    conv_store_pixel<T>(pr, d0, ndst, Cout, pix, c0, v, P.diag);
    if (out_nchw) {
      float* op = out_nchw + (((long)img * pr.out_C + pr.out_coff + c0) * P.H + y) * P.W + (xp - P.halo);
      const long plane = (long)P.H * P.W;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (u < nvalid) op[u * plane] = v[u];
    }
  }
}

// Epilogue of a convolution whose only consumer is a 2x2 / stride-2 MAX pooling layer (pooling_layer.cpp:140-180, even
// resolutions): the tile is TWO image rows x BM/2 pixels (conv_ring.hip, POOL), logical tile row r < BM/2 is pixel x0 + r of
// row 2*pair and BM/2 + r the pixel below it, so every pooled pixel has its four inputs in this tile.  The maximum is taken on
// the fp32 sums (after the KSPLIT reduction): bias add, ReLU and the conversions to T / lo / fp8 are monotone, so
// max-then-convert stores exactly the bytes convert-then-pool stores (the stand-alone pooling kernel keeps the element with
// the largest (hi, lo) pair, which is the element with the largest fp32 value).  Only the POOLED tensor is written.
template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int TM, int TN, int NT = 256>
__device__ __forceinline__ void conv_epilogue_pool(const ConvParams& P, const ConvProblem& pr, floatx16 (&acc)[TM][TN], unsigned char* smem,
                                                   int kg, int wm0, int wn0, int lane, int img, int pair, int x0, int n0, bool dump = true) {
  static_assert(KSPLIT * BM * BN * 4 <= 160 * 1024, "partials fit in LDS");
  const int lrow = lane & 31, lhalf = lane >> 5;
  float* red = (float*)smem;
#ifdef RTP_EXPERIMENTS
  if (P.diag == 2) { if (acc[0][0][0] == 12345.678f) red[threadIdx.x] = acc[TM - 1][TN - 1][15]; return; }
#endif
  if (dump) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = wm0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhalf;
          const int col = wn0 + j * 32 + lrow;
          red[kg * BM * BN + red_pos<BN, 2, 1>(row, col)] = acc[i][j][q];   // (RS = 1: the item loop's lanes walk EVEN or ODD rows)
        }
  }
  constexpr int CHUNKS = BN / 16, HALF = BM / 2, ITEMS = (HALF / 2) * CHUNKS;
  static_assert(256 % CHUNKS == 0 && NT % 256 == 0, "a thread's items all belong to one 16-channel chunk");
  const int chunk = (int)threadIdx.x % CHUNKS;   // the same for every item of this thread: bias loaded once, under the accumulator dump
  const int c0 = n0 + chunk * 16;
  floatx4 bias4[4];
  {
    const floatx4* bsrc = (const floatx4*)(pr.bias + c0);
#pragma unroll
    for (int u = 0; u < 4; ++u) bias4[u] = bsrc[u];
  }
  const ConvDst d0 = pr.dst[0];
  const int ndst = pr.ndst, Cout = pr.Cout, relu = P.relu;
  __syncthreads();
  for (int item = threadIdx.x; item < ITEMS; item += NT) {
    const int k = item / CHUNKS;
    int x = x0 + 2 * k, pr_ = pair;
    if (x >= P.pool_wq) { x -= P.pool_wq; ++pr_; }  // the tile walked past the pitch (at most once, pitch > BM/2): these columns open the next row pair
    if (pr_ >= P.H / 2 || x >= P.W || c0 >= Cout) continue;
    const long out_row = (long)img * P.pool_img_pix + (long)(pr_ + P.pool_halo) * P.pool_Wp + P.pool_halo;
    float v[16];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // (0,0) (0,1) (1,0) (1,1)
      const int row = (e >> 1) * HALF + 2 * k + (e & 1);
      float t[16];
      const int key = red_key<BN, 2, 1>(row);
      const float* src = red + row * BN;
#pragma unroll
      for (int u = 0; u < 4; ++u) { const floatx4 w = *(const floatx4*)(src + (((chunk * 4 + u) ^ key) << 2)); t[4 * u] = w[0]; t[4 * u + 1] = w[1]; t[4 * u + 2] = w[2]; t[4 * u + 3] = w[3]; }
#pragma unroll
      for (int k2 = 1; k2 < KSPLIT; ++k2) {
        const float* s2 = red + (k2 * BM + row) * BN;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const floatx4 w = *(const floatx4*)(s2 + (((chunk * 4 + u) ^ key) << 2)); t[4 * u] += w[0]; t[4 * u + 1] += w[1]; t[4 * u + 2] += w[2]; t[4 * u + 3] += w[3]; }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (e == 0 || t[u] > v[u]) ? t[u] : v[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const floatx4 bb = bias4[u];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        float t = v[4 * u + e2] + bb[e2];
        if (relu) t = t > 0.f ? t : 0.f;
        v[4 * u + e2] = t;
      }
    }
#ifdef RTP_EXPERIMENTS
    if (P.diag == 1 && v[3] != 12345.678f) continue;
#endif
    conv_store_pixel<T>(pr, d0, ndst, Cout, out_row + (x >> 1), c0, v);
  }
}

}  // namespace rtp
