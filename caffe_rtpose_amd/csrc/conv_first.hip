// conv_first.hip — conv1_1 (3x3, 3 -> 64, ReLU) straight from the fp32 NCHW net input.
//
// Replaces the first cudnnConvolutionForward of the reference's Forward (cudnn_conv_layer.cu:21-37 on
// model/coco/pose_deploy_linevec.prototxt:6-28) and the H2D'd input blob it reads (rtpose.cpp:1131-1133).
// The generic route (aux_kernels.hip pack_input_kernel: image -> 32-channel im2col tensor, then conv_igemm as a 1x1
// layer with K = 32) writes 31 MB and reads it back per 2 frames for a layer whose real traffic is 6 MB in, 62 MB out;
// here the im2col operand never exists in memory.  HBM-bound on the output write: 128 B per pixel.
//
// One workgroup = one output row of one frame.  The three input rows (3 channels, x = -1 .. W, zero outside) go to LDS
// as fp16 [r][x][c] — u8/256 - 0.5 is exact in fp16 — so that the 27 taps of a pixel are 3 runs of 9 consecutive
// halves.  MFMA 32x32x16 f16 with the WEIGHTS as the A operand (M = output channels) and 32 pixels as B (N): lane
// (pixel l & 31, half h = l >> 5) gathers its 8 + 8 taps k = j*16 + h*8 + e, k = (r*3 + s)*3 + c (the K order of the
// im2col route; k >= 27 reads a zero slot), and owns after 2 K-steps x 2 M-tiles 2 x 16 output channels of its pixel.
// The row index -> channel map of the A operand is chosen so that these are two 16-byte runs per tile:
//   tile t, accumulator q of half h  <->  channel t*32 + (q >> 3)*16 + h*8 + (q & 7).
// Bias enters as the initial accumulator.
#include "conv_common.h"

namespace rtp {

namespace {
constexpr int FIRST_ZERO_SLOT = 8;  // halves
}

__global__ __launch_bounds__(256) void conv_first_kernel(FirstParams Q) {
  const KStamp kstamp_(Q.stamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_first[];
  _Float16* tile = (_Float16*)smem_first;
  const int W = Q.g.W, H = Q.g.H, TW = W + 2, TW3 = TW * 3;
  const int n = blockIdx.x / H, y = blockIdx.x - n * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lpx = lane & 31, h = lane >> 5;
  const int zero_at = 3 * TW3;

  // ---- weights (A fragments, prepacked per lane) and bias: requested first, used after the tile is in place ----
  uint4 wa[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) wa[t][j] = Q.wfrag[(t * 2 + j) * 64 + lane];
  floatx16 bz[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const floatx4 b4 = *(const floatx4*)(Q.bias + t * 32 + (u >> 1) * 16 + h * 8 + (u & 1) * 4);
      bz[t][4 * u] = b4[0]; bz[t][4 * u + 1] = b4[1]; bz[t][4 * u + 2] = b4[2]; bz[t][4 * u + 3] = b4[3];
    }

  // ---- image rows y-1 .. y+1 -> LDS, fp16, [r][x + 1][c] ----
  const float* img = Q.in + (long)n * 3 * H * W;
  for (int idx = tid; idx < 9 * W; idx += 256) {
    const int rc = idx / W, x = idx - rc * W;
    const int r = rc / 3, c = rc - r * 3;
    const int yy = y + r - 1;
    const float v = (yy >= 0 && yy < H) ? img[((long)c * H + yy) * W + x] : 0.f;
    tile[(r * TW + x + 1) * 3 + c] = (_Float16)v;
  }
  if (tid < 18) {  // columns x = -1 and x = W
    const int r = tid / 6, rem = tid - r * 6, side = rem / 3, c = rem - side * 3;
    tile[(r * TW + (side ? W + 1 : 0)) * 3 + c] = (_Float16)0.f;
  }
  if (tid < FIRST_ZERO_SLOT) tile[zero_at + tid] = (_Float16)0.f;
  __syncthreads();

  // ---- per-lane tap offsets (halves, relative to 3 * x): k -> (k / 9) * TW3 + k % 9; taps >= 27 read the zero slot ----
  int off[2][8], mul[2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e2 = 0; e2 < 8; ++e2) {
      const int k = j * 16 + h * 8 + e2;
      const bool real = k < 27;
      off[j][e2] = real ? (k / 9) * TW3 + (k % 9) : zero_at;
      mul[j][e2] = real ? 3 : 0;
    }

  _Float16* orow = Q.out + (((long)n * Q.g.Hp + y + Q.g.halo) * Q.g.Wp + Q.g.halo) * Q.Cp;
  for (int x0 = wave * 32; x0 < W; x0 += 128) {
    const int x = x0 + lpx;
    const int xr = x < W ? x : W - 1;
    half8_t b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) b[j][e2] = tile[xr * mul[j][e2] + off[j][e2]];
    floatx16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, wa[t][0]), b[0], bz[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, wa[t][1]), b[1], acc[t], 0, 0, 0);
    }
    if (x < W) {
      _Float16* op = orow + (long)x * Q.Cp;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          half8_t o;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            float v = acc[t][g * 8 + u];
            if (Q.relu) v = v > 0.f ? v : 0.f;
            o[u] = (_Float16)v;
          }
          *(half8_t*)(op + t * 32 + g * 16 + h * 8) = o;
        }
    }
  }
}

hipError_t launch_conv_first(const FirstParams& Q, hipStream_t stream) {
  const size_t lds = ((size_t)3 * (Q.g.W + 2) * 3 + FIRST_ZERO_SLOT) * sizeof(_Float16);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(conv_first_kernel, dim3(Q.g.N * Q.g.H), dim3(256), lds, stream, Q);
  return hipGetLastError();
}

// Host side of the A-operand layout: fragment (tile t, K-step j, lane l) holds for row i = l & 31 the 8 taps
// k = j*16 + (l >> 5)*8 + e of output channel t*32 + channel_of_row(i).
int conv_first_channel_of_row(int i) {
  const int hh = (i >> 2) & 1, q = (i & 3) + 4 * (i >> 3);
  return (q >> 3) * 16 + hh * 8 + (q & 7);
}

}  // namespace rtp
