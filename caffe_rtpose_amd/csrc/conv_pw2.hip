// conv_pw2.hip — two chained 1x1 convolutions in ONE launch:  Y = W2 * relu(W1 * X + b1) + b2
//
// Replaces two cudnnConvolutionForward + cudnnAddTensor (+ ReLUForward) calls of the reference per branch
// (cudnn_conv_layer.cu:21-37) for the branch tails of the linevec net: Mconv6_stageK (128 -> 128, ReLU) ->
// Mconv7_stageK (128 -> 38 / 19) and conv5_4_CPM (128 -> 512, ReLU) -> conv5_5_CPM (512 -> 38 / 19)
// (model/coco/pose_deploy_linevec.prototxt:694-730, 1050-1110, ...).  They hold 0.3 % of the FLOPs but were a tenth
// of the conv stack's time: four launch boundaries and two round trips through HBM per stage for GEMMs of 0.27 GFLOP.
//
// One workgroup (4 waves) owns BM = 64 pixels (flat padded pixel index, as in conv_igemm.hip) of one branch:
//   X tile (64 x 128 ch) -> LDS once;  for every 128-channel chunk c of the middle layer:
//     W1[c] (128 x 128) -> LDS, GEMM1 on MFMA 32x32x16 (wave w owns middle channels [32w, 32w+32)), + b1, ReLU,
//     H_c -> LDS as fp16 (and to the middle layer's own blob tensor, so that every blob can still be tapped),
//     W2[:, c] (64 x 128) -> LDS, GEMM2 accumulates into the wave's 32 x 32 output tile;
//   epilogue = conv_epilogue (bias, fp16 NHWC destinations incl. concat slices, fp32 planar low-res maps).
// Weight blocks are fetched into registers while the previous block is being multiplied (global_load early, ds_write
// after the barrier), so the L2 latency of each block hides behind the previous GEMM.
// Split precision (RTP_PREC_MIXED / F16X3): X, H, W1, W2 each optionally carry a lo part; the passes
// a_hi*W_hi + a_lo*W_hi + a_hi*W_lo accumulate into the same fp32 accumulators (see ConvParams::nchunk).
#include "conv_common.h"
#include <atomic>
#include <type_traits>

namespace rtp {

namespace {
constexpr int PW_BM = 64;
constexpr int PW_ROW = 256;           // 128 fp16 channels
constexpr int PW_STR = PW_ROW + 16;   // padded LDS row: the 16 rows of a ds_read_b128 lane group hit 16 distinct bank quads
constexpr int PW_X = PW_BM * PW_STR;  // one 64-row operand image (X_hi, X_lo, H_hi, H_lo)
constexpr int PW_W = 128 * PW_STR;    // one weight block (W1 chunk: 128 rows; W2 slice: the first 64 rows)
constexpr int PW_LDS = 4 * PW_X + PW_W;  // 104,448 B
static_assert(PW_LDS >= 64 * 64 * 4, "epilogue scratch fits");
}  // namespace

__global__ __launch_bounds__(256) void conv_pw2_kernel(Pw2Params Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sXh = smem;
  unsigned char* sXl = smem + PW_X;
  unsigned char* sHh = smem + 2 * PW_X;
  unsigned char* sHl = smem + 3 * PW_X;
  unsigned char* sW = smem + 4 * PW_X;

  const ConvParams& P = Q.P2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int total_m = P.tiles_per_img * P.nimg;
  const int prob = blockIdx.x / total_m;
  const int mt = blockIdx.x % total_m;
  const int img = mt / P.tiles_per_img;
  const int m0 = (mt % P.tiles_per_img) * PW_BM;
  const ConvProblem& pr = P.prob[prob];
  const long pix0 = (long)img * P.img_pix + (long)P.halo * P.Wp + m0;  // first pixel of the tile (flat padded index)
  const int Mtot = P.H * P.Wp;

  // ---- X tile: 64 rows x 256 B (+ lo block) -> LDS ---------------------------------------------------------
  {
    const _Float16* xin = (const _Float16*)Q.x_in[prob] + pix0 * Q.x_cstride;
    for (int v = tid; v < PW_BM * 16; v += 256) {
      const int row = v >> 4, seg = v & 15;
      const uint4 h = *(const uint4*)((const unsigned char*)(xin + (long)row * Q.x_cstride) + seg * 16);
      *(uint4*)(sXh + row * PW_STR + seg * 16) = h;
      if (Q.x_lo_off) {
        const uint4 l = *(const uint4*)((const unsigned char*)(xin + (long)row * Q.x_cstride + Q.x_lo_off) + seg * 16);
        *(uint4*)(sXl + row * PW_STR + seg * 16) = l;
      }
    }
  }
  // weight block fetch: rows x 256 B contiguous in global -> registers (8 uint4 per thread for 128 rows)
  uint4 wreg[8];
  auto fetch = [&](const unsigned char* src, int rows) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = tid + i * 256;
      if (v < rows * 16) wreg[i] = *(const uint4*)(src + (size_t)v * 16);
    }
  };
  auto stash = [&](int rows) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = tid + i * 256;
      if (v < rows * 16) *(uint4*)(sW + (v >> 4) * PW_STR + (v & 15) * 16) = wreg[i];
    }
  };
  // one GEMM pass over K = 128 of NT 32-row tiles: acc[i] += A(rows a_row0 + i*32 + lrow of a_img) x B(rows b_row0 + lrow of sW)
  auto gemm = [&](const unsigned char* a_img, int a_row0, auto ntm_tag, int b_row0, floatx16* acc) {
    constexpr int NT = decltype(ntm_tag)::value;
    const unsigned char* pb = sW + (b_row0 + lrow) * PW_STR + lhalf * 16;
    const unsigned char* pa = a_img + (a_row0 + lrow) * PW_STR + lhalf * 16;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const uint4 fb = *(const uint4*)(pb + g * 32);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const uint4 fa = *(const uint4*)(pa + i * 32 * PW_STR + g * 32);
        Mma<_Float16>::run(fa, fb, acc[i]);
      }
    }
  };
  const std::integral_constant<int, 2> two{};
  const std::integral_constant<int, 1> one{};

  const int w1_parts = Q.split_w1 ? 2 : 1, w2_parts = Q.split_w2 ? 2 : 1;
  const unsigned char* w1 = (const unsigned char*)Q.w1[prob];
  const unsigned char* w2 = (const unsigned char*)pr.w;
  const size_t W1_BLK = 128 * PW_ROW, W2_BLK = 64 * PW_ROW;

  floatx16 acc2;  // wave (w>>1, w&1): rows (w>>1)*32.., output channels (w&1)*32..
#pragma unroll
  for (int q = 0; q < 16; ++q) acc2[q] = 0.f;

  fetch(w1, 128);  // W1_hi of chunk 0
  for (int c = 0; c < Q.c1_chunks; ++c) {
    floatx16 acc1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc1[i][q] = 0.f;
    // ---- GEMM1: H_c = X x W1[c]^T ----
    __syncthreads();              // X tile stored (c == 0) / previous readers of sW and sH are done
    stash(128);
    if (Q.split_w1) fetch(w1 + ((size_t)c * w1_parts + 1) * W1_BLK, 128);
    else fetch(w2 + (size_t)c * w2_parts * W2_BLK, 64);
    __syncthreads();
    gemm(sXh, 0, two, wave * 32, acc1);
    if (Q.x_lo_off) gemm(sXl, 0, two, wave * 32, acc1);
    if (Q.split_w1) {
      __syncthreads();
      stash(128);
      fetch(w2 + (size_t)c * w2_parts * W2_BLK, 64);
      __syncthreads();
      gemm(sXh, 0, two, wave * 32, acc1);
    }
    // ---- + b1, ReLU, H -> LDS (fp16 hi / lo) ----
    {
      const float bias = Q.b1[prob][c * 128 + wave * 32 + lrow];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhalf;
          float v = acc1[i][q] + bias;
          if (Q.relu1) v = v > 0.f ? v : 0.f;
          const _Float16 hi = (_Float16)v;
          *(_Float16*)(sHh + row * PW_STR + (wave * 32 + lrow) * 2) = hi;
          if (Q.h_lo) *(_Float16*)(sHl + row * PW_STR + (wave * 32 + lrow) * 2) = (_Float16)(v - (float)hi);
        }
    }
    __syncthreads();              // H complete; every wave is past its reads of sW
    stash(64);                    // W2_hi[:, c]
    if (Q.split_w2) fetch(w2 + ((size_t)c * w2_parts + 1) * W2_BLK, 64);
    else if (c + 1 < Q.c1_chunks) fetch(w1 + (size_t)(c + 1) * w1_parts * W1_BLK, 128);
    // the middle layer's own blob (interior pixels only; the halo stays zero)
    if (Q.mid[prob].base) {
      const ConvDst& md = Q.mid[prob];
      for (int v = tid; v < PW_BM * 16; v += 256) {
        const int row = v >> 4, seg = v & 15;
        const int m = m0 + row;
        const int xp = m - (m / P.Wp) * P.Wp;
        if (m >= Mtot || xp < P.halo || xp >= P.halo + P.W) continue;
        _Float16* dp = (_Float16*)md.base + (pix0 + row) * md.cstride + md.coff + c * 128 + seg * 8;
        *(uint4*)dp = *(const uint4*)(sHh + row * PW_STR + seg * 16);
        if (md.lo_off) *(uint4*)(dp + md.lo_off) = *(const uint4*)(sHl + row * PW_STR + seg * 16);
      }
    }
    __syncthreads();
    // ---- GEMM2: Y += H_c x W2[:, c]^T ----
    gemm(sHh, (wave >> 1) * 32, one, (wave & 1) * 32, &acc2);
    if (Q.h_lo) gemm(sHl, (wave >> 1) * 32, one, (wave & 1) * 32, &acc2);
    if (Q.split_w2) {
      __syncthreads();
      stash(64);
      if (c + 1 < Q.c1_chunks) fetch(w1 + (size_t)(c + 1) * w1_parts * W1_BLK, 128);
      __syncthreads();
      gemm(sHh, (wave >> 1) * 32, one, (wave & 1) * 32, &acc2);
    }
  }
  __syncthreads();  // LDS is reused by the epilogue
  floatx16 acc[1][1] = {{acc2}};
  conv_epilogue<_Float16, 64, 64, 2, 2, 1, 1, 1>(P, pr, acc, smem, 0, wave, (wave >> 1) * 32, (wave & 1) * 32, lane, img, m0, 0);
}

hipError_t launch_conv_pw2(const Pw2Params& Q, int nprob, int nimg, hipStream_t stream) {
  static std::atomic<unsigned> attr_mask{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_mask.load(std::memory_order_relaxed) & (1u << (dev & 31)))) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_pw2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
    if (e != hipSuccess) return e;
    attr_mask.fetch_or(1u << (dev & 31), std::memory_order_relaxed);
  }
  dim3 grid(Q.P2.tiles_per_img * nimg * nprob);
  hipLaunchKernelGGL(conv_pw2_kernel, grid, dim3(256), PW_LDS, stream, Q);
  return hipGetLastError();
}

}  // namespace rtp
