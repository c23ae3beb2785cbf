// conv_igemm.hip — implicit-GEMM convolution on CDNA4 MFMA for the linevec conv stack.
//
// Replaces cudnnConvolutionForward + cudnnAddTensor + ReLUForward of the reference
// (src/caffe/layers/cudnn_conv_layer.cu:21-37, relu_layer.cu:9-26): stride-1 cross-correlation,
// pad (k-1)/2, + bias, optional ReLU, for k in {1,3,7}.
//
// Formulation (see kernels.h for the halo'd NHWC layout): the GEMM M axis is the FLAT padded
// pixel index of the interior rows (m in [0, H*Wp)), N axis = output channels, K = taps x Cin.
// Because the halo is materialised as zeros, the A operand of tap (r,s) is the input tensor
// shifted by (r-pad)*Wp + (s-pad) pixels — a plain strided 2-D block.  For one filter ROW r a
// workgroup stages a strip of BM+KS-1 consecutive pixels x one 128-byte channel chunk in LDS
// once and reuses it for all KS taps of that row (tap s just reads the strip s rows lower), so
// activation traffic from L2 is ~1/KS of a naive implicit GEMM; the weight tile of each tap is
// staged per step.  Steps (r, chunk, s) are software-pipelined through registers
// (global_load -> VGPR -> ds_write_b128) with double-buffered LDS and one barrier per step.
// LDS rows are 16 bytes longer than the payload so that the 16 lanes of a ds_read_b128 lane
// group hit 16 distinct bank quads (row stride 144 B = 36 banks; 36*r mod 64 is injective on
// r mod 16).
//
// MFMA: v_mfma_f32_32x32x16_f16 (fp16 path) or 4 x v_mfma_f32_32x32x2_f32 (exact-f32 path) per
// 32-byte k-group; a lane's 16-byte fragment is bytes [(lane>>5)*16, +16) of the k-group for
// row (lane&31) for BOTH operands, so the hardware pairs identical k's of A and B.
// Accumulators are fp32; C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//
// The chip has 256 CUs but the dominant layers are tiny GEMMs (M = 46*88 = 4048, N = 128), so
// the launch pairs the independent L1/L2 branch convolutions (blockIdx.z) and small tiles can
// split the k-groups of a chunk across the 4 waves of a workgroup (KSPLIT) and reduce through
// LDS at the end, instead of shrinking the per-wave tile below 64x64 (LDS-read bound).
#include "conv_common.h"
#include <atomic>

namespace rtp {

template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int ROWB>
struct ConvTraits {
  static constexpr int NT = 256;
  static constexpr int STR = ROWB + 16;
  static constexpr int AROWS = BM + KS - 1;
  static constexpr int A_BYTES = AROWS * STR;
  static constexpr int B_BYTES = BN * STR;
  static constexpr int VPR = ROWB / 16;
  static constexpr int A_VECS = AROWS * VPR;
  static constexpr int B_VECS = BN * VPR;
  static constexpr int A_IT = (A_VECS + NT - 1) / NT;
  static constexpr int B_IT = (B_VECS + NT - 1) / NT;
  static constexpr int G = ROWB / 32;
  static constexpr int TM = BM / WM / 32;
  static constexpr int TN = BN / WN / 32;
  static constexpr int STAGE_BYTES = 2 * A_BYTES + 2 * B_BYTES;
  static constexpr int RED_BYTES = KSPLIT * BM * BN * 4;  // epilogue: all partials as [kg][row][col] fp32
  static constexpr int LDS_BYTES = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
  static_assert(WM * WN * KSPLIT == 4, "4 waves per workgroup");
  static_assert(G % KSPLIT == 0, "k-groups must split evenly");
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile is a multiple of 32x32");
};

template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int ROWB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams P) {
  const KStamp kstamp_(P.stamp);
  using TR = ConvTraits<T, BM, BN, WM, WN, KSPLIT, KS, ROWB>;
  constexpr int NT = TR::NT, STR = TR::STR, VPR = TR::VPR;
  constexpr int A_IT = TR::A_IT, B_IT = TR::B_IT, G = TR::G, TM = TR::TM, TN = TR::TN;
  constexpr int PAD = KS / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                     // 2 strips
  unsigned char* sB = smem + 2 * TR::A_BYTES;   // 2 weight tiles

  const BlockCoord bc = decode_block(P, P.CoutP / BN, P.tiles_per_img * P.nimg);
  const ConvProblem& pr = P.prob[bc.prob];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kg = wave / (WM * WN);
  const int wrem = wave % (WM * WN);
  const int wm0 = (wrem / WN) * (BM / WM);
  const int wn0 = (wrem % WN) * (BN / WN);
  const int lrow = lane & 31, lhalf = lane >> 5;

  const int img = bc.img;
  const int m0 = bc.mtile * BM;
  const int n0 = bc.ntile * BN;
  const int nchunk = P.nchunk;
  const long pix_bytes = (long)P.in_cstride * (long)sizeof(T);
  // byte address of the strip origin for filter row 0 (pixel m0 shifted by (-PAD rows, -PAD cols))
  const unsigned char* in_base = (const unsigned char*)pr.in +
      ((long)img * P.img_pix + (long)P.halo * P.Wp + m0 - (long)PAD * P.Wp - PAD) * pix_bytes;
  const unsigned char* w_base = (const unsigned char*)pr.w + (long)n0 * ROWB;
  const long w_tap_stride = (long)nchunk * P.CoutP * ROWB;  // bytes per tap
  const long w_chunk_stride = (long)P.CoutP * ROWB;

  // per-thread staging slots
  int a_off_g[A_IT], a_off_l[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int v = tid + i * NT;
    const int row = v / VPR, seg = v % VPR;
    a_ok[i] = v < TR::A_VECS;
    a_off_l[i] = row * STR + seg * 16;
    a_off_g[i] = row * (int)pix_bytes + seg * 16;
  }
  int b_off_g[B_IT], b_off_l[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int v = tid + i * NT;
    const int row = v / VPR, seg = v % VPR;
    b_ok[i] = v < TR::B_VECS;
    b_off_l[i] = row * STR + seg * 16;
    b_off_g[i] = row * ROWB + seg * 16;
  }

  uint4 ra[A_IT], rb[B_IT];
  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // step = (r, chunk, s); the strip depends on (r, chunk) only
  auto load_a = [&](int r, int chunk) {
    const int pchunk = (P.wrap_at && chunk >= P.wrap_at) ? chunk - P.wrap_at : chunk;  // split precision: see ConvParams::nchunk
    const unsigned char* p = in_base + (long)r * P.Wp * pix_bytes + (long)pchunk * ROWB;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (a_ok[i]) ra[i] = *(const uint4*)(p + a_off_g[i]);
  };
  auto load_b = [&](int r, int chunk, int s) {
    const unsigned char* p = w_base + (long)(r * KS + s) * w_tap_stride + (long)chunk * w_chunk_stride;
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if (b_ok[i]) rb[i] = *(const uint4*)(p + b_off_g[i]);
  };
  auto store_a = [&](int buf) {
    unsigned char* p = sA + buf * TR::A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (a_ok[i]) *(uint4*)(p + a_off_l[i]) = ra[i];
  };
  auto store_b = [&](int buf) {
    unsigned char* p = sB + buf * TR::B_BYTES;
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if (b_ok[i]) *(uint4*)(p + b_off_l[i]) = rb[i];
  };

  load_a(0, 0);
  load_b(0, 0, 0);
  store_a(0);
  store_b(0);
  __syncthreads();

  int abuf = 0, bbuf = 0;
  int r = 0, chunk = 0, s = 0;
  const int nsteps = KS * nchunk * KS;
  for (int step = 0; step < nsteps; ++step) {
    // next step coordinates
    int ns = s + 1, nc = chunk, nr = r;
    bool new_strip = false;
    if (ns == KS) {
      ns = 0;
      new_strip = true;
      nc = chunk + 1;
      if (nc == nchunk) { nc = 0; nr = r + 1; }
    }
    const bool has_next = (step + 1) < nsteps;
    if (has_next) {
      load_b(nr, nc, ns);
      if (new_strip) load_a(nr, nc);
    }
    // ---- MFMA on the current step ----
    {
      const unsigned char* pa = sA + abuf * TR::A_BYTES + (wm0 + lrow + s) * STR + lhalf * 16;
      const unsigned char* pb = sB + bbuf * TR::B_BYTES + (wn0 + lrow) * STR + lhalf * 16;
#pragma unroll
      for (int gi = 0; gi < G / KSPLIT; ++gi) {
        const int g = gi * KSPLIT + kg;
        uint4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(pa + i * 32 * STR + g * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *(const uint4*)(pb + j * 32 * STR + g * 32);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
      }
    }
    if (has_next) {
      store_b(bbuf ^ 1);
      if (new_strip) store_a(abuf ^ 1);
    }
    __syncthreads();
    bbuf ^= 1;
    if (new_strip) abuf ^= 1;
    s = ns; chunk = nc; r = nr;
  }

  conv_epilogue<T, BM, BN, WM, WN, KSPLIT, TM, TN>(P, pr, acc, smem, kg, wrem, wm0, wn0, lane, img, m0, n0);
}

template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int ROWB>
static hipError_t launch_one(const ConvParams& P, int nprob, int N, hipStream_t stream) {
  using TR = ConvTraits<T, BM, BN, WM, WN, KSPLIT, KS, ROWB>;
  auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, KSPLIT, KS, ROWB>;
  // the >64 KiB dynamic-LDS opt-in is per device: remember which devices have it
  static std::atomic<unsigned> attr_mask{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_mask.load(std::memory_order_relaxed) & (1u << (dev & 31)))) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TR::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_mask.fetch_or(1u << (dev & 31), std::memory_order_relaxed);
  }
  dim3 grid(P.tiles_per_img * N * (P.CoutP / BN) * nprob);
  hipLaunchKernelGGL(kern, grid, dim3(256), TR::LDS_BYTES, stream, P);
  return hipGetLastError();
}

template <typename T, int KS, int ROWB>
static hipError_t launch_cfg(int cfg, const ConvParams& P, int nprob, int N, hipStream_t stream) {
  switch (cfg) {
    case CFG_128x128: return launch_one<T, 128, 128, 2, 2, 1, KS, ROWB>(P, nprob, N, stream);
    case CFG_64x128: return launch_one<T, 64, 128, 1, 2, 2, KS, ROWB>(P, nprob, N, stream);
    case CFG_64x64:
      return launch_one<T, 64, 64, 1, 1, 4, KS, ROWB>(P, nprob, N, stream);
    case CFG_128x64: return launch_one<T, 128, 64, 2, 1, 2, KS, ROWB>(P, nprob, N, stream);
    default: return hipErrorInvalidValue;
  }
}

template <typename T>
static hipError_t launch_ks(int cfg, int ks, int rowb, const ConvParams& P, int nprob, int N, hipStream_t stream) {
  if (rowb == 128) {
    if (ks == 1) return launch_cfg<T, 1, 128>(cfg, P, nprob, N, stream);
    if (ks == 3) return launch_cfg<T, 3, 128>(cfg, P, nprob, N, stream);
    if (ks == 7) return launch_cfg<T, 7, 128>(cfg, P, nprob, N, stream);
  } else if (rowb == 64 && ks == 1 && cfg == CFG_128x64) {
    // only the im2col-packed first layer (32 fp16 channels = 64 bytes per pixel)
    return launch_one<T, 128, 64, 2, 1, 2, 1, 64>(P, nprob, N, stream);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_conv(int prec, int cfg, int ks, int rowb, const ConvParams& P, int nprob, int N,
                       hipStream_t stream) {
  if (prec == 0) return launch_ks<_Float16>(cfg, ks, rowb, P, nprob, N, stream);
  return launch_ks<float>(cfg, ks, rowb, P, nprob, N, stream);
}

}  // namespace rtp
