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
constexpr int PW_MAXMID = 512;         // middle channels whose bias is staged in LDS
constexpr int PW_LDS = 4 * PW_X + PW_W + PW_MAXMID * 4;  // 106,496 B
static_assert(PW_LDS >= 64 * 64 * 4, "epilogue scratch fits");
}  // namespace

// weight blocks in flight live in SSA vectors (an array of uint4 ends up in scratch: the unrolled indices are not constant
// yet when allocas are promoted)
typedef unsigned int u32x32 __attribute__((ext_vector_type(32)));
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
template <typename V>
__device__ __forceinline__ void pw_fetch(V& reg, const unsigned char* src, int tid) {
  constexpr int NV = sizeof(V) / 16;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const uint4 t = *(const uint4*)(src + (size_t)(tid + i * 256) * 16);
    reg[4 * i] = t.x; reg[4 * i + 1] = t.y; reg[4 * i + 2] = t.z; reg[4 * i + 3] = t.w;
  }
}
template <typename V>
__device__ __forceinline__ void pw_stash(const V& reg, unsigned char* sW, int tid) {
  constexpr int NV = sizeof(V) / 16;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * 256;
    uint4 t;
    t.x = reg[4 * i]; t.y = reg[4 * i + 1]; t.z = reg[4 * i + 2]; t.w = reg[4 * i + 3];
    *(uint4*)(sW + (v >> 4) * PW_STR + (v & 15) * 16) = t;
  }
}

__global__ __launch_bounds__(256) void conv_pw2_kernel(Pw2Params Q) {
  const KStamp kstamp_(Q.P2.stamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sXh = smem;
  unsigned char* sXl = smem + PW_X;
  unsigned char* sHh = smem + 2 * PW_X;
  unsigned char* sHl = smem + 3 * PW_X;
  unsigned char* sW = smem + 4 * PW_X;
  float* sB1 = (float*)(smem + 4 * PW_X + PW_W);

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
  // diagnostics (RTP_PW2_PROBE): 100 MHz wall-clock stamps of workgroup 0 at the phase boundaries
  int stamp_i = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (P.clkprobe && blockIdx.x == 0 && tid == 0 && stamp_i < 31) P.clkprobe[1 + stamp_i++] = wall_clock64();
  };
  stamp();

  // requests in the order they are needed: middle bias, X tile, weight blocks (vmcnt waits are in order)
  // two-entry kernarg arrays are selected, not indexed: a dynamic index turns into a dependent global load of the pointer
  const float* b1p = prob ? Q.b1[1] : Q.b1[0];
  const int nmid = Q.c1_chunks * 128;
  const float b_a = tid < nmid ? b1p[tid] : 0.f;
  const float b_b = tid + 256 < nmid ? b1p[tid + 256] : 0.f;
  // ---- X tile: 64 rows x 256 B (+ lo block): all requests first (4 + 4 uint4 per thread), LDS stores after the weight requests
  u32x16 r_xh, r_xl;
  {
    const _Float16* xin = (const _Float16*)(prob ? Q.x_in[1] : Q.x_in[0]) + pix0 * Q.x_cstride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + i * 256, row = v >> 4, seg = v & 15;
      const unsigned char* src = (const unsigned char*)(xin + (long)row * Q.x_cstride) + seg * 16;
      const uint4 h = *(const uint4*)src;
      r_xh[4 * i] = h.x; r_xh[4 * i + 1] = h.y; r_xh[4 * i + 2] = h.z; r_xh[4 * i + 3] = h.w;
      if (Q.x_lo_off) {
        const uint4 l = *(const uint4*)(src + (size_t)Q.x_lo_off * 2);
        r_xl[4 * i] = l.x; r_xl[4 * i + 1] = l.y; r_xl[4 * i + 2] = l.z; r_xl[4 * i + 3] = l.w;
      }
    }
  }
  // Weight blocks travel global -> registers -> LDS.  Every block of the current 128-channel chunk of the middle layer
  // has its OWN registers (W1 hi / lo: 8 uint4 each, W2 hi / lo: 4 each = 96 VGPRs; one wave per SIMD, registers are free)
  // and is requested a whole chunk ahead: at kernel start, then again as soon as its registers have been stashed.  The L2 /
  // HBM latency (2.5-3 us per block when waited for, RTP_PW2_PROBE) is paid once, under the X tile load, instead of per block.
  // Barriers are LDS-only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would wait for vmcnt(0), i.e. for every
  // prefetch in flight.
  u32x32 r_w1h, r_w1l;
  u32x16 r_w2h, r_w2l;
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // one GEMM pass over K = 128 of NT 32-row tiles: acc[i] += A(rows a_row0 + i*32 + lrow of a_img) x B(rows b_row0 + lrow of sW)
  auto gemm = [&](const unsigned char* a_img, int a_row0, auto ntm_tag, int b_row0, floatx16* acc) __attribute__((always_inline)) {
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
  const unsigned char* w1 = (const unsigned char*)(prob ? Q.w1[1] : Q.w1[0]);
  const unsigned char* w2 = (const unsigned char*)(prob ? P.prob[1].w : P.prob[0].w);
  const size_t W1_BLK = 128 * PW_ROW, W2_BLK = 64 * PW_ROW;
  auto w1_blk = [&](int c, int part) { return w1 + ((size_t)c * w1_parts + part) * W1_BLK; };
  auto w2_blk = [&](int c, int part) { return w2 + ((size_t)c * w2_parts + part) * W2_BLK; };

  floatx16 acc2;  // wave (w>>1, w&1): rows (w>>1)*32.., output channels (w&1)*32..
#pragma unroll
  for (int q = 0; q < 16; ++q) acc2[q] = 0.f;

  pw_fetch(r_w1h, w1_blk(0, 0), tid);
  if (Q.split_w1) pw_fetch(r_w1l, w1_blk(0, 1), tid);
  pw_fetch(r_w2h, w2_blk(0, 0), tid);
  if (Q.split_w2) pw_fetch(r_w2l, w2_blk(0, 1), tid);
  sB1[tid] = b_a;
  sB1[tid + 256] = b_b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = tid + i * 256, row = v >> 4, seg = v & 15;
    uint4 t;
    t.x = r_xh[4 * i]; t.y = r_xh[4 * i + 1]; t.z = r_xh[4 * i + 2]; t.w = r_xh[4 * i + 3];
    *(uint4*)(sXh + row * PW_STR + seg * 16) = t;
    if (Q.x_lo_off) {
      t.x = r_xl[4 * i]; t.y = r_xl[4 * i + 1]; t.z = r_xl[4 * i + 2]; t.w = r_xl[4 * i + 3];
      *(uint4*)(sXl + row * PW_STR + seg * 16) = t;
    }
  }
  stamp();
  for (int c = 0; c < Q.c1_chunks; ++c) {
    const bool more = c + 1 < Q.c1_chunks;
    floatx16 acc1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc1[i][q] = 0.f;
    // ---- GEMM1: H_c = X x W1[c]^T ----
    lds_barrier();                // X tile stored (c == 0) / previous readers of sW and sH are done
    pw_stash(r_w1h, sW, tid);
    if (more) pw_fetch(r_w1h, w1_blk(c + 1, 0), tid);
    lds_barrier();
    stamp();
    gemm(sXh, 0, two, wave * 32, acc1);
    if (Q.x_lo_off) gemm(sXl, 0, two, wave * 32, acc1);
    stamp();
    if (Q.split_w1) {
      lds_barrier();
      pw_stash(r_w1l, sW, tid);
      if (more) pw_fetch(r_w1l, w1_blk(c + 1, 1), tid);
      lds_barrier();
      gemm(sXh, 0, two, wave * 32, acc1);
      stamp();
    }
    // ---- + b1, ReLU, H -> LDS (fp16 hi / lo) ----
    {
      const float bias = sB1[c * 128 + wave * 32 + lrow];
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
    stamp();
    lds_barrier();                // H complete; every wave is past its reads of sW
    pw_stash(r_w2h, sW, tid);                 // W2_hi[:, c]
    if (more) pw_fetch(r_w2h, w2_blk(c + 1, 0), tid);
    lds_barrier();
    stamp();
    // ---- GEMM2: Y += H_c x W2[:, c]^T ----
    gemm(sHh, (wave >> 1) * 32, one, (wave & 1) * 32, &acc2);
    if (Q.h_lo) gemm(sHl, (wave >> 1) * 32, one, (wave & 1) * 32, &acc2);
    stamp();
    // the middle layer's own blob (interior pixels only; the halo stays zero): plain stores, nobody waits for them
    const ConvDst& md = prob ? Q.mid[1] : Q.mid[0];
    if (md.base) {
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
    stamp();
    if (Q.split_w2) {
      lds_barrier();
      pw_stash(r_w2l, sW, tid);
      if (more) pw_fetch(r_w2l, w2_blk(c + 1, 1), tid);
      lds_barrier();
      gemm(sHh, (wave >> 1) * 32, one, (wave & 1) * 32, &acc2);
      stamp();
    }
  }
  __syncthreads();  // LDS is reused by the epilogue
  stamp();
  floatx16 acc[1][1] = {{acc2}};
  conv_epilogue<_Float16, 64, 64, 2, 2, 1, 1, 1>(P, pr, acc, smem, 0, wave, (wave >> 1) * 32, (wave & 1) * 32, lane, img, m0, 0);
  stamp();
  if (P.clkprobe && blockIdx.x == 0 && tid == 0) P.clkprobe[0] = stamp_i;
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
