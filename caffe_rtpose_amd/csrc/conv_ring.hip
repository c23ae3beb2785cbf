// conv_ring.hip — the k x k (k = 3, 7) implicit-GEMM convolution with an LDS-DMA ring.
//
// Same GEMM formulation as conv_igemm.hip (flat padded pixel index = M, one A strip of
// BM+KS-1 pixels per filter row reused by its KS taps, one weight tile per tap), but the
// staging is asynchronous and deep:
//   * global -> LDS goes through `global_load_lds_dwordx4` (no VGPR round trip, 1 KiB per
//     wave-instruction), weight tiles into a 4-stage ring (3 taps in flight), the strip of the
//     NEXT filter row into the other half of a double buffer while the current row is consumed;
//   * one raw s_barrier per tap; the only waits are COUNTED `s_waitcnt vmcnt(N)` (never 0 in the
//     loop): N = DMA instructions this wave issued after the tile it is about to read.  The tail
//     issues dummy re-loads so that N follows one formula for every step.
// With ~56 B/clk/CU of L2 bandwidth and ~1000 cycles of latency a CU needs ~56 KB of loads in
// flight to stream the 0.9 MB of weights + activations one 64x64x6272 tile consumes; 3 weight
// tiles (48 KB) + the next strip (18 KB) in flight provide that, where the register-staged
// kernel had one 8 KB tile in flight and paid the full L2 latency on every tap.
//
// LDS image: DMA writes rows linearly (wave-uniform base + lane*16), CHB bytes per row with no
// padding, so bank conflicts are removed by an XOR swizzle of the 16-byte chunk index with the
// row number, applied to the SOURCE address of the DMA (activations) or baked into the packed
// weight layout, and again on the ds_read_b128 address.  CHB = 256: chunk ^= row & 15;
// CHB = 128: chunk ^= (row >> 1) & 7 — either way the 16 rows of a ds_read_b128 lane group hit
// 16 distinct 16-byte slots of the 256-byte bank row.
#include <atomic>

#include "conv_common.h"

namespace rtp {

template <int CHB> __device__ __host__ __forceinline__ int ring_swz(int row) {
  return CHB == 256 ? (row & 15) : ((row >> 1) & 7);
}

template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int CHB>
struct RingTraits {
  static constexpr int NCH = CHB / 16;
  static constexpr int G = CHB / 32;
  static constexpr int GPW = G / KSPLIT;
  static constexpr int SB = 4;
  static constexpr int RPI = 1024 / CHB;  // rows per DMA wave-instruction
  static constexpr int AROWS_RAW = BM + KS - 1;
  static constexpr int A_INSTR = ((AROWS_RAW + RPI - 1) / RPI + 3) / 4 * 4;
  static constexpr int A_PW = A_INSTR / 4;
  static constexpr int AROWS = A_INSTR * RPI;
  static constexpr int A_BYTES = AROWS * CHB;
  static constexpr int B_BYTES = BN * CHB;
  static constexpr int B_INSTR = B_BYTES / 1024;
  static constexpr int B_PW = B_INSTR / 4;
  static constexpr int TM = BM / WM / 32;
  static constexpr int TN = BN / WN / 32;
  static constexpr int STAGE_BYTES = 2 * A_BYTES + SB * B_BYTES;
  static constexpr int RED_BYTES = (KSPLIT - 1) * WM * WN * TM * TN * 16 * 64 * 4;
  static constexpr int LDS_BYTES = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
  static_assert(WM * WN * KSPLIT == 4, "4 waves");
  static_assert(G % KSPLIT == 0 && GPW >= 1, "k-groups split evenly");
  static_assert(B_INSTR % 4 == 0 && B_PW >= 1, "weight tile splits evenly over the waves");
  static_assert(KS == 3 || KS == 7, "strip reuse needs k > 1");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes, lane l lands at LDS byte lds_wave_base + 16*l.
// Issued from inline asm on purpose: hipcc cannot tell the ring stages apart, so with the builtin
// it drains the whole DMA queue (s_waitcnt vmcnt(0)) before the first ds_read of every step.
// Hidden in asm, the only VMEM waits in the main loop are the counted ones we place ourselves
// (no other VMEM instruction is in flight there).  M0 carries the LDS base and is restored.
__device__ __forceinline__ void dma16(const unsigned char* gsrc, unsigned char* lds_wave_base) {
  const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}

template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int CHB>
__global__ __launch_bounds__(256) void conv_ring_kernel(ConvParams P) {
  using TR = RingTraits<T, BM, BN, WM, WN, KSPLIT, KS, CHB>;
  constexpr int NCH = TR::NCH, GPW = TR::GPW, SB = TR::SB, RPI = TR::RPI;
  constexpr int A_PW = TR::A_PW, B_PW = TR::B_PW, TM = TR::TM, TN = TR::TN;
  constexpr int PAD = KS / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * TR::A_BYTES;

  const ConvProblem& pr = P.prob[blockIdx.z];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave / (WM * WN);
  const int wrem = wave % (WM * WN);
  const int wm0 = (wrem / WN) * (BM / WM);
  const int wn0 = (wrem % WN) * (BN / WN);
  const int lrow = lane & 31, lhalf = lane >> 5;

  const int img = blockIdx.x / P.tiles_per_img;
  const int m0 = (blockIdx.x % P.tiles_per_img) * BM;
  const int n0 = blockIdx.y * BN;
  const int nchunk = P.nchunk;
  const long pix_bytes = (long)P.in_cstride * (long)sizeof(T);
  const unsigned char* in_base = (const unsigned char*)pr.in +
      ((long)img * P.img_pix + (long)P.halo * P.Wp + m0 - (long)PAD * P.Wp - PAD) * pix_bytes;
  const unsigned char* w_base = (const unsigned char*)pr.w + (long)n0 * CHB;
  const long w_chunk_stride = (long)P.CoutP * CHB;
  const long w_tap_stride = (long)nchunk * w_chunk_stride;
  const long row_step_bytes = (long)P.Wp * pix_bytes;

  // ---- per-lane DMA source offsets -------------------------------------------------------
  // A strip: DMA instruction j covers rows [j*RPI, (j+1)*RPI); lane -> (row, physical chunk).
  // This wave issues instructions j = wave*A_PW + q.
  int a_src_off[A_PW];
#pragma unroll
  for (int q = 0; q < A_PW; ++q) {
    const int j = wave * A_PW + q;
    const int row = j * RPI + lane / NCH;
    const int cphys = lane % NCH;
    a_src_off[q] = row * (int)pix_bytes + ((cphys ^ ring_swz<CHB>(row)) * 16);
  }
  const int a_lds_off = wave * A_PW * 1024;  // + q*1024 (wave-uniform)
  const int b_src_off = wave * B_PW * 1024 + lane * 16;  // + q*1024: packed weights are already swizzled
  const int b_lds_off = wave * B_PW * 1024;

  auto issue_a = [&](int r, int chunk, int buf) {
    const unsigned char* p = in_base + (long)r * row_step_bytes + (long)chunk * CHB;
    unsigned char* l = sA + buf * TR::A_BYTES + a_lds_off;
#pragma unroll
    for (int q = 0; q < A_PW; ++q) dma16(p + a_src_off[q], l + q * 1024);
  };
  auto issue_b = [&](int r, int chunk, int s, int stage) {
    const unsigned char* p = w_base + (long)(r * KS + s) * w_tap_stride + (long)chunk * w_chunk_stride + b_src_off;
    unsigned char* l = sB + stage * TR::B_BYTES + b_lds_off;
#pragma unroll
    for (int q = 0; q < B_PW; ++q) dma16(p + q * 1024, l + q * 1024);
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int nstrips = KS * nchunk;
  const int T_steps = nstrips * KS;

  // issue-side cursor: coordinates of the next weight tile to fetch (runs 3 steps ahead)
  int ir = 0, ic = 0, is = 0, istep = 0;
  auto issue_next_b = [&]() {
    issue_b(ir, ic, is, istep & (SB - 1));
    ++istep;
    if (istep < T_steps) {  // past the end: keep re-issuing the last tile (dummy, keeps vmcnt uniform)
      if (++is == KS) { is = 0; if (++ic == nchunk) { ic = 0; ++ir; } }
    }
  };

  // ---- prologue: strip 0, weight tiles 0..2 ------------------------------------------------
  issue_a(0, 0, 0);
  issue_next_b();
  issue_next_b();
  issue_next_b();

  const int brow0 = wn0 + lrow;
  const int bswz = ring_swz<CHB>(brow0);

  int r = 0, chunk = 0, s = 0, abuf = 0;
  for (int t = 0; t < T_steps; ++t) {
    // Need: weight tile t (issued 3 steps ago) and, at s == 0, this strip (issued >= KS steps ago).
    // Issued after tile t: tiles t+1, t+2 and possibly one A strip (if a step with s == 0 lies in
    // [t-2, t-1], i.e. s is 1 or 2; a step issues its strip BEFORE its weight tile).
    const bool a_recent = (t >= 1) && (s == 1 || s == 2);
    if (a_recent) wait_vmcnt<2 * B_PW + A_PW>();
    else wait_vmcnt<2 * B_PW>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // refill (strip FIRST, then the weight tile, so that the strip a step needs is always older
    // than the weight tile it waits for): the buffers read in step t-1 are free now
    if (s == 0) {
      int nr = r, nc = chunk + 1;
      if (nc == nchunk) { nc = 0; nr = r + 1; }
      if (nr == KS) { nr = r; nc = chunk; }  // last strip: dummy reload of itself
      issue_a(nr, nc, abuf ^ 1);
    }
    issue_next_b();

    // ---- MFMA on tile t -------------------------------------------------------------------
    {
      const int arow = wm0 + lrow + s;
      const int aswz = ring_swz<CHB>(arow);
      const unsigned char* pa = sA + abuf * TR::A_BYTES + arow * CHB;
      const unsigned char* pb = sB + (t & (SB - 1)) * TR::B_BYTES + brow0 * CHB;
#pragma unroll
      for (int gi = 0; gi < GPW; ++gi) {
        const int cl = 2 * (kg * GPW + gi) + lhalf;
        uint4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(pa + i * 32 * CHB + ((cl ^ aswz) * 16));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *(const uint4*)(pb + j * 32 * CHB + ((cl ^ bswz) * 16));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
      }
    }
    if (++s == KS) {
      s = 0;
      abuf ^= 1;
      if (++chunk == nchunk) { chunk = 0; ++r; }
    }
  }
  // drain the dummy DMAs before LDS is reused / the workgroup exits
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  conv_epilogue<T, BM, BN, WM, WN, KSPLIT, TM, TN>(P, pr, acc, smem, kg, wrem, wm0, wn0, lane, img, m0, n0);
}

template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int CHB>
static hipError_t ring_launch_one(const ConvParams& P, int nprob, int N, hipStream_t stream) {
  using TR = RingTraits<T, BM, BN, WM, WN, KSPLIT, KS, CHB>;
  auto kern = conv_ring_kernel<T, BM, BN, WM, WN, KSPLIT, KS, CHB>;
  static std::atomic<unsigned> attr_mask{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_mask.load(std::memory_order_relaxed) & (1u << (dev & 31)))) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TR::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_mask.fetch_or(1u << (dev & 31), std::memory_order_relaxed);
  }
  dim3 grid(P.tiles_per_img * N, P.CoutP / BN, nprob);
  hipLaunchKernelGGL(kern, grid, dim3(256), TR::LDS_BYTES, stream, P);
  return hipGetLastError();
}

template <typename T, int KS>
static hipError_t ring_launch_cfg(int cfg, int chb, const ConvParams& P, int nprob, int N, hipStream_t stream) {
  if (chb == 256) {
    if (cfg == CFG_64x64) return ring_launch_one<T, 64, 64, 1, 1, 4, KS, 256>(P, nprob, N, stream);
    return hipErrorInvalidValue;
  }
  if (chb != 128) return hipErrorInvalidValue;
  switch (cfg) {
    case CFG_128x128: return ring_launch_one<T, 128, 128, 2, 2, 1, KS, 128>(P, nprob, N, stream);
    case CFG_64x128: return ring_launch_one<T, 64, 128, 1, 2, 2, KS, 128>(P, nprob, N, stream);
    case CFG_64x64: return ring_launch_one<T, 64, 64, 1, 1, 4, KS, 128>(P, nprob, N, stream);
    case CFG_128x64: return ring_launch_one<T, 128, 64, 2, 1, 2, KS, 128>(P, nprob, N, stream);
    default: return hipErrorInvalidValue;
  }
}

// chb: channel bytes per step (128 or 256; 256 only with the 64x64 tile)
hipError_t launch_conv_ring(int prec, int cfg, int ks, int chb, const ConvParams& P, int nprob, int N, hipStream_t stream) {
  if (prec == 0) {
    if (ks == 3) return ring_launch_cfg<_Float16, 3>(cfg, chb, P, nprob, N, stream);
    if (ks == 7) return ring_launch_cfg<_Float16, 7>(cfg, chb, P, nprob, N, stream);
  } else {
    if (ks == 3) return ring_launch_cfg<float, 3>(cfg, chb, P, nprob, N, stream);
    if (ks == 7) return ring_launch_cfg<float, 7>(cfg, chb, P, nprob, N, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace rtp
