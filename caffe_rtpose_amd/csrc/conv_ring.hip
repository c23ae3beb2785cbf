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
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace rtp {

template <int CHB> __device__ __host__ __forceinline__ int ring_swz(int row) {
  return CHB == 256 ? (row & 15) : ((row >> 1) & 7);
}

// A-strip swizzle.  ILV (interleaved fragment rows, see the consumer loop): the 16 lanes of a ds_read_b128 group read rows
// c + TM*l, so the swizzle key is row / TM (CHB = 256: 16 distinct 16-byte slots again; CHB = 128 keeps the plain key and
// takes a 2-way conflict on the few fresh reads that remain).
template <int CHB, int TM, bool ILV> __device__ __forceinline__ int ring_swz_a(int row) {
  if constexpr (ILV && CHB == 256) return (row / TM) & 15;
  else return ring_swz<CHB>(row);
}
// ILV with 128-byte rows (the 3x3 trunk layers' fp8-compensated launches, the 185-channel stage-entry layers): one 256-byte bank row
// holds a PAIR of strip rows, and the 16 lanes of a ds_read_b128 group read rows c + TM*l of one parity — a 3-bit key on the chunk
// index can give them only 8 distinct slots (2-way conflict, SQ_LDS_BANK_CONFLICT 692k per conv3_2 launch against 89k for the
// 256-byte dominant shape).  PAIR mode swizzles the 4-bit slot (parity, chunk) of the pair row R = row / 2 with (row / TM) & 15 —
// a function of R for even TM — so the 16 lanes hit 16 distinct 16-byte slots; consecutive rows (the DMA side) stay conflict-free.
template <int CHB, int TM, bool ILV> constexpr bool ring_pair_swz = ILV && CHB == 128 && (TM % 2) == 0;
// byte offset of the 16-byte piece `cl` of strip row `row` inside a strip buffer
template <int CHB, int TM, bool ILV> __device__ __forceinline__ int ring_a_offset(int row, int cl) {
  if constexpr (ring_pair_swz<CHB, TM, ILV>) {
    const int slot = (((row & 1) << 3) | cl) ^ ((row / TM) & 15);
    return (row >> 1) * 256 + slot * 16;
  } else
    return row * CHB + ((cl ^ ring_swz_a<CHB, TM, ILV>(row)) * 16);
}

// every lane takes the value of the next higher lane (v_mov_b32_dpp wave_shl:1).  bound_ctrl: lane 63 (no source) gets 0 and
// the destination is not tied to an "old" value (with old = src hipcc emitted a v_mov in front of every DPP move)
__device__ __forceinline__ uint4 lanes_down1(const uint4& v) {
  uint4 r;
  r.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x130, 0xf, 0xf, true);
  r.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x130, 0xf, 0xf, true);
  r.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x130, 0xf, 0xf, true);
  r.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x130, 0xf, 0xf, true);
  return r;
}

// ds_read_b128 into `v` for the lanes of `mask` only; the other lanes of v keep their value.  Hidden in asm on purpose: as a
// divergent `if` the partial overwrite of a fragment made hipcc spill ~800 registers in the fp8 variant (8-register MFMA
// operands).  The compiler does not see this LDS read, so EVERY use of the fragment must sit behind an explicit lgkmcnt(0)
// (the wait at the top of the next tap; one more in front of the last tap).
typedef unsigned uint4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_read_lanes(uint4& v, unsigned lds_addr, unsigned long long mask) {
  uint4v_t t = __builtin_bit_cast(uint4v_t, v);
  unsigned long long keep;
  asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %3\n\ts_nop 0\n\tds_read_b128 %0, %2\n\ts_mov_b64 exec, %1"
               : "+v"(t), "=&s"(keep) : "v"(lds_addr), "s"(mask));
  v = __builtin_bit_cast(uint4, t);
}

template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int CHB, int SB_>
struct RingTraits {
  static constexpr int NCH = CHB / 16;
  static constexpr int G = CHB / 32;
  static constexpr int GPW = G / KSPLIT;
  static constexpr int SB = SB_;  // weight-tile ring stages (SB-1 taps in flight)
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
  static constexpr int RED_BYTES = KSPLIT * BM * BN * 4;  // epilogue: all partials as [kg][row][col] fp32
  static constexpr int LDS_BYTES = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
  static_assert(WM * WN * KSPLIT == 4, "4 waves");
  static_assert(G % KSPLIT == 0 && GPW >= 1, "k-groups split evenly");
  static_assert(B_INSTR % 4 == 0 && B_PW >= 1, "weight tile splits evenly over the waves");
  static_assert(KS == 3 || KS == 7, "strip reuse needs k > 1");
  static_assert(SB >= 3 && (SB - 2) * B_PW + 2 * A_PW <= 63 && (SB - 1) * B_PW + A_PW <= 63, "vmcnt is a 6-bit counter");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// Counted VMEM wait + "all of this wave's LDS reads have RETURNED".  The second half matters: the
// compiler is free to sink the MFMAs of a step (and the lgkmcnt wait in front of them) below the
// next step's barrier, so without it a wave could pass the barrier with ds_reads of the stage that
// another wave is about to overwrite by DMA still in flight (seen as run-to-run differences when
// a co-resident workgroup keeps the LDS pipeline busy).
// Issued through the builtin (not asm) so that hipcc's own waitcnt bookkeeping sees "all LDS reads
// returned" and does not re-wait (lgkmcnt(k)) in front of the MFMAs that consume the previous tap's
// fragments.  gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14].
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "6-bit vmcnt");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (0 << 8));
  asm volatile("" ::: "memory");
}

// LDS-DMA groups.  Issued from inline asm on purpose: hipcc cannot tell the ring stages apart, so
// with the builtin it drains the whole DMA queue (s_waitcnt vmcnt(0)) before the first ds_read of
// every step.  Hidden in asm, the only VMEM waits in the main loop are the counted ones we place
// ourselves (no other VMEM instruction is in flight there).  One group = NQ wave-instructions of
// 64 lanes x 16 bytes: global address = sbase + voff + q*1024, LDS address = lds_addr + q*1024 +
// 16*lane (the instruction offset applies to both sides).  M0 carries the LDS base; restored.
template <int NQ>
__device__ __forceinline__ void dma_group_same(const unsigned char* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  static_assert(NQ >= 1 && NQ <= 4, "13-bit instruction offset");
  if constexpr (NQ == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  else if constexpr (NQ == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  else if constexpr (NQ == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// per-instruction offsets: voff[q] must already have q*1024 subtracted
template <int NQ>
__device__ __forceinline__ void dma_group_each(const unsigned char* sbase, const unsigned* voff, unsigned lds_addr) {
  unsigned keep;
  static_assert(NQ >= 1 && NQ <= 4, "13-bit instruction offset");
  if constexpr (NQ == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, %5 offset:2048\n\tglobal_load_lds_dwordx4 %4, %5 offset:3072\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sbase), "s"(lds_addr) : "memory");
  else if constexpr (NQ == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %4\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, %4 offset:2048\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(sbase), "s"(lds_addr) : "memory");
  else if constexpr (NQ == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "s"(sbase), "s"(lds_addr) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff[0]), "s"(sbase), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ unsigned lds_addr_of(const unsigned char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

// Pipeline (P = the tap whose fragments are being PREFETCHED, tap P-1 is being multiplied):
//   prologue : DMA strip 0, weight tile 0, strip 1, weight tiles 1..SB-1; wait strip 0 + tile 0; read fragments(0)
//   iteration P = 1..T-1:
//       wait (counted) for tile P [+ its strip when s(P) == 0], lgkmcnt(0), s_barrier
//         -> every wave has finished READING tile P-1 (its fragments sit in registers), so that
//            ring stage and the strip buffer of the previous filter row are free
//       DMA [strip of filter row sc(P)+1, if s(P) == 0], then weight tile P-1+SB (dummy past the end)
//       ds_read fragments(P) -> next        } the LDS latency of tap P hides under
//       MFMAs of tap P-1 on cur; cur = next } the MFMAs of tap P-1
//   epilogue : MFMAs of tap T-1
// vmcnt to wait for at the top of iteration P (s = s(P), FIRST = P is in the first strip):
//  (1) tile P was issued SB-1 iterations ago.  Younger than it: the SB-2 tiles of iterations
//      P-(SB-2)..P-1 and one strip for every such iteration u with s(u) == 0 (an iteration issues
//      its strip BEFORE its weight tile); in the first strip only iterations u >= 1 exist.
//  (2) at s == 0 (from the third strip on; strips 0 and 1 are older than weight tile 1) the strip of
//      this filter row is needed; it was issued KS iterations ago: younger = KS tiles.
// SB <= KS+1 keeps "at most one strip per window" and makes FIRST the only special case.
template <int KS, int SB, bool FIRST, int B_PW, int A_PW>
constexpr int ring_wait_count(int s) {
  int na = 0;
  for (int d = 1; d <= SB - 2; ++d) {
    if (FIRST && d >= s) continue;  // iteration P-d would be <= 0 (prologue)
    int sd = s - d;
    while (sd < 0) sd += KS;
    if (sd == 0) ++na;
  }
  int n = (SB - 2) * B_PW + na * A_PW;
  if (s == 0 && !FIRST) {
    const int n2 = KS * B_PW;
    if (n2 < n) n = n2;
  }
  return n;
}

// The LAST strip of a tile issues nothing that does not exist (no strip for a filter row behind the last one, no weight tile behind tile
// T-1): what is younger than tile P shrinks to the tiles that are left, min(SB-2, KS-1-s), and the last tap waits for vmcnt(0) — every DMA
// has landed when the K loop ends.  (Rounds 2-5 issued dummy re-loads through the tail so that one formula served every step; the DMA waves then
// had to wait for those dummies — a trip to L2 / memory issued one tap earlier — before the barrier that lets the consumers start the epilogue.)
template <int KS, int SB, int B_PW>
constexpr int ring_wait_last(int s) {
  const int left = KS - 1 - s;
  int n = (left < SB - 2 ? left : SB - 2) * B_PW;
  if (s == 0 && KS * B_PW < n) n = KS * B_PW;
  return n;
}

// SPEC = wave specialisation: 8 waves, one producer + one consumer per SIMD.  Waves 4..7 only issue
// the LDS-DMA (and wait for it, counted); waves 0..3 only ds_read + MFMA (+ the epilogue).  The
// in-order VMEM issue of a wave (~50 cycles per 1 KiB piece, 3-5 pieces per tap) then no longer sits
// in front of the same wave's MFMAs.  Same LDS image, same wait counts, one barrier per tap for all.
// VAR (experiments on the wave-specialised consumer loop, selected by ConvParams::variant; 0 = production):
//   production: all ds_reads of the next tap are issued during the FIRST half of the tap's MFMAs (2 per MFMA), so their LDS
//       latency is covered by the second half instead of being waited for in front of the barrier (-4 % on the dominant launch)
//   4 = the round-1 schedule: one ds_read per MFMA over the whole tap
//   17 = 16 with all-zero operands (data dependence of the MFMA rate)
//   2 = s_setprio 3 on the consumer waves; 3 = accumulators in AGPRs (timing only)
//   ablations (wrong results, timing only): 11 = no ds_reads, 12 = no MFMAs, 13 = no LDS-DMA,
//   15 = no per-tap barriers (and no counted waits), 16 = 15 + no ds_reads (a bare MFMA stream)
//   100 = the fp8-compensated variant of the kernel (layers with q chunks, ConvParams::q_from): its own instantiation, so that
//       the plain kernel's register allocation (254 VGPRs with the front-loaded reads) is not disturbed; reads one per MFMA
//   ILV (wave-specialised kernels): A fragments are NOT re-read from LDS for every tap.  Fragment i of a wave holds the pixels
//       TM*l + i (l = lane & 31) instead of 32*i + l, so tap s+1's fragment i < TM-1 is tap s's fragment i+1 (a register rename
//       in the unrolled code) and fragment TM-1 is tap s's fragment 0 moved down one lane (4 v_mov_b32_dpp) with ONE fresh row
//       read by lanes 31 / 63.  Only the first tap of a strip reads all A fragments from LDS: 6/7 (7x7) or 2/3 (3x3) of the A
//       reads = 43 % / 33 % of all ds_read traffic disappear.  The ablation showed the launch is bound by the matrix pipe at a
//       clock that drops when the LDS is busy (no ds_reads: 2.2 instead of 1.8 GHz); same MFMA order, bit-identical results.
//   POOL: the convolution's only consumer is a 2x2 / stride-2 max pooling layer.  The M tile is TWO image rows x BM/2 pixels instead
//       of BM consecutive flat pixels: LDS strip rows [0, BM/2+KS-1) hold the pixels of row 2*pair around x0 .. x0+BM/2-1, rows
//       [BM/2+KS-1, 2*(BM/2+KS-1)) the same columns one image row below (only the per-lane DMA source offsets differ — they are
//       loop-invariant anyway), wave row wm multiplies tile row wm, and the epilogue pools in LDS and writes only the pooled
//       tensor (conv_common.h).  Tiles walk the (row pair, x) space with an EVEN pitch pool_wq >= W + pad, so x0 is always even.
template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int CHB, int SB_, int MINW, bool SPEC, int VAR = 0, bool ILV = false, bool POOL = false>
__global__ __launch_bounds__(SPEC ? 512 : 256, MINW) void conv_ring_kernel(ConvParams P) {
  static_assert(!ILV || SPEC, "interleaved fragment rows exist in the wave-specialised kernels only");
  static_assert(!POOL || (!ILV && WM == 2 && BM == 128), "fused pooling: two wave rows, one per image row of the tile");
  using TR = RingTraits<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_>;
  constexpr int NCH = TR::NCH, GPW = TR::GPW, SB = TR::SB, RPI = TR::RPI;
  constexpr int A_PW = TR::A_PW, B_PW = TR::B_PW, TM = TR::TM, TN = TR::TN;
  constexpr int PAD = KS / 2;
  static_assert(SB <= KS + 1, "ring deeper than a filter row is not supported by the wait-count formula");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * TR::A_BYTES;

  const KStamp kstamp_(P.stamp);
  const BlockCoord bc = decode_block(P, P.CoutP / BN, P.tiles_per_img * P.nimg);
  const ConvProblem& pr = P.prob[bc.prob];
  const int tid = threadIdx.x;
  unsigned long long clk0 = 0, wall0 = 0;
  if (P.clkprobe && tid == 0 && blockIdx.x == 0) { clk0 = clock64(); wall0 = wall_clock64(); }
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = SPEC && wave_all >= 4;
  const int wave = SPEC ? (wave_all & 3) : wave_all;  // consumer role / share of the DMA pieces
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
  constexpr int PHALF = BM / 2 + KS - 1;  // POOL: LDS strip rows per image row of the tile
  static_assert(!POOL || 2 * PHALF <= TR::AROWS, "both image rows of a pooled tile fit the strip buffer");
  int pool_pair = 0, pool_x0 = 0;
  if constexpr (POOL) {
    const int q0 = bc.mtile * (BM / 2);
    pool_pair = q0 / P.pool_wq;
    pool_x0 = q0 - pool_pair * P.pool_wq;
  }
  // strip (r, chunk) starts at a_ptr; consecutive strips: chunk += 1 (CHB bytes), then next filter row
  const unsigned char* a_ptr = (const unsigned char*)pr.in +
      ((long)img * P.img_pix + (long)P.halo * P.Wp + (POOL ? 2 * pool_pair * P.Wp : m0) - (long)PAD * P.Wp - PAD) * pix_bytes;
  const long row_step_bytes = (long)P.Wp * pix_bytes;
  // weights are packed in STEP order [r][chunk][s][CoutP][CHB]: the tile pointer just increments
  const long w_tile_stride = (long)P.CoutP * CHB;
  const unsigned char* b_ptr = (const unsigned char*)pr.w + (long)n0 * CHB;

  // ---- per-lane DMA source offsets (loop invariant) ---------------------------------------
  unsigned a_voff[A_PW];
#pragma unroll
  for (int q = 0; q < A_PW; ++q) {
    const int j = wave * A_PW + q;
    const int row = j * RPI + lane / NCH;
    const int cphys = lane % NCH;
    int src_pix = row;  // pixel of LDS strip row `row`, relative to a_ptr
    if constexpr (POOL) {
      const int sel = row >= PHALF ? 1 : 0;
      // x position in the (row pair, x) walk (real image columns; element (y, x) lives at padded column x + halo).  [0, W) are
      // pixels; [W, pitch) are zero columns — the right halo of this pair and the left halo of the next one — served by the
      // gap pixel that follows the row in memory; every further `pitch` positions = one row pair down.
      // (the engine fuses only where the pitch exceeds a tile row + halo: at most ONE wrap per tile, no division on this path —
      // a workgroup of conv1_2 lives ~8000 cycles, its prologue is not hidden behind anything)
      const int a = pool_x0 - PAD + (row - sel * PHALF);
      const bool wrap = a >= P.pool_wq;
      const int ar = wrap ? a - P.pool_wq : a;
      src_pix = (wrap ? 2 * P.Wp : 0) + (ar < P.W ? ar : P.W) + P.halo + PAD + sel * P.Wp;
    }
    if constexpr (ring_pair_swz<CHB, TR::TM, ILV>) {  // LDS position (row, cphys) holds the logical piece (row', chunk') with the pair row's key (POOL is never ILV)
      const int slot = (((row & 1) << 3) | cphys) ^ ((((row >> 1) * 2) / TR::TM) & 15);
      a_voff[q] = (unsigned)(((row & ~1) + (slot >> 3)) * (int)pix_bytes + (slot & 7) * 16 - (q & 3) * 1024);
    } else
    a_voff[q] = (unsigned)(src_pix * (int)pix_bytes + ((cphys ^ ring_swz_a<CHB, TR::TM, ILV>(row)) * 16) - (q & 3) * 1024);
  }
  const unsigned b_voff = (unsigned)(wave * B_PW * 1024 + lane * 16);
  const unsigned sA_addr = lds_addr_of(sA) + wave * A_PW * 1024;
  const unsigned sB_addr = lds_addr_of(sB) + wave * B_PW * 1024;

  int a_chunk = 0;  // chunk index of the strip a_ptr points to
  auto issue_a = [&](int buf) __attribute__((always_inline)) {  // strip at a_ptr -> sA[buf]; advances a_ptr to the next strip
    const unsigned l = sA_addr + buf * TR::A_BYTES;
    if constexpr (A_PW <= 4) dma_group_each<A_PW>(a_ptr, a_voff, l);
    else if constexpr (A_PW <= 8) { dma_group_each<4>(a_ptr, a_voff, l); dma_group_each<A_PW - 4>(a_ptr, a_voff + 4, l + 4096); }
    else { dma_group_each<4>(a_ptr, a_voff, l); dma_group_each<4>(a_ptr, a_voff + 4, l + 4096); dma_group_each<A_PW - 8>(a_ptr, a_voff + 8, l + 8192); }
    // past the last strip this keeps walking: harmless dummy reads of arena memory (tail pad)
    if (++a_chunk == nchunk) { a_chunk = 0; a_ptr += row_step_bytes - (long)P.row_back; }
    else if (a_chunk == P.wrap_at) a_ptr -= (long)(P.wrap_at - 1) * CHB;  // split precision: the a_hi chunks once more (x W_lo)
    else if (a_chunk == P.q_from) a_ptr += P.jump_delta;                   // fp8 compensation: on to the tensor's q block
    else a_ptr += CHB;
  };
  auto issue_b = [&](int stage) __attribute__((always_inline)) {  // tile at b_ptr -> sB[stage]; advances b_ptr
    const unsigned l = sB_addr + stage * TR::B_BYTES;
    if constexpr (B_PW <= 4) dma_group_same<B_PW>(b_ptr, b_voff, l);
    else { dma_group_same<4>(b_ptr, b_voff, l); dma_group_same<B_PW - 4>(b_ptr + 4096, b_voff, l + 4096); }
    b_ptr += w_tile_stride;
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int nstrips = KS * nchunk;
  const int brow0 = wn0 + lrow;
  const int bswz = ring_swz<CHB>(brow0);
  const unsigned char* pb_lane = sB + brow0 * CHB;
  const int arow_base = (POOL ? (wrem / WN) * PHALF : wm0) + lrow;

  uint4 fa[GPW][TM], fb[GPW][TN];      // fragments of the tap being multiplied
  uint4 na_[GPW][TM], nb_[GPW][TN];    // fragments of the tap being prefetched
  auto read_frags = [&](int s, int abuf, int stage, uint4 (&ra)[GPW][TM], uint4 (&rb)[GPW][TN]) {
    const int arow = arow_base + s;
    const int aswz = ring_swz<CHB>(arow);
    const unsigned char* pa = sA + abuf * TR::A_BYTES + arow * CHB;
    const unsigned char* pb = pb_lane + stage * TR::B_BYTES;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int cl = 2 * (kg * GPW + gi) + lhalf;
#pragma unroll
      for (int i = 0; i < TM; ++i) ra[gi][i] = *(const uint4*)(pa + i * 32 * CHB + ((cl ^ aswz) * 16));
#pragma unroll
      for (int j = 0; j < TN; ++j) rb[gi][j] = *(const uint4*)(pb + j * 32 * CHB + ((cl ^ bswz) * 16));
    }
  };
  auto mma_all = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(fa[gi][i], fb[gi][j], acc[i][j]);
  };
  auto rotate = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[gi][i] = na_[gi][i];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[gi][j] = nb_[gi][j];
    }
  };

  if constexpr (SPEC) {
    int abuf = 0, st = 1, ist = 0;
    if (producer) {
      // strip 0 and tile 0 FIRST: the first tap starts when these ~36-50 KiB have landed, with strip 1 and tiles 1..SB-1 still in
      // flight behind them (every CU runs its prologue at the same time at ~11 B/clk/CU: each KiB less in front of the first MFMA
      // is ~90 cycles of every workgroup's life).  Loads return in issue order, so the steady-state wait counts are unchanged:
      // strip 1 is older than tile 1, which the first counted wait of the loop asks for.
      issue_a(0);
      issue_b(0);
      issue_a(1);
#pragma unroll
      for (int i = 1; i < SB; ++i) issue_b(i);
      wait_vmcnt<(SB - 1) * B_PW + A_PW>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      auto pstep = [&](auto s_tag, auto first_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr int s = decltype(s_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;   // the tile's last strip: see ring_wait_last
        if constexpr (VAR != 15 && VAR != 16 && VAR != 17) {
          if constexpr (LAST) wait_vmcnt<ring_wait_last<KS, SB, B_PW>(s)>();
          else wait_vmcnt<ring_wait_count<KS, SB, FIRST, B_PW, A_PW>(s)>();
          __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        if constexpr (VAR != 13) {
          if constexpr (s == 0 && !LAST) { abuf ^= 1; issue_a(abuf ^ 1); }
          if constexpr (!LAST || s + SB - 1 <= KS - 1) issue_b(ist);
        }
        ist = (ist + 1 == SB) ? 0 : ist + 1;
      };
      const std::false_type NL{};
      const std::true_type LS{};
      pstep(std::integral_constant<int, 1>{}, std::true_type{}, NL);
      pstep(std::integral_constant<int, 2>{}, std::true_type{}, NL);
      if constexpr (KS == 7) {
        pstep(std::integral_constant<int, 3>{}, std::true_type{}, NL);
        pstep(std::integral_constant<int, 4>{}, std::true_type{}, NL);
        pstep(std::integral_constant<int, 5>{}, std::true_type{}, NL);
        pstep(std::integral_constant<int, 6>{}, std::true_type{}, NL);
      }
      for (int sc = 1; sc < nstrips - 1; ++sc) {
        pstep(std::integral_constant<int, 0>{}, std::false_type{}, NL);
        pstep(std::integral_constant<int, 1>{}, std::false_type{}, NL);
        pstep(std::integral_constant<int, 2>{}, std::false_type{}, NL);
        if constexpr (KS == 7) {
          pstep(std::integral_constant<int, 3>{}, std::false_type{}, NL);
          pstep(std::integral_constant<int, 4>{}, std::false_type{}, NL);
          pstep(std::integral_constant<int, 5>{}, std::false_type{}, NL);
          pstep(std::integral_constant<int, 6>{}, std::false_type{}, NL);
        }
      }
      // the last strip (nstrips = KS * chunks >= 3: never the first one): only what exists is issued, the last tap's wait is vmcnt(0)
      pstep(std::integral_constant<int, 0>{}, std::false_type{}, LS);
      pstep(std::integral_constant<int, 1>{}, std::false_type{}, LS);
      pstep(std::integral_constant<int, 2>{}, std::false_type{}, LS);
      if constexpr (KS == 7) {
        pstep(std::integral_constant<int, 3>{}, std::false_type{}, LS);
        pstep(std::integral_constant<int, 4>{}, std::false_type{}, LS);
        pstep(std::integral_constant<int, 5>{}, std::false_type{}, LS);
        pstep(std::integral_constant<int, 6>{}, std::false_type{}, LS);
      }
      wait_vmcnt<0>();  // (nothing is in flight any more: the last tap waited for everything)
      __builtin_amdgcn_s_barrier();
      // the DMA waves stay for the epilogue: they take half of the (pixel, 16-channel chunk) items of the staged tile (dump = false)
      if constexpr (POOL) conv_epilogue_pool<T, BM, BN, WM, WN, KSPLIT, TM, TN, 512>(P, pr, acc, smem, kg, wm0, wn0, lane, img, pool_pair, pool_x0, n0, false);
      else conv_epilogue<T, BM, BN, WM, WN, KSPLIT, TM, TN, ILV, 512>(P, pr, acc, smem, kg, wrem, wm0, wn0, lane, img, m0, n0, false);
      return;
    }
    // ---- consumers ----
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (P.clkprobe && tid == 0 && blockIdx.x == 0) P.clkprobe[2] = wall_clock64() - wall0;   // strip 0 + weight tile 0 have landed
    if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(3);
    // ILV: address of the 16-byte piece `cl` of this lane's row of A fragment i for tap s
    auto a_ilv = [&](int i, int s, int abuf_, int cl) __attribute__((always_inline)) {
      const int row = wm0 + TM * lrow + i + s;
      return (const uint4*)(sA + abuf_ * TR::A_BYTES + ring_a_offset<CHB, TM, true>(row, cl));
    };
    if constexpr (ILV) {
      const unsigned char* pb0 = pb_lane;
#pragma unroll
      for (int gi = 0; gi < GPW; ++gi) {
        const int cl = 2 * (kg * GPW + gi) + lhalf;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[gi][i] = *a_ilv(i, 0, 0, cl);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[gi][j] = *(const uint4*)(pb0 + j * 32 * CHB + ((cl ^ bswz) * 16));
      }
    } else {
      read_frags(0, 0, 0, fa, fb);
    }
    if constexpr (VAR == 17) {
#pragma unroll
      for (int gi = 0; gi < GPW; ++gi) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[gi][i] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[gi][j] = make_uint4(0, 0, 0, 0);
      }
    }
    // ---- The fragment registers fa/fb (being multiplied) and na_/nb_ (being read) serve both chunk types: an fp16 tap uses
    //      every 16-byte fragment as one k-group (32x32x16 MFMA); an fp8 compensation tap (q chunk) uses fragments 2jj, 2jj+1
    //      together as the 8-register operand of one k-block of v_mfma_scale_f32_32x32x64_f8f6f4 (CANQ: the wave's share of a
    //      chunk is a whole number of 64-byte k-blocks). ----
    constexpr bool QVAR = VAR >= 100 && VAR <= 103;  // 100 = production q kernel; 101..103 = timing-only experiments on it (wrong results)
    constexpr bool CANQ = QVAR && std::is_same<T, _Float16>::value && (GPW % 2 == 0);
    constexpr int NKB = CANQ ? GPW / 2 : 1;
    constexpr int NRD_C = GPW * (TM + TN);
    const int sb_even = 127 - P.wq_exp, sb_odd = 127 - P.wq_exp - 11;  // E8M0 scales of W8 / W_lo8
    // Fragment read #idx of the tap being prefetched: k-group gi = 32 bytes, lane half lhalf takes 16 of them — the SAME
    // addresses for both chunk types.  For a q chunk that hands lane half lhalf the 16-byte pieces #lhalf and #(2+lhalf) of
    // each 64-byte k-block instead of 32 contiguous bytes; A and B are read alike, so the MFMA still pairs equal K positions
    // (a dot product does not care about the order of its terms).
    auto read_one_c = [&](auto s_tag, int idx, const unsigned char* pa, const unsigned char* pb, int aswz, int abuf_) __attribute__((always_inline)) {
      constexpr int s = decltype(s_tag)::value;
      const int gi = idx / (TM + TN), q = idx % (TM + TN);
      const int cl = 2 * (kg * GPW + gi) + lhalf;
      if (q >= TM) nb_[gi][q - TM] = *(const uint4*)(pb + (q - TM) * 32 * CHB + ((cl ^ bswz) * 16));
      else if constexpr (!ILV) na_[gi][q] = *(const uint4*)(pa + q * 32 * CHB + ((cl ^ aswz) * 16));
      else if constexpr (s == 0) na_[gi][q] = *a_ilv(q, 0, abuf_, cl);       // first tap of a strip: fresh rows
      else if (q < TM - 1) na_[gi][q] = fa[gi][q + 1];                        // tap s wants the pixels TM*l + q + s = fragment q+1 of tap s-1 (held in fa)
      else na_[gi][q] = lanes_down1(fa[gi][0]);                              // TM*l + TM-1 + s = TM*(l+1) + s-1: tap s-1's fragment 0, one lane up
    };
    // ILV: the one row the shift cannot supply (lanes 31 and 63 of fragment TM-1): exec-masked reads, see lds_read_lanes
    auto read_boundary_c = [&](auto s_tag, int abuf_) __attribute__((always_inline)) {
      constexpr int s = decltype(s_tag)::value;
#pragma unroll
      for (int gi = 0; gi < GPW; ++gi)
        lds_read_lanes(na_[gi][TM - 1], lds_addr_of((const unsigned char*)a_ilv(TM - 1, s, abuf_, 2 * (kg * GPW + gi) + lhalf)), 0x8000000080000000ull);
    };
    // MFMA #m of the tap being multiplied.  MQ: fp8 compensation tap — k-block jj of this wave is the (kg*NKB + jj)-th of the
    // chunk: even = a_lo8 x W8, odd = a8 x W_lo8.
    auto mma_one_c = [&](auto mq_tag, int m) __attribute__((always_inline)) {
      constexpr bool MQ = decltype(mq_tag)::value;
      if constexpr (MQ && CANQ) {
        const int jj = m / (TM * TN), r = m % (TM * TN);
        const bool odd = ((kg * NKB + jj) & 1) != 0;
        if constexpr (VAR == 101)       // timing only: the same registers read as fp4 (A) x fp6 e2m3 (B): the 2x-rate MFMA of an MX fp4 / fp6 correction
          mma_q<4, 2>(fa[2 * jj][r / TN], fa[2 * jj + 1][r / TN], fb[2 * jj][r % TN], fb[2 * jj + 1][r % TN], acc[r / TN][r % TN], odd ? Q_SA_HI : Q_SA_LO, odd ? sb_odd : sb_even);
        else if constexpr (VAR == 102) {  // timing only: ONE of the two corrections (the odd k-blocks' MFMAs are not issued)
          if (!odd) mma_fp8(fa[2 * jj][r / TN], fa[2 * jj + 1][r / TN], fb[2 * jj][r % TN], fb[2 * jj + 1][r % TN], acc[r / TN][r % TN], Q_SA_LO, sb_even);
        } else if constexpr (VAR == 103) {  // timing only: no correction MFMAs at all (the q bytes still travel)
        } else
        mma_fp8(fa[2 * jj][r / TN], fa[2 * jj + 1][r / TN], fb[2 * jj][r % TN], fb[2 * jj + 1][r % TN], acc[r / TN][r % TN],
                odd ? Q_SA_HI : Q_SA_LO, odd ? sb_odd : sb_even);
      } else if constexpr (VAR == 3) {  // timing only: accumulators forced into AGPRs (asm MFMA: no hazard padding by hipcc)
        const int gi = m / (TM * TN), r = m % (TM * TN);
        const half8_t av = __builtin_bit_cast(half8_t, fa[gi][r / TN]);
        const half8_t bv = __builtin_bit_cast(half8_t, fb[gi][r % TN]);
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[r / TN][r % TN]) : "v"(av), "v"(bv));
      } else if constexpr (VAR != 12) {
        const int gi = m / (TM * TN), r = m % (TM * TN);
        Mma<T>::run(fa[gi][r / TN], fb[gi][r % TN], acc[r / TN][r % TN]);
      }
    };
    // one tap: multiply tap P-1 (type MQ) while the fragments of tap P (type RQ) are read
    auto cstep = [&](auto s_tag, auto mq_tag, auto rq_tag) __attribute__((always_inline)) {
      constexpr int s = decltype(s_tag)::value;
      constexpr bool MQ = decltype(mq_tag)::value && CANQ;
      constexpr bool RQ = decltype(rq_tag)::value && CANQ;
      constexpr int NMMA_C = (MQ ? NKB : GPW) * TM * TN;
      if constexpr (VAR != 15 && VAR != 16 && VAR != 17) {
        wait_vmcnt<63>();  // lgkmcnt(0): this wave's ds_reads of the stage about to be overwritten have returned
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
      if constexpr (s == 0) abuf ^= 1;
      const int arow = arow_base + s;
      const int aswz = ring_swz<CHB>(arow);
      const unsigned char* pa = sA + abuf * TR::A_BYTES + arow * CHB;
      const unsigned char* pb = pb_lane + st * TR::B_BYTES;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NMMA_C; ++m) {
        mma_one_c(mq_tag, m);
        if constexpr (VAR != 4 && (!QVAR || ILV) && VAR != 11 && VAR != 16 && VAR != 17) {  // reads front-loaded over the first half of the tap
          constexpr int HALF = NMMA_C / 2 > 0 ? NMMA_C / 2 : 1;
          if (m < HALF) {
#pragma unroll
            for (int rd = m * NRD_C / HALF; rd < (m + 1) * NRD_C / HALF; ++rd) read_one_c(s_tag, rd, pa, pb, aswz, abuf);
          }
          if constexpr (ILV && s != 0) { if (m == HALF - 1) read_boundary_c(s_tag, abuf); }
        } else if constexpr (VAR == 4 || (QVAR && !ILV)) {
#pragma unroll
          for (int rd = m * NRD_C / NMMA_C; rd < (m + 1) * NRD_C / NMMA_C; ++rd) read_one_c(s_tag, rd, pa, pb, aswz, abuf);
          if constexpr (ILV && s != 0) { if (m == NMMA_C - 1) read_boundary_c(s_tag, abuf); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      st = (st + 1 == SB) ? 0 : st + 1;
      rotate();
    };
    const std::true_type QT{};
    const std::false_type QF{};
    // taps 1..KS-1 of a strip whose chunk type is known
    auto strip_tail = [&](auto q_tag) __attribute__((always_inline)) {
      cstep(std::integral_constant<int, 1>{}, q_tag, q_tag);
      cstep(std::integral_constant<int, 2>{}, q_tag, q_tag);
      if constexpr (KS == 7) {
        cstep(std::integral_constant<int, 3>{}, q_tag, q_tag);
        cstep(std::integral_constant<int, 4>{}, q_tag, q_tag);
        cstep(std::integral_constant<int, 5>{}, q_tag, q_tag);
        cstep(std::integral_constant<int, 6>{}, q_tag, q_tag);
      }
    };
    bool last_q = false;
    if constexpr (CANQ) {
      // fp8-compensated layer: every filter row = ncp fp16 strips, then ncp q strips.  The chunk type of every step is
      // STATIC in this nest (a dynamic type per strip made the register state at the loop header the union of both).
      const int ncp = P.q_from;
      auto row_body = [&]() __attribute__((always_inline)) {  // entered with the fragments of tap 0 of the row's first fp16 strip in fa / fb
        strip_tail(QF);
        for (int c = 1; c < ncp; ++c) { cstep(std::integral_constant<int, 0>{}, QF, QF); strip_tail(QF); }
        cstep(std::integral_constant<int, 0>{}, QF, QT);
        strip_tail(QT);
        for (int c = 1; c < ncp; ++c) { cstep(std::integral_constant<int, 0>{}, QT, QT); strip_tail(QT); }
      };
      row_body();  // filter row 0 (peeled: the loop below is entered and re-entered with q fragments live, never fp16 ones)
      for (int r = 1; r < KS; ++r) {
        cstep(std::integral_constant<int, 0>{}, QT, QF);  // last q tap of the previous row | first fp16 strip of this one
        row_body();
      }
      last_q = true;
    } else {
      strip_tail(QF);
      for (int sc = 1; sc < nstrips; ++sc) { cstep(std::integral_constant<int, 0>{}, QF, QF); strip_tail(QF); }
    }
    // last tap
    if constexpr (ILV) {  // lgkmcnt(0): the masked boundary reads are invisible to the compiler; no MFMA may be scheduled above the wait
      wait_vmcnt<63>();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (CANQ && last_q) {
#pragma unroll
      for (int m = 0; m < NKB * TM * TN; ++m) mma_one_c(QT, m);
    } else {
#pragma unroll
      for (int m = 0; m < GPW * TM * TN; ++m) mma_one_c(QF, m);
    }
    wait_vmcnt<63>();
    __builtin_amdgcn_s_barrier();  // pairs with the producers' drain barrier
    if (P.clkprobe && tid == 0 && blockIdx.x == 0) P.clkprobe[3] = wall_clock64() - wall0;   // K loop done
    if constexpr (POOL) conv_epilogue_pool<T, BM, BN, WM, WN, KSPLIT, TM, TN, 512>(P, pr, acc, smem, kg, wm0, wn0, lane, img, pool_pair, pool_x0, n0);
    else conv_epilogue<T, BM, BN, WM, WN, KSPLIT, TM, TN, ILV, 512>(P, pr, acc, smem, kg, wrem, wm0, wn0, lane, img, m0, n0);
    if (P.clkprobe && tid == 0 && blockIdx.x == 0) {  // shader-clock cycles vs 100 MHz wall clock over this workgroup's life
      P.clkprobe[0] = clock64() - clk0;
      P.clkprobe[1] = wall_clock64() - wall0;
    }
    return;
  }
  // ---- prologue ----------------------------------------------------------------------------
  issue_a(0);
  issue_b(0);
  issue_a(1);
#pragma unroll
  for (int i = 1; i < SB; ++i) issue_b(i);
  wait_vmcnt<(SB - 1) * B_PW + A_PW>();  // strip 0 and tile 0 are the oldest
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(0, 0, 0, fa, fb);
  int abuf = 0;   // strip buffer of the tap being prefetched
  int st = 1;     // ring stage of the tap being prefetched (P % SB)
  int ist = 0;    // ring stage the next weight tile goes to ((P-1) % SB)

  // fragment read #idx of tap s (idx enumerates group-major: TM A-fragments then TN B-fragments)
  auto read_one = [&](int idx, const unsigned char* pa, const unsigned char* pb, int aswz) __attribute__((always_inline)) {
    const int gi = idx / (TM + TN), q = idx % (TM + TN);
    const int cl = 2 * (kg * GPW + gi) + lhalf;
    if (q < TM) na_[gi][q] = *(const uint4*)(pa + q * 32 * CHB + ((cl ^ aswz) * 16));
    else nb_[gi][q - TM] = *(const uint4*)(pb + (q - TM) * 32 * CHB + ((cl ^ bswz) * 16));
  };
  auto mma_one = [&](int idx) __attribute__((always_inline)) {
    const int gi = idx / (TM * TN), r = idx % (TM * TN);
    Mma<T>::run(fa[gi][r / TN], fb[gi][r % TN], acc[r / TN][r % TN]);
  };
  constexpr int NMMA = GPW * TM * TN, NRD = GPW * (TM + TN);

  // One tap.  After the barrier the MFMAs of tap P-1 (operands already in registers) start at once
  // and the DMA issue, the address math and tap P's ds_reads are slotted BETWEEN them, so the
  // matrix pipe is busy while everything else issues (pinned with sched_barrier: left alone hipcc
  // clusters the MFMAs and leaves the pipe idle during the ~150 cycles of issue in front of them).
  auto step = [&](auto s_tag, auto first_tag) __attribute__((always_inline)) {
    constexpr int s = decltype(s_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    wait_vmcnt<ring_wait_count<KS, SB, FIRST, B_PW, A_PW>(s)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (s == 0) abuf ^= 1;  // tap P opens a new filter row: its strip is in the other buffer
    const int arow = arow_base + s;
    const int aswz = ring_swz<CHB>(arow);
    const unsigned char* pa = sA + abuf * TR::A_BYTES + arow * CHB;
    const unsigned char* pb = pb_lane + st * TR::B_BYTES;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NMMA; ++m) {
      mma_one(m);
      if (m == 0) {
        if constexpr (s == 0) issue_a(abuf ^ 1);  // the previous row's buffer is free for the row after this one
        issue_b(ist);
      }
#pragma unroll
      for (int rd = m * NRD / NMMA; rd < (m + 1) * NRD / NMMA; ++rd) read_one(rd, pa, pb, aswz);
      __builtin_amdgcn_sched_barrier(0);
    }
    ist = (ist + 1 == SB) ? 0 : ist + 1;
    st = (st + 1 == SB) ? 0 : st + 1;
    rotate();
  };

  // first strip: taps 1..KS-1 (tap 0 was read in the prologue)
  step(std::integral_constant<int, 1>{}, std::true_type{});
  step(std::integral_constant<int, 2>{}, std::true_type{});
  if constexpr (KS == 7) {
    step(std::integral_constant<int, 3>{}, std::true_type{});
    step(std::integral_constant<int, 4>{}, std::true_type{});
    step(std::integral_constant<int, 5>{}, std::true_type{});
    step(std::integral_constant<int, 6>{}, std::true_type{});
  }
  for (int sc = 1; sc < nstrips; ++sc) {
    step(std::integral_constant<int, 0>{}, std::false_type{});
    step(std::integral_constant<int, 1>{}, std::false_type{});
    step(std::integral_constant<int, 2>{}, std::false_type{});
    if constexpr (KS == 7) {
      step(std::integral_constant<int, 3>{}, std::false_type{});
      step(std::integral_constant<int, 4>{}, std::false_type{});
      step(std::integral_constant<int, 5>{}, std::false_type{});
      step(std::integral_constant<int, 6>{}, std::false_type{});
    }
  }
  mma_all();  // last tap

  // drain the dummy DMAs before LDS is reused / the workgroup exits
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  if constexpr (POOL) conv_epilogue_pool<T, BM, BN, WM, WN, KSPLIT, TM, TN>(P, pr, acc, smem, kg, wm0, wn0, lane, img, pool_pair, pool_x0, n0);
  else conv_epilogue<T, BM, BN, WM, WN, KSPLIT, TM, TN>(P, pr, acc, smem, kg, wrem, wm0, wn0, lane, img, m0, n0);
}

#ifndef RTP_RING_NO_LAUNCHERS  // (tools/ring_probe.hip instantiates single kernels to inspect their ISA)
template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int CHB, int SB_, int MINW, bool SPEC, int VAR = 0, bool ILV = false, bool POOL = false>
static hipError_t ring_launch_spec(const ConvParams& P, int nprob, int N, hipStream_t stream);

template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int CHB, int SB_, int MINW = 1>
static hipError_t ring_launch_one(const ConvParams& P, int nprob, int N, hipStream_t stream) {
  if (P.pool) {  // fused 2x2 max pooling: the 3x3 trunk layers in front of a pooling layer (fp16 storage, 128-pixel tiles, 128-byte chunks)
    if constexpr (std::is_same<T, _Float16>::value && KS == 3 && BM == 128 && WM == 2 && CHB == 128) {
      if (P.q_from > 0) {
        if constexpr (((CHB / 32) / KSPLIT) % 2 == 0) return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 100, false, true>(P, nprob, N, stream);
        else return hipErrorInvalidValue;
      }
      return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 0, false, true>(P, nprob, N, stream);
    } else
      return hipErrorInvalidValue;
  }
  if (P.q_from > 0) {  // fp8-compensated layer: always the wave-specialised kernel, q instantiation
    if constexpr (std::is_same<T, _Float16>::value && ((CHB / 32) / KSPLIT) % 2 == 0) {
#ifdef RTP_EXPERIMENTS  // timing-only variants of the q kernel (RTP_RING_VAR=101/102/103; wrong results)
      if constexpr (BM == 128 && (BN == 64 || BN == 128)) {
        if (P.variant == 101) return P.ilv ? ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 101, true>(P, nprob, N, stream) : ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 101>(P, nprob, N, stream);
        if (P.variant == 102) return P.ilv ? ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 102, true>(P, nprob, N, stream) : ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 102>(P, nprob, N, stream);
        if (P.variant == 103) return P.ilv ? ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 103, true>(P, nprob, N, stream) : ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 103>(P, nprob, N, stream);
      }
#endif
      if (P.ilv) return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 100, true>(P, nprob, N, stream);
      return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 100>(P, nprob, N, stream);
    } else
      return hipErrorInvalidValue;
  }
  if (P.spec) {
    if constexpr (std::is_same<T, _Float16>::value) {
      if (P.ilv && P.variant == 0) return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 0, true>(P, nprob, N, stream);
    }
#ifdef RTP_EXPERIMENTS  // ablation variants (some compute wrong results on purpose: timing only) exist in librtpose_mi355x_exp.so only
    // experiment variants exist for the dominant plan only (fp16, 7x7, tile 128x64, 256-byte chunks)
    if constexpr (std::is_same<T, _Float16>::value && BM == 128 && BN == 64 && KS == 7 && CHB == 256) {
      switch (P.variant) {
        case 4: return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 4>(P, nprob, N, stream);
        case 17: return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 17>(P, nprob, N, stream);
        case 11: return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 11>(P, nprob, N, stream);
        case 12: return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 12>(P, nprob, N, stream);
        case 13: return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 13>(P, nprob, N, stream);
        case 2: return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 2>(P, nprob, N, stream);
        case 3: return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 3>(P, nprob, N, stream);
        case 15: return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 15>(P, nprob, N, stream);
        case 16: return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true, 16>(P, nprob, N, stream);
        default: break;
      }
    }
#endif
    return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, 1, true>(P, nprob, N, stream);
  }
#ifdef RTP_EXPERIMENTS
  return ring_launch_spec<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, MINW, false>(P, nprob, N, stream);  // RTP_RING_SPEC=0: every wave loads and multiplies
#else
  return hipErrorInvalidValue;  // the production plan always runs the wave-specialised kernels
#endif
}

template <typename T, int BM, int BN, int WM, int WN, int KSPLIT, int KS, int CHB, int SB_, int MINW, bool SPEC, int VAR, bool ILV, bool POOL>
static hipError_t ring_launch_spec(const ConvParams& P, int nprob, int N, hipStream_t stream) {
  using TR = RingTraits<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_>;
  auto kern = conv_ring_kernel<T, BM, BN, WM, WN, KSPLIT, KS, CHB, SB_, MINW, SPEC, VAR, ILV, POOL>;
  static std::atomic<unsigned> attr_mask{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_mask.load(std::memory_order_relaxed) & (1u << (dev & 31)))) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_mask.fetch_or(1u << (dev & 31), std::memory_order_relaxed);
  }
  dim3 grid(P.tiles_per_img * N * (P.CoutP / BN) * nprob);
  static const char* ldsmax = RTP_EXP_ENV("RTP_RING_LDS_MAX");
  const int lds = (ldsmax && ldsmax[0] == '1') ? 160 * 1024 : TR::LDS_BYTES;
  hipLaunchKernelGGL(kern, grid, dim3(SPEC ? 512 : 256), lds, stream, P);
  return hipGetLastError();
}

template <typename T, int KS>
static hipError_t ring_launch_cfg(int cfg, int chb, const ConvParams& P, int nprob, int N, hipStream_t stream) {
  if (chb == 256) {
    if (cfg == CFG_128x32) return ring_launch_one<T, 128, 32, 2, 1, 2, KS, 256, 4>(P, nprob, N, stream);
    if (cfg == CFG_128x64) {
#ifdef RTP_EXPERIMENTS
      if constexpr (KS == 7 && std::is_same<T, _Float16>::value) {  // ring-depth experiments (RTP_RING_SB=3/5) on the dominant plan
        if (P.ring_sb == 3) return ring_launch_one<T, 128, 64, 2, 1, 2, KS, 256, 3>(P, nprob, N, stream);
        if (P.ring_sb == 5) return ring_launch_one<T, 128, 64, 2, 1, 2, KS, 256, 5>(P, nprob, N, stream);
      }
#endif
      return ring_launch_one<T, 128, 64, 2, 1, 2, KS, 256, 4>(P, nprob, N, stream);
    }
    if (cfg == CFG_64x64) {
      if constexpr (KS == 7) {
        if (P.ring_sb != 4) return ring_launch_one<T, 64, 64, 1, 1, 4, KS, 256, 6>(P, nprob, N, stream);
      }
      return ring_launch_one<T, 64, 64, 1, 1, 4, KS, 256, 4>(P, nprob, N, stream);
    }
    return hipErrorInvalidValue;
  }
  if (chb != 128) return hipErrorInvalidValue;
  switch (cfg) {
    case CFG_128x128: return ring_launch_one<T, 128, 128, 2, 2, 1, KS, 128, 4>(P, nprob, N, stream);
    case CFG_64x128: return ring_launch_one<T, 64, 128, 1, 2, 2, KS, 128, 4>(P, nprob, N, stream);
    case CFG_64x64:
      // fp8-compensated layers need >= 64 bytes of K per wave and chunk: waves 2 (M) x 1 (N) x 2 (K) instead of 4-way K split
      if (P.q_from > 0) return ring_launch_one<T, 64, 64, 2, 1, 2, KS, 128, 4>(P, nprob, N, stream);
      // 56 KiB of LDS and <= 128 registers: two workgroups (of different frames) share a CU
      return ring_launch_one<T, 64, 64, 1, 1, 4, KS, 128, 4, 2>(P, nprob, N, stream);
    case CFG_128x64: return ring_launch_one<T, 128, 64, 2, 1, 2, KS, 128, 4>(P, nprob, N, stream);
#ifdef RTP_EXPERIMENTS
    case CFG_256x64:
      // 256 pixels x 64 output channels per workgroup, 128-byte chunks: the K loop of the 128x64 / 256-byte tile (16 MFMAs per wave
      // and step) over twice as many steps — one pipeline fill and one epilogue per 98 K steps instead of 49.  fp16 7x7 layers only.
      // Measured (profiles/r05_experiments.txt): faster alone, no gain in the pipeline: experiments build only.
      if constexpr (std::is_same<T, _Float16>::value && KS == 7) return ring_launch_one<T, 256, 64, 2, 1, 2, KS, 128, 4>(P, nprob, N, stream);
      else return hipErrorInvalidValue;
#endif
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

#endif  // RTP_RING_NO_LAUNCHERS

}  // namespace rtp
