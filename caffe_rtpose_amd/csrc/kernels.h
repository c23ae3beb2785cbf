// kernels.h — device-side contracts shared by the HIP kernels and the engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Experiment knobs (tile overrides, ablation variants of the ring kernel — some of which compute WRONG results on purpose, stream
// plans, diagnostic probes) exist only in the second build target librtpose_mi355x_exp.so (-DRTP_EXPERIMENTS, used by tools/):
// in the production library the macro is a null pointer and the knob's name is not even a string in the binary
// (tests/test_host_cpu.py greps for them).  The production library reads ONE environment variable: RTP_EXEC=eager|graph.
#ifdef RTP_EXPERIMENTS
#define RTP_EXP_ENV(name) getenv(name)
#else
#define RTP_EXP_ENV(name) ((const char*)nullptr)
#endif

namespace rtp {

// Kernel residency stamps (rtp_stamp_probe, bench.py `idle_frac_kernel_stamps`): when a launch carries a slot pointer, thread 0 of every
// workgroup folds the 100 MHz wall clock (s_memrealtime: one clock for all XCDs, tools/clock_sync_probe.hip) into the slot when the
// workgroup starts and when it ends: slot[0] = max(~start) = ~(first workgroup's start), slot[1] = max(end) = the last workgroup's end.
// Null pointer (every launch outside a probe run): one scalar compare per workgroup.
#if defined(__HIPCC__)
struct KStamp {
  unsigned long long* s;
  __device__ __forceinline__ explicit KStamp(unsigned long long* p) : s(p) {
    if (s && threadIdx.x == 0 && threadIdx.y == 0) atomicMax(s, ~wall_clock64());
  }
  __device__ __forceinline__ ~KStamp() {
    if (s && threadIdx.x == 0 && threadIdx.y == 0) atomicMax(s + 1, wall_clock64());
  }
};
#endif

// ---------------------------------------------------------------------------------------
// Activation layout in HBM: NHWC with a zero spatial halo, one geometry per resolution level.
//   element (n, y, x, c) lives at  base + ((n*Hp + y+halo)*Wp + x+halo)*Cp + c
// Hp = H+2*halo, Wp = W+2*halo.  Halo pixels and pad channels are zeroed once at allocation
// and never written, so a k x k convolution with pad <= halo needs no bounds test: tap (r,s)
// of output pixel p (flat padded index) reads flat pixel p + (r-pad)*Wp + (s-pad).
// ---------------------------------------------------------------------------------------
struct Geom {
  int N, H, W, halo, Hp, Wp;
  long img_pix;  // Hp*Wp
};

#define RTP_MAX_DST 6
struct ConvDst {
  void* base;   // element pointer of (n=0, padded pixel 0, channel 0)
  int cstride;  // channels per pixel of the destination tensor (2*Cp when it carries a lo block)
  int coff;     // first channel this conv writes
  int lo_off;   // 0, or the channel offset of the tensor's lo block: channel c also gets lo = T(v - float(T(v))) at lo_off + c
  int q_off;    // 0, or the ELEMENT offset of the tensor's q block (fp8 error-compensation operands): per 64-channel group g,
                // bytes [128g, 128g+64) = fp8(lo * 2^12), bytes [128g+64, 128g+128) = fp8(T(v) * 2^2)   (e4m3, clamped to +-448)
};

struct ConvProblem {
  const void* in;     // element pointer of (n=0, padded pixel 0, channel 0)
  const void* w;      // packed weights [tap][chunk][CoutP][ROWB bytes]
  const float* bias;  // [CoutP] fp32
  ConvDst dst[RTP_MAX_DST];
  int ndst;
  float* out_nchw;  // optional fp32 planar output [N][out_C][H][W]
  int out_C, out_coff;
  int Cout;
};

struct ConvParams {
  ConvProblem prob[2];  // blockIdx.z selects (the L1 / L2 branch pair shares every shape)
  int H, W, Wp, halo;
  long img_pix;
  int in_cstride;  // channels per pixel of the input tensor (2*Cp when it carries a lo block)
  int nchunk;      // K chunks per filter row.  Plain layer: Cin_p*sizeof(T)/ROWB.  Split-precision layer: the passes
                   // [a_hi x W_hi][a_lo x W_hi][a_hi x W_lo] are further chunks of the same K loop (weights are
                   // packed in that order); the ACTIVATION chunk of virtual chunk v is v, or v - wrap_at from wrap_at on
  int wrap_at;     // 0 = no wrap (the chunks are contiguous in the input tensor)
  int last_phys;   // physical chunk index of the last virtual chunk (nchunk-1 without a wrap)
  // fp8-compensated split layers (ring kernels): virtual chunks [q_from, nchunk) are q chunks — the same channel groups once
  // more, as fp8 pairs (a_lo8 x W8, a8 x W_lo8) multiplied by v_mfma_scale_f32_32x32x64_f8f6f4 at twice the fp16 rate
  int q_from;      // 0 = none
  int wq_exp;      // the weights' q chunks hold fp8(W * 2^wq_exp) and fp8(W_lo * 2^(wq_exp + 11)); one exponent for both branches
                   // of a paired launch (a per-problem field read through prob[blockIdx-dependent] made hipcc keep the DMA base
                   // pointer in VGPRs)
  int jump_delta;  // bytes from the last hi chunk to the first q chunk of the input tensor (instead of + CHB)
  int row_back;    // byte offset of a filter row's last virtual chunk relative to its first (ring kernels)
  int CoutP;       // Cout rounded up to a multiple of BN
  int tiles_per_img;
  int nimg;    // images (scales) in the batch
  int relu;
  int rotate;  // ring kernel: rotate the filter-row order per tile (speed only)
  int xcdmap;  // 1: contiguous logical range per XCD, pixel tiles fastest (decode_block), 2: the same with N tiles fastest, 0: dispatch order
  int ring_sb; // ring depth request (4 or 6) where both are instantiated
  int spec;    // ring kernel: 1 = wave-specialised variant (4 DMA waves + 4 MFMA waves)
  int variant; // ring kernel experiments (env RTP_RING_VAR; see conv_ring.hip), 0 = production
  int ilv;     // ring kernel (wave-specialised, fp16): interleaved A-fragment rows, taps 1.. of a strip shift registers instead of re-reading LDS
  // fused 2x2 max pooling (ring kernels, conv_ring.hip POOL): the M tile is 2 image rows x BM/2 pixels, the epilogue writes only the
  // pooled tensor.  pool_wq = W + pad rounded up to even (pitch of the tile walk, so that every tile starts on an even x); pool_* =
  // geometry of the pooled (next) level
  int pool, pool_wq, pool_Wp, pool_halo;
  long pool_img_pix;
  int diag;    // experiments build (env RTP_EPI_DIAG; timing only, wrong results): 1 = the epilogue computes but does not store, 2 = no epilogue at all
  unsigned long long* stamp;     // kernel residency slot (KStamp) or null
  unsigned long long* clkprobe;  // diagnostics: {shader clock cycles, wall clock ticks} of workgroup 0 (spec ring kernels)
};

// which tile configuration a conv launch uses
enum ConvCfg { CFG_128x128 = 0, CFG_64x128 = 1, CFG_64x64 = 2, CFG_128x64 = 3, CFG_128x32 = 4, CFG_256x64 = 5, CFG_COUNT = 6 };

struct ConvCfgInfo { int BM, BN; };
inline ConvCfgInfo conv_cfg_info(int cfg) {
  switch (cfg) {
    case CFG_128x128: return {128, 128};
    case CFG_64x128: return {64, 128};
    case CFG_64x64: return {64, 64};
    case CFG_128x32: return {128, 32};
    case CFG_256x64: return {256, 64};   // 7x7 layers at batches of >= 4 images: twice the pixels per workgroup (one fill + epilogue per 98 K steps)
    default: return {128, 64};
  }
}

// prec: 0 = fp16 (MFMA f16, fp32 accumulate), 1 = fp32 (exact-f32 MFMA).  rowb: 64 or 128.
// Returns hipSuccess or the launch error.
hipError_t launch_conv(int prec, int cfg, int ks, int rowb, const ConvParams& P, int nprob, int N,
                       hipStream_t stream);

// LDS-DMA ring variant for k in {3,7}: chb = channel bytes per step (128, or 256 with the 64x64
// tile); weights packed with the matching 16-byte-chunk swizzle (see conv_ring.hip).
hipError_t launch_conv_ring(int prec, int cfg, int ks, int chb, const ConvParams& P, int nprob, int N,
                            hipStream_t stream);
// Two chained 1x1 convolutions in one launch (conv_pw2.hip): Y = W2 * act(W1 * X + b1) + b2, fp16 MFMA, Cin = 128 channels,
// middle layer a multiple of 128 channels, Cout <= 64.  P2 describes the SECOND convolution (destinations, bias, geometry;
// prob[].w = W2 packed [chunk][part][64][128]); the fields below the first one.
struct Pw2Params {
  ConvParams P2;
  const void* x_in[2];     // input tensor of the first convolution (element pointer of padded pixel 0)
  int x_cstride, x_lo_off; // channels per pixel; 0 or the offset of the input's lo block (split precision)
  const void* w1[2];       // W1 packed [chunk][part (hi, lo)][128][128]
  const float* b1[2];
  ConvDst mid[2];          // the middle layer's own blob tensor (base may be null)
  int c1_chunks;           // middle channels / 128
  int relu1;
  int split_w1, split_w2, h_lo;  // split precision: lo weights of either layer; lo part of the middle activations
};
hipError_t launch_conv_pw2(const Pw2Params& Q, int nprob, int nimg, hipStream_t stream);

inline int conv_ring_swz(int chb, int row) { return chb == 256 ? (row & 15) : ((row >> 1) & 7); }

// NCHW fp32 [N][3][H][W] -> level-0 tensor with 32 channels = 3x3 im2col of the image
// (channel (r*3+s)*3+c = in[c][y+r-1][x+s-1], zero outside; channels 27..31 zero).
hipError_t launch_pack_input(int prec, const float* in_nchw, void* out, Geom g, int Cp, hipStream_t stream);
// conv1_1 straight from the fp32 NCHW input (conv_first.hip): fp16 storage, 64 output channels, no split parts
struct FirstParams {
  const float* in;      // [N][3][H][W]
  Geom g;               // level-0 geometry (N = images in this launch)
  const uint4* wfrag;   // [tile 2][K-step 2][lane 64] A-operand fragments
  const float* bias;    // 64 floats in the kernel's channel order == reference order
  _Float16* out;        // halo'd NHWC destination, padded pixel 0
  int Cp;               // its channels per pixel (elements)
  int relu;
  unsigned long long* stamp;  // kernel residency slot (KStamp) or null
};
hipError_t launch_conv_first(const FirstParams& Q, hipStream_t stream);
int conv_first_channel_of_row(int i);
// 2x2 stride-2 MAX pooling between two halo'd NHWC tensors (pooling_layer.cpp:140-180).
// lo_i / lo_o: 0, or the channel offset of the lo block of a split-precision tensor — the pooled element keeps ITS lo
// part (the maximum of hi + lo, not two independent maxima).
// q_i / q_o: 0, or the element offset of the q block (fp8 compensation operands, ConvDst::q_off) — copied from the selected element.
hipError_t launch_maxpool(int prec, const void* in, Geom gi, int Cpi, void* out, Geom go, int Cpo, int C, int lo_i, int lo_o, int q_i, int q_o,
                          hipStream_t stream);
// halo'd NHWC (T) -> planar fp32 [N][C][H][W] (debug tap; channel map: out c reads in chmap[c]).
// lo_off != 0: the tensor carries a lo block, the exported value is hi + lo.
// q_off != 0 (and no lo block): the exported value is hi + fp8 lo part / 2^12.
hipError_t launch_export(int prec, const void* in, Geom g, int Cp, const int* chmap_dev, int C, int lo_off, int q_off, float* out,
                         hipStream_t stream);

// ---------------------------------------------------------------------------------------
// Pre-processing on the device (row a1): see preproc.hip
// ---------------------------------------------------------------------------------------
// warp: BicubicTab_i of OpenCV's initInterTab2D, [fy][fx][k1*4+k2] 15-bit fixed-point weights (device pointer, 32 KiB)
struct AreaScale {                    // cv::resize(INTER_AREA) tables of one pyramid level (device pointers)
  int tw, th, identity;
  int fast_x, fast_y;                 // > 0: integer scale on both axes (resizeAreaFast_): block sums instead of the tables
  const int* xstart; const int* xsi; const float* xalpha;   // entries of dst column x: [xstart[x], xstart[x+1])
  const int* ystart; const int* ysi; const float* yalpha;
  // linear != 0: an axis of this level is ENLARGED — cv::resize(INTER_AREA) then runs its bilinear kernel with area-mode coefficients
  // (preprocess.cpp linear_area_tab): lx [tw][4] / ly [th][4] = {source index, its clipped neighbour, 11-bit weight, 11-bit weight}
  int linear;
  const int* lx; const int* ly;
};
hipError_t launch_warp(unsigned long long* stamp, const unsigned char* src, int sw, int sh, double inv, const short* tab2d, unsigned char* dst, int dw, int dh,
                       hipStream_t stream);
hipError_t launch_area_pad(unsigned long long* stamp, const unsigned char* disp, int dw, int dh, const AreaScale* scales, int nscales, float* out, int net_w, int net_h,
                           hipStream_t stream);

// ---------------------------------------------------------------------------------------
// Renderer (row 8f-3): render.hip
// ---------------------------------------------------------------------------------------
struct RenderParams {
  const unsigned char* src;  // display image, u8 BGR HWC (device)
  unsigned char* dst;        // rendered image, same layout
  int w, h;
  const float* poses;        // joints [max_people][num_parts][3], display coordinates (device)
  const int* num_people;     // device
  float* tab;                // scratch, render_tab_floats(max_people)
  int model, googly, max_people;
};
hipError_t launch_render(const RenderParams& p, hipStream_t stream);
size_t render_tab_floats(int max_people);
// the --part_to_show views of render() (rtpose.cpp:270-299): one heat-map channel, all part maps, or PAF channels over the frame
struct RenderViewParams {
  const unsigned char* src;  // display image, u8 BGR HWC (device)
  unsigned char* dst;
  int w, h;
  const float* maps;         // net-resolution maps [C][net_h][net_w] (what the Nms layer reads), device
  int net_w, net_h;
  int model, part_to_show;   // part_to_show > 0 as FLAGS_part_to_show
};
hipError_t launch_render_view(const RenderViewParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------
// Post-processing (bit-exact restatements of the reference's CUDA kernels / host loop).
// ---------------------------------------------------------------------------------------
struct ResizeParams {
  const float* src;  // [num][C][h][w]
  float* dst;        // [C][th][tw]
  int num, C, h, w, tw, th;
  float start_scale, scale_gap;
};
hipError_t launch_resize(const ResizeParams& p, hipStream_t stream);

struct NmsParams {
  const float* src;  // resized map [src_planes][H][W]
  float* peaks;      // [num_parts][max_peaks+1][3]  (in/out)
  int* strip_count;  // [num_parts][nstrips]
  int* strip_list;   // [num_parts][nstrips][max_peaks]  flat pixel index of the first max_peaks maxima of the strip
  int src_planes, H, W, num_parts, max_peaks, nstrips, strip_rows;
  float threshold;
  unsigned long long* probe;  // diagnostics (RTP_NMS_PROBE): wall-clock stamps of the phases of one strip workgroup, [0] = count
  unsigned long long* stamp;  // kernel residency slots (KStamp) or null: [0,1] strip kernel, [2,3] write kernel
  int* clear_flag;            // production chain: the connect kernels' people counter / error flag, reset here (the write kernel runs right in
                              // front of them on the same stream) instead of by a 4-byte fill launch of its own between the two
};
hipError_t launch_nms(const NmsParams& p, hipStream_t stream);
// Production path: the same peaks WITHOUT materialising the resized map — each strip workgroup
// evaluates ImResize for its rows (+1 halo row) in LDS, the centroid windows are evaluated on demand.
// `p.src` is ignored; `r.dst` is ignored.
hipError_t launch_nms_fused(const NmsParams& p, const ResizeParams& r, hipStream_t stream);

struct ConnectParams {
  unsigned long long* stamp;  // kernel residency slots (KStamp) or null: [0,1] pairs, [2,3] match, [4,5] assemble
  int counter_cleared;  // 1: *num_people was reset by the NMS write kernel in front of this chain (NmsParams::clear_flag)
  int* tickets;         // [num_limbs + 1] zeros (device): non-null = the production chain runs as ONE launch (connect_chain_kernel): per-limb tickets of
                        // the pair workgroups + one of the limbs; the kernel leaves them zero again
  const float* heat;   // resized map [C][net_h][net_w]
  const float* peaks;  // [num_parts][max_peaks+1][3]
  float* joints;       // [max_people][num_parts][3]
  int* num_people;     // [1]  (<0: error code)
  // scratch
  float* cand_score;   // [num_limbs][max_peaks*max_peaks]
  int* cand_ij;        // [num_limbs][max_peaks*max_peaks]  (i<<16|j), raster (i,j) order, compacted
  int* cand_count;     // [num_limbs]
  int* cand_blk;       // [num_limbs][ceil(max_peaks^2 / 256)] survivors per block of 256 pairs
  int* conn;           // [num_limbs][max_peaks][2]  (indexA, indexB) flat peak-score indices
  float* conn_score;   // [num_limbs][max_peaks]
  int* conn_count;     // [num_limbs]
  int* subset_idx;     // [max_rows][num_parts]
  double* subset_score;  // [max_rows]
  int* subset_cnt;     // [max_rows]
  int max_rows;
  int assemble_preload;  // set by launch_connect: the assembly kernel copies its inputs to LDS first
  int pairs_full;        // experiments (RTP_PAIRS_FULL=1): the pair kernel evaluates all 10 samples of every pair (round 5's behaviour; same results)
  int diag_stages;       // diagnostics (RTP_DIAG_SKIP_POST): 0 = all kernels, 1 = pairs only, 2 = pairs + match, 3 = all
  int model;           // 0 COCO_18, 1 MPI_15
  int num_parts, num_limbs, max_peaks;
  int net_w, net_h, disp_w, disp_h;
  float inter_threshold;
  int inter_min_above;
  int min_subset_cnt;
  float min_subset_score;
  int max_people;
};
hipError_t launch_connect(const ConnectParams& p, hipStream_t stream);
// Production path: PAF samples are evaluated from the low-res maps on demand (`p.heat` ignored).
hipError_t launch_connect_fused(const ConnectParams& p, const ResizeParams& r, hipStream_t stream);

}  // namespace rtp
