// postproc.hip — ImResize, Nms and connectLimbs* as wavefront kernels, BIT-EXACT restatements.
//
// Compiled with -ffp-contract=off: every float/double operation below rounds exactly where the
// reference's source rounds (no FMA fusion), so given the same low-res heat maps these kernels
// reproduce the oracle's resized map, peak list and joints bit for bit.  f32 divide and sqrt are
// correctly rounded on gfx950 under hipcc's default flags.
//
//   imresize_cubic_kernel    src/caffe/cpm/layers/imresize_layer.cu:9-18,99-155
//   nms_register/scan/write  src/caffe/cpm/layers/nms_layer.cu:15-113,117-184
//   connectLimbs(COCO)       examples/rtpose/rtpose.cpp:549-751, 808-1076
//
// Structure (MI355X-first, not the reference's launch pattern):
//   resize : ONE launch over (x, y, c) instead of 57 per-channel launches.
//   nms    : the reference does per part {flag kernel, thrust::exclusive_scan over 241k ints,
//            write kernel} = 54 launches + 18 scans.  Here: strip kernel (8 image rows per
//            workgroup; wave ballots + popcounts give each maximum its raster ordinal inside the
//            strip, and only the first max_peaks pixel indices per strip are kept) and a write
//            kernel that scans the <=46 strip totals and does the 7x7 centroid refine.  Raster
//            order — which decides WHICH peaks survive the max_peaks cap — is preserved exactly.
//   connect: score kernel (one workgroup per limb: PAF line integrals for all (i,j) pairs in
//            raster order, ballot-compacted; then lane 0 runs the libstdc++-exact sort and the
//            greedy assignment) and a single-wavefront assembly kernel that keeps the person
//            table in LDS and parallelises the row searches over the 64 lanes.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "kernels.h"
#include "stdsort_replica.h"

// No packed-f32 VALU (v_pk_add/mul/fma_f32) in these kernels: this file is built with
// -fno-slp-vectorize -fno-vectorize (csrc/Makefile BITEXACT; `make check-nopk` and a CPU test count
// the instructions).  hipcc's SLP vectoriser pairs the x/y halves of the scalar float math below
// into packed ops; with that, nms_fused_write_kernel returned run-to-run different bits for the SAME
// inputs (whole 16-lane groups, inside divergent code) whenever MFMA-heavy convolution workgroups
// of another frame shared its CU, and never on an idle chip (tools/race_probe_post.py; in-kernel
// re-evaluation self-check).  Scalar f32 ops are bit-identical by definition, and with them 0
// mismatches in the same stress.  (A per-function target("no-packed-fp32-ops") also removes them
// but blocks inlining of the HIP header/ockl functions: __syncthreads became a call.)

namespace rtp {

// ---------------------------------------------------------------------------------------
// ImResize
// ---------------------------------------------------------------------------------------
// imresize_layer.cu:14-17 with C++'s usual arithmetic conversions spelled out, in two halves: the three sub-expressions that do
// not depend on the fraction (8 consecutive outputs along an axis share them), and the evaluation at a fraction.  Same
// operations in the same order as the one-piece expression (no contraction in this file): cubic_interp IS this pair.
struct CubicCoef { float a; double b; float c; float v1; };
__device__ __forceinline__ CubicCoef cubic_coef(float v0, float v1, float v2, float v3) {
  CubicCoef k;
  k.a = (-0.5f * v0 + 1.5f * v1 - 1.5f * v2 + 0.5f * v3);
  k.b = ((double)(v0 - 2.5f * v1) + 2.0 * (double)v2 - 0.5 * (double)v3);
  k.c = (-0.5f * v0 + 0.5f * v2);
  k.v1 = v1;
  return k;
}
__device__ __forceinline__ float cubic_eval(const CubicCoef& k, float dx) {
  const float t1 = k.a * dx * dx * dx;
  const double t2 = k.b * (double)dx * (double)dx;
  const float t3 = k.c * dx;
  return (float)((((double)t1 + t2) + (double)t3) + (double)k.v1);
}
__device__ __forceinline__ float cubic_interp(float v0, float v1, float v2, float v3, float dx) {
  return cubic_eval(cubic_coef(v0, v1, v2, v3), dx);
}

// Geometry of scale n (imresize_layer.cu:99-131)
struct ScaleGeo {
  int padw, padh, ow, oh, rw;
  float offset_x, offset_y, fx, fy;  // fx = (float)ow / tw
};
__device__ __forceinline__ ScaleGeo scale_geo(const ResizeParams& p, int n) {
  ScaleGeo g;
  g.padw = (int)floorf((float)(p.w / 2) * (1 - p.start_scale + n * p.scale_gap));
  g.padh = (int)floorf((float)(p.h / 2) * (1 - p.start_scale + n * p.scale_gap));
  g.ow = p.w - 2 * g.padw;
  g.oh = p.h - 2 * g.padh;
  g.offset_x = (float)((double)(p.tw / (float)g.ow / 2) - 0.5);
  g.offset_y = (float)((double)(p.th / (float)g.oh / 2) - 0.5);
  g.rw = g.ow + 2 * g.padw;
  g.fx = (float)g.ow / p.tw;
  g.fy = (float)g.oh / p.th;
  return g;
}
// neighbour indices (already padded) and fraction along one axis
__device__ __forceinline__ float axis_nb(int x, float offset, float f, int osize, int pad, int nb[4]) {
  const float x_on = (x - offset) * f;
  int xn1 = (int)((double)x_on + 1e-5);
  xn1 = (xn1 < 0) ? 0 : xn1;
  const float dx = x_on - xn1;
  nb[0] = ((xn1 - 1 < 0) ? xn1 : (xn1 - 1)) + pad;
  const int xn2 = (xn1 + 1 >= osize) ? (osize - 1) : (xn1 + 1);
  nb[3] = ((xn2 + 1 >= osize) ? (osize - 1) : (xn2 + 1)) + pad;
  nb[1] = xn1 + pad;
  nb[2] = xn2 + pad;
  return dx;
}
// ONE output of the resized map, computed exactly as resize_kernel computes it (same operations in
// the same order per scale, same accumulation over scales).
// chan0: the channel's map at scale 0; scale_stride: floats between consecutive scales of that channel;
// geo: the per-scale geometry (scale_geo), computed once per workgroup
__device__ __forceinline__ float resized_from(const ScaleGeo* geo, int num, const float* chan0, long scale_stride, int y, int x) {
  float sum = 0.f;
  for (int n = 0; n < num; ++n) {
    const ScaleGeo g = geo[n];
    const float* sp = chan0 + (long)n * scale_stride;
    int xn[4], yn[4];
    const float dx = axis_nb(x, g.offset_x, g.fx, g.ow, g.padw, xn);
    const float dy = axis_nb(y, g.offset_y, g.fy, g.oh, g.padh, yn);
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* rr = sp + yn[i] * g.rw;
      t[i] = cubic_interp(rr[xn[0]], rr[xn[1]], rr[xn[2]], rr[xn[3]], dx);
    }
    const float d = cubic_interp(t[0], t[1], t[2], t[3], dy);
    sum = sum + d;
  }
  return sum / num;
}
// two channels at the same output pixel (the x and y PAF of a limb): the neighbourhood is shared
__device__ __forceinline__ void resized_pair(const ScaleGeo* geo, int num, const float* cx0, const float* cy0, long scale_stride, int y, int x,
                                             float* vx, float* vy) {
  float sx = 0.f, sy = 0.f;
  for (int n = 0; n < num; ++n) {
    const ScaleGeo g = geo[n];
    const float* spx = cx0 + (long)n * scale_stride;
    const float* spy = cy0 + (long)n * scale_stride;
    int xn[4], yn[4];
    const float dx = axis_nb(x, g.offset_x, g.fx, g.ow, g.padw, xn);
    const float dy = axis_nb(y, g.offset_y, g.fy, g.oh, g.padh, yn);
    float tx[4], ty[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ro = yn[i] * g.rw;
      tx[i] = cubic_interp(spx[ro + xn[0]], spx[ro + xn[1]], spx[ro + xn[2]], spx[ro + xn[3]], dx);
      ty[i] = cubic_interp(spy[ro + xn[0]], spy[ro + xn[1]], spy[ro + xn[2]], spy[ro + xn[3]], dx);
    }
    sx = sx + cubic_interp(tx[0], tx[1], tx[2], tx[3], dy);
    sy = sy + cubic_interp(ty[0], ty[1], ty[2], ty[3], dy);
  }
  *vx = sx / num;
  *vy = sy / num;
}
#define RTP_MAX_SCALES 16
// every thread calls; geo[] (LDS) is valid after the barrier inside
__device__ __forceinline__ void fill_scale_geo(ScaleGeo* geo, const ResizeParams& p) {
  if ((int)threadIdx.x < p.num) geo[threadIdx.x] = scale_geo(p, threadIdx.x);
  __syncthreads();
}
__device__ __forceinline__ float resized_at(const ScaleGeo* geo, const ResizeParams& p, int c, int y, int x) {
  const long plane = (long)p.h * p.w;
  return resized_from(geo, p.num, p.src + (long)c * plane, (long)p.C * plane, y, x);
}

// One thread = an 8x8 block of outputs of one channel: x0 = 8k-4, y0 = 8m-4.  For the full-size
// scale such a block shares ONE 4x4 low-res neighbourhood, so the 4 row interpolations t[i] of an
// output column are computed once and reused by the 8 output rows (1.5 cubic evaluations per output
// instead of 5) and the 16 neighbours are loaded once per block.  Per-output arithmetic and
// rounding are exactly the reference kernel's; whenever the neighbourhood of an output differs
// from the previous one (other scales, borders) it is simply recomputed.
__global__ __launch_bounds__(128) void resize_kernel(ResizeParams p) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;  // 8-wide column strip
  const int x0 = 8 * k - 4;
  const int y0 = 8 * (int)blockIdx.y - 4;
  const int c = blockIdx.z;
  if (x0 >= p.tw) return;
  const long plane = (long)p.h * p.w;
  const float* src_c = p.src + (long)c * plane;
  const long src_offset = (long)p.C * plane;
  float sum[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[r][j] = 0.f;
  for (int n = 0; n < p.num; ++n) {
    const int padw = (int)floorf((float)(p.w / 2) * (1 - p.start_scale + n * p.scale_gap));
    const int padh = (int)floorf((float)(p.h / 2) * (1 - p.start_scale + n * p.scale_gap));
    const int ow = p.w - 2 * padw, oh = p.h - 2 * padh;
    const float* sp = src_c + n * src_offset;
    const float offset_x = (float)((double)(p.tw / (float)ow / 2) - 0.5);
    const float offset_y = (float)((double)(p.th / (float)oh / 2) - 0.5);
    const int rw = ow + 2 * padw;
    float t[4][8];   // row interpolations of the current y-neighbourhood, per output column
    int prev_y = -1;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int y = y0 + r;
      if (y < 0 || y >= p.th) continue;
      const float y_on = (y - offset_y) * ((float)oh / p.th);
      int yn1 = (int)((double)y_on + 1e-5);
      yn1 = (yn1 < 0) ? 0 : yn1;
      const float dy = y_on - yn1;
      if (yn1 != prev_y) {
        prev_y = yn1;
        int yn[4];
        yn[0] = ((yn1 - 1 < 0) ? yn1 : (yn1 - 1)) + padh;
        yn[2] = (yn1 + 1 >= oh) ? (oh - 1) : (yn1 + 1);
        yn[3] = ((yn[2] + 1 >= oh) ? (oh - 1) : (yn[2] + 1)) + padh;
        yn[1] = yn1 + padh;
        yn[2] += padh;
        float v[4][4];
        int prev_x = -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int x = x0 + j;
          if (x < 0 || x >= p.tw) continue;
          const float x_on = (x - offset_x) * ((float)ow / p.tw);
          int xn1 = (int)((double)x_on + 1e-5);
          xn1 = (xn1 < 0) ? 0 : xn1;
          const float dx = x_on - xn1;
          if (xn1 != prev_x) {
            prev_x = xn1;
            const int xn0 = ((xn1 - 1 < 0) ? xn1 : (xn1 - 1)) + padw;
            const int xn2 = (xn1 + 1 >= ow) ? (ow - 1) : (xn1 + 1);
            const int xn3 = ((xn2 + 1 >= ow) ? (ow - 1) : (xn2 + 1)) + padw;
            const int xa = xn1 + padw, xb = xn2 + padw;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float* rr = sp + yn[i] * rw;
              v[i][0] = rr[xn0]; v[i][1] = rr[xa]; v[i][2] = rr[xb]; v[i][3] = rr[xn3];
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) t[i][j] = cubic_interp(v[i][0], v[i][1], v[i][2], v[i][3], dx);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int x = x0 + j;
        if (x < 0 || x >= p.tw) continue;
        const float d = cubic_interp(t[0][j], t[1][j], t[2][j], t[3][j], dy);
        sum[r][j] = sum[r][j] + d;
      }
    }
  }
  float* o = p.dst + (long)c * p.th * p.tw;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int y = y0 + r;
    if (y < 0 || y >= p.th) continue;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int x = x0 + j;
      if (x >= 0 && x < p.tw) o[(long)y * p.tw + x] = sum[r][j] / p.num;
    }
  }
}

hipError_t launch_resize(const ResizeParams& p, hipStream_t stream) {
  const int strips = (p.tw + 4 + 7) / 8;  // x0 = 8k-4 covers x in [-4, tw)
  const int bands = (p.th + 4 + 7) / 8;   // y0 = 8m-4
  dim3 grid((strips + 127) / 128, bands, p.C);
  hipLaunchKernelGGL(resize_kernel, grid, dim3(128), 0, stream, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// NMS
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int nms_flag(const float* s, int x, int y, int W, int H, float thr) {
  // nms_register_kernel, nms_layer.cu:15-46
  if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
    const float v = s[y * W + x];
    if (v > thr) {
      const float top = s[(y - 1) * W + x], bottom = s[(y + 1) * W + x];
      const float left = s[y * W + x - 1], right = s[y * W + x + 1];
      const float tl = s[(y - 1) * W + x - 1], tr = s[(y - 1) * W + x + 1];
      const float bl = s[(y + 1) * W + x - 1], br = s[(y + 1) * W + x + 1];
      if (v > top && v > bottom && v > left && v > right && v > tl && v > bl && v > br && v > tr) return 1;
    }
  }
  return 0;
}

__global__ __launch_bounds__(256) void nms_strip_kernel(NmsParams p) {
  __shared__ int wave_cnt[4];
  __shared__ int running;
  const int strip = blockIdx.x, part = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* s = p.src + (long)part * p.H * p.W;
  const int y0 = strip * p.strip_rows;
  const int y1 = min(y0 + p.strip_rows, p.H);
  const int npix = (y1 - y0) * p.W;
  const int pix0 = y0 * p.W;
  int* list = p.strip_list + ((long)part * p.nstrips + strip) * p.max_peaks;
  if (tid == 0) running = 0;
  __syncthreads();
  for (int base = 0; base < npix; base += 256) {
    const int q = base + tid;
    int f = 0;
    if (q < npix) {
      const int g = pix0 + q;
      f = nms_flag(s, g % p.W, g / p.W, p.W, p.H, p.threshold);
    }
    const unsigned long long bal = __ballot(f);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int before = running;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    const int ord = before + __popcll(bal & ((1ull << lane) - 1ull));
    if (f && ord < p.max_peaks) list[ord] = pix0 + q;
    __syncthreads();
    if (tid == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) p.strip_count[part * p.nstrips + strip] = running;
}

__global__ __launch_bounds__(256) void nms_write_kernel(NmsParams p) {
  extern __shared__ int prefix[];  // [nstrips+1]
  const int part = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < p.nstrips; ++i) {
      prefix[i] = run;
      run += p.strip_count[part * p.nstrips + i];
    }
    prefix[p.nstrips] = run;
  }
  __syncthreads();
  const int total = prefix[p.nstrips];
  const int W = p.W;
  const long offset = (long)p.H * p.W;
  const float* s = p.src + (long)part * offset;
  float* dst = p.peaks + (long)part * (p.max_peaks + 1) * 3;
  const int n = total < p.max_peaks ? total : p.max_peaks;
  for (int e = tid; e < n; e += blockDim.x) {
    int st = 0;
    while (prefix[st + 1] <= e) ++st;  // strip holding ordinal e
    const int g = p.strip_list[((long)part * p.nstrips + st) * p.max_peaks + (e - prefix[st])];
    const int px = g % W, py = g / W;
    // writeResultKernel, nms_layer.cu:70-105 (window bound on y is `width`; index 0 excluded)
    float x_acc = 0.f, y_acc = 0.f, score_acc = 0.f;
    for (int dy = -3; dy < 4; ++dy) {
      if ((py + dy) > 0 && (py + dy) < W) {
        for (int dx = -3; dx < 4; ++dx) {
          if ((px + dx) > 0 && (px + dx) < W) {
            const long idx = (long)(py + dy) * W + px + dx;
            float score = 0.f;
            if ((long)part * offset + idx < (long)p.src_planes * offset) score = s[idx];
            const float fx = (float)(px + dx), fy = (float)(py + dy);
            if (score > 0) {
              x_acc += fx * score;
              y_acc += fy * score;
              score_acc += score;
            }
          }
        }
      }
    }
    const int oi = (e + 1) * 3;
    dst[oi] = x_acc / score_acc;
    dst[oi + 1] = y_acc / score_acc;
    dst[oi + 2] = s[py * W + px];
  }
  if (tid == 0) dst[0] = (float)total;  // unclamped total, nms_layer.cu:110
}

// ---- map-free NMS -----------------------------------------------------------------------------
// Strip kernel: resized rows y0-1 .. y1 of one part are built in LDS (row interpolations T of the
// low-res rows the strip touches, then the column interpolation, accumulated over the scales in
// scale order — the arithmetic of resize_kernel), then flagged exactly like nms_strip_kernel.
#define NMSF_TROWS 8
// The kernel is ALU-bound (828 workgroups per frame, ~10 % of the chip next to the convolutions), so nothing is evaluated per
// item that is constant along a row or a column: loops run row by row (no integer division), the x-axis neighbour / fraction
// of every column is tabulated once per scale (xt), the y-axis ones are uniform per row, the first scale assigns 0 + d instead
// of a zero pass, and the final division is skipped for one scale (x / 1 == x).  Same operations on the same values.
struct YTab { int o0, o1, o2, o3; float dy; };  // one output row: offsets of its four row interpolations in T, fraction
struct XTab { int xn1; float dx; };  // axis_nb's clamped integer position (before padding) and fraction of one output column
#define NMS_STAMP() do { if (probe_me) { __syncthreads(); if (threadIdx.x == 0 && stamp_i < 30) p.probe[++stamp_i] = wall_clock64(); } } while (0)
__global__ __launch_bounds__(256) void nms_fused_strip_kernel(NmsParams p, ResizeParams r) {
  const KStamp kstamp_(p.stamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const bool probe_me = p.probe && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0;  // uniform
  int stamp_i = 0;
  NMS_STAMP();
  const int W = p.W, H = p.H;
  float* out = (float*)lds_raw;                 // [strip_rows + 2][W]
  float* T = out + (p.strip_rows + 2) * W;      // [NMSF_TROWS][W]
  XTab* xt = (XTab*)(T + NMSF_TROWS * W);       // [W]
  float* colmax = T + (NMSF_TROWS - 1) * W;     // [r.w <= W]: max |low-res value| per low-res column over the rows this strip touches; lives in the LAST
                                                // row of T, which is only written when a strip needs all NMSF_TROWS rows — then col_skip is off (below)
  __shared__ int wave_cnt[4];
  __shared__ float strip_max, strip_min, wave_min[4];
  __shared__ YTab ytab[24];
  const int strip = blockIdx.x, part = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y0 = strip * p.strip_rows;
  const int y1 = min(y0 + p.strip_rows, H);
  const int ya = max(y0 - 1, 0), yb = min(y1, H - 1);  // rows held in LDS: ya..yb
  const int nrow = yb - ya + 1;
  const long plane = (long)r.h * r.w;
  // ---- what cannot hold a maximum is not evaluated (exactly) -------------------------------------------------------------
  // A resized value is sum_n cubic(cubic(...)) / num of a 4x4 low-res neighbourhood per scale.  |cubic(v0..v3, d)| <=
  // S * max|v_i| with S = sum_i |w_i(d)| <= 1.375 for d in [-0.5, 1] (1.25 inside, 1.375 at the border where axis_nb's d is
  // negative, imresize_layer.cu:123-128), so |value| <= 1.375^2 * max|neighbourhood| = 1.8906 * max; NMS_BOUND adds 3 % for the
  // float / double roundings of cubic_interp.  nms_register_kernel (nms_layer.cu:15-46) flags only pixels with v > threshold:
  // where NMS_BOUND * max|low-res neighbourhood| <= threshold there is no flag, and such a pixel is also no obstacle to a
  // neighbour's flag (that neighbour has v > threshold >= it).  Strip level (any number of scales): nothing in the low-res rows
  // of this strip is large enough -> count 0, done.  Column level (one scale): columns whose 4 low-res columns are too small
  // get the placeholder `threshold` instead of their value.  On real heat maps most of a part's plane is background.
  constexpr float NMS_BOUND = 1.95f;
  float smax = 0.f, smin = __builtin_inff();  // largest / smallest column maximum
  for (int n = 0; n < r.num; ++n) {
    const ScaleGeo g = scale_geo(r, n);
    const float* sp = r.src + ((long)n * r.C + part) * plane;
    int nb[4];
    (void)axis_nb(ya, g.offset_y, g.fy, g.oh, g.padh, nb);
    const int rlo = nb[0];
    (void)axis_nb(yb, g.offset_y, g.fy, g.oh, g.padh, nb);
    const int rhi = nb[3];
    float m = 0.f;
    for (int c = tid; c < g.ow; c += 256) {
      float cm = 0.f;
      for (int rr = rlo; rr <= rhi; ++rr) {
        const float a = fabsf(sp[rr * g.rw + g.padw + c]);
        if (!(a <= cm)) cm = (a == a) ? a : __builtin_inff();
      }
      if (r.num == 1) colmax[c] = cm;
      m = cm > m ? cm : m;
      smin = cm < smin ? cm : smin;
    }
    smax = m > smax ? m : smax;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(smax, o), u = __shfl_xor(smin, o);
    smax = t > smax ? t : smax;
    smin = u < smin ? u : smin;
  }
  if (lane == 0) { ((float*)wave_cnt)[wave] = smax; wave_min[wave] = smin; }
  __syncthreads();
  if (tid == 0) {
    const float* wm = (const float*)wave_cnt;
    float m = wm[0], mn = wave_min[0];
    for (int w = 1; w < 4; ++w) { m = wm[w] > m ? wm[w] : m; mn = wave_min[w] < mn ? wave_min[w] : mn; }
    strip_max = m;
    strip_min = mn;
  }
  __syncthreads();
  NMS_STAMP();  // 2: bound pre-pass
  if (!(NMS_BOUND * strip_max > p.threshold)) {
    if (tid == 0) p.strip_count[part * p.nstrips + strip] = 0;
    return;
  }
  // column level only where it can skip something (noise maps: every column is above the bound; the tests below would be overhead)
  bool col_skip = r.num == 1 && !(NMS_BOUND * strip_min > p.threshold);
  if (col_skip) {  // colmax shares T's last row: only while the strip's low-res rows leave that row free (always with net/8 maps and 8- or 16-row strips)
    const ScaleGeo g0 = scale_geo(r, 0);
    int nb[4];
    (void)axis_nb(ya, g0.offset_y, g0.fy, g0.oh, g0.padh, nb);
    const int lo = nb[0];
    (void)axis_nb(yb, g0.offset_y, g0.fy, g0.oh, g0.padh, nb);
    if (nb[3] - lo + 1 >= NMSF_TROWS) col_skip = false;
  }
  for (int n = 0; n < r.num; ++n) {
    const ScaleGeo g = scale_geo(r, n);
    const float* sp = r.src + ((long)n * r.C + part) * plane;
    int nb[4];
    (void)axis_nb(ya, g.offset_y, g.fy, g.oh, g.padh, nb);
    const int rlo = nb[0];
    (void)axis_nb(yb, g.offset_y, g.fy, g.oh, g.padh, nb);
    const int rhi = nb[3];
    const int nt = rhi - rlo + 1;
    if (nt <= NMSF_TROWS) {
      for (int x = tid; x < W; x += 256) {  // axis_nb(x, ..) once per column
        const float x_on = (x - g.offset_x) * g.fx;
        int xn1 = (int)((double)x_on + 1e-5);
        xn1 = (xn1 < 0) ? 0 : xn1;
        if (col_skip) {  // one scale: is any of this column's 4 low-res columns large enough?  (xn1 = -1 marks "no")
          const int x0 = (xn1 - 1 < 0) ? xn1 : (xn1 - 1);
          const int x2 = (xn1 + 1 >= g.ow) ? (g.ow - 1) : (xn1 + 1);
          const int x3 = (x2 + 1 >= g.ow) ? (g.ow - 1) : (x2 + 1);
          float m = colmax[x0];
          m = colmax[xn1] > m ? colmax[xn1] : m;
          m = colmax[x2] > m ? colmax[x2] : m;
          m = colmax[x3] > m ? colmax[x3] : m;
          if (!(NMS_BOUND * m > p.threshold)) xn1 = -1;
        }
        xt[x].xn1 = xn1;
        xt[x].dx = x_on - xn1;
      }
      __syncthreads();
      NMS_STAMP();  // 3: column table
      // Row interpolations.  The ~8 output columns of one low-res cell share cubic_coef (19 of cubic_interp's 33 operations and
      // its 4 loads): where T has room behind its nt rows, the coefficients are tabulated per (low-res row, cell) first.
      float4* ctab = (float4*)(T + nt * W);   // [nt][ow] {a, c, b as two words}
      const bool use_ctab = (long)nt * g.ow * 4 <= (long)(NMSF_TROWS - nt) * W && (W & 3) == 0;
      if (use_ctab) {
        for (int it = tid; it < nt * g.ow; it += 256) {
          const int rr = it / g.ow, c = it - rr * g.ow;
          const float* row = sp + (rlo + rr) * g.rw + g.padw;
          const int x0 = (c - 1 < 0) ? c : (c - 1);
          const int x2 = (c + 1 >= g.ow) ? (g.ow - 1) : (c + 1);
          const int x3 = (x2 + 1 >= g.ow) ? (g.ow - 1) : (x2 + 1);
          const CubicCoef k = cubic_coef(row[x0], row[c], row[x2], row[x3]);
          float4 q;
          q.x = k.a; q.y = k.c;
          __builtin_memcpy(&q.z, &k.b, 8);
          ctab[it] = q;
        }
        __syncthreads();
      }
      NMS_STAMP();  // 4: coefficient table
      for (int rr = 0; rr < nt; ++rr) {
        const float* row = sp + (rlo + rr) * g.rw + g.padw;
        float* trow = T + rr * W;
        for (int x = tid; x < W; x += 256) {
          const XTab e = xt[x];
          if (e.xn1 < 0) continue;   // column below the bound (col_skip)
          if (use_ctab) {
            const float4 q = ctab[rr * g.ow + e.xn1];
            CubicCoef k;
            k.a = q.x; k.c = q.y; k.v1 = row[e.xn1];
            __builtin_memcpy(&k.b, &q.z, 8);
            trow[x] = cubic_eval(k, e.dx);
          } else {
            const int x0 = (e.xn1 - 1 < 0) ? e.xn1 : (e.xn1 - 1);
            const int x2 = (e.xn1 + 1 >= g.ow) ? (g.ow - 1) : (e.xn1 + 1);
            const int x3 = (x2 + 1 >= g.ow) ? (g.ow - 1) : (x2 + 1);
            trow[x] = cubic_interp(row[x0], row[e.xn1], row[x2], row[x3], e.dx);
          }
        }
      }
      __syncthreads();
      NMS_STAMP();  // 5: row interpolations
      // Column interpolation, one thread per output column: the rows of one low-res cell share cubic_coef of the four row
      // interpolations above / below (the y neighbourhood and fraction of every row are tabulated first: uniform per row)
      if (tid < nrow) {
        int yn[4];
        ytab[tid].dy = axis_nb(ya + tid, g.offset_y, g.fy, g.oh, g.padh, yn);
        ytab[tid].o0 = (yn[0] - rlo) * W; ytab[tid].o1 = (yn[1] - rlo) * W; ytab[tid].o2 = (yn[2] - rlo) * W; ytab[tid].o3 = (yn[3] - rlo) * W;
      }
      __syncthreads();
      // (round 5) a thread's columns x, x + 256, x + 512 are evaluated TOGETHER, row by row: the rows of one low-res cell share the row's
      // table entry and the branch on it (uniform), and three independent cubic_eval chains are in flight instead of one — the phase was a
      // latency chain (5.2 of the workgroup's 21 us for ~26 evaluations per thread).  Same operations on the same values.
      constexpr int XI = 3;
      for (int xb = tid; xb < W; xb += 256 * XI) {
        int xs[XI];
        bool act[XI];
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          xs[i] = xb + i * 256;
          act[i] = xs[i] < W;
          if (act[i] && col_skip && xt[xs[i]].xn1 < 0) {  // (one scale: this is the final value)
            for (int yy = 0; yy < nrow; ++yy) out[yy * W + xs[i]] = p.threshold;
            act[i] = false;
          }
          if (!act[i]) xs[i] = 0;   // (a safe column for the unconditional loads below; its results are not stored)
        }
        int prev = -1;
        CubicCoef k[XI];
        for (int yy = 0; yy < nrow; ++yy) {
          const YTab yt = ytab[yy];
          if (yt.o1 != prev) {
            prev = yt.o1;
#pragma unroll
            for (int i = 0; i < XI; ++i) k[i] = cubic_coef(T[yt.o0 + xs[i]], T[yt.o1 + xs[i]], T[yt.o2 + xs[i]], T[yt.o3 + xs[i]]);
          }
          float d[XI], prevv[XI];
#pragma unroll
          for (int i = 0; i < XI; ++i) { d[i] = cubic_eval(k[i], yt.dy); prevv[i] = n == 0 ? 0.f : out[yy * W + xs[i]]; }
#pragma unroll
          for (int i = 0; i < XI; ++i) if (act[i]) out[yy * W + xs[i]] = prevv[i] + d[i];
        }
      }
      __syncthreads();
      NMS_STAMP();  // 6: column interpolations
    } else {  // a strip spanning more low-res rows than the table holds (not with net/8 maps): per pixel
      for (int i = tid; i < nrow * W; i += 256) {
        const int yy = i / W, x = i - yy * W;
        int xn[4], yn[4];
        const float dx = axis_nb(x, g.offset_x, g.fx, g.ow, g.padw, xn);
        const float dy = axis_nb(ya + yy, g.offset_y, g.fy, g.oh, g.padh, yn);
        float t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* row = sp + yn[q] * g.rw;
          t[q] = cubic_interp(row[xn[0]], row[xn[1]], row[xn[2]], row[xn[3]], dx);
        }
        out[i] = (n == 0 ? 0.f : out[i]) + cubic_interp(t[0], t[1], t[2], t[3], dy);
      }
      __syncthreads();
    }
  }
  if (r.num != 1) {
    for (int i = tid; i < nrow * W; i += 256) out[i] = out[i] / r.num;
    __syncthreads();
  }
  int* list = p.strip_list + ((long)part * p.nstrips + strip) * p.max_peaks;
  const float* s = out - (long)ya * W;  // s[y * W + x] for y in ya..yb
  // nms_register_kernel, nms_layer.cu:15-46, in raster order.  Round 5: a lane owns ONE column of a 64-column block and walks down the strip's
  // rows with the 3x3 window in registers (3 LDS reads per pixel instead of 9: the phase was 6 of the workgroup's 21 us); one ballot per
  // (row, block) = 64 consecutive pixels of a row, so the groups in (row, block) order ARE raster order.  T is free now: ballots and
  // their exclusive prefix live there.
  const int NG = (W + 63) >> 6;                 // 64-column blocks per row
  const int nr = y1 - y0;                       // rows of this strip
  const int ngr = nr * NG;
  unsigned long long* bals = (unsigned long long*)T;      // [nr][NG]
  int* gbase = (int*)(bals + ngr);                        // [ngr + 1]: flags in front of group g (raster order)
  {
    const float thr = p.threshold;
    for (int c = wave; c < NG; c += 4) {
      const int x = (c << 6) + lane;
      const bool xin = x > 0 && x < W - 1;
      const int xc = xin ? x : 1;               // any interior column when out of range (the flag is dropped)
      auto rowp = [&](int yy) __attribute__((always_inline)) { return s + (long)(yy < ya ? ya : (yy > yb ? yb : yy)) * W + xc; };
      const float* pa = rowp(y0 - 1);
      const float* pb = rowp(y0);
      float a0 = pa[-1], a1 = pa[0], a2 = pa[1];
      float b0 = pb[-1], b1 = pb[0], b2 = pb[1];
      for (int y = y0; y < y1; ++y) {
        const float* pc = rowp(y + 1);
        const float c0 = pc[-1], c1 = pc[0], c2 = pc[1];
        const bool in = xin && y > 0 && y < H - 1;
        const float v = b1;
        const bool f = in && v > thr && v > a1 && v > c1 && v > b0 && v > b2 && v > a0 && v > c0 && v > c2 && v > a2;
        const unsigned long long bal = __ballot(f);
        if (lane == 0) bals[(y - y0) * NG + c] = bal;
        a0 = b0; a1 = b1; a2 = b2;
        b0 = c0; b1 = c1; b2 = c2;
      }
    }
  }
  __syncthreads();
  NMS_STAMP();  // 7: flags + ballots
  if (wave == 0) {   // exclusive prefix of the groups' flag counts, 64 groups per step
    int run = 0;
    for (int g0 = 0; g0 < ngr; g0 += 64) {
      const int g = g0 + lane;
      const int cnt = g < ngr ? __popcll(bals[g]) : 0;
      int inc = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      if (g < ngr) gbase[g] = run + inc - cnt;
      run += __shfl(inc, 63);
    }
    if (lane == 0) { gbase[ngr] = run; p.strip_count[part * p.nstrips + strip] = run; }
  }
  __syncthreads();
  for (int g = wave; g < ngr; g += 4) {
    const int base = gbase[g];
    if (base >= p.max_peaks) break;             // prefixes grow with g: nothing further fits under the cap
    const unsigned long long bal = bals[g];
    const int ord = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (((bal >> lane) & 1) && ord < p.max_peaks) {
      const int r = g / NG, c = g - r * NG;
      list[ord] = (y0 + r) * W + (c << 6) + lane;
    }
  }
  NMS_STAMP();  // 8: list
  if (probe_me && tid == 0) p.probe[0] = stamp_i;
}

// Write kernel: the 49 window values of every kept peak are evaluated on demand by all threads,
// then one thread per peak accumulates them in the reference's (dy, dx) order.
__global__ __launch_bounds__(256) void nms_fused_write_kernel(NmsParams p, ResizeParams r) {
  const KStamp kstamp_(p.stamp ? p.stamp + 2 : nullptr);
  extern __shared__ int dyn_i[];
  int* prefix = dyn_i;                                   // [nstrips+1]
  float* win = (float*)(dyn_i + p.nstrips + 1);          // [max_peaks][50]: 49 window values + centre
  int* pix = (int*)(win + p.max_peaks * 50);             // [max_peaks]
  const int part = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ ScaleGeo geo[RTP_MAX_SCALES];
  if (p.clear_flag && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *p.clear_flag = 0;
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < p.nstrips; ++i) {
      prefix[i] = run;
      run += p.strip_count[part * p.nstrips + i];
    }
    prefix[p.nstrips] = run;
  }
  fill_scale_geo(geo, r);
  const int total = prefix[p.nstrips];
  const int W = p.W, H = p.H;
  float* dst = p.peaks + (long)part * (p.max_peaks + 1) * 3;
  const int nall = total < p.max_peaks ? total : p.max_peaks;
  // this workgroup's share of the part's peaks: ordinals e0 .. e0+n-1 (local index e below)
  const int share = (nall + (int)gridDim.y - 1) / (int)gridDim.y;
  const int e0 = (int)blockIdx.y * share;
  const int n = max(0, min(share, nall - e0));
  for (int e = tid; e < n; e += 256) {
    const int ge = e0 + e;
    int st = 0;
    while (prefix[st + 1] <= ge) ++st;  // strip holding ordinal ge
    pix[e] = p.strip_list[((long)part * p.nstrips + st) * p.max_peaks + (ge - prefix[st])];
  }
  __syncthreads();
  for (int it = tid; it < n * 50; it += 256) {
    const int e = it / 50, wq = it - e * 50;
    const int px = pix[e] % W, py = pix[e] / W;
    float score = 0.f;
    if (wq == 49) score = resized_at(geo, r, part, py, px);
    else {
      const int dy = wq / 7 - 3, dx = wq % 7 - 3;
      // writeResultKernel, nms_layer.cu:70-105: the bound on y is `width`, so the window may run
      // past the last row of this part into the first rows of the next plane (or past the map: 0)
      if ((py + dy) > 0 && (py + dy) < W && (px + dx) > 0 && (px + dx) < W) {
        const int row = py + dy;
        const int pl = part + row / H;
        if (pl < p.src_planes) score = resized_at(geo, r, pl, row % H, px + dx);
      }
    }
    win[it] = score;
  }
  __syncthreads();
  for (int e = tid; e < n; e += 256) {
    const int px = pix[e] % W, py = pix[e] / W;
    float x_acc = 0.f, y_acc = 0.f, score_acc = 0.f;
    for (int dy = -3; dy < 4; ++dy) {
      if ((py + dy) > 0 && (py + dy) < W) {
        for (int dx = -3; dx < 4; ++dx) {
          if ((px + dx) > 0 && (px + dx) < W) {
            const float score = win[e * 50 + (dy + 3) * 7 + dx + 3];
            const float fx = (float)(px + dx), fy = (float)(py + dy);
            if (score > 0) {
              x_acc += fx * score;
              y_acc += fy * score;
              score_acc += score;
            }
          }
        }
      }
    }
    const int oi = (e0 + e + 1) * 3;
    dst[oi] = x_acc / score_acc;
    dst[oi + 1] = y_acc / score_acc;
    dst[oi + 2] = win[e * 50 + 49];
  }
  if (tid == 0 && blockIdx.y == 0) dst[0] = (float)total;  // unclamped total, nms_layer.cu:110
}

// >64 KiB dynamic LDS opt-in, once per (kernel, device, size): hipFuncSetAttribute is a driver call
// (it was issued per launch) and has no place inside a stream capture.
template <auto KERN>  // one cache per kernel (not per kernel TYPE: several kernels share a signature)
static hipError_t ensure_lds(size_t bytes) {
  static std::atomic<size_t> have[32];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<size_t>& h = have[dev & 31];
  if (h.load(std::memory_order_relaxed) >= bytes) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) h.store(bytes, std::memory_order_relaxed);
  return e;
}

// Dynamic-LDS budget of the post-processing kernels: the CU's 160 KiB minus what the kernels declare statically (wave counters, the
// per-scale geometry, YTab, the sort replica's explicit stack: < 2 KiB in every kernel of this file; 4 KiB reserved).
constexpr size_t kPostDynLdsMax = 160 * 1024 - 4096;

hipError_t launch_nms_fused(const NmsParams& p, const ResizeParams& r, hipStream_t stream) {
  const size_t lds1 = (size_t)(p.strip_rows + 2 + NMSF_TROWS) * p.W * sizeof(float) + (size_t)p.W * sizeof(XTab);
  if (lds1 > kPostDynLdsMax) return hipErrorInvalidValue;
  if (lds1 > 64 * 1024) {
    hipError_t e = ensure_lds<nms_fused_strip_kernel>(lds1);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(nms_fused_strip_kernel, dim3(p.nstrips, p.num_parts), dim3(256), lds1, stream, p, r);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const size_t lds2 = (p.nstrips + 1) * sizeof(int) + (size_t)p.max_peaks * 51 * sizeof(float);
  hipLaunchKernelGGL(nms_fused_write_kernel, dim3(p.num_parts, 4), dim3(256), lds2, stream, p, r);
  return hipGetLastError();
}

hipError_t launch_nms(const NmsParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(nms_strip_kernel, dim3(p.nstrips, p.num_parts), dim3(256), 0, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nms_write_kernel, dim3(p.num_parts), dim3(256), (p.nstrips + 1) * sizeof(int), stream, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// connectLimbs / connectLimbsCOCO
// ---------------------------------------------------------------------------------------
__constant__ int kCocoLimb[38] = {1, 2, 1, 5, 2, 3, 3, 4, 5, 6, 6, 7, 1, 8, 8, 9, 9, 10, 1, 11, 11, 12, 12, 13, 1, 0, 0, 14, 14, 16, 0, 15, 15, 17, 2, 16, 5, 17};
__constant__ int kCocoMap[38] = {31, 32, 39, 40, 33, 34, 35, 36, 41, 42, 43, 44, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 47, 48, 49, 50, 53, 54, 51, 52, 55, 56, 37, 38, 45, 46};
__constant__ int kMpiLimb[28] = {0, 1, 1, 2, 2, 3, 3, 4, 1, 5, 5, 6, 6, 7, 1, 14, 14, 11, 11, 12, 12, 13, 14, 8, 8, 9, 9, 10};
__constant__ int kMpiMap[28] = {16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 38, 39, 40, 41, 42, 43, 32, 33, 34, 35, 36, 37};

#define CONNECT_ERR_RANGE (-34)

// connectLimbs* in three kernels (all bit-exact restatements; -ffp-contract=off):
//  connect_pairs_kernel    grid (ceil(max_peaks^2/256), limbs): the PAF line integral of every (i,j)
//      candidate pair (rtpose.cpp:897-951 / :611-651), one pair per thread, result to global
//      scratch: the survivors of each block of 256 pairs (loop order q = (i-1)*nB + (j-1)) compacted
//      at the head of the block's slots, their count in cand_blk.
//  connect_match_kernel    one workgroup per limb: compacts the survivors in loop order, orders them
//      as std::sort(.., ColumnCompare) would (:953-954) and runs the greedy assignment (:956-980).
//      If all scores are distinct the sorted order is unique, so a bitonic sort on the 64-bit key
//      (score descending, loop order ascending) gives it.  std::sort leaves the relative order of
//      EQUAL scores implementation-defined, but that only matters if two tied candidates are both
//      still free when the greedy scan reaches them: the scan detects exactly that case (and NaNs)
//      and only then lane 0 runs the libstdc++-exact replica.
//  connect_assemble_kernel one workgroup: person assembly + emission.
// Short kernels with <= 56 KiB of LDS: a convolution workgroup of another frame can share the CU.
__device__ __forceinline__ void limb_setup(const ConnectParams& p, int k, const float*& candA, const float*& candB, int& nA, int& nB) {
  const bool coco = p.model == 0;
  const int* limbSeq = coco ? kCocoLimb : kMpiLimb;
  const int peaks_offset = 3 * (p.max_peaks + 1);
  candA = p.peaks + limbSeq[2 * k] * peaks_offset;
  candB = p.peaks + limbSeq[2 * k + 1] * peaks_offset;
  nA = (int)candA[0];
  nB = (int)candB[0];
  if (nA > p.max_peaks) nA = p.max_peaks;  // defined-behaviour clamp (see oracle NOTE)
  if (nB > p.max_peaks) nB = p.max_peaks;
}

// Workgroups of 1024 threads = four 256-pair blocks of one limb: the limb's two PAF planes are staged into LDS once for all four, and
// the 304 small workgroups that used to land one or two per CU (each blocking a whole CU for a convolution workgroup of another
// frame) become 76 dense ones.  Survivors are still compacted per block of 256 pairs (cand_blk), so the results are unchanged.
#define PAIRS_WG 1024
template <bool FUSED>
__device__ __forceinline__ void connect_pairs_body(const ConnectParams& p, const ResizeParams& r, const int stage, unsigned char* lds_raw) {
  const int k = blockIdx.y;
  const int cap = p.max_peaks * p.max_peaks;
  const bool coco = p.model == 0;
  const int* mapIdx = coco ? kCocoMap : kMpiMap;
  const int NW = p.net_w, NH = p.net_h;
  const float* map_x = p.heat + (long)mapIdx[2 * k] * NH * NW;
  const float* map_y = p.heat + (long)mapIdx[2 * k + 1] * NH * NW;
  const float *candA, *candB;
  int nA, nB;
  limb_setup(p, k, candA, candB, nA, nB);
  __shared__ int wave_cnt[PAIRS_WG / 64];
  const int q = blockIdx.x * PAIRS_WG + threadIdx.x;
  const int sub = threadIdx.x >> 8;                      // 256-pair block of this thread inside the workgroup
  const int blk256 = blockIdx.x * (PAIRS_WG / 256) + sub;  // ... and inside the limb (the unit of cand_blk)
  const int nblk256 = (p.max_peaks * p.max_peaks + 255) / 256;
  const int npairs = nA * nB;
  if (blockIdx.x * PAIRS_WG >= npairs) {  // whole workgroup past the last pair (uniform)
    if ((threadIdx.x & 255) == 0 && blk256 < nblk256) p.cand_blk[k * nblk256 + blk256] = 0;
    return;
  }
  // FUSED: the limb's two PAF channels (all scales) go to LDS once per workgroup; the 20 bicubic
  // samples of a pair (16 taps each per scale) then never leave the CU
  const long lplane = (long)r.h * r.w;
  float* lmap = (float*)lds_raw;  // [2][num][h][w]
  __shared__ ScaleGeo geo[RTP_MAX_SCALES];
  if (FUSED) fill_scale_geo(geo, r);
  if (FUSED && stage) {
    for (int which = 0; which < 2; ++which)          // plane by plane: no integer division per element
      for (int n = 0; n < r.num; ++n) {
        const float* src = r.src + ((long)n * r.C + mapIdx[2 * k + which]) * lplane;
        float* dst = lmap + ((long)which * r.num + n) * lplane;
        for (int o = threadIdx.x; o < (int)lplane; o += PAIRS_WG) dst[o] = src[o];
      }
    __syncthreads();
  }
  const int num_inter = 10;
  int pass = 0;
  float conn_score = 0.f;
  const int i = q / (nB > 0 ? nB : 1) + 1, j = q % (nB > 0 ? nB : 1) + 1;
  if (q < npairs) {
  const float s_x = candA[i * 3];
  const float s_y = candA[i * 3 + 1];
  const float d_x = candB[j * 3] - candA[i * 3];
  const float d_y = candB[j * 3 + 1] - candA[i * 3 + 1];
  float norm_vec;
  if (coco) norm_vec = sqrtf(d_x * d_x + d_y * d_y);
  else norm_vec = (float)sqrt((double)d_x * (double)d_x + (double)d_y * (double)d_y);  // pow(d,2) in double
  if (!(norm_vec < 1e-6)) {
    const float vec_x = d_x / norm_vec;
    const float vec_y = d_y / norm_vec;
    int idxs[10];
    bool bad = false;
#pragma unroll
    for (int lm = 0; lm < num_inter; lm++) {
      const float ry = roundf(s_y + lm * d_y / num_inter), rx = roundf(s_x + lm * d_x / num_inter);
      int my = (int)ry;
      int mx = (int)rx;
      if (coco) {
        if (mx >= NW) mx = NW - 1;
        if (my >= NH) my = NH - 1;
      }
      // A NaN or a value beyond int is "integer indefinite" = INT_MIN in the reference's host conversion (cvttss2si), i.e. its
      // CHECK_GE(mx, 0) fails; v_cvt_i32_f32 would give 0 / INT_MAX instead.  Reached by a net that is taller than wide: the write
      // kernel's centroid window is bounded by `width` in y too (nms_layer.cu:79) and divides 0 by 0 for peaks below row width + 3.
      const bool indef = !(rx >= -2147483648.f && rx < 2147483648.f) || !(ry >= -2147483648.f && ry < 2147483648.f);
      if (indef || mx < 0 || my < 0 || mx >= NW || my >= NH) { bad = true; mx = 0; my = 0; }
      idxs[lm] = my * NW + mx;
    }
    // The 10 samples in order (rtpose.cpp:931-947).  A pair survives only with count > inter_min_above: once more samples have failed than
    // that leaves room for, the pair is rejected whatever the remaining samples say, and the wave stops evaluating when that holds for EVERY
    // lane (noise maps: 4096 candidate pairs per limb, almost all of which fail within the first samples; each sample is 2 x 16-tap bicubic
    // evaluations per scale).  Survivors have run all 10 samples in the reference's order: the same sum, the same count.  (The range check of
    // the sample coordinates — `bad`, the reference's CHECK — covered all 10 above and does not depend on this loop.)
    float sum = 0;
    int count = 0;
    const int allowed_fail = num_inter - (p.inter_min_above + 1);   // COCO: 0 (all 10 must pass), MPI: 1
    bool alive = true;
#pragma unroll
    for (int lm = 0; lm < num_inter; lm++) {
      if (!p.pairs_full && !__any(alive)) break;   // wave-uniform
      float pxl, pyl;
      if (FUSED) {  // the two PAF samples straight from the low-res maps (no resized map in memory)
        const int my = idxs[lm] / NW, mx = idxs[lm] - my * NW;
        if (stage) resized_pair(geo, r.num, lmap, lmap + r.num * lplane, lplane, my, mx, &pxl, &pyl);
        else resized_pair(geo, r.num, r.src + (long)mapIdx[2 * k] * lplane, r.src + (long)mapIdx[2 * k + 1] * lplane, (long)r.C * lplane, my, mx, &pxl, &pyl);
      } else {
        pxl = map_x[idxs[lm]];
        pyl = map_y[idxs[lm]];
      }
      const float score = (vec_x * pxl + vec_y * pyl);
      if (score > p.inter_threshold) {
        sum = sum + score;
        count++;
      }
      if (lm + 1 - count > allowed_fail) alive = false;   // (NaN scores fail the compare like in the reference and count as failed)
    }
    if (bad) *p.num_people = CONNECT_ERR_RANGE;  // the reference CHECK-fails here (rtpose.cpp:928)
    else if (count > p.inter_min_above) {
      pass = 1;
      conn_score = sum / count;
    }
  }
  }
  // survivors of every 256-pair block, compacted in loop order at the head of the block's 256 slots
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(pass);
  if (lane == 0) wave_cnt[wave] = __popcll(bal);
  __syncthreads();
  int before = 0;
  for (int w = sub * 4; w < wave; ++w) before += wave_cnt[w];
  const int lo = before + __popcll(bal & ((1ull << lane) - 1ull));
  if (pass) {
    p.cand_score[(long)k * cap + blk256 * 256 + lo] = conn_score;
    p.cand_ij[(long)k * cap + blk256 * 256 + lo] = (i << 16) | j;
  }
  if ((threadIdx.x & 255) == 0 && blk256 < nblk256)
    p.cand_blk[k * nblk256 + blk256] = wave_cnt[sub * 4] + wave_cnt[sub * 4 + 1] + wave_cnt[sub * 4 + 2] + wave_cnt[sub * 4 + 3];
}
template <bool FUSED>
__global__ __launch_bounds__(PAIRS_WG) void connect_pairs_kernel(ConnectParams p, ResizeParams r, int stage) {
  const KStamp kstamp_(p.stamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  connect_pairs_body<FUSED>(p, r, stage, lds_raw);
}

// 64-bit sort key: ascending key order == (score descending, loop order ascending)
__device__ __forceinline__ unsigned long long match_key(float score, int ord, int i, int j) {
  unsigned int u = __float_as_uint(score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotonic in the float order
  return ((unsigned long long)(~u) << 32) | (unsigned int)((ord << 14) | (i << 7) | j);
}
__device__ __forceinline__ float key_score(unsigned long long key) {
  unsigned int u = ~(unsigned int)(key >> 32);
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}

// one limb of connect_match_kernel (k was blockIdx.x); every thread of the workgroup calls it
__device__ __forceinline__ void connect_match_limb(const ConnectParams& p, const int k, unsigned char* lds_raw) {
  unsigned long long* keys = (unsigned long long*)lds_raw;  // [pow2 >= survivors]
  __shared__ int running;
  __shared__ int flags;  // 1: NaN score seen   2: greedy scan met an ambiguous tie
  __shared__ int s_cnt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cap = p.max_peaks * p.max_peaks;
  const bool coco = p.model == 0;
  const int* limbSeq = coco ? kCocoLimb : kMpiLimb;
  const int peaks_offset = 3 * (p.max_peaks + 1);
  const float *candA, *candB;
  int nA, nB;
  limb_setup(p, k, candA, candB, nA, nB);
  if (tid == 0) { running = 0; flags = 0; s_cnt = 0; }
  __syncthreads();
  if (nA == 0 || nB == 0) {
    if (tid == 0) { p.cand_count[k] = 0; p.conn_count[k] = 0; }
    return;
  }
  const int npairs = nA * nB;
  const float* gs = p.cand_score + (long)k * cap;
  const int* gij = p.cand_ij + (long)k * cap;
  // ---- survivors in loop order: the pair kernel compacted each block of 256; gather the blocks
  const int nblk = (npairs + 255) / 256, blk_stride = (cap + 255) / 256;
  __shared__ int blk_off[65];
  if (tid == 0) {
    int run = 0;
    for (int b = 0; b < nblk; ++b) { blk_off[b] = run; run += p.cand_blk[k * blk_stride + b]; }
    blk_off[nblk] = run;
    running = run;
  }
  __syncthreads();
  auto gather = [&](auto&& put) {
    for (int b0 = 0; b0 < nblk; b0 += 8) {
      float sc[8];
      int ij[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // 16 independent loads in flight
        const int idx = min((b0 + u) * 256 + tid, cap - 1);
        sc[u] = gs[idx];
        ij[u] = gij[idx];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u;
        if (b < nblk) {
          const int o = blk_off[b], c = blk_off[b + 1] - o;
          if (tid < c) put(o + tid, sc[u], ij[u]);
        }
      }
    }
  };
  gather([&](int ord, float sc, int ij) {
    if (!(sc == sc)) flags = 1;  // NaN breaks the ordering: force the exact path
    keys[ord] = match_key(sc, ord, ij >> 16, ij & 0xffff);
  });
  const int nc = running;
  int n2 = 64;
  while (n2 < nc) n2 <<= 1;
  for (int t = nc + tid; t < n2; t += 256) keys[t] = ~0ull;
  __syncthreads();
  // ---- bitonic sort, ascending keys
  const int half = n2 >> 1;
  for (int kk = 2; kk <= n2; kk <<= 1) {
    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
      if (half >= 2048) {  // 8+ pairs per thread: all reads of a group first, branch-free writes
        for (int t0 = tid; t0 < half; t0 += 2048) {
          unsigned long long ka[8], kb[8];
          int ia[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int t = t0 + u * 256;
            ia[u] = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
            ka[u] = keys[ia[u]];
            kb[u] = keys[ia[u] | jj];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const bool sw = (ka[u] > kb[u]) == ((ia[u] & kk) == 0);
            keys[ia[u]] = sw ? kb[u] : ka[u];
            keys[ia[u] | jj] = sw ? ka[u] : kb[u];
          }
        }
      } else {
        for (int t = tid; t < half; t += 256) {
          const int a = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
          const int b = a | jj;
          const unsigned long long ka = keys[a], kb = keys[b];
          const bool sw = (ka > kb) == ((a & kk) == 0);
          keys[a] = sw ? kb : ka;
          keys[b] = sw ? ka : kb;
        }
      }
      __syncthreads();
    }
  }
  // ---- greedy assignment (:956-980) on wave 0, 64 sorted rows per step
  const int num = nA < nB ? nA : nB;
  int* conn = p.conn + (long)k * p.max_peaks * 2;
  float* cs = p.conn_score + (long)k * p.max_peaks;
  const int partA_off = limbSeq[2 * k] * peaks_offset, partB_off = limbSeq[2 * k + 1] * peaks_offset;
  if (wave == 0) {
    bool ambiguous = flags != 0;
    int cnt = 0;
    if (!ambiguous) {
      unsigned long long occA[2] = {0, 0}, occB[2] = {0, 0};  // wave-uniform copies (max_peaks <= 127)
      for (int base = 0; base < nc && cnt < num && !ambiguous; base += 64) {
        const int row = base + lane;
        const unsigned long long key = row < nc ? keys[row] : 0ull;
        const float score = key_score(key);
        const int i = row < nc ? (int)((key >> 7) & 127) : 1, j = row < nc ? (int)(key & 127) : 1;
        unsigned long long done = 0;  // lanes of this chunk already decided
        while (cnt < num) {
          const bool free_ = row < nc && !((done >> lane) & 1) && !((occA[(i - 1) >> 6] >> ((i - 1) & 63)) & 1) &&
                             !((occB[(j - 1) >> 6] >> ((j - 1) & 63)) & 1);
          const unsigned long long bal = __ballot(free_);
          if (!bal) break;
          const int first = __ffsll((long long)bal) - 1;
          const float sc = __shfl(score, first);
          const int wi = __shfl(i, first), wj = __shfl(j, first);
          // any OTHER free row with the same score (later in this chunk or in the following rows)?
          bool amb = __any(free_ && lane != first && score == sc);
          if (!amb) {
            for (int r2 = base + 64; r2 < nc; ++r2) {  // tied rows may continue past this chunk
              const unsigned long long k2 = keys[r2];
              if (!(key_score(k2) == sc)) break;
              const int i2 = (int)((k2 >> 7) & 127), j2 = (int)(k2 & 127);
              if (!((occA[(i2 - 1) >> 6] >> ((i2 - 1) & 63)) & 1) && !((occB[(j2 - 1) >> 6] >> ((j2 - 1) & 63)) & 1)) { amb = true; break; }
            }
          }
          if (amb) { ambiguous = true; break; }
          if (lane == 0) {
            conn[cnt * 2] = partA_off + wi * 3 + 2;
            conn[cnt * 2 + 1] = partB_off + wj * 3 + 2;
            cs[cnt] = sc;
          }
          cnt++;
          occA[(wi - 1) >> 6] |= 1ull << ((wi - 1) & 63);
          occB[(wj - 1) >> 6] |= 1ull << ((wj - 1) & 63);
          done |= (2ull << first) - 1ull;  // rows up to and including `first` are decided
        }
      }
    }
    if (lane == 0) { if (ambiguous) flags |= 2; s_cnt = cnt; }
  }
  __syncthreads();
  if (flags) {  // exact path: the survivors again in loop order, then libstdc++'s introsort on lane 0
    Cand* cands = (Cand*)lds_raw;
    gather([&](int ord, float sc, int ij) { cands[ord].score = sc; cands[ord].ij = ij; });
    __syncthreads();
    if (tid == 0) {
      std_sort_replica(cands, nc);
      int cnt = 0;
      unsigned long long occA[2] = {0, 0}, occB[2] = {0, 0};
      for (int row = 0; row < nc; ++row) {
        if (cnt == num) break;
        const Cand c = cands[row];
        const int i = c.ij >> 16, j = c.ij & 0xffff;
        const unsigned long long ba = 1ull << ((i - 1) & 63), bb = 1ull << ((j - 1) & 63);
        if (!(occA[(i - 1) >> 6] & ba) && !(occB[(j - 1) >> 6] & bb)) {
          conn[cnt * 2] = partA_off + i * 3 + 2;
          conn[cnt * 2 + 1] = partB_off + j * 3 + 2;
          cs[cnt] = c.score;
          cnt++;
          occA[(i - 1) >> 6] |= ba;
          occB[(j - 1) >> 6] |= bb;
        }
      }
      s_cnt = cnt;
    }
  }
  if (tid == 0) {
    p.cand_count[k] = nc | (flags ? (1 << 30) : 0);  // bit 30: the exact path ran (diagnostics only)
    p.conn_count[k] = s_cnt;
  }
}

// One workgroup per limb by default.  RTP_MATCH_WGS = n lets n workgroups take the limbs in turn: an experiment on WHY the
// post-processing costs the pipeline 3x its CU time (profiles/r03_pool_and_postproc.txt section 5).  Hypothesis: 19 workgroups of ~95 us
// hold 19 CUs while convolution launches need 248 at once.  Result: fewer workgroups are SLOWER (19: 1053 frames/s, 8: 1028, 4: 1017,
// 2: 973) — what the chain costs is its latency (a frame holds its pipeline slot until its joints are on the host), not the CUs.
__global__ __launch_bounds__(256) void connect_match_kernel(ConnectParams p) {
  const KStamp kstamp_(p.stamp ? p.stamp + 2 : nullptr);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  for (int k = blockIdx.x; k < p.num_limbs; k += gridDim.x) {
    connect_match_limb(p, k, lds_raw);
    __syncthreads();  // the next limb re-initialises the shared state
  }
}

// Person assembly (rtpose.cpp:982-1046 / :684-722) + emission (:1051-1073 / :726-748), one workgroup.
// The subset table lives in LDS: idx[max_rows][num_parts] (flat index of the part's score in the
// peaks array, 0 = absent; int16), score[max_rows] (double), cnt[max_rows].
// The reference walks connections in order and for each scans all rows for subset[j][partA] ==
// indexA.  Within one limb every A peak belongs to at most one connection and only column partB is
// written, so the scan can be turned inside out: ONE pass over the rows looks up the connection
// owning the row's partA peak (conn_of[]), updates the row, and marks the connection as matched;
// unmatched connections then append their rows in connection order, exactly as the serial loop.
__device__ __forceinline__ void connect_assemble_body(const ConnectParams& p, unsigned char* lds_raw) {
  const int NP = p.num_parts;
  double* sscore = (double*)lds_raw;                       // [max_rows]
  short* scnt = (short*)(sscore + p.max_rows);             // [max_rows]
  short* sidx = scnt + p.max_rows;                         // [max_rows][NP]
  __shared__ short conn_of[128];   // A-peak ordinal -> connection (or -1); single-sided: presence flag
  __shared__ int matched[128];
  __shared__ int wave_cnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool coco = p.model == 0;
  const int* limbSeq = coco ? kCocoLimb : kMpiLimb;
  const int peaks_offset = 3 * (p.max_peaks + 1);
  if (*p.num_people < 0) return;  // the pair kernel flagged out-of-contract data
  // Everything the serial limb loop reads (peaks, connections, their scores and counts) is pulled
  // into LDS first with independent loads; the loop then never waits on global memory.
  const float* peaks = p.peaks;
  const int* conn_all = p.conn;
  const float* cs_all = p.conn_score;
  const int* ncon_all = p.conn_count;
  if (p.assemble_preload) {
    float* l_peaks = (float*)(lds_raw + (((size_t)p.max_rows * (sizeof(double) + sizeof(short) + sizeof(short) * NP) + 7) & ~(size_t)7));
    const int npk = NP * peaks_offset;
    int* l_conn = (int*)(l_peaks + npk);
    float* l_cs = (float*)(l_conn + p.num_limbs * p.max_peaks * 2);
    int* l_ncon = (int*)(l_cs + p.num_limbs * p.max_peaks);
    for (int i = tid; i < npk; i += 256) l_peaks[i] = p.peaks[i];
    for (int i = tid; i < p.num_limbs * p.max_peaks * 2; i += 256) l_conn[i] = p.conn[i];
    for (int i = tid; i < p.num_limbs * p.max_peaks; i += 256) l_cs[i] = p.conn_score[i];
    if (tid < p.num_limbs) l_ncon[tid] = p.conn_count[tid];
    __syncthreads();
    peaks = l_peaks; conn_all = l_conn; cs_all = l_cs; ncon_all = l_ncon;
  }
  int nrows = 0;                  // uniform over the workgroup

  // rows for the items (tid < n) whose `want` is set, appended in item order
  auto append_rows = [&](int n, bool want, int partA, int idxA, int partB, int idxB, int cnt, double score) {
    const unsigned long long bal = __ballot(want && tid < n);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    const int row = nrows + before + __popcll(bal & ((1ull << lane) - 1ull));
    if (want && tid < n && row < p.max_rows) {
      for (int q = 0; q < NP; ++q) sidx[row * NP + q] = 0;
      sidx[row * NP + partA] = (short)idxA;
      if (partB >= 0) sidx[row * NP + partB] = (short)idxB;
      scnt[row] = (short)cnt;
      sscore[row] = score;
    }
    __syncthreads();
    nrows = nrows + total < p.max_rows ? nrows + total : p.max_rows;
  };

  for (int k = 0; k < p.num_limbs; ++k) {
    const int partA = limbSeq[2 * k], partB = limbSeq[2 * k + 1];
    const float* candA = peaks + partA * peaks_offset;
    const float* candB = peaks + partB * peaks_offset;
    int nA = (int)candA[0], nB = (int)candB[0];
    if (nA > p.max_peaks) nA = p.max_peaks;  // defined-behaviour clamp (see oracle NOTE)
    if (nB > p.max_peaks) nB = p.max_peaks;
    if (nA == 0 && nB == 0) continue;
    if (nA == 0 || nB == 0) {
      const int part = (nA == 0) ? partB : partA;
      const float* cand = (nA == 0) ? candB : candA;
      const int n = (nA == 0) ? nB : nA;
      if (tid < 128) conn_of[tid] = 0;
      __syncthreads();
      if (coco) {  // rtpose.cpp:849-858, 873-881: only peaks no row holds yet; the MPI version appends unconditionally
        for (int j = tid; j < nrows; j += 256) {
          const int v = sidx[j * NP + part];
          if (v) conn_of[(v - part * peaks_offset - 2) / 3] = 1;
        }
        __syncthreads();
      }
      const int i = tid + 1;
      const bool want = tid < n && !conn_of[i < 128 ? i : 0];
      append_rows(n, want, part, part * peaks_offset + i * 3 + 2, -1, 0, 1, tid < n ? (double)cand[i * 3 + 2] : 0.0);
      continue;
    }
    const int nconn = ncon_all[k];
    const int* conn = conn_all + (long)k * p.max_peaks * 2;
    const float* cs = cs_all + (long)k * p.max_peaks;
    if (k != 0 && nconn == 0) continue;
    int indexA = 0, indexB = 0;
    float csv = 0.f;
    if (tid < nconn) { indexA = conn[tid * 2]; indexB = conn[tid * 2 + 1]; csv = cs[tid]; }
    if (k != 0) {
      if (tid < 128) { conn_of[tid] = -1; matched[tid] = 0; }
      __syncthreads();
      if (tid < nconn) conn_of[(indexA - partA * peaks_offset - 2) / 3] = (short)tid;
      __syncthreads();
      for (int j = tid; j < nrows; j += 256) {
        const int v = sidx[j * NP + partA];
        if (!v) continue;
        const int c = conn_of[(v - partA * peaks_offset - 2) / 3];
        if (c < 0) continue;
        const int iB = conn[c * 2 + 1];
        sidx[j * NP + partB] = (short)iB;
        scnt[j] = (short)(scnt[j] + 1);
        sscore[j] = (sscore[j] + (double)peaks[iB]) + (double)cs[c];
        matched[c] = 1;
      }
      __syncthreads();
    }
    const bool want = tid < nconn && (k == 0 || !matched[tid]);
    const double sc = tid < nconn ? (double)(peaks[indexA] + peaks[indexB]) + (double)csv : 0.0;
    append_rows(nconn, want, partA, indexA, partB, indexB, 2, sc);
  }
  __syncthreads();

  // emit rows that pass the subset thresholds, in row order, at most max_people
  int out = 0;
  for (int base = 0; base < nrows && out < p.max_people; base += 256) {
    const int j = base + tid;
    int ok = 0;
    if (j < nrows) {
      const double c = (double)scnt[j];
      ok = (c >= (double)p.min_subset_cnt) && ((sscore[j] / c) > (double)p.min_subset_score);
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    const int slot = out + before + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok && slot < p.max_people) {
      for (int q = 0; q < NP; ++q) {
        const int idx = sidx[j * NP + q];
        float* o = p.joints + ((long)slot * NP + q) * 3;
        if (idx) {
          o[2] = peaks[idx];
          o[1] = peaks[idx - 1] * p.disp_h / (float)p.net_h;
          o[0] = peaks[idx - 2] * p.disp_w / (float)p.net_w;
        } else {
          o[0] = 0; o[1] = 0; o[2] = 0;
        }
      }
    }
    __syncthreads();
    out += total;
  }
  if (tid == 0) *p.num_people = out < p.max_people ? out : p.max_people;
}
__global__ __launch_bounds__(256) void connect_assemble_kernel(ConnectParams p) {
  const KStamp kstamp_(p.stamp ? p.stamp + 4 : nullptr);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  connect_assemble_body(p, lds_raw);
}

#ifdef RTP_EXPERIMENTS
// The whole connect chain in ONE launch (experiments build, RTP_CHAIN_CONNECT=1; round 4): the grid of the pair kernel.  The LAST workgroup of a limb to
// finish its pairs (a ticket per limb) orders that limb's candidates and runs its greedy assignment with its first four waves (the other
// twelve retire: s_barrier only waits for the surviving waves of a workgroup), and the last LIMB to finish assembles the people.  Two
// launches and their dependency latencies less per frame; the arithmetic is the three kernels', called as they are.
// Visibility: every producer ends with __threadfence() before its ticket (release), every consumer starts with one after it (acquire:
// nothing of the produced data was read by this workgroup before, and the vector L1 is invalidated by the fence).
__global__ __launch_bounds__(PAIRS_WG) void connect_chain_kernel(ConnectParams p, ResizeParams r, int stage) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int s_last;
  connect_pairs_body<true>(p, r, stage, lds_raw);
  // release: every thread's stores have left the CU (vmcnt 0), the barrier collects them, ONE thread issues the device-scope fence (on a
  // multi-XCD chip that is an L2 write-back: 1024 threads x 76 workgroups doing it each was the first version's 24 us) and takes the ticket
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence(); s_last = atomicAdd(&p.tickets[blockIdx.y], 1) == (int)gridDim.x - 1; if (s_last) __threadfence(); }
  __syncthreads();
  if (!s_last || threadIdx.x >= 256) return;   // (uniform per wave)
  connect_match_limb(p, blockIdx.y, lds_raw);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence(); s_last = atomicAdd(&p.tickets[p.num_limbs], 1) == p.num_limbs - 1; if (s_last) __threadfence(); }
  __syncthreads();
  if (!s_last) return;
  if ((int)threadIdx.x <= p.num_limbs) p.tickets[threadIdx.x] = 0;   // ready for the next frame (nobody else is alive in this launch)
  connect_assemble_body(p, lds_raw);
}
#endif

static hipError_t launch_connect_impl(const ConnectParams& p, const ResizeParams* r, hipStream_t stream) {
  hipError_t e = p.counter_cleared ? hipSuccess : hipMemsetAsync(p.num_people, 0, sizeof(int), stream);
  if (e != hipSuccess) return e;
  const int cap = p.max_peaks * p.max_peaks;
  size_t n2 = 64;
  while ((int)n2 < cap) n2 <<= 1;
  const size_t lds1 = n2 * sizeof(unsigned long long);
  size_t lds2 = (size_t)p.max_rows * (sizeof(double) + sizeof(short) + sizeof(short) * p.num_parts);
  if (p.max_peaks > 127) return hipErrorInvalidValue;
  ConnectParams pa = p;
  {
    const size_t extra = 8 + ((size_t)p.num_parts * 3 * (p.max_peaks + 1) + (size_t)p.num_limbs * p.max_peaks * 3 + p.num_limbs) * 4;
    static const char* np = RTP_EXP_ENV("RTP_ASSEMBLE_PRELOAD");  // experiments: 0 = no LDS copy of the assembly inputs
    pa.assemble_preload = (lds2 + extra <= kPostDynLdsMax && !(np && np[0] == '0')) ? 1 : 0;
    if (pa.assemble_preload) lds2 += extra;
  }
  if (lds1 > kPostDynLdsMax || lds2 > kPostDynLdsMax) return hipErrorInvalidValue;   // (static LDS of the kernels is inside the reserve)
  if (lds1 > 64 * 1024) {
    e = ensure_lds<connect_match_kernel>(lds1);
    if (e != hipSuccess) return e;
  }
  if (lds2 > 64 * 1024) {
    e = ensure_lds<connect_assemble_kernel>(lds2);
    if (e != hipSuccess) return e;
  }
#ifdef RTP_EXPERIMENTS
  if (r && p.tickets && p.counter_cleared && !p.diag_stages) {   // pairs -> match -> assemble in one launch
    const size_t lds0 = (size_t)2 * r->num * r->h * r->w * sizeof(float);
    const int stage = lds0 <= 96 * 1024 ? 1 : 0;
    const size_t lds = std::max(std::max(stage ? lds0 : (size_t)0, lds1), lds2);
    if (lds > 64 * 1024) {
      e = ensure_lds<connect_chain_kernel>(lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(connect_chain_kernel, dim3((cap + PAIRS_WG - 1) / PAIRS_WG, p.num_limbs), dim3(PAIRS_WG), lds, stream, pa, *r, stage);
    return hipGetLastError();
  }
#endif
  if (r) {
    const size_t lds0 = (size_t)2 * r->num * r->h * r->w * sizeof(float);
    const int stage = lds0 <= 96 * 1024 ? 1 : 0;
    if (stage && lds0 > 64 * 1024) {
      e = ensure_lds<connect_pairs_kernel<true>>(lds0);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(connect_pairs_kernel<true>, dim3((cap + PAIRS_WG - 1) / PAIRS_WG, p.num_limbs), dim3(PAIRS_WG), stage ? lds0 : 0, stream, p, *r, stage);
  } else hipLaunchKernelGGL(connect_pairs_kernel<false>, dim3((cap + PAIRS_WG - 1) / PAIRS_WG, p.num_limbs), dim3(PAIRS_WG), 0, stream, p, ResizeParams{}, 0);
  e = hipGetLastError();
  if (e != hipSuccess || p.diag_stages == 1) return e;
  static const char* mw = RTP_EXP_ENV("RTP_MATCH_WGS");  // experiments: workgroups of the match kernel (default: one per limb)
  int match_wgs = mw ? atoi(mw) : p.num_limbs;
  if (match_wgs < 1 || match_wgs > p.num_limbs) match_wgs = p.num_limbs;
  hipLaunchKernelGGL(connect_match_kernel, dim3(match_wgs), dim3(256), lds1, stream, p);
  e = hipGetLastError();
  if (e != hipSuccess || p.diag_stages == 2) return e;
  hipLaunchKernelGGL(connect_assemble_kernel, dim3(1), dim3(256), lds2, stream, pa);
  return hipGetLastError();
}

hipError_t launch_connect(const ConnectParams& p, hipStream_t stream) { return launch_connect_impl(p, nullptr, stream); }
hipError_t launch_connect_fused(const ConnectParams& p, const ResizeParams& r, hipStream_t stream) { return launch_connect_impl(p, &r, stream); }

}  // namespace rtp
