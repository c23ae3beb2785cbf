// preproc.hip — row a1 on the GPU: from the decoded u8 BGR frame to the net input,
// bit-identical to the host restatement in preprocess.cpp (which restates OpenCV's warpAffine
// INTER_CUBIC / resize INTER_AREA as used by rtpose.cpp:322-368):
//   warp_cubic_kernel : display-fit scale, 1/32-pixel fixed-point coordinates, OpenCV's 2-D table of 15-bit cubic weights,
//                       one rounding per pixel, BORDER_CONSTANT 0 — pure integer arithmetic.
//   area_pad_kernel   : per scale, fractional-area resize (float accumulation in the host's order:
//                       x-sum per source row, then beta-weighted row sum), round-to-nearest-even — or, where the level ENLARGES
//                       an axis, OpenCV's fixed-point bilinear kernel with area-mode coefficients —,
//                       u8/256 - 0.5, centre zero-pad into the net frame (process_and_pad_image).
// Compiled with -ffp-contract=off.  HBM-bound streaming kernels: 2.8 MB in, 11.6 MB out per 720p frame.
#include <cstring>

#include "kernels.h"

// built with the BITEXACT flags of csrc/Makefile (no packed-f32 VALU; see postproc.hip)

namespace rtp {

__global__ __launch_bounds__(256) void warp_cubic_kernel(unsigned long long* stamp, const unsigned char* __restrict__ src, int sw, int sh, double inv,
                                                         const short* __restrict__ tab2d, unsigned char* __restrict__ dst, int dw, int dh) {
  const KStamp kstamp_(stamp);
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= dw) return;
  const int INTER_BITS = 5, INTER_TAB = 1 << INTER_BITS, AB_BITS = 10, AB_SCALE = 1 << AB_BITS, COEF_BITS = 15;
  const int round_delta = AB_SCALE / INTER_TAB / 2;
  const int Y0 = (int)__double2ll_rn((inv * y) * AB_SCALE) + round_delta;
  const int Y = Y0 >> (AB_BITS - INTER_BITS);
  const int sy = (Y >> INTER_BITS) - 1, fy = Y & (INTER_TAB - 1);
  const int X = (round_delta + (int)__double2ll_rn(inv * x * AB_SCALE)) >> (AB_BITS - INTER_BITS);
  const int sx = (X >> INTER_BITS) - 1, fx = X & (INTER_TAB - 1);
  const short* w = tab2d + ((size_t)fy * 32 + fx) * 16;   // remapBicubic: 16 fixed-point weights, one rounding per pixel
  int acc[3] = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int yy = sy + r;
    if (yy < 0 || yy >= sh) continue;   // BORDER_CONSTANT, value 0
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int xx = sx + q;
      if (xx < 0 || xx >= sw) continue;
      const unsigned char* p = src + ((size_t)yy * sw + xx) * 3;
      const int wv = w[r * 4 + q];
      acc[0] += p[0] * wv; acc[1] += p[1] * wv; acc[2] += p[2] * wv;
    }
  }
  unsigned char* o = dst + ((size_t)y * dw + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int v = (acc[c] + (1 << (COEF_BITS - 1))) >> COEF_BITS;
    o[c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

// One thread per pixel of the (net_w x net_h) frame of scale `blockIdx.z`.
__global__ __launch_bounds__(256) void area_pad_kernel(unsigned long long* stamp, const unsigned char* __restrict__ disp, int dw, int dh, AreaScale sc0, AreaScale sc1,
                                                       AreaScale sc2, AreaScale sc3, int nscales_in_launch, float* __restrict__ out, int net_w,
                                                       int net_h, int scale_base) {
  const KStamp kstamp_(stamp);
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int si = blockIdx.z;
  if (x >= net_w || si >= nscales_in_launch) return;
  const AreaScale sc = si == 0 ? sc0 : (si == 1 ? sc1 : (si == 2 ? sc2 : sc3));
  const int padw = (net_w - sc.tw) / 2, padh = (net_h - sc.th) / 2;
  float* o = out + (size_t)(scale_base + si) * 3 * net_h * net_w + (size_t)y * net_w + x;
  const size_t plane = (size_t)net_h * net_w;
  const int ox = x - padw, oy = y - padh;
  if (ox < 0 || ox >= sc.tw || oy < 0 || oy >= sc.th) { o[0] = 0.f; o[plane] = 0.f; o[2 * plane] = 0.f; return; }
  float r3[3];
  if (sc.identity) {
    const unsigned char* p = disp + ((size_t)oy * dw + ox) * 3;
    r3[0] = p[0]; r3[1] = p[1]; r3[2] = p[2];
  } else if (sc.linear) {  // an enlarged axis: HResizeLinear into int rows, VResizeLinear's u8 specialisation (integer arithmetic throughout)
    const int* tx = sc.lx + ox * 4;
    const int* ty = sc.ly + oy * 4;
    const unsigned char* r0 = disp + (size_t)ty[0] * dw * 3;
    const unsigned char* r1 = disp + (size_t)ty[1] * dw * 3;
    const int x0 = tx[0] * 3, x1 = tx[1] * 3, a0 = tx[2], a1 = tx[3], b0 = ty[2], b1 = ty[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int d0 = r0[x0 + c] * a0 + r0[x1 + c] * a1;
      const int d1 = r1[x0 + c] * a0 + r1[x1 + c] * a1;
      r3[c] = (float)(unsigned char)((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2);
    }
  } else if (sc.fast_x > 0) {  // resizeAreaFast_: integer scale; 2x2 rounds half up, other areas go through float * (1.f/area)
    int sum[3] = {0, 0, 0};
    for (int yy = 0; yy < sc.fast_y; ++yy)
      for (int xx = 0; xx < sc.fast_x; ++xx) {
        const unsigned char* p = disp + ((size_t)(oy * sc.fast_y + yy) * dw + ox * sc.fast_x + xx) * 3;
        sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2];
      }
    const float inv_area = 1.f / (float)(sc.fast_x * sc.fast_y);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int v = (sc.fast_x == 2 && sc.fast_y == 2) ? ((sum[c] + 2) >> 2) : (int)rintf((float)sum[c] * inv_area);
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      r3[c] = (float)v;
    }
  } else {
    const int xs = sc.xstart[ox], xe = sc.xstart[ox + 1];
    const int ys = sc.ystart[oy], ye = sc.ystart[oy + 1];
    float sum[3] = {0.f, 0.f, 0.f};
    for (int yi = ys; yi < ye; ++yi) {
      const unsigned char* srow = disp + (size_t)sc.ysi[yi] * dw * 3;
      const float beta = sc.yalpha[yi];
      float b[3] = {0.f, 0.f, 0.f};
      for (int xi = xs; xi < xe; ++xi) {
        const unsigned char* p = srow + sc.xsi[xi] * 3;
        const float a = sc.xalpha[xi];
        b[0] += p[0] * a; b[1] += p[1] * a; b[2] += p[2] * a;
      }
      sum[0] += beta * b[0]; sum[1] += beta * b[1]; sum[2] += beta * b[2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int v = (int)rintf(sum[c]);  // cvRound: round half to even
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      r3[c] = (float)v;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c * plane] = r3[c] / 256.0f - 0.5f;
}

hipError_t launch_warp(unsigned long long* stamp, const unsigned char* src, int sw, int sh, double inv, const short* tab2d, unsigned char* dst, int dw, int dh,
                       hipStream_t stream) {
  dim3 grid((dw + 255) / 256, dh);
  hipLaunchKernelGGL(warp_cubic_kernel, grid, dim3(256), 0, stream, stamp, src, sw, sh, inv, tab2d, dst, dw, dh);
  return hipGetLastError();
}

hipError_t launch_area_pad(unsigned long long* stamp, const unsigned char* disp, int dw, int dh, const AreaScale* scales, int nscales, float* out, int net_w, int net_h,
                           hipStream_t stream) {
  for (int base = 0; base < nscales; base += 4) {
    const int n = nscales - base < 4 ? nscales - base : 4;
    AreaScale z;
    memset(&z, 0, sizeof z);
    const AreaScale& a = scales[base];
    const AreaScale& b = n > 1 ? scales[base + 1] : z;
    const AreaScale& c = n > 2 ? scales[base + 2] : z;
    const AreaScale& d = n > 3 ? scales[base + 3] : z;
    dim3 grid((net_w + 255) / 256, net_h, n);
    hipLaunchKernelGGL(area_pad_kernel, grid, dim3(256), 0, stream, stamp, disp, dw, dh, a, b, c, d, n, out, net_w, net_h, base);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace rtp
