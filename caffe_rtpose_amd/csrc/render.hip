// render.hip — SURVEY.md §8(f)-3: the reference's pose overlay (renderFunctions.cu:124-240 MPI,
// :394-636 COCO; part_to_show == 0) on the display-resolution frame, followed by
// postProcessFrame's float -> u8 conversion (rtpose.cpp:1286-1293).  Oracle: orc_render_pose.
//
// MI355X-first shape, not the reference's: the reference evaluates atan2f/sinf/cosf of every limb
// of every person in EVERY pixel thread and keeps a float canvas (11 MB at 720p) that travels
// H2D and back.  Here a one-workgroup prep kernel computes the per-person box / scale and the
// per-limb ellipse frame once (same inputs -> the same values every pixel would have computed), and
// the pixel kernel reads the u8 display image the pre-processing kernels already left on the
// device, blends in registers with the reference's arithmetic (C's usual conversions: several MPI
// blends are double) and writes u8.  Built with the BITEXACT flags (no FMA contraction).
#include "kernels.h"

namespace rtp {

__constant__ int kRColorCoco[18 * 3] = {255, 0, 0, 255, 85, 0, 255, 170, 0, 255, 255, 0, 170, 255, 0, 85, 255, 0, 0, 255, 0, 0, 255, 85,
                                        0, 255, 170, 0, 255, 255, 0, 170, 255, 0, 85, 255, 0, 0, 255, 85, 0, 255, 170, 0, 255, 255, 0, 255,
                                        255, 0, 170, 255, 0, 85};
__constant__ int kRColorMpi[9 * 3] = {255, 0, 0, 255, 170, 0, 170, 255, 0, 0, 255, 0, 0, 255, 170, 0, 170, 255, 0, 0, 255, 170, 0, 255, 255, 0, 170};
__constant__ int kRLimbCoco[17 * 2] = {1, 2, 1, 5, 2, 3, 3, 4, 5, 6, 6, 7, 1, 8, 8, 9, 9, 10, 1, 11, 11, 12, 12, 13, 1, 0, 0, 14, 14, 16, 0, 15, 15, 17};
__constant__ int kRLimbMpi[9 * 2] = {0, 1, 2, 3, 3, 4, 5, 6, 6, 7, 8, 9, 9, 10, 11, 12, 12, 13};

// table layout (floats): [0] = number of people; person p at 8 + p * RTAB_PERSON:
//   [0..3] box min x, min y, max x, max y   [4] scale   [8 + l*6 ..] limb l: valid, mid_x, mid_y, dir_s, dir_c, half_len2
#define RTAB_PERSON (8 + 17 * 6)

__global__ __launch_bounds__(128) void render_prep_kernel(RenderParams p) {
  int n = *p.num_people;
  n = n < 0 ? 0 : (n > p.max_people ? p.max_people : n);
  if (threadIdx.x == 0) p.tab[0] = (float)n;
  const bool coco = p.model == 0;
  const int NP = coco ? 18 : 15, NL = coco ? 17 : 9;
  const float threshold = coco ? 0.01f : 0.0f;
  const int* limb = coco ? kRLimbCoco : kRLimbMpi;
  for (int q = threadIdx.x; q < n; q += blockDim.x) {
    float* t = p.tab + 8 + (size_t)q * RTAB_PERSON;
    const float* ps = p.poses + (size_t)q * NP * 3;
    float mnx = p.w, mny = p.h, mxx = 0, mxy = 0, sx = 1.f;
    if (coco) {
      for (int part = 0; part < NP; part++) {
        const float x = ps[part * 3], y = ps[part * 3 + 1], z = ps[part * 3 + 2];
        if (z > threshold) {
          if (x < mnx) mnx = x;
          if (x > mxx) mxx = x;
          if (y < mny) mny = y;
          if (y > mxy) mxy = y;
        }
      }
      sx = mxx - mnx;
      const float sy = mxy - mny;
      sx = (sx + sy) / 2.0;
      if (sx < 200) {
        sx = sx / 200;
        if (sx < 0.33) sx = 0.33;
      } else {
        sx = 1.0;
      }
      mxx += 50; mxy += 50; mnx -= 50; mny -= 50;
    }
    t[0] = mnx; t[1] = mny; t[2] = mxx; t[3] = mxy; t[4] = sx;
    for (int l = 0; l < NL; l++) {
      const int a = limb[2 * l], b = limb[2 * l + 1];
      const float x_a = ps[a * 3], x_b = ps[b * 3], y_a = ps[a * 3 + 1], y_b = ps[b * 3 + 1];
      float* o = t + 8 + l * 6;
      if (ps[a * 3 + 2] > threshold && ps[b * 3 + 2] > threshold) {
        const float mid_x = (x_a + x_b) / 2, mid_y = (y_a + y_b) / 2;
        const float angle = atan2f(y_b - y_a, x_b - x_a);
        o[0] = 1.f; o[1] = mid_x; o[2] = mid_y; o[3] = sinf(angle); o[4] = cosf(angle);
        o[5] = (x_a - mid_x) * (x_a - mid_x) + (y_a - mid_y) * (y_a - mid_y);
      } else {
        o[0] = 0.f;
      }
    }
  }
}

__global__ __launch_bounds__(256) void render_pose_kernel(RenderParams p) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.w || y >= p.h) return;
  const unsigned char* s = p.src + ((size_t)y * p.w + x) * 3;
  float b = s[0], g = s[1], r = s[2];
  const int n = (int)p.tab[0];
  const bool coco = p.model == 0;
  if (coco) {
    const float threshold = 0.01f;
    const float radius = 2 * p.h / 200.0f;
    const float limb_width = p.h / 120.0f;
    for (int q = 0; q < n; q++) {
      const float* t = p.tab + 8 + (size_t)q * RTAB_PERSON;
      if (x > t[2] || x < t[0] || y > t[3] || y < t[1]) continue;
      const float sc = t[4];
      const float* ps = p.poses + (size_t)q * 18 * 3;
      for (int l = 0; l < 17; l++) {
        const float* o = t + 8 + l * 6;
        if (o[0] == 0.f) continue;
        const float half_wid2 = sc * sc * limb_width * limb_width;
        const float alpha = 0.5;
        const float mid_x = o[1], mid_y = o[2], dir_s = o[3], dir_c = o[4], half_len2 = o[5];
        const float A = dir_c * (x - mid_x) + dir_s * (y - mid_y);
        const float B = dir_s * (x - mid_x) - dir_c * (y - mid_y);
        const float ellipse_lhs = A * A / half_len2 + B * B / half_wid2;
        if (ellipse_lhs >= 0 && ellipse_lhs <= 1) {
          b = (1 - alpha) * b + alpha * kRColorCoco[l * 3 + 2];
          g = (1 - alpha) * g + alpha * kRColorCoco[l * 3 + 1];
          r = (1 - alpha) * r + alpha * kRColorCoco[l * 3 + 0];
        }
      }
      for (int i = 0; i < 18; i++) {
        const float jx = ps[i * 3], jy = ps[i * 3 + 1], value = ps[i * 3 + 2];
        if (value > threshold) {
          const float d2 = (x - jx) * (x - jx) + (y - jy) * (y - jy);
          float ring_lo2 = 0;
          float ring_hi2 = sc * sc * radius * radius;
          float alpha = 0.6;
          float cx = kRColorCoco[i * 3 + 0], cy = kRColorCoco[i * 3 + 1], cz = kRColorCoco[i * 3 + 2];
          if (p.googly && (i == 14 || i == 15)) {
            ring_hi2 = sc * sc * 2.5 * 2.5 * radius * radius;
            ring_lo2 = sc * sc * (2.5 * radius - 2) * (2.5 * radius - 2);
            alpha = 0.9;
            cx = 0; cy = 0; cz = 0;
            if (d2 <= ring_hi2) {
              if (d2 <= ring_lo2) { cx = 255; cy = 255; cz = 255; }
              if (d2 <= ring_lo2 * 0.6) {
                const float d2_pupil = (x - 4 - jx) * (x - 4 - jx) + (y - jy + 4) * (y - jy + 4);
                if (d2_pupil > 3.75 * 3.75) { cx = 0; cy = 0; cz = 0; }
              }
              b = (1 - alpha) * b + alpha * cz;
              g = (1 - alpha) * g + alpha * cy;
              r = (1 - alpha) * r + alpha * cx;
            }
          } else if (d2 >= ring_lo2 && d2 <= ring_hi2) {
            b = (1 - alpha) * b + alpha * cz;
            g = (1 - alpha) * g + alpha * cy;
            r = (1 - alpha) * r + alpha * cx;
          }
        }
      }
    }
  } else {
    const float threshold = 0.0f;
    const float radius = 3 * p.h / 200.0f;
    const float limb_width = p.h / 60.0f;
    for (int q = 0; q < n; q++) {
      const float* t = p.tab + 8 + (size_t)q * RTAB_PERSON;
      const float* ps = p.poses + (size_t)q * 15 * 3;
      for (int l = 0; l < 9; l++) {
        const float* o = t + 8 + l * 6;
        if (o[0] == 0.f) continue;
        float half_wid2 = limb_width * limb_width;
        const float alpha = 0.6;
        const float mid_x = o[1], mid_y = o[2], dir_s = o[3], dir_c = o[4];
        float half_len2 = o[5];
        if (l == 0) {
          half_len2 *= 1.2;
          half_wid2 = half_len2;
        }
        const float A = dir_c * (x - mid_x) + dir_s * (y - mid_y);
        const float B = dir_s * (x - mid_x) - dir_c * (y - mid_y);
        const float ellipse_lhs = A * A / half_len2 + B * B / half_wid2;
        float minV = 0;
        if (l == 0) minV = 0.8;
        if (ellipse_lhs >= minV && ellipse_lhs <= 1) {
          b = (1 - alpha) * b + alpha * kRColorMpi[l * 3 + 2];
          g = (1 - alpha) * g + alpha * kRColorMpi[l * 3 + 1];
          r = (1 - alpha) * r + alpha * kRColorMpi[l * 3];
        }
      }
      for (int i = 0; i < 15; i++) {
        const float px = ps[i * 3], py = ps[i * 3 + 1], value = ps[i * 3 + 2];
        if (value > threshold) {
          if ((x - px) * (x - px) + (y - py) * (y - py) <= radius * radius) {
            b = 0.6 * b + 0.4 * kRColorMpi[(i % 9) * 3 + 2];
            g = 0.6 * g + 0.4 * kRColorMpi[(i % 9) * 3 + 1];
            r = 0.6 * r + 0.4 * kRColorMpi[(i % 9) * 3];
          }
        }
      }
    }
  }
  unsigned char* o = p.dst + ((size_t)y * p.w + x) * 3;
  const float v3[3] = {b, g, r};
#pragma unroll
  for (int c = 0; c < 3; c++) {
    int value = int(v3[c] + 0.5);
    value = value < 0 ? 0 : (value > 255 ? 255 : value);
    o[c] = (unsigned char)value;
  }
}

hipError_t launch_render(const RenderParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(render_prep_kernel, dim3(1), dim3(128), 0, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(render_pose_kernel, dim3((p.w + 63) / 64, (p.h + 3) / 4), dim3(256), 0, stream, p);
  return hipGetLastError();
}

size_t render_tab_floats(int max_people) { return 8 + (size_t)max_people * RTAB_PERSON; }

// ---------------------------------------------------------------------------------------------------------------
// The --part_to_show views (renderFunctions.cu:242-329 MPI heat map, :638-724 COCO heat map, :726-836 all COCO parts,
// :838-975 COCO PAFs; colour maps :12-109, cubic :111-120; dispatch rtpose.cpp:270-299 + renderFunctions.cu:331-389,
// :978-1080).  One kernel for all of them, u8 in / u8 out like the pose overlay; every expression keeps the reference's
// operand types (float / double literals), built without FMA contraction: bit-identical to the reference's source
// semantics except atan2 of the PAF view (ocml vs the host libm, then rounded to float).  Oracle: orc_render_view.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void heat_colour(float* c, float v, float vmin, float vmax) {
  c[0] = c[1] = c[2] = 255;
  if (v < vmin) v = vmin;
  if (v > vmax) v = vmax;
  const float dv = vmax - vmin;
  if (v < (vmin + 0.125 * dv)) {
    c[0] = 256 * (0.5 + (v * 4));
    c[1] = c[2] = 0;
  } else if (v < (vmin + 0.375 * dv)) {
    c[0] = 255;
    c[1] = 256 * (v - 0.125) * 4;
    c[2] = 0;
  } else if (v < (vmin + 0.625 * dv)) {
    c[0] = 256 * (-4 * v + 2.5);
    c[1] = 255;
    c[2] = 256 * (4 * (v - 0.375));
  } else if (v < (vmin + 0.875 * dv)) {
    c[0] = 0;
    c[1] = 256 * (-4 * v + 3.5);
    c[2] = 255;
  } else {
    c[0] = 0;
    c[1] = 0;
    c[2] = 256 * (-4 * v + 4.5);
  }
}
__device__ __forceinline__ void hue_colour(float* c, float v) {  // 55-step colour wheel, v already in [0, 1]
  if (v < 0.f) v = 0.f;
  if (v > 1.f) v = 1.f;
  v = 55 * v;
  if (v < 15) {
    c[0] = 255; c[1] = 255 * (v / (15)); c[2] = 0;
  } else if (v < 21) {
    c[0] = 255 - 255 * ((v - 15) / (6)); c[1] = 255; c[2] = 0;
  } else if (v < 25) {
    c[0] = 0; c[1] = 255; c[2] = 255 * ((v - 15 - 6) / (4));
  } else if (v < 36) {
    c[0] = 0; c[1] = 255 - 255 * ((v - 15 - 6 - 4) / (11)); c[2] = 255;
  } else if (v < 49) {
    c[0] = 255 * ((v - 15 - 6 - 4 - 11) / (13)); c[1] = 0; c[2] = 255;
  } else if (v < 55) {
    c[0] = 255; c[1] = 0; c[2] = 255 - 255 * ((v - 15 - 6 - 4 - 11 - 13) / (6));
  } else {
    c[0] = 255; c[1] = 0; c[2] = 0;
  }
}
__device__ __forceinline__ void paf_colour(float* c, float vx, float vy) {
  float rad = sqrtf(vx * vx + vy * vy);
  const float a = atan2((double)(-vy), (double)(-vx)) / 3.14159265358979323846;
  float fk = (a + 1) / 2.0;
  if (fk != fk) fk = 0;
  if (rad > 1) rad = 1;
  hue_colour(c, fk);
  c[0] = 255 * (rad * (c[0] / 255));
  c[1] = 255 * (rad * (c[1] / 255));
  c[2] = 255 * (rad * (c[2] / 255));
}
__device__ __forceinline__ float view_cubic(float v0, float v1, float v2, float v3, float dx) {
  return (-0.5f * v0 + 1.5f * v1 - 1.5f * v2 + 0.5f * v3) * dx * dx * dx + (v0 - 2.5f * v1 + 2.0 * v2 - 0.5 * v3) * dx * dx + (-0.5f * v0 + 0.5f * v2) * dx + v1;
}
struct ViewTap {
  bool inside;
  int xn[4], yn[4];
  float dx, dy;
};
__device__ __forceinline__ int nb_axis(float pos, int n, int* nb, float* frac) {
  nb[1] = int(pos + 1e-5);
  nb[1] = (nb[1] < 0) ? 0 : nb[1];
  nb[0] = (nb[1] - 1 < 0) ? nb[1] : (nb[1] - 1);
  nb[2] = (nb[1] + 1 >= n) ? (n - 1) : (nb[1] + 1);
  nb[3] = (nb[2] + 1 >= n) ? (n - 1) : (nb[2] + 1);
  *frac = pos - nb[1];
  return 0;
}
__device__ __forceinline__ float view_bicubic(const float* plane, int nw, const ViewTap& t) {
  float row[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float* q = plane + t.yn[i] * nw;
    row[i] = view_cubic(q[t.xn[0]], q[t.xn[1]], q[t.xn[2]], q[t.xn[3]], t.dx);
  }
  return view_cubic(row[0], row[1], row[2], row[3], t.dy);
}
__device__ __forceinline__ float view_bilinear(const float* plane, int nw, const ViewTap& t) {
  const float a = plane[t.yn[1] * nw + t.xn[1]], b = plane[t.yn[1] * nw + t.xn[2]];
  const float c = plane[t.yn[2] * nw + t.xn[1]], d = plane[t.yn[2] * nw + t.xn[2]];
  return (1 - t.dx) * (1 - t.dy) * a + (t.dx) * (1 - t.dy) * b + (1 - t.dx) * (t.dy) * c + (t.dx) * (t.dy) * d;
}
}  // namespace

__global__ __launch_bounds__(256) void render_view_kernel(RenderViewParams p) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.w || y >= p.h) return;
  const unsigned char* s = p.src + ((size_t)y * p.w + x) * 3;
  float b = s[0], g = s[1], r = s[2];
  ViewTap t;
  {
    const float h_inv = (float)p.net_h / (float)p.h;
    const float w_inv = (float)p.net_w / (float)p.w;
    const float x_on = w_inv * x + (0.5 * w_inv - 0.5);
    const float y_on = h_inv * y + (0.5 * h_inv - 0.5);
    t.inside = x_on >= 0 && x_on < p.net_w && y_on >= 0 && y_on < p.net_h;
    nb_axis(x_on, p.net_w, t.xn, &t.dx);
    nb_axis(y_on, p.net_h, t.yn, &t.dy);
  }
  const long plane = (long)p.net_w * p.net_h;
  const int nw = p.net_w;
  const int shown = p.part_to_show - 1;
  if (p.model != 0) {  // MPI: one map; maps 16.. are PAF channels in [-1, 1]
    float value = (shown == 14) ? 1 : 0;
    if (t.inside) value = view_bicubic(p.maps + shown * plane, nw, t);
    float c[3];
    if (shown < 16) heat_colour(c, value, 0, 1);
    else heat_colour(c, value, -1, 1);
    b = 0.5 * b + 0.5 * c[0];
    g = 0.5 * g + 0.5 * c[1];
    r = 0.5 * r + 0.5 * c[2];
  } else if (shown == 18) {  // every part map in its own colour, nearest sample
    float c[3] = {0, 0, 0};
    if (t.inside) {
      const int o = t.yn[1] * nw + t.xn[1];
      for (int part = 0; part < 18; part++) {
        const float value = p.maps[part * plane + o];
        c[0] += value * kRColorCoco[part * 3 + 0];
        c[1] += value * kRColorCoco[part * 3 + 1];
        c[2] += value * kRColorCoco[part * 3 + 2];
      }
    }
    const float alpha = 0.7;
    b = (1 - alpha) * b + alpha * c[2];
    g = (1 - alpha) * g + alpha * c[1];
    r = (1 - alpha) * r + alpha * c[0];
  } else if (shown < 18) {  // one part map
    float value = (shown == 17) ? 1 : 0;
    if (t.inside) value = view_bicubic(p.maps + shown * plane, nw, t);
    float c[3];
    heat_colour(c, value, 0, 1);
    const float alpha = 0.7;
    b = (1 - alpha) * b + alpha * c[2];
    g = (1 - alpha) * g + alpha * c[1];
    r = (1 - alpha) * r + alpha * c[0];
  } else {  // PAFs: part_to_show 20 = all 19 limbs (nearest sample), 21.. = one limb (2x2 blend)
    int first = (shown - 18 - 1) * 2;
    int count = 1;
    if (first == 0) count = 19;
    else first = first - 2;
    first += 1 + 18;
    float c[3] = {0, 0, 0};
    if (t.inside) {
      for (int ch = first; ch < first + count * 2; ch += 2) {
        float vx, vy;
        if (count == 1) {
          vx = view_bilinear(p.maps + ch * plane, nw, t);
          vy = view_bilinear(p.maps + (ch + 1) * plane, nw, t);
        } else {
          vx = p.maps[ch * plane + t.yn[1] * nw + t.xn[1]];
          vy = p.maps[(ch + 1) * plane + t.yn[1] * nw + t.xn[1]];
        }
        float c2[3];
        paf_colour(c2, vx, vy);
        c[0] += c2[0];
        c[1] += c2[1];
        c[2] += c2[2];
      }
    }
    if (c[0] > 255) c[0] = 255;
    if (c[1] > 255) c[1] = 255;
    if (c[2] > 255) c[2] = 255;
    const float alpha = 0.7;
    b = (1 - alpha) * b + alpha * c[2];
    g = (1 - alpha) * g + alpha * c[1];
    r = (1 - alpha) * r + alpha * c[0];
  }
  unsigned char* o = p.dst + ((size_t)y * p.w + x) * 3;
  const float v3[3] = {b, g, r};
#pragma unroll
  for (int c = 0; c < 3; c++) {
    int value = int(v3[c] + 0.5);
    value = value < 0 ? 0 : (value > 255 ? 255 : value);
    o[c] = (unsigned char)value;
  }
}

hipError_t launch_render_view(const RenderViewParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(render_view_kernel, dim3((p.w + 63) / 64, (p.h + 3) / 4), dim3(256), 0, stream, p);
  return hipGetLastError();
}

}  // namespace rtp
