// preprocess.cpp — host side of row a1: what getFrameFromDir/getFrameFromCam do to a
// decoded frame before it reaches the net (rtpose.cpp:322-368, 474-518):
//   display-fit scale -> warpAffine(INTER_CUBIC, BORDER_CONSTANT 0) to the display resolution ->
//   per scale: cv::resize(INTER_AREA) to 16*ceil(net*s/16) -> process_and_pad_image(normalize=1).
// OpenCV is a third-party dependency of the reference that is absent here (and whose version the
// reference does not pin: Makefile:197-202): the two OpenCV primitives below restate OpenCV's
// published algorithms (imgproc/imgwarp.cpp warpAffine + remapBicubic 8u: 1/32-pixel fixed-point
// coordinates, the 2-D 15-bit fixed-point weight table of initInterTab2D with A = -0.75, ONE rounding
// per pixel; resize: scale = 1/(dsize/ssize), resizeAreaFast_ for integer scales, else
// computeResizeAreaTab + resizeArea_: fractional-area weights, float accumulation, round half to even).
// Third-party arithmetic: pinned only by tests/_cvref.py, an independent numpy restatement of the same
// published algorithm (tests/test_preprocess_cli.py, tests/test_gpu_parity.py), not by OpenCV itself.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/rtpose_mi355x.h"

bool rtp_internal_area_fast(int sw, int sh, int dw, int dh, int* ix, int* iy);

namespace {

inline int cv_round(double v) { return (int)std::nearbyint(v); }  // cvRound: round half to even
inline unsigned char sat_u8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---- cv::resize(..., INTER_AREA), 8UC3, shrinking in both directions -------------------------
struct AreaTab { int di, si; float alpha; };
void area_tab(int ssize, int dsize, double scale, std::vector<AreaTab>& tab) {
  tab.clear();
  for (int dx = 0; dx < dsize; dx++) {
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cell = std::min(scale, ssize - fsx1);
    int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    if (sx1 - fsx1 > 1e-3) tab.push_back({dx, sx1 - 1, (float)((sx1 - fsx1) / cell)});
    for (int sx = sx1; sx < sx2; sx++) tab.push_back({dx, sx, (float)(1.0 / cell)});
    if (fsx2 - sx2 > 1e-3) tab.push_back({dx, sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
  }
}

// cv::resize(INTER_AREA) when an axis is ENLARGED (display resolution smaller than a pyramid level): OpenCV implements true area
// interpolation only for scale_x >= 1 && scale_y >= 1 and otherwise runs its bilinear kernel with `area_mode` coefficients:
//   sx = cvFloor(dx * scale); fx = (float)((dx + 1) - (sx + 1) * inv_scale); fx = fx <= 0 ? 0 : fx - cvFloor(fx)
// (source index clamped at both ends with fx = 0), 11-bit fixed-point weights saturate_cast<short>(w * 2048), HResizeLinear into int
// rows and the u8 VResizeLinear  dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  Restated independently in
// tests/_cvref.py; the device path (preproc.hip) uses the tables built here.
void linear_area_tab(int ssize, int dsize, std::vector<int>* tab /* [dsize][4]: s0, s1, a0, a1 */) {
  const double inv_scale = (double)dsize / ssize, scale = 1.0 / inv_scale;
  tab->assign((size_t)dsize * 4, 0);
  for (int d = 0; d < dsize; ++d) {
    int sx = (int)std::floor(d * scale);
    float fx = (float)((d + 1) - (sx + 1) * inv_scale);
    fx = fx <= 0 ? 0.f : fx - std::floor(fx);
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
    const float c0 = 1.f - fx, c1 = fx;
    auto sat_short = [](float v) { const int q = cv_round(v); return q < -32768 ? -32768 : (q > 32767 ? 32767 : q); };
    int* t = tab->data() + (size_t)d * 4;
    t[0] = sx; t[1] = std::min(sx + 1, ssize - 1); t[2] = sat_short(c0 * 2048.f); t[3] = sat_short(c1 * 2048.f);
  }
}

void resize_linear_u8(const unsigned char* src, int sw, int sh, unsigned char* dst, int dw, int dh) {
  std::vector<int> xt, yt;
  linear_area_tab(sw, dw, &xt);
  linear_area_tab(sh, dh, &yt);
  for (int y = 0; y < dh; ++y) {
    const int* ty = yt.data() + (size_t)y * 4;
    const unsigned char* r0 = src + (size_t)ty[0] * sw * 3;
    const unsigned char* r1 = src + (size_t)ty[1] * sw * 3;
    for (int x = 0; x < dw; ++x) {
      const int* tx = xt.data() + (size_t)x * 4;
      for (int c = 0; c < 3; ++c) {
        const int d0 = r0[tx[0] * 3 + c] * tx[2] + r0[tx[1] * 3 + c] * tx[3];
        const int d1 = r1[tx[0] * 3 + c] * tx[2] + r1[tx[1] * 3 + c] * tx[3];
        dst[((size_t)y * dw + x) * 3 + c] = (unsigned char)((((ty[2] * (d0 >> 4)) >> 16) + ((ty[3] * (d1 >> 4)) >> 16) + 2) >> 2);
      }
    }
  }
}

}  // namespace
// is_area_fast of cv::resize: both scale factors are integers (compared with DBL_EPSILON)
bool rtp_internal_area_fast(int sw, int sh, int dw, int dh, int* ix, int* iy) {
  const double sx = 1.0 / ((double)dw / sw), sy = 1.0 / ((double)dh / sh);
  *ix = (int)std::lrint(sx);
  *iy = (int)std::lrint(sy);
  return std::fabs(sx - *ix) < 2.220446049250313e-16 && std::fabs(sy - *iy) < 2.220446049250313e-16 && *ix >= 1 && *iy >= 1 && !(*ix == 1 && *iy == 1);
}
namespace {
void resize_area_u8(const unsigned char* src, int sw, int sh, unsigned char* dst, int dw, int dh) {
  if (dw == sw && dh == sh) { memcpy(dst, src, (size_t)sw * sh * 3); return; }
  if (dw > sw || dh > sh) { resize_linear_u8(src, sw, sh, dst, dw, dh); return; }
  const double sx = 1.0 / ((double)dw / sw), sy = 1.0 / ((double)dh / sh);  // resize(): scale_x = 1./inv_scale_x
  int ix = 0, iy = 0;
  if (rtp_internal_area_fast(sw, sh, dw, dh, &ix, &iy)) {  // resizeAreaFast_: integer scale on both axes
    const float inv_area = 1.f / (float)(ix * iy);
    for (int dy = 0; dy < dh; ++dy)
      for (int dx = 0; dx < dw; ++dx)
        for (int c = 0; c < 3; ++c) {
          int sum = 0;
          for (int yy = 0; yy < iy; ++yy)
            for (int xx = 0; xx < ix; ++xx) sum += src[((size_t)(dy * iy + yy) * sw + dx * ix + xx) * 3 + c];
          dst[((size_t)dy * dw + dx) * 3 + c] = (ix == 2 && iy == 2) ? (unsigned char)((sum + 2) >> 2) : sat_u8(cv_round((float)sum * inv_area));
        }
    return;
  }
  std::vector<AreaTab> xt, yt;
  area_tab(sw, dw, sx, xt);
  area_tab(sh, dh, sy, yt);
  std::vector<float> buf((size_t)dw * 3), sum((size_t)dw * 3);
  size_t yi = 0;
  for (int dy = 0; dy < dh; ++dy) {
    std::fill(sum.begin(), sum.end(), 0.f);
    for (; yi < yt.size() && yt[yi].di == dy; ++yi) {
      const unsigned char* srow = src + (size_t)yt[yi].si * sw * 3;
      const float beta = yt[yi].alpha;
      std::fill(buf.begin(), buf.end(), 0.f);
      for (const AreaTab& t : xt)
        for (int c = 0; c < 3; ++c) buf[t.di * 3 + c] += srow[t.si * 3 + c] * t.alpha;
      for (int i = 0; i < dw * 3; ++i) sum[i] += beta * buf[i];
    }
    for (int i = 0; i < dw * 3; ++i) dst[(size_t)dy * dw * 3 + i] = sat_u8(cv_round(sum[i]));
  }
}

// ---- cv::warpAffine(src, dst, M = diag(scale), dsize, INTER_CUBIC, BORDER_CONSTANT, 0) ---------
void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

// BicubicTab_i of initInterTab2D(INTER_CUBIC, fixpt): for every (fy, fx) phase pair the 16 weights
// saturate_cast<short>(wy[k1] * wx[k2] * 32768) of the FLOAT 1-D weights; when they do not sum to 32768 the difference
// goes to the largest (sum too small) or the smallest (sum too large) of the four entries k1, k2 in {2, 3}.
void cubic_tab2d(short* tab /* [32][32][16] */) {
  float t1[32][4];
  const float scale = 1.f / 32;
  for (int i = 0; i < 32; ++i) cubic_coeffs(i * scale, t1[i]);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      short* it = tab + ((size_t)i * 32 + j) * 16;
      int isum = 0;
      for (int k1 = 0; k1 < 4; ++k1) {
        const float vy = t1[i][k1];
        for (int k2 = 0; k2 < 4; ++k2) {
          const float v = vy * t1[j][k2];
          const int q = cv_round(v * 32768.f);
          it[k1 * 4 + k2] = (short)(q < -32768 ? -32768 : (q > 32767 ? 32767 : q));
          isum += it[k1 * 4 + k2];
        }
      }
      if (isum != 32768) {
        const int diff = isum - 32768;
        int Mk = 2 * 4 + 2, mk = 2 * 4 + 2;
        for (int k1 = 2; k1 < 4; ++k1)
          for (int k2 = 2; k2 < 4; ++k2) {
            if (it[k1 * 4 + k2] < it[mk]) mk = k1 * 4 + k2;
            else if (it[k1 * 4 + k2] > it[Mk]) Mk = k1 * 4 + k2;
          }
        if (diff < 0) it[Mk] = (short)(it[Mk] - diff);
        else it[mk] = (short)(it[mk] - diff);
      }
    }
}

// warpAffine inverts M = diag(s, s) itself: D = 1/(M0*M4 - M1*M3); A11 = M4*D
double warp_inverse_scale(double s) {
  double D = s * s;
  D = D != 0 ? 1.0 / D : 0;
  return s * D;
}

void warp_scale_cubic_u8(const unsigned char* src, int sw, int sh, double scale, unsigned char* dst, int dw, int dh) {
  const int INTER_BITS = 5, INTER_TAB = 1 << INTER_BITS, AB_BITS = 10, AB_SCALE = 1 << AB_BITS, COEF_BITS = 15;
  static const std::vector<short> tab = [] { std::vector<short> t(32 * 32 * 16); cubic_tab2d(t.data()); return t; }();
  const double inv = warp_inverse_scale(scale);
  const int round_delta = AB_SCALE / INTER_TAB / 2;
  for (int y = 0; y < dh; ++y) {
    const int Y0 = (int)std::lrint((inv * y) * AB_SCALE) + round_delta;
    const int Y = Y0 >> (AB_BITS - INTER_BITS);
    const int sy = (Y >> INTER_BITS) - 1, fy = Y & (INTER_TAB - 1);
    for (int x = 0; x < dw; ++x) {
      const int X = (round_delta + (int)std::lrint(inv * x * AB_SCALE)) >> (AB_BITS - INTER_BITS);  // (X0 + adelta[x]) >> 5, X0 = 16
      const int sx = (X >> INTER_BITS) - 1, fx = X & (INTER_TAB - 1);
      const short* w = tab.data() + ((size_t)fy * 32 + fx) * 16;
      for (int c = 0; c < 3; ++c) {
        int acc = 0;
        for (int r = 0; r < 4; ++r) {
          const int yy = sy + r;
          if (yy < 0 || yy >= sh) continue;  // BORDER_CONSTANT 0
          for (int q = 0; q < 4; ++q) {
            const int xx = sx + q;
            if (xx < 0 || xx >= sw) continue;
            acc += src[((size_t)yy * sw + xx) * 3 + c] * w[r * 4 + q];
          }
        }
        dst[((size_t)y * dw + x) * 3 + c] = sat_u8((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS);  // FixedPtCast<int, uchar, 15>
      }
    }
  }
}

}  // namespace

// shared with the device path (engine.cpp uploads exactly these tables)
void rtp_internal_cubic_tab2d(short* tab) { cubic_tab2d(tab); }
double rtp_internal_warp_inverse_scale(double s) { return warp_inverse_scale(s); }
void rtp_internal_linear_area_table(int ssize, int dsize, std::vector<int>* tab) { linear_area_tab(ssize, dsize, tab); }
int rtp_internal_area_table(int ssize, int dsize, std::vector<int>* start, std::vector<int>* si, std::vector<float>* alpha) {
  if (dsize > ssize) return -1;  // enlarging: the host path falls back to linear interpolation
  std::vector<AreaTab> tab;
  area_tab(ssize, dsize, 1.0 / ((double)dsize / ssize), tab);
  start->assign(dsize + 1, 0);
  si->clear();
  alpha->clear();
  int cur = 0;
  for (size_t i = 0; i < tab.size(); ++i) {
    while (cur <= tab[i].di) (*start)[cur++] = (int)i;
    si->push_back(tab[i].si);
    alpha->push_back(tab[i].alpha);
  }
  while (cur <= dsize) (*start)[cur++] = (int)tab.size();
  return 0;
}

extern "C" int rtp_internal_load_png_jpeg(const char* path, unsigned char* out_bgr, size_t capacity, int* w, int* h);

extern "C" {

// rtpose.cpp:324-329: scale-to-fit, top-left anchored
double rtp_display_fit_scale(int ow, int oh, int disp_w, int disp_h) {
  if (ow / (double)oh > disp_w / (double)disp_h) return disp_w / (double)ow;
  return disp_h / (double)oh;
}

int rtp_resize_area(const unsigned char* bgr, int sw, int sh, unsigned char* out, int dw, int dh) {
  if (!bgr || !out || sw < 1 || sh < 1 || dw < 1 || dh < 1) return RTP_EINVAL;
  resize_area_u8(bgr, sw, sh, out, dw, dh);
  return RTP_OK;
}

int rtp_warp_display(const unsigned char* bgr, int sw, int sh, unsigned char* out, int disp_w, int disp_h, double* scale_out) {
  if (!bgr || !out || sw < 1 || sh < 1 || disp_w < 1 || disp_h < 1) return RTP_EINVAL;
  const double s = rtp_display_fit_scale(sw, sh, disp_w, disp_h);
  warp_scale_cubic_u8(bgr, sw, sh, s, out, disp_w, disp_h);
  if (scale_out) *scale_out = s;
  return RTP_OK;
}

// The producer's per-frame work (rtpose.cpp:322-368): returns the net input (num_scales x 3 x net_h x
// net_w) and Frame::scale.  display_bgr (disp_h x disp_w x 3) may be NULL.
int rtp_preprocess_frame(const unsigned char* bgr, int w, int h, int disp_w, int disp_h, int net_w, int net_h, int num_scales,
                         double start_scale, double scale_gap, float* net_input, unsigned char* display_bgr, float* frame_scale) {
  if (!bgr || !net_input || w < 1 || h < 1 || num_scales < 1 || disp_w < 1 || disp_h < 1 || net_w < 16 || net_h < 16 || (net_w % 16) || (net_h % 16) ||
      !(start_scale == start_scale) || !(scale_gap == scale_gap))
    return RTP_EINVAL;
  std::vector<unsigned char> disp((size_t)disp_w * disp_h * 3);
  double s = 0;
  int rc = rtp_warp_display(bgr, w, h, disp.data(), disp_w, disp_h, &s);
  if (rc) return rc;
  if (frame_scale) *frame_scale = (float)s;
  if (display_bgr) memcpy(display_bgr, disp.data(), disp.size());
  const size_t offset = (size_t)3 * net_h * net_w;
  std::vector<unsigned char> tmp;
  for (int i = 0; i < num_scales; ++i) {
    const float scale = (float)(start_scale - i * scale_gap);
    const int tw = (int)(16 * std::ceil(net_w * scale / 16));
    const int th = (int)(16 * std::ceil(net_h * scale / 16));
    if (tw > net_w || th > net_h || tw < 16 || th < 16) return RTP_EINVAL;  // CHECK_LE(target_width, NET_RESOLUTION_WIDTH)
    tmp.resize((size_t)tw * th * 3);
    resize_area_u8(disp.data(), disp_w, disp_h, tmp.data(), tw, th);
    rc = rtp_process_and_pad_image(net_input + i * offset, tmp.data(), tw, th, net_w, net_h, 1);
    if (rc) return rc;
  }
  return RTP_OK;
}

// ---- image files we can decode without OpenCV: binary PPM (P6) and 24-bit uncompressed BMP ------
extern "C" int rtp_internal_codec_fail(int code, const char* msg);
int rtp_load_image(const char* path, unsigned char* out_bgr, size_t capacity, int* w, int* h) {
  if (!path || !w || !h) return RTP_EINVAL;
  std::ifstream f(path, std::ios::binary);
  if (!f) return rtp_internal_codec_fail(RTP_EIO, (std::string("cannot open ") + path).c_str());
  unsigned char magic[2] = {0, 0};
  f.read((char*)magic, 2);
  if (magic[0] == 'P' && magic[1] == '6') {
    auto next_int = [&]() {
      int c = f.get();
      for (;;) {
        while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = f.get();
        if (c == '#') { while (c != '\n' && c != EOF) c = f.get(); continue; }
        break;
      }
      int v = 0;
      while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); c = f.get(); }
      return v;
    };
    const int W = next_int(), H = next_int(), maxv = next_int();
    if (W < 1 || H < 1 || maxv != 255 || (long long)W * H > (1LL << 26)) return rtp_internal_codec_fail(RTP_EIO, "PPM: bad header (need P6, maxval 255, <= 64 Mpixel)");  // same 64 Mpixel cap as codecs.cpp
    *w = W; *h = H;
    if (!out_bgr) return RTP_OK;
    if (capacity < (size_t)W * H * 3) return RTP_EINVAL;
    f.read((char*)out_bgr, (std::streamsize)W * H * 3);
    if (!f) return rtp_internal_codec_fail(RTP_EIO, "PPM: truncated pixel data");
    for (size_t i = 0; i < (size_t)W * H; ++i) std::swap(out_bgr[i * 3], out_bgr[i * 3 + 2]);  // RGB -> BGR
    return RTP_OK;
  }
  if (magic[0] == 'B' && magic[1] == 'M') {
    unsigned char hdr[52];
    f.read((char*)hdr, 52);
    if (!f) return rtp_internal_codec_fail(RTP_EIO, "BMP: truncated header");
    auto u32 = [&](int o) { return (uint32_t)hdr[o] | ((uint32_t)hdr[o + 1] << 8) | ((uint32_t)hdr[o + 2] << 16) | ((uint32_t)hdr[o + 3] << 24); };
    const uint32_t data_off = u32(8);
    const int32_t W = (int32_t)u32(16), Hs = (int32_t)u32(20);
    const int bpp = hdr[26] | (hdr[27] << 8);
    const uint32_t comp = u32(28);
    if (W < 1 || Hs == 0 || Hs == INT32_MIN || bpp != 24 || comp != 0 || (long long)W * (Hs < 0 ? -(long long)Hs : Hs) > (1LL << 26)) return rtp_internal_codec_fail(RTP_EIO, "BMP: only 24-bit uncompressed files up to 64 Mpixel are supported");
    const int H = Hs < 0 ? -Hs : Hs;
    *w = W; *h = H;
    if (!out_bgr) return RTP_OK;
    if (capacity < (size_t)W * H * 3) return RTP_EINVAL;
    const size_t stride = ((size_t)W * 3 + 3) & ~(size_t)3;
    std::vector<unsigned char> row(stride);
    f.seekg(data_off);
    for (int y = 0; y < H; ++y) {
      f.read((char*)row.data(), (std::streamsize)stride);
      if (!f) return rtp_internal_codec_fail(RTP_EIO, "BMP: truncated pixel data");
      const int dy = Hs < 0 ? y : H - 1 - y;
      memcpy(out_bgr + (size_t)dy * W * 3, row.data(), (size_t)W * 3);
    }
    return RTP_OK;
  }
  f.close();
  return rtp_internal_load_png_jpeg(path, out_bgr, capacity, w, h);  // codecs.cpp
}

// Procedural frame `index` of the synthetic video (BASELINE config 2: "synthetic 720p video"):
// a textured background with moving bright stick figures; deterministic in (seed, index).
int rtp_synth_frame(unsigned char* out_bgr, int w, int h, int index, uint64_t seed) {
  if (!out_bgr || w < 16 || h < 16) return RTP_EINVAL;
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
  auto rnd = [&]() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const unsigned v = (unsigned)(((x * 7 + y * 13 + index * 3) ^ (x * y >> 4)) & 63);
      unsigned char* p = out_bgr + ((size_t)y * w + x) * 3;
      p[0] = (unsigned char)(40 + v); p[1] = (unsigned char)(50 + v / 2); p[2] = (unsigned char)(60 + (v ^ 21) / 2);
    }
  const int people = 3;
  for (int pi = 0; pi < people; ++pi) {
    const double cx = (double)(rnd() % 1000) / 1000.0 * 0.6 + 0.2 + 0.15 * std::sin(index * 0.05 + pi);
    const double cy = 0.15 + 0.05 * pi;
    const double size = 0.5 + 0.1 * pi;
    auto line = [&](double x0, double y0, double x1, double y1, int thick, unsigned char b, unsigned char g, unsigned char r) {
      const int n = (int)(std::max(std::fabs(x1 - x0), std::fabs(y1 - y0)) * std::max(w, h)) + 1;
      for (int i = 0; i <= n; ++i) {
        const int px = (int)((x0 + (x1 - x0) * i / n) * w), py = (int)((y0 + (y1 - y0) * i / n) * h);
        for (int dy = -thick; dy <= thick; ++dy)
          for (int dx = -thick; dx <= thick; ++dx) {
            const int xx = px + dx, yy = py + dy;
            if (xx >= 0 && xx < w && yy >= 0 && yy < h) { unsigned char* p = out_bgr + ((size_t)yy * w + xx) * 3; p[0] = b; p[1] = g; p[2] = r; }
          }
      }
    };
    const double sw = 0.08 * size, top = cy, neck = cy + 0.08 * size, hip = cy + 0.45 * size, foot = cy + 0.9 * size;
    line(cx, top, cx, neck, 6, 200, 180, 220);
    line(cx - sw, neck, cx + sw, neck, 4, 90, 200, 240);
    line(cx, neck, cx, hip, 5, 90, 200, 240);
    line(cx - sw, neck, cx - 1.5 * sw, hip, 3, 240, 160, 90);
    line(cx + sw, neck, cx + 1.5 * sw, hip, 3, 240, 160, 90);
    line(cx, hip, cx - sw, foot, 4, 120, 240, 120);
    line(cx, hip, cx + sw, foot, 4, 120, 240, 120);
  }
  return RTP_OK;
}

}  // extern "C"
