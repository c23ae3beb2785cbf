// codecs.cpp — SURVEY.md §8(f)-2: the reference reads frames with cv::imread / cv::VideoCapture
// (rtpose.cpp:323, 402-411, 431); OpenCV is absent here, so the decoders it would have used are
// restated for the formats the CLI accepts:
//   JPEG  Huffman sequential (baseline / extended, single- or multi-scan) AND progressive (SOF2: DC/AC
//         first + refinement scans, EOB runs), 8 bit, 1 or 3 components (4:4:4, 4:2:2, 4:2:0,
//         4:4:0), restart intervals.  The arithmetic is libjpeg(-turbo)'s DEFAULT decode path —
//         what cv::imread and PIL both run —: dequantise, jidctint.c `jpeg_idct_islow`
//         (CONST_BITS 13, PASS1_BITS 2), jdsample.c fancy ("triangle") up-sampling h2v1 / h2v2 /
//         h1v2 when the down-sampled width > 2 (else replication), jdcolor.c YCbCr->RGB tables
//         (SCALEBITS 16).  PINNED bit-for-bit against PIL's (libjpeg-turbo) decode of the fixtures in
//         tests/golden/codecs (tools/make_codec_fixtures.py).  Arithmetic-coded / lossless / 12-bit /
//         CMYK files are rejected with a message.
//   PNG   all colour types and bit depths, Adam7, through zlib's inflate (the one library
//         dependency; libz ships with the ROCm image); alpha stripped, 16 bit -> high byte,
//         low-bit grey expanded — libpng's transforms under cv::IMREAD_COLOR.  Lossless: pinned
//         against the source arrays of the fixtures.
//   Y4M   8-bit 4:2:0 / 4:2:2 / 4:4:4 / mono "video": BT.601 limited range, chroma replicated.
//         PARITY UNPINNED (the reference would go through ffmpeg's swscale, version unknown).
//   MJPEG a raw stream of concatenated JPEG images (no AVI container).
// Everything returns BGR HWC u8, the layout of cv::Mat the rest of the host path consumes.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/rtpose_mi355x.h"

namespace {

const long long kMaxPixels = 1LL << 26;  // 64 Mpixel: refuse absurd headers before allocating for them
thread_local std::string g_codec_err;
int cfail(int code, const std::string& m) { g_codec_err = m; return code; }

// =================================================================================================
// JPEG
// =================================================================================================
const int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool set = false;
  int mincode[17], maxcode[18], valptr[17];
  unsigned char vals[256];
  // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
  unsigned short fast[512];
};

// false: the code lengths over-subscribe the code space (not a Huffman table)
bool build_huff(Huff* h, const unsigned char bits[17], const unsigned char* vals, int nvals) {
  memcpy(h->vals, vals, nvals);
  int code = 0, k = 0;
  std::vector<int> codes(nvals), sizes(nvals);
  for (int l = 1; l <= 16; ++l) {
    h->valptr[l] = k;
    h->mincode[l] = code;
    for (int i = 0; i < bits[l]; ++i) { codes[k] = code++; sizes[k] = l; ++k; }
    if (code > (1 << l)) return false;
    h->maxcode[l] = bits[l] ? code - 1 : -1;
    code <<= 1;
  }
  h->maxcode[17] = 0x7fffffff;
  memset(h->fast, 0, sizeof h->fast);
  for (int i = 0; i < nvals; ++i) {
    if (sizes[i] > 9) continue;
    const int first = codes[i] << (9 - sizes[i]);
    for (int j = 0; j < (1 << (9 - sizes[i])); ++j) h->fast[first + j] = (unsigned short)((sizes[i] << 8) | vals[i]);
  }
  h->set = true;
  return true;
}

struct BitReader {
  const unsigned char* p;
  const unsigned char* end;
  uint64_t acc = 0;
  int nbits = 0;
  bool hit_marker = false;
  void fill() {
    while (nbits <= 48) {
      int b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;       // stuffed byte
          else { hit_marker = true; b = 0; }             // marker: feed zeros (libjpeg does the same)
        } else ++p;
      }
      acc = (acc << 8) | (uint64_t)b;
      nbits += 8;
    }
  }
  inline int peek(int n) { if (nbits < n) fill(); return (int)((acc >> (nbits - n)) & ((1u << n) - 1)); }
  inline void skip(int n) { nbits -= n; }
  inline int get(int n) { if (n == 0) return 0; const int v = peek(n); skip(n); return v; }
  void align_reset() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int huff_decode(BitReader& br, const Huff& h) {
  const int look = br.peek(9);
  const unsigned short f = h.fast[look];
  if (f) { br.skip(f >> 8); return f & 255; }
  int code = look, l = 9;
  br.skip(9);
  for (;;) {
    ++l;
    if (l > 16) return -1;
    code = (code << 1) | br.get(1);
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l]) break;
  }
  return h.vals[h.valptr[l] + code - h.mincode[l]];
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// post-IDCT range limit with libjpeg's RANGE_MASK wrap (jdmaster.c prepare_range_limit_table)
inline unsigned char idct_limit(int x) {
  const int i = x & 1023;
  if (i < 128) return (unsigned char)(128 + i);
  if (i < 512) return 255;
  if (i < 896) return 0;
  return (unsigned char)(i - 896);
}
inline unsigned char clamp255(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// jidctint.c jpeg_idct_islow: coef = dequantised coefficients in natural order, out: 8 rows
void idct_islow(const int* coef, unsigned char* out, int stride) {
  const int CONST_BITS = 13, PASS1_BITS = 2;
  const long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
             F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
             F_2_562915447 = 20995, F_3_072711026 = 25172;
  auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
  long ws[64];
  for (int c = 0; c < 8; ++c) {
    const int* in = coef + c;
    long z2 = in[16], z3 = in[48];
    long z1 = (z2 + z3) * F_0_541196100;
    long tmp2 = z1 + z3 * (-F_1_847759065);
    long tmp3 = z1 + z2 * F_0_765366865;
    z2 = in[0]; z3 = in[32];
    long tmp0 = (z2 + z3) * (1L << CONST_BITS);
    long tmp1 = (z2 - z3) * (1L << CONST_BITS);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    long* w = ws + c;
    w[0] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);  w[56] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
    w[8] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);  w[48] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
    w[16] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS); w[40] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
    w[24] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS); w[32] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
  }
  for (int r = 0; r < 8; ++r) {
    const long* w = ws + r * 8;
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F_0_541196100;
    long tmp2 = z1 + z3 * (-F_1_847759065);
    long tmp3 = z1 + z2 * F_0_765366865;
    long tmp0 = (w[0] + w[4]) * (1L << CONST_BITS);
    long tmp1 = (w[0] - w[4]) * (1L << CONST_BITS);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int S = CONST_BITS + PASS1_BITS + 3;
    unsigned char* o = out + r * stride;
    o[0] = idct_limit((int)descale(tmp10 + tmp3, S)); o[7] = idct_limit((int)descale(tmp10 - tmp3, S));
    o[1] = idct_limit((int)descale(tmp11 + tmp2, S)); o[6] = idct_limit((int)descale(tmp11 - tmp2, S));
    o[2] = idct_limit((int)descale(tmp12 + tmp1, S)); o[5] = idct_limit((int)descale(tmp12 - tmp1, S));
    o[3] = idct_limit((int)descale(tmp13 + tmp0, S)); o[4] = idct_limit((int)descale(tmp13 - tmp0, S));
  }
}

struct JComp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int bw = 0, bh = 0;          // blocks per row / column (padded to whole MCUs)
  int dw = 0, dh = 0;          // down-sampled size in samples (ceil)
  int pred = 0;
  std::vector<short> coef;           // bw*bh blocks of 64 coefficients (natural order, not dequantised)
  std::vector<unsigned char> plane;  // bw*8 x bh*8
};

// jdsample.c: one output row of h2v1 fancy up-sampling
void h2v1_fancy_row(const unsigned char* in, int dw, unsigned char* out) {
  const unsigned char* p = in;
  int invalue = *p++;
  *out++ = (unsigned char)invalue;
  *out++ = (unsigned char)((invalue * 3 + p[0] + 2) >> 2);
  for (int c = dw - 2; c > 0; --c) {
    invalue = (*p++) * 3;
    *out++ = (unsigned char)((invalue + p[-2] + 1) >> 2);
    *out++ = (unsigned char)((invalue + p[0] + 2) >> 2);
  }
  invalue = *p;
  *out++ = (unsigned char)((invalue * 3 + p[-1] + 1) >> 2);
  *out++ = (unsigned char)invalue;
}
// one output row of h2v2 fancy: in0 = the nearer input row (weight 3), in1 = the farther one
void h2v2_fancy_row(const unsigned char* in0, const unsigned char* in1, int dw, unsigned char* out) {
  int thiscolsum = in0[0] * 3 + in1[0];
  int nextcolsum = in0[1] * 3 + in1[1];
  int lastcolsum;
  *out++ = (unsigned char)((thiscolsum * 4 + 8) >> 4);
  *out++ = (unsigned char)((thiscolsum * 3 + nextcolsum + 7) >> 4);
  lastcolsum = thiscolsum; thiscolsum = nextcolsum;
  for (int c = 2; c < dw; ++c) {
    nextcolsum = in0[c] * 3 + in1[c];
    *out++ = (unsigned char)((thiscolsum * 3 + lastcolsum + 8) >> 4);
    *out++ = (unsigned char)((thiscolsum * 3 + nextcolsum + 7) >> 4);
    lastcolsum = thiscolsum; thiscolsum = nextcolsum;
  }
  *out++ = (unsigned char)((thiscolsum * 3 + lastcolsum + 8) >> 4);
  *out++ = (unsigned char)((thiscolsum * 4 + 7) >> 4);
}

// full-resolution plane of a component (W x H samples) from its down-sampled plane
int upsample(const JComp& c, int hmax, int vmax, int W, int H, std::vector<unsigned char>* full) {
  const int stride = c.bw * 8;
  const int hs = hmax / c.h, vs = vmax / c.v;
  if (hmax % c.h || vmax % c.v) return cfail(RTP_EINVAL, "JPEG: fractional sampling ratios are not supported");
  const int ow = c.dw * hs, oh = c.dh * vs;  // >= W, H
  std::vector<unsigned char> tmp((size_t)ow * oh);
  auto row = [&](int y) { return c.plane.data() + (size_t)std::min(std::max(y, 0), c.dh - 1) * stride; };
  const bool fancy = c.dw > 2;
  if (hs == 1 && vs == 1) {
    for (int y = 0; y < oh; ++y) memcpy(&tmp[(size_t)y * ow], row(y), ow);
  } else if (hs == 2 && vs == 1) {
    for (int y = 0; y < oh; ++y) {
      if (fancy) h2v1_fancy_row(row(y), c.dw, &tmp[(size_t)y * ow]);
      else for (int x = 0; x < ow; ++x) tmp[(size_t)y * ow + x] = row(y)[x >> 1];
    }
  } else if (hs == 2 && vs == 2) {
    for (int y = 0; y < oh; ++y) {
      const int iy = y >> 1;
      if (fancy) h2v2_fancy_row(row(iy), row((y & 1) ? iy + 1 : iy - 1), c.dw, &tmp[(size_t)y * ow]);  // context rows replicate at the edges
      else for (int x = 0; x < ow; ++x) tmp[(size_t)y * ow + x] = row(iy)[x >> 1];
    }
  } else if (hs == 1 && vs == 2) {  // h1v2_fancy_upsample (libjpeg-turbo): bias 1 for the upper row, 2 for the lower
    for (int y = 0; y < oh; ++y) {
      const int iy = y >> 1;
      const unsigned char* in0 = row(iy);
      const unsigned char* in1 = row((y & 1) ? iy + 1 : iy - 1);
      const int bias = (y & 1) ? 2 : 1;
      for (int x = 0; x < ow; ++x) tmp[(size_t)y * ow + x] = (unsigned char)((in0[x] * 3 + in1[x] + bias) >> 2);
    }
  } else {
    // generic integral replication (jdsample.c int_upsample)
    for (int y = 0; y < oh; ++y)
      for (int x = 0; x < ow; ++x) tmp[(size_t)y * ow + x] = row(y / vs)[x / hs];
  }
  full->resize((size_t)W * H);
  for (int y = 0; y < H; ++y) memcpy(&(*full)[(size_t)y * W], &tmp[(size_t)y * ow], W);
  return RTP_OK;
}

int decode_jpeg(const unsigned char* d, size_t n, unsigned char* out, size_t cap, int* ow_, int* oh_) {
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return cfail(RTP_EIO, "JPEG: no SOI");
  uint16_t qt[4][64];
  bool qset[4] = {false, false, false, false};
  Huff hdc[4], hac[4];
  std::vector<JComp> comps;
  int W = 0, H = 0, restart = 0;
  int adobe_transform = -1;
  bool have_sof = false, progressive = false, geometry_done = false;
  int hmax = 1, vmax = 1, mcux = 0, mcuy = 0, scans_done = 0;
  size_t pos = 2;
  auto u16 = [&](size_t o) { return (d[o] << 8) | d[o + 1]; };
  while (pos + 4 <= n) {
    if (d[pos] != 0xFF) { ++pos; continue; }
    int m = d[pos + 1];
    if (m == 0xFF) { ++pos; continue; }
    pos += 2;
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9) break;
    if (pos + 2 > n) break;
    const int len = u16(pos);
    if (len < 2 || pos + len > n) return cfail(RTP_EIO, "JPEG: truncated segment");
    const unsigned char* s = d + pos + 2;
    const int sl = len - 2;
    if (m == 0xDB) {  // DQT
      int o = 0;
      while (o < sl) {
        const int pq = s[o] >> 4, tq = s[o] & 15;
        if (tq > 3 || o + 1 + (pq ? 128 : 64) > sl) return cfail(RTP_EIO, "JPEG: bad DQT");
        ++o;
        for (int i = 0; i < 64; ++i) {
          if (pq) { qt[tq][i] = (uint16_t)((s[o] << 8) | s[o + 1]); o += 2; }
          else qt[tq][i] = s[o++];
        }
        qset[tq] = true;
      }
    } else if (m == 0xC4) {  // DHT
      int o = 0;
      while (o + 17 <= sl) {
        const int tc = s[o] >> 4, th = s[o] & 15;
        if (tc > 1 || th > 3) return cfail(RTP_EIO, "JPEG: bad DHT");
        unsigned char bits[17] = {0};
        int cnt = 0;
        for (int i = 1; i <= 16; ++i) { bits[i] = s[o + i]; cnt += bits[i]; }
        o += 17;
        if (cnt > 256 || o + cnt > sl) return cfail(RTP_EIO, "JPEG: bad DHT");
        if (!build_huff(tc ? &hac[th] : &hdc[th], bits, s + o, cnt)) return cfail(RTP_EIO, "JPEG: invalid Huffman table");
        o += cnt;
      }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {  // SOF0/1 sequential, SOF2 progressive
      progressive = m == 0xC2;
      if (sl < 6) return cfail(RTP_EIO, "JPEG: truncated SOF");
      if (s[0] != 8) return cfail(RTP_EINVAL, "JPEG: only 8-bit samples are supported");
      H = u16(pos + 3); W = u16(pos + 5);
      const int nc = s[5];
      if (W < 1 || H < 1 || (nc != 1 && nc != 3)) return cfail(RTP_EINVAL, "JPEG: only 1- or 3-component images are supported (no CMYK)");
      if ((long long)W * H > kMaxPixels) return cfail(RTP_EINVAL, "JPEG: image larger than 64 Mpixel");
      if (sl < 6 + 3 * nc) return cfail(RTP_EIO, "JPEG: truncated SOF");
      if (have_sof) return cfail(RTP_EIO, "JPEG: second frame header");
      comps.resize(nc);
      for (int i = 0; i < nc; ++i) {
        comps[i].id = s[6 + 3 * i];
        comps[i].h = s[7 + 3 * i] >> 4;
        comps[i].v = s[7 + 3 * i] & 15;
        comps[i].tq = s[8 + 3 * i];
        if (comps[i].h < 1 || comps[i].h > 4 || comps[i].v < 1 || comps[i].v > 4 || comps[i].tq > 3) return cfail(RTP_EIO, "JPEG: bad SOF");
      }
      have_sof = true;
      if (!out) { *ow_ = W; *oh_ = H; return RTP_OK; }  // size query: the frame header is enough
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return cfail(RTP_EINVAL, "JPEG: this coding process is not supported (Huffman sequential and progressive only)");
    } else if (m == 0xDD) {
      if (sl < 2) return cfail(RTP_EIO, "JPEG: truncated DRI");
      restart = u16(pos + 2);
    } else if (m == 0xEE && sl >= 12 && !memcmp(s, "Adobe", 5)) {
      adobe_transform = s[11];
    } else if (m == 0xDA) {  // SOS: one scan into the coefficient buffers
      if (!have_sof) return cfail(RTP_EIO, "JPEG: SOS before SOF");
      const int ns = sl >= 1 ? s[0] : 0;
      if (ns < 1 || ns > (int)comps.size() || sl < 1 + 2 * ns + 3) return cfail(RTP_EIO, "JPEG: bad SOS");
      JComp* sc[4];
      for (int i = 0; i < ns; ++i) {
        const int cid = s[1 + 2 * i];
        JComp* c = nullptr;
        for (auto& cc : comps) if (cc.id == cid) c = &cc;
        if (!c) return cfail(RTP_EIO, "JPEG: SOS names an unknown component");
        c->td = s[2 + 2 * i] >> 4;
        c->ta = s[2 + 2 * i] & 15;
        if (c->td > 3 || c->ta > 3) return cfail(RTP_EIO, "JPEG: bad table selector");
        sc[i] = c;
      }
      const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
      if (!progressive && (Ss != 0 || Se != 63 || Ah || Al)) return cfail(RTP_EIO, "JPEG: spectral selection in a sequential file");
      if (progressive && (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13)) return cfail(RTP_EIO, "JPEG: bad progressive scan parameters");
      for (int i = 0; i < ns; ++i) {
        if ((Ss == 0 && !Ah && !hdc[sc[i]->td].set) || (Se > 0 && !hac[sc[i]->ta].set)) return cfail(RTP_EIO, "JPEG: missing Huffman table");
      }
      pos += len;
      if (!geometry_done) {
        hmax = 1; vmax = 1;
        for (auto& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
        if (comps.size() == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }  // a single-component image is never interleaved
        mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (auto& c : comps) {
          c.bw = mcux * c.h; c.bh = mcuy * c.v;
          c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
          c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        geometry_done = true;
      }
      for (int i = 0; i < ns; ++i) sc[i]->pred = 0;
      BitReader br;
      br.p = d + pos; br.end = d + n;
      int eobrun = 0;
      // one block of scan data for component c at block (bx, by)
      auto decode_block = [&](JComp& c, int bx, int by) -> int {
        short* blk = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
        if (!progressive) {
          int s0 = huff_decode(br, hdc[c.td]);
          if (s0 < 0 || s0 > 15) return cfail(RTP_EIO, "JPEG: corrupt DC code");
          if (s0) c.pred += extend(br.get(s0), s0);
          blk[0] = (short)c.pred;
          for (int k = 1; k < 64;) {
            const int rs = huff_decode(br, hac[c.ta]);
            if (rs < 0) return cfail(RTP_EIO, "JPEG: corrupt AC code");
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
              if (r == 15) { k += 16; continue; }
              break;  // EOB
            }
            k += r;
            if (k > 63) return cfail(RTP_EIO, "JPEG: corrupt AC run");
            blk[kZigzag[k]] = (short)extend(br.get(sz), sz);
            ++k;
          }
          return RTP_OK;
        }
        if (Ss == 0) {  // DC scan (ITU T.81 G.1.2.1)
          if (Ah == 0) {
            int s0 = huff_decode(br, hdc[c.td]);
            if (s0 < 0 || s0 > 15) return cfail(RTP_EIO, "JPEG: corrupt DC code");
            if (s0) c.pred += extend(br.get(s0), s0);
            blk[0] = (short)(c.pred * (1 << Al));
          } else if (br.get(1)) blk[0] |= (short)(1 << Al);
          return RTP_OK;
        }
        if (Ah == 0) {  // AC first pass (G.1.2.2)
          if (eobrun > 0) { --eobrun; return RTP_OK; }
          for (int k = Ss; k <= Se;) {
            const int rs = huff_decode(br, hac[c.ta]);
            if (rs < 0) return cfail(RTP_EIO, "JPEG: corrupt AC code");
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
              if (r < 15) {
                eobrun = (1 << r) - 1;
                if (r) eobrun += br.get(r);
                break;
              }
              k += 16;
              continue;
            }
            k += r;
            if (k > 63) return cfail(RTP_EIO, "JPEG: corrupt AC run");
            blk[kZigzag[k]] = (short)(extend(br.get(sz), sz) * (1 << Al));
            ++k;
          }
          return RTP_OK;
        }
        // AC refinement (G.1.2.3; jdphuff.c decode_mcu_AC_refine)
        const int p1 = 1 << Al, m1 = -(1 << Al);
        int k = Ss;
        if (eobrun == 0) {
          for (; k <= Se; ++k) {
            const int rs = huff_decode(br, hac[c.ta]);
            if (rs < 0) return cfail(RTP_EIO, "JPEG: corrupt AC code");
            int r = rs >> 4;
            const int sz = rs & 15;
            int value = 0;
            if (sz) {
              value = br.get(1) ? p1 : m1;
            } else if (r != 15) {
              eobrun = 1 << r;
              if (r) eobrun += br.get(r);
              break;
            }
            do {
              short* cp = blk + kZigzag[k];
              if (*cp != 0) {
                if (br.get(1) && (*cp & p1) == 0) *cp = (short)(*cp + (*cp >= 0 ? p1 : m1));
              } else if (--r < 0) break;
              ++k;
            } while (k <= Se);
            if (sz && k <= Se) blk[kZigzag[k]] = (short)value;
          }
        }
        if (eobrun > 0) {
          for (; k <= Se; ++k) {
            short* cp = blk + kZigzag[k];
            if (*cp != 0 && br.get(1) && (*cp & p1) == 0) *cp = (short)(*cp + (*cp >= 0 ? p1 : m1));
          }
          --eobrun;
        }
        return RTP_OK;
      };
      auto do_restart = [&]() {
        br.align_reset();
        // the RSTn marker is normally right here; a damaged stream is searched forward at most once per byte
        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) {
          if (br.p[0] == 0xFF && br.p[1] != 0x00 && br.p[1] != 0xFF) { br.hit_marker = true; break; }  // another marker: the scan is over
          ++br.p;
        }
        if (!br.hit_marker && br.p + 1 < br.end) br.p += 2;
        for (int i = 0; i < ns; ++i) sc[i]->pred = 0;
        eobrun = 0;
      };
      int until_restart = restart;
      if (ns == 1) {  // non-interleaved: the component's own blocks in raster order (A.2.2)
        JComp& c = *sc[0];
        const int nbx = (c.dw + 7) / 8, nby = (c.dh + 7) / 8;
        for (int by = 0; by < nby; ++by)
          for (int bx = 0; bx < nbx; ++bx) {
            if (restart && until_restart == 0) { do_restart(); until_restart = restart; }
            const int rc = decode_block(c, bx, by);
            if (rc) return rc;
            if (restart) --until_restart;
          }
      } else {
        for (int my = 0; my < mcuy; ++my)
          for (int mx = 0; mx < mcux; ++mx) {
            if (restart && until_restart == 0) { do_restart(); until_restart = restart; }
            for (int i = 0; i < ns; ++i)
              for (int by = 0; by < sc[i]->v; ++by)
                for (int bx = 0; bx < sc[i]->h; ++bx) {
                  const int rc = decode_block(*sc[i], mx * sc[i]->h + bx, my * sc[i]->v + by);
                  if (rc) return rc;
                }
            if (restart) --until_restart;
          }
      }
      // continue after the entropy-coded segment: the next marker (the reader stopped in front of it)
      pos = (size_t)(br.p - d);
      while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] != 0x00 && !(d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7))) ++pos;
      scans_done++;
      continue;
    }
    pos += len;
  }
  if (!scans_done) return cfail(RTP_EIO, "JPEG: no scan found");
  {
    {
      // ---- coefficients -> samples: dequantise + IDCT of every block ----
      for (auto& c : comps) {
        if (!qset[c.tq]) return cfail(RTP_EIO, "JPEG: missing quantisation table");
        c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
        int coef[64];
        for (int by = 0; by < c.bh; ++by)
          for (int bx = 0; bx < c.bw; ++bx) {
            const short* blk = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
            for (int k = 0; k < 64; ++k) coef[kZigzag[k]] = blk[kZigzag[k]] * qt[c.tq][k];
            idct_islow(coef, c.plane.data() + ((size_t)by * 8) * (c.bw * 8) + (size_t)bx * 8, c.bw * 8);
          }
      }
      // ---- up-sample + colour ----
      *ow_ = W; *oh_ = H;
      if (!out) return RTP_OK;
      if (cap < (size_t)W * H * 3) return cfail(RTP_EINVAL, "output buffer too small");
      if (comps.size() == 1) {
        const int stride = comps[0].bw * 8;
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            const unsigned char v = comps[0].plane[(size_t)y * stride + x];
            unsigned char* o = out + ((size_t)y * W + x) * 3;
            o[0] = o[1] = o[2] = v;
          }
        return RTP_OK;
      }
      std::vector<unsigned char> f[3];
      for (int i = 0; i < 3; ++i) {
        const int rc = upsample(comps[i], hmax, vmax, W, H, &f[i]);
        if (rc) return rc;
      }
      const bool rgb = adobe_transform == 0 || (adobe_transform < 0 && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
      if (rgb) {
        for (size_t i = 0; i < (size_t)W * H; ++i) { out[i * 3] = f[2][i]; out[i * 3 + 1] = f[1][i]; out[i * 3 + 2] = f[0][i]; }
        return RTP_OK;
      }
      // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert
      struct YccTab {  // built once, thread-safely (C++11 static initialisation): decoders run on several threads
        int cr_r[256], cb_b[256];
        long cr_g[256], cb_g[256];
        YccTab() {
          for (int i = 0; i < 256; ++i) {
            const long x = i - 128;
            cr_r[i] = (int)((91881L * x + 32768L) >> 16);
            cb_b[i] = (int)((116130L * x + 32768L) >> 16);
            cr_g[i] = -46802L * x;
            cb_g[i] = -22554L * x + 32768L;
          }
        }
      };
      static const YccTab T;
      const int* cr_r = T.cr_r; const int* cb_b = T.cb_b;
      const long* cr_g = T.cr_g; const long* cb_g = T.cb_g;
      for (size_t i = 0; i < (size_t)W * H; ++i) {
        const int y = f[0][i], cb = f[1][i], cr = f[2][i];
        out[i * 3 + 2] = clamp255(y + cr_r[cr]);
        out[i * 3 + 1] = clamp255(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
        out[i * 3] = clamp255(y + cb_b[cb]);
      }
      return RTP_OK;
    }
  }
}

// =================================================================================================
// PNG
// =================================================================================================
inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// un-filter `rows` scanlines of `rowbytes` bytes (bpp = bytes per complete pixel, >= 1); in = filter byte + data per line
int png_unfilter(const unsigned char* in, size_t avail, int rows, size_t rowbytes, int bpp, std::vector<unsigned char>* out) {
  if (avail < (size_t)rows * (rowbytes + 1)) return cfail(RTP_EIO, "PNG: not enough image data");
  out->assign((size_t)rows * rowbytes, 0);
  std::vector<unsigned char> zero(rowbytes, 0);
  for (int y = 0; y < rows; ++y) {
    const unsigned char* s = in + (size_t)y * (rowbytes + 1);
    const int ft = s[0];
    ++s;
    unsigned char* o = out->data() + (size_t)y * rowbytes;
    const unsigned char* up = y ? o - rowbytes : zero.data();
    for (size_t i = 0; i < rowbytes; ++i) {
      const int a = i >= (size_t)bpp ? o[i - bpp] : 0, b = up[i], c = i >= (size_t)bpp ? up[i - bpp] : 0;
      int v = s[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: return cfail(RTP_EIO, "PNG: bad filter type");
      }
      o[i] = (unsigned char)v;
    }
  }
  return RTP_OK;
}

int decode_png(const unsigned char* d, size_t n, unsigned char* out, size_t cap, int* ow, int* oh) {
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (n < 33 || memcmp(d, sig, 8)) return cfail(RTP_EIO, "PNG: bad signature");
  size_t pos = 8;
  int W = 0, H = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<unsigned char> idat, pal;
  bool have_ihdr = false;
  while (pos + 12 <= n) {
    const uint32_t len = be32(d + pos);
    const unsigned char* type = d + pos + 4;
    const unsigned char* data = d + pos + 8;
    if (pos + 12 + (size_t)len > n) return cfail(RTP_EIO, "PNG: truncated chunk");
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return cfail(RTP_EIO, "PNG: bad IHDR");
      W = (int)be32(data); H = (int)be32(data + 4);
      depth = data[8]; ctype = data[9]; interlace = data[12];
      if (W < 1 || H < 1 || data[10] || data[11] || interlace > 1) return cfail(RTP_EIO, "PNG: bad IHDR");
      if ((long long)W * H > kMaxPixels) return cfail(RTP_EINVAL, "PNG: image larger than 64 Mpixel");
      have_ihdr = true;
    } else if (!memcmp(type, "PLTE", 4)) pal.assign(data, data + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!memcmp(type, "IEND", 4)) break;
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr) return cfail(RTP_EIO, "PNG: no IHDR");
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return cfail(RTP_EIO, "PNG: bad colour type");
  }
  if (!(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return cfail(RTP_EIO, "PNG: bad bit depth");
  if (ctype == 3 && (depth == 16 || pal.size() < 3)) return cfail(RTP_EIO, "PNG: bad palette image");
  *ow = W; *oh = H;
  if (!out) return RTP_OK;
  if (cap < (size_t)W * H * 3) return cfail(RTP_EINVAL, "output buffer too small");
  const int bits_pp = depth * channels;
  const int bpp = std::max(1, bits_pp / 8);
  // inflate
  size_t raw_cap = 0;
  struct Pass { int x0, y0, dx, dy; };
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  const Pass whole = {0, 0, 1, 1};
  const int npass = interlace ? 7 : 1;
  auto pass_dim = [&](int p, int* pw, int* ph) {
    const Pass& a = interlace ? adam7[p] : whole;
    *pw = (W - a.x0 + a.dx - 1) / a.dx;
    *ph = (H - a.y0 + a.dy - 1) / a.dy;
    if (*pw < 0) *pw = 0;
    if (*ph < 0) *ph = 0;
  };
  for (int p = 0; p < npass; ++p) {
    int pw, ph;
    pass_dim(p, &pw, &ph);
    if (pw && ph) raw_cap += (size_t)ph * (((size_t)pw * bits_pp + 7) / 8 + 1);
  }
  std::vector<unsigned char> raw(raw_cap);
  uLongf got = (uLongf)raw_cap;
  const int zr = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
  if (zr != Z_OK && !(zr == Z_BUF_ERROR && got == raw_cap)) return cfail(RTP_EIO, "PNG: inflate failed");
  if (got < raw_cap) return cfail(RTP_EIO, "PNG: not enough image data");
  // sample fetch of un-filtered line data
  auto put_pixel = [&](const unsigned char* line, int xi, int x, int y) {
    int r, g, b;
    auto samp = [&](int ch) -> int {
      if (depth == 8) return line[xi * channels + ch];
      if (depth == 16) return line[(xi * channels + ch) * 2];  // png_set_strip_16: the high byte
      const int per = 8 / depth;
      const int byte = line[xi / per];
      return (byte >> ((per - 1 - xi % per) * depth)) & ((1 << depth) - 1);
    };
    if (ctype == 3) {
      const int idx = samp(0);
      if ((size_t)idx * 3 + 2 < pal.size()) { r = pal[idx * 3]; g = pal[idx * 3 + 1]; b = pal[idx * 3 + 2]; }
      else r = g = b = 0;
    } else if (ctype == 0 || ctype == 4) {
      int v = samp(0);
      if (depth < 8) v = v * 255 / ((1 << depth) - 1);  // png_set_expand_gray_1_2_4_to_8
      r = g = b = v;
    } else { r = samp(0); g = samp(1); b = samp(2); }
    unsigned char* o = out + ((size_t)y * W + x) * 3;
    o[0] = (unsigned char)b; o[1] = (unsigned char)g; o[2] = (unsigned char)r;
  };
  size_t off = 0;
  std::vector<unsigned char> lines;
  for (int p = 0; p < npass; ++p) {
    int pw, ph;
    pass_dim(p, &pw, &ph);
    if (!pw || !ph) continue;
    const size_t rowbytes = ((size_t)pw * bits_pp + 7) / 8;
    const int rc = png_unfilter(raw.data() + off, raw.size() - off, ph, rowbytes, bpp, &lines);
    if (rc) return rc;
    off += (size_t)ph * (rowbytes + 1);
    const Pass& a = interlace ? adam7[p] : whole;
    for (int yy = 0; yy < ph; ++yy)
      for (int xx = 0; xx < pw; ++xx) put_pixel(lines.data() + (size_t)yy * rowbytes, xx, a.x0 + xx * a.dx, a.y0 + yy * a.dy);
  }
  return RTP_OK;
}

bool read_file(const char* path, std::vector<unsigned char>* buf) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  f.seekg(0, std::ios::end);
  const std::streamoff sz = f.tellg();
  if (sz < 0) return false;
  f.seekg(0);
  buf->resize((size_t)sz);
  if (sz) f.read((char*)buf->data(), sz);
  return (bool)f;
}

}  // namespace

// =================================================================================================
// JPEG encoder: cv::imwrite(name, img, {CV_IMWRITE_JPEG_QUALITY, q}) (rtpose.cpp:1367-1381) =
// libjpeg's defaults: JFIF 1.01, YCbCr 4:2:0, baseline, standard Huffman tables (Annex K),
// jccolor.c RGB->YCC tables, jcsample.c h2v2 down-sampling with alternating bias, edge replication
// (jcprepct.c / jcsample.c) and dummy edge blocks (jccoefct.c), jfdctint.c forward DCT, jcdctmgr.c
// rounding division.  PINNED byte-for-byte against Pillow's (libjpeg-turbo) files in tests/golden/codecs.
// =================================================================================================
namespace {

const unsigned char kStdLumQ[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const unsigned char kStdChrQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                                    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const unsigned char kDcLumBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const unsigned char kDcChrBits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const unsigned char kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const unsigned char kAcLumBits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const unsigned char kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
const unsigned char kAcChrBits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const unsigned char kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};

struct EncHuff { unsigned short code[256]; unsigned char size[256]; };
void build_enc_huff(EncHuff* h, const unsigned char bits[17], const unsigned char* vals) {
  memset(h, 0, sizeof *h);
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < bits[l]; ++i) { h->code[vals[k]] = (unsigned short)code++; h->size[vals[k]] = (unsigned char)l; ++k; }
    code <<= 1;
  }
}

struct BitWriter {
  unsigned char* p = nullptr;  // the caller guarantees room (worst case 2 bytes per 8 bits written)
  uint64_t acc = 0;
  int n = 0;
  inline void put(unsigned code, int size) {
    acc = (acc << size) | (uint64_t)(code & ((1u << size) - 1));
    n += size;
    while (n >= 8) {
      const unsigned char b = (unsigned char)(acc >> (n - 8));
      *p++ = b;
      if (b == 0xFF) *p++ = 0;
      n -= 8;
    }
  }
  void flush() { put(0x7F, 7); acc = 0; n = 0; }  // libjpeg pads the last byte with 1-bits
};

// jfdctint.c jpeg_fdct_islow, in place on level-shifted samples; output scaled by 8
void fdct_islow(int* data) {
  const int CONST_BITS = 13, PASS1_BITS = 2;
  typedef int32_t long_t;  // libjpeg's JLONG: every intermediate fits 32 bits for 8-bit samples
#define long long_t
  const long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
             F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
             F_2_562915447 = 20995, F_3_072711026 = 25172;
  auto descale = [](long x, int n) { return (int)((x + ((long)1 << (n - 1))) >> n); };
  for (int pass = 0; pass < 2; ++pass) {
    const int step = pass ? 8 : 1, next = pass ? 1 : 8;
    for (int i = 0; i < 8; ++i) {
      int* d = data + i * next;
      const long tmp0 = d[0] + d[7 * step], tmp7 = d[0] - d[7 * step], tmp1 = d[step] + d[6 * step], tmp6 = d[step] - d[6 * step];
      const long tmp2 = d[2 * step] + d[5 * step], tmp5 = d[2 * step] - d[5 * step], tmp3 = d[3 * step] + d[4 * step], tmp4 = d[3 * step] - d[4 * step];
      const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      long z1 = (tmp12 + tmp13) * F_0_541196100;
      if (!pass) {
        d[0] = (int)((tmp10 + tmp11) * (1 << PASS1_BITS));
        d[4 * step] = (int)((tmp10 - tmp11) * (1 << PASS1_BITS));
        d[2 * step] = descale(z1 + tmp13 * F_0_765366865, CONST_BITS - PASS1_BITS);
        d[6 * step] = descale(z1 + tmp12 * (-F_1_847759065), CONST_BITS - PASS1_BITS);
      } else {
        d[0] = descale(tmp10 + tmp11, PASS1_BITS);
        d[4 * step] = descale(tmp10 - tmp11, PASS1_BITS);
        d[2 * step] = descale(z1 + tmp13 * F_0_765366865, CONST_BITS + PASS1_BITS);
        d[6 * step] = descale(z1 + tmp12 * (-F_1_847759065), CONST_BITS + PASS1_BITS);
      }
      z1 = tmp4 + tmp7;
      long z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
      const long z5 = (z3 + z4) * F_1_175875602;
      const long t4 = tmp4 * F_0_298631336, t5 = tmp5 * F_2_053119869, t6 = tmp6 * F_3_072711026, t7 = tmp7 * F_1_501321110;
      z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
      z3 += z5; z4 += z5;
      const int sh = pass ? CONST_BITS + PASS1_BITS : CONST_BITS - PASS1_BITS;
      d[7 * step] = descale(t4 + z1 + z3, sh);
      d[5 * step] = descale(t5 + z2 + z4, sh);
      d[3 * step] = descale(t6 + z2 + z3, sh);
      d[step] = descale(t7 + z1 + z4, sh);
    }
  }
#undef long
}

inline int bit_category(int v) { const unsigned a = (unsigned)(v < 0 ? -v : v); return a ? 32 - __builtin_clz(a) : 0; }

}  // namespace

static long encode_jpeg_impl(const unsigned char* bgr, int W, int H, int quality, unsigned char* out, size_t capacity);
extern "C" long rtp_encode_jpeg(const unsigned char* bgr, int W, int H, int quality, unsigned char* out, size_t capacity) {
  try {
    return encode_jpeg_impl(bgr, W, H, quality, out, capacity);
  } catch (const std::exception& ex) {  // nothing may unwind through the C boundary
    return cfail(RTP_ENOMEM, std::string("JPEG encode: ") + ex.what());
  }
}
static long encode_jpeg_impl(const unsigned char* bgr, int W, int H, int quality, unsigned char* out, size_t capacity) {
  if (!bgr || W < 1 || H < 1 || W > 65535 || H > 65535) return RTP_EINVAL;
  if (quality < 1) quality = 1;
  if (quality > 100) quality = 100;
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;  // jpeg_quality_scaling
  unsigned char qt[2][64];
  for (int t = 0; t < 2; ++t)
    for (int i = 0; i < 64; ++i) {
      long v = ((long)(t ? kStdChrQ[i] : kStdLumQ[i]) * scale + 50L) / 100L;
      if (v <= 0) v = 1;
      if (v > 255) v = 255;  // force_baseline
      qt[t][i] = (unsigned char)v;
    }
  // ---- planes: Y at full resolution, Cb/Cr down-sampled 2x2; all padded to whole 16x16 MCUs
  const int mcux = (W + 15) / 16, mcuy = (H + 15) / 16;
  const int yw_blocks = (W + 7) / 8, yh_blocks = (H + 7) / 8;              // real blocks (width_in_blocks)
  const int cw_blocks = ((W + 1) / 2 + 7) / 8, ch_blocks = ((H + 1) / 2 + 7) / 8;
  const int YW = mcux * 16, YH = mcuy * 16, CW = mcux * 8, CH = mcuy * 8;
  std::vector<unsigned char> Y((size_t)YW * YH), Cb((size_t)CW * CH), Cr((size_t)CW * CH);
  {
    // colour conversion of the real pixels (jccolor.c rgb_ycc_convert)
    const int H2 = (H + 1) & ~1;                 // rows after padding to max_v_samp_factor (replicate the last row)
    const int fullw = std::max(yw_blocks * 8, cw_blocks * 16);  // expand_right_edge targets
    std::vector<unsigned char> fy((size_t)fullw * H2), fcb((size_t)fullw * H2), fcr((size_t)fullw * H2);
    for (int y = 0; y < H; ++y) {
      const unsigned char* px = bgr + (size_t)y * W * 3;
      unsigned char* py = &fy[(size_t)y * fullw];
      unsigned char* pcb = &fcb[(size_t)y * fullw];
      unsigned char* pcr = &fcr[(size_t)y * fullw];
      for (int x = 0; x < W; ++x, px += 3) {
        const int r = px[2], g = px[1], b = px[0];
        py[x] = (unsigned char)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
        pcb[x] = (unsigned char)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
        pcr[x] = (unsigned char)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
      }
      for (int x = W; x < fullw; ++x) { py[x] = py[W - 1]; pcb[x] = pcb[W - 1]; pcr[x] = pcr[W - 1]; }  // expand_right_edge
    }
    if (H2 > H) {  // expand_bottom_edge to max_v_samp_factor rows
      memcpy(&fy[(size_t)H * fullw], &fy[(size_t)(H - 1) * fullw], fullw);
      memcpy(&fcb[(size_t)H * fullw], &fcb[(size_t)(H - 1) * fullw], fullw);
      memcpy(&fcr[(size_t)H * fullw], &fcr[(size_t)(H - 1) * fullw], fullw);
    }
    // luma: copy; rows below the image replicate the last (padded) row up to the iMCU height
    for (int y = 0; y < YH; ++y) {
      const int sy = std::min(y, H2 - 1);
      const int real = std::min(YW, yw_blocks * 8);
      memcpy(&Y[(size_t)y * YW], &fy[(size_t)sy * fullw], real);
      for (int x = real; x < YW; ++x) Y[(size_t)y * YW + x] = fy[(size_t)sy * fullw + real - 1];
    }
    // chroma: h2v2_downsample with bias 1,2,1,2 along the row
    const int crow = H2 / 2, ccol = cw_blocks * 8;
    for (int y = 0; y < CH; ++y) {
      const int sy = std::min(y, crow - 1);
      for (int x = 0; x < CW; ++x) {
        const int sx = std::min(x, ccol - 1);
        const int bias = (sx & 1) ? 2 : 1;
        const size_t o0 = (size_t)(2 * sy) * fullw + 2 * sx, o1 = o0 + fullw;
        Cb[(size_t)y * CW + x] = (unsigned char)((fcb[o0] + fcb[o0 + 1] + fcb[o1] + fcb[o1 + 1] + bias) >> 2);
        Cr[(size_t)y * CW + x] = (unsigned char)((fcr[o0] + fcr[o0 + 1] + fcr[o1] + fcr[o1 + 1] + bias) >> 2);
      }
    }
  }
  std::vector<unsigned char> o;
  o.reserve((size_t)W * H);
  auto put16 = [&](int v) { o.push_back((unsigned char)(v >> 8)); o.push_back((unsigned char)v); };
  o.push_back(0xFF); o.push_back(0xD8);
  const unsigned char app0[] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  o.insert(o.end(), app0, app0 + sizeof app0);
  for (int t = 0; t < 2; ++t) {
    o.push_back(0xFF); o.push_back(0xDB); put16(67); o.push_back((unsigned char)t);
    for (int i = 0; i < 64; ++i) o.push_back(qt[t][kZigzag[i]]);
  }
  o.push_back(0xFF); o.push_back(0xC0); put16(17); o.push_back(8); put16(H); put16(W); o.push_back(3);
  o.push_back(1); o.push_back(0x22); o.push_back(0);
  o.push_back(2); o.push_back(0x11); o.push_back(1);
  o.push_back(3); o.push_back(0x11); o.push_back(1);
  auto dht = [&](int tc_th, const unsigned char* bits, const unsigned char* vals, int nv) {
    o.push_back(0xFF); o.push_back(0xC4); put16(2 + 1 + 16 + nv); o.push_back((unsigned char)tc_th);
    for (int i = 1; i <= 16; ++i) o.push_back(bits[i]);
    o.insert(o.end(), vals, vals + nv);
  };
  dht(0x00, kDcLumBits, kDcVals, 12);
  dht(0x10, kAcLumBits, kAcLumVals, 162);
  dht(0x01, kDcChrBits, kDcVals, 12);
  dht(0x11, kAcChrBits, kAcChrVals, 162);
  const unsigned char sos[] = {0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
  o.insert(o.end(), sos, sos + sizeof sos);
  EncHuff hdc[2], hac[2];
  build_enc_huff(&hdc[0], kDcLumBits, kDcVals); build_enc_huff(&hac[0], kAcLumBits, kAcLumVals);
  build_enc_huff(&hdc[1], kDcChrBits, kDcVals); build_enc_huff(&hac[1], kAcChrBits, kAcChrVals);
  const size_t header_bytes = o.size();
  // worst case per coefficient: 16-bit Huffman code + 11 value bits, every output byte 0xFF and stuffed => < 8 bytes
  o.resize(header_bytes + (size_t)mcux * mcuy * 6 * 64 * 8 + 64);
  uint64_t recip[2][64];
  for (int t = 0; t < 2; ++t)
    for (int i = 0; i < 64; ++i) { const uint64_t dv = (uint64_t)qt[t][i] << 3; recip[t][i] = ((1ull << 32) + dv - 1) / dv; }
  BitWriter bw;
  bw.p = o.data() + header_bytes;
  int pred[3] = {0, 0, 0};
  int blk[64], prev_q0 = 0;
  auto code_block = [&](const unsigned char* plane, int stride, int bx, int by, bool real, int comp, int tq) {
    int q[64];
    if (real) {
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) blk[r * 8 + c] = (int)plane[(size_t)(by * 8 + r) * stride + bx * 8 + c] - 128;
      fdct_islow(blk);
      for (int i = 0; i < 64; ++i) {  // jcdctmgr.c: round-half-away division by 8*Q (exact reciprocal: |t| < 2^17, 8*Q < 2^11)
        const int qval = qt[tq][i] << 3;
        const int t = blk[i];
        const unsigned a = (unsigned)(t < 0 ? -t : t) + (unsigned)(qval >> 1);
        const int m = (int)(((uint64_t)a * recip[tq][i]) >> 32);
        q[i] = t < 0 ? -m : m;
      }
    } else {  // dummy edge block: zero AC, DC of the previous block (jccoefct.c)
      memset(q, 0, sizeof q);
      q[0] = prev_q0;
    }
    prev_q0 = q[0];
    const int hidx = comp ? 1 : 0;
    int diff = q[0] - pred[comp];
    pred[comp] = q[0];
    int s = bit_category(diff);
    bw.put(hdc[hidx].code[s], hdc[hidx].size[s]);
    if (s) bw.put((unsigned)(diff < 0 ? diff - 1 : diff), s);
    int run = 0;
    for (int k = 1; k < 64; ++k) {
      const int v = q[kZigzag[k]];
      if (v == 0) { ++run; continue; }
      while (run > 15) { bw.put(hac[hidx].code[0xF0], hac[hidx].size[0xF0]); run -= 16; }
      s = bit_category(v);
      const int sym = (run << 4) | s;
      // code and value bits in one go (<= 16 + 10 bits)
      bw.put(((unsigned)hac[hidx].code[sym] << s) | ((unsigned)(v < 0 ? v - 1 : v) & ((1u << s) - 1)), hac[hidx].size[sym] + s);
      run = 0;
    }
    if (run) bw.put(hac[hidx].code[0], hac[hidx].size[0]);
  };
  for (int my = 0; my < mcuy; ++my)
    for (int mx = 0; mx < mcux; ++mx) {
      for (int by = 0; by < 2; ++by)
        for (int bx = 0; bx < 2; ++bx) {
          const int gx = mx * 2 + bx, gy = my * 2 + by;
          code_block(Y.data(), YW, gx, gy, gx < yw_blocks && gy < yh_blocks, 0, 0);
        }
      code_block(Cb.data(), CW, mx, my, mx < cw_blocks && my < ch_blocks, 1, 1);
      code_block(Cr.data(), CW, mx, my, mx < cw_blocks && my < ch_blocks, 2, 1);
    }
  bw.flush();
  o.resize((size_t)(bw.p - o.data()));
  o.push_back(0xFF); o.push_back(0xD9);
  if (out) {
    if (capacity < o.size()) return cfail(RTP_EINVAL, "output buffer too small");
    memcpy(out, o.data(), o.size());
  }
  return (long)o.size();
}

// =================================================================================================
// Video readers
// =================================================================================================
struct rtp_video {
  int kind = 0;  // 1 = Y4M, 2 = raw MJPEG
  int w = 0, h = 0, nframes = -1;
  std::vector<unsigned char> data;  // whole file (MJPEG) — the CLI's clips are small; Y4M streams from the file
  std::ifstream f;
  size_t pos = 0;
  int chroma = 420;  // Y4M
  std::vector<unsigned char> yuv;
  std::string err;
};

extern "C" {

const char* rtp_codec_last_error(void) { return g_codec_err.c_str(); }
int rtp_internal_codec_fail(int code, const char* msg) { return cfail(code, msg ? msg : ""); }  // for the PPM/BMP loader in preprocess.cpp

// cv::imread(path, IMREAD_COLOR) for PNG / JPEG byte strings -> BGR HWC.  out_bgr may be NULL to query the size.
int rtp_decode_image(const unsigned char* bytes, size_t n, unsigned char* out_bgr, size_t capacity, int* w, int* h) {
  if (!bytes || !w || !h) return RTP_EINVAL;
  try {
    if (n >= 8 && bytes[0] == 0x89 && bytes[1] == 'P') return decode_png(bytes, n, out_bgr, capacity, w, h);
    if (n >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8) return decode_jpeg(bytes, n, out_bgr, capacity, w, h);
  } catch (const std::exception& ex) {  // nothing may unwind through the C boundary
    return cfail(RTP_ENOMEM, std::string("image decode: ") + ex.what());
  }
  return cfail(RTP_EIO, "not a PNG or JPEG byte string");
}

int rtp_internal_load_png_jpeg(const char* path, unsigned char* out_bgr, size_t capacity, int* w, int* h) {
  std::vector<unsigned char> buf;
  if (!read_file(path, &buf)) return cfail(RTP_EIO, std::string("cannot read ") + path);
  return rtp_decode_image(buf.data(), buf.size(), out_bgr, capacity, w, h);
}

// cv::VideoCapture(path) for the two container-less formats decodable here.
static int video_open_impl(const char* path, rtp_video** out, int* w, int* h, int* nframes);
static int video_read_impl(rtp_video* v, unsigned char* out_bgr, size_t capacity);
int rtp_video_open(const char* path, rtp_video** out, int* w, int* h, int* nframes) {
  try {
    return video_open_impl(path, out, w, h, nframes);
  } catch (const std::exception& ex) {
    return cfail(RTP_ENOMEM, std::string("video open: ") + ex.what());
  }
}
int rtp_video_read(rtp_video* v, unsigned char* out_bgr, size_t capacity) {
  try {
    return video_read_impl(v, out_bgr, capacity);
  } catch (const std::exception& ex) {
    return cfail(RTP_ENOMEM, std::string("video read: ") + ex.what());
  }
}
static int video_open_impl(const char* path, rtp_video** out, int* w, int* h, int* nframes) {
  if (!path || !out) return RTP_EINVAL;
  rtp_video* v = new rtp_video();
  v->f.open(path, std::ios::binary);
  if (!v->f) { delete v; return cfail(RTP_EIO, std::string("cannot open ") + path); }
  char magic[10] = {0};
  v->f.read(magic, 9);
  v->f.seekg(0);
  if (!memcmp(magic, "YUV4MPEG2", 9)) {
    std::string hdr;
    std::getline(v->f, hdr);
    v->kind = 1;
    size_t i = 9;
    while (i < hdr.size()) {
      while (i < hdr.size() && hdr[i] == ' ') ++i;
      if (i >= hdr.size()) break;
      const char tag = hdr[i++];
      size_t j = i;
      while (j < hdr.size() && hdr[j] != ' ') ++j;
      const std::string val = hdr.substr(i, j - i);
      if (tag == 'W') v->w = atoi(val.c_str());
      else if (tag == 'H') v->h = atoi(val.c_str());
      else if (tag == 'C') {
        if (val.rfind("420", 0) == 0) v->chroma = 420;
        else if (val.rfind("422", 0) == 0) v->chroma = 422;
        else if (val.rfind("444", 0) == 0) v->chroma = 444;
        else if (val.rfind("mono", 0) == 0) v->chroma = 400;
        else { delete v; return cfail(RTP_EINVAL, "Y4M: unsupported chroma format " + val); }
        if (val.find("p1") != std::string::npos && val.find("p1") > 2) { delete v; return cfail(RTP_EINVAL, "Y4M: only 8-bit samples"); }
      }
      i = j;
    }
    if (v->w < 1 || v->h < 1 || (long long)v->w * v->h > kMaxPixels) { delete v; return cfail(RTP_EIO, "Y4M: bad header"); }
    const size_t cw = v->chroma == 444 ? v->w : (v->chroma == 400 ? 0 : (v->w + 1) / 2);
    const size_t ch = v->chroma == 420 ? (v->h + 1) / 2 : (v->chroma == 400 ? 0 : v->h);
    v->yuv.resize((size_t)v->w * v->h + 2 * cw * ch);
    const std::streamoff start = v->f.tellg();
    v->f.seekg(0, std::ios::end);
    const std::streamoff total = v->f.tellg();
    v->f.seekg(start);
    v->nframes = (int)((total - start) / (std::streamoff)(v->yuv.size() + 6));
  } else if ((unsigned char)magic[0] == 0xFF && (unsigned char)magic[1] == 0xD8) {
    v->kind = 2;
    v->f.close();
    if (!read_file(path, &v->data)) { delete v; return cfail(RTP_EIO, std::string("cannot read ") + path); }
    // count frames and take the size of the first
    int cnt = 0;
    for (size_t p = 0; p + 1 < v->data.size(); ++p)
      if (v->data[p] == 0xFF && v->data[p + 1] == 0xD9) ++cnt;
    v->nframes = cnt;
    int rc = rtp_decode_image(v->data.data(), v->data.size(), nullptr, 0, &v->w, &v->h);
    if (rc) { delete v; return rc; }
  } else {
    delete v;
    return cfail(RTP_EINVAL, "video: only Y4M (YUV4MPEG2) and raw MJPEG streams can be read without OpenCV/ffmpeg");
  }
  if (w) *w = v->w;
  if (h) *h = v->h;
  if (nframes) *nframes = v->nframes;
  *out = v;
  return RTP_OK;
}

// next frame -> BGR HWC; RTP_EAGAIN at the end of the stream
static int video_read_impl(rtp_video* v, unsigned char* out_bgr, size_t capacity) {
  if (!v || !out_bgr) return RTP_EINVAL;
  if (capacity < (size_t)v->w * v->h * 3) return cfail(RTP_EINVAL, "output buffer too small");
  if (v->kind == 1) {
    std::string line;
    if (!std::getline(v->f, line)) return RTP_EAGAIN;
    if (line.rfind("FRAME", 0) != 0) return cfail(RTP_EIO, "Y4M: FRAME marker expected");
    v->f.read((char*)v->yuv.data(), (std::streamsize)v->yuv.size());
    if (!v->f) return RTP_EAGAIN;
    const int W = v->w, H = v->h;
    const int cw = v->chroma == 444 ? W : (W + 1) / 2;
    const int chh = v->chroma == 420 ? (H + 1) / 2 : H;
    const unsigned char* Y = v->yuv.data();
    const unsigned char* U = Y + (size_t)W * H;
    const unsigned char* V = U + (size_t)cw * chh;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const int c = 298 * (Y[(size_t)y * W + x] - 16);
        int d = 0, e = 0;
        if (v->chroma != 400) {
          const int cx = v->chroma == 444 ? x : x >> 1, cy = v->chroma == 420 ? y >> 1 : y;
          d = U[(size_t)cy * cw + cx] - 128;
          e = V[(size_t)cy * cw + cx] - 128;
        }
        unsigned char* o = out_bgr + ((size_t)y * W + x) * 3;
        o[2] = clamp255((c + 409 * e + 128) >> 8);
        o[1] = clamp255((c - 100 * d - 208 * e + 128) >> 8);
        o[0] = clamp255((c + 516 * d + 128) >> 8);
      }
    return RTP_OK;
  }
  // MJPEG: next SOI..EOI
  const std::vector<unsigned char>& d = v->data;
  size_t p = v->pos;
  while (p + 1 < d.size() && !(d[p] == 0xFF && d[p + 1] == 0xD8)) ++p;
  if (p + 1 >= d.size()) return RTP_EAGAIN;
  size_t q = p + 2;
  while (q + 1 < d.size() && !(d[q] == 0xFF && d[q + 1] == 0xD9)) ++q;
  if (q + 1 >= d.size()) return RTP_EAGAIN;
  q += 2;
  int w = 0, h = 0;
  const int rc = rtp_decode_image(d.data() + p, q - p, nullptr, 0, &w, &h);
  if (rc) return rc;
  if (w != v->w || h != v->h) return cfail(RTP_EINVAL, "MJPEG: frame size changes inside the stream");
  v->pos = q;
  return rtp_decode_image(d.data() + p, q - p, out_bgr, capacity, &w, &h);
}

void rtp_video_close(rtp_video* v) { delete v; }

}  // extern "C"
