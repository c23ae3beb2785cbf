// aux_kernels.hip — the HBM-bound glue of the conv stack: input packing, 2x2 max pooling and
// the debug export.  All are pure streaming kernels: 16-byte vector accesses, one pass.
#include "conv_common.h"

namespace rtp {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// ---- input pack: NCHW fp32 image -> 3x3 im2col, 32 channels, halo'd NHWC -------------------
// Replaces the H2D'd blobs()[0] (rtpose.cpp:1131-1133) as conv1_1's input: with the 27 taps of
// the Cin=3 first layer laid out as channels, conv1_1 becomes a 1x1 convolution with K=32 and
// runs on the same MFMA kernel as every other layer (model/coco/pose_deploy_linevec.prototxt:6-28).
template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ in, T* __restrict__ out,
                                                          Geom g, int Cp) {
  const long total = (long)g.N * g.H * g.W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % g.W);
    const int y = (int)((idx / g.W) % g.H);
    const int n = (int)(idx / ((long)g.W * g.H));
    const float* ip = in + (long)n * 3 * g.H * g.W;
    T vals[32];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int yy = y + r - 1, xx = x + s - 1;
        const bool ok = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          vals[(r * 3 + s) * 3 + c] = ok ? (T)ip[((long)c * g.H + yy) * g.W + xx] : (T)0.f;
      }
#pragma unroll
    for (int j = 27; j < 32; ++j) vals[j] = (T)0.f;
    T* op = out + (((long)n * g.Hp + y + g.halo) * g.Wp + x + g.halo) * Cp;
    constexpr int VEC = 16 / sizeof(T);
#pragma unroll
    for (int v = 0; v < 32 / VEC; ++v) {
      uint4 pk;
      __builtin_memcpy(&pk, &vals[v * VEC], 16);
      *(uint4*)(op + v * VEC) = pk;
    }
  }
}

hipError_t launch_pack_input(int prec, const float* in_nchw, void* out, Geom g, int Cp, hipStream_t stream) {
  const long total = (long)g.N * g.H * g.W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (prec == 0) hipLaunchKernelGGL(pack_input_kernel<_Float16>, dim3(blocks), dim3(256), 0, stream, in_nchw, (_Float16*)out, g, Cp);
  else hipLaunchKernelGGL(pack_input_kernel<float>, dim3(blocks), dim3(256), 0, stream, in_nchw, (float*)out, g, Cp);
  return hipGetLastError();
}

// OCP e4m3 byte -> float (q blocks: fp8 error-compensation operands, see ConvDst::q_off)
__device__ __forceinline__ float e4m3_to_float(unsigned b) {
  const unsigned s = b & 0x80, e = (b >> 3) & 15, m = b & 7;
  const float mag = e ? __uint_as_float(((e + 120) << 23) | (m << 20)) : (float)m * 0.001953125f;  // 2^(e-7) * (1 + m/8); subnormal m * 2^-9
  return s ? -mag : mag;
}

// ---- 2x2 / stride 2 MAX pooling (pooling_layer.cpp:140-180; resolutions are even) ----------
template <typename T, bool SPLIT>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, Geom gi, int Cpi, T* __restrict__ out,
                                                       Geom go, int Cpo, int C, int lo_i, int lo_o, int q_i, int q_o) {
  constexpr int VEC = 16 / sizeof(T);
  const int cv = C / VEC;
  const long total = (long)go.N * go.H * go.W * cv;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * VEC;
    long p = idx / cv;
    const int x = (int)(p % go.W);
    p /= go.W;
    const int y = (int)(p % go.H);
    const int n = (int)(p / go.H);
    const T* ip = in + (((long)n * gi.Hp + 2 * y + gi.halo) * gi.Wp + 2 * x + gi.halo) * Cpi + c;
    const long offs[4] = {0, Cpi, (long)gi.Wp * Cpi, (long)gi.Wp * Cpi + Cpi};
    T h[4][VEC], o[VEC];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = *(const uint4*)(ip + offs[q]);
      __builtin_memcpy(h[q], &v, 16);
    }
    T* op = out + (((long)n * go.Hp + y + go.halo) * go.Wp + x + go.halo) * Cpo + c;
    if constexpr (!SPLIT) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        T m = h[0][i];
        m = h[1][i] > m ? h[1][i] : m;
        m = h[2][i] > m ? h[2][i] : m;
        m = h[3][i] > m ? h[3][i] : m;
        o[i] = m;
      }
    } else {  // split precision: the value is hi + lo; the first maximum in (0,0),(0,1),(1,0),(1,1) order keeps all its parts
      T l[4][VEC], ol[VEC];
      unsigned char ql[4][VEC], qh[4][VEC], oql[VEC], oqh[VEC];
      float lf[4][VEC];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (lo_i) {
          const uint4 v = *(const uint4*)(ip + offs[q] + lo_i);
          __builtin_memcpy(l[q], &v, 16);
        }
        if constexpr (sizeof(T) == 2) {
          if (q_i) {  // 8 lo8 bytes and 8 hi8 bytes of these 8 channels (group of 64 channels = 128 bytes)
            const unsigned char* qp = (const unsigned char*)(ip - c + offs[q] + q_i) + (c >> 6) * 128 + (c & 63);
            const uint2 a = *(const uint2*)qp, b = *(const uint2*)(qp + 64);
            __builtin_memcpy(ql[q], &a, 8);
            __builtin_memcpy(qh[q], &b, 8);
          }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) lf[q][i] = lo_i ? (float)l[q][i] : ((sizeof(T) == 2 && q_i) ? e4m3_to_float(ql[q][i & 7]) : 0.f);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        T mh = h[0][i], ml = lo_i ? l[0][i] : (T)0.f;
        float mlf = lf[0][i];
        unsigned char mql = ql[0][i & 7], mqh = qh[0][i & 7];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          const bool gt = h[q][i] > mh || (h[q][i] == mh && lf[q][i] > mlf);
          mh = gt ? h[q][i] : mh;
          ml = gt ? (lo_i ? l[q][i] : (T)0.f) : ml;
          mlf = gt ? lf[q][i] : mlf;
          mql = gt ? ql[q][i & 7] : mql;
          mqh = gt ? qh[q][i & 7] : mqh;
        }
        o[i] = mh;
        ol[i] = ml;
        oql[i & 7] = mql;
        oqh[i & 7] = mqh;
      }
      if (lo_o && lo_i) {
        uint4 rl;
        __builtin_memcpy(&rl, ol, 16);
        *(uint4*)(op + lo_o) = rl;
      }
      if constexpr (sizeof(T) == 2) {
        if (q_o && q_i) {
          unsigned char* qp = (unsigned char*)(op - c + q_o) + (c >> 6) * 128 + (c & 63);
          uint2 a, b;
          __builtin_memcpy(&a, oql, 8);
          __builtin_memcpy(&b, oqh, 8);
          *(uint2*)qp = a;
          *(uint2*)(qp + 64) = b;
        }
      }
    }
    uint4 r;
    __builtin_memcpy(&r, o, 16);
    *(uint4*)op = r;
  }
}

// 2x2 max pooling of a tensor that carries fp8 compensation operands (q block, ConvDst::q_off): 8 channels per thread.
// The value of an element is hi + lo8 / 2^12; the first maximum in (0,0),(0,1),(1,0),(1,1) order keeps its parts (ties on hi
// are broken by lo8, compared through an order-preserving integer key).  hi8 = fp8(hi * 2^2) is a function of hi alone, so
// it is recomputed for the selected element with the conversion the conv epilogue uses instead of being read (a quarter
// less read traffic).  Packed integer state: the byte-array version of this kernel needed 234 VGPRs.
__global__ __launch_bounds__(256) void maxpool_q_kernel(const _Float16* __restrict__ in, Geom gi, int Cpi, _Float16* __restrict__ out,
                                                        Geom go, int Cpo, int C, int q_i, int q_o) {
  const int cv = C / 8;
  const long total = (long)go.N * go.H * go.W * cv;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * 8;
    long p = idx / cv;
    const int x = (int)(p % go.W);
    p /= go.W;
    const int y = (int)(p % go.H);
    const int n = (int)(p / go.H);
    const _Float16* ip = in + (((long)n * gi.Hp + 2 * y + gi.halo) * gi.Wp + 2 * x + gi.halo) * Cpi + c;
    const long offs[4] = {0, Cpi, (long)gi.Wp * Cpi, (long)gi.Wp * Cpi + Cpi};
    half8_t h[4];
    uint2 l8[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = *(const uint4*)(ip + offs[q]);
      __builtin_memcpy(&h[q], &v, 16);
      l8[q] = *(const uint2*)((const unsigned char*)(ip - c + offs[q] + q_i) + (c >> 6) * 128 + (c & 63));
    }
    half8_t oh;
    unsigned ol8[2] = {0u, 0u};
    float ohf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      auto lo_byte = [&](int q) { return ((i < 4 ? l8[q].x : l8[q].y) >> (8 * (i & 3))) & 0xffu; };
      auto key = [](unsigned b) { return (b & 0x80u) ? 0x80 - (int)(b & 0x7fu) : 0x80 + (int)b; };  // monotonic in the e4m3 value, -0 == +0
      _Float16 mh = h[0][i];
      unsigned mb = lo_byte(0);
      int mk = key(mb);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        const _Float16 hq = h[q][i];
        const unsigned b = lo_byte(q);
        const int k = key(b);
        const bool gt = hq > mh || (hq == mh && k > mk);
        mh = gt ? hq : mh;
        mb = gt ? b : mb;
        mk = gt ? k : mk;
      }
      oh[i] = mh;
      ohf[i] = (float)mh * (float)(1 << Q_HI_EXP);
      ol8[i >> 2] |= mb << (8 * (i & 3));
    }
    _Float16* op = out + (((long)n * go.Hp + y + go.halo) * go.Wp + x + go.halo) * Cpo + c;
    uint4 r;
    __builtin_memcpy(&r, &oh, 16);
    *(uint4*)op = r;
    if (q_o) {
      unsigned char* qp = (unsigned char*)(op - c + q_o) + (c >> 6) * 128 + (c & 63);
      uint2 a, b;
      a.x = ol8[0]; a.y = ol8[1];
      b.x = pack4_fp8(ohf[0], ohf[1], ohf[2], ohf[3]); b.y = pack4_fp8(ohf[4], ohf[5], ohf[6], ohf[7]);
      *(uint2*)qp = a;
      *(uint2*)(qp + 64) = b;
    }
  }
}

hipError_t launch_maxpool(int prec, const void* in, Geom gi, int Cpi, void* out, Geom go, int Cpo, int C, int lo_i, int lo_o, int q_i, int q_o,
                          hipStream_t stream) {
  const int vec = prec == 0 ? 8 : 4;
  const long total = (long)go.N * go.H * go.W * (C / vec);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (prec == 0 && q_i && !lo_i)
    hipLaunchKernelGGL(maxpool_q_kernel, dim3(blocks), dim3(256), 0, stream, (const _Float16*)in, gi, Cpi, (_Float16*)out, go, Cpo, C, q_i, q_o);
  else if (prec == 0 && (lo_i || q_i))
    hipLaunchKernelGGL((maxpool_kernel<_Float16, true>), dim3(blocks), dim3(256), 0, stream, (const _Float16*)in, gi, Cpi, (_Float16*)out, go, Cpo, C, lo_i, lo_o, q_i, q_o);
  else if (prec == 0)
    hipLaunchKernelGGL((maxpool_kernel<_Float16, false>), dim3(blocks), dim3(256), 0, stream, (const _Float16*)in, gi, Cpi, (_Float16*)out, go, Cpo, C, 0, 0, 0, 0);
  else hipLaunchKernelGGL((maxpool_kernel<float, false>), dim3(blocks), dim3(256), 0, stream, (const float*)in, gi, Cpi, (float*)out, go, Cpo, C, 0, 0, 0, 0);
  return hipGetLastError();
}

// ---- debug export: halo'd NHWC -> planar fp32 NCHW (Net::blob_by_name()->cpu_data() tap) ----
template <typename T>
__global__ __launch_bounds__(256) void export_kernel(const T* __restrict__ in, Geom g, int Cp, const int* __restrict__ chmap,
                                                      int C, int lo_off, int q_off, float* __restrict__ out) {
  const long total = (long)g.N * C * g.H * g.W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % g.W);
    long p = idx / g.W;
    const int y = (int)(p % g.H);
    p /= g.H;
    const int c = (int)(p % C);
    const int n = (int)(p / C);
    const int ci = chmap ? chmap[c] : c;
    const T* ip = in + (((long)n * g.Hp + y + g.halo) * g.Wp + x + g.halo) * Cp + ci;
    float v = (float)ip[0];
    if (lo_off) v += (float)ip[lo_off];
    else if (q_off && sizeof(T) == 2) v += e4m3_to_float(((const unsigned char*)(ip - ci + q_off))[(ci >> 6) * 128 + (ci & 63)]) * (1.f / 4096.f);
    out[idx] = v;
  }
}

hipError_t launch_export(int prec, const void* in, Geom g, int Cp, const int* chmap_dev, int C, int lo_off, int q_off, float* out,
                         hipStream_t stream) {
  const long total = (long)g.N * C * g.H * g.W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (prec == 0) hipLaunchKernelGGL(export_kernel<_Float16>, dim3(blocks), dim3(256), 0, stream, (const _Float16*)in, g, Cp, chmap_dev, C, lo_off, q_off, out);
  else hipLaunchKernelGGL(export_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)in, g, Cp, chmap_dev, C, lo_off, 0, out);
  return hipGetLastError();
}

}  // namespace rtp
