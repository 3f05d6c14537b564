// Convolution-shaped kernels of the MViT path (all HBM-bound, channels-last token layout):
//   * im2col for the 3-D patch embedding (video NDHWC fp32 -> [tokens, Kp] operand of the MFMA GEMM)
//   * depthwise 3x3x3 conv pooling of q/k/v heads (forward, data gradient, weight gradient)
//   * max-pool of the skip path (forward with argmax, backward)
// Tokens are [B, N, H, C] views (strides in elements, C contiguous); token 0 is the class token
// when has_cls != 0 and bypasses the pooling; the remaining tokens form a (T, Hs, Ws) grid.
#include "common.h"

// Minimum waves per SIMD the register allocation must leave room for (A/B builds: -DDW3_RUN_OCC=n ...).  hipcc allocates up to 512
// registers per lane for a 256-thread kernel when nothing asks for more waves.
#ifndef DW3_RUN_OCC
#define DW3_RUN_OCC 1
#endif
#ifndef DW3_BWD_WIDE_OCC
#define DW3_BWD_WIDE_OCC 1
#endif
#ifndef DW3_WGRAD_MARCH_OCC
#define DW3_WGRAD_MARCH_OCC 1
#endif
#include <stdlib.h>

namespace {

struct PoolGeom {
  int B, H, C;          // batch, heads, channels per head
  int T, Hs, Ws;        // input grid
  int To, Ho, Wo;       // output grid
  int st, sh, sw;       // strides
  int kt, kh, kw;       // kernel
  int pt, ph, pw;       // padding
  int has_cls;
  int64_t sxb, sxn, sxh;  // input strides
  int64_t syb, syn, syh;  // output strides
};

// Hardware block b runs on XCD b % 8.  This logical order gives every XCD a CONTIGUOUS range of the grid, so the rows and
// frames that neighbouring runs share are fetched into one L2 instead of all eight (PMC: L2 misses of the stride-1 pool
// were 4.8x its input with the round-robin order).
__device__ __forceinline__ unsigned xcd_block() {
  const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, qb = nb >> 3, rb = nb & 7;
  return xcd * qb + (xcd < rb ? xcd : rb) + (blockIdx.x >> 3);
}
// ---------------------------------------------------------------- patch-embed im2col
// col[tok][(kt*KH + kh)*KW*Cin + kw*Cin + c], zero padded to Kp; one wave per output token.
template <typename T>
__global__ __launch_bounds__(256) void im2col3d_kernel(const float* __restrict__ x, T* __restrict__ col, int B, int Tn,
                                                       int Hn, int Wn, int Cin, int KT, int KH, int KW, int st, int sh,
                                                       int sw, int pt, int ph, int pw, int To, int Ho, int Wo, int Kp) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t ntok = (int64_t)B * To * Ho * Wo;
  if (tok >= ntok) return;
  const int wo = tok % Wo, ho = (tok / Wo) % Ho, to = (tok / ((int64_t)Wo * Ho)) % To;
  const int b = tok / ((int64_t)Wo * Ho * To);
  const int run = KW * Cin;  // contiguous run in the input per (kt, kh)
  const int K = KT * KH * run;
  const float* xb = x + (int64_t)b * Tn * Hn * Wn * Cin;
  T* dst = col + tok * Kp;
  for (int i = lane; i < Kp; i += 64) {
    float v = 0.0f;
    if (i < K) {
      const int r = i % run, kh = (i / run) % KH, kt = i / (run * KH);
      const int kw = r / Cin, c = r % Cin;
      const int t = to * st - pt + kt, hh = ho * sh - ph + kh, ww = wo * sw - pw + kw;
      if (t >= 0 && t < Tn && hh >= 0 && hh < Hn && ww >= 0 && ww < Wn)
        v = xb[(((int64_t)t * Hn + hh) * Wn + ww) * Cin + c];
    }
    dst[i] = from_f32<T>(v);
  }
}

struct FastDiv { unsigned d; unsigned long long m; };
inline FastDiv mkdiv(unsigned d) { FastDiv f; f.d = d ? d : 1; f.m = ((1ull << 40) / f.d) + 1; return f; }
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return (unsigned)(((unsigned long long)n * f.m) >> 40); }

// Chunked variant (the one the patch embedding uses): a thread owns one 16-byte chunk of the column index and
// walks tokens, so the (kt, kh, kw, c) decomposition of its CV elements is done once, token coordinates come from
// multiply-shift divisions, padding reads zeros through out-of-range buffer offsets and the row is written with
// 16-byte stores (the generic kernel spends ~40 instructions and a 2-byte store per element).
struct Im2colGeom {
  int B, Tn, Hn, Wn, Cin, KT, KH, KW, st, sh, sw, pt, ph, pw, To, Ho, Wo, Kp;
  int cpt, tpb;          // chunks per token, tokens per block
  unsigned ntok, xbytes;
  FastDiv d_cpt, d_Wo, d_Ho, d_To;
};
// TI = float: frames already normalised (the reference's host pipeline); TI = uint8_t: raw pixels, (x - mean[c]) / std[c]
// is applied here (utils/video.py:278-288 on the fly: no fp32 copy of the clip ever exists, 4x less H2D traffic).  Padding
// taps are zeros of the NORMALISED tensor in both cases (Conv3d pads after the transform).
template <typename T, typename TI>
__global__ __launch_bounds__(256) void im2col3d_chunk_kernel(const TI* __restrict__ x, T* __restrict__ col, Im2colGeom g,
                                                             const float* __restrict__ mean, const float* __restrict__ stdv) {
  constexpr int CV = Vec16<T>::N;
  constexpr bool U8 = sizeof(TI) == 1;
  const unsigned tl = fdiv(threadIdx.x, g.d_cpt);
  const int ch = threadIdx.x - tl * g.cpt;
  if ((int)tl >= g.tpb) return;
  const int run = g.KW * g.Cin, K = g.KT * g.KH * run;
  int doff[CV], kt[CV], kh[CV], kw[CV];
  float nsub[CV], nmul[CV];
#pragma unroll
  for (int e = 0; e < CV; ++e) {
    const int i = ch * CV + e;
    nsub[e] = 0.0f; nmul[e] = 1.0f;
    if (i < K) {
      const int r = i % run;
      kh[e] = (i / run) % g.KH; kt[e] = i / (run * g.KH); kw[e] = r / g.Cin;
      doff[e] = ((kt[e] * g.Hn + kh[e]) * g.Wn + kw[e]) * g.Cin + (r % g.Cin);
      if constexpr (U8) { nsub[e] = mean[r % g.Cin]; nmul[e] = 1.0f / stdv[r % g.Cin]; }
    } else { kt[e] = 1 << 20; kh[e] = kw[e] = doff[e] = 0; }  // zero padding of the column index: never in range
  }
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<TI*>(x), 0, (int)g.xbytes, 0x00020000);
  for (unsigned tok = blockIdx.x * g.tpb + tl; tok < g.ntok; tok += gridDim.x * g.tpb) {  // (XCD-contiguous order measured slower here: 598 vs 540 us)
    const unsigned q1 = fdiv(tok, g.d_Wo);
    const int wo = tok - q1 * g.Wo;
    const unsigned q2 = fdiv(q1, g.d_Ho);
    const int ho = q1 - q2 * g.Ho;
    const unsigned b = fdiv(q2, g.d_To);
    const int to = q2 - b * g.To;
    const int t0 = to * g.st - g.pt, h0 = ho * g.sh - g.ph, w0 = wo * g.sw - g.pw;
    const int obase = (((int)b * g.Tn + t0) * g.Hn + h0) * g.Wn * g.Cin + w0 * g.Cin;
    float v[CV];
#pragma unroll
    for (int e = 0; e < CV; ++e) {
      const bool ok = (unsigned)(t0 + kt[e]) < (unsigned)g.Tn && (unsigned)(h0 + kh[e]) < (unsigned)g.Hn && (unsigned)(w0 + kw[e]) < (unsigned)g.Wn;
      if constexpr (U8) {
        const unsigned raw = __builtin_amdgcn_raw_buffer_load_b8(rsx, ok ? (unsigned)(obase + doff[e]) : 0xF0000000u, 0, 0);
        v[e] = ok ? ((float)(raw & 0xffu) - nsub[e]) * nmul[e] : 0.0f;
      } else {
        v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, ok ? (unsigned)(obase + doff[e]) * 4u : 0xF0000000u, 0, 0));
      }
    }
    T* dst = col + (int64_t)tok * g.Kp + ch * CV;
    if constexpr (sizeof(T) == 2) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      *reinterpret_cast<bf16x8*>(dst) = o;
    } else {
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    }
  }
}

// Row variant (bf16 columns): a block owns one output row (clip, frame to, row ho) = Wo tokens.  The KT * KH input rows that
// row touches are staged ONCE in LDS as bf16 (coalesced 16-byte loads, the normalisation of uint8 frames applied here, zero
// rows / zero side margins for the padding), then every token's column row is assembled from LDS: a (kt, kh) segment of a
// token is KW * Cin CONTIGUOUS LDS elements starting at wo * sw * Cin.  The chunk kernel above gathers every column element with
// its own 4-byte global load — a wave-wide load there touches 20-40 cache lines and the kernel runs at the texture unit's line
// rate (540 us for 1.03 GB at the headline shape); here global memory only sees full-row reads and 16-byte column stores.
struct Im2colRow { int rowp, nrows, run, K, margin; };   // LDS row pitch (elements), KT*KH, KW*Cin, KT*KH*KW*Cin, pw*Cin

template <typename TI>
__global__ __launch_bounds__(256) void im2col3d_row_kernel(const TI* __restrict__ x, bf16_t* __restrict__ col, Im2colGeom g,
                                                           Im2colRow rg, const float* __restrict__ mean,
                                                           const float* __restrict__ stdv) {
  constexpr bool U8 = sizeof(TI) == 1;
  extern __shared__ __attribute__((aligned(16))) bf16_t irow[];   // [nrows][rowp]
  const int tid = threadIdx.x;
  const unsigned rowid = blockIdx.x;                 // (b, to, ho)
  const unsigned q2 = fdiv(rowid, g.d_Ho);
  const int ho = rowid - q2 * g.Ho;
  const unsigned b = fdiv(q2, g.d_To);
  const int to = q2 - b * g.To;
  const int t0 = to * g.st - g.pt, h0 = ho * g.sh - g.ph;
  const int wc = g.Wn * g.Cin;                      // elements of one input row
  // ---- stage: zero margins, then the rows (4 elements per thread step)
  for (int i = tid; i < rg.nrows * 2 * rg.margin; i += 256) {
    const int r = i / (2 * rg.margin), m = i - r * 2 * rg.margin;
    irow[r * rg.rowp + (m < rg.margin ? m : wc + m)] = (bf16_t)0.0f;
  }
  const int q4 = wc / 4;                            // wc % 4 == 0 (checked by the launcher)
  for (int i = tid; i < rg.nrows * q4; i += 256) {
    const int r = i / q4, c4 = i - r * q4;
    const int kt = r / g.KH, kh = r - kt * g.KH;
    const int t = t0 + kt, hh = h0 + kh;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if ((unsigned)t < (unsigned)g.Tn && (unsigned)hh < (unsigned)g.Hn) {
      const int64_t src = (((int64_t)b * g.Tn + t) * g.Hn + hh) * wc + c4 * 4;
      if constexpr (U8) {
        const unsigned raw = *reinterpret_cast<const unsigned*>(x + src);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = (c4 * 4 + e) % g.Cin;
          v[e] = ((float)((raw >> (8 * e)) & 0xffu) - mean[c]) * (1.0f / stdv[c]);   // same arithmetic as the chunk kernel
        }
      } else {
        const f32x4 f = *reinterpret_cast<const f32x4*>(x + src);
        v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
      }
    }
    bf16_t* d = irow + r * rg.rowp + rg.margin + c4 * 4;
    d[0] = (bf16_t)v[0]; d[1] = (bf16_t)v[1]; d[2] = (bf16_t)v[2]; d[3] = (bf16_t)v[3];
  }
  __syncthreads();
  // ---- assemble: thread = (chunk of 8 column elements, token slot); the chunk's LDS offsets are fixed per thread
  const unsigned tl = fdiv(tid, g.d_cpt);
  const int ch = tid - tl * g.cpt;
  if ((int)tl >= g.tpb) return;
  int off[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = ch * 8 + e;
    const int r = i / rg.run;
    off[e] = i < rg.K ? r * rg.rowp + (i - r * rg.run) : -1;   // -1: zero padding of the column index
  }
  const int step = g.sw * g.Cin;
  bf16_t* out = col + ((int64_t)rowid * g.Wo) * g.Kp + ch * 8;
  for (int wo = tl; wo < g.Wo; wo += g.tpb) {
    const bf16_t* base = irow + wo * step;           // margin cancels: column w0 + kw = wo * sw - pw + kw sits at margin + that
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = off[e] >= 0 ? base[off[e]] : (bf16_t)0.0f;
    *reinterpret_cast<bf16x8*>(out + (int64_t)wo * g.Kp) = o;
  }
}

// launches the row kernel when its domain holds; false -> the caller uses the chunk kernel
template <typename TI>
bool launch_im2col_row(const TI* x, bf16_t* col, Im2colGeom g, const float* mean, const float* stdv, hipStream_t s) {
  static const bool on = !(getenv("DCC_IM2COL_ROW") && atoi(getenv("DCC_IM2COL_ROW")) == 0);   // A/B knob
  const int wc = g.Wn * g.Cin;
  Im2colRow rg;
  rg.margin = g.pw * g.Cin;
  rg.rowp = (wc + 2 * rg.margin + 7) & ~7;
  rg.nrows = g.KT * g.KH; rg.run = g.KW * g.Cin; rg.K = rg.nrows * rg.run;
  const size_t lds = (size_t)rg.nrows * rg.rowp * 2;
  // the right margin must cover the last token's window: (Wo-1)*sw - pw + KW <= Wn + pw
  if (!on || (wc % 4) || lds > 60 * 1024 || (g.Wo - 1) * g.sw - g.pw + g.KW > g.Wn + g.pw || rg.margin < 0) return false;
  if (sizeof(TI) == 1 ? ((uintptr_t)x & 3) || ((wc % 4) != 0) : ((uintptr_t)x & 15) != 0) return false;
  const int64_t nrows_out = (int64_t)g.B * g.To * g.Ho;
  if (nrows_out >= (1LL << 31)) return false;
  auto kern = im2col3d_row_kernel<TI>;
  static unsigned long long configured = 0;
  const unsigned long long devbit = dcc_dev_bit();
  if (!(configured & devbit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024) != hipSuccess) return false;
    configured |= devbit;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nrows_out), dim3(256), lds, s, x, col, g, rg, mean, stdv);
  return true;
}

// ---------------------------------------------------------------- depthwise conv / max pool forward
// one thread = one 16-byte channel chunk of one output token-head.
template <typename T, bool MAXPOOL>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       const float* __restrict__ w /*[C][taps]*/,
                                                       uint8_t* __restrict__ argmax, PoolGeom g) {
  constexpr int VEC = Vec16<T>::N;
  using V = vec16_t<T>;
  extern __shared__ __attribute__((aligned(16))) float wlds[];  // [taps][C]
  const int taps = g.kt * g.kh * g.kw;
  if (!MAXPOOL) {
    for (int i = threadIdx.x; i < taps * g.C; i += blockDim.x) {
      const int k = i / g.C, c = i % g.C;
      wlds[i] = w[c * taps + k];
    }
    __syncthreads();
  }
  const int cpr = g.C / VEC;
  const int No = g.To * g.Ho * g.Wo + g.has_cls;
  const int64_t total = (int64_t)g.B * No * g.H * cpr;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = id % cpr;
    const int hd = (id / cpr) % g.H;
    const int n = (id / ((int64_t)cpr * g.H)) % No;
    const int b = id / ((int64_t)cpr * g.H * No);
    const T* xb = x + b * g.sxb + hd * g.sxh + c * VEC;
    T* yo = y + b * g.syb + (int64_t)n * g.syn + hd * g.syh + c * VEC;
    if (g.has_cls && n == 0) {
      *reinterpret_cast<V*>(yo) = *reinterpret_cast<const V*>(xb);
      if (MAXPOOL && argmax) {
        uint8_t* am = argmax + ((((int64_t)b * No + n) * g.H + hd) * g.C + c * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) am[e] = 255;
      }
      continue;
    }
    const int o = n - g.has_cls;
    const int wo = o % g.Wo, ho = (o / g.Wo) % g.Ho, to = o / (g.Wo * g.Ho);
    float acc[VEC];
    int am[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[e] = MAXPOOL ? -INFINITY : 0.0f; am[e] = 255; }
    for (int kt = 0; kt < g.kt; ++kt) {
      const int t = to * g.st - g.pt + kt;
      if (t < 0 || t >= g.T) continue;
      for (int kh = 0; kh < g.kh; ++kh) {
        const int hh = ho * g.sh - g.ph + kh;
        if (hh < 0 || hh >= g.Hs) continue;
        for (int kw = 0; kw < g.kw; ++kw) {
          const int ww = wo * g.sw - g.pw + kw;
          if (ww < 0 || ww >= g.Ws) continue;
          const int nin = g.has_cls + (t * g.Hs + hh) * g.Ws + ww;
          const V v = *reinterpret_cast<const V*>(xb + (int64_t)nin * g.sxn);
          const int tap = (kt * g.kh + kh) * g.kw + kw;
          if (MAXPOOL) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float f = to_f32(v[e]);
              if (f > acc[e]) { acc[e] = f; am[e] = tap; }
            }
          } else {
            const float* wk = wlds + tap * g.C + c * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += to_f32(v[e]) * wk[e];
          }
        }
      }
    }
    V out;
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[e] = from_f32<T>(acc[e]);
    *reinterpret_cast<V*>(yo) = out;
    if (MAXPOOL && argmax) {
      uint8_t* amp = argmax + ((((int64_t)b * No + n) * g.H + hd) * g.C + c * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) amp[e] = (uint8_t)am[e];
    }
  }
}

// ---------------------------------------------------------------- data gradient (gather form)
// dx[in] = sum over (output o, tap k) with in = o*s - p + k of dy[o] * w[k]   (or dy[o] if argmax[o]==k)
// x-strides describe dx (the destination, e.g. a slice of the dqkv buffer); y-strides describe dy.
template <typename T, typename TD, bool MAXPOOL>
__global__ __launch_bounds__(256) void pool_bwd_data_kernel(const TD* __restrict__ dy, T* __restrict__ dx,
                                                            const float* __restrict__ w,
                                                            const uint8_t* __restrict__ argmax, PoolGeom g) {
  constexpr int VEC = Vec16<T>::N;
  using V = vec16_t<T>;
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int taps = g.kt * g.kh * g.kw;
  if (!MAXPOOL) {
    for (int i = threadIdx.x; i < taps * g.C; i += blockDim.x) {
      const int k = i / g.C, c = i % g.C;
      wlds[i] = w[c * taps + k];
    }
    __syncthreads();
  }
  const int cpr = g.C / VEC;
  const int Ni = g.T * g.Hs * g.Ws + g.has_cls;
  const int No = g.To * g.Ho * g.Wo + g.has_cls;
  const int64_t total = (int64_t)g.B * Ni * g.H * cpr;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = id % cpr;
    const int hd = (id / cpr) % g.H;
    const int n = (id / ((int64_t)cpr * g.H)) % Ni;
    const int b = id / ((int64_t)cpr * g.H * Ni);
    const TD* dyb = dy + b * g.syb + hd * g.syh + c * VEC;
    T* dxo = dx + b * g.sxb + (int64_t)n * g.sxn + hd * g.sxh + c * VEC;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.0f;
    if (g.has_cls && n == 0) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = to_f32(dyb[e]);
    } else {
      const int i = n - g.has_cls;
      const int ww = i % g.Ws, hh = (i / g.Ws) % g.Hs, t = i / (g.Ws * g.Hs);
      for (int kt = 0; kt < g.kt; ++kt) {
        const int tn = t + g.pt - kt;
        if (tn < 0 || tn % g.st) continue;
        const int to = tn / g.st;
        if (to >= g.To) continue;
        for (int kh = 0; kh < g.kh; ++kh) {
          const int hn = hh + g.ph - kh;
          if (hn < 0 || hn % g.sh) continue;
          const int ho = hn / g.sh;
          if (ho >= g.Ho) continue;
          for (int kw = 0; kw < g.kw; ++kw) {
            const int wn = ww + g.pw - kw;
            if (wn < 0 || wn % g.sw) continue;
            const int wo = wn / g.sw;
            if (wo >= g.Wo) continue;
            const int no = g.has_cls + (to * g.Ho + ho) * g.Wo + wo;
            const int tap = (kt * g.kh + kh) * g.kw + kw;
            const TD* d = dyb + (int64_t)no * g.syn;
            if (MAXPOOL) {
              const uint8_t* am = argmax + ((((int64_t)b * No + no) * g.H + hd) * g.C + c * VEC);
#pragma unroll
              for (int e = 0; e < VEC; ++e)
                if (am[e] == tap) acc[e] += to_f32(d[e]);
            } else {
              const float* wk = wlds + tap * g.C + c * VEC;
#pragma unroll
              for (int e = 0; e < VEC; ++e) acc[e] += to_f32(d[e]) * wk[e];
            }
          }
        }
      }
    }
    V out;
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[e] = from_f32<T>(acc[e]);
    *reinterpret_cast<V*>(dxo) = out;
  }
}

// ---------------------------------------------------------------- depthwise weight gradient
// dw[c][tap] += sum over (b, head, o) dy[o][c] * x[in(o, tap)][c].
// Work item = (output token-head, temporal tap kt): one 16-byte dy chunk against the 9 spatial taps of one input
// frame -> 9 independent 16-byte loads in flight per thread and 72 fp32 accumulators, so 4+ blocks fit per CU
// (the first version kept 27x4 accumulators per thread, one block per CU, and was latency-bound).
// Thread = (slot, chunk): slot % 3 = kt stays fixed over the grid-stride loop (the stride is a multiple of 3),
// so the per-thread sums fold into LDS red[27][C] once, then the block's partial is added into dw.
template <typename T, typename TD>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const TD* __restrict__ dy, const T* __restrict__ x,
                                                                float* __restrict__ ws, PoolGeom g) {
  constexpr int CV = Vec16<T>::N;  // channels per thread (8 bf16 / 4 fp32): 16-byte x loads
  using V = vec16_t<T>;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [27][C]
  const int taps = g.kt * g.kh * g.kw;
  const int cpr = g.C / CV;
  const int slots = (blockDim.x / cpr) / 3 * 3;  // items per block iteration, multiple of 3
  const int c = threadIdx.x % cpr, slot = threadIdx.x / cpr;
  for (int i = threadIdx.x; i < taps * g.C; i += blockDim.x) red[i] = 0.0f;
  __syncthreads();
  float acc[9][CV];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < CV; ++e) acc[k][e] = 0.0f;
  const int Ngrid = g.To * g.Ho * g.Wo;
  const int64_t total = (int64_t)g.B * g.H * Ngrid * 3;  // (token-head, kt) items; kt fastest
  const int kt = slot % 3;
  if (slot < slots && kt < g.kt) {
    for (int64_t id = (int64_t)blockIdx.x * slots + slot; id < total; id += (int64_t)gridDim.x * slots) {
      const int64_t th = id / 3;
      const int o = th % Ngrid;
      const int hd = (th / Ngrid) % g.H;
      const int b = th / ((int64_t)Ngrid * g.H);
      const int wo = o % g.Wo, ho = (o / g.Wo) % g.Ho, to = o / (g.Wo * g.Ho);
      const int t = to * g.st - g.pt + kt;
      if (t < 0 || t >= g.T) continue;
      float dv[CV];
      {
        const TD* d = dy + b * g.syb + (int64_t)(o + g.has_cls) * g.syn + hd * g.syh + c * CV;
#pragma unroll
        for (int e = 0; e < CV; ++e) dv[e] = to_f32(d[e]);
      }
      const T* xb = x + b * g.sxb + hd * g.sxh + c * CV;
      V xv[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = ho * g.sh - g.ph + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ww = wo * g.sw - g.pw + kw;
          const bool ok = kh < g.kh && kw < g.kw && hh >= 0 && hh < g.Hs && ww >= 0 && ww < g.Ws;
          V v;
#pragma unroll
          for (int e = 0; e < CV; ++e) v[e] = (T)0.0f;
          if (ok) v = *reinterpret_cast<const V*>(xb + (int64_t)(g.has_cls + (t * g.Hs + hh) * g.Ws + ww) * g.sxn);
          xv[kh * 3 + kw] = v;
        }
      }
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < CV; ++e) acc[k][e] += dv[e] * to_f32(xv[k][e]);
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        if (kh < g.kh && kw < g.kw) {
          const int tap = (kt * g.kh + kh) * g.kw + kw;
#pragma unroll
          for (int e = 0; e < CV; ++e) atomicAdd(&red[tap * g.C + c * CV + e], acc[kh * 3 + kw][e]);
        }
      }
  }
  __syncthreads();
  // per-block partial -> dw [C][taps] by float atomics in dw's own order (a wave-instruction adds 256 contiguous bytes: the
  // shape the memory-side atomic units take at full rate; <= 1024 blocks x 10 KB = 8 us of atomic bandwidth).  r02 wrote the
  // partials to a workspace and folded them with a second launch (dwconv_reduce_kernel: 22-36 us a time, 48 launches per step).
  for (int o = threadIdx.x; o < taps * g.C; o += blockDim.x) {
    const int cc = o / taps, k = o - cc * taps;
    atomicAdd(ws + o, red[k * g.C + cc]);
  }
}

// =====================================================================================================
// Specialised depthwise 3x3x3 kernels (padding 1, temporal stride 1, spatial stride S in {1,2,4,8}: every
// q/k/v pool of MViT-v2-S).  The generic kernels above spend most of their instructions on 64-bit index
// arithmetic and re-fetch every input up to 27 times; here
//   * a thread owns one 16-byte channel chunk of a RUN of WT consecutive output columns of one output row, so an
//     input row segment is fetched once per (kt, kh) and feeds all 3 kw taps of all WT outputs (11.6 instead of
//     27 vector loads per output for S = 1, WT = 7; MViT's grids 56/28/14/7 are multiples of 7);
//   * indices are 32-bit with multiply-shift division by launch constants, taps are compile-time;
//   * the data gradient of a stride-1 pool is the same stencil with mirrored taps; the data gradient of a
//     strided pool enumerates exactly the taps that reach an input position (1-2 per axis instead of 3 tests);
//   * the weight gradient keeps one temporal tap per thread (72 accumulators), folds the threads of a block
//     through LDS without atomics and adds one partial per block into dw with contiguous float atomics.

struct RunGeom {
  PoolGeom g;
  int runsW, runs, cpr, slots;
  FastDiv d_cpr, d_runsW, d_Ho, d_To, d_H, d_Ws, d_Hs, d_T, d_Wo;
  // buffer addressing of the stencil input ("x" side) and of dy ("y" side): spans and strides in BYTES (32-bit)
  unsigned xspan, bxb, bxn, bxh;
  unsigned yspan, byb, byn, byh;
};

// Stencil inputs are read with raw buffer loads: a lane whose tap falls into the zero padding gets an offset >= the
// buffer size and the hardware returns zeros, so no per-element select and no address clamp is needed (the selects
// and 64-bit address arithmetic were 2/3 of the instruction stream of the first version).
constexpr unsigned OOB = 0xF0000000u;   // offsets are combined with a saturating add: OOB + anything stays out of range
__device__ __forceinline__ unsigned sat_add(unsigned a, unsigned b) { return __builtin_elementwise_add_sat(a, b); }
template <typename TD> __device__ __forceinline__ __amdgpu_buffer_rsrc_t span_rsrc(const TD* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<TD*>(p), 0, (int)bytes, 0x00020000);
}

// w [C][27] fp32 -> LDS [27][C] (MIRROR: taps reversed).  Coalesced reads, eight in flight per thread: the first version
// (one strided load, wait, one LDS store per iteration) cost every block ten dependent L2 round trips before its first
// stencil load.
template <bool MIRROR> __device__ __forceinline__ void stage_w27(float* wl, const float* __restrict__ w, int C) {
  const int n = 27 * C;
  for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u;
      v[u] = i < n ? w[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u;
      if (i < n) {
        const int c = i / 27, k = i - c * 27;
        wl[(MIRROR ? 26 - k : k) * C + c] = v[u];
      }
    }
  }
}

// raw (unconverted) CV-element vector: issue all loads of a phase first, convert when consumed
template <typename TD, int CV> struct RawCV { f32x4 q[sizeof(TD) * CV / 16]; };
template <typename TD, int CV> __device__ __forceinline__ RawCV<TD, CV> load_raw(const TD* p) {
  RawCV<TD, CV> r;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(TD) * CV / 16); ++i) r.q[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p) + 16 * i);
  return r;
}
template <typename TD, int CV> __device__ __forceinline__ RawCV<TD, CV> load_raw_buf(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  RawCV<TD, CV> r;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(TD) * CV / 16); ++i)  // the 16*i goes into soffset, which the range check ignores
    r.q[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 16 * i, 0));
  return r;
}
template <typename TD, int CV> __device__ __forceinline__ void raw_to_f32(const RawCV<TD, CV>& r, float* f) {
  if constexpr (sizeof(TD) == 2) {
    const bf16x8 v = __builtin_bit_cast(bf16x8, r.q[0]);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
  } else {
#pragma unroll
    for (int i = 0; i < CV / 4; ++i) { f[4 * i] = r.q[i][0]; f[4 * i + 1] = r.q[i][1]; f[4 * i + 2] = r.q[i][2]; f[4 * i + 3] = r.q[i][3]; }
  }
}

template <typename TD, int CV> __device__ __forceinline__ void load_cv(const TD* p, float* f) {
  if constexpr (sizeof(TD) == 2) {
    static_assert(CV == 8, "bf16 chunks are 8 wide");
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
  } else {
#pragma unroll
    for (int q = 0; q < CV / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
      f[4 * q] = v[0]; f[4 * q + 1] = v[1]; f[4 * q + 2] = v[2]; f[4 * q + 3] = v[3];
    }
  }
}
template <typename T, int CV> __device__ __forceinline__ void store_cv(T* p, const float* f) {
  if constexpr (sizeof(T) == 2) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)f[e];
    *reinterpret_cast<bf16x8*>(p) = o;
  } else {
#pragma unroll
    for (int q = 0; q < CV / 4; ++q) *reinterpret_cast<f32x4*>(p + 4 * q) = f32x4{f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
  }
}

// run index -> (b, head, to, ho, first output column)
struct RunPos { int b, hd, to, ho, w0; };
__device__ __forceinline__ RunPos run_pos(unsigned run, const RunGeom& r, int WT) {
  RunPos p;
  const unsigned t1 = fdiv(run, r.d_runsW);
  p.w0 = (run - t1 * r.runsW) * WT;
  const unsigned t2 = fdiv(t1, r.d_Ho);
  p.ho = t1 - t2 * r.g.Ho;
  const unsigned t3 = fdiv(t2, r.d_To);
  p.to = t2 - t3 * r.g.To;
  const unsigned t4 = fdiv(t3, r.d_H);
  p.hd = t3 - t4 * r.g.H;
  p.b = t4;
  return p;
}

// forward (x = input, y = output) or stride-1 data gradient (MIRROR: x = dy, y = dx, taps mirrored)
template <typename T, typename TD, int S, int WT, bool MIRROR>
__global__ __launch_bounds__(256, (S == 1 && sizeof(TD) == 2 ? DW3_RUN_OCC : 1)) void dw3_run_kernel(const TD* __restrict__ x, T* __restrict__ y,
                                                      const float* __restrict__ w, RunGeom r) {
  constexpr int CV = Vec16<T>::N;
  constexpr int SEG = (WT - 1) * S + 3;
  extern __shared__ __attribute__((aligned(16))) float wl3[];  // [27][C]
  const PoolGeom& g = r.g;
  stage_w27<MIRROR>(wl3, w, g.C);
  __syncthreads();
  const unsigned rs = fdiv(threadIdx.x, r.d_cpr);
  const int c = threadIdx.x - rs * r.cpr;
  if ((int)rs >= r.slots) return;
  const unsigned run = xcd_block() * r.slots + rs;
  if ((int)run >= r.runs) {  // class tokens bypass the pooling
    const int i = run - r.runs;
    if (g.has_cls && i < g.B * g.H) {
      const int b = i / g.H, hd = i - b * g.H;
      float f[CV];
      load_cv<TD, CV>(x + b * g.sxb + hd * g.sxh + c * CV, f);
      store_cv<T, CV>(y + b * g.syb + hd * g.syh + c * CV, f);
    }
    return;
  }
  const RunPos p = run_pos(run, r, WT);
  const __amdgpu_buffer_rsrc_t rsx = span_rsrc(x, r.xspan);
  const unsigned xbase = p.b * r.bxb + p.hd * r.bxh + c * (CV * (unsigned)sizeof(TD));
  T* yb = y + p.b * g.syb + p.hd * g.syh + c * CV;
  float acc[WT][CV];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int e = 0; e < CV; ++e) acc[i][e] = 0.0f;
  // All loads of a temporal slab (3 rows x SEG columns) are in flight together; padding taps read zeros (OOB offset).
  const int col0 = p.w0 * S - 1;
  unsigned coff[SEG];
#pragma unroll
  for (int j = 0; j < SEG; ++j) {
    const int col = col0 + j;
    coff[j] = (col >= 0 && col < g.Ws) ? (unsigned)col * r.bxn : OOB;
  }
#pragma unroll 1
  for (int kt = 0; kt < 3; ++kt) {  // rolled: one slab of loads in flight is what the register file holds
    const int t = p.to + kt - 1;
    const bool tok = t >= 0 && t < g.T;
    RawCV<TD, CV> raw[3][SEG];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = p.ho * S + kh - 1;
      const bool ok = tok && hh >= 0 && hh < g.Hs;
      const unsigned row = ok ? xbase + (unsigned)(g.has_cls + (t * g.Hs + hh) * g.Ws) * r.bxn : OOB;
#pragma unroll
      for (int j = 0; j < SEG; ++j) {
        if (S > 2 && (j % S) > 2) continue;  // columns between the windows of a wide stride are never used
        raw[kh][j] = load_raw_buf<TD, CV>(rsx, sat_add(row, coff[j]));
      }
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      float wt[3][CV];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float* wp = wl3 + ((kt * 3 + kh) * 3 + kw) * g.C + c * CV;
#pragma unroll
        for (int q = 0; q < CV / 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(wp + 4 * q);
          wt[kw][4 * q] = v[0]; wt[kw][4 * q + 1] = v[1]; wt[kw][4 * q + 2] = v[2]; wt[kw][4 * q + 3] = v[3];
        }
      }
#pragma unroll
      for (int j = 0; j < SEG; ++j) {
        if (S > 2 && (j % S) > 2) continue;
        float f[CV];
        raw_to_f32<TD, CV>(raw[kh][j], f);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          if ((j - kw) >= 0 && (j - kw) % S == 0 && (j - kw) / S < WT) {
#pragma unroll
            for (int e = 0; e < CV; ++e) acc[(j - kw) / S][e] += f[e] * wt[kw][e];
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int wo = p.w0 + i;
    if (wo < g.Wo) store_cv<T, CV>(yb + (int64_t)(g.has_cls + (p.to * g.Ho + p.ho) * g.Wo + wo) * g.syn, acc[i]);
  }
}

// =====================================================================================================
// Fused q/k/v pool: depthwise 3x3x3 conv + the per-head LayerNorm(96) that follows it (torchvision mvit.Pool =
// conv -> norm), bf16, head_dim 96.  One launch instead of two, the conv output is not read back from HBM, and the
// thread mapping is chosen for the LayerNorm: a 16-lane DPP row owns one run of WT output columns with 6 channels
// (12 bytes) per lane, so the row statistics are four v_add_dpp steps and no lane idles (the stand-alone LayerNorm
// kernel packs a 96-wide row into 12 of 16 lanes).  One accumulator set per thread (WT x 6) and a two-row ring of raw
// loads keep the kernel at 3-4 waves per SIMD; the run kernel above holds a 27-load slab and 56 accumulators (2 waves).
// Outputs: y = conv rounded to bf16 (saved for the LayerNorm backward), o = LayerNorm(y) with its own row stride (the
// [.., DQK] attention operand), mean / rstd per row.  The class token bypasses the conv and is normalised like any row.
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));   // register triple of the 12-byte buffer loads only:
struct alignas(4) Raw6 { unsigned q[3]; };                     // memory accesses go through this 12-byte, 4-aligned type
__device__ __forceinline__ Raw6 raw6_load_buf(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0);
  return Raw6{{v[0], v[1], v[2]}};
}

__device__ __forceinline__ float row16_sum(float s) {  // all-reduce over the 16 lanes of a DPP row
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x128, 0xf, 0xf, false));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x124, 0xf, 0xf, false));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x122, 0xf, 0xf, false));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x121, 0xf, 0xf, false));
  return s;
}
__device__ __forceinline__ void raw6_to_f32(const Raw6& r, float* f) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f[2 * i] = __builtin_bit_cast(float, r.q[i] << 16);
    f[2 * i + 1] = __builtin_bit_cast(float, r.q[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ Raw6 f32_to_raw6(const float* f) {
  Raw6 r;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bf16x2 t = {(bf16_t)f[2 * i], (bf16_t)f[2 * i + 1]};
    r.q[i] = __builtin_bit_cast(unsigned, t);
  }
  return r;
}

// one row of 96 (6 per lane, v = the bf16-rounded conv output): statistics as norm.hip (two-pass, fp32), store o / mean / rstd
__device__ __forceinline__ void ln96_row_store(const float* v, const float* gam, const float* bet, float eps, bf16_t* orow,
                                               float* mean, float* rstd, int64_t row, bool lane0) {
  float s = 0.0f;
#pragma unroll
  for (int e = 0; e < 6; ++e) s += v[e];
  const float mu = row16_sum(s) * (1.0f / 96.0f);
  float q = 0.0f, d[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) { d[e] = v[e] - mu; q += d[e] * d[e]; }
  const float rs = rsqrtf(row16_sum(q) * (1.0f / 96.0f) + eps);
  float out[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) out[e] = d[e] * rs * gam[e] + bet[e];
  *reinterpret_cast<Raw6*>(orow) = f32_to_raw6(out);
  if (lane0) { mean[row] = mu; rstd[row] = rs; }
}

// One-hot key augmentation of the rel-pos contraction (relpos.hip: e(k) = [1{k_h} | 1{k_w} | 1{k_t}] behind the key, zeros for
// the class token) written by the k pool itself: columns D .. D + aug of the row at `orow`, aug / 16 columns per lane.
__device__ __forceinline__ void kaug_row_store(bf16_t* orow_l0, int l, int aug, int ch, int cw, int ct) {
  const int per = aug >> 4;   // 2 (aug 32) or 4 (aug 64) columns per lane
  unsigned v[2] = {0u, 0u};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < per) {
      const int col = l * per + e;
      if (col == ch || col == cw || col == ct) v[e >> 1] |= 0x3f80u << (16 * (e & 1));   // bf16 1.0
    }
  }
  unsigned* dst = reinterpret_cast<unsigned*>(orow_l0 + 96 + l * per);
  dst[0] = v[0];
  if (per > 2) dst[1] = v[1];
}

// one pooling branch of a launch; blockIdx.y picks between two branches of identical geometry (the k and v pools of a block run
// as ONE grid: each alone fills a third of the chip at stage 2, together they take the time of one)
struct PoolLnBr {
  const bf16_t* x; bf16_t* y; bf16_t* o; int64_t ldo;
  const float* w; const float* gamma; const float* beta; float* mean; float* rstd; int aug;
};

template <int S, int WT, int PD>
__global__ __launch_bounds__(256, PD == 0 ? 4 : 2) void dw3_ln_kernel(PoolLnBr br0, PoolLnBr br1, float eps, RunGeom r) {
  const PoolLnBr& br = blockIdx.y ? br1 : br0;
  const bf16_t* __restrict__ x = br.x;
  bf16_t* __restrict__ y = br.y;
  bf16_t* __restrict__ o = br.o;
  const int64_t ldo = br.ldo;
  const float* __restrict__ w = br.w;
  const float* __restrict__ gamma = br.gamma;
  const float* __restrict__ beta = br.beta;
  float* __restrict__ mean = br.mean;
  float* __restrict__ rstd = br.rstd;
  const int aug = br.aug;
  constexpr int NC = (WT - 1) * S + 3;
  __shared__ f32x4 wA[27 * 16];   // channels 6l .. 6l+3 of tap k at [k * 16 + l]
  __shared__ f32x2 wB[27 * 16];   // channels 6l+4, 6l+5
  __shared__ float gbs[2 * 96];
  const PoolGeom& g = r.g;
  {  // weights [96][27] -> the LDS images: coalesced loads, all in flight together (a load-wait-store loop costs a block
     // ten dependent L2 round trips before its first stencil load)
    float wv[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const int idx = threadIdx.x + 256 * i;
      wv[i] = idx < 27 * 96 ? w[idx] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int c = idx / 27, k = idx - c * 27, lc = c / 6, e = c - lc * 6;
      if (idx < 27 * 96) {
        if (e < 4) reinterpret_cast<float*>(wA)[(k * 16 + lc) * 4 + e] = wv[i];
        else reinterpret_cast<float*>(wB)[(k * 16 + lc) * 2 + (e - 4)] = wv[i];
      }
    }
  }
  if (threadIdx.x < 96) gbs[threadIdx.x] = gamma[threadIdx.x];
  else if (threadIdx.x < 192) gbs[threadIdx.x] = beta[threadIdx.x - 96];
  __syncthreads();
  const int l = threadIdx.x & 15;
  const unsigned run = xcd_block() * 16 + (threadIdx.x >> 4);
  float gam[6], bet[6];
  const int No = g.To * g.Ho * g.Wo + g.has_cls;
  if ((int)run >= r.runs) {  // class tokens: no conv, same LayerNorm
    const int i = run - r.runs;
    if (g.has_cls && i < g.B * g.H) {
      const int b = i / g.H, hd = i - b * g.H;
      const Raw6 q = *reinterpret_cast<const Raw6*>(x + b * g.sxb + hd * g.sxh + l * 6);
      float v[6];
      raw6_to_f32(q, v);
      *reinterpret_cast<Raw6*>(y + b * g.syb + hd * g.syh + l * 6) = q;
#pragma unroll
      for (int e = 0; e < 6; ++e) { gam[e] = gbs[l * 6 + e]; bet[e] = gbs[96 + l * 6 + e]; }
      const int64_t row = ((int64_t)b * No) * g.H + hd;
      ln96_row_store(v, gam, bet, eps, o + row * ldo + l * 6, mean, rstd, row, l == 0);
      if (aug) kaug_row_store(o + row * ldo, l, aug, -1, -1, -1);   // class token: zeros
    }
    return;
  }
  const RunPos p = run_pos(run, r, WT);
  const __amdgpu_buffer_rsrc_t rsx = span_rsrc(x, r.xspan);
  const unsigned xbase = p.b * r.bxb + p.hd * r.bxh + l * 12u;
  float acc[WT][6];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int e = 0; e < 6; ++e) acc[i][e] = 0.0f;
  const int col0 = p.w0 * S - 1;
  unsigned coff[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int col = col0 + j;
    coff[j] = (col >= 0 && col < g.Ws) ? (unsigned)col * r.bxn : OOB;
  }
  // The 9 (kt, kh) input rows are walked in order by a ROLLED loop (unrolled, the compiler sinks all the arithmetic below
  // all the loads and parks 80 loads in AGPRs); PD = 1: two row buffers, the next row's loads go out before the current
  // row is consumed.  Out-of-range rows read zeros through the OOB offset, so the body has no branches.
  auto issue = [&](int step, Raw6 (&buf)[NC]) {
    const int kt = (step * 11) >> 5, kh = step - kt * 3;  // step / 3 for 0..8
    const int t = p.to + kt - 1, hh = p.ho * S + kh - 1;
    const bool ok = ((unsigned)t < (unsigned)g.T) & ((unsigned)hh < (unsigned)g.Hs);
    const unsigned rowoff = xbase + (unsigned)(g.has_cls + (t * g.Hs + hh) * g.Ws) * r.bxn;
    const unsigned row = ok ? rowoff : OOB;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (S > 2 && (j % S) > 2) continue;  // columns between the windows of a wide stride are never used
      buf[j] = raw6_load_buf(rsx, sat_add(row, coff[j]));
    }
  };
  auto consume = [&](int step, const Raw6 (&buf)[NC]) {
    float wt[3][6];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const f32x4 a = wA[(step * 3 + kw) * 16 + l];
      const f32x2 b = wB[(step * 3 + kw) * 16 + l];
      wt[kw][0] = a[0]; wt[kw][1] = a[1]; wt[kw][2] = a[2]; wt[kw][3] = a[3]; wt[kw][4] = b[0]; wt[kw][5] = b[1];
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (S > 2 && (j % S) > 2) continue;
      float f[6];
      raw6_to_f32(buf[j], f);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        if ((j - kw) >= 0 && (j - kw) % S == 0 && (j - kw) / S < WT) {
#pragma unroll
          for (int e = 0; e < 6; ++e) acc[(j - kw) / S][e] += f[e] * wt[kw][e];
        }
      }
    }
  };
  if constexpr (PD == 0) {
    Raw6 A[NC];
#pragma unroll 1
    for (int step = 0; step < 9; ++step) {
      issue(step, A);
      consume(step, A);
    }
  } else {
    Raw6 A[NC], Bf[NC];
    issue(0, A);
#pragma unroll 1
    for (int s2 = 0; s2 < 8; s2 += 2) {
      issue(s2 + 1, Bf);
      __builtin_amdgcn_sched_barrier(0);
      consume(s2, A);
      __builtin_amdgcn_sched_barrier(0);
      issue(s2 + 2, A);
      __builtin_amdgcn_sched_barrier(0);
      consume(s2 + 1, Bf);
      __builtin_amdgcn_sched_barrier(0);
    }
    consume(8, A);
  }
#pragma unroll
  for (int e = 0; e < 6; ++e) { gam[e] = gbs[l * 6 + e]; bet[e] = gbs[96 + l * 6 + e]; }
  const int n0 = g.has_cls + (p.to * g.Ho + p.ho) * g.Wo + p.w0;
  bf16_t* yrow = y + p.b * g.syb + p.hd * g.syh + (int64_t)n0 * g.syn + l * 6;
  int64_t row = ((int64_t)p.b * No + n0) * g.H + p.hd;
  bf16_t* orow = o + row * ldo + l * 6;
  const int64_t ostep = g.H * ldo;
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    if (p.w0 + i < g.Wo) {  // uniform over the 16 lanes of the run
      const Raw6 q = f32_to_raw6(acc[i]);
      *reinterpret_cast<Raw6*>(yrow) = q;
      float v[6];
      raw6_to_f32(q, v);
      ln96_row_store(v, gam, bet, eps, orow, mean, rstd, row, l == 0);
      if (aug) kaug_row_store(orow - l * 6, l, aug, p.ho, g.Ho + p.w0 + i, g.Ho + g.Wo + p.to);
    }
    yrow += g.syn; orow += ostep; row += g.H;
  }
}

// data gradient of a strided pool (S >= 2): one thread per input position chunk, exact tap enumeration.
// x-strides describe dx, y-strides dy (as dcc_pool_bwd_data).
template <typename T, typename TD, int S>
__global__ __launch_bounds__(256) void dw3_bwd_data_strided_kernel(const TD* __restrict__ dy, T* __restrict__ dx,
                                                                   const float* __restrict__ w, RunGeom r) {
  constexpr int CV = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) float wl3[];
  const PoolGeom& g = r.g;
  stage_w27<false>(wl3, w, g.C);
  __syncthreads();
  const unsigned rs = fdiv(threadIdx.x, r.d_cpr);
  const int c = threadIdx.x - rs * r.cpr;
  if ((int)rs >= r.slots) return;
  const unsigned id = xcd_block() * r.slots + rs;  // position index over (b, head, t, h, w), then class tokens
  const int npos = g.B * g.H * g.T * g.Hs * g.Ws;
  float acc[CV];
#pragma unroll
  for (int e = 0; e < CV; ++e) acc[e] = 0.0f;
  if ((int)id >= npos) {
    const int i = id - npos;
    if (g.has_cls && i < g.B * g.H) {
      const int b = i / g.H, hd = i - b * g.H;
      load_cv<TD, CV>(dy + b * g.syb + hd * g.syh + c * CV, acc);
      store_cv<T, CV>(dx + b * g.sxb + hd * g.sxh + c * CV, acc);
    }
    return;
  }
  const unsigned t1 = fdiv(id, r.d_Ws);
  const int ww = id - t1 * g.Ws;
  const unsigned t2 = fdiv(t1, r.d_Hs);
  const int hh = t1 - t2 * g.Hs;
  const unsigned t3 = fdiv(t2, r.d_T);
  const int t = t2 - t3 * g.T;
  const unsigned t4 = fdiv(t3, r.d_H);
  const int hd = t3 - t4 * g.H, b = t4;
  // taps (k, o) with o * S - 1 + k == coordinate
  int khs[2], hos[2], nh = 0, kws[2], wos[2], nw = 0;
  auto taps = [](int x, int lim, int* ks, int* os) {
    int n = 0;
    if (S == 2) {
      if ((x & 1) == 0) { ks[0] = 1; os[0] = x >> 1; n = 1; }
      else { ks[0] = 0; os[0] = (x + 1) >> 1; ks[1] = 2; os[1] = (x - 1) >> 1; n = 2; }
    } else {
      const int rem = (x + 1) & (S - 1);
      if (rem <= 2) { ks[0] = rem; os[0] = (x + 1 - rem) / S; n = 1; }
    }
    int m = 0;
    for (int i = 0; i < n; ++i)
      if (os[i] < lim) { ks[m] = ks[i]; os[m] = os[i]; ++m; }
    return m;
  };
  nh = taps(hh, g.Ho, khs, hos);
  nw = taps(ww, g.Wo, kws, wos);
  const TD* dyb = dy + b * g.syb + hd * g.syh + c * CV;
  if (nh && nw) {
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int to = t + 1 - kt;
      if (to < 0 || to >= g.To) continue;
      for (int a = 0; a < nh; ++a)
        for (int bb = 0; bb < nw; ++bb) {
          float f[CV];
          load_cv<TD, CV>(dyb + (int64_t)(g.has_cls + (to * g.Ho + hos[a]) * g.Wo + wos[bb]) * g.syn, f);
          const float* wp = wl3 + ((kt * 3 + khs[a]) * 3 + kws[bb]) * g.C + c * CV;
#pragma unroll
          for (int e = 0; e < CV; ++e) acc[e] += f[e] * wp[e];
        }
    }
  }
  store_cv<T, CV>(dx + b * g.sxb + (int64_t)(g.has_cls + (t * g.Hs + hh) * g.Ws + ww) * g.sxn + hd * g.sxh + c * CV, acc);
}

// data gradient of a stride-2 pool whose input grid is exactly twice the output grid (every MViT stage transition):
// a thread owns one 16-byte channel chunk of ONE input row over a run of WQ column pairs.  An even row 2a is reached
// from output row a through kh = 1 only; an odd row 2a+1 from rows a (kh = 2) and a+1 (kh = 0).  Within a row, input
// column 2b takes dy[b] (kw = 1) and column 2b+1 takes dy[b] (kw = 2) and dy[b+1] (kw = 0), so the run needs WQ+1 dy
// columns per (kt, output row): 1.7-3.4 loads per output chunk instead of 6.75, and runs are ordered parity-major so
// that the "second output row" branch is wave-uniform.
template <typename T, typename TD, int WQ>
__global__ __launch_bounds__(256) void dw3_bwd_data_s2_kernel(const TD* __restrict__ dy, T* __restrict__ dx,
                                                              const float* __restrict__ w, RunGeom r) {
  constexpr int CV = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) float wl3[];  // [27][C]
  const PoolGeom& g = r.g;
  stage_w27<false>(wl3, w, g.C);
  __syncthreads();
  const unsigned rs = fdiv(threadIdx.x, r.d_cpr);
  const int c = threadIdx.x - rs * r.cpr;
  if ((int)rs >= r.slots) return;
  const unsigned run = xcd_block() * r.slots + rs;
  const int nruns = r.runs * 2;  // r.runs = B * H * To * Ho * runsW (one per output row run), two input rows each
  if ((int)run >= nruns) {
    const int i = run - nruns;
    if (g.has_cls && i < g.B * g.H) {
      const int b = i / g.H, hd = i - b * g.H;
      float f[CV];
      load_cv<TD, CV>(dy + b * g.syb + hd * g.syh + c * CV, f);
      store_cv<T, CV>(dx + b * g.sxb + hd * g.sxh + c * CV, f);
    }
    return;
  }
  // run -> (b, head, t, parity, a, first column pair)
  const unsigned t1 = fdiv(run, r.d_runsW);
  const int q0 = (run - t1 * r.runsW) * WQ;
  const unsigned t2 = fdiv(t1, r.d_Ho);
  const int a = t1 - t2 * g.Ho;
  const int par = t2 & 1;
  const unsigned t3 = t2 >> 1;
  const unsigned t4 = fdiv(t3, r.d_T);
  const int t = t3 - t4 * g.T;
  const unsigned t5 = fdiv(t4, r.d_H);
  const int hd = t4 - t5 * g.H, b = t5;
  const __amdgpu_buffer_rsrc_t rsy = span_rsrc(dy, r.yspan);
  const unsigned ybase = b * r.byb + hd * r.byh + c * (CV * (unsigned)sizeof(TD));
  unsigned coff[WQ + 1];
#pragma unroll
  for (int j = 0; j <= WQ; ++j) coff[j] = (q0 + j < g.Wo) ? (unsigned)(q0 + j) * r.byn : OOB;
  float acc[2 * WQ][CV];
#pragma unroll
  for (int i = 0; i < 2 * WQ; ++i)
#pragma unroll
    for (int e = 0; e < CV; ++e) acc[i][e] = 0.0f;
  // output rows feeding this input row: (ho, kh) = (a, par ? 2 : 1) and, for odd rows, (a + 1, 0)
  const int nrow = par ? 2 : 1;
#pragma unroll 1
  for (int kt = 0; kt < 3; ++kt) {
    const int to = t + 1 - kt;
    const bool tok = to >= 0 && to < g.To;
    RawCV<TD, CV> raw[2][WQ + 1];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      if (rr < nrow) {
        const int ho = a + rr;
        const unsigned row = (tok && ho < g.Ho) ? ybase + (unsigned)(g.has_cls + (to * g.Ho + ho) * g.Wo) * r.byn : OOB;
#pragma unroll
        for (int j = 0; j <= WQ; ++j) raw[rr][j] = load_raw_buf<TD, CV>(rsy, sat_add(row, coff[j]));
      }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      if (rr < nrow) {
        const int kh = par ? (rr ? 0 : 2) : 1;
        float wt[3][CV];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float* wp = wl3 + ((kt * 3 + kh) * 3 + kw) * g.C + c * CV;
#pragma unroll
          for (int q = 0; q < CV / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wp + 4 * q);
            wt[kw][4 * q] = v[0]; wt[kw][4 * q + 1] = v[1]; wt[kw][4 * q + 2] = v[2]; wt[kw][4 * q + 3] = v[3];
          }
        }
#pragma unroll
        for (int j = 0; j <= WQ; ++j) {
          float f[CV];
          raw_to_f32<TD, CV>(raw[rr][j], f);
          if (j < WQ) {
#pragma unroll
            for (int e = 0; e < CV; ++e) { acc[2 * j][e] += f[e] * wt[1][e]; acc[2 * j + 1][e] += f[e] * wt[2][e]; }
          }
          if (j >= 1) {
#pragma unroll
            for (int e = 0; e < CV; ++e) acc[2 * j - 1][e] += f[e] * wt[0][e];
          }
        }
      }
    }
  }
  T* xb = dx + b * g.sxb + hd * g.sxh + c * CV + (int64_t)(g.has_cls + (t * g.Hs + 2 * a + par) * g.Ws + 2 * q0) * g.sxn;
#pragma unroll
  for (int i = 0; i < 2 * WQ; ++i)
    if (q0 + i / 2 < g.Wo) store_cv<T, CV>(xb + (int64_t)i * g.sxn, acc[i]);
}

// data gradient of a stride-4 / stride-8 pool: windows do not overlap, so an input position has at most one tap per
// axis and S-3 of every S rows / columns are plain zeros.  A thread owns one input row over a run of WQ output
// columns (S * WQ input columns): 3 dy loads (one per temporal tap) give the 3 non-zero columns of a group, the rest
// of the row segment is zero-filled — the kernel is a streaming write of dx.
template <typename T, typename TD, int S, int WQ>
__global__ __launch_bounds__(256, DW3_BWD_WIDE_OCC) void dw3_bwd_data_wide_kernel(const TD* __restrict__ dy, T* __restrict__ dx,
                                                                const float* __restrict__ w, RunGeom r) {
  static_assert(S >= 4 && (S & (S - 1)) == 0, "non-overlapping windows");
  constexpr int CV = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) float wl3[];  // [27][C]
  const PoolGeom& g = r.g;
  stage_w27<false>(wl3, w, g.C);
  __syncthreads();
  const unsigned rs = fdiv(threadIdx.x, r.d_cpr);
  const int c = threadIdx.x - rs * r.cpr;
  if ((int)rs >= r.slots) return;
  const unsigned run = xcd_block() * r.slots + rs;
  const int nruns = g.B * g.H * g.T * g.Hs * r.runsW;
  if ((int)run >= nruns) {
    const int i = run - nruns;
    if (g.has_cls && i < g.B * g.H) {
      const int b = i / g.H, hd = i - b * g.H;
      float f[CV];
      load_cv<TD, CV>(dy + b * g.syb + hd * g.syh + c * CV, f);
      store_cv<T, CV>(dx + b * g.sxb + hd * g.sxh + c * CV, f);
    }
    return;
  }
  const unsigned t1 = fdiv(run, r.d_runsW);
  const int q0 = (run - t1 * r.runsW) * WQ;
  const unsigned t2 = fdiv(t1, r.d_Hs);
  const int h = t1 - t2 * g.Hs;
  const unsigned t3 = fdiv(t2, r.d_T);
  const int t = t2 - t3 * g.T;
  const unsigned t4 = fdiv(t3, r.d_H);
  const int hd = t3 - t4 * g.H, b = t4;
  const int kh = (h + 1) & (S - 1), ho = (h + 1) / S;
  const bool live = kh <= 2 && ho < g.Ho;
  T* xrow = dx + b * g.sxb + hd * g.sxh + c * CV + (int64_t)(g.has_cls + (t * g.Hs + h) * g.Ws) * g.sxn;
  float zero[CV];
#pragma unroll
  for (int e = 0; e < CV; ++e) zero[e] = 0.0f;
  const bool last = q0 + WQ >= g.Wo;
  if (!live) {  // a row no window touches
    const int c0 = q0 == 0 ? 0 : S * q0 - 1, c1 = last ? g.Ws : S * (q0 + WQ) - 1;
    for (int col = c0; col < c1; ++col) store_cv<T, CV>(xrow + (int64_t)col * g.sxn, zero);
    return;
  }
  const __amdgpu_buffer_rsrc_t rsy = span_rsrc(dy, r.yspan);
  const unsigned ybase = b * r.byb + hd * r.byh + c * (CV * (unsigned)sizeof(TD)) + (unsigned)(g.has_cls + ho * g.Wo + q0) * r.byn;
  RawCV<TD, CV> raw[3][WQ];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    const int to = t + 1 - kt;
    const unsigned row = (to >= 0 && to < g.To) ? ybase + (unsigned)(to * g.Ho * g.Wo) * r.byn : OOB;
#pragma unroll
    for (int j = 0; j < WQ; ++j) raw[kt][j] = load_raw_buf<TD, CV>(rsy, sat_add(row, q0 + j < g.Wo ? j * r.byn : OOB));
  }
  float wt[3][3][CV];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float* wp = wl3 + ((kt * 3 + kh) * 3 + kw) * g.C + c * CV;
#pragma unroll
      for (int q = 0; q < CV / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wp + 4 * q);
        wt[kt][kw][4 * q] = v[0]; wt[kt][kw][4 * q + 1] = v[1]; wt[kt][kw][4 * q + 2] = v[2]; wt[kt][kw][4 * q + 3] = v[3];
      }
    }
#pragma unroll
  for (int j = 0; j < WQ; ++j) {
    if (q0 + j >= g.Wo) break;
    float acc[3][CV];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int e = 0; e < CV; ++e) acc[kw][e] = 0.0f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      float f[CV];
      raw_to_f32<TD, CV>(raw[kt][j], f);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int e = 0; e < CV; ++e) acc[kw][e] += f[e] * wt[kt][kw][e];
    }
    const int cb = S * (q0 + j) - 1;  // column of tap kw = 0
#pragma unroll
    for (int kw = 0; kw < S; ++kw) {
      const int col = cb + kw;
      if (col >= 0 && col < g.Ws) store_cv<T, CV>(xrow + (int64_t)col * g.sxn, kw < 3 ? acc[kw < 3 ? kw : 0] : zero);
    }
  }
  if (last)
    for (int col = S * g.Wo - 1; col < g.Ws; ++col) store_cv<T, CV>(xrow + (int64_t)col * g.sxn, zero);
}

// skip-path max pool forward: MaxPool3d k(1,3,3) s(1,2,2) p(0,1,1), one thread per output position chunk; the 9 taps
// are read with unconditional buffer loads (padding taps masked to -inf afterwards), argmax = first maximum in tap order
// like the generic kernel (which walks them behind three nested bounds branches with 64-bit index arithmetic).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_s2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             uint8_t* __restrict__ argmax, RunGeom r) {
  constexpr int CV = Vec16<T>::N;
  const PoolGeom& g = r.g;
  const unsigned rs = fdiv(threadIdx.x, r.d_cpr);
  const int c = threadIdx.x - rs * r.cpr;
  if ((int)rs >= r.slots) return;
  const unsigned id = xcd_block() * r.slots + rs;
  const int npos = g.B * g.H * g.To * g.Ho * g.Wo;
  const int No = g.To * g.Ho * g.Wo + g.has_cls;
  if ((int)id >= npos) {
    const int i = id - npos;
    if (g.has_cls && i < g.B * g.H) {
      const int b = i / g.H, hd = i - b * g.H;
      float f[CV];
      load_cv<T, CV>(x + b * g.sxb + hd * g.sxh + c * CV, f);
      store_cv<T, CV>(y + b * g.syb + hd * g.syh + c * CV, f);
      if (argmax) {
        uint8_t* am = argmax + ((((int64_t)b * No) * g.H + hd) * g.C + c * CV);
#pragma unroll
        for (int e = 0; e < CV; ++e) am[e] = 255;
      }
    }
    return;
  }
  const unsigned t1 = fdiv(id, r.d_Wo);
  const int wo = id - t1 * g.Wo;
  const unsigned t2 = fdiv(t1, r.d_Ho);
  const int ho = t1 - t2 * g.Ho;
  const unsigned t3 = fdiv(t2, r.d_To);
  const int to = t2 - t3 * g.To;
  const unsigned b = fdiv(t3, r.d_H);
  const int hd = t3 - b * g.H;
  const __amdgpu_buffer_rsrc_t rsx = span_rsrc(x, r.xspan);
  const unsigned xbase = b * r.bxb + hd * r.bxh + c * (CV * (unsigned)sizeof(T));
  RawCV<T, CV> raw[9];
  bool ok[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hh = 2 * ho - 1 + kh;
    const bool rok = hh >= 0 && hh < g.Hs;
    const unsigned row = rok ? xbase + (unsigned)(g.has_cls + (to * g.Hs + hh) * g.Ws) * r.bxn : OOB;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = 2 * wo - 1 + kw;
      ok[kh * 3 + kw] = rok && ww >= 0 && ww < g.Ws;
      raw[kh * 3 + kw] = load_raw_buf<T, CV>(rsx, sat_add(row, (ww >= 0 && ww < g.Ws) ? (unsigned)ww * r.bxn : OOB));
    }
  }
  float acc[CV];
  int am[CV];
#pragma unroll
  for (int e = 0; e < CV; ++e) { acc[e] = -INFINITY; am[e] = 255; }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float f[CV];
    raw_to_f32<T, CV>(raw[k], f);
#pragma unroll
    for (int e = 0; e < CV; ++e) {
      const bool better = ok[k] && f[e] > acc[e];
      acc[e] = better ? f[e] : acc[e];
      am[e] = better ? k : am[e];
    }
  }
  const int n = g.has_cls + (to * g.Ho + ho) * g.Wo + wo;
  store_cv<T, CV>(y + b * g.syb + (int64_t)n * g.syn + hd * g.syh + c * CV, acc);
  if (argmax) {
    uint8_t* amp = argmax + ((((int64_t)b * No + n) * g.H + hd) * g.C + c * CV);
    if constexpr (CV == 8) {
      u32x2 pk;
      pk[0] = (unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24);
      pk[1] = (unsigned)am[4] | ((unsigned)am[5] << 8) | ((unsigned)am[6] << 16) | ((unsigned)am[7] << 24);
      *reinterpret_cast<u32x2*>(amp) = pk;
    } else {
      *reinterpret_cast<uint32_t*>(amp) = (unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24);
    }
  }
}

// skip-path max-pool data gradient: MaxPool3d k(1,3,3) s(1,2,2) p(0,1,1) with the argmax recorded by the forward
// (tap = kh * 3 + kw).  One thread per input position chunk; the <= 2 x 2 windows that contain the position are
// enumerated exactly (the generic kernel tests all 9 taps with 64-bit index arithmetic).
template <typename T, typename TD>
__global__ __launch_bounds__(256) void maxpool_s2_bwd_kernel(const TD* __restrict__ dy, T* __restrict__ dx,
                                                             const uint8_t* __restrict__ argmax, RunGeom r) {
  constexpr int CV = Vec16<T>::N;
  const PoolGeom& g = r.g;
  const unsigned rs = fdiv(threadIdx.x, r.d_cpr);
  const int c = threadIdx.x - rs * r.cpr;
  if ((int)rs >= r.slots) return;
  const unsigned id = xcd_block() * r.slots + rs;
  const int npos = g.B * g.H * g.T * g.Hs * g.Ws;
  float acc[CV];
#pragma unroll
  for (int e = 0; e < CV; ++e) acc[e] = 0.0f;
  if ((int)id >= npos) {
    const int i = id - npos;
    if (g.has_cls && i < g.B * g.H) {
      const int b = i / g.H, hd = i - b * g.H;
      load_cv<TD, CV>(dy + b * g.syb + hd * g.syh + c * CV, acc);
      store_cv<T, CV>(dx + b * g.sxb + hd * g.sxh + c * CV, acc);
    }
    return;
  }
  const unsigned t1 = fdiv(id, r.d_Ws);
  const int ww = id - t1 * g.Ws;
  const unsigned t2 = fdiv(t1, r.d_Hs);
  const int hh = t1 - t2 * g.Hs;
  const unsigned t3 = fdiv(t2, r.d_T);
  const int t = t2 - t3 * g.T;
  const unsigned t4 = fdiv(t3, r.d_H);
  const int hd = t3 - t4 * g.H, b = t4;
  const int No = g.To * g.Ho * g.Wo + g.has_cls;
  // windows o with 2 o - 1 + k == coordinate: even -> (k 1, o x/2); odd -> (k 0, o (x+1)/2), (k 2, o (x-1)/2)
  int khs[2], hos[2], kws[2], wos[2];
  const int nh = (hh & 1) ? 2 : 1, nw = (ww & 1) ? 2 : 1;
  if (hh & 1) { khs[0] = 0; hos[0] = (hh + 1) >> 1; khs[1] = 2; hos[1] = (hh - 1) >> 1; } else { khs[0] = 1; hos[0] = hh >> 1; khs[1] = 1; hos[1] = hh >> 1; }
  if (ww & 1) { kws[0] = 0; wos[0] = (ww + 1) >> 1; kws[1] = 2; wos[1] = (ww - 1) >> 1; } else { kws[0] = 1; wos[0] = ww >> 1; kws[1] = 1; wos[1] = ww >> 1; }
  for (int a = 0; a < nh; ++a) {
    if (hos[a] >= g.Ho) continue;
    for (int bb = 0; bb < nw; ++bb) {
      if (wos[bb] >= g.Wo) continue;
      const int no = g.has_cls + (t * g.Ho + hos[a]) * g.Wo + wos[bb];
      const int tap = khs[a] * 3 + kws[bb];
      float f[CV];
      load_cv<TD, CV>(dy + b * g.syb + (int64_t)no * g.syn + hd * g.syh + c * CV, f);
      const uint8_t* am = argmax + ((((int64_t)b * No + no) * g.H + hd) * g.C + c * CV);
      uint8_t a8[CV];
      if constexpr (CV == 8) { const u32x2 v = *reinterpret_cast<const u32x2*>(am); __builtin_memcpy(a8, &v, 8); }
      else { const uint32_t v = *reinterpret_cast<const uint32_t*>(am); __builtin_memcpy(a8, &v, 4); }
#pragma unroll
      for (int e = 0; e < CV; ++e) acc[e] += (a8[e] == tap) ? f[e] : 0.0f;
    }
  }
  store_cv<T, CV>(dx + b * g.sxb + (int64_t)(g.has_cls + (t * g.Hs + hh) * g.Ws + ww) * g.sxn + hd * g.sxh + c * CV, acc);
}

// weight gradient: thread = (chunk, temporal tap kt, run slot); block partial -> ws[blockIdx.x][27][C]
template <typename T, typename TD, int S, int WT>
__global__ __launch_bounds__(256) void dw3_wgrad_kernel(const TD* __restrict__ dy, const T* __restrict__ x,
                                                        float* __restrict__ ws, RunGeom r, int iters) {
  constexpr int CV = Vec16<T>::N;
  constexpr int SEG = (WT - 1) * S + 3;
  constexpr int PITCH = 9 * CV + 1;
  extern __shared__ __attribute__((aligned(16))) float red3[];  // [256][PITCH]
  const PoolGeom& g = r.g;
  const unsigned slot = fdiv(threadIdx.x, r.d_cpr);
  const int c = threadIdx.x - slot * r.cpr;
  const int slots3 = r.slots / 3;            // run slots per block
  const int kt = slot % 3, rsl = slot / 3;
  float acc[9][CV];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < CV; ++e) acc[k][e] = 0.0f;
  const __amdgpu_buffer_rsrc_t rsx = span_rsrc(x, r.xspan), rsy = span_rsrc(dy, r.yspan);
  const unsigned lblk = xcd_block();
  if ((int)slot < slots3 * 3) {
    for (int it = 0; it < iters; ++it) {
      const unsigned run = (lblk * iters + it) * slots3 + rsl;
      if ((int)run >= r.runs) break;
      const RunPos p = run_pos(run, r, WT);
      const int t = p.to + kt - 1;
      const bool tok = t >= 0 && t < g.T;
      // every load of the run (WT dy chunks + 3 input row segments) is issued before the first FMA; padding taps and
      // columns past the end of a partial run read zeros through out-of-range buffer offsets
      RawCV<TD, CV> draw[WT];
      const unsigned ybase = p.b * r.byb + p.hd * r.byh + c * (CV * (unsigned)sizeof(TD)) +
                             (unsigned)(g.has_cls + (p.to * g.Ho + p.ho) * g.Wo + p.w0) * r.byn;
#pragma unroll
      for (int i = 0; i < WT; ++i) draw[i] = load_raw_buf<TD, CV>(rsy, p.w0 + i < g.Wo ? ybase + i * r.byn : OOB);
      const unsigned xbase = p.b * r.bxb + p.hd * r.bxh + c * (CV * (unsigned)sizeof(T));
      const int col0 = p.w0 * S - 1;
      RawCV<T, CV> raw[3][SEG];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = p.ho * S + kh - 1;
        const bool ok = tok && hh >= 0 && hh < g.Hs;
        const unsigned row = ok ? xbase + (unsigned)(g.has_cls + (t * g.Hs + hh) * g.Ws) * r.bxn : OOB;
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
          if (S > 2 && (j % S) > 2) continue;
          const int col = col0 + j;
          raw[kh][j] = load_raw_buf<T, CV>(rsx, sat_add(row, (col >= 0 && col < g.Ws) ? (unsigned)col * r.bxn : OOB));
        }
      }
      float dv[WT][CV];
#pragma unroll
      for (int i = 0; i < WT; ++i) raw_to_f32<TD, CV>(draw[i], dv[i]);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
          if (S > 2 && (j % S) > 2) continue;
          float f[CV];
          raw_to_f32<T, CV>(raw[kh][j], f);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            if ((j - kw) >= 0 && (j - kw) % S == 0 && (j - kw) / S < WT) {
#pragma unroll
              for (int e = 0; e < CV; ++e) acc[kh * 3 + kw][e] += dv[(j - kw) / S][e] * f[e];
            }
          }
        }
      }
    }
  }
  float* mine = red3 + threadIdx.x * PITCH;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < CV; ++e) mine[k * CV + e] = acc[k][e];
  __syncthreads();
  for (int o = threadIdx.x; o < 27 * g.C; o += 256) {   // o walks dw [C][27]: contiguous float atomics (see dwconv_bwd_weight_kernel)
    const int ch = o / 27, tap = o - ch * 27;
    const int okt = tap / 9, k9 = tap - okt * 9;
    const int oc = ch / CV, oe = ch - oc * CV;
    float sum = 0.0f;
    for (int s3 = 0; s3 < slots3; ++s3) sum += red3[((s3 * 3 + okt) * r.cpr + oc) * PITCH + k9 * CV + oe];
    atomicAdd(ws + o, sum);
  }
}

// Weight gradient, marching variant: thread = (chunk, row tap kh, run slot) walks the output frames to = 0..To-1 of one
// (clip, head, output row, column run).  The input row of frame t is loaded ONCE and serves the temporal taps kt = 0, 1, 2
// of the outputs to = t+1, t, t-1 from a rolling three-row window in registers, and the dy run is loaded once per frame:
// 16 loads per (output run, frame, kh) instead of 34 per (output run, kt).  The next frame's row / dy run are requested
// before the current frame's FMAs.  Partials: same block fold as dw3_wgrad_kernel (tap = (kt*3 + kh)*3 + kw).
template <typename T, typename TD, int S, int WT>
__global__ __launch_bounds__(256, (WT == 2 ? DW3_WGRAD_MARCH_OCC : 1)) void dw3_wgrad_march_kernel(const TD* __restrict__ dy, const T* __restrict__ x,
                                                              float* __restrict__ ws, RunGeom r, int iters, int nruns) {
  constexpr int CV = Vec16<T>::N;
  constexpr int SEG = (WT - 1) * S + 3;
  constexpr int PITCH = 9 * CV + 1;
  extern __shared__ __attribute__((aligned(16))) float red3[];  // [256][PITCH]
  const PoolGeom& g = r.g;
  const unsigned slot = fdiv(threadIdx.x, r.d_cpr);
  const int c = threadIdx.x - slot * r.cpr;
  const int slots3 = r.slots / 3;
  const int kh = slot % 3, rsl = slot / 3;
  float acc[9][CV];  // [kt*3 + kw]
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < CV; ++e) acc[k][e] = 0.0f;
  const __amdgpu_buffer_rsrc_t rsx = span_rsrc(x, r.xspan), rsy = span_rsrc(dy, r.yspan);
  const unsigned lblk = xcd_block();
  if ((int)slot < slots3 * 3) {
    for (int it = 0; it < iters; ++it) {
      const unsigned run = (lblk * iters + it) * slots3 + rsl;
      if ((int)run >= nruns) break;
      // run -> (b, head, ho, first output column)
      const unsigned t1 = fdiv(run, r.d_runsW);
      const int w0 = (run - t1 * r.runsW) * WT;
      const unsigned t2 = fdiv(t1, r.d_Ho);
      const int ho = t1 - t2 * g.Ho;
      const unsigned b = fdiv(t2, r.d_H);
      const int hd = t2 - b * g.H;
      const int hh = ho * S + kh - 1;
      const bool rowok = hh >= 0 && hh < g.Hs;
      const unsigned xbase = b * r.bxb + hd * r.bxh + c * (CV * (unsigned)sizeof(T));
      const unsigned ybase = b * r.byb + hd * r.byh + c * (CV * (unsigned)sizeof(TD)) + (unsigned)(g.has_cls + ho * g.Wo + w0) * r.byn;
      const int col0 = w0 * S - 1;
      unsigned coff[SEG], doff[WT];
#pragma unroll
      for (int j = 0; j < SEG; ++j) {
        const int col = col0 + j;
        coff[j] = (col >= 0 && col < g.Ws) ? (unsigned)col * r.bxn : OOB;
      }
#pragma unroll
      for (int i = 0; i < WT; ++i) doff[i] = (w0 + i < g.Wo) ? i * r.byn : OOB;
      auto load_row = [&](RawCV<T, CV> (&dst)[SEG], int t) {
        const unsigned row = (rowok && t >= 0 && t < g.T) ? xbase + (unsigned)(g.has_cls + (t * g.Hs + hh) * g.Ws) * r.bxn : OOB;
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
          if (S > 2 && (j % S) > 2) continue;
          dst[j] = load_raw_buf<T, CV>(rsx, sat_add(row, coff[j]));
        }
      };
      auto load_dy = [&](RawCV<TD, CV> (&dst)[WT], int to) {
        const unsigned row = (to < g.To) ? ybase + (unsigned)(to * g.Ho * g.Wo) * r.byn : OOB;
#pragma unroll
        for (int i = 0; i < WT; ++i) dst[i] = load_raw_buf<TD, CV>(rsy, sat_add(row, doff[i]));
      };
      RawCV<T, CV> rows[3][SEG];
      RawCV<TD, CV> dcur[WT], dnext[WT];
      load_row(rows[0], -1);  // zeros
      load_row(rows[1], 0);
      load_dy(dcur, 0);
#pragma unroll 1
      for (int to = 0; to < g.To; ++to) {
        load_row(rows[2], to + 1);
        load_dy(dnext, to + 1);
        float dv[WT][CV];
#pragma unroll
        for (int i = 0; i < WT; ++i) raw_to_f32<TD, CV>(dcur[i], dv[i]);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
          for (int j = 0; j < SEG; ++j) {
            if (S > 2 && (j % S) > 2) continue;
            float f[CV];
            raw_to_f32<T, CV>(rows[kt][j], f);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              if ((j - kw) >= 0 && (j - kw) % S == 0 && (j - kw) / S < WT) {
#pragma unroll
                for (int e = 0; e < CV; ++e) acc[kt * 3 + kw][e] += dv[(j - kw) / S][e] * f[e];
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < SEG; ++j) { rows[0][j] = rows[1][j]; rows[1][j] = rows[2][j]; }
#pragma unroll
        for (int i = 0; i < WT; ++i) dcur[i] = dnext[i];
      }
    }
  }
  float* mine = red3 + threadIdx.x * PITCH;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < CV; ++e) mine[k * CV + e] = acc[k][e];
  __syncthreads();
  for (int o = threadIdx.x; o < 27 * g.C; o += 256) {   // o walks dw [C][27]: contiguous float atomics (see dwconv_bwd_weight_kernel)
    const int ch = o / 27, tap = o - ch * 27;
    const int okt = tap / 9, okh = (tap - okt * 9) / 3, okw = tap % 3;
    const int oc = ch / CV, oe = ch - oc * CV;
    float sum = 0.0f;
    for (int s3 = 0; s3 < slots3; ++s3) sum += red3[((s3 * 3 + okh) * r.cpr + oc) * PITCH + (okt * 3 + okw) * CV + oe];
    atomicAdd(ws + o, sum);
  }
}

RunGeom make_run_geom(const PoolGeom& g, int vec, int WT) {
  RunGeom r;
  r.g = g;
  r.cpr = g.C / vec;
  r.slots = 256 / r.cpr;
  r.runsW = (g.Wo + WT - 1) / WT;
  r.runs = g.B * g.H * g.To * g.Ho * r.runsW;
  r.d_cpr = mkdiv(r.cpr); r.d_runsW = mkdiv(r.runsW); r.d_Ho = mkdiv(g.Ho); r.d_To = mkdiv(g.To); r.d_H = mkdiv(g.H);
  r.d_Ws = mkdiv(g.Ws); r.d_Hs = mkdiv(g.Hs); r.d_T = mkdiv(g.T); r.d_Wo = mkdiv(g.Wo);
  r.xspan = r.yspan = 0; r.bxb = r.bxn = r.bxh = r.byb = r.byn = r.byh = 0;
  return r;
}

// byte spans / strides for buffer addressing; esx / esy = element sizes of the tensors behind the x- and y-strides.
// false when a tensor is too large for 32-bit offsets (callers then use the generic kernels).
bool set_spans(RunGeom& r, int esx, int esy) {
  const PoolGeom& g = r.g;
  const int64_t nx = (int64_t)g.T * g.Hs * g.Ws + g.has_cls, ny = (int64_t)g.To * g.Ho * g.Wo + g.has_cls;
  const int64_t xs = ((g.B - 1) * g.sxb + (nx - 1) * g.sxn + (g.H - 1) * g.sxh + g.C) * esx;
  const int64_t ys = ((g.B - 1) * g.syb + (ny - 1) * g.syn + (g.H - 1) * g.syh + g.C) * esy;
  if (xs >= 0x7fffffffLL || ys >= 0x7fffffffLL || g.sxb < 0 || g.sxn < 0 || g.sxh < 0 || g.syb < 0 || g.syn < 0 || g.syh < 0) return false;
  r.xspan = (unsigned)xs; r.bxb = (unsigned)(g.sxb * esx); r.bxn = (unsigned)(g.sxn * esx); r.bxh = (unsigned)(g.sxh * esx);
  r.yspan = (unsigned)ys; r.byb = (unsigned)(g.syb * esy); r.byn = (unsigned)(g.syn * esy); r.byh = (unsigned)(g.syh * esy);
  return true;
}

// kernel 3x3x3, padding 1, temporal stride 1, equal spatial strides in {1,2,4,8}, index ranges the multiply-shift
// division covers
int dw3_stride(const PoolGeom& g, int vec) {
  if (g.kt != 3 || g.kh != 3 || g.kw != 3 || g.pt != 1 || g.ph != 1 || g.pw != 1 || g.st != 1 || g.sh != g.sw) return 0;
  if (g.sh != 1 && g.sh != 2 && g.sh != 4 && g.sh != 8) return 0;
  if (g.C % vec || g.C / vec > 64) return 0;
  if ((int64_t)g.B * g.H * g.T * g.Hs * g.Ws >= (1 << 25)) return 0;
  return g.sh;
}

inline unsigned blocks_for(int64_t total, int per_block, int cap) {
  int64_t b = (total + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

int geom_ok(const PoolGeom& g, int vec) {
  if (g.B <= 0 || g.H <= 0 || g.C <= 0 || g.T <= 0 || g.Hs <= 0 || g.Ws <= 0) return DCC_ERR_ARG;
  if (g.C % vec || g.C > 1024) return DCC_ERR_UNSUPPORTED;
  if (g.kt < 1 || g.kt > 3 || g.kh < 1 || g.kh > 3 || g.kw < 1 || g.kw > 3) return DCC_ERR_UNSUPPORTED;
  const int64_t s[] = {g.sxb, g.sxn, g.sxh, g.syb, g.syn, g.syh};
  for (int64_t v : s)
    if (v % vec) return DCC_ERR_UNSUPPORTED;
  return DCC_OK;
}

PoolGeom make_geom(const int* d, const int64_t* strides) {
  PoolGeom g;
  g.B = d[0]; g.H = d[1]; g.C = d[2]; g.T = d[3]; g.Hs = d[4]; g.Ws = d[5];
  g.kt = d[6]; g.kh = d[7]; g.kw = d[8]; g.st = d[9]; g.sh = d[10]; g.sw = d[11];
  g.pt = d[12]; g.ph = d[13]; g.pw = d[14]; g.has_cls = d[15];
  g.To = (g.T + 2 * g.pt - g.kt) / g.st + 1;
  g.Ho = (g.Hs + 2 * g.ph - g.kh) / g.sh + 1;
  g.Wo = (g.Ws + 2 * g.pw - g.kw) / g.sw + 1;
  g.sxb = strides[0]; g.sxn = strides[1]; g.sxh = strides[2];
  g.syb = strides[3]; g.syn = strides[4]; g.syh = strides[5];
  return g;
}

}  // namespace

extern "C" int dcc_im2col3d(const float* x, void* col, int B, int T, int H, int W, int Cin, int KT, int KH, int KW,
                            int st, int sh, int sw, int pt, int ph, int pw, int Kp, int dtype, void* stream) {
  if (!x || !col || B <= 0) return DCC_ERR_ARG;
  const int To = (T + 2 * pt - KT) / st + 1, Ho = (H + 2 * ph - KH) / sh + 1, Wo = (W + 2 * pw - KW) / sw + 1;
  if (Kp < KT * KH * KW * Cin) return DCC_ERR_ARG;
  const int64_t ntok = (int64_t)B * To * Ho * Wo;
  dim3 grid((unsigned)((ntok + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  {
    const int cv = dtype == DCC_BF16 ? 8 : 4;
    const int64_t xelems = (int64_t)B * T * H * W * Cin;
    if ((dtype == DCC_BF16 || dtype == DCC_F32) && Kp % cv == 0 && Kp / cv <= 256 && xelems * 4 < 0x7fffffffLL && ntok < (1LL << 30) &&
        (((uintptr_t)col) & 15) == 0) {
      Im2colGeom g{B, T, H, W, Cin, KT, KH, KW, st, sh, sw, pt, ph, pw, To, Ho, Wo, Kp};
      g.cpt = Kp / cv; g.tpb = 256 / g.cpt; g.ntok = (unsigned)ntok; g.xbytes = (unsigned)(xelems * 4);
      g.d_cpt = mkdiv(g.cpt); g.d_Wo = mkdiv(Wo); g.d_Ho = mkdiv(Ho); g.d_To = mkdiv(To);
      int64_t nb = (ntok + g.tpb - 1) / g.tpb;
      if (nb > 16384) nb = 16384;
      if (dtype == DCC_BF16 && launch_im2col_row<float>(x, (bf16_t*)col, g, nullptr, nullptr, s)) return dcc_check_launch();
      if (dtype == DCC_BF16) hipLaunchKernelGGL((im2col3d_chunk_kernel<bf16_t, float>), dim3((unsigned)nb), dim3(256), 0, s, x, (bf16_t*)col, g, nullptr, nullptr);
      else hipLaunchKernelGGL((im2col3d_chunk_kernel<float, float>), dim3((unsigned)nb), dim3(256), 0, s, x, (float*)col, g, nullptr, nullptr);
      return dcc_check_launch();
    }
  }
  if (dtype == DCC_BF16)
    hipLaunchKernelGGL((im2col3d_kernel<bf16_t>), grid, dim3(256), 0, s, x, (bf16_t*)col, B, T, H, W, Cin, KT, KH, KW,
                       st, sh, sw, pt, ph, pw, To, Ho, Wo, Kp);
  else if (dtype == DCC_F32)
    hipLaunchKernelGGL((im2col3d_kernel<float>), grid, dim3(256), 0, s, x, (float*)col, B, T, H, W, Cin, KT, KH, KW, st,
                       sh, sw, pt, ph, pw, To, Ho, Wo, Kp);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

// uint8 frames + per-channel mean / std (pixel units) -> normalised im2col rows; same column order as dcc_im2col3d
extern "C" int dcc_im2col3d_u8(const uint8_t* x, const float* mean, const float* stdv, void* col, int B, int T, int H, int W,
                               int Cin, int KT, int KH, int KW, int st, int sh, int sw, int pt, int ph, int pw, int Kp,
                               int dtype, void* stream) {
  if (!x || !mean || !stdv || !col || B <= 0) return DCC_ERR_ARG;
  const int To = (T + 2 * pt - KT) / st + 1, Ho = (H + 2 * ph - KH) / sh + 1, Wo = (W + 2 * pw - KW) / sw + 1;
  if (Kp < KT * KH * KW * Cin) return DCC_ERR_ARG;
  const int64_t ntok = (int64_t)B * To * Ho * Wo;
  const int cv = dtype == DCC_BF16 ? 8 : 4;
  const int64_t xelems = (int64_t)B * T * H * W * Cin;
  if (!(dtype == DCC_BF16 || dtype == DCC_F32) || Kp % cv || Kp / cv > 256 || xelems >= 0x7fffffffLL || ntok >= (1LL << 30) ||
      (((uintptr_t)col) & 15))
    return DCC_ERR_UNSUPPORTED;
  Im2colGeom g{B, T, H, W, Cin, KT, KH, KW, st, sh, sw, pt, ph, pw, To, Ho, Wo, Kp};
  g.cpt = Kp / cv; g.tpb = 256 / g.cpt; g.ntok = (unsigned)ntok; g.xbytes = (unsigned)xelems;
  g.d_cpt = mkdiv(g.cpt); g.d_Wo = mkdiv(Wo); g.d_Ho = mkdiv(Ho); g.d_To = mkdiv(To);
  int64_t nb = (ntok + g.tpb - 1) / g.tpb;
  if (nb > 16384) nb = 16384;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DCC_BF16 && launch_im2col_row<uint8_t>(x, (bf16_t*)col, g, mean, stdv, s)) return dcc_check_launch();
  if (dtype == DCC_BF16) hipLaunchKernelGGL((im2col3d_chunk_kernel<bf16_t, uint8_t>), dim3((unsigned)nb), dim3(256), 0, s, x, (bf16_t*)col, g, mean, stdv);
  else hipLaunchKernelGGL((im2col3d_chunk_kernel<float, uint8_t>), dim3((unsigned)nb), dim3(256), 0, s, x, (float*)col, g, mean, stdv);
  return dcc_check_launch();
}

template <typename T, typename TD, int S, bool MIRROR>
int launch_dw3_run(const void* x, void* y, const float* w, const PoolGeom& g, hipStream_t s) {
  constexpr int vec = Vec16<T>::N;
  const int cpr = g.C / vec;
  // runs of 7 output columns when the grid allows (56/28/14/7) and still leaves >= 2 waves per SIMD, else runs of 4
  const int64_t threads7 = (int64_t)g.B * g.H * g.To * g.Ho * ((g.Wo + 6) / 7) * cpr;
  bool w7 = (g.Wo % 7 == 0) && threads7 >= 131072;
  static const int pool_wt = getenv("DCC_POOL_WT") ? atoi(getenv("DCC_POOL_WT")) : 0;  // A/B knob for tools/pool_bench.py (read once)
  if (pool_wt) w7 = w7 && pool_wt == 7;
  RunGeom r = make_run_geom(g, vec, w7 ? 7 : 4);
  if (!set_spans(r, sizeof(TD), sizeof(T))) return DCC_ERR_UNSUPPORTED;  // caller falls back to the generic kernel
  const int items = r.runs + (g.has_cls ? g.B * g.H : 0);
  const dim3 grid((items + r.slots - 1) / r.slots);
  const size_t lds = (size_t)27 * g.C * 4;
  if (w7) hipLaunchKernelGGL((dw3_run_kernel<T, TD, S, 7, MIRROR>), grid, dim3(256), lds, s, (const TD*)x, (T*)y, w, r);
  else hipLaunchKernelGGL((dw3_run_kernel<T, TD, S, 4, MIRROR>), grid, dim3(256), lds, s, (const TD*)x, (T*)y, w, r);
  return dcc_check_launch();
}

template <typename T, typename TD, bool MIRROR>
int launch_dw3_run_s(int S, const void* x, void* y, const float* w, const PoolGeom& g, hipStream_t s) {
  if (S == 1) return launch_dw3_run<T, TD, 1, MIRROR>(x, y, w, g, s);
  if constexpr (!MIRROR) {
    if (S == 2) return launch_dw3_run<T, TD, 2, false>(x, y, w, g, s);
    if (S == 4) return launch_dw3_run<T, TD, 4, false>(x, y, w, g, s);
    if (S == 8) return launch_dw3_run<T, TD, 8, false>(x, y, w, g, s);
  }
  return DCC_ERR_UNSUPPORTED;
}

// dims: [B,H,C,T,Hs,Ws,kt,kh,kw,st,sh,sw,pt,ph,pw,has_cls]; strides: [sxb,sxn,sxh,syb,syn,syh]
// mode 0 = depthwise conv (w = [C][kt*kh*kw] fp32), mode 1 = max pool (argmax uint8 [B,No,H,C] or null)
extern "C" int dcc_pool_fwd(const void* x, void* y, const float* w, uint8_t* argmax, const int* dims,
                            const int64_t* strides, int mode, int dtype, void* stream) {
  if (!x || !y || !dims || !strides || (mode == 0 && !w)) return DCC_ERR_ARG;
  PoolGeom g = make_geom(dims, strides);
  const int vec = dtype == DCC_BF16 ? 8 : 4;
  int rc = geom_ok(g, vec);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) {
    // measured (round 1, tools/pool_bench.py today): with buffer-addressed loads the run kernel wins for every stride of the model
    const int S = dw3_stride(g, vec);
    int rr = DCC_ERR_UNSUPPORTED;
    static const int smax = getenv("DCC_POOL_RUN_SMAX") ? atoi(getenv("DCC_POOL_RUN_SMAX")) : 8;
    if (S >= 1 && S <= smax && dtype == DCC_BF16) rr = launch_dw3_run_s<bf16_t, bf16_t, false>(S, x, y, w, g, s);
    else if (S >= 1 && S <= smax && dtype == DCC_F32) rr = launch_dw3_run_s<float, float, false>(S, x, y, w, g, s);
    if (rr != DCC_ERR_UNSUPPORTED) return rr;
  }
  if (mode == 1 && g.kt == 1 && g.kh == 3 && g.kw == 3 && g.st == 1 && g.sh == 2 && g.sw == 2 && g.pt == 0 && g.ph == 1 &&
      g.pw == 1 && (g.C / vec) <= 128 && (int64_t)g.B * g.H * g.T * g.Hs * g.Ws < (1 << 25) &&
      (dtype == DCC_BF16 || dtype == DCC_F32)) {
    RunGeom r = make_run_geom(g, vec, 4);
    if (set_spans(r, dtype == DCC_BF16 ? 2 : 4, dtype == DCC_BF16 ? 2 : 4)) {
      const int items = g.B * g.H * g.To * g.Ho * g.Wo + (g.has_cls ? g.B * g.H : 0);
      const dim3 grid2((items + r.slots - 1) / r.slots);
      if (dtype == DCC_BF16) hipLaunchKernelGGL((maxpool_s2_fwd_kernel<bf16_t>), grid2, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, argmax, r);
      else hipLaunchKernelGGL((maxpool_s2_fwd_kernel<float>), grid2, dim3(256), 0, s, (const float*)x, (float*)y, argmax, r);
      return dcc_check_launch();
    }
  }
  const int64_t total = (int64_t)g.B * (g.To * g.Ho * g.Wo + g.has_cls) * g.H * (g.C / vec);
  dim3 grid(blocks_for(total, 256, 8192));
  const size_t lds = mode == 0 ? (size_t)g.kt * g.kh * g.kw * g.C * 4 : 0;
  if (dtype == DCC_BF16) {
    if (mode == 0) hipLaunchKernelGGL((pool_fwd_kernel<bf16_t, false>), grid, dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, w, argmax, g);
    else hipLaunchKernelGGL((pool_fwd_kernel<bf16_t, true>), grid, dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, w, argmax, g);
  } else if (dtype == DCC_F32) {
    if (mode == 0) hipLaunchKernelGGL((pool_fwd_kernel<float, false>), grid, dim3(256), lds, s, (const float*)x, (float*)y, w, argmax, g);
    else hipLaunchKernelGGL((pool_fwd_kernel<float, true>), grid, dim3(256), lds, s, (const float*)x, (float*)y, w, argmax, g);
  } else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

static int pool_ln_launch(const PoolLnBr& b0, const PoolLnBr& b1, int nbr, const int* dims, const int64_t* strides, float eps, int dtype,
                          hipStream_t s) {
  const PoolLnBr* bs[2] = {&b0, &b1};
  for (int i = 0; i < nbr; ++i) {   // argument errors before domain checks
    const PoolLnBr& b = *bs[i];
    if (!b.x || !b.y || !b.o || !b.w || !b.gamma || !b.beta || !b.mean || !b.rstd) return DCC_ERR_ARG;
    if (b.aug < 0 || b.aug % 32) return DCC_ERR_ARG;
  }
  for (int i = 0; i < nbr; ++i)
    if (bs[i]->aug > 64) return DCC_ERR_UNSUPPORTED;   // wider one-hot key augmentations (k_t + k_h + k_w > 64): conv + LayerNorm + relpos_kaug
  if (dtype != DCC_BF16) return DCC_ERR_UNSUPPORTED;
  PoolGeom g = make_geom(dims, strides);
  if (g.B <= 0 || g.H <= 0 || g.T <= 0 || g.Hs <= 0 || g.Ws <= 0) return DCC_ERR_ARG;
  if (g.C != 96) return DCC_ERR_UNSUPPORTED;
  const int S = dw3_stride(g, 2);
  if (S == 0) return DCC_ERR_UNSUPPORTED;
  const int64_t st[] = {g.sxb, g.sxn, g.sxh, g.syb, g.syn, g.syh};
  for (int64_t v : st)
    if (v % 2) return DCC_ERR_UNSUPPORTED;  // 4-byte aligned 12-byte accesses
  for (int i = 0; i < nbr; ++i) {
    const PoolLnBr& b = *bs[i];
    if ((b.ldo % 2) || (b.aug && (b.ldo < 96 + b.aug || g.To + g.Ho + g.Wo > b.aug))) return DCC_ERR_UNSUPPORTED;
    if ((((uintptr_t)b.x) | ((uintptr_t)b.y) | ((uintptr_t)b.o)) & 3) return DCC_ERR_UNSUPPORTED;
  }
  const int WT = (S == 1 && g.Wo % 7 == 0) ? 7 : 4;
  RunGeom r = make_run_geom(g, 6, WT);
  if (!set_spans(r, 2, 2)) return DCC_ERR_UNSUPPORTED;
  const int items = r.runs + (g.has_cls ? g.B * g.H : 0);
  unsigned gx = (items + 15) / 16;
  if (nbr == 2) gx = (gx + 7) & ~7u;   // both y slices start on XCD 0: the XCD-contiguous order holds per slice (extra blocks exit)
  const dim3 grid(gx, nbr);
  static const int pd = getenv("DCC_POOL_LN_PD") ? atoi(getenv("DCC_POOL_LN_PD")) : 1;  // A/B knob of tools/pool_bench.py
#define POOL_LN(S_, WT_)                                                                                                        \
  do {                                                                                                                          \
    if (pd == 0) hipLaunchKernelGGL((dw3_ln_kernel<S_, WT_, 0>), grid, dim3(256), 0, s, b0, b1, eps, r); \
    else hipLaunchKernelGGL((dw3_ln_kernel<S_, WT_, 1>), grid, dim3(256), 0, s, b0, b1, eps, r); \
  } while (0)
  if (S == 1 && WT == 7) POOL_LN(1, 7);
  else if (S == 1) POOL_LN(1, 4);
  else if (S == 2) POOL_LN(2, 4);
  else if (S == 4) POOL_LN(4, 4);
  else POOL_LN(8, 4);
#undef POOL_LN
  return dcc_check_launch();
}

// Fused depthwise-conv pool + LayerNorm over the head dimension (bf16, C == 96, kernel 3x3x3 / padding 1 / temporal
// stride 1 / spatial stride 1, 2, 4 or 8).  y: conv output [B,No,H,C] (y-strides), o: LayerNorm(y) rows [(b,n,h)] with row
// stride ldo, mean / rstd fp32 per row.  DCC_ERR_UNSUPPORTED outside that domain (callers use dcc_pool_fwd + dcc_layernorm_fwd).
extern "C" int dcc_pool_ln_fwd(const void* x, void* y, void* o, int64_t ldo, const float* w, const float* gamma,
                               const float* beta, float* mean, float* rstd, const int* dims, const int64_t* strides, float eps,
                               int aug, int dtype, void* stream) {
  if (!dims || !strides) return DCC_ERR_ARG;
  const PoolLnBr b{(const bf16_t*)x, (bf16_t*)y, (bf16_t*)o, ldo, w, gamma, beta, mean, rstd, aug};
  return pool_ln_launch(b, b, 1, dims, strides, eps, dtype, (hipStream_t)stream);
}

// Two pools of identical geometry and strides (a block's k and v) in one launch; ptrs = {x, y, o, w, gamma, beta, mean, rstd} per branch.
extern "C" int dcc_pool_ln_fwd2(const void* const* ptrs0, const void* const* ptrs1, int64_t ldo0, int64_t ldo1, int aug0, int aug1,
                                const int* dims, const int64_t* strides, float eps, int dtype, void* stream) {
  if (!ptrs0 || !ptrs1 || !dims || !strides) return DCC_ERR_ARG;
  auto mk = [](const void* const* p, int64_t ldo, int aug) {
    return PoolLnBr{(const bf16_t*)p[0], (bf16_t*)p[1], (bf16_t*)p[2], ldo, (const float*)p[3], (const float*)p[4], (const float*)p[5],
                    (float*)p[6], (float*)p[7], aug};
  };
  return pool_ln_launch(mk(ptrs0, ldo0, aug0), mk(ptrs1, ldo1, aug1), 2, dims, strides, eps, dtype, (hipStream_t)stream);
}

// dy has the y-strides, dx the x-strides. grad_dtype = element type of dy (T or fp32).
extern "C" int dcc_pool_bwd_data(const void* dy, void* dx, const float* w, const uint8_t* argmax, const int* dims,
                                 const int64_t* strides, int mode, int dtype, int grad_dtype, void* stream) {
  if (!dy || !dx || !dims || !strides || (mode == 0 && !w) || (mode == 1 && !argmax)) return DCC_ERR_ARG;
  PoolGeom g = make_geom(dims, strides);
  const int vec = dtype == DCC_BF16 ? 8 : 4;
  int rc = geom_ok(g, vec);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 1 && g.kt == 1 && g.kh == 3 && g.kw == 3 && g.st == 1 && g.sh == 2 && g.sw == 2 && g.pt == 0 && g.ph == 1 &&
      g.pw == 1 && (g.C / vec) <= 128 && (int64_t)g.B * g.H * g.T * g.Hs * g.Ws < (1 << 25)) {
    const RunGeom r = make_run_geom(g, vec, 4);
    const int items = g.B * g.H * g.T * g.Hs * g.Ws + (g.has_cls ? g.B * g.H : 0);
    const dim3 grid((items + r.slots - 1) / r.slots);
    if (dtype == DCC_BF16 && grad_dtype == DCC_BF16) hipLaunchKernelGGL((maxpool_s2_bwd_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, argmax, r);
    else if (dtype == DCC_BF16 && grad_dtype == DCC_F32) hipLaunchKernelGGL((maxpool_s2_bwd_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const float*)dy, (bf16_t*)dx, argmax, r);
    else if (dtype == DCC_F32 && grad_dtype == DCC_F32) hipLaunchKernelGGL((maxpool_s2_bwd_kernel<float, float>), grid, dim3(256), 0, s, (const float*)dy, (float*)dx, argmax, r);
    else return DCC_ERR_UNSUPPORTED;
    return dcc_check_launch();
  }
  if (mode == 0) {
    const int S = dw3_stride(g, vec);
    if (S == 1) {
      // stride 1: the mirrored stencil over dy; the run kernel reads through the x-strides and writes through the y-strides
      PoolGeom m = g;
      m.sxb = g.syb; m.sxn = g.syn; m.sxh = g.syh; m.syb = g.sxb; m.syn = g.sxn; m.syh = g.sxh;
      int rr = DCC_ERR_UNSUPPORTED;
      if (dtype == DCC_BF16 && grad_dtype == DCC_BF16) rr = launch_dw3_run_s<bf16_t, bf16_t, true>(1, dy, dx, w, m, s);
      else if (dtype == DCC_BF16 && grad_dtype == DCC_F32) rr = launch_dw3_run_s<bf16_t, float, true>(1, dy, dx, w, m, s);
      else if (dtype == DCC_F32 && grad_dtype == DCC_F32) rr = launch_dw3_run_s<float, float, true>(1, dy, dx, w, m, s);
      if (rr != DCC_ERR_UNSUPPORTED) return rr;
    } else if (S > 1) {
      static const bool quad = !getenv("DCC_POOL_NO_QUAD");
      if (quad && S == 2 && g.Hs == 2 * g.Ho && g.Ws == 2 * g.Wo) {
        const bool w7 = (g.Wo % 7 == 0);
        RunGeom rq = make_run_geom(g, vec, w7 ? 7 : 4);
        if (set_spans(rq, dtype == DCC_BF16 ? 2 : 4, grad_dtype == DCC_BF16 ? 2 : 4)) {
          const int items = rq.runs * 2 + (g.has_cls ? g.B * g.H : 0);
          const dim3 grid((items + rq.slots - 1) / rq.slots);
          const size_t ldsq = (size_t)27 * g.C * 4;
#define DW3Q(T_, TD_) do { if (w7) hipLaunchKernelGGL((dw3_bwd_data_s2_kernel<T_, TD_, 7>), grid, dim3(256), ldsq, s, (const TD_*)dy, (T_*)dx, w, rq); \
                           else hipLaunchKernelGGL((dw3_bwd_data_s2_kernel<T_, TD_, 4>), grid, dim3(256), ldsq, s, (const TD_*)dy, (T_*)dx, w, rq); \
                           return dcc_check_launch(); } while (0)
          if (dtype == DCC_BF16 && grad_dtype == DCC_BF16) DW3Q(bf16_t, bf16_t);
          if (dtype == DCC_BF16 && grad_dtype == DCC_F32) DW3Q(bf16_t, float);
          if (dtype == DCC_F32 && grad_dtype == DCC_F32) DW3Q(float, float);
#undef DW3Q
        }
      }
      if (quad && S >= 4) {
        const bool w7 = (g.Wo % 7 == 0);
        RunGeom rq = make_run_geom(g, vec, w7 ? 7 : 4);
        if (set_spans(rq, dtype == DCC_BF16 ? 2 : 4, grad_dtype == DCC_BF16 ? 2 : 4)) {
          const int items = g.B * g.H * g.T * g.Hs * rq.runsW + (g.has_cls ? g.B * g.H : 0);
          const dim3 grid((items + rq.slots - 1) / rq.slots);
          const size_t ldsq = (size_t)27 * g.C * 4;
#define DW3WD(T_, TD_, S_) do { if (w7) hipLaunchKernelGGL((dw3_bwd_data_wide_kernel<T_, TD_, S_, 7>), grid, dim3(256), ldsq, s, (const TD_*)dy, (T_*)dx, w, rq); \
                                else hipLaunchKernelGGL((dw3_bwd_data_wide_kernel<T_, TD_, S_, 4>), grid, dim3(256), ldsq, s, (const TD_*)dy, (T_*)dx, w, rq); \
                                return dcc_check_launch(); } while (0)
#define DW3WD_S(T_, TD_) do { if (S == 4) DW3WD(T_, TD_, 4); else DW3WD(T_, TD_, 8); } while (0)
          if (dtype == DCC_BF16 && grad_dtype == DCC_BF16) DW3WD_S(bf16_t, bf16_t);
          if (dtype == DCC_BF16 && grad_dtype == DCC_F32) DW3WD_S(bf16_t, float);
          if (dtype == DCC_F32 && grad_dtype == DCC_F32) DW3WD_S(float, float);
#undef DW3WD_S
#undef DW3WD
        }
      }
      const RunGeom r = make_run_geom(g, vec, 4);
      const int items = g.B * g.H * g.T * g.Hs * g.Ws + (g.has_cls ? g.B * g.H : 0);
      const dim3 grid((items + r.slots - 1) / r.slots);
      const size_t lds3 = (size_t)27 * g.C * 4;
#define DW3S(T_, TD_, S_) hipLaunchKernelGGL((dw3_bwd_data_strided_kernel<T_, TD_, S_>), grid, dim3(256), lds3, s, (const TD_*)dy, (T_*)dx, w, r)
#define DW3S_ALL(T_, TD_) do { if (S == 2) DW3S(T_, TD_, 2); else if (S == 4) DW3S(T_, TD_, 4); else DW3S(T_, TD_, 8); return dcc_check_launch(); } while (0)
      if (dtype == DCC_BF16 && grad_dtype == DCC_BF16) DW3S_ALL(bf16_t, bf16_t);
      if (dtype == DCC_BF16 && grad_dtype == DCC_F32) DW3S_ALL(bf16_t, float);
      if (dtype == DCC_F32 && grad_dtype == DCC_F32) DW3S_ALL(float, float);
#undef DW3S_ALL
#undef DW3S
    }
  }
  const int64_t total = (int64_t)g.B * (g.T * g.Hs * g.Ws + g.has_cls) * g.H * (g.C / vec);
  dim3 grid(blocks_for(total, 256, 8192));
  const size_t lds = mode == 0 ? (size_t)g.kt * g.kh * g.kw * g.C * 4 : 0;
#define PBD(T_, TD_, MP) hipLaunchKernelGGL((pool_bwd_data_kernel<T_, TD_, MP>), grid, dim3(256), lds, s, (const TD_*)dy, (T_*)dx, w, argmax, g)
  if (dtype == DCC_BF16 && grad_dtype == DCC_BF16) { if (mode == 0) PBD(bf16_t, bf16_t, false); else PBD(bf16_t, bf16_t, true); }
  else if (dtype == DCC_BF16 && grad_dtype == DCC_F32) { if (mode == 0) PBD(bf16_t, float, false); else PBD(bf16_t, float, true); }
  else if (dtype == DCC_F32 && grad_dtype == DCC_F32) { if (mode == 0) PBD(float, float, false); else PBD(float, float, true); }
  else return DCC_ERR_UNSUPPORTED;
#undef PBD
  return dcc_check_launch();
}

// dw [C][taps] fp32 is ACCUMULATED into (float atomics, one 10-KB partial per block).  workspace: unused, may be null.
extern "C" int dcc_dwconv_bwd_weight(const void* dy, const void* x, float* dw, float* workspace, const int* dims,
                                     const int64_t* strides, int dtype, int grad_dtype, void* stream) {
  if (!dy || !x || !dw || !dims || !strides) return DCC_ERR_ARG;
  (void)workspace;   // r02's per-block partial slabs: unused since every block adds into dw itself (kept in the ABI)
  PoolGeom g = make_geom(dims, strides);
  int rc = geom_ok(g, dtype == DCC_BF16 ? 8 : 4);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  {
    const int vec = dtype == DCC_BF16 ? 8 : 4;
    const int S = dw3_stride(g, vec);
    if (S && (256 / (g.C / vec)) >= 3) {
      // marching kernel (default): runs of WM = 2 / 4 output columns keep the three-row window + 72 accumulators at
      // 2-3 waves per SIMD; DCC_WGRAD_NO_MARCH=1 selects the per-temporal-tap kernel with runs of 7 / 4
      static const bool march = !getenv("DCC_WGRAD_NO_MARCH");
      static const int wm_env = getenv("DCC_WGRAD_WM") ? atoi(getenv("DCC_WGRAD_WM")) : 0;
      const int WM = wm_env == 2 || wm_env == 4 ? wm_env : ((g.Wo % 4 == 0 && S == 1) ? 4 : 2);
      const bool w7 = (g.Wo % 7 == 0);
      RunGeom r = make_run_geom(g, vec, march ? WM : (w7 ? 7 : 4));
      const bool spans_ok = set_spans(r, dtype == DCC_BF16 ? 2 : 4, grad_dtype == DCC_BF16 ? 2 : 4);
      const int slots3 = r.slots / 3;
      const int nruns = march ? g.B * g.H * g.Ho * r.runsW : r.runs;  // marching threads walk the frames themselves
      const int total_iters = (nruns + slots3 - 1) / slots3;
      int nblocks = total_iters < 512 ? total_iters : 512;  // <= DCC_DWCONV_WS_BLOCKS; two 75-KB blocks per CU
      if (nblocks < 1) nblocks = 1;
      const int iters = (total_iters + nblocks - 1) / nblocks;
      nblocks = (total_iters + iters - 1) / iters;
      const size_t lds3 = (size_t)256 * (9 * vec + 1) * 4;
#define DW3K(KERN_, ...)                                                                                            \
  do {                                                                                                              \
    auto kern = KERN_;                                                                                              \
    static unsigned long long cfg = 0;                                                                              \
    if (!(cfg & dcc_dev_bit())) {                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess) return DCC_ERR_LAUNCH; \
      cfg |= dcc_dev_bit();                                                                                         \
    }                                                                                                               \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds3, s, __VA_ARGS__);                                       \
  } while (0)
#define DW3W(T_, TD_, S_)                                                                                           \
  do {                                                                                                              \
    if (march && WM == 4) DW3K((dw3_wgrad_march_kernel<T_, TD_, S_, 4>), (const TD_*)dy, (const T_*)x, dw, r, iters, nruns); \
    else if (march) DW3K((dw3_wgrad_march_kernel<T_, TD_, S_, 2>), (const TD_*)dy, (const T_*)x, dw, r, iters, nruns); \
    else if (w7) DW3K((dw3_wgrad_kernel<T_, TD_, S_, 7>), (const TD_*)dy, (const T_*)x, dw, r, iters);       \
    else DW3K((dw3_wgrad_kernel<T_, TD_, S_, 4>), (const TD_*)dy, (const T_*)x, dw, r, iters);               \
  } while (0)
#define DW3W_ALL(T_, TD_) do { if (S == 1) DW3W(T_, TD_, 1); else if (S == 2) DW3W(T_, TD_, 2); else if (S == 4) DW3W(T_, TD_, 4); else DW3W(T_, TD_, 8); } while (0)
      bool done = true;
      if (!spans_ok) done = false;
      else if (dtype == DCC_BF16 && grad_dtype == DCC_BF16) DW3W_ALL(bf16_t, bf16_t);
      else if (dtype == DCC_BF16 && grad_dtype == DCC_F32) DW3W_ALL(bf16_t, float);
      else if (dtype == DCC_F32 && grad_dtype == DCC_F32) DW3W_ALL(float, float);
      else done = false;
#undef DW3W_ALL
#undef DW3W
#undef DW3K
      if (done) return dcc_check_launch();   // every block added its partial into dw itself
    }
  }
  const int cv = dtype == DCC_BF16 ? 8 : 4;
  if (g.C % cv || g.C / cv > 85) return DCC_ERR_UNSUPPORTED;
  const int slots = (256 / (g.C / cv)) / 3 * 3;
  if (slots < 3) return DCC_ERR_UNSUPPORTED;
  const int taps = g.kt * g.kh * g.kw;
  const int64_t total = (int64_t)g.B * g.H * g.To * g.Ho * g.Wo * 3;
  dim3 grid(blocks_for(total, slots * 4, 1024));  // <= DCC_DWCONV_WS_BLOCKS
  const size_t lds = (size_t)taps * g.C * 4;
  if (dtype == DCC_BF16 && grad_dtype == DCC_BF16)
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<bf16_t, bf16_t>), grid, dim3(256), lds, s, (const bf16_t*)dy, (const bf16_t*)x, dw, g);
  else if (dtype == DCC_BF16 && grad_dtype == DCC_F32)
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<bf16_t, float>), grid, dim3(256), lds, s, (const float*)dy, (const bf16_t*)x, dw, g);
  else if (dtype == DCC_F32 && grad_dtype == DCC_F32)
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<float, float>), grid, dim3(256), lds, s, (const float*)dy, (const float*)x, dw, g);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}
