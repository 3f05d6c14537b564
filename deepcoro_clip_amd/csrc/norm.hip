// LayerNorm forward / backward (HBM-bound; wave-shuffle reductions, 16-byte accesses).
// A row of C elements is owned by LPR lanes (16/32/64), each holding ITERS 16-byte chunks, so a
// 96-wide row (MViT head_dim / block-0 width) packs 4 rows per wave instead of idling 40 lanes.
// Statistics are fp32 regardless of the element type. Rows may be strided (ldx / ldy) so the
// normalised q/k can be written straight into the wider [.. , DQK] attention operand.
#include "common.h"

#ifndef LN_BWD_OCC
#define LN_BWD_OCC 1   // minimum waves per SIMD of the LayerNorm backward (A/B builds)
#endif

namespace {

template <typename T, int LPR, int ITERS>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                     int64_t ldy, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows, int C, float eps) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int RPW = 64 / LPR;  // rows per wave
  using V = vec16_t<T>;
  const int lane = threadIdx.x & 63, sub = lane % LPR, rw = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int nch = C / VEC;
  for (int64_t r0 = wave0 * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t row = r0 + rw;
    const bool rok = row < rows;
    float v[ITERS][VEC];
    float s = 0.0f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int c = it * LPR + sub;
      if (rok && c < nch) {
        const V t = *reinterpret_cast<const V*>(x + row * ldx + c * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { v[it][e] = to_f32(t[e]); s += v[it][e]; }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[it][e] = 0.0f;
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s / C;
    float q = 0.0f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int c = it * LPR + sub;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float d = v[it][e] - mu; q += d * d; }
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rs = rsqrtf(q / C + eps);
    if (rok) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int c = it * LPR + sub;
        if (c < nch) {
          V o;
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            o[e] = from_f32<T>((v[it][e] - mu) * rs * gamma[c * VEC + e] + beta[c * VEC + e]);
          *reinterpret_cast<V*>(y + row * ldy + c * VEC) = o;
        }
      }
      if (sub == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
      }
    }
  }
}

template <typename TD, int VEC> __device__ __forceinline__ void load_vecf(const TD* p, float (&o)[VEC]) {
  if constexpr (sizeof(TD) == 2) {
    static_assert(VEC == 8, "bf16 rows use 8-element chunks");
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)t[e];
  } else {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
      o[4 * q + 0] = t[0]; o[4 * q + 1] = t[1]; o[4 * q + 2] = t[2]; o[4 * q + 3] = t[3];
    }
  }
}
template <typename TD, int VEC> __device__ __forceinline__ void store_vecf(TD* p, const float (&o)[VEC]) {
  if constexpr (sizeof(TD) == 2) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)o[e];
    *reinterpret_cast<bf16x8*>(p) = t;
  } else {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      f32x4 t = {o[4 * q + 0], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
      *reinterpret_cast<f32x4*>(p + 4 * q) = t;
    }
  }
}

// raw 16-byte registers -> VEC floats
template <typename TD, int VEC> __device__ __forceinline__ void unpack_f32(const f32x4* r, float (&o)[VEC]) {
  if constexpr (sizeof(TD) == 2) {
    const bf16x8 t = __builtin_bit_cast(bf16x8, r[0]);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)t[e];
  } else {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) { o[4 * q] = r[q][0]; o[4 * q + 1] = r[q][1]; o[4 * q + 2] = r[q][2]; o[4 * q + 3] = r[q][3]; }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dgamma += dy * xhat, dbeta += dy
// TD = element type of dy / dx (may be fp32 while x is bf16, e.g. gradients reduced with atomics).
// Parameter gradients: registers over the row loop -> LDS per block -> ONE global atomic per column per block
// (a few hundred blocks), instead of per-wave global atomics that serialise on C addresses.
template <typename T, typename TD, int LPR, int ITERS, int U>
__global__ __launch_bounds__(256, (sizeof(TD) == 2 && ITERS == 1 && U == 4 ? LN_BWD_OCC : 1)) void ln_bwd_kernel(const TD* __restrict__ dy, int64_t lddy, const T* __restrict__ x,
                                                     int64_t ldx, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     TD* __restrict__ dx, int64_t lddx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t rows, int C,
                                                     const TD* __restrict__ dres, float* __restrict__ ws) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int RPW = 64 / LPR;
  using V = vec16_t<T>;
  extern __shared__ float red[];  // [2][C]
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) red[i] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane % LPR, rw = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int nch = C / VEC;
  float gam[ITERS][VEC], dg[ITERS][VEC], db[ITERS][VEC];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int c = it * LPR + sub;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      gam[it][e] = (c < nch) ? gamma[c * VEC + e] : 0.0f;
      dg[it][e] = 0.0f;
      db[it][e] = 0.0f;
    }
  }
  // Branch-free row loop.  A wave handles U groups of RPW rows per iteration through buffer descriptors based at its
  // first row (wave-uniform, so any tensor size works with 32-bit offsets): rows past the end and the idle lanes of a
  // narrow row read zeros / drop their stores through out-of-range offsets, so every load of the iteration is in flight
  // before the first reduction — the kernel is bound by bytes in flight, and conditional loads made the compiler wait
  // for each one inside its branch.
  constexpr int G = RPW * U;
  constexpr unsigned NOPE = 0xF0000000u;
  auto rsrc = [](const void* p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes), 0x00020000);
  };
  for (int64_t r0 = wave0 * G; r0 < rows; r0 += nwaves * G) {
    const int64_t nrow = rows - r0 < G ? rows - r0 : G;
    const __amdgpu_buffer_rsrc_t rsx = rsrc(x + r0 * ldx, ((nrow - 1) * ldx + C) * (int64_t)sizeof(T));
    const __amdgpu_buffer_rsrc_t rsy = rsrc(dy + r0 * lddy, ((nrow - 1) * lddy + C) * (int64_t)sizeof(TD));
    const __amdgpu_buffer_rsrc_t rsd = rsrc(dx + r0 * lddx, ((nrow - 1) * lddx + C) * (int64_t)sizeof(TD));
    const __amdgpu_buffer_rsrc_t rsr = rsrc(dres ? dres + r0 * lddx : nullptr, dres ? ((nrow - 1) * lddx + C) * (int64_t)sizeof(TD) : 0);
    const __amdgpu_buffer_rsrc_t rsm = rsrc(mean + r0, nrow * 4), rss = rsrc(rstd + r0, nrow * 4);
    f32x4 tx[U][ITERS][sizeof(T) * VEC / 16], td[U][ITERS][sizeof(TD) * VEC / 16], tr[U][ITERS][sizeof(TD) * VEC / 16];
    float mu[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int lrow = u * RPW + rw;
      mu[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsm, lrow * 4, 0, 0));
      rs[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rss, lrow * 4, 0, 0));
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int c = it * LPR + sub;
        const bool live = c < nch;
        const unsigned ox = live ? (unsigned)((lrow * ldx + c * VEC) * sizeof(T)) : NOPE;
        const unsigned oy = live ? (unsigned)((lrow * lddy + c * VEC) * sizeof(TD)) : NOPE;
        const unsigned od = live ? (unsigned)((lrow * lddx + c * VEC) * sizeof(TD)) : NOPE;
#pragma unroll
        for (int q = 0; q < (int)(sizeof(T) * VEC / 16); ++q) tx[u][it][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, ox, 16 * q, 0));
#pragma unroll
        for (int q = 0; q < (int)(sizeof(TD) * VEC / 16); ++q) {
          td[u][it][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy, oy, 16 * q, 0));
          tr[u][it][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, od, 16 * q, 0));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int lrow = u * RPW + rw;
      float xh[ITERS][VEC], g[ITERS][VEC];
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        float xv[VEC], d[VEC];
        unpack_f32<T, VEC>(tx[u][it], xv);
        unpack_f32<TD, VEC>(td[u][it], d);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          xh[it][e] = (xv[e] - mu[u]) * rs[u];
          g[it][e] = d[e] * gam[it][e];
          s1 += g[it][e];
          s2 += g[it][e] * xh[it][e];
          dg[it][e] += d[e] * xh[it][e];
          db[it][e] += d[e];
        }
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      s1 /= C; s2 /= C;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int c = it * LPR + sub;
        float o[VEC], rr[VEC];
        unpack_f32<TD, VEC>(tr[u][it], rr);  // gradient arriving through the other use of x (residual / skip path); zeros without dres
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = rs[u] * (g[it][e] - s1 - xh[it][e] * s2) + rr[e];
        const unsigned od = c < nch ? (unsigned)((lrow * lddx + c * VEC) * sizeof(TD)) : NOPE;
        if constexpr (sizeof(TD) == 2) {
          bf16x8 t;
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = (bf16_t)o[e];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rsd, od, 0, 0);
        } else {
#pragma unroll
          for (int q = 0; q < VEC / 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]}), rsd, od, 16 * q, 0);
        }
      }
    }
  }
  // fold the wave's row groups together, then LDS across the block's waves, then one atomic per column
#pragma unroll
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
#pragma unroll
      for (int o = 32; o >= LPR; o >>= 1) {
        dg[it][e] += __shfl_xor(dg[it][e], o);
        db[it][e] += __shfl_xor(db[it][e], o);
      }
    }
  if (rw == 0) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int c = it * LPR + sub;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          atomicAdd(&red[c * VEC + e], dg[it][e]);
          atomicAdd(&red[C + c * VEC + e], db[it][e]);
        }
      }
    }
  }
  __syncthreads();
  if (!ws) {  // no workspace: one atomic per column per block straight into the gradients (grid capped at 512 blocks)
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
      if (dgamma) atomicAdd(dgamma + i, red[i]);
      if (dbeta) atomicAdd(dbeta + i, red[C + i]);
    }
    return;
  }
  // With a workspace: plain stores of the block partial, folded by ln_reduce_kernel (no atomics, so the grid size is
  // not tied to an atomic tail: ~18 ns per block on the few L2 channels that hold dgamma / dbeta)
  float* part = ws + (size_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) part[i] = red[i];
}

// out[i] += sum_b ws[b][i], i < 2C (first C -> dgamma, rest -> dbeta); 32 columns x 8 partial groups per block, the
// partials split over gridDim.y slices (a single slice walks them in latency-bound rounds of 64)
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int nblocks, int C) {
  __shared__ float part[8][33];
  const int il = threadIdx.x & 31, bg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + il, n = 2 * C;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  float acc = 0.0f;
  if (i < n) {
    int b = b0 + bg;
    for (; b + 56 < b1; b += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws[(int64_t)(b + 8 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < b1; b += 8) acc += ws[(int64_t)b * n + i];
  }
  part[bg][il] = acc;
  __syncthreads();
  if (bg == 0 && i < n) {
    float t = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) t += part[u][il];
    if (i < C) { if (dgamma) atomicAdd(dgamma + i, t); }
    else if (dbeta) atomicAdd(dbeta + i - C, t);
  }
}

inline int pick(int nch, int& lpr, int& iters) {
  if (nch <= 16) { lpr = 16; iters = 1; }
  else if (nch <= 32) { lpr = 32; iters = 1; }
  else { lpr = 64; iters = (nch + 63) / 64; }
  return iters <= 4 ? DCC_OK : DCC_ERR_UNSUPPORTED;
}

inline unsigned grid_for(int64_t rows, int rpw) {
  const int64_t waves = (rows + rpw - 1) / rpw;
  int64_t blocks = (waves + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

inline unsigned lnb_grid(int64_t rows, int rows_per_wave_iter, unsigned cap) {
  const int64_t per_block = (int64_t)rows_per_wave_iter * 4;
  int64_t blocks = (rows + per_block - 1) / per_block;
  if (blocks > cap) blocks = cap;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

template <typename T>
int ln_fwd_t(const void* x, int64_t ldx, void* y, int64_t ldy, const float* g, const float* b, float* mean, float* rstd,
             int64_t rows, int C, float eps, hipStream_t st) {
  int lpr, iters;
  if (pick(C / Vec16<T>::N, lpr, iters)) return DCC_ERR_UNSUPPORTED;
  const T* xp = (const T*)x; T* yp = (T*)y;
#define LN_F(LP, IT) hipLaunchKernelGGL((ln_fwd_kernel<T, LP, IT>), dim3(grid_for(rows, 64 / LP)), dim3(256), 0, st, xp, ldx, yp, ldy, g, b, mean, rstd, rows, C, eps)
  if (lpr == 16) LN_F(16, 1);
  else if (lpr == 32) LN_F(32, 1);
  else if (iters == 1) LN_F(64, 1);
  else if (iters == 2) LN_F(64, 2);
  else if (iters == 3) LN_F(64, 3);
  else LN_F(64, 4);
#undef LN_F
  return dcc_check_launch();
}

template <typename T, typename TD>
int ln_bwd_t(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* mean, const float* rstd,
             const float* g, void* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int C, const void* dres,
             float* ws, hipStream_t st) {
  int lpr, iters;
  if (pick(C / Vec16<T>::N, lpr, iters)) return DCC_ERR_UNSUPPORTED;
  const TD* dyp = (const TD*)dy; const T* xp = (const T*)x; TD* dxp = (TD*)dx; const TD* drp = (const TD*)dres;
  static const unsigned lnb_cap = getenv("DCC_LNB_CAP") ? (unsigned)atoi(getenv("DCC_LNB_CAP")) : 512u;
  // measured (round 1): 512 blocks up to ~30 M elements, 1024 beyond (block 0/1 per-head norms)
  static const unsigned lnb_cap_env = getenv("DCC_LNB_CAP_WS") ? (unsigned)atoi(getenv("DCC_LNB_CAP_WS")) : 0u;
  const unsigned lnb_cap_ws = lnb_cap_env ? lnb_cap_env : (rows * C >= (32 << 20) ? 1024u : 512u);
  const unsigned ws_fit = (1u << 20) / (unsigned)(2 * C);  // block partials the workspace holds (DCC_LN_WS_FLOATS in include/dcc.h)
  const unsigned ws_cap = ws_fit < lnb_cap_ws ? ws_fit : lnb_cap_ws;
  unsigned nb = 0;
  // few rows (the text tower's 768-token norms, the aggregator): a grid of <= 128 blocks pays ~2 us for the atomic tail,
  // less than the fold kernel's launch
  if (ws && rows <= 128 * 8) ws = nullptr;
  // DCC_LN_ATOMIC=1 (A/B): always the one-launch form — every block adds its dgamma / dbeta partial with contiguous float
  // atomics (256 bytes per wave-instruction) instead of storing it for ln_reduce_kernel
  static const bool ln_atomic = getenv("DCC_LN_ATOMIC") && atoi(getenv("DCC_LN_ATOMIC")) != 0;
  if (ln_atomic) ws = nullptr;
  // r03: 4 row groups in flight for one-chunk-per-lane rows (C <= 512; 170 VGPRs, three waves per SIMD) — in the step 41.47 vs 41.77 ms
  // and 40.97 vs 41.11 ms against 2 on two boxes (alternating runs); 8 (256 VGPRs) is no better
  static const int lnb_u = getenv("DCC_LNB_U") ? atoi(getenv("DCC_LNB_U")) : 4;
#define LN_BU(LP, IT, UU) hipLaunchKernelGGL((ln_bwd_kernel<T, TD, LP, IT, UU>), dim3(nb = lnb_grid(rows, 64 / LP * UU, ws ? ws_cap : lnb_cap)), dim3(256), 2 * C * sizeof(float), st, dyp, lddy, xp, ldx, mean, rstd, g, dxp, lddx, dgamma, dbeta, rows, C, drp, ws)
  // U = row groups a wave keeps in flight per iteration (only the combinations used are instantiated)
#define LN_B(LP, IT) do { if (lnb_u >= 8 && IT == 1 && LP <= 32) LN_BU(LP, IT, ((IT == 1 && LP <= 32) ? 8 : 1)); \
                          else if (lnb_u >= 4 && IT == 1) LN_BU(LP, IT, (IT == 1 ? 4 : 1)); \
                          else if (lnb_u >= 2 && IT <= 2) LN_BU(LP, IT, (IT <= 2 ? 2 : 1)); else LN_BU(LP, IT, 1); } while (0)
  if (lpr == 16) LN_B(16, 1);
  else if (lpr == 32) LN_B(32, 1);
  else if (iters == 1) LN_B(64, 1);
  else if (iters == 2) LN_B(64, 2);
  else if (iters == 3) LN_B(64, 3);
  else LN_B(64, 4);
#undef LN_B
#undef LN_BU
  int rc = dcc_check_launch();
  if (rc || !ws) return rc;
  hipLaunchKernelGGL(ln_reduce_kernel, dim3((2 * C + 31) / 32, (nb + 63) / 64), dim3(256), 0, st, ws, dgamma, dbeta, (int)nb, C);
  return dcc_check_launch();
}

}  // namespace

extern "C" int dcc_layernorm_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma,
                                 const float* beta, float* mean, float* rstd, int64_t rows, int C, float eps, int dtype,
                                 void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0) return DCC_ERR_ARG;
  const int vec = dtype == DCC_BF16 ? 8 : 4;
  if ((C % vec) || (ldx % vec) || (ldy % vec) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return DCC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCC_BF16) return ln_fwd_t<bf16_t>(x, ldx, y, ldy, gamma, beta, mean, rstd, rows, C, eps, st);
  if (dtype == DCC_F32) return ln_fwd_t<float>(x, ldx, y, ldy, gamma, beta, mean, rstd, rows, C, eps, st);
  return DCC_ERR_UNSUPPORTED;
}

// dgamma / dbeta are ACCUMULATED into (fp32, caller zeroes or owns the running gradient).
extern "C" int dcc_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* mean,
                                 const float* rstd, const float* gamma, void* dx, int64_t lddx, float* dgamma,
                                 float* dbeta, const void* dres, float* workspace, int64_t rows, int C, int dtype,
                                 int grad_dtype, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || rows <= 0 || C <= 0) return DCC_ERR_ARG;
  const int vec = dtype == DCC_BF16 ? 8 : 4;
  if ((C % vec) || (ldx % vec) || ((uintptr_t)x & 15)) return DCC_ERR_UNSUPPORTED;
  const int gvec = grad_dtype == DCC_BF16 ? 8 : 4;
  if ((lddy % gvec) || (lddx % gvec) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15) || ((uintptr_t)dres & 15)) return DCC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCC_BF16 && grad_dtype == DCC_BF16)
    return ln_bwd_t<bf16_t, bf16_t>(dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, rows, C, dres, workspace, st);
  if (dtype == DCC_BF16 && grad_dtype == DCC_F32)
    return ln_bwd_t<bf16_t, float>(dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, rows, C, dres, workspace, st);
  if (dtype == DCC_F32 && grad_dtype == DCC_F32)
    return ln_bwd_t<float, float>(dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, rows, C, dres, workspace, st);
  return DCC_ERR_UNSUPPORTED;
}
