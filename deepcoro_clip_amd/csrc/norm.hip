// LayerNorm forward / backward (HBM-bound; wave-shuffle reductions, 16-byte accesses).
// A row of C elements is owned by LPR lanes (16/32/64), each holding ITERS 16-byte chunks, so a
// 96-wide row (MViT head_dim / block-0 width) packs 4 rows per wave instead of idling 40 lanes.
// Statistics are fp32 regardless of the element type. Rows may be strided (ldx / ldy) so the
// normalised q/k can be written straight into the wider [.. , DQK] attention operand.
#include "common.h"

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

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dgamma += dy * xhat, dbeta += dy
// TD = element type of dy / dx (may be fp32 while x is bf16, e.g. gradients reduced with atomics).
// Parameter gradients: registers over the row loop -> LDS per block -> ONE global atomic per column per block
// (a few hundred blocks), instead of per-wave global atomics that serialise on C addresses.
template <typename T, typename TD, int LPR, int ITERS>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TD* __restrict__ dy, int64_t lddy, const T* __restrict__ x,
                                                     int64_t ldx, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     TD* __restrict__ dx, int64_t lddx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t rows, int C,
                                                     const TD* __restrict__ dres) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int RPW = 64 / LPR;
  using V = vec16_t<T>;
  extern __shared__ float red[];  // [2][C]
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) red[i] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane % LPR, rw = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
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
  for (int64_t r0 = wave0 * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t row = r0 + rw;
    const bool rok = row < rows;
    const float mu = rok ? mean[row] : 0.0f, rs = rok ? rstd[row] : 0.0f;
    float xh[ITERS][VEC], g[ITERS][VEC];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int c = it * LPR + sub;
      if (rok && c < nch) {
        const V tx = *reinterpret_cast<const V*>(x + row * ldx + c * VEC);
        float d[VEC];
        load_vecf<TD, VEC>(dy + row * lddy + c * VEC, d);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          xh[it][e] = (to_f32(tx[e]) - mu) * rs;
          g[it][e] = d[e] * gam[it][e];
          s1 += g[it][e];
          s2 += g[it][e] * xh[it][e];
          dg[it][e] += d[e] * xh[it][e];
          db[it][e] += d[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { xh[it][e] = 0.0f; g[it][e] = 0.0f; }
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    s1 /= C; s2 /= C;
    if (rok) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int c = it * LPR + sub;
        if (c < nch) {
          float o[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = rs * (g[it][e] - s1 - xh[it][e] * s2);
          if (dres) {  // gradient arriving through the other use of x (residual / skip path): summed here, not by autograd
            float rr[VEC];
            load_vecf<TD, VEC>(dres + row * lddx + c * VEC, rr);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] += rr[e];
          }
          store_vecf<TD, VEC>(dx + row * lddx + c * VEC, o);
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
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    if (dgamma) atomicAdd(dgamma + i, red[i]);
    if (dbeta) atomicAdd(dbeta + i, red[C + i]);
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
             hipStream_t st) {
  int lpr, iters;
  if (pick(C / Vec16<T>::N, lpr, iters)) return DCC_ERR_UNSUPPORTED;
  const TD* dyp = (const TD*)dy; const T* xp = (const T*)x; TD* dxp = (TD*)dx; const TD* drp = (const TD*)dres;
#define LN_B(LP, IT) hipLaunchKernelGGL((ln_bwd_kernel<T, TD, LP, IT>), dim3(grid_for(rows, 64 / LP) > 512 ? 512 : grid_for(rows, 64 / LP)), dim3(256), 2 * C * sizeof(float), st, dyp, lddy, xp, ldx, mean, rstd, g, dxp, lddx, dgamma, dbeta, rows, C, drp)
  if (lpr == 16) LN_B(16, 1);
  else if (lpr == 32) LN_B(32, 1);
  else if (iters == 1) LN_B(64, 1);
  else if (iters == 2) LN_B(64, 2);
  else if (iters == 3) LN_B(64, 3);
  else LN_B(64, 4);
#undef LN_B
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
                                 float* dbeta, const void* dres, int64_t rows, int C, int dtype, int grad_dtype,
                                 void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || rows <= 0 || C <= 0) return DCC_ERR_ARG;
  const int vec = dtype == DCC_BF16 ? 8 : 4;
  if ((C % vec) || (ldx % vec) || ((uintptr_t)x & 15)) return DCC_ERR_UNSUPPORTED;
  const int gvec = grad_dtype == DCC_BF16 ? 8 : 4;
  if ((lddy % gvec) || (lddx % gvec) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15) || ((uintptr_t)dres & 15)) return DCC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCC_BF16 && grad_dtype == DCC_BF16)
    return ln_bwd_t<bf16_t, bf16_t>(dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, rows, C, dres, st);
  if (dtype == DCC_BF16 && grad_dtype == DCC_F32)
    return ln_bwd_t<bf16_t, float>(dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, rows, C, dres, st);
  if (dtype == DCC_F32 && grad_dtype == DCC_F32)
    return ln_bwd_t<float, float>(dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, rows, C, dres, st);
  return DCC_ERR_UNSUPPORTED;
}
