// Multi-instance pooling kernels for the token-level consumers of the video encoder:
// models/multi_instance_linear_probing.py:494-538 (masked mean / max, gated attention pooling, hierarchical variants) and the
// study-level patch pooling of utils/loss/multitask_loss.py:270-289.  Everything here is small ([B, N, D] with N = videos
// per study, or [B*N, L, D] patch tokens): one thread per (sample, channel) or one wave per row, fp32 statistics.
#include "common.h"

namespace {

inline unsigned gsm(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// gated attention features: out = tanh(v) * sigmoid(u)
template <typename T>
__global__ void gate_fwd_kernel(const T* __restrict__ v, const T* __restrict__ u, T* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float tv = tanhf(to_f32(v[i])), su = 1.0f / (1.0f + __expf(-to_f32(u[i])));
    out[i] = from_f32<T>(tv * su);
  }
}
template <typename T>
__global__ void gate_bwd_kernel(const T* __restrict__ d, const T* __restrict__ v, const T* __restrict__ u, T* __restrict__ dv,
                                T* __restrict__ du, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float tv = tanhf(to_f32(v[i])), su = 1.0f / (1.0f + __expf(-to_f32(u[i]))), g = to_f32(d[i]);
    dv[i] = from_f32<T>(g * su * (1.0f - tv * tv));
    du[i] = from_f32<T>(g * tv * su * (1.0f - su));
  }
}

// w[b][:] = softmax over the N valid entries of s[b][:] (mask[b][n] != 0 = valid; all-masked rows give zeros); fp32
__global__ void masked_softmax_fwd_kernel(const float* __restrict__ s, const uint8_t* __restrict__ mask, float* __restrict__ w,
                                          int B, int N) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float mx = -INFINITY;
  for (int n = lane; n < N; n += 64)
    if (!mask || mask[(int64_t)b * N + n]) mx = fmaxf(mx, s[(int64_t)b * N + n]);
  mx = wave_max(mx);
  float se = 0.0f;
  for (int n = lane; n < N; n += 64)
    if (!mask || mask[(int64_t)b * N + n]) se += __expf(s[(int64_t)b * N + n] - mx);
  se = wave_sum(se);
  const float inv = se > 0.0f ? 1.0f / se : 0.0f;
  for (int n = lane; n < N; n += 64) {
    const bool ok = !mask || mask[(int64_t)b * N + n];
    w[(int64_t)b * N + n] = ok ? __expf(s[(int64_t)b * N + n] - mx) * inv : 0.0f;
  }
}
// ds = w * (dw - sum_m w_m dw_m)
__global__ void masked_softmax_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dw, float* __restrict__ ds, int B, int N) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float dot = 0.0f;
  for (int n = lane; n < N; n += 64) dot += w[(int64_t)b * N + n] * dw[(int64_t)b * N + n];
  dot = wave_sum(dot);
  for (int n = lane; n < N; n += 64) ds[(int64_t)b * N + n] = w[(int64_t)b * N + n] * (dw[(int64_t)b * N + n] - dot);
}

// out[b][d] = sum_n w[b][n] x[b][n][d]   (thread per (b, d))
template <typename T>
__global__ void wsum_fwd_kernel(const float* __restrict__ w, const T* __restrict__ x, T* __restrict__ out, int B, int N, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * D) return;
  const int b = i / D, d = i % D;
  float acc = 0.0f;
  for (int n = 0; n < N; ++n) acc += w[(int64_t)b * N + n] * to_f32(x[((int64_t)b * N + n) * D + d]);
  out[i] = from_f32<T>(acc);
}
// dx[b][n][d] = w[b][n] dout[b][d];  dw[b][n] = <dout[b], x[b][n]>  (one wave per (b, n))
template <typename T>
__global__ void wsum_bwd_kernel(const float* __restrict__ w, const T* __restrict__ x, const T* __restrict__ dout, T* __restrict__ dx,
                                float* __restrict__ dw, int B, int N, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (int64_t)B * N) return;
  const int b = r / N;
  const float wv = w[r];
  float dot = 0.0f;
  for (int d = lane; d < D; d += 64) {
    const float g = to_f32(dout[(int64_t)b * D + d]);
    dot += g * to_f32(x[r * D + d]);
    dx[r * D + d] = from_f32<T>(wv * g);
  }
  dot = wave_sum(dot);
  if (lane == 0) dw[r] = dot;
}

// masked mean (mode 0: sum over valid / max(count, 1)) and masked max (mode 1, argmax kept) over the instance axis
template <typename T>
__global__ void mreduce_fwd_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask, T* __restrict__ out,
                                   int* __restrict__ arg, int B, int N, int D, int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * D) return;
  const int b = i / D, d = i % D;
  if (mode == 0) {
    float acc = 0.0f, cnt = 0.0f;
    for (int n = 0; n < N; ++n)
      if (!mask || mask[(int64_t)b * N + n]) { acc += to_f32(x[((int64_t)b * N + n) * D + d]); cnt += 1.0f; }
    out[i] = from_f32<T>(acc / fmaxf(cnt, 1.0f));
  } else {
    float best = -INFINITY;
    int bi = -1;
    for (int n = 0; n < N; ++n)
      if (!mask || mask[(int64_t)b * N + n]) {
        const float v = to_f32(x[((int64_t)b * N + n) * D + d]);
        if (v > best) { best = v; bi = n; }
      }
    out[i] = from_f32<T>(best);
    arg[i] = bi;
  }
}
template <typename T>
__global__ void mreduce_bwd_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ mask, const int* __restrict__ arg,
                                   T* __restrict__ dx, int B, int N, int D, int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * N * D) return;
  const int d = i % D, n = (i / D) % N, b = i / ((int64_t)D * N);
  const bool ok = !mask || mask[(int64_t)b * N + n];
  float g = 0.0f;
  if (mode == 0) {
    if (ok) {
      float cnt = 0.0f;
      for (int m = 0; m < N; ++m) cnt += (!mask || mask[(int64_t)b * N + m]) ? 1.0f : 0.0f;
      g = to_f32(dout[(int64_t)b * D + d]) / fmaxf(cnt, 1.0f);
    }
  } else if (arg[(int64_t)b * D + d] == n) {
    g = to_f32(dout[(int64_t)b * D + d]);
  }
  dx[i] = from_f32<T>(g);
}

// view-angle embeddings (models/multi_instance_linear_probing.py:142-145,265-269): out[r][l][:] = x[r][l][:] + table[ids[r]][:]
// for rows r = (study, video), L patch tokens per row (L = 1 for [B, N, D] inputs); fp32
__global__ void view_embed_add_kernel(const float* __restrict__ x, const int64_t* __restrict__ ids, const float* __restrict__ table,
                                      float* __restrict__ out, int64_t rows, int L, int D) {
  const int64_t n = rows * L * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ((int64_t)L * D);
    const int d = (int)(i % D);
    out[i] = x[i] + table[ids[r] * D + d];
  }
}
// dtable[ids[r]][:] += sum_l dy[r][l][:]  (one block per row, fp32 atomics: few rows share a class)
__global__ void view_embed_bwd_kernel(const float* __restrict__ dy, const int64_t* __restrict__ ids, float* __restrict__ dtable,
                                      int L, int D) {
  const int64_t r = blockIdx.x;
  const float* src = dy + r * L * D;
  float* dst = dtable + ids[r] * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float acc = 0.0f;
    for (int l = 0; l < L; ++l) acc += src[(int64_t)l * D + d];
    atomicAdd(dst + d, acc);
  }
}
}  // namespace

#define ST (hipStream_t) stream
extern "C" int dcc_gate_fwd(const void* v, const void* u, void* out, int64_t n, int dtype, void* stream) {
  if (!v || !u || !out || n <= 0) return DCC_ERR_ARG;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((gate_fwd_kernel<bf16_t>), dim3(gsm(n)), dim3(256), 0, ST, (const bf16_t*)v, (const bf16_t*)u, (bf16_t*)out, n);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((gate_fwd_kernel<float>), dim3(gsm(n)), dim3(256), 0, ST, (const float*)v, (const float*)u, (float*)out, n);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}
extern "C" int dcc_gate_bwd(const void* d, const void* v, const void* u, void* dv, void* du, int64_t n, int dtype, void* stream) {
  if (!d || !v || !u || !dv || !du || n <= 0) return DCC_ERR_ARG;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((gate_bwd_kernel<bf16_t>), dim3(gsm(n)), dim3(256), 0, ST, (const bf16_t*)d, (const bf16_t*)v, (const bf16_t*)u, (bf16_t*)dv, (bf16_t*)du, n);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((gate_bwd_kernel<float>), dim3(gsm(n)), dim3(256), 0, ST, (const float*)d, (const float*)v, (const float*)u, (float*)dv, (float*)du, n);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}
extern "C" int dcc_masked_softmax_fwd(const float* s, const uint8_t* mask, float* w, int B, int N, void* stream) {
  if (!s || !w || B <= 0 || N <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(masked_softmax_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, s, mask, w, B, N);
  return dcc_check_launch();
}
extern "C" int dcc_masked_softmax_bwd(const float* w, const float* dw, float* ds, int B, int N, void* stream) {
  if (!w || !dw || !ds || B <= 0 || N <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(masked_softmax_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, w, dw, ds, B, N);
  return dcc_check_launch();
}
extern "C" int dcc_wsum_fwd(const float* w, const void* x, void* out, int B, int N, int D, int dtype, void* stream) {
  if (!w || !x || !out || B <= 0 || N <= 0 || D <= 0) return DCC_ERR_ARG;
  const dim3 g(gsm((int64_t)B * D));
  if ((int64_t)B * D > 8192LL * 256) return DCC_ERR_UNSUPPORTED;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((wsum_fwd_kernel<bf16_t>), g, dim3(256), 0, ST, w, (const bf16_t*)x, (bf16_t*)out, B, N, D);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((wsum_fwd_kernel<float>), g, dim3(256), 0, ST, w, (const float*)x, (float*)out, B, N, D);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}
extern "C" int dcc_wsum_bwd(const float* w, const void* x, const void* dout, void* dx, float* dw, int B, int N, int D, int dtype,
                            void* stream) {
  if (!w || !x || !dout || !dx || !dw || B <= 0 || N <= 0 || D <= 0) return DCC_ERR_ARG;
  const int64_t rows = (int64_t)B * N;
  if (rows > 4LL * 0x7fffffff) return DCC_ERR_UNSUPPORTED;
  const dim3 g((unsigned)((rows + 3) / 4));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((wsum_bwd_kernel<bf16_t>), g, dim3(256), 0, ST, w, (const bf16_t*)x, (const bf16_t*)dout, (bf16_t*)dx, dw, B, N, D);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((wsum_bwd_kernel<float>), g, dim3(256), 0, ST, w, (const float*)x, (const float*)dout, (float*)dx, dw, B, N, D);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}
// mode 0 = masked mean, 1 = masked max (arg: int32 [B, D] workspace / saved argmax, mode 1 only)
extern "C" int dcc_masked_reduce_fwd(const void* x, const uint8_t* mask, void* out, int* arg, int B, int N, int D, int mode, int dtype,
                                     void* stream) {
  if (!x || !out || B <= 0 || N <= 0 || D <= 0 || (mode == 1 && !arg) || (mode != 0 && mode != 1)) return DCC_ERR_ARG;
  const int64_t n = (int64_t)B * D;
  if (n > 8192LL * 256 * 64) return DCC_ERR_UNSUPPORTED;
  const dim3 g((unsigned)((n + 255) / 256));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((mreduce_fwd_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)x, mask, (bf16_t*)out, arg, B, N, D, mode);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((mreduce_fwd_kernel<float>), g, dim3(256), 0, ST, (const float*)x, mask, (float*)out, arg, B, N, D, mode);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}
extern "C" int dcc_masked_reduce_bwd(const void* dout, const uint8_t* mask, const int* arg, void* dx, int B, int N, int D, int mode,
                                     int dtype, void* stream) {
  if (!dout || !dx || B <= 0 || N <= 0 || D <= 0 || (mode == 1 && !arg) || (mode != 0 && mode != 1)) return DCC_ERR_ARG;
  const int64_t n = (int64_t)B * N * D;
  if ((n + 255) / 256 > 0x7fffffffLL) return DCC_ERR_UNSUPPORTED;
  const dim3 g((unsigned)((n + 255) / 256));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((mreduce_bwd_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)dout, mask, arg, (bf16_t*)dx, B, N, D, mode);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((mreduce_bwd_kernel<float>), g, dim3(256), 0, ST, (const float*)dout, mask, arg, (float*)dx, B, N, D, mode);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

// view-angle embedding add and its table gradient (ids: int64 [rows], validated on the host: 0 <= id < classes)
extern "C" int dcc_view_embed_add(const float* x, const int64_t* ids, const float* table, float* out, int64_t rows, int L, int D,
                                  void* stream) {
  if (!x || !ids || !table || !out || rows <= 0 || L <= 0 || D <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(view_embed_add_kernel, dim3(gsm(rows * L * D)), dim3(256), 0, ST, x, ids, table, out, rows, L, D);
  return dcc_check_launch();
}
extern "C" int dcc_view_embed_bwd(const float* dy, const int64_t* ids, float* dtable, int64_t rows, int L, int D, void* stream) {
  if (!dy || !ids || !dtable || rows <= 0 || L <= 0 || D <= 0) return DCC_ERR_ARG;
  if (rows > 0x7fffffffLL) return DCC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(view_embed_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, ST, dy, ids, dtable, L, D);
  return dcc_check_launch();
}
