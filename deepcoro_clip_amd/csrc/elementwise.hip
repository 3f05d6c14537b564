// Streaming (HBM-bound) helpers: casts, axpby, column sums (bias gradients), class-token rows,
// token mean pooling, BERT embedding gather/scatter, 3-D RoPE rotation, learned-query softmax
// pooling of the video aggregator, nan_to_num, dropout.
#include "common.h"
#include <type_traits>

namespace {

inline unsigned gs_blocks(int64_t n, int per_thread = 1) {
  int64_t b = (n + 256LL * per_thread - 1) / (256LL * per_thread);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// 8 elements per thread per trip (16-byte accesses on the 2-byte side, 2 x 16 on the 4-byte side), scalar tail
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
  const int64_t n8 = ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) ? n / 8 : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    if constexpr (sizeof(TS) == 2) {
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(s + 8 * i);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>(s + 8 * i), b = *reinterpret_cast<const f32x4*>(s + 8 * i + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    if constexpr (sizeof(TD) == 2) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      *reinterpret_cast<bf16x8*>(d + 8 * i) = o;
    } else {
      f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
      *reinterpret_cast<f32x4*>(d + 8 * i) = a;
      *reinterpret_cast<f32x4*>(d + 8 * i + 4) = b;
    }
  }
  for (int64_t i = 8 * n8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = from_f32<TD>(to_f32(s[i]));
}

// fp32 <-> IEEE fp16: the wire format of SiglipLossDDP's embedding all-gather (utils/loss/losses.py:243-248 rounds the
// embeddings to half before the gather, and autograd rounds the returning gradient to half on the way back)
__global__ void cast_f32_f16_kernel(const float* __restrict__ s, _Float16* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (_Float16)s[i];
}
__global__ void cast_f16_f32_kernel(const _Float16* __restrict__ s, float* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (float)s[i];
}

// strided 2-D cast/copy: d[r*ldd + c] = s[r*lds + c]
template <typename TS, typename TD>
__global__ void cast2d_kernel(const TS* __restrict__ s, int64_t lds_, TD* __restrict__ d, int64_t ldd, int64_t rows,
                              int cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = i % cols;
    d[r * ldd + c] = from_f32<TD>(to_f32(s[r * lds_ + c]));
  }
}

template <typename T>
__global__ void axpby_kernel(T* __restrict__ out, float a, const T* __restrict__ x, float b, const T* __restrict__ y,
                             int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = a * to_f32(x[i]);
    if (y) v += b * to_f32(y[i]);
    out[i] = from_f32<T>(v);
  }
}

// out[r][c] = resid[r][c] + x[r][c] * scale[r / rows_per_sample]   (stochastic depth "row" mode: one Bernoulli
// keep/(1-p) factor per clip; resid may be null -> the backward's d_branch = dout * scale)
template <typename T>
__global__ void scale_rows_kernel(const T* __restrict__ x, const float* __restrict__ scale, const T* __restrict__ resid,
                                  T* __restrict__ out, int64_t n8, int64_t per_sample8) {
  constexpr int VEC = Vec16<T>::N;
  using V = vec16_t<T>;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const float sc = scale[i / per_sample8];
    const V xv = *reinterpret_cast<const V*>(x + i * VEC);
    V o;
    if (resid) {
      const V rv = *reinterpret_cast<const V*>(resid + i * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(to_f32(rv[e]) + to_f32(xv[e]) * sc);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = from_f32<T>(to_f32(xv[e]) * sc);
    }
    *reinterpret_cast<V*>(out + i * VEC) = o;
  }
}

// uint8 frames -> normalised fp32: (x - mean[c]) / std[c] with the channel innermost (torchvision v2.Normalize as
// used by utils/video.py:278-288, applied on the GPU so that the host ships 1 byte per sample instead of 4)
__global__ void u8_normalize_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t n4, int C,
                                    const float* __restrict__ mean, const float* __restrict__ stdv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t v = reinterpret_cast<const uint32_t*>(src)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = (int)((i * 4 + e) % C);
      o[e] = __fdiv_rn(__fsub_rn((float)((v >> (8 * e)) & 0xffu), mean[c]), stdv[c]);
    }
    reinterpret_cast<f32x4*>(dst)[i] = o;
  }
}

// out = dy * gelu'(pre)
template <typename T>
__global__ void dgelu_mul_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = from_f32<T>(to_f32(dy[i]) * dgelu_for<T>(to_f32(pre[i])));
}

// out[n] += sum_m x[m*ldx + n].  VEC: rows are 4-element aligned (N % 4 == 0, ldx % 4 == 0, aligned base): one 8-byte (bf16)
// / 16-byte (fp32) load per 4 columns, two row groups in flight per thread.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ out,
                                                     int64_t M, int N) {
  extern __shared__ float red[];
  for (int i = threadIdx.x; i < N; i += blockDim.x) red[i] = 0.0f;
  __syncthreads();
  const int cpr = (N + 3) / 4;
  const int lanes = cpr < 256 ? cpr : 256;
  const int rpb = 256 / lanes;
  const int r_in = threadIdx.x / lanes, c0 = threadIdx.x % lanes;
  if (r_in < rpb) {
    for (int cc = c0; cc < cpr; cc += lanes) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      const int64_t step = (int64_t)gridDim.x * rpb;
      if constexpr (VEC) {
        using V4 = typename std::conditional<sizeof(T) == 2, bf16x4, f32x4>::type;
        int64_t r = (int64_t)blockIdx.x * rpb + r_in;
        for (; r + step < M; r += 2 * step) {
          const V4 a = *reinterpret_cast<const V4*>(x + r * ldx + cc * 4);
          const V4 b = *reinterpret_cast<const V4*>(x + (r + step) * ldx + cc * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += (float)a[e] + (float)b[e];
        }
        if (r < M) {
          const V4 a = *reinterpret_cast<const V4*>(x + r * ldx + cc * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += (float)a[e];
        }
      } else {
        for (int64_t r = (int64_t)blockIdx.x * rpb + r_in; r < M; r += step) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (cc * 4 + e < N) acc[e] += to_f32(x[r * ldx + cc * 4 + e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (cc * 4 + e < N) atomicAdd(&red[cc * 4 + e], acc[e]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += blockDim.x) atomicAdd(out + i, red[i]);
}

template <typename T>
__global__ void broadcast_row_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t ld, int64_t rows,
                                     int C) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[(i / C) * ld + (i % C)] = from_f32<T>(src[i % C]);
}

// y[b][d] = mean_n x[b][n][d]; block = (b, 64 columns), the 4 waves take interleaved token rows with 8 loads in
// flight each and fold through LDS (one thread per output walked the N rows in a serial latency chain)
template <typename T>
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int N, int D) {
  __shared__ float part[4][64];
  const int nd = (D + 63) / 64;
  const int b = blockIdx.x / nd, d = (blockIdx.x - b * nd) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  float acc = 0.0f;
  if (d < D) {
    const T* xb = x + (int64_t)b * N * D + d;
    int n = g;
    for (; n + 28 < N; n += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = to_f32(xb[(int64_t)(n + 4 * u) * D]);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; n < N; n += 4) acc += to_f32(xb[(int64_t)n * D]);
  }
  part[g][threadIdx.x & 63] = acc;
  __syncthreads();
  if (g == 0 && d < D) y[(int64_t)b * D + d] = from_f32<T>((part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / N);
}
template <typename T>
__global__ void token_mean_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int N, int D) {
  const int64_t total = (int64_t)B * N * D;
  const float inv = 1.0f / N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = i % D;
    const int64_t b = i / ((int64_t)N * D);
    dx[i] = from_f32<T>(to_f32(dy[b * D + d]) * inv);
  }
}

// BERT embeddings: out[b*S+s] = word[ids] + pos[s] + type[0]   (fp32 tables -> T)
template <typename T>
__global__ void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                 const float* __restrict__ pos, const float* __restrict__ type, T* __restrict__ out,
                                 int64_t rows, int S, int D, int vocab) {
  const int64_t total = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int d = i % D;
    int64_t id = ids[r];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    out[i] = from_f32<T>(word[id * D + d] + pos[(r % S) * D + d] + type[d]);
  }
}
template <typename T>
__global__ void embed_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dy, float* __restrict__ dword,
                                 float* __restrict__ dpos, float* __restrict__ dtype_, int64_t rows, int S, int D,
                                 int vocab) {
  const int64_t total = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int d = i % D;
    int64_t id = ids[r];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const float g = to_f32(dy[i]);
    if (dword) atomicAdd(dword + id * D + d, g);
    if (dpos) atomicAdd(dpos + (r % S) * D + d, g);
    if (dtype_) atomicAdd(dtype_ + d, g);
  }
}

// interleaved-pair rotation: y[2i] = x[2i] c[2i] - sgn x[2i+1] s[2i];  y[2i+1] = x[2i+1] c[2i+1] + sgn x[2i] s[2i+1]
// x: [B, N, H, D] view (strides), tables [N, D] fp32. sgn = +1 forward, -1 backward (transpose of the rotation).
template <typename T>
__global__ void rope_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ cs,
                            const float* __restrict__ sn, int B, int N, int H, int D, int64_t sb, int64_t sn_, int64_t sh,
                            int64_t yb, int64_t yn, int64_t yh, float sgn) {
  const int half = D / 2;
  const int64_t total = (int64_t)B * N * H * half;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = i % half;
    const int hd = (i / half) % H;
    const int n = (i / ((int64_t)half * H)) % N;
    const int b = i / ((int64_t)half * H * N);
    const T* xp = x + b * sb + (int64_t)n * sn_ + hd * sh + 2 * p;
    T* yp = y + b * yb + (int64_t)n * yn + hd * yh + 2 * p;
    const float x0 = to_f32(xp[0]), x1 = to_f32(xp[1]);
    const float c0 = cs[(int64_t)n * D + 2 * p], c1 = cs[(int64_t)n * D + 2 * p + 1];
    const float s0 = sn[(int64_t)n * D + 2 * p], s1 = sn[(int64_t)n * D + 2 * p + 1];
    float y0, y1;
    if (sgn > 0) { y0 = x0 * c0 - x1 * s0; y1 = x1 * c1 + x0 * s1; }
    else { y0 = x0 * c0 + x1 * s1; y1 = x1 * c1 - x0 * s0; }
    yp[0] = from_f32<T>(y0);
    yp[1] = from_f32<T>(y1);
  }
}

// learned-query softmax pooling over N tokens (EnhancedVideoAggregator tail): one block per batch row.
// scores[n] = <q, x[n]>, w = softmax(scores), out = sum_n w[n] x[n]
__global__ __launch_bounds__(256) void qpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ q,
                                                        float* __restrict__ out, float* __restrict__ wout, int N,
                                                        int D) {
  extern __shared__ float sm[];  // scores[N]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (int64_t)b * N * D;
  for (int n = wave; n < N; n += 4) {
    float acc = 0.0f;
    for (int d = lane; d < D; d += 64) acc += q[d] * xb[(int64_t)n * D + d];
    acc = wave_sum(acc);
    if (lane == 0) sm[n] = acc;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int n = 0; n < N; ++n) mx = fmaxf(mx, sm[n]);
  float den = 0.0f;
  for (int n = 0; n < N; ++n) den += __expf(sm[n] - mx);
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    const float w = __expf(sm[n] - mx) / den;
    sm[n] = w;
    wout[(int64_t)b * N + n] = w;
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    float acc = 0.0f;
    for (int n = 0; n < N; ++n) acc += sm[n] * xb[(int64_t)n * D + d];
    out[(int64_t)b * D + d] = acc;
  }
}
// dx[n] = w[n] dout + ds[n] q;  ds = w (dw - sum w dw), dw[n] = <dout, x[n]>;  dq += sum_n ds[n] x[n]
__global__ __launch_bounds__(256) void qpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ q,
                                                        const float* __restrict__ w, const float* __restrict__ dout,
                                                        float* __restrict__ dx, float* __restrict__ dq, int N, int D) {
  extern __shared__ float sm[];  // ds[N]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (int64_t)b * N * D;
  const float* wb = w + (int64_t)b * N;
  const float* db = dout + (int64_t)b * D;
  for (int n = wave; n < N; n += 4) {
    float acc = 0.0f;
    for (int d = lane; d < D; d += 64) acc += db[d] * xb[(int64_t)n * D + d];
    acc = wave_sum(acc);
    if (lane == 0) sm[n] = acc;
  }
  __syncthreads();
  float dot = 0.0f;
  for (int n = 0; n < N; ++n) dot += wb[n] * sm[n];
  __syncthreads();
  for (int n = tid; n < N; n += 256) sm[n] = wb[n] * (sm[n] - dot);
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    float dqa = 0.0f;
    for (int n = 0; n < N; ++n) {
      dx[((int64_t)b * N + n) * D + d] = wb[n] * db[d] + sm[n] * q[d];
      dqa += sm[n] * xb[(int64_t)n * D + d];
    }
    atomicAdd(dq + d, dqa);
  }
}

template <typename T>
__global__ void nan_to_num_kernel(T* __restrict__ x, int64_t n, float nanv, float posv, float negv,
                                  int* __restrict__ count) {
  int local = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = to_f32(x[i]);
    if (v != v) { x[i] = from_f32<T>(nanv); ++local; }
    else if (v == INFINITY) { x[i] = from_f32<T>(posv); ++local; }
    else if (v == -INFINITY) { x[i] = from_f32<T>(negv); ++local; }
  }
  if (count && local) atomicAdd(count, local);
}

// y = x * keep / (1-p); keep regenerated from (seed, index) so no mask is stored.  `step` (nullable, device): a counter
// mixed into the seed, so that a launch replayed from a HIP graph draws a new mask whenever the counter was advanced.
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p, uint64_t seed,
                               const int64_t* __restrict__ step) {
  if (step) seed += (uint64_t)(*step) * 0x9E3779B97F4A7C15ULL;
  const float scale = 1.0f / (1.0f - p);
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool keep = hash32(seed * 0x100000001b3ULL + (uint64_t)i) >= thr;
    y[i] = keep ? from_f32<T>(to_f32(x[i]) * scale) : from_f32<T>(0.0f);
  }
}

// x[i] *= *coef  (device-side coefficient, e.g. gradient clipping)
__global__ void scale_by_dev_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ coef) {
  const float c = *coef;
  if (c == 1.0f) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= c;
}

// ReLU (nn.TransformerEncoderLayer's default activation inside AttentionPoolWithCLS): y = max(x, 0); dx = dy * [y > 0]
template <typename T>
__global__ void relu_kernel(const T* __restrict__ x, const T* __restrict__ gate, T* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = to_f32(gate ? gate[i] : x[i]);
    y[i] = g > 0.0f ? x[i] : from_f32<T>(0.0f);
  }
}

// ---- fp8 quantisation (per-tensor scale) for the fp8 GEMMs of BASELINE configs[4] ------------------------------------
// One atomicMax per BLOCK on the tensor's amax word (waves fold through LDS first), from a grid of at most 1024 blocks: atomics
// on one address retire serially at ~12 ns each, and the first versions' one atomic per WAVE (19-33 k of them) was the whole
// cost of quantising an activation — 230 of 260 us for a [50 208, 384] tensor (tools/gemm_bench.py, r03).
__device__ __forceinline__ void block_amax(float m, float* __restrict__ amax) {
  __shared__ float part[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (m > 0.0f) atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(m));
  }
}
// amax[0] = max(amax[0], max |x|): float bit patterns of non-negative values order like unsigned integers
template <typename T>
__global__ void absmax_kernel(const T* __restrict__ x, int64_t rows, int64_t cols, int64_t ldx, float* __restrict__ amax) {
  float m = 0.0f;
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(to_f32(x[(i / cols) * ldx + (i % cols)])));
  block_amax(m, amax);
}
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d, float lim, int bf8) {
  a = fminf(fmaxf(a, -lim), lim); b = fminf(fmaxf(b, -lim), lim); c = fminf(fmaxf(c, -lim), lim); d = fminf(fmaxf(d, -lim), lim);
  int v = 0;
  if (bf8) { v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, v, false); v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true); }
  else { v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false); v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true); }
  return (unsigned)v;
}
// q = fp8(x * FMAX / amax), scale_out[0] = amax / FMAX (the dequantisation multiplier); cols % 4 == 0
template <typename T>
__global__ void fp8_quant_kernel(const T* __restrict__ x, int64_t rows, int64_t cols, int64_t ldx, const float* __restrict__ amax,
                                 unsigned* __restrict__ out, int64_t ldo4, float fmax_, int bf8, float* __restrict__ scale_out) {
  const float am = *amax;
  const float mul = am > 0.0f ? fmax_ / am : 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = am > 0.0f ? am / fmax_ : 1.0f;
  const int64_t c4 = cols / 4, n4 = rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4, c = (i % c4) * 4;
    const T* p = x + r * ldx + c;
    out[r * ldo4 + c / 4] = pack4_fp8(to_f32(p[0]) * mul, to_f32(p[1]) * mul, to_f32(p[2]) * mul, to_f32(p[3]) * mul, fmax_, bf8);
  }
}
// Contiguous bf16 activations (the case every layer hits: ldx == cols, 16-byte aligned, n % 8 == 0): 16-byte loads, no index
// arithmetic — the generic kernels above spend a 64-bit division per ELEMENT and ran at 0.3 TB/s (tools/gemm_bench.py, r03:
// 375 us to quantise a [50 208, 384] activation, three times its GEMM); these stream at HBM rate.
__global__ __launch_bounds__(256) void absmax_bf16_flat_kernel(const uint4* __restrict__ x, int64_t n8, float* __restrict__ amax) {
  unsigned m = 0;   // max over |x| as bf16 bit patterns (sign cleared): orders like the values
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const uint4 v = x[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned lo = w[e] & 0x7fffu, hi = (w[e] >> 16) & 0x7fffu;
      m = max(m, lo > 0x7f80u ? 0u : lo);   // NaN patterns are skipped, like fmaxf does in the generic kernel
      m = max(m, hi > 0x7f80u ? 0u : hi);
    }
  }
  block_amax(__uint_as_float(m << 16), amax);
}
__global__ __launch_bounds__(256) void fp8_quant_bf16_flat_kernel(const uint4* __restrict__ x, int64_t n8, const float* __restrict__ amax,
                                                                  uint2* __restrict__ out, float fmax_, int bf8,
                                                                  float* __restrict__ scale_out) {
  const float am = *amax;
  const float mul = am > 0.0f ? fmax_ / am : 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = am > 0.0f ? am / fmax_ : 1.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const uint4 v = x[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = __uint_as_float(w[e] << 16) * mul;
      f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u) * mul;
    }
    uint2 o;
    o.x = pack4_fp8(f[0], f[1], f[2], f[3], fmax_, bf8);
    o.y = pack4_fp8(f[4], f[5], f[6], f[7], fmax_, bf8);
    out[i] = o;
  }
}
// transposed output: out8[c][r] = fp8(x[r][c] * mul): 32 x 32 tiles through LDS (weights only: small)
template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_t_kernel(const T* __restrict__ x, int rows, int cols, int64_t ldx,
                                                          const float* __restrict__ amax, uint8_t* __restrict__ out, int64_t ldo,
                                                          float fmax_, int bf8, float* __restrict__ scale_out) {
  __shared__ float tile[32][33];
  const float am = *amax;
  const float mul = am > 0.0f ? fmax_ / am : 1.0f;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *scale_out = am > 0.0f ? am / fmax_ : 1.0f;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int r = r0 + i / 32, c = c0 + i % 32;
    tile[i / 32][i % 32] = (r < rows && c < cols) ? to_f32(x[(int64_t)r * ldx + c]) * mul : 0.0f;
  }
  __syncthreads();
  // thread = (output row c, 4 consecutive output columns r): one packed store
  const int oc = threadIdx.x / 8, og = threadIdx.x % 8;
  const int c = c0 + oc, r = r0 + 4 * og;
  if (c < cols && r < rows) {  // rows % 4 == 0 is required by the caller
    const unsigned v = pack4_fp8(tile[4 * og][oc], tile[4 * og + 1][oc], tile[4 * og + 2][oc], tile[4 * og + 3][oc], fmax_, bf8);
    *reinterpret_cast<unsigned*>(out + (int64_t)c * ldo + r) = v;
  }
}

// dst[c][r] = (bf16) src[r][c]: the transposed bf16 shadow of an fp32 weight (the streaming GEMM takes W as [N][K] rows, a
// data-gradient product g W needs W^T in that form); 32 x 32 tiles through LDS, weights only (<= a few MB per step)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, int64_t lds_, bf16_t* __restrict__ dst,
                                                             int64_t ldd, int rows, int cols) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int r = r0 + i / 32, c = c0 + i % 32;
    tile[i / 32][i % 32] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int c = c0 + i / 32, r = r0 + i % 32;
    if (c < cols && r < rows) dst[(int64_t)c * ldd + r] = (bf16_t)tile[i % 32][i / 32];
  }
}

}  // namespace

#define ST (hipStream_t) stream

extern "C" int dcc_cast_transpose(const float* src, int64_t lds_, void* dst, int64_t ldd, int rows, int cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, ST, src, lds_, (bf16_t*)dst, ldd, rows, cols);
  return dcc_check_launch();
}

// gate == NULL: y = relu(x);  gate != NULL: y = x * [gate > 0]  (backward with gate = the forward output)
extern "C" int dcc_relu(const void* x, const void* gate, void* y, int64_t n, int dtype, void* stream) {
  if (!x || !y || n <= 0) return DCC_ERR_ARG;
  dim3 g(gs_blocks(n, 4));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((relu_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)gate, (bf16_t*)y, n);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((relu_kernel<float>), g, dim3(256), 0, ST, (const float*)x, (const float*)gate, (float*)y, n);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}
extern "C" int dcc_absmax(const void* x, int64_t rows, int64_t cols, int64_t ldx, int dtype, float* amax, void* stream) {
  if (!x || !amax || rows <= 0 || cols <= 0) return DCC_ERR_ARG;
  if (dtype == DCC_BF16 && ldx == cols && ((rows * cols) % 8) == 0 && !((uintptr_t)x & 15)) {
    const int64_t n8 = rows * cols / 8;
    unsigned gb = gs_blocks(n8, 2);
    if (gb > 1024) gb = 1024;
    hipLaunchKernelGGL(absmax_bf16_flat_kernel, dim3(gb), dim3(256), 0, ST, (const uint4*)x, n8, amax);
    return dcc_check_launch();
  }
  dim3 g(gs_blocks(rows * cols, 8));
  if (g.x > 1024) g.x = 1024;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((absmax_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)x, rows, cols, ldx, amax);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((absmax_kernel<float>), g, dim3(256), 0, ST, (const float*)x, rows, cols, ldx, amax);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}
// fmt: 0 = e4m3 (max 448), 1 = e5m2 (max 57344).  transpose != 0 writes out8[c][r] (ldo = row pitch of the output in bytes).
extern "C" int dcc_fp8_quantize(const void* x, int64_t rows, int64_t cols, int64_t ldx, int dtype, const float* amax, void* out8,
                                int64_t ldo, int transpose, int fmt, float* scale_out, void* stream) {
  if (!x || !amax || !out8 || !scale_out || rows <= 0 || cols <= 0 || (fmt != 0 && fmt != 1)) return DCC_ERR_ARG;
  const float fmax_ = fmt == 0 ? 448.0f : 57344.0f;
  if (transpose) {
    if ((rows % 4) || (ldo % 4) || ((uintptr_t)out8 & 3) || rows > 0x7fffffff || cols > 0x7fffffff) return DCC_ERR_UNSUPPORTED;
    dim3 g((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    if (dtype == DCC_BF16) hipLaunchKernelGGL((fp8_quant_t_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)x, (int)rows, (int)cols, ldx, amax, (uint8_t*)out8, ldo, fmax_, fmt, scale_out);
    else if (dtype == DCC_F32) hipLaunchKernelGGL((fp8_quant_t_kernel<float>), g, dim3(256), 0, ST, (const float*)x, (int)rows, (int)cols, ldx, amax, (uint8_t*)out8, ldo, fmax_, fmt, scale_out);
    else return DCC_ERR_UNSUPPORTED;
    return dcc_check_launch();
  }
  if ((cols % 4) || (ldo % 4) || ((uintptr_t)out8 & 3)) return DCC_ERR_UNSUPPORTED;
  if (dtype == DCC_BF16 && ldx == cols && ldo == cols && ((rows * cols) % 8) == 0 && !((uintptr_t)x & 15) && !((uintptr_t)out8 & 7)) {
    const int64_t n8 = rows * cols / 8;
    hipLaunchKernelGGL(fp8_quant_bf16_flat_kernel, dim3(gs_blocks(n8, 2)), dim3(256), 0, ST, (const uint4*)x, n8, amax, (uint2*)out8, fmax_,
                       fmt, scale_out);
    return dcc_check_launch();
  }
  dim3 g(gs_blocks(rows * cols / 4, 4));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((fp8_quant_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)x, rows, cols, ldx, amax, (unsigned*)out8, ldo / 4, fmax_, fmt, scale_out);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((fp8_quant_kernel<float>), g, dim3(256), 0, ST, (const float*)x, rows, cols, ldx, amax, (unsigned*)out8, ldo / 4, fmax_, fmt, scale_out);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream) {
  if (!src || !dst || n <= 0) return DCC_ERR_ARG;
  dim3 g(gs_blocks(n, 8));
  if (src_dtype == DCC_F32 && dst_dtype == DCC_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, dim3(256), 0, ST, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == DCC_BF16 && dst_dtype == DCC_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, dim3(256), 0, ST, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == DCC_F32 && dst_dtype == DCC_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, dim3(256), 0, ST, (const float*)src, (float*)dst, n);
  else if (src_dtype == DCC_BF16 && dst_dtype == DCC_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)src, (bf16_t*)dst, n);
  else if (src_dtype == DCC_F32 && dst_dtype == DCC_F16) hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(gs_blocks(n, 1)), dim3(256), 0, ST, (const float*)src, (_Float16*)dst, n);
  else if (src_dtype == DCC_F16 && dst_dtype == DCC_F32) hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(gs_blocks(n, 1)), dim3(256), 0, ST, (const _Float16*)src, (float*)dst, n);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_cast2d(const void* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows, int cols, int src_dtype,
                          int dst_dtype, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return DCC_ERR_ARG;
  dim3 g(gs_blocks(rows * cols, 4));
  if (src_dtype == DCC_F32 && dst_dtype == DCC_BF16) hipLaunchKernelGGL((cast2d_kernel<float, bf16_t>), g, dim3(256), 0, ST, (const float*)src, lds_, (bf16_t*)dst, ldd, rows, cols);
  else if (src_dtype == DCC_BF16 && dst_dtype == DCC_F32) hipLaunchKernelGGL((cast2d_kernel<bf16_t, float>), g, dim3(256), 0, ST, (const bf16_t*)src, lds_, (float*)dst, ldd, rows, cols);
  else if (src_dtype == DCC_F32 && dst_dtype == DCC_F32) hipLaunchKernelGGL((cast2d_kernel<float, float>), g, dim3(256), 0, ST, (const float*)src, lds_, (float*)dst, ldd, rows, cols);
  else if (src_dtype == DCC_BF16 && dst_dtype == DCC_BF16) hipLaunchKernelGGL((cast2d_kernel<bf16_t, bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)src, lds_, (bf16_t*)dst, ldd, rows, cols);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_axpby(void* out, float a, const void* x, float b, const void* y, int64_t n, int dtype, void* stream) {
  if (!out || !x || n <= 0) return DCC_ERR_ARG;
  dim3 g(gs_blocks(n, 4));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((axpby_kernel<bf16_t>), g, dim3(256), 0, ST, (bf16_t*)out, a, (const bf16_t*)x, b, (const bf16_t*)y, n);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((axpby_kernel<float>), g, dim3(256), 0, ST, (float*)out, a, (const float*)x, b, (const float*)y, n);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_u8_normalize(const uint8_t* src, float* dst, int64_t n, int C, const float* mean, const float* stdv, void* stream) {
  if (!src || !dst || !mean || !stdv || n <= 0 || C <= 0 || (n % 4) || ((uintptr_t)src & 3) || ((uintptr_t)dst & 15)) return DCC_ERR_ARG;
  dim3 g(gs_blocks(n / 4, 2));
  hipLaunchKernelGGL(u8_normalize_kernel, g, dim3(256), 0, ST, src, dst, n / 4, C, mean, stdv);
  return dcc_check_launch();
}

extern "C" int dcc_scale_rows(const void* x, const float* scale, const void* resid, void* out, int64_t samples,
                              int64_t per_sample, int dtype, void* stream) {
  if (!x || !scale || !out || samples <= 0 || per_sample <= 0) return DCC_ERR_ARG;
  const int vec = dtype == DCC_BF16 ? 8 : 4;
  if ((per_sample % vec) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)resid & 15)) return DCC_ERR_UNSUPPORTED;
  const int64_t n8 = samples * per_sample / vec;
  dim3 g(gs_blocks(n8, 2));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((scale_rows_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)x, scale, (const bf16_t*)resid, (bf16_t*)out, n8, per_sample / vec);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((scale_rows_kernel<float>), g, dim3(256), 0, ST, (const float*)x, scale, (const float*)resid, (float*)out, n8, per_sample / vec);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_dgelu_mul(const void* dy, const void* pre, void* out, int64_t n, int dtype, void* stream) {
  if (!dy || !pre || !out || n <= 0) return DCC_ERR_ARG;
  dim3 g(gs_blocks(n, 4));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((dgelu_mul_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)dy, (const bf16_t*)pre, (bf16_t*)out, n);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((dgelu_mul_kernel<float>), g, dim3(256), 0, ST, (const float*)dy, (const float*)pre, (float*)out, n);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_colsum(const void* x, int64_t ldx, float* out, int64_t M, int N, int dtype, void* stream) {
  if (!x || !out || M <= 0 || N <= 0 || N > 8192) return DCC_ERR_ARG;
  int64_t blocks = (M + 255) / 256;
  if (blocks > 1024) blocks = 1024;  // each block ends with N global atomics
  dim3 g((unsigned)blocks);
  const int es = dtype == DCC_BF16 ? 2 : 4;
  const bool vec = (N % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x) % (4 * es) == 0);
  if (dtype == DCC_BF16) {
    if (vec) hipLaunchKernelGGL((colsum_kernel<bf16_t, true>), g, dim3(256), N * 4, ST, (const bf16_t*)x, ldx, out, M, N);
    else hipLaunchKernelGGL((colsum_kernel<bf16_t, false>), g, dim3(256), N * 4, ST, (const bf16_t*)x, ldx, out, M, N);
  } else if (dtype == DCC_F32) {
    if (vec) hipLaunchKernelGGL((colsum_kernel<float, true>), g, dim3(256), N * 4, ST, (const float*)x, ldx, out, M, N);
    else hipLaunchKernelGGL((colsum_kernel<float, false>), g, dim3(256), N * 4, ST, (const float*)x, ldx, out, M, N);
  } else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_broadcast_row(const float* src, void* dst, int64_t ld, int64_t rows, int C, int dtype, void* stream) {
  if (!src || !dst || rows <= 0 || C <= 0) return DCC_ERR_ARG;
  dim3 g(gs_blocks(rows * C));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((broadcast_row_kernel<bf16_t>), g, dim3(256), 0, ST, src, (bf16_t*)dst, ld, rows, C);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((broadcast_row_kernel<float>), g, dim3(256), 0, ST, src, (float*)dst, ld, rows, C);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_token_mean(const void* in, void* out, int B, int N, int D, int backward, int dtype, void* stream) {
  if (!in || !out || B <= 0 || N <= 0 || D <= 0) return DCC_ERR_ARG;
  const int64_t total = backward ? (int64_t)B * N * D : (int64_t)B * D;
  dim3 g(gs_blocks(total));
  if (dtype == DCC_BF16) {
    if (backward) hipLaunchKernelGGL((token_mean_bwd_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)in, (bf16_t*)out, B, N, D);
    else hipLaunchKernelGGL((token_mean_fwd_kernel<bf16_t>), dim3(B * ((D + 63) / 64)), dim3(256), 0, ST, (const bf16_t*)in, (bf16_t*)out, B, N, D);
  } else if (dtype == DCC_F32) {
    if (backward) hipLaunchKernelGGL((token_mean_bwd_kernel<float>), g, dim3(256), 0, ST, (const float*)in, (float*)out, B, N, D);
    else hipLaunchKernelGGL((token_mean_fwd_kernel<float>), dim3(B * ((D + 63) / 64)), dim3(256), 0, ST, (const float*)in, (float*)out, B, N, D);
  } else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type, void* out,
                             int64_t rows, int S, int D, int vocab, int dtype, void* stream) {
  if (!ids || !word || !pos || !type || !out || rows <= 0) return DCC_ERR_ARG;
  dim3 g(gs_blocks(rows * D));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((embed_fwd_kernel<bf16_t>), g, dim3(256), 0, ST, ids, word, pos, type, (bf16_t*)out, rows, S, D, vocab);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((embed_fwd_kernel<float>), g, dim3(256), 0, ST, ids, word, pos, type, (float*)out, rows, S, D, vocab);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_embed_bwd(const int64_t* ids, const void* dy, float* dword, float* dpos, float* dtype_, int64_t rows,
                             int S, int D, int vocab, int dtype, void* stream) {
  if (!ids || !dy || rows <= 0) return DCC_ERR_ARG;
  dim3 g(gs_blocks(rows * D));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((embed_bwd_kernel<bf16_t>), g, dim3(256), 0, ST, ids, (const bf16_t*)dy, dword, dpos, dtype_, rows, S, D, vocab);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((embed_bwd_kernel<float>), g, dim3(256), 0, ST, ids, (const float*)dy, dword, dpos, dtype_, rows, S, D, vocab);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

// strides: [x_b, x_n, x_h, y_b, y_n, y_h]
extern "C" int dcc_rope(const void* x, void* y, const float* cos_t, const float* sin_t, int B, int N, int H, int D,
                        const int64_t* strides, int backward, int dtype, void* stream) {
  if (!x || !y || !cos_t || !sin_t || !strides || (D & 1)) return DCC_ERR_ARG;
  dim3 g(gs_blocks((int64_t)B * N * H * (D / 2)));
  const float sgn = backward ? -1.0f : 1.0f;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((rope_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, cos_t, sin_t, B, N, H, D, strides[0], strides[1], strides[2], strides[3], strides[4], strides[5], sgn);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((rope_kernel<float>), g, dim3(256), 0, ST, (const float*)x, (float*)y, cos_t, sin_t, B, N, H, D, strides[0], strides[1], strides[2], strides[3], strides[4], strides[5], sgn);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_qpool_fwd(const float* x, const float* q, float* out, float* w, int B, int N, int D, void* stream) {
  if (!x || !q || !out || !w || B <= 0 || N <= 0 || N > 8192) return DCC_ERR_ARG;
  hipLaunchKernelGGL(qpool_fwd_kernel, dim3(B), dim3(256), N * 4, ST, x, q, out, w, N, D);
  return dcc_check_launch();
}
extern "C" int dcc_qpool_bwd(const float* x, const float* q, const float* w, const float* dout, float* dx, float* dq,
                             int B, int N, int D, void* stream) {
  if (!x || !q || !w || !dout || !dx || !dq || B <= 0 || N <= 0 || N > 8192) return DCC_ERR_ARG;
  hipLaunchKernelGGL(qpool_bwd_kernel, dim3(B), dim3(256), N * 4, ST, x, q, w, dout, dx, dq, N, D);
  return dcc_check_launch();
}

extern "C" int dcc_nan_to_num(void* x, int64_t n, float nanv, float posv, float negv, int* count, int dtype, void* stream) {
  if (!x || n <= 0) return DCC_ERR_ARG;
  dim3 g(gs_blocks(n, 4));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((nan_to_num_kernel<bf16_t>), g, dim3(256), 0, ST, (bf16_t*)x, n, nanv, posv, negv, count);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((nan_to_num_kernel<float>), g, dim3(256), 0, ST, (float*)x, n, nanv, posv, negv, count);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

// The same (seed) regenerates the same keep-mask: call with x = dy for the backward.
extern "C" int dcc_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, const int64_t* step, int dtype,
                           void* stream) {
  if (!x || !y || n <= 0 || p < 0.0f || p >= 1.0f) return DCC_ERR_ARG;
  dim3 g(gs_blocks(n, 4));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((dropout_kernel<bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, n, p, seed, step);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((dropout_kernel<float>), g, dim3(256), 0, ST, (const float*)x, (float*)y, n, p, seed, step);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

extern "C" int dcc_scale_by_dev(float* x, int64_t n, const float* coef, void* stream) {
  if (!x || !coef || n <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(scale_by_dev_kernel, dim3(gs_blocks(n, 4)), dim3(256), 0, ST, x, n, coef);
  return dcc_check_launch();
}
