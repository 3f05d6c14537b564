// MViT decomposed relative-position bias as an augmented attention contraction.
//   bias[q][k] = <q_vec, Rh[q_h][k_h]> + <q_vec, Rw[q_w][k_w]> + <q_vec, Rt[q_t][k_t]>
// (reference: torchvision mvit._add_rel_pos, called from models/video_encoder.py:354-364).
// Instead of gathering a bias per score element, the per-query tables
//   tab(q) = [ <q,Rh[q_h][0..k_h)> | <q,Rw[q_w][0..k_w)> | <q,Rt[q_t][0..k_t)> ] / scale
// are written behind q in the attention's Q operand (columns D .. D+A) and a one-hot row
//   e(k) = [ 1{k_h} | 1{k_w} | 1{k_t} ]
// behind k in the K operand, so scale * <Qf, Kf> = scale * <q,k> + bias with the MFMA doing the
// "gather".  Class token rows/cols carry zeros (the bias applies to attn[:, :, 1:, 1:] only).
//
// The inner products themselves are ONE MFMA GEMM  P = q [rows, D] x Rcat^T [D, Lh+Lw+Lt]  against the three
// tables stacked (every table row, not only the k_h+k_w+k_t that a query needs: ~10x redundant FLOPs on the
// matrix cores beat a VALU gather by >10x); the kernels here only move P's entries into place:
//   gather  (forward):  Qf[row][D + j]  = P[row][base_a + idx_a[coord_a(row)][j_a]] / scale
//   scatter (backward): dP[row][l]      = dQf[row][D + off_a + inv_a[coord_a(row)][l - base_a]] / scale  (0 if none)
// after which d tables = dP^T q and dq += dP Rcat are GEMMs again.
#include "common.h"

namespace {

struct RelGeom {
  int B, H, D, A, L;     // D = head dim (96), A = augmentation width, L = padded stacked table length
  int qt, qh, qw, kt, kh, kw;
  int Lh, Lw, Lt;        // table lengths
  int has_cls;
  int64_t sb, sn, sh;    // strides of the Qf-like operand (b, token, head)
};

template <typename T> __device__ __forceinline__ void store8(T* dst, const float* v) {
  if constexpr (sizeof(T) == 2) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8*>(dst) = o;
  } else {
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}

// One thread = 8 consecutive columns of one (b, token, head) row (16-byte bf16 / 2 x 16-byte fp32 stores); the row's
// grid coordinates are decoded once per thread with 32-bit arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void relpos_gather_kernel(T* __restrict__ Qf, const T* __restrict__ P,
                                                            const int* __restrict__ idx_h, const int* __restrict__ idx_w,
                                                            const int* __restrict__ idx_t, RelGeom g, float inv_scale) {
  const int Nq = g.qt * g.qh * g.qw + g.has_cls;
  const int cpr = g.A / 8;  // A is a multiple of 32
  const unsigned total = (unsigned)g.B * Nq * g.H * cpr;
  const unsigned id = blockIdx.x * 256u + threadIdx.x;
  if (id >= total) return;
  const unsigned row = id / cpr;  // (b, n, head) row of P
  const int j0 = (id - row * cpr) * 8;
  const unsigned bn = row / g.H;
  const int hd = row - bn * g.H;
  const unsigned b = bn / Nq;
  const int n = bn - b * Nq;
  // Branch-free: the 8 table lookups are issued together, then the 8 gathers (an if-chain with a load in every arm made
  // the compiler wait for each load inside its branch: 16 serialised round trips per thread).
  float v[8];
  const bool tokrow = !(g.has_cls && n == 0);
  const int i = tokrow ? n - g.has_cls : 0;
  const int t = i / (g.qw * g.qh), r = i - t * (g.qw * g.qh);
  const int hh = r / g.qw, w = r - hh * g.qw;
  const T* prow = P + (int64_t)row * g.L;
  int l[8];
  bool ok[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int j = j0 + e;
    ok[e] = tokrow && j < g.kh + g.kw + g.kt;
    const int ax = (j >= g.kh) + (j >= g.kh + g.kw);
    const int* tp = ax == 0 ? idx_h : (ax == 1 ? idx_w : idx_t);
    const int off = ax == 0 ? hh * g.kh + j : (ax == 1 ? w * g.kw + (j - g.kh) : t * g.kt + (j - g.kh - g.kw));
    l[e] = tp[ok[e] ? off : 0] + (ax == 0 ? 0 : (ax == 1 ? g.Lh : g.Lh + g.Lw));
  }
  T pv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) pv[e] = prow[ok[e] ? l[e] : 0];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = ok[e] ? to_f32(pv[e]) * inv_scale : 0.0f;
  store8<T>(Qf + b * g.sb + (int64_t)n * g.sn + hd * g.sh + g.D + j0, v);
}

template <typename T>
__global__ __launch_bounds__(256) void relpos_scatter_kernel(const T* __restrict__ dQf, T* __restrict__ dP,
                                                             const int* __restrict__ inv_h, const int* __restrict__ inv_w,
                                                             const int* __restrict__ inv_t, RelGeom g, float inv_scale) {
  const int Nq = g.qt * g.qh * g.qw + g.has_cls;
  const int cpr = g.L / 8;  // L is a multiple of 32
  const unsigned total = (unsigned)g.B * Nq * g.H * cpr;
  const unsigned id = blockIdx.x * 256u + threadIdx.x;
  if (id >= total) return;
  const unsigned row = id / cpr;
  const int l0 = (id - row * cpr) * 8;
  const unsigned bn = row / g.H;
  const int hd = row - bn * g.H;
  const unsigned b = bn / Nq;
  const int n = bn - b * Nq;
  float v[8];
  const bool tokrow = !(g.has_cls && n == 0);
  const int i = tokrow ? n - g.has_cls : 0;
  const int t = i / (g.qw * g.qh), r = i - t * (g.qw * g.qh);
  const int hh = r / g.qw, w = r - hh * g.qw;
  const T* src = dQf + b * g.sb + (int64_t)n * g.sn + hd * g.sh + g.D;
  int j[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {  // branch-free, see relpos_gather_kernel
    const int l = l0 + e;
    const bool ok = tokrow && l < g.Lh + g.Lw + g.Lt;
    const int ax = (l >= g.Lh) + (l >= g.Lh + g.Lw);
    const int* tp = ax == 0 ? inv_h : (ax == 1 ? inv_w : inv_t);
    const int off = ax == 0 ? hh * g.Lh + l : (ax == 1 ? w * g.Lw + (l - g.Lh) : t * g.Lt + (l - g.Lh - g.Lw));
    const int jj = tp[ok ? off : 0];
    j[e] = (ok && jj >= 0) ? jj + (ax == 0 ? 0 : (ax == 1 ? g.kh : g.kh + g.kw)) : -1;
  }
  T sv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sv[e] = src[j[e] >= 0 ? j[e] : 0];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = j[e] >= 0 ? to_f32(sv[e]) * inv_scale : 0.0f;
  store8<T>(dP + (int64_t)row * g.L + l0, v);
}

// one-hot key augmentation, one thread per (b, n, head, j)
template <typename T>
__global__ void relpos_kaug_kernel(T* __restrict__ Kf, RelGeom g) {
  const int Nk = g.kt * g.kh * g.kw + g.has_cls;
  const int64_t total = (int64_t)g.B * Nk * g.H * g.A;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int j = id % g.A;
    const int hd = (id / g.A) % g.H;
    const int n = (id / ((int64_t)g.A * g.H)) % Nk;
    const int b = id / ((int64_t)g.A * g.H * Nk);
    float v = 0.0f;
    if (!(g.has_cls && n == 0)) {
      const int i = n - g.has_cls;
      const int w = i % g.kw, hh = (i / g.kw) % g.kh, t = i / (g.kw * g.kh);
      if (j == hh || j == g.kh + w || j == g.kh + g.kw + t) v = 1.0f;
    }
    Kf[b * g.sb + (int64_t)n * g.sn + hd * g.sh + g.D + j] = from_f32<T>(v);
  }
}

inline unsigned gsb(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (unsigned)b;
}

RelGeom mk(const int* d, const int64_t* s) {
  RelGeom g;
  g.B = d[0]; g.H = d[1]; g.D = d[2]; g.A = d[3]; g.L = d[4];
  g.qt = d[5]; g.qh = d[6]; g.qw = d[7]; g.kt = d[8]; g.kh = d[9]; g.kw = d[10];
  g.Lh = d[11]; g.Lw = d[12]; g.Lt = d[13]; g.has_cls = d[14];
  g.sb = s[0]; g.sn = s[1]; g.sh = s[2];
  return g;
}

}  // namespace

// dims: [B,H,D,A,L,qt,qh,qw,kt,kh,kw,Lh,Lw,Lt,has_cls]; strides: [b, token, head] of Qf (elements)
// P: dense [B*Nq*H, L] inner products against the stacked tables (rows in Qf's (b, token, head) order)
extern "C" int dcc_relpos_gather(void* Qf, const void* P, const int* idx_h, const int* idx_w, const int* idx_t,
                                 const int* dims, const int64_t* strides, float scale, int dtype, void* stream) {
  if (!Qf || !P || !idx_h || !idx_w || !idx_t || !dims || !strides) return DCC_ERR_ARG;
  RelGeom g = mk(dims, strides);
  if (g.A < g.kh + g.kw + g.kt || g.L < g.Lh + g.Lw + g.Lt) return DCC_ERR_ARG;
  if ((g.A % 8) || (g.L % 8)) return DCC_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)g.B * (g.qt * g.qh * g.qw + g.has_cls) * g.H * (g.A / 8);
  if (total >= (1LL << 31)) return DCC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((relpos_gather_kernel<bf16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (bf16_t*)Qf, (const bf16_t*)P, idx_h, idx_w, idx_t, g, 1.0f / scale);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((relpos_gather_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (float*)Qf, (const float*)P, idx_h, idx_w, idx_t, g, 1.0f / scale);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

// inv_a: int32 [q_a][L_a], the column j_a with idx_a[coord][j_a] == l, or -1
extern "C" int dcc_relpos_scatter(const void* dQf, void* dP, const int* inv_h, const int* inv_w, const int* inv_t,
                                  const int* dims, const int64_t* strides, float scale, int dtype, void* stream) {
  if (!dQf || !dP || !inv_h || !inv_w || !inv_t || !dims || !strides) return DCC_ERR_ARG;
  RelGeom g = mk(dims, strides);
  if ((g.A % 8) || (g.L % 8)) return DCC_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)g.B * (g.qt * g.qh * g.qw + g.has_cls) * g.H * (g.L / 8);
  if (total >= (1LL << 31)) return DCC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((relpos_scatter_kernel<bf16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)dQf, (bf16_t*)dP, inv_h, inv_w, inv_t, g, 1.0f / scale);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((relpos_scatter_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)dQf, (float*)dP, inv_h, inv_w, inv_t, g, 1.0f / scale);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

// dims as above (k grid used); strides of Kf
extern "C" int dcc_relpos_kaug(void* Kf, const int* dims, const int64_t* strides, int dtype, void* stream) {
  if (!Kf || !dims || !strides) return DCC_ERR_ARG;
  RelGeom g = mk(dims, strides);
  const int64_t total = (int64_t)g.B * (g.kt * g.kh * g.kw + g.has_cls) * g.H * g.A;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((relpos_kaug_kernel<bf16_t>), dim3(gsb(total)), dim3(256), 0, st, (bf16_t*)Kf, g);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((relpos_kaug_kernel<float>), dim3(gsb(total)), dim3(256), 0, st, (float*)Kf, g);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}
