// MViT decomposed relative-position bias as an augmented attention contraction.
//   bias[q][k] = <q_vec, Rh[q_h][k_h]> + <q_vec, Rw[q_w][k_w]> + <q_vec, Rt[q_t][k_t]>
// (reference: torchvision mvit._add_rel_pos, called from models/video_encoder.py:354-364).
// Instead of gathering a bias per score element, the per-query tables
//   tab(q) = [ <q,Rh[q_h][0..k_h)> | <q,Rw[q_w][0..k_w)> | <q,Rt[q_t][0..k_t)> ] / scale
// are written behind q in the attention's Q operand (columns 96 .. 96+A) and a one-hot row
//   e(k) = [ 1{k_h} | 1{k_w} | 1{k_t} ]
// behind k in the K operand, so scale * <Qf, Kf> = scale * <q,k> + bias with the MFMA doing the
// "gather".  Class token rows/cols carry zeros (the bias applies to attn[:, :, 1:, 1:] only).
#include "common.h"

namespace {

struct RelGeom {
  int B, H, D, A;        // D = head dim (96), A = augmentation width
  int qt, qh, qw, kt, kh, kw;
  int has_cls;
  int64_t sb, sn, sh;    // strides of the Qf-like operand (b, token, head)
};

// one thread per (b, n, head, j)
template <typename T>
__global__ void relpos_tab_fwd_kernel(T* __restrict__ Qf, const float* __restrict__ rel_h,
                                      const float* __restrict__ rel_w, const float* __restrict__ rel_t,
                                      const int* __restrict__ idx_h, const int* __restrict__ idx_w,
                                      const int* __restrict__ idx_t, RelGeom g, float inv_scale) {
  const int Nq = g.qt * g.qh * g.qw + g.has_cls;
  const int64_t total = (int64_t)g.B * Nq * g.H * g.A;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int j = id % g.A;
    const int hd = (id / g.A) % g.H;
    const int n = (id / ((int64_t)g.A * g.H)) % Nq;
    const int b = id / ((int64_t)g.A * g.H * Nq);
    T* row = Qf + b * g.sb + (int64_t)n * g.sn + hd * g.sh;
    float v = 0.0f;
    if (!(g.has_cls && n == 0) && j < g.kh + g.kw + g.kt) {
      const int i = n - g.has_cls;
      const int w = i % g.qw, hh = (i / g.qw) % g.qh, t = i / (g.qw * g.qh);
      const float* r;
      if (j < g.kh) r = rel_h + (int64_t)idx_h[hh * g.kh + j] * g.D;
      else if (j < g.kh + g.kw) r = rel_w + (int64_t)idx_w[w * g.kw + (j - g.kh)] * g.D;
      else r = rel_t + (int64_t)idx_t[t * g.kt + (j - g.kh - g.kw)] * g.D;
      float acc = 0.0f;
      for (int d = 0; d < g.D; ++d) acc += to_f32(row[d]) * r[d];
      v = acc * inv_scale;
    }
    row[g.D + j] = from_f32<T>(v);
  }
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

// dq[n][c] = dQf[n][c] + [residual && n >= has_cls] dO[n][c] + sum_j (dQf[n][D+j] / scale) * R_j[c]
// one thread per (b, n, head, 4-channel chunk); dq is dense [B, Nq, H, D].
template <typename T>
__global__ void relpos_tab_bwd_q_kernel(const T* __restrict__ dQf, const T* __restrict__ dO, int64_t ob, int64_t on,
                                        int64_t oh, T* __restrict__ dq, const float* __restrict__ rel_h,
                                        const float* __restrict__ rel_w, const float* __restrict__ rel_t,
                                        const int* __restrict__ idx_h, const int* __restrict__ idx_w,
                                        const int* __restrict__ idx_t, RelGeom g, float inv_scale) {
  const int Nq = g.qt * g.qh * g.qw + g.has_cls;
  const int cpr = g.D / 4;
  const int64_t total = (int64_t)g.B * Nq * g.H * cpr;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (id % cpr) * 4;
    const int hd = (id / cpr) % g.H;
    const int n = (id / ((int64_t)cpr * g.H)) % Nq;
    const int b = id / ((int64_t)cpr * g.H * Nq);
    const T* drow = dQf + b * g.sb + (int64_t)n * g.sn + hd * g.sh;
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = to_f32(drow[c + e]);
    if (!(g.has_cls && n == 0)) {
      if (dO) {
        const T* orow = dO + b * ob + (int64_t)n * on + hd * oh;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += to_f32(orow[c + e]);
      }
      const int i = n - g.has_cls;
      const int w = i % g.qw, hh = (i / g.qw) % g.qh, t = i / (g.qw * g.qh);
      const int na = g.kh + g.kw + g.kt;
      for (int j = 0; j < na; ++j) {
        const float dt = to_f32(drow[g.D + j]) * inv_scale;
        const float* r;
        if (j < g.kh) r = rel_h + (int64_t)idx_h[hh * g.kh + j] * g.D;
        else if (j < g.kh + g.kw) r = rel_w + (int64_t)idx_w[w * g.kw + (j - g.kh)] * g.D;
        else r = rel_t + (int64_t)idx_t[t * g.kt + (j - g.kh - g.kw)] * g.D;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += dt * r[c + e];
      }
    } else if (dO && !g.has_cls) {
      // no class token: residual applies to every row (handled above); nothing here
    }
    T* o = dq + ((((int64_t)b * Nq + n) * g.H + hd) * g.D) + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(acc[e]);
  }
}

// table gradient for one axis: drel[idx[g][ka]][c] += sum_{tokens with coord == g} (dQf[tok][D+off+ka]/scale) * q[tok][c]
// grid = (groups, splits); thread tasks = (ka, 4-channel chunk).
template <typename T>
__global__ __launch_bounds__(256) void relpos_tab_bwd_rel_kernel(const T* __restrict__ Qf, const T* __restrict__ dQf,
                                                                 float* __restrict__ drel, const int* __restrict__ idx,
                                                                 RelGeom g, int axis, float inv_scale) {
  const int grp = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  int ka_n, off, n1, n2;
  if (axis == 0) { ka_n = g.kh; off = 0; n1 = g.qt; n2 = g.qw; }
  else if (axis == 1) { ka_n = g.kw; off = g.kh; n1 = g.qt; n2 = g.qh; }
  else { ka_n = g.kt; off = g.kh + g.kw; n1 = g.qh; n2 = g.qw; }
  const int cpr = g.D / 4;
  const int ntask = ka_n * cpr;
  const int64_t members = (int64_t)g.B * g.H * n1 * n2;
  const int64_t per = (members + nsplit - 1) / nsplit;
  const int64_t m0 = split * per;
  int64_t m1 = m0 + per; if (m1 > members) m1 = members;
  for (int task = threadIdx.x; task < ntask; task += blockDim.x) {
    const int ka = task / cpr, c = (task % cpr) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t m = m0; m < m1; ++m) {
      const int i2 = m % n2;
      const int i1 = (m / n2) % n1;
      const int hd = (m / ((int64_t)n2 * n1)) % g.H;
      const int b = m / ((int64_t)n2 * n1 * g.H);
      int t, hh, w;
      if (axis == 0) { t = i1; hh = grp; w = i2; }
      else if (axis == 1) { t = i1; hh = i2; w = grp; }
      else { t = grp; hh = i1; w = i2; }
      const int n = g.has_cls + (t * g.qh + hh) * g.qw + w;
      const int64_t o = b * g.sb + (int64_t)n * g.sn + hd * g.sh;
      const float dt = to_f32(dQf[o + g.D + off + ka]) * inv_scale;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += dt * to_f32(Qf[o + c + e]);
    }
    float* dst = drel + (int64_t)idx[grp * ka_n + ka] * g.D + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dst + e, acc[e]);
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
  g.B = d[0]; g.H = d[1]; g.D = d[2]; g.A = d[3];
  g.qt = d[4]; g.qh = d[5]; g.qw = d[6]; g.kt = d[7]; g.kh = d[8]; g.kw = d[9]; g.has_cls = d[10];
  g.sb = s[0]; g.sn = s[1]; g.sh = s[2];
  return g;
}

}  // namespace

// dims: [B,H,D,A,qt,qh,qw,kt,kh,kw,has_cls]; strides: [b, token, head] of Qf (elements)
extern "C" int dcc_relpos_tab_fwd(void* Qf, const float* rel_h, const float* rel_w, const float* rel_t,
                                  const int* idx_h, const int* idx_w, const int* idx_t, const int* dims,
                                  const int64_t* strides, float scale, int dtype, void* stream) {
  if (!Qf || !rel_h || !rel_w || !rel_t || !idx_h || !idx_w || !idx_t || !dims || !strides) return DCC_ERR_ARG;
  RelGeom g = mk(dims, strides);
  if (g.A < g.kh + g.kw + g.kt || g.D % 4) return DCC_ERR_ARG;
  const int64_t total = (int64_t)g.B * (g.qt * g.qh * g.qw + g.has_cls) * g.H * g.A;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((relpos_tab_fwd_kernel<bf16_t>), dim3(gsb(total)), dim3(256), 0, st, (bf16_t*)Qf, rel_h, rel_w, rel_t, idx_h, idx_w, idx_t, g, 1.0f / scale);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((relpos_tab_fwd_kernel<float>), dim3(gsb(total)), dim3(256), 0, st, (float*)Qf, rel_h, rel_w, rel_t, idx_h, idx_w, idx_t, g, 1.0f / scale);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

// dims as above (the q* entries are ignored; k grid used); strides of Kf
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

// Backward: dq (dense [B,Nq,H,D]) and the three table gradients (fp32, accumulated into).
// dO/ostrides: upstream gradient of the attention output for the residual-pooling shortcut (null if unused).
extern "C" int dcc_relpos_tab_bwd(const void* Qf, const void* dQf, const void* dO, const int64_t* ostrides, void* dq,
                                  const float* rel_h, const float* rel_w, const float* rel_t, float* drel_h,
                                  float* drel_w, float* drel_t, const int* idx_h, const int* idx_w, const int* idx_t,
                                  const int* dims, const int64_t* strides, float scale, int dtype, void* stream) {
  if (!Qf || !dQf || !dq || !rel_h || !rel_w || !rel_t || !idx_h || !idx_w || !idx_t || !dims || !strides)
    return DCC_ERR_ARG;
  if (dO && !ostrides) return DCC_ERR_ARG;
  RelGeom g = mk(dims, strides);
  if (g.D % 4) return DCC_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)g.B * (g.qt * g.qh * g.qw + g.has_cls) * g.H * (g.D / 4);
  const float inv = 1.0f / scale;
  const int64_t ob = dO ? ostrides[0] : 0, on = dO ? ostrides[1] : 0, oh = dO ? ostrides[2] : 0;
  if (dtype == DCC_BF16) hipLaunchKernelGGL((relpos_tab_bwd_q_kernel<bf16_t>), dim3(gsb(total)), dim3(256), 0, st, (const bf16_t*)dQf, (const bf16_t*)dO, ob, on, oh, (bf16_t*)dq, rel_h, rel_w, rel_t, idx_h, idx_w, idx_t, g, inv);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((relpos_tab_bwd_q_kernel<float>), dim3(gsb(total)), dim3(256), 0, st, (const float*)dQf, (const float*)dO, ob, on, oh, (float*)dq, rel_h, rel_w, rel_t, idx_h, idx_w, idx_t, g, inv);
  else return DCC_ERR_UNSUPPORTED;
  int rc = dcc_check_launch();
  if (rc) return rc;
  float* drels[3] = {drel_h, drel_w, drel_t};
  const int* idxs[3] = {idx_h, idx_w, idx_t};
  const int groups[3] = {g.qh, g.qw, g.qt};
  for (int a = 0; a < 3; ++a) {
    if (!drels[a]) continue;
    const int64_t members = (int64_t)g.B * g.H * (g.qt * g.qh * g.qw / groups[a]);
    int nsplit = (int)((members + 127) / 128);
    const int cap = (2048 + groups[a] - 1) / groups[a];
    if (nsplit > cap) nsplit = cap;
    if (nsplit < 1) nsplit = 1;
    dim3 grid(groups[a], nsplit);
    if (dtype == DCC_BF16) hipLaunchKernelGGL((relpos_tab_bwd_rel_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)Qf, (const bf16_t*)dQf, drels[a], idxs[a], g, a, inv);
    else hipLaunchKernelGGL((relpos_tab_bwd_rel_kernel<float>), grid, dim3(256), 0, st, (const float*)Qf, (const float*)dQf, drels[a], idxs[a], g, a, inv);
    rc = dcc_check_launch();
    if (rc) return rc;
  }
  return DCC_OK;
}
