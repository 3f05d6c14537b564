// Contrastive loss kernels (reference: utils/loss/losses.py:23-64,173-211; utils/loss/contrastive.py:125-164,
// 230-315; utils/loss/weighted_siglip.py:18-51; runners/video_constrative_learning_runner.py:1256-1296).
// Everything is fp32 (the reference disables autocast for the loss).  The similarity GEMMs go
// through dcc_gemm (fp32 MFMA); these kernels do the L2 normalisation, the logit transform
// (temperature / SiLU gate / bias+clamp), row and column log-sum-exp with wavefront shuffles, the
// closed-form logit gradients, and the scalar reductions (loss, d log_temp, d bias, diagnostics).
#include "common.h"

namespace {

// ---- L2 normalise: y = x / max(||x||, eps) -------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                         float* __restrict__ norms, int R, int D, float eps,
                                                         float* __restrict__ norm_sum) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* xr = x + (int64_t)row * ldx;
  float s = 0.0f;
  for (int d = lane; d < D; d += 64) s += xr[d] * xr[d];
  s = wave_sum(s);
  const float nrm = sqrtf(s);
  const float inv = 1.0f / fmaxf(nrm, eps);
  for (int d = lane; d < D; d += 64) y[(int64_t)row * D + d] = xr[d] * inv;
  if (lane == 0 && norms) norms[row] = nrm;
  if (lane == 0 && norm_sum) atomicAdd(norm_sum, nrm);
}
// dx = (dy - y <y,dy>) / max(||x||, eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ norms, float* __restrict__ dx,
                                                         int64_t lddx, int R, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* dyr = dy + (int64_t)row * D;
  const float* yr = y + (int64_t)row * D;
  float s = 0.0f;
  for (int d = lane; d < D; d += 64) s += dyr[d] * yr[d];
  s = wave_sum(s);
  const float inv = 1.0f / fmaxf(norms[row], eps);
  for (int d = lane; d < D; d += 64) dx[(int64_t)row * lddx + d] = (dyr[d] - yr[d] * s) * inv;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float temp_of(const float* log_temp, int clamp_temp, bool* clamped) {
  float t = __expf(*log_temp);
  *clamped = false;
  if (clamp_temp && t < 1e-4f) { t = 1e-4f; *clamped = true; }
  return t;
}

// ---- logits: mode 0: s/T   1: s*sigmoid(s)/T   2: clamp(s/T + bias, +-30);  + col_margin[m] -------
__global__ void logits_fwd_kernel(const float* __restrict__ sim, float* __restrict__ z, int B, int M, int mode,
                                  const float* __restrict__ log_temp, int clamp_temp, const float* __restrict__ bias,
                                  const float* __restrict__ col_margin) {
  bool cl;
  const float invT = 1.0f / temp_of(log_temp, clamp_temp, &cl);
  const float bv = (mode == 2 && bias) ? *bias : 0.0f;
  const int64_t n = (int64_t)B * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float s = sim[i];
    float v;
    if (mode == 0) v = s * invT;
    else if (mode == 1) v = s * sigmoidf_(s) * invT;
    else v = fminf(fmaxf(s * invT + bv, -30.0f), 30.0f);
    if (col_margin) v += col_margin[i % M];
    z[i] = v;
  }
}

// dsim = dz * dz/ds, scal[1] += d log_temp, scal[2] += d bias
__global__ __launch_bounds__(256) void logits_bwd_kernel(const float* __restrict__ sim, const float* __restrict__ dz,
                                                         float* __restrict__ dsim, int B, int M, int mode,
                                                         const float* __restrict__ log_temp, int clamp_temp,
                                                         const float* __restrict__ bias, float* __restrict__ scal) {
  bool cl;
  const float invT = 1.0f / temp_of(log_temp, clamp_temp, &cl);
  const float bv = (mode == 2 && bias) ? *bias : 0.0f;
  const int64_t n = (int64_t)B * M;
  float dlt = 0.0f, dbs = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float s = sim[i], g = dz[i];
    float ds, zt;  // zt = the temperature-dependent part of z
    if (mode == 0) { ds = invT; zt = s * invT; }
    else if (mode == 1) {
      const float sg = sigmoidf_(s);
      ds = (sg + s * sg * (1.0f - sg)) * invT;
      zt = s * sg * invT;
    } else {
      const float u = s * invT + bv;
      const bool inside = (u >= -30.0f) && (u <= 30.0f);
      ds = inside ? invT : 0.0f;
      zt = inside ? s * invT : 0.0f;
      if (inside) dbs += g;
    }
    dsim[i] = g * ds;
    dlt -= g * zt;
  }
  dlt = wave_sum(dlt);
  dbs = wave_sum(dbs);
  if ((threadIdx.x & 63) == 0) {
    if (!cl && dlt != 0.0f) atomicAdd(scal + 1, dlt);
    if (dbs != 0.0f) atomicAdd(scal + 2, dbs);
  }
}

// ---- log-sum-exp over rows (wave per row) and columns (thread per column) -------------------------
__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ z, float* __restrict__ lse, int B, int M) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* zr = z + (int64_t)row * M;
  float mx = -INFINITY;
  for (int j = lane; j < M; j += 64) mx = fmaxf(mx, zr[j]);
  mx = wave_max(mx);
  float s = 0.0f;
  for (int j = lane; j < M; j += 64) s += __expf(zr[j] - mx);
  s = wave_sum(s);
  if (lane == 0) lse[row] = mx + __logf(s);
}
__global__ __launch_bounds__(256) void lse_cols_kernel(const float* __restrict__ z, float* __restrict__ lse, int B, int M) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= M) return;
  float mx = -INFINITY, s = 0.0f;
  for (int i = 0; i < B; ++i) {
    const float v = z[(int64_t)i * M + col];
    if (v > mx) { s = s * __expf(mx - v) + 1.0f; mx = v; }
    else s += __expf(v - mx);
  }
  lse[col] = mx + __logf(s);
}

// ---- symmetric diagonal-target cross entropy: loss = 0.5 (mean_i CE_row_i + mean_j CE_col_j) ------
// dz = 0.5/B (p_row - (1-e) I - e/M) + 0.5/M (p_col - (1-e) I - e/B)
// scal[0] += loss, scal[3] += mean_i (z_ii - lse_row_i)  (alignment log-prob), needs B == M.
__global__ __launch_bounds__(256) void ce_diag_kernel(const float* __restrict__ z, const float* __restrict__ lse_r,
                                                      const float* __restrict__ lse_c, float* __restrict__ dz, int B,
                                                      float smooth, float* __restrict__ scal,
                                                      const float* __restrict__ sim) {
  const int M = B;
  const int64_t n = (int64_t)B * M;
  const float a = 0.5f / B;
  float loss = 0.0f, align = 0.0f, cosd = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = i / M, c = i % M;
    const float v = z[i];
    const float pr = __expf(v - lse_r[r]), pc = __expf(v - lse_c[c]);
    const float tgt = (r == c ? (1.0f - smooth) : 0.0f) + smooth / M;
    if (dz) dz[i] = a * (pr - tgt) + a * (pc - tgt);
    // CE_row_r = sum_c tgt (lse_r - z);  CE_col_c = sum_r tgt (lse_c - z)
    loss += a * tgt * (lse_r[r] - v) + a * tgt * (lse_c[c] - v);
    if (r == c) {
      align += (v - lse_r[r]) / B;
      if (sim) cosd += sim[i] / B;
    }
  }
  loss = wave_sum(loss);
  align = wave_sum(align);
  cosd = wave_sum(cosd);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(scal + 0, loss);
    atomicAdd(scal + 3, align);
    if (sim) atomicAdd(scal + 4, cosd);
  }
}

// ---- row / column sums of a [B,M] matrix ---------------------------------------------------------
__global__ void rowcol_sums_kernel(const float* __restrict__ w, float* __restrict__ rs, float* __restrict__ cs, int B,
                                   int M) {
  const int64_t n = (int64_t)B * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = fmaxf(w[i], 0.0f);
    if (v != 0.0f) {
      atomicAdd(rs + i / M, v);
      atomicAdd(cs + i % M, v);
    }
  }
}

// ---- weighted bidirectional log-softmax CE (WeightedSigLIPLoss) -------------------------------------
// L = 0.5 (mean_i [-sum_j pos_ij lp_row_ij / max(S_i,eps)] + mean_j [-sum_i pos_ij lp_col_ij / max(C_j,eps)])
__global__ __launch_bounds__(256) void weighted_ce_kernel(const float* __restrict__ z, const float* __restrict__ pos,
                                                          const float* __restrict__ lse_r, const float* __restrict__ lse_c,
                                                          const float* __restrict__ rs, const float* __restrict__ cs,
                                                          float* __restrict__ dz, int B, int M, float eps,
                                                          float* __restrict__ scal) {
  const int64_t n = (int64_t)B * M;
  float loss = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = i / M, c = i % M;
    const float v = z[i], p = fmaxf(pos[i], 0.0f);
    const float dr = fmaxf(rs[r], eps), dc = fmaxf(cs[c], eps);
    const float lpr = v - lse_r[r], lpc = v - lse_c[c];
    loss += -0.5f * p * lpr / (dr * B) - 0.5f * p * lpc / (dc * M);
    if (dz) dz[i] = 0.5f / B * (__expf(lpr) * rs[r] - p) / dr + 0.5f / M * (__expf(lpc) * cs[c] - p) / dc;
  }
  loss = wave_sum(loss);
  if ((threadIdx.x & 63) == 0) atomicAdd(scal + 0, loss);
}

// ---- weighted BCE-with-logits: loss = sum w bce(z,y) / denom;  dz = w (sigmoid(z) - y) / denom ------
// denom = denom_host if denom_dev == null else max(1, *denom_dev)
// weights: w = (y > thr) ? pw[i] * pos_weight (or pos_weight if pw null) : neg_weight
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                  const float* __restrict__ pw, float pos_weight, float neg_weight,
                                                  float thr, float denom_host, const float* __restrict__ denom_dev,
                                                  float* __restrict__ dz, int64_t n, float* __restrict__ scal) {
  const float denom = denom_dev ? fmaxf(1.0f, *denom_dev) : denom_host;
  float loss = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = z[i], t = y[i];
    const float w = (t > thr) ? (pw ? pw[i] * pos_weight : pos_weight) : neg_weight;
    const float l = fmaxf(v, 0.0f) - v * t + log1pf(__expf(-fabsf(v)));
    loss += w * l / denom;
    if (dz) dz[i] = w * (sigmoidf_(v) - t) / denom;
  }
  loss = wave_sum(loss);
  if ((threadIdx.x & 63) == 0) atomicAdd(scal + 0, loss);
}

inline unsigned gsb(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---- multi-positive alignment diagnostics (runner.py:1298-1313 of the reference, under no_grad) --------------------------
// one wave per video row: out[0] += sum_j logp_ij t_ij w_ij / sum_j t_ij w_ij, out[1] += [that row sum > 0],
// out[2..3] the same with the weights replaced by 1 (the reference's fallback when every weight is zero),
// out[4] += sum_{t_ij != 0} sim_ij, out[5] += #{t_ij != 0}, out[6] += sum_j |w_ij|.  logp = z - logsumexp_j z.
__global__ __launch_bounds__(256) void mp_align_kernel(const float* __restrict__ z, const float* __restrict__ sim,
                                                       const float* __restrict__ t, const float* __restrict__ w,
                                                       int B, int M, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* zr = z + (int64_t)row * M;
  float mx = -INFINITY;
  for (int j = lane; j < M; j += 64) mx = fmaxf(mx, zr[j]);
  mx = wave_max(mx);
  float se = 0.0f;
  for (int j = lane; j < M; j += 64) se += __expf(zr[j] - mx);
  const float lse = mx + __logf(wave_sum(se));
  float a_tw = 0.0f, s_tw = 0.0f, a_t = 0.0f, s_t = 0.0f, c_sum = 0.0f, c_cnt = 0.0f, w_abs = 0.0f;
  for (int j = lane; j < M; j += 64) {
    const int64_t i = (int64_t)row * M + j;
    const float tv = t[i], wv = w ? w[i] : 1.0f, lp = zr[j] - lse;
    a_tw += lp * tv * wv; s_tw += tv * wv; a_t += lp * tv; s_t += tv;
    if (tv != 0.0f) { c_sum += sim[i]; c_cnt += 1.0f; }
    w_abs += fabsf(wv);
  }
  a_tw = wave_sum(a_tw); s_tw = wave_sum(s_tw); a_t = wave_sum(a_t); s_t = wave_sum(s_t);
  c_sum = wave_sum(c_sum); c_cnt = wave_sum(c_cnt); w_abs = wave_sum(w_abs);
  if (lane == 0) {
    if (s_tw > 0.0f) { atomicAdd(out + 0, a_tw / s_tw); atomicAdd(out + 1, 1.0f); }
    if (s_t > 0.0f) { atomicAdd(out + 2, a_t / s_t); atomicAdd(out + 3, 1.0f); }
    atomicAdd(out + 4, c_sum); atomicAdd(out + 5, c_cnt); atomicAdd(out + 6, w_abs);
  }
}

// out = (sum |w| > 0) ? t * w : t   (reference :1276-1278: targets * positive_weights unless every weight is zero)
__global__ void abs_sum_kernel(const float* __restrict__ w, int64_t n, float* __restrict__ out) {
  float s = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += fabsf(w[i]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0 && s != 0.0f) atomicAdd(out, s);
}
__global__ void select_mul_kernel(const float* __restrict__ t, const float* __restrict__ w, const float* __restrict__ nz,
                                  float* __restrict__ out, int64_t n) {
  const bool use_w = *nz > 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = use_w ? t[i] * w[i] : t[i];
}

// ---- auto-balanced positive weights (utils/loss/contrastive.py:291-296, siglip_pairwise.py:161-166) ---------------------
// out[i][j] = max((T - n_i) / n_i, 1) with n_i = max(sum_j counts[i][j], 1): one wave per row
__global__ __launch_bounds__(256) void auto_balance_kernel(const float* __restrict__ counts, float* __restrict__ out, int B, int T) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float n = 0.0f;
  for (int j = lane; j < T; j += 64) n += counts[(int64_t)row * T + j];
  n = fmaxf(wave_sum(n), 1.0f);
  const float ratio = fmaxf(((float)T - n) / n, 1.0f);
  for (int j = lane; j < T; j += 64) out[(int64_t)row * T + j] = ratio;
}

// ---- entropy regulariser (utils/loss/contrastive.py:19-68): L_ent = relu(thr - mean_i H_i), H_i = -sum_j p_ij log(p_ij + 1e-10)
// pass 1: row entropies -> ent[0] += H_i, ent[1] = min, ent[2] = max (atomics on the bit patterns of non-negative floats)
__global__ __launch_bounds__(256) void entropy_rows_kernel(const float* __restrict__ z, int B, int M, float* __restrict__ ent) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* zr = z + (int64_t)row * M;
  float mx = -INFINITY;
  for (int j = lane; j < M; j += 64) mx = fmaxf(mx, zr[j]);
  mx = wave_max(mx);
  float se = 0.0f;
  for (int j = lane; j < M; j += 64) se += __expf(zr[j] - mx);
  const float inv = 1.0f / wave_sum(se);
  float h = 0.0f;
  for (int j = lane; j < M; j += 64) {
    const float p = __expf(zr[j] - mx) * inv;
    h -= p * __logf(p + 1e-10f);
  }
  h = wave_sum(h);
  if (lane == 0) {
    atomicAdd(ent + 0, h);
    const float hc = fmaxf(h, 0.0f);  // H >= 0 up to rounding: integer order == float order
    atomicMin(reinterpret_cast<unsigned*>(ent + 1), __float_as_uint(hc));
    atomicMax(reinterpret_cast<unsigned*>(ent + 2), __float_as_uint(hc));
  }
}
// pass 2: scal[0] += weight * deficit; dz_ij += weight / B * p_ij (g_ij - sum_k p_ik g_ik), g = log(p + eps) + p / (p + eps),
// only while the deficit is positive (decided on the device)
__global__ __launch_bounds__(256) void entropy_grad_kernel(const float* __restrict__ z, float* __restrict__ dz, int B, int M,
                                                           float weight, float thr, const float* __restrict__ ent,
                                                           float* __restrict__ scal) {
  const float deficit = thr - ent[0] / (float)B;
  if (deficit <= 0.0f) return;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  if (row == 0 && lane == 0) atomicAdd(scal + 0, weight * deficit);
  if (!dz) return;
  const float* zr = z + (int64_t)row * M;
  float mx = -INFINITY;
  for (int j = lane; j < M; j += 64) mx = fmaxf(mx, zr[j]);
  mx = wave_max(mx);
  float se = 0.0f;
  for (int j = lane; j < M; j += 64) se += __expf(zr[j] - mx);
  const float inv = 1.0f / wave_sum(se);
  float gb = 0.0f;
  for (int j = lane; j < M; j += 64) {
    const float p = __expf(zr[j] - mx) * inv;
    gb += p * (__logf(p + 1e-10f) + p / (p + 1e-10f));
  }
  gb = wave_sum(gb);
  const float c = weight / (float)B;
  for (int j = lane; j < M; j += 64) {
    const float p = __expf(zr[j] - mx) * inv;
    dz[(int64_t)row * M + j] += c * p * (__logf(p + 1e-10f) + p / (p + 1e-10f) - gb);
  }
}

// ---- symmetric multi-positive InfoNCE on given logits (utils/loss/multi_positive_infonce.py:9-110) -----------------------
// per-row / per-column weighted CE with the weights normalised by max(sum, 1); rows / columns without a positive in the
// MASK are left out; reduction over the concatenation of the selected rows and columns (mean | sum | importance-weighted).
// Kernel 1 (one block): coefficient of every row / column + the scalar loss, from the per-row statistics.
//   wsum_r/c: sum of clamped weights, msum_r/c: sum of the mask, imp_r/c: importance base sums (nullable),
//   nll_r/c: -sum_j w_ij logp_ij (unnormalised).  coef_r[i] = multiplier of loss_i in the total (0 when not selected).
__global__ __launch_bounds__(256) void mpnce_coef_kernel(const float* wsum_r, const float* msum_r, const float* imp_r,
                                                         const float* nll_r, float* coef_r, int N, const float* wsum_c,
                                                         const float* msum_c, const float* imp_c, const float* nll_c,
                                                         float* coef_c, int M, int mean, float imp_eps, float* scal) {
  __shared__ float red[2][4];
  float cnt = 0.0f, tw = 0.0f;
  const bool use_imp = imp_r != nullptr;
  for (int i = threadIdx.x; i < N + M; i += 256) {
    const bool row = i < N;
    const int k = row ? i : i - N;
    const bool sel = (row ? msum_r[k] : msum_c[k]) > 0.0f;
    if (sel) { cnt += 1.0f; if (use_imp) tw += fmaxf(row ? imp_r[k] : imp_c[k], imp_eps); }
  }
  cnt = wave_sum(cnt); tw = wave_sum(tw);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = cnt; red[1][threadIdx.x >> 6] = tw; }
  __syncthreads();
  cnt = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  tw = fmaxf(red[1][0] + red[1][1] + red[1][2] + red[1][3], imp_eps);
  float loss = 0.0f;
  for (int i = threadIdx.x; i < N + M; i += 256) {
    const bool row = i < N;
    const int k = row ? i : i - N;
    const bool sel = (row ? msum_r[k] : msum_c[k]) > 0.0f;
    float c = 0.0f;
    if (sel) {
      if (use_imp) c = fmaxf(row ? imp_r[k] : imp_c[k], imp_eps) / (mean ? tw : 1.0f);
      else c = mean ? 1.0f / cnt : 1.0f;
    }
    const float li = (row ? nll_r[k] : nll_c[k]) / fmaxf(row ? wsum_r[k] : wsum_c[k], 1.0f);
    (row ? coef_r : coef_c)[k] = c;
    loss += c * li;
  }
  loss = wave_sum(loss);
  if ((threadIdx.x & 63) == 0) atomicAdd(scal + 0, loss);
}
// Kernel 0: nll_r[i] += -w_ij (z_ij - lse_r[i]), nll_c[j] likewise (atomics; matrices here are a few hundred squared)
__global__ void mpnce_nll_kernel(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ lse_r,
                                 const float* __restrict__ lse_c, float* __restrict__ nll_r, float* __restrict__ nll_c, int N, int M) {
  const int64_t n = (int64_t)N * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = i / M, c = i % M;
    const float wv = fmaxf(w[i], 0.0f);
    if (wv != 0.0f) { atomicAdd(nll_r + r, -wv * (z[i] - lse_r[r])); atomicAdd(nll_c + c, -wv * (z[i] - lse_c[c])); }
  }
}
// Kernel 2: dz_ij = coef_r[i] (p_row_ij W_i - w_ij) / max(W_i, 1) + coef_c[j] (p_col_ij C_j - w_ij) / max(C_j, 1)
__global__ void mpnce_grad_kernel(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ lse_r,
                                  const float* __restrict__ lse_c, const float* __restrict__ wsum_r, const float* __restrict__ wsum_c,
                                  const float* __restrict__ coef_r, const float* __restrict__ coef_c, float* __restrict__ dz, int N, int M) {
  const int64_t n = (int64_t)N * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = i / M, c = i % M;
    const float wv = fmaxf(w[i], 0.0f), v = z[i];
    dz[i] = coef_r[r] * (__expf(v - lse_r[r]) * wsum_r[r] - wv) / fmaxf(wsum_r[r], 1.0f) +
            coef_c[c] * (__expf(v - lse_c[c]) * wsum_c[c] - wv) / fmaxf(wsum_c[c], 1.0f);
  }
}

}  // namespace

#define ST (hipStream_t) stream

// out = max(a * b, 0) (b nullable: max(a, 0)) — the clamped positive weights of the multi-positive losses
__global__ void mul_clamp0_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = fmaxf(b ? a[i] * b[i] : a[i], 0.0f);
}
extern "C" int dcc_mul_clamp0(const float* a, const float* b, float* out, int64_t n, void* stream) {
  if (!a || !out || n <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(mul_clamp0_kernel, dim3(gsb(n)), dim3(256), 0, ST, a, b, out, n);
  return dcc_check_launch();
}
__global__ void clamp_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float lo, float hi) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = fminf(fmaxf(x[i], lo), hi);
}
extern "C" int dcc_clamp(const float* x, float* out, int64_t n, float lo, float hi, void* stream) {
  if (!x || !out || n <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(clamp_kernel, dim3(gsb(n)), dim3(256), 0, ST, x, out, n, lo, hi);
  return dcc_check_launch();
}
extern "C" int dcc_auto_balance_weights(const float* counts, float* out, int B, int T, void* stream) {
  if (!counts || !out || B <= 0 || T <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(auto_balance_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, counts, out, B, T);
  return dcc_check_launch();
}
// ent: fp32[4] workspace initialised by the caller to {0, +inf, 0, 0}; afterwards {sum_i H_i, min_i H_i, max_i H_i, -}.
// scal[0] += weight * relu(thr - mean H); dz (nullable) += its gradient.
extern "C" int dcc_entropy_reg(const float* z, float* dz, int B, int M, float weight, float thr, float* ent, float* scal,
                               void* stream) {
  if (!z || !ent || !scal || B <= 0 || M <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(entropy_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, z, B, M, ent);
  hipLaunchKernelGGL(entropy_grad_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, z, dz, B, M, weight, thr, ent, scal);
  return dcc_check_launch();
}
// ws: fp32 workspace of 2 * (N + M) zero-initialised floats (nll_r, nll_c, coef_r, coef_c)
extern "C" int dcc_mp_infonce(const float* z, const float* w, const float* lse_r, const float* lse_c, const float* wsum_r,
                              const float* wsum_c, const float* msum_r, const float* msum_c, const float* imp_r,
                              const float* imp_c, float* ws, float* dz, int N, int M, int mean, float imp_eps, float* scal,
                              void* stream) {
  if (!z || !w || !lse_r || !lse_c || !wsum_r || !wsum_c || !msum_r || !msum_c || !ws || !scal || N <= 0 || M <= 0) return DCC_ERR_ARG;
  if ((imp_r == nullptr) != (imp_c == nullptr)) return DCC_ERR_ARG;
  float *nll_r = ws, *nll_c = ws + N, *coef_r = ws + N + M, *coef_c = ws + 2 * N + M;
  hipLaunchKernelGGL(mpnce_nll_kernel, dim3(gsb((int64_t)N * M)), dim3(256), 0, ST, z, w, lse_r, lse_c, nll_r, nll_c, N, M);
  hipLaunchKernelGGL(mpnce_coef_kernel, dim3(1), dim3(256), 0, ST, wsum_r, msum_r, imp_r, nll_r, coef_r, N, wsum_c, msum_c,
                     imp_c, nll_c, coef_c, M, mean, imp_eps, scal);
  if (dz)
    hipLaunchKernelGGL(mpnce_grad_kernel, dim3(gsb((int64_t)N * M)), dim3(256), 0, ST, z, w, lse_r, lse_c, wsum_r, wsum_c,
                       coef_r, coef_c, dz, N, M);
  return dcc_check_launch();
}

extern "C" int dcc_mp_align(const float* z, const float* sim, const float* targets, const float* weights, int B, int M,
                            float* out, void* stream) {
  if (!z || !sim || !targets || !out || B <= 0 || M <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(mp_align_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, z, sim, targets, weights, B, M, out);
  return dcc_check_launch();
}
// nz: fp32[1] zero-initialised workspace (receives sum |weights|)
extern "C" int dcc_mp_pos_weights(const float* targets, const float* weights, float* out, int64_t n, float* nz, void* stream) {
  if (!targets || !weights || !out || !nz || n <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(abs_sum_kernel, dim3(gsb(n)), dim3(256), 0, ST, weights, n, nz);
  hipLaunchKernelGGL(select_mul_kernel, dim3(gsb(n)), dim3(256), 0, ST, targets, weights, nz, out, n);
  return dcc_check_launch();
}

extern "C" int dcc_l2norm_fwd(const float* x, int64_t ldx, float* y, float* norms, int R, int D, float eps,
                              float* norm_sum, void* stream) {
  if (!x || !y || R <= 0 || D <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, x, ldx, y, norms, R, D, eps, norm_sum);
  return dcc_check_launch();
}
extern "C" int dcc_l2norm_bwd(const float* dy, const float* y, const float* norms, float* dx, int64_t lddx, int R, int D,
                              float eps, void* stream) {
  if (!dy || !y || !norms || !dx || R <= 0 || D <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, dy, y, norms, dx, lddx, R, D, eps);
  return dcc_check_launch();
}
extern "C" int dcc_logits_fwd(const float* sim, float* z, int B, int M, int mode, const float* log_temp, int clamp_temp,
                              const float* bias, const float* col_margin, void* stream) {
  if (!sim || !z || !log_temp || B <= 0 || M <= 0 || mode < 0 || mode > 2) return DCC_ERR_ARG;
  hipLaunchKernelGGL(logits_fwd_kernel, dim3(gsb((int64_t)B * M)), dim3(256), 0, ST, sim, z, B, M, mode, log_temp,
                     clamp_temp, bias, col_margin);
  return dcc_check_launch();
}
// scal: fp32[8] device scalars {loss, dlog_temp, dbias, align_logprob, ...}, accumulated into
extern "C" int dcc_logits_bwd(const float* sim, const float* dz, float* dsim, int B, int M, int mode,
                              const float* log_temp, int clamp_temp, const float* bias, float* scal, void* stream) {
  if (!sim || !dz || !dsim || !log_temp || !scal || B <= 0 || M <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(logits_bwd_kernel, dim3(gsb((int64_t)B * M)), dim3(256), 0, ST, sim, dz, dsim, B, M, mode,
                     log_temp, clamp_temp, bias, scal);
  return dcc_check_launch();
}
extern "C" int dcc_lse(const float* z, float* lse_rows, float* lse_cols, int B, int M, void* stream) {
  if (!z || B <= 0 || M <= 0) return DCC_ERR_ARG;
  if (lse_rows) hipLaunchKernelGGL(lse_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, z, lse_rows, B, M);
  if (lse_cols) hipLaunchKernelGGL(lse_cols_kernel, dim3((M + 255) / 256), dim3(256), 0, ST, z, lse_cols, B, M);
  return dcc_check_launch();
}
extern "C" int dcc_ce_diag(const float* z, const float* lse_r, const float* lse_c, float* dz, int B, float smooth,
                           float* scal, const float* sim, void* stream) {
  if (!z || !lse_r || !lse_c || !scal || B <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(ce_diag_kernel, dim3(gsb((int64_t)B * B)), dim3(256), 0, ST, z, lse_r, lse_c, dz, B, smooth, scal, sim);
  return dcc_check_launch();
}
extern "C" int dcc_rowcol_sums(const float* w, float* rs, float* cs, int B, int M, void* stream) {
  if (!w || !rs || !cs || B <= 0 || M <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(rowcol_sums_kernel, dim3(gsb((int64_t)B * M)), dim3(256), 0, ST, w, rs, cs, B, M);
  return dcc_check_launch();
}
extern "C" int dcc_weighted_ce(const float* z, const float* pos, const float* lse_r, const float* lse_c, const float* rs,
                               const float* cs, float* dz, int B, int M, float eps, float* scal, void* stream) {
  if (!z || !pos || !lse_r || !lse_c || !rs || !cs || !scal || B <= 0 || M <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(weighted_ce_kernel, dim3(gsb((int64_t)B * M)), dim3(256), 0, ST, z, pos, lse_r, lse_c, rs, cs, dz,
                     B, M, eps, scal);
  return dcc_check_launch();
}
extern "C" int dcc_bce(const float* z, const float* y, const float* pw, float pos_weight, float neg_weight, float thr,
                       float denom_host, const float* denom_dev, float* dz, int64_t n, float* scal, void* stream) {
  if (!z || !y || !scal || n <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(bce_kernel, dim3(gsb(n)), dim3(256), 0, ST, z, y, pw, pos_weight, neg_weight, thr, denom_host,
                     denom_dev, dz, n, scal);
  return dcc_check_launch();
}
