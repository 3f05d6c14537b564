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

// multiply-shift division by launch constants (exact for n < 2^40 / d; rows stay far below)
struct FastDiv { unsigned d; unsigned long long m; };
inline FastDiv mkdiv(unsigned d) { FastDiv f; f.d = d ? d : 1; f.m = ((1ull << 40) / f.d) + 1; return f; }
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return (unsigned)(((unsigned long long)n * f.m) >> 40); }
struct RelDiv { FastDiv H, Nq, plane, qw, cl, ca; };

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
                                                            const int* __restrict__ idx_t, RelGeom g, RelDiv dv, float inv_scale) {
  const int Nq = g.qt * g.qh * g.qw + g.has_cls;
  const int cpr = g.A / 8;  // A is a multiple of 32
  const unsigned total = (unsigned)g.B * Nq * g.H * cpr;
  const unsigned id = blockIdx.x * 256u + threadIdx.x;
  if (id >= total) return;
  const unsigned row = fdiv(id, dv.ca);  // (b, n, head) row of P
  const int j0 = (id - row * cpr) * 8;
  const unsigned bn = fdiv(row, dv.H);
  const int hd = row - bn * g.H;
  const unsigned b = fdiv(bn, dv.Nq);
  const int n = bn - b * Nq;
  // Branch-free: the 8 table lookups are issued together, then the 8 gathers (an if-chain with a load in every arm made
  // the compiler wait for each load inside its branch: 16 serialised round trips per thread).
  float v[8];
  const bool tokrow = !(g.has_cls && n == 0);
  const int i = tokrow ? n - g.has_cls : 0;
  const int t = fdiv(i, dv.plane), r = i - t * (g.qw * g.qh);
  const int hh = fdiv(r, dv.qw), w = r - hh * g.qw;
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

// LDS-staged scatter: a block owns RPB rows.  The three inverse tables (a few KB) and the rows' augmentation columns are
// staged in LDS once, so a thread's 8 lookups + 8 picks are LDS reads (the direct version issues 16 dependent global
// loads per 16 bytes of output and runs at ~1 TB/s).
constexpr int RPB = 32;
template <typename T, bool STAGE_TABLES>
__global__ __launch_bounds__(256) void relpos_scatter_lds_kernel(const T* __restrict__ dQf, T* __restrict__ dP,
                                                                 const int* __restrict__ inv_h, const int* __restrict__ inv_w,
                                                                 const int* __restrict__ inv_t, RelGeom g, RelDiv dv, float inv_scale) {
  extern __shared__ __attribute__((aligned(16))) int lds_i[];
  const int nh = g.qh * g.Lh, nw = g.qw * g.Lw, nt = g.qt * g.Lt;
  // small tables (the 14x14 / 7x7 stages: < 8 KB) are staged too; the 56x56 stage's 50 KB stay in global memory / L1
  const int* th = inv_h; const int* tw = inv_w; const int* tt = inv_t;
  T* aug = reinterpret_cast<T*>(lds_i);   // [RPB][A]
  if constexpr (STAGE_TABLES) {
    int* sh_ = lds_i; int* sw_ = sh_ + nh; int* st_ = sw_ + nw;
    aug = reinterpret_cast<T*>(lds_i + ((nh + nw + nt + 3) & ~3));
    for (int i = threadIdx.x; i < nh; i += 256) sh_[i] = inv_h[i];
    for (int i = threadIdx.x; i < nw; i += 256) sw_[i] = inv_w[i];
    for (int i = threadIdx.x; i < nt; i += 256) st_[i] = inv_t[i];
    th = sh_; tw = sw_; tt = st_;
  }
  const int Nq = g.qt * g.qh * g.qw + g.has_cls;
  const unsigned rows = (unsigned)g.B * Nq * g.H;
  const unsigned row0 = blockIdx.x * RPB;
  const int ca = g.A / 8;
  for (int i = threadIdx.x; i < RPB * ca; i += 256) {
    const unsigned lr = fdiv(i, dv.ca), c = i - lr * ca, row = row0 + lr;
    if (row < rows) {
      const unsigned bn = fdiv(row, dv.H);
      const int hd = row - bn * g.H;
      const unsigned b = fdiv(bn, dv.Nq);
      const int n = bn - b * Nq;
      const T* src = dQf + b * g.sb + (int64_t)n * g.sn + hd * g.sh + g.D + c * 8;
      if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x8*>(aug + lr * g.A + c * 8) = *reinterpret_cast<const bf16x8*>(src);
      else {
        *reinterpret_cast<f32x4*>(aug + lr * g.A + c * 8) = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(aug + lr * g.A + c * 8 + 4) = *reinterpret_cast<const f32x4*>(src + 4);
      }
    }
  }
  __syncthreads();
  const int cl = g.L / 8;
  for (int i = threadIdx.x; i < RPB * cl; i += 256) {
    const unsigned lr = fdiv(i, dv.cl), row = row0 + lr;
    if (row >= rows) break;
    const int l0 = (i - lr * cl) * 8;
    const unsigned bn = fdiv(row, dv.H);
    const unsigned b = fdiv(bn, dv.Nq);
    const int n = bn - b * Nq;
    float v[8];
    const bool tokrow = !(g.has_cls && n == 0);
    const int p = tokrow ? n - g.has_cls : 0;
    const int t = fdiv(p, dv.plane), r = p - t * (g.qw * g.qh);
    const int hh = fdiv(r, dv.qw), w = r - hh * g.qw;
    const T* arow = aug + lr * g.A;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int l = l0 + e;
      const bool ok = tokrow && l < g.Lh + g.Lw + g.Lt;
      const int ax = (l >= g.Lh) + (l >= g.Lh + g.Lw);
      const int* tp = ax == 0 ? th : (ax == 1 ? tw : tt);
      const int off = ax == 0 ? hh * g.Lh + l : (ax == 1 ? w * g.Lw + (l - g.Lh) : t * g.Lt + (l - g.Lh - g.Lw));
      const int jj = tp[ok ? off : 0];
      const int j = (ok && jj >= 0) ? jj + (ax == 0 ? 0 : (ax == 1 ? g.kh : g.kh + g.kw)) : -1;
      v[e] = j >= 0 ? to_f32(arow[j]) * inv_scale : 0.0f;
    }
    store8<T>(dP + (int64_t)row * g.L + l0, v);
  }
}

// one-hot key augmentation: one thread = 8 columns of one (b, n, head) row (a 16-byte store), 32-bit index arithmetic
// with multiply-shift divisions (the first version spent ~100 instructions per ELEMENT on 64-bit div / mod).
template <typename T>
__global__ __launch_bounds__(256) void relpos_kaug_kernel(T* __restrict__ Kf, RelGeom g, RelDiv dv) {
  const int Nk = g.kt * g.kh * g.kw + g.has_cls;
  const int ca = g.A / 8;
  const unsigned total = (unsigned)g.B * Nk * g.H * ca;
  const unsigned id = blockIdx.x * 256u + threadIdx.x;
  if (id >= total) return;
  const unsigned row = fdiv(id, dv.ca);
  const int j0 = (id - row * ca) * 8;
  const unsigned bn = fdiv(row, dv.H);
  const int hd = row - bn * g.H;
  const unsigned b = fdiv(bn, dv.Nq);   // dv.Nq / plane / qw are built from the KEY grid for this kernel
  const int n = bn - b * Nk;
  float v[8];
  const bool tok = !(g.has_cls && n == 0);
  const int i = tok ? n - g.has_cls : 0;
  const int t = fdiv(i, dv.plane), r = i - t * (g.kw * g.kh);
  const int hh = fdiv(r, dv.qw), w = r - hh * g.kw;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int j = j0 + e;
    v[e] = (tok && (j == hh || j == g.kh + w || j == g.kh + g.kw + t)) ? 1.0f : 0.0f;
  }
  store8<T>(Kf + b * g.sb + (int64_t)n * g.sn + hd * g.sh + g.D + j0, v);
}

// =====================================================================================================
// Fused rel-pos kernels (bf16, head_dim 96): forward and backward work on tiles of 64 consecutive (clip, token, head) rows and keep
// the tile's slice of P / dP ([rows, L], L = 48 .. 240 against 22 .. 36 used entries per row; 385 MB each way in block 0) in LDS,
// so the tensor never exists in HBM; the products stay on the matrix cores with the arithmetic of the GEMM path.
// History of this section (measured, tools/relpos_bench.py): a VALU forward (v_dot2c_f32_bf16 dot products against only the used
// table rows) beat GEMM + gather for 32 augmentation columns but lost for 64 and was replaced by the MFMA tile kernel below,
// which wins everywhere (stage 2: 34 -> 19 us, block 1: 77 -> 50); a VALU backward with LDS float atomics for the table gradient
// was 6-30x slower than scatter + two GEMMs (ds_add_f32 retires about one lane per clock per CU).
// Table rows of idx_a: idx_a[coord][j] (the reference's integer-truncated distance), staged in LDS.
typedef __bf16 rbf2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct RowPos { int b, n, hd, t, hh, w; bool tok; };
__device__ __forceinline__ RowPos row_pos(unsigned row, const RelGeom& g, const RelDiv& dv, int Nq) {
  RowPos p;
  const unsigned bn = fdiv(row, dv.H);
  p.hd = row - bn * g.H;
  p.b = fdiv(bn, dv.Nq);
  p.n = bn - p.b * Nq;
  p.tok = !(g.has_cls && p.n == 0);
  const int i = p.tok ? p.n - g.has_cls : 0;
  p.t = fdiv(i, dv.plane);
  const int r = i - p.t * (g.qw * g.qh);
  p.hh = fdiv(r, dv.qw);
  p.w = r - p.hh * g.qw;
  return p;
}

// ------------------------------------------------------------------------------- fused backward on the matrix cores
// One kernel instead of scatter + two GEMMs, and no [rows, L] dP tensor in HBM.  A block walks tiles of 64 consecutive
// (clip, token, head) rows: it builds the tile's dP [64][L] in LDS exactly as the scatter kernel would write it (zeros, then the
// k_h + k_w + k_t augmentation gradients of every row at their table rows, scaled and rounded to bf16), then
//   dq [64][96]  = dP . Rcat + dQf[:, :96]      (3 waves, one 32-column slab each; tables resident in LDS),
//   dR [L][96]  += dP^T . q                      (all 4 waves; accumulators stay in registers across the block's tiles),
// both on v_mfma_f32_32x32x16_bf16 with fragments read straight from the LDS images (row-major dP serves product 1 as a
// k-contiguous operand and product 2 as a k-major operand through ds_read_tr16_b64).  Each block's dR goes to a workspace that
// relpos_fold_kernel sums.  Same arithmetic as the three-kernel path (bf16 dP, fp32 accumulation).
// (An earlier VALU version of this backward used LDS float atomics for the table gradient and was 6-30x slower than the GEMM path.)
__device__ __forceinline__ bf16x8 rfrag_kc(const char* base, int pitch, int r0, int s, int lane) {
  return *reinterpret_cast<const bf16x8*>(base + (r0 + (lane & 31)) * pitch + (16 * s + 8 * (lane >> 5)) * 2);
}
// LDS rows = k, columns = the m / n index (c0 + 32 of them), transposing reads
__device__ __forceinline__ bf16x8 rfrag_km(const char* base, int pitch, int c0, int s, int lane) {
  const int h = lane >> 5, i = lane & 15, q = i >> 2, pp = i & 3, g = (lane >> 4) & 1;
  bf16x8 f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int krow = 16 * s + 8 * h + 4 * half + q;
    const char* p = base + krow * pitch + (c0 + 16 * g + 4 * pp) * 2;
    const bf16x4 tb = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p)));
    f[4 * half + 0] = tb[0]; f[4 * half + 1] = tb[1]; f[4 * half + 2] = tb[2]; f[4 * half + 3] = tb[3];
  }
  return f;
}

#ifdef RB_PROF
__device__ unsigned long long g_rbprof[12];
#define RBT() __builtin_amdgcn_s_memtime()
#define RBP(i, t0) do { if (tid == 0) atomicAdd(&g_rbprof[i], RBT() - (t0)); } while (0)
#else
#define RBT() 0ull
#define RBP(i, t0) do { } while (0)
#endif
// LN: the per-head LayerNorm backward of the pooled queries (Pool.norm_act, the op autograd runs next on dq) in the epilogue:
// what is stored is dconv = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dq * gamma, xhat = (conv - mean) * rstd, with dq
// rounded to bf16 first exactly as the two-kernel path stores it; the block's dgamma / dbeta partials follow its table gradient
// in the workspace row.  Saves the dq round trip through HBM and two launches per block (ln_bwd + ln_reduce).
struct RelLn { const bf16_t* conv; const float* mean; const float* rstd; const float* gamma; };
template <int NMT, bool LN>   // NMT: 32-row slabs of dR per wave: ceil(LPp / 32 / 4)
__global__ __launch_bounds__(256, NMT == 1 ? 2 : 1) void relpos_bwd_mfma_kernel(const bf16_t* __restrict__ dQf, const bf16_t* __restrict__ Qf,
                                                              bf16_t* __restrict__ dq, const bf16_t* __restrict__ Rcat,
                                                              float* __restrict__ ws, const int* __restrict__ idx_h,
                                                              const int* __restrict__ idx_w, const int* __restrict__ idx_t,
                                                              RelGeom g, RelDiv dv, float inv_scale, unsigned rows, int LPp,
                                                              int dpp, unsigned ntiles, RelLn ln) {
  extern __shared__ __attribute__((aligned(16))) char rlds[];
  char* Rs = rlds;                       // [LPp][192 B]   stacked tables, zero rows past g.L
  char* dP = Rs + LPp * 192;             // [64][dpp]
  char* Qs = dP + 64 * dpp;              // [64][192 B]   q rows of the tile
  constexpr int RZP = 208;               // residual tile row pitch (192 + 16: the epilogue's 8-byte reads stay conflict-light)
  char* Rz = Qs + 64 * 192;              // [64][RZP]     dQf[:, :96] rows of the tile
  float* lnp = reinterpret_cast<float*>(Rz + 64 * RZP);   // LN: [3][64][2]  per-wave (32-column) row sums of g and g * xhat
  float* gml = lnp + 3 * 64 * 2;                          // LN: [96] gamma
  int* ih = reinterpret_cast<int*>(Rz + 64 * RZP + (LN ? 3 * 64 * 8 + 96 * 4 : 0));
  int* iw = ih + g.qh * g.kh;
  int* it = iw + g.qw * g.kw;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const unsigned long long tp0 = RBT();
  // LN: this lane's 16 epilogue columns are 32 * wave + 8 * g4 + 4 * h + e
  // LN parameter gradients: after each tile a butterfly over the 32 rows a half-wave holds leaves ONE of the half-wave's 32 column
  // sums (16 of dy * xhat, 16 of dy) in each lane; that lane accumulates it over the block's tiles
  float pgrad = 0.0f;
  if constexpr (LN) {
    if (tid < 96) gml[tid] = ln.gamma[tid];
  }
  for (int i = tid; i < LPp * 12; i += 256) {
    const int l = i / 12;
    const u32x4 z = {0u, 0u, 0u, 0u};
    reinterpret_cast<u32x4*>(Rs)[i] = l < g.L ? reinterpret_cast<const u32x4*>(Rcat)[i] : z;
  }
  for (int i = tid; i < g.qh * g.kh; i += 256) ih[i] = idx_h[i];
  for (int i = tid; i < g.qw * g.kw; i += 256) iw[i] = idx_w[i] + g.Lh;
  for (int i = tid; i < g.qt * g.kt; i += 256) it[i] = idx_t[i] + g.Lh + g.Lw;
  const int Nq = g.qt * g.qh * g.qw + g.has_cls;
  const int areal = g.kh + g.kw + g.kt;
  f32x16 dR[NMT][3];
#pragma unroll
  for (int i = 0; i < NMT; ++i)
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) dR[i][n][e] = 0.0f;
  const int r = tid >> 2, part = tid & 3;     // this thread's tile row and quarter
  const int zunits = 64 * dpp / 16;
  const int apq = g.A >> 2;                   // this thread's share of a row's augmentation gradients: A/4 = 8 or 16 columns
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  // The global loads of tile i+1 (q row, residual row, augmentation gradients: 8 x 16 bytes per thread) are issued at the START
  // of tile i, right after tile i's registers went to LDS, so a whole tile of work covers their latency (in-kernel timers: with
  // the loads issued mid-tile, and the residual fetched in the epilogue, three exposed memory waits were 90 % of a tile).
  struct Pre { u32x4 q[3], z[3], g[2]; RowPos p; bool live; };
  Pre preA, preB;
  auto prefetch = [&](Pre& pr, unsigned tile) {
    const unsigned row = tile * 64u + r;
    pr.live = tile < ntiles && row < rows;
    pr.p = row_pos(pr.live ? row : 0u, g, dv, Nq);
    const int64_t off = pr.p.b * g.sb + (int64_t)pr.p.n * g.sn + pr.p.hd * g.sh;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pr.q[c] = pr.live ? reinterpret_cast<const u32x4*>(Qf + off)[part * 3 + c] : z4;
      pr.z[c] = pr.live ? reinterpret_cast<const u32x4*>(dQf + off)[part * 3 + c] : z4;
    }
    pr.g[0] = pr.live ? reinterpret_cast<const u32x4*>(dQf + off + g.D + part * apq)[0] : z4;
    pr.g[1] = (pr.live && apq > 8) ? reinterpret_cast<const u32x4*>(dQf + off + g.D + part * apq)[1] : z4;
  };
  int64_t pend = -1;   // !LN: tile whose finished dq sits in Rz
  auto copy_out = [&]() {
    if (pend < 0) return;
    const int64_t orow = pend * 64 + r;
    if (orow < (int64_t)rows) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        reinterpret_cast<u32x4*>(dq + orow * g.D)[part * 3 + c] = reinterpret_cast<const u32x4*>(Rz + r * RZP)[part * 3 + c];
    }
  };
  auto do_tile = [&](Pre& pr, unsigned tile) {
    const unsigned long long t1 = RBT();
    __syncthreads();   // the previous tile's fragment reads are done (and, first time, the tables are staged)
    RBP(1, t1);
    const unsigned long long t2 = RBT();
    const RowPos p = pr.p;
    const bool live = pr.live;
    const u32x4 gr[2] = {pr.g[0], pr.g[1]};
    for (int i = tid; i < zunits; i += 256) reinterpret_cast<u32x4*>(dP)[i] = z4;
    if constexpr (!LN) copy_out();   // the previous tile's dq rows (each thread re-reads the three pieces it is about to overwrite)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      reinterpret_cast<u32x4*>(Qs + r * 192)[part * 3 + c] = pr.q[c];
      reinterpret_cast<u32x4*>(Rz + r * RZP)[part * 3 + c] = pr.z[c];
    }
    pend = (int64_t)tile;
    prefetch(pr, tile + (NMT == 2 ? 2 : 1) * gridDim.x);
    __syncthreads();   // zeros are in place
    RBP(2, t2);
    const unsigned long long t3 = RBT();
    if (live && p.tok) {
      // the column arithmetic below is tile-invariant; computed here each tile (a few VALU ops per column) rather than hoisted
      // out of the tile loop, where it held ~35 registers across the whole kernel
      int jb = part * apq;
      asm volatile("" : "+v"(jb));
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e >= apq) break;
        const int j = jb + e;
        if (j < areal) {
          const int ax = (j >= g.kh) + (j >= g.kh + g.kw);
          const int l = ax == 0 ? ih[p.hh * g.kh + j] : (ax == 1 ? iw[p.w * g.kw + (j - g.kh)] : it[p.t * g.kt + (j - g.kh - g.kw)]);
          const unsigned v = gr[e >> 3][(e >> 1) & 3];
          const float gf = (e & 1) ? __builtin_bit_cast(float, v & 0xffff0000u) : __builtin_bit_cast(float, v << 16);
          *reinterpret_cast<bf16_t*>(dP + r * dpp + l * 2) = (bf16_t)(gf * inv_scale);
        }
      }
    }
    __syncthreads();   // dP, Qs and the residual tile are complete
    RBP(3, t3);
    // LN: the epilogue lanes fetch their own pieces of the tile's conv rows and the rows' statistics here; both products run
    // before the first use (fetched before the scatter they cost 18 spilled registers)
    u32x2 cz[LN ? 2 : 1][LN ? 4 : 1];
    float mu[2] = {0.0f, 0.0f}, rsd[2] = {0.0f, 0.0f};
    if constexpr (LN) {
      if (wave < 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned orow = tile * 64u + 32 * i + (lane & 31);
          const bool ok = orow < rows;
          const bf16_t* src = ln.conv + (int64_t)(ok ? orow : 0u) * g.D + 32 * wave + 4 * h;
          const u32x2 z2 = {0u, 0u};
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) cz[i][g4] = ok ? *reinterpret_cast<const u32x2*>(src + 8 * g4) : z2;
          mu[i] = ok ? ln.mean[orow] : 0.0f;
          rsd[i] = ok ? ln.rstd[orow] : 0.0f;
        }
      }
    }

    const unsigned long long t4 = RBT();
    // ---- product 2: dR[l][c] += sum_rows dP[row][l] q[row][c]   (k = the 64 tile rows)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 fq[3];
#pragma unroll
      for (int n = 0; n < 3; ++n) fq[n] = rfrag_km(Qs, 192, 32 * n, s, lane);
#pragma unroll
      for (int i = 0; i < NMT; ++i) {
        const int mt = wave + 4 * i;
        if (32 * mt < LPp) {   // wave-uniform
          const bf16x8 fa = rfrag_km(dP, dpp, 32 * mt, s, lane);
#pragma unroll
          for (int n = 0; n < 3; ++n) dR[i][n] = mma16(fa, fq[n], dR[i][n]);
        }
      }
    }
    RBP(4, t4);
    const unsigned long long t5 = RBT();
    // ---- product 1: dq[row][c] = sum_l dP[row][l] Rcat[l][c] + residual   (waves 0..2: 32 columns each; C^T form, lanes = rows)
    f32x16 acc[2];
    if (wave < 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
      for (int s = 0; s < LPp / 16; ++s) {
        const bf16x8 fb = rfrag_km(Rs, 192, 32 * wave, s, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = mma16(fb, rfrag_kc(dP, dpp, 32 * i, s, lane), acc[i]);
      }
      RBP(5, t5);
      if constexpr (!LN) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned orow = tile * 64u + 32 * i + (lane & 31);
          if (orow < rows) {
            const char* res = Rz + (32 * i + (lane & 31)) * RZP + (32 * wave + 4 * h) * 2;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const bf16x4 rv = *reinterpret_cast<const bf16x4*>(res + 16 * g4);
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[i][4 * g4 + e] + (float)rv[e]);
              // the finished value replaces its residual in LDS; the tile leaves at the start of the next one as whole rows
              // (48 bytes per thread, 4 threads per 192-byte row) instead of 8-byte pieces of 32 different rows per store
              *reinterpret_cast<bf16x4*>(const_cast<char*>(res) + 16 * g4) = o;
            }
          }
        }
      } else {
        // phase 1: dq (rounded as the two-kernel path stores it) stays in acc; row sums over this wave's 32 columns -> LDS
        float pv[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) pv[k] = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const char* res = Rz + (32 * i + (lane & 31)) * RZP + (32 * wave + 4 * h) * 2;
          float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(res + 16 * g4);
            const bf16x4 cv = __builtin_bit_cast(bf16x4, cz[i][g4]);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gml + 32 * wave + 8 * g4 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = 4 * g4 + e;
              const float d = (float)(bf16_t)(acc[i][k] + (float)rv[e]);
              const float xh = ((float)cv[e] - mu[i]) * rsd[i];
              const float gg = d * gm[e];
              acc[i][k] = d;
              s1 += gg;
              s2 += gg * xh;
              pv[k] += d * xh;
              pv[16 + k] += d;
            }
          }
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 32);
          if (h == 0) *reinterpret_cast<f32x2*>(lnp + ((wave * 64) + 32 * i + (lane & 31)) * 2) = f32x2{s1, s2};
        }
        // reduce-scatter over the half-wave's 32 lanes: 16 + 8 + 4 + 2 + 1 exchanges; lane l ends with the sum of value l & 31
#pragma unroll
        for (int w = 16; w > 0; w >>= 1) {
          const bool up = (lane & w) != 0;
#pragma unroll
          for (int j = 0; j < w; ++j) {
            const float keep = up ? pv[w + j] : pv[j], send = up ? pv[j] : pv[w + j];
            pv[j] = keep + __shfl_xor(send, w);
          }
        }
        pgrad += pv[0];
      }
    }
    RBP(7, t5);
    if constexpr (LN) {
      const unsigned long long t8 = RBT();
      __syncthreads();   // the three column slabs' row sums are in LDS (next written after the next tile's barriers)
      RBP(8, t8);
      if (wave < 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = 32 * i + (lane & 31);
          const unsigned orow = tile * 64u + row;
          if (orow < rows) {
            const f32x2 p0 = *reinterpret_cast<const f32x2*>(lnp + row * 2), p1 = *reinterpret_cast<const f32x2*>(lnp + (64 + row) * 2),
                        p2 = *reinterpret_cast<const f32x2*>(lnp + (128 + row) * 2);
            const float m1 = (p0[0] + p1[0] + p2[0]) / 96.0f, m2 = (p0[1] + p1[1] + p2[1]) / 96.0f;
            bf16_t* dst = dq + (int64_t)orow * g.D + 32 * wave + 4 * h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const bf16x4 cv = __builtin_bit_cast(bf16x4, cz[i][g4]);
              const f32x4 gm = *reinterpret_cast<const f32x4*>(gml + 32 * wave + 8 * g4 + 4 * h);
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int k = 4 * g4 + e;
                const float xh = ((float)cv[e] - mu[i]) * rsd[i];
                o[e] = (bf16_t)(rsd[i] * (acc[i][k] * gm[e] - m1 - xh * m2));
              }
              *reinterpret_cast<bf16x4*>(dst + 8 * g4) = o;
            }
          }
        }
      }
    }
    RBP(9, t5);
  };
  // NMT == 2 (the 240-row tables of block 0: one block per CU anyway): two register sets, two tiles of loads in flight
  // (349 -> 269 us).  NMT == 1: one set — the second costs 130 VGPRs and the second block per CU (stage 2: 57 -> 64 us).
  constexpr int DIST = NMT == 2 ? 2 : 1;
  prefetch(preA, blockIdx.x);
  if constexpr (DIST == 2) prefetch(preB, blockIdx.x + gridDim.x);
  RBP(0, tp0);
  for (unsigned tile = blockIdx.x; tile < ntiles; tile += DIST * gridDim.x) {
    do_tile(preA, tile);
    if constexpr (DIST == 2) {
      if (tile + gridDim.x < ntiles) do_tile(preB, tile + gridDim.x);   // block-uniform
    }
  }
  if constexpr (!LN) {
    __syncthreads();   // the last tile's epilogue writes
    copy_out();
  }
  const unsigned long long t6 = RBT();
  // ---- this block's table gradient -> workspace [block][g.L][96] (+ [2][96] LayerNorm parameter partials with LN)
  float* out = ws + (int64_t)blockIdx.x * (g.L * 96 + (LN ? 192 : 0));
  if constexpr (LN) {
    if (wave < 3) {
      const int k = lane & 15;   // value lane & 31: k of dgamma below 16, k of dbeta from 16
      out[g.L * 96 + ((lane & 16) ? 96 : 0) + 32 * wave + 8 * (k >> 2) + 4 * h + (k & 3)] = pgrad;
    }
  }
#pragma unroll
  for (int i = 0; i < NMT; ++i) {
    const int mt = wave + 4 * i;
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = 32 * mt + acc_row(e, h);
        if (m < g.L) out[m * 96 + 32 * n + (lane & 31)] = dR[i][n][e];
      }
  }
  RBP(6, t6);
}

// Forward on the matrix cores, same tiling: P = q Rcat^T for a 64-row tile lands in LDS (bf16, as the GEMM would write it), the
// tile's threads pick each row's k_h + k_w + k_t entries, scale and store the augmentation columns.  Bit-identical to dcc_gemm +
// dcc_relpos_gather (same roundings) without the [rows, L] tensor; serves every augmentation width.
template <int NNT>   // 32-row slabs of the tables per wave: ceil(LPp / 32 / 4)
__global__ __launch_bounds__(256, NNT == 1 ? 2 : 1) void relpos_fwd_mfma_kernel(bf16_t* __restrict__ Qf, const bf16_t* __restrict__ Rcat,
                                                                                const int* __restrict__ idx_h, const int* __restrict__ idx_w,
                                                                                const int* __restrict__ idx_t, RelGeom g, RelDiv dv,
                                                                                float inv_scale, unsigned rows, int LPp, int ppitch,
                                                                                unsigned ntiles) {
  extern __shared__ __attribute__((aligned(16))) char rlds[];
  constexpr int QP = 208;                // q tile row pitch: 13 16-byte units (odd: the k-contiguous fragment reads spread over the banks)
  char* Rs = rlds;                       // [LPp][QP]     stacked tables (k-contiguous B operand), zero rows past g.L
  char* Qs = Rs + LPp * QP;              // [64][QP]
  char* Ps = Qs + 64 * QP;               // [64][ppitch]  P tile, bf16
  int* ih = reinterpret_cast<int*>(Ps + 64 * ppitch);
  int* iw = ih + g.qh * g.kh;
  int* it = iw + g.qw * g.kw;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  for (int i = tid; i < LPp * 12; i += 256) {
    const int l = i / 12, c = i - l * 12;
    *reinterpret_cast<u32x4*>(Rs + l * QP + c * 16) = l < g.L ? reinterpret_cast<const u32x4*>(Rcat)[i] : z4;
  }
  for (int i = tid; i < g.qh * g.kh; i += 256) ih[i] = idx_h[i];
  for (int i = tid; i < g.qw * g.kw; i += 256) iw[i] = idx_w[i] + g.Lh;
  for (int i = tid; i < g.qt * g.kt; i += 256) it[i] = idx_t[i] + g.Lh + g.Lw;
  const int Nq = g.qt * g.qh * g.qw + g.has_cls;
  const int areal = g.kh + g.kw + g.kt;
  const int r = tid >> 2, part = tid & 3, apq = g.A >> 2;
  u32x4 pq[3];
  RowPos pp_;
  bool plive;
  auto prefetch = [&](unsigned tile) {
    const unsigned row = tile * 64u + r;
    plive = tile < ntiles && row < rows;
    pp_ = row_pos(plive ? row : 0u, g, dv, Nq);
    const int64_t off = pp_.b * g.sb + (int64_t)pp_.n * g.sn + pp_.hd * g.sh;
#pragma unroll
    for (int c = 0; c < 3; ++c) pq[c] = plive ? reinterpret_cast<const u32x4*>(Qf + off)[part * 3 + c] : z4;
  };
  prefetch(blockIdx.x);
  for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();   // the previous tile's P reads are done (and, first time, the tables are staged)
    const RowPos p = pp_;
    const bool live = plive;
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<u32x4*>(Qs + r * QP + (part * 3 + c) * 16) = pq[c];
    prefetch(tile + gridDim.x);
    __syncthreads();   // the q tile is complete
    // P^T form: lanes = rows, a lane's registers = 4 consecutive table rows -> 8-byte LDS writes of bf16
#pragma unroll
    for (int i = 0; i < NNT; ++i) {
      const int nt = wave + 4 * i;
      if (32 * nt < LPp) {   // wave-uniform
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          const bf16x8 fb = rfrag_kc(Rs, QP, 32 * nt, s, lane);
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = mma16(fb, rfrag_kc(Qs, QP, 32 * m, s, lane), acc[m]);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          char* prow = Ps + (32 * m + (lane & 31)) * ppitch + (32 * nt + 4 * h) * 2;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)acc[m][4 * g4 + e];
            *reinterpret_cast<bf16x4*>(prow + 16 * g4) = o;
          }
        }
      }
    }
    __syncthreads();   // the P tile is complete
    if (live) {        // thread (row, quarter): augmentation columns quarter * A/4 .. of its row
      unsigned out[8];
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        if (e >= apq) break;
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = part * apq + e + u;
          v[u] = 0.0f;
          if (p.tok && j < areal) {
            const int ax = (j >= g.kh) + (j >= g.kh + g.kw);
            const int l = ax == 0 ? ih[p.hh * g.kh + j] : (ax == 1 ? iw[p.w * g.kw + (j - g.kh)] : it[p.t * g.kt + (j - g.kh - g.kw)]);
            v[u] = (float)*reinterpret_cast<const bf16_t*>(Ps + r * ppitch + l * 2) * inv_scale;
          }
        }
        const rbf2 t = {(bf16_t)v[0], (bf16_t)v[1]};
        out[e >> 1] = __builtin_bit_cast(unsigned, t);
      }
      u32x4* dst = reinterpret_cast<u32x4*>(Qf + p.b * g.sb + (int64_t)p.n * g.sn + p.hd * g.sh + g.D + part * apq);
      dst[0] = u32x4{out[0], out[1], out[2], out[3]};
      if (apq > 8) dst[1] = u32x4{out[4], out[5], out[6], out[7]};
    }
  }
}

// out[i] += sum_b ws[b][i]  (i over n4 float4 columns): block = 64 columns x 4 row lanes, gridDim.y row slices, the four lanes
// fold through LDS and every slice adds its sum with one atomic per element.
// columns [0, nr4) of a block's workspace row go to out, the next 24 to dgamma, the last 24 to dbeta (n4 = nr4 or nr4 + 48)
__global__ __launch_bounds__(256) void relpos_fold_kernel(const float* __restrict__ ws, float* __restrict__ out, int nb, int n4, int nr4,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ f32x4 red[256];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
  if (c < n4)
    for (int b = blockIdx.y * 4 + rl; b < nb; b += gridDim.y * 4) s += reinterpret_cast<const f32x4*>(ws)[(int64_t)b * n4 + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < n4) {
    s = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
    float* dst = c < nr4 ? out + 4 * c : (c < nr4 + 24 ? dgamma + 4 * (c - nr4) : dbeta + 4 * (c - nr4 - 24));
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dst + e, s[e]);
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
  const RelDiv dv{mkdiv(g.H), mkdiv(g.qt * g.qh * g.qw + g.has_cls), mkdiv(g.qw * g.qh), mkdiv(g.qw), mkdiv(g.L / 8), mkdiv(g.A / 8)};
  if (dtype == DCC_BF16) hipLaunchKernelGGL((relpos_gather_kernel<bf16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (bf16_t*)Qf, (const bf16_t*)P, idx_h, idx_w, idx_t, g, dv, 1.0f / scale);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((relpos_gather_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (float*)Qf, (const float*)P, idx_h, idx_w, idx_t, g, dv, 1.0f / scale);
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
  {
    const int esz = dtype == DCC_BF16 ? 2 : 4;
    const int ntab = g.qh * g.Lh + g.qw * g.Lw + g.qt * g.Lt;
    const bool stage = ntab <= 2048;
    const size_t lds = (stage ? (size_t)((ntab + 3) & ~3) * 4 : 0) + (size_t)RPB * g.A * esz;
    const int64_t rows = (int64_t)g.B * (g.qt * g.qh * g.qw + g.has_cls) * g.H;
    static const bool use_lds = !getenv("DCC_RELPOS_NO_LDS");
    if (use_lds && g.kh + g.kw + g.kt <= g.A && rows < (1LL << 31) && (dtype == DCC_BF16 || dtype == DCC_F32)) {
      const dim3 grid((unsigned)((rows + RPB - 1) / RPB));
      RelDiv dv{mkdiv(g.H), mkdiv(g.qt * g.qh * g.qw + g.has_cls), mkdiv(g.qw * g.qh), mkdiv(g.qw), mkdiv(g.L / 8), mkdiv(g.A / 8)};
#define RPS(T_, ST_) hipLaunchKernelGGL((relpos_scatter_lds_kernel<T_, ST_>), grid, dim3(256), lds, st, (const T_*)dQf, (T_*)dP, inv_h, inv_w, inv_t, g, dv, 1.0f / scale)
      if (dtype == DCC_BF16) { if (stage) RPS(bf16_t, true); else RPS(bf16_t, false); }
      else { if (stage) RPS(float, true); else RPS(float, false); }
#undef RPS
      return dcc_check_launch();
    }
  }
  if (dtype == DCC_BF16) hipLaunchKernelGGL((relpos_scatter_kernel<bf16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)dQf, (bf16_t*)dP, inv_h, inv_w, inv_t, g, 1.0f / scale);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((relpos_scatter_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)dQf, (float*)dP, inv_h, inv_w, inv_t, g, 1.0f / scale);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

// dims as above (k grid used); strides of Kf
extern "C" int dcc_relpos_kaug(void* Kf, const int* dims, const int64_t* strides, int dtype, void* stream) {
  if (!Kf || !dims || !strides) return DCC_ERR_ARG;
  RelGeom g = mk(dims, strides);
  if (g.A % 8) return DCC_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)g.B * (g.kt * g.kh * g.kw + g.has_cls) * g.H * (g.A / 8);
  if (total >= (1LL << 31)) return DCC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const RelDiv dv{mkdiv(g.H), mkdiv(g.kt * g.kh * g.kw + g.has_cls), mkdiv(g.kw * g.kh), mkdiv(g.kw), mkdiv(1), mkdiv(g.A / 8)};
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == DCC_BF16) hipLaunchKernelGGL((relpos_kaug_kernel<bf16_t>), grid, dim3(256), 0, st, (bf16_t*)Kf, g, dv);
  else if (dtype == DCC_F32) hipLaunchKernelGGL((relpos_kaug_kernel<float>), grid, dim3(256), 0, st, (float*)Kf, g, dv);
  else return DCC_ERR_UNSUPPORTED;
  return dcc_check_launch();
}

// Fused backward: dq [B*Nq*H][D] = dQf[:, :D] + (dQf[:, D:] / scale) against the table rows; dRcat [L][D] fp32 is ACCUMULATED
// into (the gradient of the three stacked tables); ws: scratch of max_blocks * L * D floats for the per-block partials (no
// initialisation needed).  = dcc_relpos_scatter + dcc_gemm (dP Rcat + residual) + dcc_gemm (dP^T q).
// With ln: what is stored is the LayerNorm backward of dq (see the kernel), dgamma / dbeta [D] fp32 accumulated into, and
// the workspace holds max_blocks * (L + 2) * D floats.
static int relpos_bwd_launch(const void* dQf, const void* Qf, void* dq, const void* Rcat, float* dRcat, float* ws, int max_blocks,
                             const int* idx_h, const int* idx_w, const int* idx_t, const int* dims, const int64_t* strides,
                             float scale, int dtype, void* stream, const RelLn* lnp, float* dgamma, float* dbeta) {
  if (!dQf || !Qf || !dq || !Rcat || !dRcat || !ws || !idx_h || !idx_w || !idx_t || !dims || !strides || max_blocks < 1) return DCC_ERR_ARG;
  if (lnp && (!lnp->conv || !lnp->mean || !lnp->rstd || !lnp->gamma || !dgamma || !dbeta)) return DCC_ERR_ARG;
  if (dtype != DCC_BF16) return DCC_ERR_UNSUPPORTED;
  RelGeom g = mk(dims, strides);
  if (g.D != 96 || (g.A != 32 && g.A != 64) || g.A < g.kh + g.kw + g.kt || g.L < g.Lh + g.Lw + g.Lt || (g.L % 8)) return DCC_ERR_UNSUPPORTED;
  if ((g.sb % 8) || (g.sn % 8) || (g.sh % 8) || (((uintptr_t)dQf) & 15) || (((uintptr_t)Qf) & 15) || (((uintptr_t)dq) & 7) ||
      (((uintptr_t)Rcat) & 15) || (((uintptr_t)ws) & 15))
    return DCC_ERR_UNSUPPORTED;
  if (lnp && ((((uintptr_t)lnp->conv) & 7) || (((uintptr_t)dgamma) & 3) || (((uintptr_t)dbeta) & 3))) return DCC_ERR_UNSUPPORTED;
  const int64_t rows = (int64_t)g.B * (g.qt * g.qh * g.qw + g.has_cls) * g.H;
  const int LPp = (g.L + 31) & ~31;
  if (LPp > 256 || rows >= (1LL << 31)) return DCC_ERR_UNSUPPORTED;
  const int dpp = LPp * 2 + (((LPp / 8) & 1) ? 0 : 16);   // row pitch of dP: an odd number of 16-byte units
  const size_t lds = (size_t)LPp * 192 + (size_t)64 * dpp + 64 * 192 + 64 * 208 + (lnp ? 3 * 64 * 8 + 96 * 4 : 0) +
                     (size_t)(g.qh * g.kh + g.qw * g.kw + g.qt * g.kt) * 4;
  if (lds > 150 * 1024) return DCC_ERR_UNSUPPORTED;
  const int64_t ntiles = (rows + 63) / 64;
  const int per_cu = (int)((160 * 1024) / lds);
  int64_t nb = 256LL * (per_cu > 3 ? 3 : per_cu);   // persistent: the tables are staged once per block
  if (nb > ntiles) nb = ntiles;
  if (nb > max_blocks) nb = max_blocks;
  hipStream_t st = (hipStream_t)stream;
  const RelDiv dv{mkdiv(g.H), mkdiv(g.qt * g.qh * g.qw + g.has_cls), mkdiv(g.qw * g.qh), mkdiv(g.qw), mkdiv(1), mkdiv(1)};
  auto k1 = relpos_bwd_mfma_kernel<1, false>;
  auto k2 = relpos_bwd_mfma_kernel<2, false>;
  auto k1l = relpos_bwd_mfma_kernel<1, true>;
  auto k2l = relpos_bwd_mfma_kernel<2, true>;
  static unsigned long long configured = 0;
  const unsigned long long devbit = dcc_dev_bit();
  if (!(configured & devbit)) {
    for (const void* f : {reinterpret_cast<const void*>(k1), reinterpret_cast<const void*>(k2), reinterpret_cast<const void*>(k1l),
                          reinterpret_cast<const void*>(k2l)})
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return DCC_ERR_LAUNCH;
    configured |= devbit;
  }
  const RelLn ln = lnp ? *lnp : RelLn{nullptr, nullptr, nullptr, nullptr};
  auto kern = lnp ? (LPp <= 128 ? k1l : k2l) : (LPp <= 128 ? k1 : k2);
  hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, st, (const bf16_t*)dQf, (const bf16_t*)Qf, (bf16_t*)dq, (const bf16_t*)Rcat, ws,
                     idx_h, idx_w, idx_t, g, dv, 1.0f / scale, (unsigned)rows, LPp, dpp, (unsigned)ntiles, ln);
  int rc = dcc_check_launch();
  if (rc) return rc;
  const int nr4 = g.L * 24, n4 = nr4 + (lnp ? 48 : 0);
  hipLaunchKernelGGL(relpos_fold_kernel, dim3((n4 + 63) / 64, 8), dim3(256), 0, st, ws, dRcat, (int)nb, n4, nr4, dgamma, dbeta);
  return dcc_check_launch();
}

extern "C" int dcc_relpos_aug_bwd(const void* dQf, const void* Qf, void* dq, const void* Rcat, float* dRcat, float* ws,
                                  int max_blocks, const int* idx_h, const int* idx_w, const int* idx_t, const int* dims,
                                  const int64_t* strides, float scale, int dtype, void* stream) {
  return relpos_bwd_launch(dQf, Qf, dq, Rcat, dRcat, ws, max_blocks, idx_h, idx_w, idx_t, dims, strides, scale, dtype, stream, nullptr,
                           nullptr, nullptr);
}

// dcc_relpos_aug_bwd followed by dcc_layernorm_bwd of its result against the pooled queries (conv [B*Nq*H][D] bf16 contiguous,
// mean / rstd [B*Nq*H], gamma [D]) in one kernel: dconv replaces dq, dgamma / dbeta are accumulated into.
extern "C" int dcc_relpos_aug_bwd_ln(const void* dQf, const void* Qf, void* dconv, const void* Rcat, float* dRcat, float* ws,
                                     int max_blocks, const int* idx_h, const int* idx_w, const int* idx_t, const int* dims,
                                     const int64_t* strides, float scale, const void* conv, const float* mean, const float* rstd,
                                     const float* gamma, float* dgamma, float* dbeta, int dtype, void* stream) {
  const RelLn ln{(const bf16_t*)conv, mean, rstd, gamma};
  return relpos_bwd_launch(dQf, Qf, dconv, Rcat, dRcat, ws, max_blocks, idx_h, idx_w, idx_t, dims, strides, scale, dtype, stream, &ln,
                           dgamma, dbeta);
}

#ifdef RB_PROF
extern "C" int dcc_relpos_prof_read(unsigned long long* out12, int reset) {
  unsigned long long h[12];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rbprof), sizeof(h)) != hipSuccess) return DCC_ERR_LAUNCH;
  for (int i = 0; i < 12; ++i) out12[i] = h[i];
  if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rbprof), z, sizeof(z)) != hipSuccess) return DCC_ERR_LAUNCH; }
  return DCC_OK;
}
#endif

// Forward on the matrix cores (bf16, D == 96, A in {32, 64}, L <= 256): Qf[row][D + j] exactly as dcc_gemm + dcc_relpos_gather
// produce it, without the [rows, L] tensor.
extern "C" int dcc_relpos_aug_fwd_mfma(void* Qf, const void* Rcat, const int* idx_h, const int* idx_w, const int* idx_t,
                                       const int* dims, const int64_t* strides, float scale, int dtype, void* stream) {
  if (!Qf || !Rcat || !idx_h || !idx_w || !idx_t || !dims || !strides) return DCC_ERR_ARG;
  if (dtype != DCC_BF16) return DCC_ERR_UNSUPPORTED;
  RelGeom g = mk(dims, strides);
  if (g.D != 96 || (g.A != 32 && g.A != 64) || g.A < g.kh + g.kw + g.kt || g.L < g.Lh + g.Lw + g.Lt || (g.L % 8)) return DCC_ERR_UNSUPPORTED;
  if ((g.sb % 8) || (g.sn % 8) || (g.sh % 8) || (((uintptr_t)Qf) & 15) || (((uintptr_t)Rcat) & 15)) return DCC_ERR_UNSUPPORTED;
  const int64_t rows = (int64_t)g.B * (g.qt * g.qh * g.qw + g.has_cls) * g.H;
  const int LPp = (g.L + 31) & ~31;
  if (LPp > 256 || rows >= (1LL << 31)) return DCC_ERR_UNSUPPORTED;
  const int ppitch = LPp * 2 + 16;
  const size_t lds = (size_t)LPp * 208 + 64 * 208 + (size_t)64 * ppitch + (size_t)(g.qh * g.kh + g.qw * g.kw + g.qt * g.kt) * 4;
  if (lds > 150 * 1024) return DCC_ERR_UNSUPPORTED;
  const int64_t ntiles = (rows + 63) / 64;
  const int per_cu = (int)((160 * 1024) / lds);
  int64_t nb = 256LL * (per_cu > 2 ? 2 : per_cu);
  if (nb > ntiles) nb = ntiles;
  hipStream_t st = (hipStream_t)stream;
  const RelDiv dv{mkdiv(g.H), mkdiv(g.qt * g.qh * g.qw + g.has_cls), mkdiv(g.qw * g.qh), mkdiv(g.qw), mkdiv(1), mkdiv(1)};
  auto k1 = relpos_fwd_mfma_kernel<1>;
  auto k2 = relpos_fwd_mfma_kernel<2>;
  static unsigned long long configured = 0;
  const unsigned long long devbit = dcc_dev_bit();
  if (!(configured & devbit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
      return DCC_ERR_LAUNCH;
    configured |= devbit;
  }
  if (LPp <= 128) hipLaunchKernelGGL(k1, dim3((unsigned)nb), dim3(256), lds, st, (bf16_t*)Qf, (const bf16_t*)Rcat, idx_h, idx_w, idx_t, g, dv, 1.0f / scale, (unsigned)rows, LPp, ppitch, (unsigned)ntiles);
  else hipLaunchKernelGGL(k2, dim3((unsigned)nb), dim3(256), lds, st, (bf16_t*)Qf, (const bf16_t*)Rcat, idx_h, idx_w, idx_t, g, dv, 1.0f / scale, (unsigned)rows, LPp, ppitch, (unsigned)ntiles);
  return dcc_check_launch();
}
