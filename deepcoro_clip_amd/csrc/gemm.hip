// MFMA GEMM with fused epilogues for gfx950.
//   C[m][n] = alpha * sum_k A(m,k) * B(n,k)  (+ bias[n]) (gelu / * gelu'(aux) / + R[m][n])
// A(m,k) = A_KM ? A[k*lda + m] : A[m*lda + k];   B(n,k) = B_KM ? B[k*ldb + n] : B[n*ldb + k]
// so that a Linear layer's forward (x W^T), data gradient (dy W) and weight gradient
// (dy^T x) are the three layout variants of one kernel, with no transposes through HBM.
//
// Tile 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles.
// Operands go global -> registers (16-byte coalesced chunks) -> LDS, double buffered.
// "k-contiguous" operands live in LDS as [row][32 k] with rows padded by 16 B (odd multiple of
// 16 B => conflict-free ds_read_b128 fragment reads); "k-major" operands live as [k][128] and
// are read with ds_read_b64_tr_b16 (bf16, 256-byte rows, XOR-swizzled 16-byte chunks) or plain
// ds_read_b32 (fp32).
#include "common.h"

#define GEMM_BIAS 1
#define GEMM_GELU 2
#define GEMM_SAVE_PRE 4
#define GEMM_RESID 8
#define GEMM_DGELU 16
#define GEMM_ACCUM 32

struct GemmParams {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const void* R;
  void* aux;
  int64_t lda, ldb, ldc, ldr, ldaux;
  int64_t sA, sB, sC;
  int M, N, K;
  int batch, splitk;
  float alpha;
  int flags;
  int unaligned;  // 1: operands break the 16-byte chunk rules -> element-wise guarded loads
};

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;

template <typename T, bool KM> struct OpTile {
  static constexpr int ES = sizeof(T);
  static constexpr int CH = 16 / ES;                       // elements per 16-byte chunk
  static constexpr int ROWB = KM ? 128 * ES : BK * ES + 16;  // LDS row pitch in bytes
  static constexpr int ROWS = KM ? BK : 128;
  static constexpr int BYTES = ROWS * ROWB;
  static constexpr int NCH = 128 * BK * ES / 16 / NT;      // chunks per thread
  static constexpr int CPR = KM ? 128 / CH : BK / CH;      // chunks per LDS row
};

__device__ __forceinline__ int swz_km(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }

// global -> registers. `row0` is the tile origin in the non-reduction dimension, `dim` its extent.
template <typename T, bool KM>
__device__ __forceinline__ void g2r(vec16_t<T>* regs, const T* base, int64_t ld, int row0, int dim, int k0, int kend,
                                    int tid, bool unaligned) {
  using OT = OpTile<T, KM>;
  using V = typename Vec16<T>::type;
#pragma unroll
  for (int i = 0; i < OT::NCH; ++i) {
    const int id = tid + NT * i;
    const int r = id / OT::CPR, c = id % OT::CPR;
    V v;
#pragma unroll
    for (int e = 0; e < OT::CH; ++e) v[e] = (T)0.0f;
    if (!KM) {
      const int gr = row0 + r, gk = k0 + c * OT::CH;
      if (gr < dim && gk < kend) {
        const T* src = base + (int64_t)gr * ld + gk;
        if (!unaligned) v = *reinterpret_cast<const V*>(src);
        else {
#pragma unroll
          for (int e = 0; e < OT::CH; ++e)
            if (gk + e < kend) v[e] = src[e];
        }
      }
    } else {
      const int gk = k0 + r, gc = row0 + c * OT::CH;
      if (gk < kend && gc < dim) {
        const T* src = base + (int64_t)gk * ld + gc;
        if (!unaligned) v = *reinterpret_cast<const V*>(src);
        else {
#pragma unroll
          for (int e = 0; e < OT::CH; ++e)
            if (gc + e < dim) v[e] = src[e];
        }
      }
    }
    regs[i] = v;
  }
}

template <typename T, bool KM>
__device__ __forceinline__ void r2s(const vec16_t<T>* regs, char* buf, int tid) {
  using OT = OpTile<T, KM>;
  using V = typename Vec16<T>::type;
#pragma unroll
  for (int i = 0; i < OT::NCH; ++i) {
    const int id = tid + NT * i;
    const int r = id / OT::CPR, c = id % OT::CPR;
    int off;
    if (!KM) off = r * OT::ROWB + c * 16;
    else if (sizeof(T) == 2) off = r * OT::ROWB + ((c ^ swz_km(r)) * 16);
    else off = r * OT::ROWB + c * 16;
    *reinterpret_cast<V*>(buf + off) = regs[i];
  }
}

// fragment of K-step s (16 k) for the 32 rows/cols starting at `t0` of the tile.
template <typename T, bool KM>
__device__ __forceinline__ typename Frag<T>::type load_frag(const char* buf, int t0, int s, int lane) {
  using OT = OpTile<T, KM>;
  const int r = lane & 31, h = lane >> 5;
  typename Frag<T>::type f;
  if constexpr (!KM) {
    const char* p = buf + (t0 + r) * OT::ROWB + (16 * s + 8 * h) * OT::ES;
    if constexpr (sizeof(T) == 2) {
      f = *reinterpret_cast<const bf16x8*>(p);
    } else {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 16);
      f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
      f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    }
  } else if constexpr (sizeof(T) == 2) {
    const int i = lane & 15, q = i >> 2, pp = i & 3, g = (lane >> 4) & 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int krow = 16 * s + 8 * h + 4 * half + q;
      const int chunk = ((t0 + 16 * g) >> 3) + (pp >> 1);
      const char* p = buf + krow * OT::ROWB + ((chunk ^ swz_km(krow)) * 16) + 8 * (pp & 1);
      const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p));
      const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
      f[4 * half + 0] = tb[0]; f[4 * half + 1] = tb[1]; f[4 * half + 2] = tb[2]; f[4 * half + 3] = tb[3];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      f[j] = *reinterpret_cast<const float*>(buf + (16 * s + 8 * h + j) * OT::ROWB + (t0 + r) * 4);
  }
  return f;
}

template <typename T, typename TC, bool A_KM, bool B_KM>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmParams p) {
  using OA = OpTile<T, A_KM>;
  using OB = OpTile<T, B_KM>;
  __shared__ __attribute__((aligned(16))) char smem[2 * (OA::BYTES + OB::BYTES)];
  constexpr int STAGE = OA::BYTES + OB::BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tilesN = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int bz = blockIdx.z / p.splitk, sk = blockIdx.z % p.splitk;
  const int m0 = tm * BM, n0 = tn * BN;

  const T* A = reinterpret_cast<const T*>(p.A) + (int64_t)bz * p.sA;
  const T* B = reinterpret_cast<const T*>(p.B) + (int64_t)bz * p.sB;

  const int ksteps = (p.K + BK - 1) / BK;
  const int per = (ksteps + p.splitk - 1) / p.splitk;
  const int ks0 = sk * per;
  int ks1 = ks0 + per; if (ks1 > ksteps) ks1 = ksteps;
  const int kend = (ks1 * BK < p.K) ? ks1 * BK : p.K;
  const int nk = ks1 - ks0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  typename Vec16<T>::type ra[OA::NCH], rb[OB::NCH];
  if (nk > 0) {
    g2r<T, A_KM>(ra, A, p.lda, m0, p.M, ks0 * BK, kend, tid, p.unaligned);
    g2r<T, B_KM>(rb, B, p.ldb, n0, p.N, ks0 * BK, kend, tid, p.unaligned);
    r2s<T, A_KM>(ra, smem, tid);
    r2s<T, B_KM>(rb, smem + OA::BYTES, tid);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    char* curA = smem + cur * STAGE;
    char* curB = curA + OA::BYTES;
    char* nxtA = smem + (cur ^ 1) * STAGE;
    char* nxtB = nxtA + OA::BYTES;
    if (kt + 1 < nk) {
      g2r<T, A_KM>(ra, A, p.lda, m0, p.M, (ks0 + kt + 1) * BK, kend, tid, p.unaligned);
      g2r<T, B_KM>(rb, B, p.ldb, n0, p.N, (ks0 + kt + 1) * BK, kend, tid, p.unaligned);
    }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      typename Frag<T>::type fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = load_frag<T, A_KM>(curA, wr * 64 + 32 * i, s, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = load_frag<T, B_KM>(curB, wc * 64 + 32 * j, s, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma16(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) {
      r2s<T, A_KM>(ra, nxtA, tid);
      r2s<T, B_KM>(rb, nxtB, tid);
    }
    __syncthreads();
  }

  // epilogue
  TC* C = reinterpret_cast<TC*>(p.C) + (int64_t)bz * p.sC;
  const TC* R = reinterpret_cast<const TC*>(p.R);
  const bool atomic = (p.splitk > 1) || (p.flags & GEMM_ACCUM);
  const int h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 64 + 32 * j + (lane & 31);
      if (col >= p.N) continue;
      float bv = 0.0f;
      if ((p.flags & GEMM_BIAS) && sk == 0) bv = p.bias[col];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wr * 64 + 32 * i + acc_row(e, h);
        if (row >= p.M) continue;
        float v = acc[i][j][e] * p.alpha + bv;
        if (p.flags & GEMM_SAVE_PRE) reinterpret_cast<TC*>(p.aux)[(int64_t)row * p.ldaux + col] = from_f32<TC>(v);
        if (p.flags & GEMM_GELU) v = gelu_f(v);
        if (p.flags & GEMM_DGELU) v *= dgelu_f(to_f32(reinterpret_cast<const T*>(p.aux)[(int64_t)row * p.ldaux + col]));
        if (p.flags & GEMM_RESID) v += to_f32(R[(int64_t)row * p.ldr + col]);
        TC* dst = C + (int64_t)row * p.ldc + col;
        if (atomic) {
          if constexpr (sizeof(TC) == 4) atomicAdd(reinterpret_cast<float*>(dst), v);
        } else {
          *dst = from_f32<TC>(v);
        }
      }
    }
  }
}

template <typename T, typename TC>
int launch_layout(const GemmParams& p, int a_km, int b_km, dim3 grid, hipStream_t st) {
  if (!a_km && !b_km) hipLaunchKernelGGL((gemm_kernel<T, TC, false, false>), grid, dim3(NT), 0, st, p);
  else if (!a_km && b_km) hipLaunchKernelGGL((gemm_kernel<T, TC, false, true>), grid, dim3(NT), 0, st, p);
  else if (a_km && !b_km) hipLaunchKernelGGL((gemm_kernel<T, TC, true, false>), grid, dim3(NT), 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<T, TC, true, true>), grid, dim3(NT), 0, st, p);
  return dcc_check_launch();
}

}  // namespace

// See include/dcc.h for the contract.
extern "C" int dcc_gemm(const void* A, const void* B, void* C, const float* bias, const void* R, void* aux,
                        int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int64_t ldaux, int M, int N, int K,
                        int batch, int64_t sA, int64_t sB, int64_t sC, int splitk, float alpha, int flags, int a_km,
                        int b_km, int in_dtype, int out_dtype, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || splitk <= 0) return DCC_ERR_ARG;
  if (!A || !B || !C) return DCC_ERR_ARG;
  const int ch = in_dtype == DCC_BF16 ? 8 : 4;
  // 16-byte chunk loads need chunk-multiple pitches/extents and 16-byte aligned bases; otherwise fall back to
  // element-wise guarded loads (tiny loss-side GEMMs with odd batch sizes land here)
  int unaligned = 0;
  if ((lda % ch) || (ldb % ch)) unaligned = 1;
  if ((!a_km && (K % ch)) || (a_km && (M % ch))) unaligned = 1;
  if ((!b_km && (K % ch)) || (b_km && (N % ch))) unaligned = 1;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) unaligned = 1;
  if ((sA % ch) || (sB % ch)) unaligned = 1;
  if ((splitk > 1 || (flags & GEMM_ACCUM)) && out_dtype != DCC_F32) return DCC_ERR_ARG;
  if ((flags & GEMM_BIAS) && !bias) return DCC_ERR_ARG;
  if ((flags & GEMM_RESID) && !R) return DCC_ERR_ARG;
  if ((flags & (GEMM_SAVE_PRE | GEMM_DGELU)) && !aux) return DCC_ERR_ARG;
  if (splitk > 1 && (flags & ~(GEMM_BIAS | GEMM_ACCUM))) return DCC_ERR_ARG;  // non-linear epilogues need the full sum
  GemmParams p{A, B, C, bias, R, aux, lda, ldb, ldc, ldr, ldaux, sA, sB, sC, M, N, K, batch, splitk, alpha, flags, unaligned};
  const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles > 0x7fffffffLL || (int64_t)batch * splitk > 65535) return DCC_ERR_ARG;
  dim3 grid((unsigned)tiles, 1, (unsigned)(batch * splitk));
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == DCC_BF16 && out_dtype == DCC_BF16) return launch_layout<bf16_t, bf16_t>(p, a_km, b_km, grid, st);
  if (in_dtype == DCC_BF16 && out_dtype == DCC_F32) return launch_layout<bf16_t, float>(p, a_km, b_km, grid, st);
  if (in_dtype == DCC_F32 && out_dtype == DCC_F32) return launch_layout<float, float>(p, a_km, b_km, grid, st);
  return DCC_ERR_UNSUPPORTED;
}
