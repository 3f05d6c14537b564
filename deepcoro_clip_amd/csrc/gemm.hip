// MFMA GEMM with fused epilogues for gfx950.
//   C[m][n] = alpha * sum_k A(m,k) * B(n,k)  (+ bias[n]) (gelu / * gelu'(aux) / + R[m][n])
// A(m,k) = A_KM ? A[k*lda + m] : A[m*lda + k];   B(n,k) = B_KM ? B[k*ldb + n] : B[n*ldb + k]
// so that a Linear layer's forward (x W^T), data gradient (dy W) and weight gradient
// (dy^T x) are the three layout variants of one kernel, with no transposes through HBM.
//
// Block tile (64*TM)^2 x BK (BK = 64 bf16 / 32 fp32), 256 threads = 4 waves (2x2), each wave a TM x TM grid of
// 32x32 MFMA tiles.  Operands go global -> registers (16-byte coalesced chunks, prefetched one K-tile ahead) ->
// LDS (double buffered, one barrier per K-tile).  "k-contiguous" operands live in LDS as [row][BK] with rows padded
// by 16 B (odd multiple of 16 B => conflict-free ds_read_b128 fragment reads); "k-major" operands live as [k][cols]
// and are read with ds_read_b64_tr_b16 (bf16, XOR-swizzled 16-byte chunks) or plain ds_read_b32 (fp32).
// Epilogue: accumulators are staged through LDS as fp32 and written as whole 16-byte chunks of complete rows
// (bias / GELU / GELU' / residual applied in that coalesced layout); split-K and accumulate modes use fp32 atomics.
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of tiles.
#include "common.h"

#include "gemm_params.h"
#include <stdlib.h>

namespace {

constexpr int NT = 256;

template <typename T, bool KM, int TM, int BKT> struct OpTile {
  static constexpr int ES = sizeof(T);
  static constexpr int BK = BKT;
  static constexpr int W = 64 * TM;                         // tile extent in the non-reduction dimension
  static constexpr int CH = 16 / ES;                        // elements per 16-byte chunk
  static constexpr int ROWB = KM ? W * ES : BK * ES + 16;   // LDS row pitch in bytes
  static constexpr int ROWS = KM ? BK : W;
  static constexpr int BYTES = ROWS * ROWB;
  static constexpr int NCH = W * BK * ES / 16 / NT;         // chunks per thread
  static constexpr int CPR = KM ? W / CH : BK / CH;         // chunks per LDS row
};

// XOR swizzle of the 16-byte chunk index for k-major bf16 tiles (conflict-free ds_read_b64_tr_b16):
// 256-byte rows (W=128): 4 rows of a read block land in 4 different 64-byte quarters of the bank row;
// 128-byte rows (W=64): rows 2,3 of a block use the other 64-byte half.
template <int W> __device__ __forceinline__ int swz_km(int krow) {
  if constexpr (W == 128) return ((krow & 3) << 2) | ((krow >> 2) & 3);
  else return ((krow >> 1) & 1) << 2;
}

// global -> registers.  `base` points at the tile origin (row0 folded in), `nrem` = extent left in the
// non-reduction dimension from there.  ALIGNED path: branch-free raw buffer loads (the descriptor's range check
// keeps tile rows past the end of the allocation from faulting) + a select for the chunks that must read as zero;
// straight-line loads let hipcc keep several K-tiles in flight behind counted vmcnt waits.
template <typename T, bool KM, int TM, int BKT, bool ALIGNED>
__device__ __forceinline__ void g2r(vec16_t<T>* regs, const T* base, __amdgpu_buffer_rsrc_t rsrc, int64_t ld, int nrem,
                                    int k0, int kend, int tid) {
  using OT = OpTile<T, KM, TM, BKT>;
  using V = vec16_t<T>;
#pragma unroll
  for (int i = 0; i < OT::NCH; ++i) {
    const int id = tid + NT * i;
    const int r = id / OT::CPR, c = id % OT::CPR;
    V v;
    if constexpr (ALIGNED) {
      int off;
      bool ok;
      if (!KM) { off = (r * (int)ld + k0 + c * OT::CH) * OT::ES; ok = (r < nrem) && (k0 + c * OT::CH < kend); }
      else { off = ((k0 + r) * (int)ld + c * OT::CH) * OT::ES; ok = (k0 + r < kend) && (c * OT::CH < nrem); }
      (void)ok;  // the zero-select happens in r2s (at LDS-store time) so that the loads stay in flight until then
      v = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    } else {
#pragma unroll
      for (int e = 0; e < OT::CH; ++e) v[e] = (T)0.0f;
      if (!KM) {
        const int gk = k0 + c * OT::CH;
        if (r < nrem && gk < kend) {
          const T* src = base + (int64_t)r * ld + gk;
#pragma unroll
          for (int e = 0; e < OT::CH; ++e)
            if (gk + e < kend) v[e] = src[e];
        }
      } else {
        const int gk = k0 + r, gc = c * OT::CH;
        if (gk < kend && gc < nrem) {
          const T* src = base + (int64_t)gk * ld + gc;
#pragma unroll
          for (int e = 0; e < OT::CH; ++e)
            if (gc + e < nrem) v[e] = src[e];
        }
      }
    }
    regs[i] = v;
  }
}

template <typename T, bool KM, int TM, int BKT, bool ALIGNED>
__device__ __forceinline__ void r2s(const vec16_t<T>* regs, char* buf, int nrem, int k0, int kend, int tid) {
  using OT = OpTile<T, KM, TM, BKT>;
  using V = vec16_t<T>;
#pragma unroll
  for (int i = 0; i < OT::NCH; ++i) {
    const int id = tid + NT * i;
    const int r = id / OT::CPR, c = id % OT::CPR;
    int off;
    if (KM && sizeof(T) == 2) off = r * OT::ROWB + ((c ^ swz_km<OT::W>(r)) * 16);
    else off = r * OT::ROWB + c * 16;
    V v = regs[i];
    if constexpr (ALIGNED) {  // chunks outside the operand read as zero (the buffer load fetched whatever was there)
      const bool ok = KM ? ((k0 + r < kend) && (c * OT::CH < nrem)) : ((r < nrem) && (k0 + c * OT::CH < kend));
      const u32x4 zero = {0u, 0u, 0u, 0u};
      v = __builtin_bit_cast(V, ok ? __builtin_bit_cast(u32x4, v) : zero);
    }
    *reinterpret_cast<V*>(buf + off) = v;
  }
}

// fragment of K-step s (16 k) for the 32 rows/cols starting at `t0` of the tile.
template <typename T, bool KM, int TM, int BKT>
__device__ __forceinline__ frag_t<T> load_frag(const char* buf, int t0, int s, int lane) {
  using OT = OpTile<T, KM, TM, BKT>;
  const int r = lane & 31, h = lane >> 5;
  frag_t<T> f;
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
      const char* p = buf + krow * OT::ROWB + ((chunk ^ swz_km<OT::W>(krow)) * 16) + 8 * (pp & 1);
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

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <typename T, typename TC, bool A_KM, bool B_KM, int TM, int BK, bool ALIGNED>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(GemmParams p) {
  using OA = OpTile<T, A_KM, TM, BK>;
  using OB = OpTile<T, B_KM, TM, BK>;
  constexpr int BMN = 64 * TM;
  constexpr int STAGE = OA::BYTES + OB::BYTES;
  constexpr int CPITCH = BMN + 4;                       // fp32 staging pitch of the epilogue (floats)
  constexpr int CROWS = BMN / 2;                        // the epilogue stages half of the tile rows at a time
  constexpr int SMEM = cmax(2 * STAGE, CROWS * CPITCH * 4);
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tilesN = (p.N + BMN - 1) / BMN;
  // XCD-aware remap: hardware deals workgroups round-robin over 8 XCDs; give each XCD a contiguous tile range
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / tilesN, tn = bid % tilesN;
  const int bz = blockIdx.z / p.splitk, sk = blockIdx.z % p.splitk;
  const int m0 = tm * BMN, n0 = tn * BMN;

  // operand bases at the tile origin + buffer descriptors covering what is left of each allocation from there
  const T* A = reinterpret_cast<const T*>(p.A) + (int64_t)bz * p.sA + (A_KM ? (int64_t)m0 : (int64_t)m0 * p.lda);
  const T* B = reinterpret_cast<const T*>(p.B) + (int64_t)bz * p.sB + (B_KM ? (int64_t)n0 : (int64_t)n0 * p.ldb);
  const int mrem = p.M - m0, nrem = p.N - n0;
  auto remaining = [](bool km, int64_t ld, int rem, int K) -> unsigned {
    const int64_t elems = rem <= 0 ? 0 : (km ? (int64_t)(K - 1) * ld + rem : (int64_t)(rem - 1) * ld + K);
    const int64_t bytes = elems * (int64_t)sizeof(T);
    return bytes > 0xffffffffLL ? 0xffffffffu : (unsigned)bytes;
  };
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A), 0, remaining(A_KM, p.lda, mrem, p.K), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(B), 0, remaining(B_KM, p.ldb, nrem, p.K), 0x00020000);

  const int ksteps = (p.K + BK - 1) / BK;
  const int per = (ksteps + p.splitk - 1) / p.splitk;
  const int ks0 = sk * per;
  int ks1 = ks0 + per; if (ks1 > ksteps) ks1 = ksteps;
  const int kend = (ks1 * BK < p.K) ? ks1 * BK : p.K;
  const int nk = ks1 - ks0;

  f32x16 acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const bool do_colsum = A_KM && (p.flags & GEMM_COLSUM_A) && tn == 0;
  float csum[OA::CH];
#pragma unroll
  for (int e = 0; e < OA::CH; ++e) csum[e] = 0.0f;

  // Two register sets = prefetch distance 2: the loads of K-tile t+2 are issued before tile t+1 is written to LDS,
  // so every global load has ~two MFMA phases to land (hipcc emits the counted vmcnt waits for plain loads).
  vec16_t<T> ra0[OA::NCH], rb0[OB::NCH], ra1[OA::NCH], rb1[OB::NCH];
  auto LOAD = [&](vec16_t<T>* ra, vec16_t<T>* rb, int kt) {
    g2r<T, A_KM, TM, BK, ALIGNED>(ra, A, rsA, p.lda, mrem, (ks0 + kt) * BK, kend, tid);
    g2r<T, B_KM, TM, BK, ALIGNED>(rb, B, rsB, p.ldb, nrem, (ks0 + kt) * BK, kend, tid);
  };
  auto STORE = [&](const vec16_t<T>* ra, const vec16_t<T>* rb, int buf, int kt) {
    r2s<T, A_KM, TM, BK, ALIGNED>(ra, smem + buf * STAGE, mrem, (ks0 + kt) * BK, kend, tid);
    r2s<T, B_KM, TM, BK, ALIGNED>(rb, smem + buf * STAGE + OA::BYTES, nrem, (ks0 + kt) * BK, kend, tid);
    if (A_KM && do_colsum) {  // k-major A: rows past kend / columns past M contribute nothing
#pragma unroll
      for (int i = 0; i < OA::NCH; ++i) {
        const int id = tid + NT * i;
        const int r = id / OA::CPR, c = id % OA::CPR;
        const bool ok = ((ks0 + kt) * BK + r < kend) && (c * OA::CH < mrem);
#pragma unroll
        for (int e = 0; e < OA::CH; ++e) csum[e] += ok ? to_f32(ra[i][e]) : 0.0f;
      }
    }
  };
  auto COMPUTE = [&](int buf) {
    const char* curA = smem + buf * STAGE;
    const char* curB = curA + OA::BYTES;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      frag_t<T> fa[TM], fb[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = load_frag<T, A_KM, TM, BK>(curA, wr * 32 * TM + 32 * i, s, lane);
#pragma unroll
      for (int j = 0; j < TM; ++j) fb[j] = load_frag<T, B_KM, TM, BK>(curB, wc * 32 * TM + 32 * j, s, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = mma16(fa[i], fb[j], acc[i][j]);
    }
  };

  if (nk > 0) {
    LOAD(ra0, rb0, 0);
    STORE(ra0, rb0, 0, 0);
    if (nk > 1) LOAD(ra0, rb0, 1);
    if (nk > 2) LOAD(ra1, rb1, 2);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // even phase: LDS buffer 0 = tile kt; set 0 = tile kt+1 (in flight); set 1 = tile kt+2 (in flight)
    COMPUTE(0);
    if (kt + 1 < nk) STORE(ra0, rb0, 1, kt + 1);
    if (kt + 3 < nk) LOAD(ra0, rb0, kt + 3);
    __syncthreads();
    if (kt + 1 >= nk) break;
    // odd phase: LDS buffer 1 = tile kt+1; set 1 = tile kt+2; set 0 = tile kt+3
    COMPUTE(1);
    if (kt + 2 < nk) STORE(ra1, rb1, 0, kt + 2);
    if (kt + 4 < nk) LOAD(ra1, rb1, kt + 4);
    __syncthreads();
  }

  TC* C = reinterpret_cast<TC*>(p.C) + (int64_t)bz * p.sC;
  const bool atomic = (p.splitk > 1) || (p.flags & GEMM_ACCUM);
  const int h = lane >> 5;

  if (A_KM && do_colsum) {
    // all chunks of a thread share one column chunk (NT is a multiple of CPR): fold the k-rows through LDS
    float* red = reinterpret_cast<float*>(smem);
    for (int i = tid; i < BMN; i += NT) red[i] = 0.0f;
    __syncthreads();
    const int c = tid % OA::CPR;
#pragma unroll
    for (int e = 0; e < OA::CH; ++e) atomicAdd(&red[c * OA::CH + e], csum[e]);
    __syncthreads();
    for (int i = tid; i < BMN; i += NT)
      if (m0 + i < p.M) atomicAdd(p.colsum + m0 + i, red[i]);
    __syncthreads();
  }

  if (atomic) {
    if constexpr (sizeof(TC) == 4) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int col = n0 + wc * 32 * TM + 32 * j + (lane & 31);
          if (col >= p.N) continue;
          const float bv = ((p.flags & GEMM_BIAS) && sk == 0) ? p.bias[col] : 0.0f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = m0 + wr * 32 * TM + 32 * i + acc_row(e, h);
            if (row < p.M) atomicAdd(reinterpret_cast<float*>(C) + (int64_t)row * p.ldc + col, acc[i][j][e] * p.alpha + bv);
          }
        }
    }
    return;
  }

  // stage the fp32 tile through LDS one half (wave row wr) at a time (the K loop's last barrier has retired all
  // fragment reads), then walk it row-major in 16-byte chunks of complete rows
  float* ct = reinterpret_cast<float*>(smem);
  constexpr int GPR = BMN / 8;  // 8-column groups per tile row
  const TC* R = reinterpret_cast<const TC*>(p.R);
  // a thread keeps the same 8-column group for all its rows (NT is a multiple of GPR): fetch its bias once
  float bias8[8];
  {
    const int col = n0 + (tid % GPR) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.0f;
    if (p.flags & GEMM_BIAS) {
      if (ALIGNED && col + 8 <= p.N) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p.bias + col);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(p.bias + col + 4);
        bias8[0] = lo[0]; bias8[1] = lo[1]; bias8[2] = lo[2]; bias8[3] = lo[3];
        bias8[4] = hi[0]; bias8[5] = hi[1]; bias8[6] = hi[2]; bias8[7] = hi[3];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (col + e < p.N) bias8[e] = p.bias[col + e];
      }
    }
  }
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
    if (wr == half) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            ct[(32 * i + acc_row(e, h)) * CPITCH + wc * 32 * TM + 32 * j + (lane & 31)] = acc[i][j][e];
    }
    __syncthreads();
    for (int gidx = tid; gidx < CROWS * GPR; gidx += NT) {
      const int r = gidx / GPR, cg = gidx % GPR;
      const int row = m0 + half * CROWS + r, col = n0 + cg * 8;
      if (row >= p.M || col >= p.N) continue;
      float v[8];
      {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ct + r * CPITCH + cg * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(ct + r * CPITCH + cg * 8 + 4);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      }
      const bool full = ALIGNED && (col + 8 <= p.N);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias8[e];
      if (p.flags & GEMM_SAVE_PRE) {
        TC* ap = reinterpret_cast<TC*>(p.aux) + (int64_t)row * p.ldaux + col;
        if (full && sizeof(TC) == 2) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x8*>(ap) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (col + e < p.N) ap[e] = from_f32<TC>(v[e]);
        }
      }
      if (p.flags & GEMM_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_for<TC>(v[e]);
      }
      if (p.flags & GEMM_DGELU) {
        const T* ap = reinterpret_cast<const T*>(p.aux) + (int64_t)row * p.ldaux + col;
        if (full && sizeof(T) == 2) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= dgelu_for<T>((float)a[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (col + e < p.N) v[e] *= dgelu_for<T>(to_f32(ap[e]));
        }
      }
      if (p.flags & GEMM_RESID) {
        const TC* rp = R + (int64_t)row * p.ldr + col;
        if (full && sizeof(TC) == 2) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(rp);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (col + e < p.N) v[e] += to_f32(rp[e]);
        }
      }
      TC* dst = C + (int64_t)row * p.ldc + col;
      if (full && sizeof(TC) == 2) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x8*>(dst) = o;
      } else if (full && sizeof(TC) == 4) {
        f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(dst) = lo;
        *reinterpret_cast<f32x4*>(dst + 4) = hi;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (col + e < p.N) dst[e] = from_f32<TC>(v[e]);
      }
    }
  }
}

template <typename T, typename TC, int TM, int BK, bool ALIGNED>
int launch_layout(const GemmParams& p, int a_km, int b_km, dim3 grid, hipStream_t st) {
  if (!a_km && !b_km) hipLaunchKernelGGL((gemm_kernel<T, TC, false, false, TM, BK, ALIGNED>), grid, dim3(NT), 0, st, p);
  else if (!a_km && b_km) hipLaunchKernelGGL((gemm_kernel<T, TC, false, true, TM, BK, ALIGNED>), grid, dim3(NT), 0, st, p);
  else if (a_km && !b_km) hipLaunchKernelGGL((gemm_kernel<T, TC, true, false, TM, BK, ALIGNED>), grid, dim3(NT), 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<T, TC, true, true, TM, BK, ALIGNED>), grid, dim3(NT), 0, st, p);
  return dcc_check_launch();
}


// ------------------------------------------------------------------------------------------------------------------
// fp32 GEMMs with a handful of rows (M <= 32: the study aggregator and projection heads work on [batch, 512] tensors).
// The tiled kernel runs them on 4 workgroups with a serial, latency-bound K loop (45-160 us each, on the critical
// path between the towers and the loss).  Here K is spread over the threads of a block instead: every thread issues
// all its loads up front, keeps an [MT][4] accumulator slab, and the block folds the partials through shuffles + LDS.
__device__ __forceinline__ float skinny_epilogue(const GemmParams& p, float v, int row, int col) {
  v = v * p.alpha + ((p.flags & GEMM_BIAS) ? p.bias[col] : 0.0f);
  if (p.flags & GEMM_SAVE_PRE) reinterpret_cast<float*>(p.aux)[(int64_t)row * p.ldaux + col] = v;
  if (p.flags & GEMM_GELU) v = gelu_f(v);
  if (p.flags & GEMM_DGELU) v *= dgelu_f(reinterpret_cast<const float*>(p.aux)[(int64_t)row * p.ldaux + col]);
  if (p.flags & GEMM_RESID) v += reinterpret_cast<const float*>(p.R)[(int64_t)row * p.ldr + col];
  return v;
}

// One step of a halving wave reduction: N values per lane -> N/2, the lanes with bit `o` set keep the upper half.
// Reducing N values over 64 lanes this way costs ~N shuffles instead of 6 N.
template <int N, int NMAX>
__device__ __forceinline__ void fold_step(float (&v)[NMAX], int lane, int o) {
  const bool up = (lane & o) != 0;
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const float send = up ? v[i] : v[i + N / 2];
    const float keep = up ? v[i + N / 2] : v[i];
    v[i] = keep + __shfl_xor(send, o);
  }
}

// B[n][k] (K contiguous): block = 4 output columns, threads stride over K in float4 steps
template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_nt_kernel(GemmParams p) {
  __shared__ float red[4][MT * 4];
  const float* A = reinterpret_cast<const float*>(p.A);
  const float* B = reinterpret_cast<const float*>(p.B);
  const int n0 = blockIdx.x * 4, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NV = MT * 4;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0f;
  for (int k = tid * 4; k < p.K; k += 1024) {
    f32x4 b[4], a[MT];
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const f32x4*>(B + (int64_t)(n0 + c < p.N ? n0 + c : p.N - 1) * p.ldb + k);
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(A + (int64_t)(m < p.M ? m : p.M - 1) * p.lda + k);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[m * 4 + c] += a[m][0] * b[c][0] + a[m][1] * b[c][1] + a[m][2] * b[c][2] + a[m][3] * b[c][3];
  }
  fold_step<NV, NV>(acc, lane, 32); fold_step<NV / 2, NV>(acc, lane, 16); fold_step<NV / 4, NV>(acc, lane, 8);
  fold_step<NV / 8, NV>(acc, lane, 4); fold_step<NV / 16, NV>(acc, lane, 2);
  {
    const int base = ((lane >> 5) & 1) * (NV / 2) + ((lane >> 4) & 1) * (NV / 4) + ((lane >> 3) & 1) * (NV / 8) +
                     ((lane >> 2) & 1) * (NV / 16) + ((lane >> 1) & 1) * (NV / 32);
#pragma unroll
    for (int i = 0; i < NV / 32; ++i) {
      const float v = acc[i] + __shfl_xor(acc[i], 1);
      if (!(lane & 1)) red[wave][base + i] = v;
    }
  }
  __syncthreads();
  if (tid < MT * 4) {
    const int m = tid >> 2, c = tid & 3, col = n0 + c;
    if (m < p.M && col < p.N) {
      const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + col] = skinny_epilogue(p, v, m, col);
    }
  }
}

// B[k][n] (N contiguous): block = 16 output columns (4 float4 lanes) x 64 K slices
template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_nn_kernel(GemmParams p) {
  __shared__ float red[4][4][MT * 4];  // [wave][quad][m*4+e]
  const float* A = reinterpret_cast<const float*>(p.A);
  const float* B = reinterpret_cast<const float*>(p.B);
  const int tid = threadIdx.x, q = tid & 3, s = tid >> 2, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16 + 4 * q;
  const int ks = ((p.K + 63) / 64 + 3) & ~3;  // K slice per thread, a multiple of 4 (A is read in float4 steps)
  const int kb = s * ks;
  constexpr int NV = MT * 4;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0f;
  const bool ncol = n0 < p.N;  // N % 4 == 0: a quad is either fully inside or outside
  for (int k = kb; k < kb + ks && k < p.K; k += 4) {
    f32x4 b[4], a[MT];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(B + (int64_t)(k + j < p.K ? k + j : p.K - 1) * p.ldb + (ncol ? n0 : 0));
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(A + (int64_t)(m < p.M ? m : p.M - 1) * p.lda + k);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float av = (k + j < p.K) ? a[m][j] : 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m * 4 + c] += av * b[j][c];
      }
  }
  // fold the 16 K slices of a wave (lanes with equal q) by halving, then the 4 waves through LDS
  fold_step<NV, NV>(acc, lane, 32); fold_step<NV / 2, NV>(acc, lane, 16); fold_step<NV / 4, NV>(acc, lane, 8);
  fold_step<NV / 8, NV>(acc, lane, 4);
  {
    const int base = ((lane >> 5) & 1) * (NV / 2) + ((lane >> 4) & 1) * (NV / 4) + ((lane >> 3) & 1) * (NV / 8) +
                     ((lane >> 2) & 1) * (NV / 16);
#pragma unroll
    for (int i = 0; i < NV / 16; ++i) red[wave][q][base + i] = acc[i];
  }
  __syncthreads();
  for (int i = tid; i < 4 * MT * 4; i += 256) {
    const int qq = i / (MT * 4), r = i - qq * (MT * 4), m = r >> 2, c = r & 3;
    const int col = blockIdx.x * 16 + 4 * qq + c;
    if (m < p.M && col < p.N) {
      const float v = red[0][qq][r] + red[1][qq][r] + red[2][qq][r] + red[3][qq][r];
      reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + col] = skinny_epilogue(p, v, m, col);
    }
  }
}

template <int MT>
int launch_skinny(const GemmParams& p, int b_km, hipStream_t st) {
  if (b_km) hipLaunchKernelGGL((gemm_skinny_nn_kernel<MT>), dim3((p.N + 15) / 16), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((gemm_skinny_nt_kernel<MT>), dim3((p.N + 3) / 4), dim3(256), 0, st, p);
  return dcc_check_launch();
}

}  // namespace

// fp8 x fp8 -> bf16: C[m][n] = alpha * sa * sb * sum_k A8(m,k) B8(n,k) (+ bias, GELU / GELU' / residual epilogues as
// dcc_gemm); both operands k-contiguous bytes.  See include/dcc.h.
extern "C" int dcc_gemm_fp8(const void* A8, const void* B8, void* C, const float* bias, const void* R, void* aux,
                            const float* scale_a, const float* scale_b, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr,
                            int64_t ldaux, int M, int N, int K, float alpha, int flags, int fmt_a, int fmt_b, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A8 || !B8 || !C) return DCC_ERR_ARG;
  if ((flags & GEMM_BIAS) && !bias) return DCC_ERR_ARG;
  if ((flags & GEMM_RESID) && !R) return DCC_ERR_ARG;
  if ((flags & (GEMM_SAVE_PRE | GEMM_DGELU)) && !aux) return DCC_ERR_ARG;
  if ((fmt_a | fmt_b) & ~1) return DCC_ERR_ARG;
  GemmParams p{A8, B8, C, bias, R, aux, nullptr, lda, ldb, ldc, ldr, ldaux, 0, 0, 0, M, N, K, 1, 1, alpha, flags & 0xff, scale_a, scale_b, fmt_a, fmt_b};
  return dcc_gemm_ring_launch_fp8(p, (hipStream_t)stream);
}

// See include/dcc.h for the contract.
extern "C" int dcc_gemm(const void* A, const void* B, void* C, const float* bias, const void* R, void* aux,
                        float* colsum, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int64_t ldaux, int M, int N,
                        int K, int batch, int64_t sA, int64_t sB, int64_t sC, int splitk, float alpha, int flags,
                        int a_km, int b_km, int in_dtype, int out_dtype, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || splitk <= 0) return DCC_ERR_ARG;
  if (!A || !B || !C) return DCC_ERR_ARG;
  const int ch = in_dtype == DCC_BF16 ? 8 : 4;
  // 16-byte chunk accesses need chunk-multiple pitches/extents and 16-byte aligned bases; otherwise the
  // element-wise guarded variant runs (tiny loss-side fp32 GEMMs with odd batch sizes land there)
  bool aligned = true;
  if ((lda % ch) || (ldb % ch) || (sA % ch) || (sB % ch)) aligned = false;
  if ((!a_km && (K % ch)) || (a_km && (M % ch))) aligned = false;
  if ((!b_km && (K % ch)) || (b_km && (N % ch))) aligned = false;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) aligned = false;
  const int och = 8;
  if ((ldc % och) || (N % och) || (sC % och)) aligned = false;
  if ((flags & GEMM_RESID) && ((ldr % och) || ((uintptr_t)R & 15))) aligned = false;
  if ((flags & GEMM_BIAS) && ((uintptr_t)bias & 15)) aligned = false;
  if ((flags & (GEMM_SAVE_PRE | GEMM_DGELU)) && ((ldaux % och) || ((uintptr_t)aux & 15))) aligned = false;
  if ((splitk > 1 || (flags & GEMM_ACCUM)) && out_dtype != DCC_F32) return DCC_ERR_ARG;
  if ((flags & GEMM_BIAS) && !bias) return DCC_ERR_ARG;
  if ((flags & GEMM_RESID) && !R) return DCC_ERR_ARG;
  if ((flags & (GEMM_SAVE_PRE | GEMM_DGELU)) && !aux) return DCC_ERR_ARG;
  if ((flags & GEMM_COLSUM_A) && (!colsum || !a_km || batch != 1)) return DCC_ERR_ARG;
  if (splitk > 1 && (flags & 0xff & ~(GEMM_BIAS | GEMM_ACCUM | GEMM_COLSUM_A))) return DCC_ERR_ARG;  // non-linear epilogues need the full sum
  if (!aligned && in_dtype != DCC_F32) return DCC_ERR_UNSUPPORTED;  // bf16 callers always meet the chunk rules
  GemmParams p{A, B, C, bias, R, aux, colsum, lda, ldb, ldc, ldr, ldaux, sA, sB, sC, M, N, K, batch, splitk, alpha, (flags & 0xff) | (((flags >> 8) & 0xff) == 4 ? 0x1000 : 0) | (((flags >> 8) & 0xff) == 7 ? 0x2000 : 0), nullptr, nullptr, 0, 0};
  if ((int64_t)batch * splitk > 65535) return DCC_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == DCC_F32 && out_dtype == DCC_F32 && aligned && !a_km && M <= 32 && batch == 1 && splitk == 1 && K >= 64 &&
      !(flags & (GEMM_ACCUM | GEMM_COLSUM_A)) && ((flags >> 8) & 0xff) != 3) {  // a handful of rows: K spread over threads
    if (M <= 8) return launch_skinny<8>(p, b_km, st);
    if (M <= 16) return launch_skinny<16>(p, b_km, st);
    if (M <= 24) return launch_skinny<24>(p, b_km, st);
    return launch_skinny<32>(p, b_km, st);
  }
  // tile choice: 128x128 unless that leaves most of the 256 CUs idle
  const int64_t t128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * batch * splitk;
  const int variant = (flags >> 8) & 0xff;  // tuning knob (bench / tests only): 1 = BK 32 for the 128-tile bf16 kernels, 2 = 64x64 tiles, 3 = no ring kernel
  const bool small = in_dtype == DCC_BF16 && (t128 < 192 || variant == 2);
  const int bmn = small ? 64 : 128;
  // persistent LDS-DMA ring kernel: the towers' large GEMMs, and (measured in round 1) also the small grids
  // of the text tower and the deep-K rel-pos table gradients, where its 3-4 stage pipeline hides the per-step load
  // latency the tiled kernel exposes (variant 6 restores the tiled kernel for small grids)
  if (in_dtype == DCC_BF16 && out_dtype == DCC_BF16 && aligned && variant == 13 && !a_km && !b_km) {  // experimental eight-phase kernel
    const int rc = dcc_gemm_p8_launch(p, st);
    if (rc != DCC_ERR_UNSUPPORTED) return rc;
  }
  if (in_dtype == DCC_BF16 && aligned && (variant == 8 || variant == 9 || variant == 11)) {  // A/B knob: force the wide-tile kernel
    const int rc = dcc_gemm_wide_launch(p, a_km, b_km, out_dtype, variant == 8 ? 256 : 128, variant == 11 ? 3 : 4, st);
    if (rc != DCC_ERR_UNSUPPORTED) return rc;
  }
  // Early MViT blocks (C = 96 / 192 at 10^5 - 10^6 rows): HBM-bound products with a reduction of a few k-steps.  The streaming
  // panel kernel (gemm_stream.hip: weight resident in LDS, every wave on its own rows, no barriers) was built for all of them and
  // MEASURED against the tiled kernels (tools/stream_bench.py).  With direct 16-byte stores from the accumulator layout it won only
  // the plain-bias wide outputs (profiles/r03_stream_bench.txt: a lane pair covers 32 bytes of a row per store instruction); since
  // its column tiles leave in pairs through a wave-private LDS strip as complete 128-byte lines (and the epilogue flags are
  // compile-time) it wins every product with K <= 192 and N >= 192 — qkv 1.03-1.29x, project 1.12-1.36x, fc1 with GELU and the
  // saved pre-activation 1.09-1.11x, gelu' 1.12-1.25x (profiles/r03_stream_bench_staged.txt) — and still loses the N = 96 outputs
  // (0.83-0.93x: three column tiles = one pair + one direct tile, and the tiled kernels stream those at 4.5-5.3 TB/s).
  // DCC_STREAM_GEMM=0 disables it, =3 restores round 3's first rule (plain bias, N >= 288), =2 (or variant 12: tests,
  // tools/stream_bench.py) widens it to its whole domain.
  static const int stream_mode = getenv("DCC_STREAM_GEMM") ? atoi(getenv("DCC_STREAM_GEMM")) : 1;
  const bool stream_all = stream_mode == 2 || variant == 12;
  const bool stream_pick = stream_mode == 3 ? (K <= 192 && N >= 288 && !(flags & (GEMM_GELU | GEMM_DGELU | GEMM_SAVE_PRE | GEMM_RESID)))
                                            : (K <= 192 && N >= 192);
  if ((stream_mode || variant == 12) && in_dtype == DCC_BF16 && out_dtype == DCC_BF16 && aligned && (variant == 0 || variant == 12) &&
      !a_km && !b_km && batch == 1 && splitk == 1 && !(flags & (GEMM_ACCUM | GEMM_COLSUM_A)) && M >= 16384 && K <= 576 &&
      (K <= 192 || N <= 192) && (stream_all || stream_pick)) {
    const int rc = dcc_gemm_stream_launch(p, st);
    if (rc != DCC_ERR_UNSUPPORTED) return rc;
  }
  // Eight-phase 256 x 256 kernel (gemm_p8.hip) where it wins in the interleaved table (profiles/r03_p8_bench.txt, second run): the
  // qkv projections — plain / bias epilogue, wide output, K = C — 57.8 vs 61.9 us at stage 2 (M 50208 N 1152 K 384), 53.1 vs 59.9 at
  // stage 3 (M 12576 N 2304 K 768), and the plain N 1536 K 384 product 77.3 vs 82.7.  Its epilogue is not covered by anything (one
  // workgroup per CU), so everything with a second output tensor or an activation stays on the 256 x 128 wide kernel.
  static const int p8_mode = getenv("DCC_P8_GEMM") ? atoi(getenv("DCC_P8_GEMM")) : 1;   // A/B knob: 0 = never
  if (p8_mode && in_dtype == DCC_BF16 && out_dtype == DCC_BF16 && aligned && variant == 0 && !a_km && !b_km && batch == 1 && splitk == 1 &&
      !(flags & (GEMM_ACCUM | GEMM_COLSUM_A | GEMM_GELU | GEMM_DGELU | GEMM_SAVE_PRE | GEMM_RESID)) && M >= 8192 && N >= 1024 &&
      K >= 384 && K <= 768) {
    const int rc = dcc_gemm_p8_launch(p, st);
    if (rc != DCC_ERR_UNSUPPORTED) return rc;
  }
  // Measured on MI355X (tools/gemm_bench.py, profiles/r02_kernel_bench.txt): forward GEMMs with a short reduction and a
  // wide output (fc1: N = 4C, qkv: N = 3C; K = C <= 768) run 12-15 % faster on 256 x 256 tiles with 8 unspecialised
  // waves (half the L2 -> LDS operand traffic per FLOP) than on the ring kernel's 128 x 128 tiles; N = C outputs, k-major
  // operands and deep reductions stay on the ring kernels (variant 10 disables this choice).
  static const int wide_mode = getenv("DCC_WIDE_MODE") ? atoi(getenv("DCC_WIDE_MODE")) : 0;  // experiments: 1-3 widen the choice
  if (in_dtype == DCC_BF16 && out_dtype == DCC_BF16 && aligned && variant == 0 && !a_km && batch == 1 && splitk == 1 &&
      !(flags & (GEMM_ACCUM | GEMM_COLSUM_A)) && M >= 8192 && K >= 128) {
    int bn = 0, ns = 4;
    if (!b_km && N >= 1024 && K <= 768) {
      // 256 x 128 tiles, two workgroups per CU (one's epilogue under the other's main loop): qkv 61.5 us against 68.5 (256 x 256) and
      // 72.8 (ring) at M 50208 N 1152 K 384; fc1 (pre-activation + GELU: two stores) 98.7 against 104.6 / 119.8, and 85.7 against
      // 95.2 / 100.3 at stage 3 (M 12576 N 3072 K 768) — tools/p8_bench.py, variants interleaved in one process
      // (profiles/r03_p8_bench.txt; round 2 had measured fc1 in sequential blocks per variant and kept it on 256 x 256)
      bn = 128; ns = 3;
      if (wide_mode == 5 && (flags & GEMM_SAVE_PRE)) { bn = 256; ns = 4; }   // A/B: the round-2 choice for fc1
    }
    if (b_km && (flags & GEMM_DGELU) && N >= 768 && K <= 768 && wide_mode != 4) { bn = 128; ns = 3; }  // 134 us against 149 (ring) at stage 2, 206 against 230 at stage 1 (N 768 K 192)
    if (wide_mode == 4 && !b_km && N >= 1024 && K <= 768) { bn = 256; ns = 4; }   // A/B: the round-2 choice
    if (wide_mode >= 1 && wide_mode <= 3 && b_km && N >= 1024 && K <= 768) { bn = 256; ns = 4; }  // dgelu / NT with a wide output
    if (wide_mode >= 2 && wide_mode <= 3 && N >= 256 && N < 1024) { bn = (N % 256 == 0) ? 256 : 128; ns = 4; }  // N = C outputs
    if (wide_mode == 3 && bn == 0 && N >= 1024) { bn = 256; ns = 4; }              // deep K too
    if (bn) {
      const int rc = dcc_gemm_wide_launch(p, a_km, b_km, out_dtype, bn, ns, st);
      if (rc != DCC_ERR_UNSUPPORTED) return rc;
    }
  }
  if (in_dtype == DCC_BF16 && aligned && variant != 3 && variant != 2 && !(small && variant == 6)) {
    const int rc = dcc_gemm_ring_launch(p, a_km, b_km, out_dtype, st);
    if (rc != DCC_ERR_UNSUPPORTED) return rc;
  }
  const int64_t tiles = (int64_t)((M + bmn - 1) / bmn) * ((N + bmn - 1) / bmn);
  if (tiles > 0x7fffffffLL) return DCC_ERR_ARG;
  dim3 grid((unsigned)tiles, 1, (unsigned)(batch * splitk));
  if (in_dtype == DCC_BF16 && out_dtype == DCC_BF16) {
    if (small) return launch_layout<bf16_t, bf16_t, 1, 64, true>(p, a_km, b_km, grid, st);
    return variant == 1 ? launch_layout<bf16_t, bf16_t, 2, 32, true>(p, a_km, b_km, grid, st)
                        : launch_layout<bf16_t, bf16_t, 2, 64, true>(p, a_km, b_km, grid, st);
  }
  if (in_dtype == DCC_BF16 && out_dtype == DCC_F32) {
    if (small) return launch_layout<bf16_t, float, 1, 64, true>(p, a_km, b_km, grid, st);
    return variant == 1 ? launch_layout<bf16_t, float, 2, 32, true>(p, a_km, b_km, grid, st)
                        : launch_layout<bf16_t, float, 2, 64, true>(p, a_km, b_km, grid, st);
  }
  if (in_dtype == DCC_F32 && out_dtype == DCC_F32)
    return aligned ? launch_layout<float, float, 2, 32, true>(p, a_km, b_km, grid, st)
                   : launch_layout<float, float, 2, 32, false>(p, a_km, b_km, grid, st);
  return DCC_ERR_UNSUPPORTED;
}
