// Flash-style multi-head attention for gfx950: forward, dQ and dK/dV kernels.
//
//   z[q][k] = scale * <Qf[q], Kf[k]> + kbias[k];  P = softmax_k(z);  O = dropout(P) V (+ Qres residual)
//
// dropout(P) = P * keep / (1 - p) is nn.MultiheadAttention(dropout=p) / HF BERT attention_probs_dropout_prob: applied to
// the NORMALISED probabilities, so the softmax denominator l sums the undropped tile and only the tile that feeds the
// P V product is masked.  keep(b, h, q, k) is a counter hash of (seed, device step counter, element) that the two
// backward kernels regenerate: dV = dropout(P)^T dO, dz = P * (keep/(1-p) * (dO V^T) - D), D = rowsum(dO * O).
//
// Qf/Kf are DQK wide, V/O are DV wide.  MViT's decomposed relative-position bias is folded into
// the contraction: Qf = [q | tab(q)/scale], Kf = [k | one-hot(k_t,k_h,k_w)], so the bias costs
// (DQK-96)/16 extra MFMA K-steps and no gather (see relpos.hip).  The score matrix is never
// written to HBM; only LSE[q] is kept for the backward.
//
// All three kernels compute the TRANSPOSED score tile  S^T = Kf Qf^T  (keys on accumulator
// rows, queries on lanes) or S (queries on rows, keys on lanes) so that the softmax statistics
// are per-lane scalars and the probability tile feeds the next MFMA straight from the
// accumulator registers (cdna guide: "accumulator tile as the next MFMA's operand").
#include <cstdlib>
#include <type_traits>

#include "common.h"

struct AttnParams {
  const void* Q; const void* K; const void* V; void* O;
  float* LSE;            // [B,H,Nq] natural-log sum-exp of z
  const float* kbias;    // [B,Nk] additive key bias (e.g. padding mask) or null
  const void* Qres;      // residual source with Q's strides (adds Qres[q][0:DV] to O for q >= res_start) or null
  // backward only
  const void* dO; const float* Dv;   // Dv[B,H,Nq] = rowsum(dO * O_attn)
  void* dQ; void* dK; void* dV;      // dK/dV: T when qsplit==1 else fp32 (atomicAdd, pre-zeroed)
  int64_t sqb, sqn, sqh, skb, skn, skh, svb, svn, svh, sob, son, soh;
  int64_t sdqb, sdqn, sdqh, sdkb, sdkn, sdkh, sdvb, sdvn, sdvh;
  int B, H, Nq, Nk;
  float scale;
  int res_start;
  int qsplit;
  // attention-probability dropout (0 = off): mask = hash32(seed', (row << 32) | key) >= thr, row = (b*H + h)*Nq + q
  float drop_p;
  uint64_t drop_seed;
  const int64_t* drop_step;   // nullable device counter mixed into the seed (HIP-graph replays draw fresh masks)
};

namespace {

// v_exp_f32 without libm's denormal rescue (3-4 extra VALU ops per element): arguments here are <= 0 and results
// below 2^-126 only ever multiply values that are added to O(1) sums, so flushing them is immaterial
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int ANT = 256;  // threads per block
// minimum waves per SIMD the register allocation must leave room for (hipcc otherwise spends up to 512 registers per lane)
#ifndef DCC_ATTN_OCC_FWD
#define DCC_ATTN_OCC_FWD 2   // measured: 168 instead of 180 registers at D = 128 / 96, three waves per SIMD, forward -3.5 % (profiles/r04_attn_pipe_bench.txt, occ2)
#endif
#ifndef DCC_ATTN_OCC_DQ
#define DCC_ATTN_OCC_DQ 1
#endif
#ifndef DCC_ATTN_OCC_DKV
#define DCC_ATTN_OCC_DKV 1
#endif
#ifndef DCC_ATTN_HAND_DQ
#define DCC_ATTN_HAND_DQ 1
#endif

struct DropCtx {
  uint64_t seed; uint32_t thr; float inv_keep;
  __device__ __forceinline__ void init(const AttnParams& p) {
    seed = p.drop_seed;
    if (p.drop_step) seed += (uint64_t)(*p.drop_step) * 0x9E3779B97F4A7C15ULL;
    seed *= 0x100000001b3ULL;
    thr = (uint32_t)((double)p.drop_p * 4294967296.0);
    inv_keep = 1.0f / (1.0f - p.drop_p);
  }
  // multiplier of P[row][key]: 0 or 1/(1-p)
  __device__ __forceinline__ float mul(uint32_t row, uint32_t key) const {
    return hash32(seed + (((uint64_t)row << 32) | key)) >= thr ? inv_keep : 0.0f;
  }
};
constexpr int TK = 64;    // rows staged per LDS tile

template <typename T, int D> struct RowTile {  // [rows][D] tile with padded pitch
  static constexpr int ES = sizeof(T);
  static constexpr int CH = 16 / ES;
  static constexpr int PITCH = D * ES + 16;
  static constexpr int CPR = D / CH;
};

// Cooperative staging of a TK-row tile, split in two halves so that the global loads of tile t+1 are in flight
// while tile t is being computed (register prefetch; a rolled load->wait->store loop serialises one HBM round
// trip per 16-byte chunk and was the bottleneck of the first version).
template <typename T, int D> struct TileRegs {
  static constexpr int NCH = TK * RowTile<T, D>::CPR / ANT;  // 16-byte chunks per thread
  vec16_t<T> v[NCH];
};

// Rows come through a buffer descriptor whose range ends with the last valid row: rows past n_valid read as zero
// without a branch (branchy loads keep hipcc from hoisting them and cost 64-bit address arithmetic per chunk).
template <typename T, int D>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const T* base, int64_t row_stride, int n_valid) {
  const int64_t bytes = n_valid > 0 ? ((int64_t)(n_valid - 1) * row_stride + D) * (int64_t)sizeof(T) : 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, bytes > 0x7fffffffLL ? 0x7fffffff : (int)bytes, 0x00020000);
}

template <typename T, int D>
__device__ __forceinline__ void tile_load(TileRegs<T, D>& r, __amdgpu_buffer_rsrc_t rsrc, int row_stride_bytes, int row0,
                                          int tid) {
  using RT = RowTile<T, D>;
  using V = vec16_t<T>;
#pragma unroll
  for (int i = 0; i < TileRegs<T, D>::NCH; ++i) {
    const int id = tid + ANT * i;
    const int rr = id / RT::CPR, c = id % RT::CPR;
    r.v[i] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (row0 + rr) * row_stride_bytes + c * 16, 0, 0));
  }
}

template <typename T, int D>
__device__ __forceinline__ void tile_store(const TileRegs<T, D>& r, char* lds, int tid) {
  using RT = RowTile<T, D>;
  using V = vec16_t<T>;
#pragma unroll
  for (int i = 0; i < TileRegs<T, D>::NCH; ++i) {
    const int id = tid + ANT * i;
    const int rr = id / RT::CPR, c = id % RT::CPR;
    *reinterpret_cast<V*>(lds + rr * RT::PITCH + c * 16) = r.v[i];
  }
}

// fragment from global memory: lane (r,h) takes 8 elements at row (row0 + r), cols 16*s + 8*h ..
template <typename T>
__device__ __forceinline__ frag_t<T> gfrag(const T* base, int64_t row_stride, int row, int n_valid, int s, int h) {
  frag_t<T> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (T)0.0f;
  if (row < n_valid) {
    const T* p = base + (int64_t)row * row_stride + 16 * s + 8 * h;
    if constexpr (sizeof(T) == 2) {
      f = *reinterpret_cast<const bf16x8*>(p);
    } else {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
      f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
      f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    }
  }
  return f;
}

// row-read fragment from a RowTile: lane (r,h): tile row (t0 + r), cols 16*s + 8*h .. +7
template <typename T, int D>
__device__ __forceinline__ frag_t<T> row_frag(const char* lds, int t0, int s, int lane) {
  using RT = RowTile<T, D>;
  const int r = lane & 31, h = lane >> 5;
  const char* p = lds + (t0 + r) * RT::PITCH + (16 * s + 8 * h) * RT::ES;
  frag_t<T> f;
  if constexpr (sizeof(T) == 2) {
    f = *reinterpret_cast<const bf16x8*>(p);
  } else {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 16);
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  }
  return f;
}

// transposed fragment matching acc_as_frag's k order: element j of lane (r,h) =
//   tile[row = t0 + 16*s2 + 8*(j>>2) + 4*h + (j&3)][col = c0 + r]
template <typename T, int D, int PITCH = RowTile<T, D>::PITCH>
__device__ __forceinline__ frag_t<T> tr_frag(const char* lds, int t0, int s2, int c0, int lane) {
  const int r = lane & 31, h = lane >> 5;
  frag_t<T> f;
  if constexpr (sizeof(T) == 2) {
    const int i = lane & 15, q = i >> 2, pp = i & 3, g = (lane >> 4) & 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = t0 + 16 * s2 + 8 * half + 4 * h + q;
      const char* p = lds + row * PITCH + (c0 + 16 * g + 4 * pp) * 2;
      const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p));
      const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
      f[4 * half + 0] = tb[0]; f[4 * half + 1] = tb[1]; f[4 * half + 2] = tb[2]; f[4 * half + 3] = tb[3];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = t0 + 16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3);
      f[j] = *reinterpret_cast<const float*>(lds + row * PITCH + (c0 + r) * 4);
    }
  }
  return f;
}

// An inline-asm MFMA is opaque to the compiler's hazard recognizer: a vector instruction that writes one of its operands right in front
// of it (a conversion that sank, an accumulator-register read-back of a parked K fragment, a zeroing move) needs wait states nobody
// inserts — the MFMA then reads the register's previous contents.  Two wait states in front of every hand-written MFMA.
#define PIPE_PRE "s_nop 1\n\t"
#ifdef DCC_ATTN_PIPE_NOPS
#define PIPE_NOPS "\n\ts_nop 15\n\ts_nop 3"
#else
#define PIPE_NOPS ""
#endif
// NB: __builtin_bit_cast applied directly to a vector ELEMENT (v[i]) reads element 0 whatever i is (hipcc 7.2): go through a scalar
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
// first MFMA of a chain: C = 0 as an inline constant.  (Zeroing the accumulator with vector moves and accumulating into it needs wait
// states between the last move and the MFMA that the compiler cannot know to insert for an inline-asm MFMA: the MFMA then reads the
// registers' previous contents.)
__device__ __forceinline__ void mma16_vinit(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile(PIPE_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" PIPE_NOPS : "=&v"(acc) : "v"(a), "v"(b));
}
// ---- hand-placed pieces of the pipelined kernel's steady state (the compiler neither counts nor waits for these LDS reads)
template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}
template <int OFF> __device__ __forceinline__ u32x4 hlds_b128(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ u32x2 hlds_tr64(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// wait until at most N younger LDS operations are outstanding; the fragment is an operand so that its consumer stays behind the wait
template <int N> __device__ __forceinline__ void hwait(u32x4& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
// An MFMA chain whose LDS fragments are read AHEAD steps in front of their MFMA (hand-issued reads, counted waits): what hipcc emits
// for `acc = mma16(frag_from_lds(k), b[k], acc)` is read -> wait -> MFMA with at most one read in flight, i.e. an LDS latency per MFMA.
// issue(k) returns fragment k (OPS LDS operations each), mma(k, fragment) consumes it.
template <int N, int OPS, typename Issue, typename Mma>
__device__ __forceinline__ void hand_chain(Issue&& issue, Mma&& mma) {
  constexpr int AHEAD = 3;
  u32x4 fr[AHEAD + 1];
  static_for<(N < AHEAD ? N : AHEAD)>([&](auto kc) { fr[decltype(kc)::value % (AHEAD + 1)] = issue(kc); });
  static_for<N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int left = N - 1 - k;
    hwait<OPS * (left < AHEAD - 1 ? left : AHEAD - 1)>(fr[k % (AHEAD + 1)]);
    if constexpr (k + AHEAD < N) fr[(k + AHEAD) % (AHEAD + 1)] = issue(std::integral_constant<int, k + AHEAD>{});
    mma(kc, __builtin_bit_cast(bf16x8, fr[k % (AHEAD + 1)]));
    __builtin_amdgcn_sched_barrier(0);
  });
}

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }

// store a transposed accumulator tile: value for (row = n, col = 32*t + acc_row(reg,h)) lives in
// acc[reg] of the lane with (lane&31) == n - n0.  4 consecutive regs = 4 consecutive columns.
// PAIR = false keeps four 8-byte stores: measured faster in the DQK = 160 backward kernels (block 1: 1 401 vs 1 452 us), whose five
// column tiles leave no registers for the packed copies
template <typename T, bool PAIR = true>
__device__ __forceinline__ void store_tile_T(T* rowptr, int t, int h, const f32x16& acc, float mul,
                                             const T* resptr) {
  if constexpr (sizeof(T) == 2 && PAIR) {
    // bf16: the lane pair (l, l + 32) holds columns 8 g4 + 4 h .. + 3 of the same row; after a half exchange per pair of column
    // groups the lower lane owns columns 16 P .. + 7 and the upper lane 16 P + 8 .. + 15: two 16-byte stores per tile instead of four
    // 8-byte ones (both lanes of a pair are always active together: the callers' row guard depends on lane & 31 only)
    uint32_t pk[8];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int col = 32 * t + 8 * g4 + 4 * h;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[4 * g4 + e] * mul;
      if (resptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += to_f32(resptr[col + e]);
      }
      bf16x2 lo, hi;
      lo[0] = (bf16_t)v[0]; lo[1] = (bf16_t)v[1]; hi[0] = (bf16_t)v[2]; hi[1] = (bf16_t)v[3];
      pk[2 * g4] = __builtin_bit_cast(uint32_t, lo);
      pk[2 * g4 + 1] = __builtin_bit_cast(uint32_t, hi);
    }
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * P], pk[4 * P + 2], false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * P + 1], pk[4 * P + 3], false, false);
      const u32x4 c = {s0[0], s1[0], s0[1], s1[1]};
      *reinterpret_cast<u32x4*>(rowptr + 32 * t + 16 * P + 8 * h) = c;
    }
  } else {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int col = 32 * t + 8 * g4 + 4 * h;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[4 * g4 + e] * mul;
      if (resptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += to_f32(resptr[col + e]);
      }
      if constexpr (sizeof(T) == 2) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x4*>(rowptr + col) = o;
      } else {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(rowptr + col) = o;
      }
    }
  }
}

// XCD-aware block order.  The hardware deals workgroups (x fastest) round-robin to the 8 XCDs, so the q tiles of one
// (clip, head) — which all read that head's K / V — and the key blocks of the backward — which all stream that head's Q / dO —
// land on 8 different L2s and every one of them fetches the shared operand (PMC: 300 MB per launch for ~110 MB of tensors).
// Logical ids give each XCD a contiguous range of (x, head, clip) triples, x fastest: the sharers run on one XCD back to back.
struct Blk3 { int x, y, z; };
__device__ __forceinline__ Blk3 xcd_blk3() {
#ifdef DCC_ATTN_NO_XCD
  return Blk3{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
#else
  const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
  const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const unsigned xcd = L & 7, q = total >> 3, r = total & 7;
  const unsigned lb = xcd * q + (xcd < r ? xcd : r) + (L >> 3);
  const unsigned t = lb / gx;
  return Blk3{(int)(lb - t * gx), (int)(t % gy), (int)(t / gy)};
#endif
}

// ------------------------------------------------------------------------------------------
// forward: one wave = 32 queries, block = 128 queries; keys/values staged 64 rows at a time.
template <typename T, int DQK, int DV, bool DROP>
__global__ __launch_bounds__(ANT, DCC_ATTN_OCC_FWD) void attn_fwd_kernel(AttnParams p) {
  using KT = RowTile<T, DQK>;
  using VT = RowTile<T, DV>;
  __shared__ __attribute__((aligned(16))) char smem[TK * KT::PITCH + TK * VT::PITCH + TK * 4];
  char* ldsK = smem;
  char* ldsV = smem + TK * KT::PITCH;
  float* ldsB = reinterpret_cast<float*>(smem + TK * KT::PITCH + TK * VT::PITCH);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const Blk3 blk = xcd_blk3();
  const int b = blk.z, hd = blk.y;
  const int q0 = blk.x * 128 + wave * 32;
  const int myq = q0 + (lane & 31);
  const T* Q = reinterpret_cast<const T*>(p.Q) + b * p.sqb + hd * p.sqh;
  const T* K = reinterpret_cast<const T*>(p.K) + b * p.skb + hd * p.skh;
  const T* V = reinterpret_cast<const T*>(p.V) + b * p.svb + hd * p.svh;

  frag_t<T> qf[DQK / 16];
#pragma unroll
  for (int s = 0; s < DQK / 16; ++s) qf[s] = gfrag<T>(Q, p.sqn, myq, p.Nq, s, h);

  f32x16 oacc[DV / 32];
#pragma unroll
  for (int t = 0; t < DV / 32; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[t][e] = 0.0f;
  float m = -INFINITY, l = 0.0f;
  const float scale2 = p.scale * LOG2E;
  DropCtx dc;
  if constexpr (DROP) dc.init(p);
  const uint32_t drow = (uint32_t)((b * p.H + hd) * p.Nq + myq);

  TileRegs<T, DQK> rk;
  TileRegs<T, DV> rv;
  const __amdgpu_buffer_rsrc_t rsK = tile_rsrc<T, DQK>(K, p.skn, p.Nk), rsV = tile_rsrc<T, DV>(V, p.svn, p.Nk);
  float rbias = -INFINITY;
  auto prefetch = [&](int kt0) {
    tile_load<T, DQK>(rk, rsK, (int)p.skn * (int)sizeof(T), kt0, tid);
    tile_load<T, DV>(rv, rsV, (int)p.svn * (int)sizeof(T), kt0, tid);
    if (tid < TK) {
      const int key = kt0 + tid;
      rbias = -INFINITY;
      if (key < p.Nk) rbias = p.kbias ? p.kbias[(int64_t)b * p.Nk + key] * LOG2E : 0.0f;
    }
  };
  prefetch(0);
  for (int kt0 = 0; kt0 < p.Nk; kt0 += TK) {
    __syncthreads();  // previous tile's fragment reads are done
    tile_store<T, DQK>(rk, ldsK, tid);
    tile_store<T, DV>(rv, ldsV, tid);
    if (tid < TK) ldsB[tid] = rbias;
    __syncthreads();
    if (kt0 + TK < p.Nk) prefetch(kt0 + TK);  // in flight during this tile's MFMAs
#pragma unroll
    for (int sub = 0; sub < TK / 32; ++sub) {
      if (kt0 + 32 * sub >= p.Nk) break;
      f32x16 S;
#pragma unroll
      for (int e = 0; e < 16; ++e) S[e] = 0.0f;
#pragma unroll
      for (int s = 0; s < DQK / 16; ++s) S = mma16(row_frag<T, DQK>(ldsK, 32 * sub, s, lane), qf[s], S);
      float tmax = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        S[e] = S[e] * scale2 + ldsB[32 * sub + acc_row(e, h)];
        tmax = fmaxf(tmax, S[e]);
      }
      tmax = fmaxf(tmax, xhalf(tmax));
      const float mnew = fmaxf(m, tmax);
      const float msafe = (mnew == -INFINITY) ? 0.0f : mnew;
      const float alpha = fast_exp2(m - msafe);  // m = -inf -> 0
      float rs = 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        S[e] = fast_exp2(S[e] - msafe);
        rs += S[e];
      }
      rs += xhalf(rs);
      l = l * alpha + rs;
      m = mnew;
      if constexpr (DROP) {  // mask the tile that feeds P V; l keeps the undropped sum (softmax, THEN dropout)
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] *= dc.mul(drow, (uint32_t)(kt0 + 32 * sub + acc_row(e, h)));
      }
#pragma unroll
      for (int t = 0; t < DV / 32; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[t][e] *= alpha;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const frag_t<T> pf = acc_as_frag<T>(S, s2);
#pragma unroll
        for (int t = 0; t < DV / 32; ++t)
          oacc[t] = mma16(tr_frag<T, DV>(ldsV, 32 * sub, s2, 32 * t, lane), pf, oacc[t]);
      }
    }
  }

  if (myq < p.Nq) {
    const float inv = (l > 0.0f) ? 1.0f / l : 0.0f;
    T* orow = reinterpret_cast<T*>(p.O) + b * p.sob + hd * p.soh + (int64_t)myq * p.son;
    const T* rrow = nullptr;
    if (p.Qres && myq >= p.res_start)
      rrow = reinterpret_cast<const T*>(p.Qres) + b * p.sqb + hd * p.sqh + (int64_t)myq * p.sqn;
#pragma unroll
    for (int t = 0; t < DV / 32; ++t) store_tile_T<T>(orow, t, h, oacc[t], inv, rrow);
    if (h == 0 && p.LSE) p.LSE[((int64_t)b * p.H + hd) * p.Nq + myq] = (m + log2f(l)) * LN2;
  }
}

// ------------------------------------------------------------------------------------------
// dQ: same decomposition as the forward (queries on lanes).
template <typename T, int DQK, int DV, bool DROP>
// DQK = 160 needs 266 registers left alone (one wave per SIMD); capped at 256 it spills 14 dwords outside the key loop and runs two:
// measured -7 ... -9 % on the three D = 160 shapes of the step (the D = 128 kernels, 238 registers, already hold two)
__global__ __launch_bounds__(ANT, (DQK >= 160 && sizeof(T) == 2 && !DROP ? 2 : DCC_ATTN_OCC_DQ)) void attn_bwd_dq_kernel(AttnParams p) {
  using KT = RowTile<T, DQK>;
  using VT = RowTile<T, DV>;
  constexpr bool HANDQ = DCC_ATTN_HAND_DQ && sizeof(T) == 2 && !DROP && DQK >= 128;   // MFMA chains with their LDS reads three steps ahead (-1 %; +0.6 % at DQK = 96: off there)
  __shared__ __attribute__((aligned(16))) char smem[TK * KT::PITCH + TK * VT::PITCH + TK * 4];
  char* ldsK = smem;
  char* ldsV = smem + TK * KT::PITCH;
  float* ldsB = reinterpret_cast<float*>(smem + TK * KT::PITCH + TK * VT::PITCH);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const Blk3 blk = xcd_blk3();
  const int b = blk.z, hd = blk.y;
  const int q0 = blk.x * 128 + wave * 32;
  const int myq = q0 + (lane & 31);
  const T* Q = reinterpret_cast<const T*>(p.Q) + b * p.sqb + hd * p.sqh;
  const T* K = reinterpret_cast<const T*>(p.K) + b * p.skb + hd * p.skh;
  const T* V = reinterpret_cast<const T*>(p.V) + b * p.svb + hd * p.svh;
  const T* dO = reinterpret_cast<const T*>(p.dO) + b * p.sob + hd * p.soh;

  frag_t<T> qf[DQK / 16], dof[DV / 16];
#pragma unroll
  for (int s = 0; s < DQK / 16; ++s) qf[s] = gfrag<T>(Q, p.sqn, myq, p.Nq, s, h);
#pragma unroll
  for (int s = 0; s < DV / 16; ++s) dof[s] = gfrag<T>(dO, p.son, myq, p.Nq, s, h);
  // D[q] = sum_d dO[q][d] * (O[q][d] - [q >= res_start] Qres[q][d]): the rows are already here as fragments (the residual
  // shortcut adds the pooled query, i.e. the first DV columns of this Q row), so the separate preparation kernel is gone;
  // the value is also stored for the dK/dV kernel, which runs after this one on the same stream.
  float lse2 = INFINITY, dv = 0.0f;
  {
    const T* O = reinterpret_cast<const T*>(p.O) + b * p.sob + hd * p.soh;
    const bool res = p.Qres != nullptr && myq >= p.res_start;
    const bool same = p.Qres == p.Q;  // the usual case: the residual is the pooled query itself
    const T* QR = reinterpret_cast<const T*>(p.Qres) + b * p.sqb + hd * p.sqh;
#pragma unroll
    for (int s = 0; s < DV / 16; ++s) {
      const frag_t<T> of = gfrag<T>(O, p.son, myq, p.Nq, s, h);
      frag_t<T> rf = qf[s < DQK / 16 ? s : 0];
      if (res && !same) rf = gfrag<T>(QR, p.sqn, myq, p.Nq, s, h);
#pragma unroll
      for (int e = 0; e < 8; ++e) dv += to_f32(dof[s][e]) * (to_f32(of[e]) - (res ? to_f32(rf[e]) : 0.0f));
    }
    dv += xhalf(dv);
  }
  if (myq < p.Nq) {
    const int64_t o = ((int64_t)b * p.H + hd) * p.Nq + myq;
    lse2 = p.LSE[o] * LOG2E;
    if (h == 0) const_cast<float*>(p.Dv)[o] = dv;
  }
  f32x16 dq[DQK / 32];
#pragma unroll
  for (int t = 0; t < DQK / 32; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[t][e] = 0.0f;
  const float scale2 = p.scale * LOG2E;
  DropCtx dc;
  if constexpr (DROP) dc.init(p);
  const uint32_t drow = (uint32_t)((b * p.H + hd) * p.Nq + myq);

  TileRegs<T, DQK> rk;
  TileRegs<T, DV> rv;
  const __amdgpu_buffer_rsrc_t rsK = tile_rsrc<T, DQK>(K, p.skn, p.Nk), rsV = tile_rsrc<T, DV>(V, p.svn, p.Nk);
  float rbias = -INFINITY;
  auto prefetch = [&](int kt0) {
    tile_load<T, DQK>(rk, rsK, (int)p.skn * (int)sizeof(T), kt0, tid);
    tile_load<T, DV>(rv, rsV, (int)p.svn * (int)sizeof(T), kt0, tid);
    if (tid < TK) {
      const int key = kt0 + tid;
      rbias = -INFINITY;
      if (key < p.Nk) rbias = p.kbias ? p.kbias[(int64_t)b * p.Nk + key] * LOG2E : 0.0f;
    }
  };
  prefetch(0);
  for (int kt0 = 0; kt0 < p.Nk; kt0 += TK) {
    __syncthreads();
    tile_store<T, DQK>(rk, ldsK, tid);
    tile_store<T, DV>(rv, ldsV, tid);
    if (tid < TK) ldsB[tid] = rbias;
    __syncthreads();
    if (kt0 + TK < p.Nk) prefetch(kt0 + TK);
#pragma unroll
    for (int sub = 0; sub < TK / 32; ++sub) {
      if (kt0 + 32 * sub >= p.Nk) break;
      f32x16 S, dP;
#pragma unroll
      for (int e = 0; e < 16; ++e) { S[e] = 0.0f; dP[e] = 0.0f; }
      if constexpr (HANDQ) {
        const uint32_t aK = (uint32_t)(uintptr_t)LDS_PTR(const char, ldsK) + (32 * sub + (lane & 31)) * KT::PITCH + 16 * h;
        const uint32_t aV = (uint32_t)(uintptr_t)LDS_PTR(const char, ldsV) + (32 * sub + (lane & 31)) * VT::PITCH + 16 * h;
        hand_chain<DQK / 16, 1>([&](auto sc) { return hlds_b128<32 * decltype(sc)::value>(aK); },
                                [&](auto sc, const bf16x8& a) { S = mma16(a, qf[decltype(sc)::value], S); });
        hand_chain<DV / 16, 1>([&](auto sc) { return hlds_b128<32 * decltype(sc)::value>(aV); },
                               [&](auto sc, const bf16x8& a) { dP = mma16(a, dof[decltype(sc)::value], dP); });
      } else {
#pragma unroll
        for (int s = 0; s < DQK / 16; ++s) S = mma16(row_frag<T, DQK>(ldsK, 32 * sub, s, lane), qf[s], S);
#pragma unroll
        for (int s = 0; s < DV / 16; ++s) dP = mma16(row_frag<T, DV>(ldsV, 32 * sub, s, lane), dof[s], dP);
      }
      if constexpr (DROP) {  // dP = keep/(1-p) * (dO V^T)
#pragma unroll
        for (int e = 0; e < 16; ++e) dP[e] *= dc.mul(drow, (uint32_t)(kt0 + 32 * sub + acc_row(e, h)));
      }
      const bool plain = !p.kbias && (kt0 + 32 * sub + 32 <= p.Nk);  // no key bias, no tail: skip the bias reads
      if (plain) {
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = fast_exp2(S[e] * scale2 - lse2) * (dP[e] - dv);  // dz^T
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pr = fast_exp2(S[e] * scale2 + ldsB[32 * sub + acc_row(e, h)] - lse2);
          S[e] = pr * (dP[e] - dv);  // dz^T
        }
      }
      if constexpr (HANDQ) {
        const frag_t<T> zf2[2] = {acc_as_frag<T>(S, 0), acc_as_frag<T>(S, 1)};
        const int i16 = lane & 15;
        const uint32_t trK = (uint32_t)(uintptr_t)LDS_PTR(const char, ldsK) + (32 * sub + 4 * h + (i16 >> 2)) * KT::PITCH +
                             (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;
        constexpr int NT = DQK / 32;
        hand_chain<2 * NT, 2>(
            [&](auto kc) {
              constexpr int k = decltype(kc)::value, s2 = k / NT, t = k % NT;
              const u32x2 lo = hlds_tr64<(16 * s2) * KT::PITCH + 64 * t>(trK), hi = hlds_tr64<(16 * s2 + 8) * KT::PITCH + 64 * t>(trK);
              return u32x4{lo[0], lo[1], hi[0], hi[1]};
            },
            [&](auto kc, const bf16x8& a) {
              constexpr int k = decltype(kc)::value;
              dq[k % NT] = mma16(a, zf2[k / NT], dq[k % NT]);
            });
      } else {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const frag_t<T> zf = acc_as_frag<T>(S, s2);
#pragma unroll
          for (int t = 0; t < DQK / 32; ++t)
            dq[t] = mma16(tr_frag<T, DQK>(ldsK, 32 * sub, s2, 32 * t, lane), zf, dq[t]);
        }
      }
    }
  }
  if (myq < p.Nq) {
    T* drow = reinterpret_cast<T*>(p.dQ) + b * p.sdqb + hd * p.sdqh + (int64_t)myq * p.sdqn;
    // residual shortcut O += Qf[:, :DV] (q >= res_start): its gradient dO lands in the first DV columns of dQ
    const T* rrow = (p.Qres && myq >= p.res_start) ? dO + (int64_t)myq * p.son : nullptr;
#pragma unroll
    for (int t = 0; t < DQK / 32; ++t) store_tile_T<T, (DQK <= 128)>(drow, t, h, dq[t], p.scale, (32 * t < DV) ? rrow : nullptr);
  }
}

// ------------------------------------------------------------------------------------------
// dK/dV: keys on lanes (one wave = KW tiles of 32 keys, block = 128 KW keys), queries staged 64 rows at a time.
// blockIdx.x = key block * qsplit + query split.
// KW = 2 (experiment, only instantiated with -DDCC_ATTN_KW2 and without this file's -amdgpu-mfma-vgpr-form flag, which cannot place
// its accumulators): every Q / dO fragment read from LDS feeds two MFMAs, halving the LDS bytes per MFMA, at the price of a doubled
// accumulator / K,V fragment set — 420+ registers with spills in the loop at DQK = 128; never faster.  The shipped form is KW = 1.
template <typename T, typename TO, int DQK, int DV, bool DROP, int KW>
__global__ __launch_bounds__(ANT, (KW > 1 ? 1 : DCC_ATTN_OCC_DKV)) void attn_bwd_dkv_kernel(AttnParams p) {
  using QT = RowTile<T, DQK>;
  using OT = RowTile<T, DV>;
  constexpr int TPITCH = DQK + 4;  // fp32 pitch of the per-wave transposition tile of the split epilogue
  constexpr int STAGE_BYTES = TK * QT::PITCH + TK * OT::PITCH + 2 * TK * 4;
  constexpr int TRANS_BYTES = sizeof(TO) == 4 ? 4 * 32 * TPITCH * 4 : 0;
  __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES > TRANS_BYTES ? STAGE_BYTES : TRANS_BYTES];
  char* ldsQ = smem;
  char* ldsO = smem + TK * QT::PITCH;
  float* ldsL = reinterpret_cast<float*>(smem + TK * QT::PITCH + TK * OT::PITCH);
  float* ldsD = ldsL + TK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const Blk3 blk = xcd_blk3();
  const int b = blk.z, hd = blk.y;
  const int kblk = blk.x / p.qsplit, qs = blk.x % p.qsplit;
  const int k0 = kblk * (128 * KW) + wave * (32 * KW);   // tile kw of this wave starts at k0 + 32 kw
  const T* Q = reinterpret_cast<const T*>(p.Q) + b * p.sqb + hd * p.sqh;
  const T* K = reinterpret_cast<const T*>(p.K) + b * p.skb + hd * p.skh;
  const T* V = reinterpret_cast<const T*>(p.V) + b * p.svb + hd * p.svh;
  const T* dO = reinterpret_cast<const T*>(p.dO) + b * p.sob + hd * p.soh;
  const float* LSE = p.LSE + ((int64_t)b * p.H + hd) * p.Nq;
  const float* Dv = p.Dv + ((int64_t)b * p.H + hd) * p.Nq;

  frag_t<T> kf[KW][DQK / 16], vf[KW][DV / 16];
  float kb2[KW];
  f32x16 dk[KW][DQK / 32], dvv[KW][DV / 32];
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) {
    const int mykey = k0 + 32 * kw + (lane & 31);
#pragma unroll
    for (int s = 0; s < DQK / 16; ++s) kf[kw][s] = gfrag<T>(K, p.skn, mykey, p.Nk, s, h);
#pragma unroll
    for (int s = 0; s < DV / 16; ++s) vf[kw][s] = gfrag<T>(V, p.svn, mykey, p.Nk, s, h);
    kb2[kw] = -INFINITY;
    if (mykey < p.Nk) kb2[kw] = p.kbias ? p.kbias[(int64_t)b * p.Nk + mykey] * LOG2E : 0.0f;
#pragma unroll
    for (int t = 0; t < DQK / 32; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) dk[kw][t][e] = 0.0f;
#pragma unroll
    for (int t = 0; t < DV / 32; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) dvv[kw][t][e] = 0.0f;
  }
  const float scale2 = p.scale * LOG2E;

  DropCtx dc;
  if constexpr (DROP) dc.init(p);
  const uint32_t drow0 = (uint32_t)((b * p.H + hd) * p.Nq);

  // query range of this split, in units of TK rows
  const int qtiles = (p.Nq + TK - 1) / TK;
  const int per = (qtiles + p.qsplit - 1) / p.qsplit;
  const int qt_begin = qs * per;
  int qt_end = qt_begin + per; if (qt_end > qtiles) qt_end = qtiles;

  TileRegs<T, DQK> rq;
  TileRegs<T, DV> ro;
  const __amdgpu_buffer_rsrc_t rsQ = tile_rsrc<T, DQK>(Q, p.sqn, p.Nq), rsO = tile_rsrc<T, DV>(dO, p.son, p.Nq);
  float rl = INFINITY, rd = 0.0f;
  auto prefetch = [&](int qq0) {
    tile_load<T, DQK>(rq, rsQ, (int)p.sqn * (int)sizeof(T), qq0, tid);
    tile_load<T, DV>(ro, rsO, (int)p.son * (int)sizeof(T), qq0, tid);
    if (tid < TK) {
      const int q = qq0 + tid;
      rl = (q < p.Nq) ? LSE[q] * LOG2E : INFINITY;
      rd = (q < p.Nq) ? Dv[q] : 0.0f;
    }
  };
  if (qt_begin < qt_end) prefetch(qt_begin * TK);
  for (int qt = qt_begin; qt < qt_end; ++qt) {
    const int qq0 = qt * TK;
    __syncthreads();
    tile_store<T, DQK>(rq, ldsQ, tid);
    tile_store<T, DV>(ro, ldsO, tid);
    if (tid < TK) { ldsL[tid] = rl; ldsD[tid] = rd; }
    __syncthreads();
    if (qt + 1 < qt_end) prefetch(qq0 + TK);
#pragma unroll
    for (int sub = 0; sub < TK / 32; ++sub) {
      if (qq0 + 32 * sub >= p.Nq || k0 >= p.Nk) break;  // no queries left / this wave owns no valid key
      f32x16 S[KW], dP[KW];
#pragma unroll
      for (int kw = 0; kw < KW; ++kw)
#pragma unroll
        for (int e = 0; e < 16; ++e) { S[kw][e] = 0.0f; dP[kw][e] = 0.0f; }
      const bool two = KW > 1 && (k0 + 32 < p.Nk);  // the second key tile holds valid keys (wave-uniform)
      constexpr bool HANDKV = DCC_ATTN_HAND_DQ && KW == 1 && sizeof(T) == 2 && !DROP && DQK >= 128;   // LDS fragments three MFMAs ahead (hand_chain)
      if constexpr (HANDKV) {
        const uint32_t aQ = (uint32_t)(uintptr_t)LDS_PTR(const char, ldsQ) + (32 * sub + (lane & 31)) * QT::PITCH + 16 * h;
        const uint32_t aO = (uint32_t)(uintptr_t)LDS_PTR(const char, ldsO) + (32 * sub + (lane & 31)) * OT::PITCH + 16 * h;
        hand_chain<DQK / 16, 1>([&](auto sc) { return hlds_b128<32 * decltype(sc)::value>(aQ); },
                                [&](auto sc, const bf16x8& a) { S[0] = mma16(a, kf[0][decltype(sc)::value], S[0]); });
        hand_chain<DV / 16, 1>([&](auto sc) { return hlds_b128<32 * decltype(sc)::value>(aO); },
                               [&](auto sc, const bf16x8& a) { dP[0] = mma16(a, vf[0][decltype(sc)::value], dP[0]); });
      } else {
#pragma unroll
      for (int s = 0; s < DQK / 16; ++s) {
        const frag_t<T> a = row_frag<T, DQK>(ldsQ, 32 * sub, s, lane);
        S[0] = mma16(a, kf[0][s], S[0]);
        if constexpr (KW > 1) { if (two) S[KW - 1] = mma16(a, kf[KW - 1][s], S[KW - 1]); }
      }
#pragma unroll
      for (int s = 0; s < DV / 16; ++s) {
        const frag_t<T> a = row_frag<T, DV>(ldsO, 32 * sub, s, lane);
        dP[0] = mma16(a, vf[0][s], dP[0]);
        if constexpr (KW > 1) { if (two) dP[KW - 1] = mma16(a, vf[KW - 1][s], dP[KW - 1]); }
      }
      }
      // rows = queries, lanes = keys
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qi = 32 * sub + acc_row(e, h);
        const float lq = ldsL[qi], dq_ = ldsD[qi];
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const float pr = fast_exp2(S[kw][e] * scale2 + kb2[kw] - lq);
          if constexpr (DROP) {
            const float mk = dc.mul(drow0 + (uint32_t)(qq0 + qi), (uint32_t)(k0 + 32 * kw + (lane & 31)));
            S[kw][e] = pr * mk;                          // dropout(P): feeds dV
            dP[kw][e] = pr * (dP[kw][e] * mk - dq_);    // dz
          } else {
            S[kw][e] = pr;                           // P
            dP[kw][e] = pr * (dP[kw][e] - dq_);      // dz
          }
        }
      }
      if constexpr (HANDKV) {
        const frag_t<T> pf2[2] = {acc_as_frag<T>(S[0], 0), acc_as_frag<T>(S[0], 1)};
        const frag_t<T> zf2[2] = {acc_as_frag<T>(dP[0], 0), acc_as_frag<T>(dP[0], 1)};
        const int i16 = lane & 15;
        const uint32_t lrow = 32 * sub + 4 * h + (i16 >> 2), lcol = (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;
        const uint32_t trO = (uint32_t)(uintptr_t)LDS_PTR(const char, ldsO) + lrow * OT::PITCH + lcol;
        const uint32_t trQ = (uint32_t)(uintptr_t)LDS_PTR(const char, ldsQ) + lrow * QT::PITCH + lcol;
        constexpr int NV = DV / 32, NQ = DQK / 32;
        hand_chain<2 * NV, 2>(
            [&](auto kc) {
              constexpr int k = decltype(kc)::value, s2 = k / NV, t = k % NV;
              const u32x2 lo = hlds_tr64<(16 * s2) * OT::PITCH + 64 * t>(trO), hi = hlds_tr64<(16 * s2 + 8) * OT::PITCH + 64 * t>(trO);
              return u32x4{lo[0], lo[1], hi[0], hi[1]};
            },
            [&](auto kc, const bf16x8& a) { constexpr int k = decltype(kc)::value; dvv[0][k % NV] = mma16(a, pf2[k / NV], dvv[0][k % NV]); });
        hand_chain<2 * NQ, 2>(
            [&](auto kc) {
              constexpr int k = decltype(kc)::value, s2 = k / NQ, t = k % NQ;
              const u32x2 lo = hlds_tr64<(16 * s2) * QT::PITCH + 64 * t>(trQ), hi = hlds_tr64<(16 * s2 + 8) * QT::PITCH + 64 * t>(trQ);
              return u32x4{lo[0], lo[1], hi[0], hi[1]};
            },
            [&](auto kc, const bf16x8& a) { constexpr int k = decltype(kc)::value; dk[0][k % NQ] = mma16(a, zf2[k / NQ], dk[0][k % NQ]); });
      } else {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        frag_t<T> pf[KW], zf[KW];
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) { pf[kw] = acc_as_frag<T>(S[kw], s2); zf[kw] = acc_as_frag<T>(dP[kw], s2); }
#pragma unroll
        for (int t = 0; t < DV / 32; ++t) {
          const frag_t<T> a = tr_frag<T, DV>(ldsO, 32 * sub, s2, 32 * t, lane);
          dvv[0][t] = mma16(a, pf[0], dvv[0][t]);
          if constexpr (KW > 1) { if (two) dvv[KW - 1][t] = mma16(a, pf[KW - 1], dvv[KW - 1][t]); }
        }
#pragma unroll
        for (int t = 0; t < DQK / 32; ++t) {
          const frag_t<T> a = tr_frag<T, DQK>(ldsQ, 32 * sub, s2, 32 * t, lane);
          dk[0][t] = mma16(a, zf[0], dk[0][t]);
          if constexpr (KW > 1) { if (two) dk[KW - 1][t] = mma16(a, zf[KW - 1], dk[KW - 1][t]); }
        }
      }
      }
    }
  }

  if (p.qsplit == 1) {
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
      const int mykey = k0 + 32 * kw + (lane & 31);
      if (mykey < p.Nk) {
        TO* dkrow = reinterpret_cast<TO*>(p.dK) + b * p.sdkb + hd * p.sdkh + (int64_t)mykey * p.sdkn;
        TO* dvrow = reinterpret_cast<TO*>(p.dV) + b * p.sdvb + hd * p.sdvh + (int64_t)mykey * p.sdvn;
#pragma unroll
        for (int t = 0; t < DQK / 32; ++t) store_tile_T<TO, (DQK <= 128)>(dkrow, t, h, dk[kw][t], p.scale, nullptr);
#pragma unroll
        for (int t = 0; t < DV / 32; ++t) store_tile_T<TO, (DQK <= 128)>(dvrow, t, h, dvv[kw][t], 1.0f, nullptr);
      }
    }
  } else if constexpr (sizeof(TO) == 4) {
    // query-split partial sums: transpose each wave's [32 keys][D] tile through LDS so that one atomic
    // wave-instruction covers 256 contiguous bytes of ONE row (row-per-lane atomics run ~17x slower)
    __syncthreads();  // all waves are done with the staged Q / dO tiles
    float* tw = reinterpret_cast<float*>(smem) + wave * (32 * TPITCH);
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
      const int kk0 = k0 + 32 * kw;
      // dK
#pragma unroll
      for (int t = 0; t < DQK / 32; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) tw[(lane & 31) * TPITCH + 32 * t + acc_row(e, h)] = dk[kw][t][e] * p.scale;
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
      for (int r = 0; r < 32; ++r) {
        const int key = kk0 + r;
        if (key >= p.Nk) break;
        float* dst = reinterpret_cast<float*>(p.dK) + b * p.sdkb + hd * p.sdkh + (int64_t)key * p.sdkn;
        for (int c = lane; c < DQK; c += 64) atomicAdd(dst + c, tw[r * TPITCH + c]);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      // dV
#pragma unroll
      for (int t = 0; t < DV / 32; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) tw[(lane & 31) * TPITCH + 32 * t + acc_row(e, h)] = dvv[kw][t][e];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      for (int r = 0; r < 32; ++r) {
        const int key = kk0 + r;
        if (key >= p.Nk) break;
        float* dst = reinterpret_cast<float*>(p.dV) + b * p.sdvb + hd * p.sdvh + (int64_t)key * p.sdvn;
        for (int c = lane; c < DV; c += 64) atomicAdd(dst + c, tw[r * TPITCH + c]);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
  }
}

// ------------------------------------------------------------------------------------------
// dK/dV, software-pipelined (bf16, no dropout).  Same decomposition as attn_bwd_dkv_kernel — one wave = 32 keys, K / V fragments and
// the dK / dV accumulators in registers — but a lone wave per SIMD no longer runs its three phases back to back:
//   A(j)  S = Q_j K^T, dP = dO_j V^T          MFMA, row fragments from LDS
//   B(j)  P = exp2(..), dz = P (dP - D)        VALU
//   C(j)  dV += P^T dO_j, dK += dz^T Q_j       MFMA, transposed fragments from LDS
// Iteration j issues C(j-1), B(j) and A(j+1): the MFMAs of two neighbouring query tiles are independent of the softmax arithmetic
// of the one between them, so the matrix pipe works while the VALU does exponentials.  Query tiles are 32 rows, four LDS buffers
// (tiles j-1, j, j+1 live, j+2 being written): one barrier per tile.
// scheduling groups (llvm.amdgcn.sched.group.barrier masks: 0x008 MFMA, 0x100 LDS read, 0x002 VALU)
template <int N> __device__ __forceinline__ void sched_prime() {
  if constexpr (N > 0) { __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); sched_prime<N - 1>(); }
}
template <int N, int NDS, int NVALU> __device__ __forceinline__ void sched_slots() {
  if constexpr (N > 0) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
    sched_slots<N - 1, NDS, NVALU>();
  }
}
#ifndef DCC_PIPE_DBG
#define DCC_PIPE_DBG 0
#endif
#ifndef PIPE_AHEAD
#define PIPE_AHEAD 3   // fragments in flight ahead of the MFMA that consumes them
#endif
#ifndef PIPE_VALU_C
#define PIPE_VALU_C 5
#endif
#ifndef PIPE_VALU_A
#define PIPE_VALU_A 3
#endif
// the same with the result retired when the statement ends: for the compiler-scheduled prologue / epilogue code, where the register
// allocator may copy an accumulator tuple right behind the MFMA (it does not know the statement's latency)
__device__ __forceinline__ void mma16_acc_retired(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile(PIPE_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mma16_vacc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile(PIPE_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" PIPE_NOPS : "+v"(acc) : "v"(a), "v"(b));
}
// dK / dV accumulate in AGPRs: the kernel is built with MFMA results in VGPRs (the softmax reads S and dP), and 112 more VGPRs
// of accumulators would leave no room for the second S / dP pair and the fragments in flight
__device__ __forceinline__ void mma16_acc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile(PIPE_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" PIPE_NOPS : "+a"(acc) : "v"(a), "v"(b));
}

constexpr int PT = 32;     // query rows per pipelined tile
#ifdef DCC_PIPE_TILE_BARRIER   // (A/B: the earlier schedule, one barrier per tile, four buffers)
constexpr int PNBUF = 4;
#else
constexpr int PNBUF = 8;   // barrier once per pair of tiles: tiles j-1 .. j+2 live, j+3 and j+4 being written
#endif

template <int DQK, int DV>
struct PipeGeo {
  using QT = RowTile<bf16_t, DQK>;
  using OT = RowTile<bf16_t, DV>;
  static constexpr int Q_BYTES = PT * QT::PITCH, O_BYTES = PT * OT::PITCH;
  // Optional second copy of the tile for the transposed reads.  A ds_read_b64_tr_b16 takes 4 consecutive rows x 64 bytes per 32
  // lanes: the rows must start 64 bytes apart modulo 256 (the 64 banks) to be conflict-free, while the row reads (16 rows x one
  // 16-byte column per 16 lanes) want a pitch of an odd number of 16-byte units — no single pitch serves both.  Measured: keeping the
  // tile twice costs more in LDS writes than the conflicts it removes (attention backward 6.20 vs 6.09 ms per step), so it is off.
#ifndef DCC_PIPE_DUAL_COPY
  static constexpr int QTR_PITCH = QT::PITCH, OTR_PITCH = OT::PITCH;
#else
  static constexpr int QTR_PITCH = ((DQK * 2 + 255) / 256) * 256 + 64 - ((DQK * 2) % 256 == 64 ? 256 : 0);
  static constexpr int OTR_PITCH = DV * 2 == 192 ? 192 : ((DV * 2 + 255) / 256) * 256 + 64;
  static_assert(QTR_PITCH >= DQK * 2 && QTR_PITCH % 256 == 64 && (OTR_PITCH % 256 == 64 || OTR_PITCH % 256 == 192), "transposed-read pitch");
#endif
#ifndef DCC_PIPE_DUAL_COPY   // default: transposed reads from the row copy (DCC_PIPE_DUAL_COPY: A/B builds of the second copy)
  static constexpr int QTR_OFF = 0, OTR_OFF = Q_BYTES, LD_OFF = Q_BYTES + O_BYTES;
#else
  static constexpr int QTR_OFF = Q_BYTES + O_BYTES, OTR_OFF = QTR_OFF + PT * QTR_PITCH, LD_OFF = OTR_OFF + PT * OTR_PITCH;
#endif
  static constexpr int TILE_BYTES = LD_OFF + 2 * PT * 4;
  static constexpr int TPITCH = DQK + 4;
  static constexpr int TRANS_BYTES = 4 * 32 * TPITCH * 4;
  static constexpr int LDS_BYTES = PNBUF * TILE_BYTES > TRANS_BYTES ? PNBUF * TILE_BYTES : TRANS_BYTES;
  static constexpr int NQ = PT * QT::CPR, NO = PT * OT::CPR;            // 16-byte chunks per tile
  static constexpr int RQ = (NQ + ANT - 1) / ANT, RO = (NO + ANT - 1) / ANT;  // chunks per thread
};

template <typename TO, int DQK, int DV>
__global__ __launch_bounds__(ANT) void attn_bwd_dkv_pipe_kernel(AttnParams p) {
  using T = bf16_t;
  using G = PipeGeo<DQK, DV>;
  using QT = typename G::QT;
  using OT = typename G::OT;
  extern __shared__ __attribute__((aligned(16))) char psmem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const Blk3 blk = xcd_blk3();
  const int b = blk.z, hd = blk.y;
  const int kblk = blk.x / p.qsplit, qs = blk.x % p.qsplit;
  const int k0 = kblk * 128 + wave * 32;
  const int mykey = k0 + (lane & 31);
  const bool active = k0 < p.Nk;  // wave-uniform: this wave owns at least one valid key
  const T* Q = reinterpret_cast<const T*>(p.Q) + b * p.sqb + hd * p.sqh;
  const T* K = reinterpret_cast<const T*>(p.K) + b * p.skb + hd * p.skh;
  const T* V = reinterpret_cast<const T*>(p.V) + b * p.svb + hd * p.svh;
  const T* dO = reinterpret_cast<const T*>(p.dO) + b * p.sob + hd * p.soh;
  const float* LSE = p.LSE + ((int64_t)b * p.H + hd) * p.Nq;
  const float* Dv = p.Dv + ((int64_t)b * p.H + hd) * p.Nq;

  frag_t<T> kf[DQK / 16], vf[DV / 16];
#pragma unroll
  for (int s = 0; s < DQK / 16; ++s) kf[s] = gfrag<T>(K, p.skn, mykey, p.Nk, s, h);
#pragma unroll
  for (int s = 0; s < DV / 16; ++s) vf[s] = gfrag<T>(V, p.svn, mykey, p.Nk, s, h);
  float kb2 = -INFINITY;
  if (mykey < p.Nk) kb2 = p.kbias ? p.kbias[(int64_t)b * p.Nk + mykey] * LOG2E : 0.0f;

  f32x16 dk[DQK / 32], dvv[DV / 32];
#pragma unroll
  for (int t = 0; t < DQK / 32; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[t][e] = 0.0f;
#pragma unroll
  for (int t = 0; t < DV / 32; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) dvv[t][e] = 0.0f;
  const float scale2 = p.scale * LOG2E;

  // query range of this split, in units of PT rows
  const int qtiles = (p.Nq + PT - 1) / PT;
  const int per = (qtiles + p.qsplit - 1) / p.qsplit;
  const int qt_begin = qs * per;
  int qt_end = qt_begin + per; if (qt_end > qtiles) qt_end = qtiles;
  const int n = qt_end - qt_begin;  // tiles of this block (may be <= 0)

  const __amdgpu_buffer_rsrc_t rsQ = tile_rsrc<T, DQK>(Q, p.sqn, p.Nq), rsO = tile_rsrc<T, DV>(dO, p.son, p.Nq);
  const int qstride = (int)p.sqn * 2, ostride = (int)p.son * 2;
  // Two register sets: a tile's loads are issued two iterations before its LDS writes (HBM / L2 latency under load is well over one
  // iteration).  Loads and LDS writes are branch-free — chunk ids past the tile are clamped (two threads then write the same bytes to
  // the same place), tiles past the block's range are clamped to its last tile and land in a buffer nobody reads — so that the tail of
  // an iteration is straight-line code and the compiler's vmcnt bookkeeping can leave the younger set's loads in flight (with
  // predicated loads it fell back to vmcnt(0) at every join).  (Hand-counted waits on inline-asm loads were tried: the register
  // allocator may copy a loaded register before the data has landed — it does at the loop's phis — so the loads stay builtins.)
  vec16_t<T> rq[2][G::RQ], ro[2][G::RO];
  float rl[2] = {0.0f, 0.0f};
  const float* ldsrc = (tid & 32) ? Dv : LSE;   // lanes 0..31 of a wave: LSE, 32..63: D
  auto gload = [&](auto setc, int j) {  // global -> registers of set SET, tile min(j, n - 1) of this block
    constexpr int SET = decltype(setc)::value;
    if (j > n - 1) j = n - 1;
    const int row0 = (qt_begin + j) * PT;
#pragma unroll
    for (int i = 0; i < G::RQ; ++i) {
      int id = tid + ANT * i; if (id > G::NQ - 1) id = G::NQ - 1;
      const int rr = id / QT::CPR, c = id % QT::CPR;
      rq[SET][i] = __builtin_bit_cast(vec16_t<T>, __builtin_amdgcn_raw_buffer_load_b128(rsQ, (row0 + rr) * qstride + c * 16, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < G::RO; ++i) {
      int id = tid + ANT * i; if (id > G::NO - 1) id = G::NO - 1;
      const int rr = id / OT::CPR, c = id % OT::CPR;
      ro[SET][i] = __builtin_bit_cast(vec16_t<T>, __builtin_amdgcn_raw_buffer_load_b128(rsO, (row0 + rr) * ostride + c * 16, 0, 0));
    }
    int q = row0 + (tid & (PT - 1)); if (q > p.Nq - 1) q = p.Nq - 1;
    rl[SET] = ldsrc[q];
  };
  auto lstore = [&](auto setc, int j) {  // registers of set SET -> LDS buffer of tile j (both copies)
    constexpr int SET = decltype(setc)::value;
    char* base = psmem + (j & (PNBUF - 1)) * G::TILE_BYTES;
#pragma unroll
    for (int i = 0; i < G::RQ; ++i) {
      int id = tid + ANT * i; if (id > G::NQ - 1) id = G::NQ - 1;
      const int rr = id / QT::CPR, c = id % QT::CPR;
      *reinterpret_cast<vec16_t<T>*>(base + rr * QT::PITCH + c * 16) = rq[SET][i];
      if constexpr (G::QTR_OFF != 0) *reinterpret_cast<vec16_t<T>*>(base + G::QTR_OFF + rr * G::QTR_PITCH + c * 16) = rq[SET][i];
    }
#pragma unroll
    for (int i = 0; i < G::RO; ++i) {
      int id = tid + ANT * i; if (id > G::NO - 1) id = G::NO - 1;
      const int rr = id / OT::CPR, c = id % OT::CPR;
      *reinterpret_cast<vec16_t<T>*>(base + G::Q_BYTES + rr * OT::PITCH + c * 16) = ro[SET][i];
      if constexpr (G::QTR_OFF != 0) *reinterpret_cast<vec16_t<T>*>(base + G::OTR_OFF + rr * G::OTR_PITCH + c * 16) = ro[SET][i];
    }
    // every wave writes the tile's 32 LSE (log2 units; +inf past Nq: P = 0) and 32 D values
    const int jj = j > n - 1 ? n - 1 : j;
    const bool valid = (qt_begin + jj) * PT + (tid & (PT - 1)) < p.Nq;
    const float v = rl[SET];
    reinterpret_cast<float*>(base + G::LD_OFF)[tid & 63] = (tid & 32) ? (valid ? v : 0.0f) : (valid ? v * LOG2E : INFINITY);
  };
  auto phaseA = [&](int j, f32x16& S, f32x16& dP) {
    const char* base = psmem + (j & (PNBUF - 1)) * G::TILE_BYTES;
#pragma unroll
    for (int e = 0; e < 16; ++e) { S[e] = 0.0f; dP[e] = 0.0f; }
#pragma unroll
    for (int s = 0; s < DQK / 16; ++s) S = mma16(row_frag<T, DQK>(base, 0, s, lane), kf[s], S);
#pragma unroll
    for (int s = 0; s < DV / 16; ++s) dP = mma16(row_frag<T, DV>(base + G::Q_BYTES, 0, s, lane), vf[s], dP);
  };
  auto phaseB = [&](int j, const f32x16& S, const f32x16& dP, frag_t<T>* pf, frag_t<T>* zf) {
    const float* ld = reinterpret_cast<const float*>(psmem + (j & (PNBUF - 1)) * G::TILE_BYTES + G::LD_OFF);
    f32x16 P, Z;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(ld + 8 * g + 4 * h);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(ld + PT + 8 * g + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = 4 * g + i;
        const float pr = fast_exp2(S[e] * scale2 + kb2 - l4[i]);
        P[e] = pr;
        Z[e] = pr * (dP[e] - d4[i]);
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) { pf[s2] = acc_as_frag<T>(P, s2); zf[s2] = acc_as_frag<T>(Z, s2); }
  };
  auto phaseC = [&](int j, const frag_t<T>* pf, const frag_t<T>* zf) {
    const char* base = psmem + (j & (PNBUF - 1)) * G::TILE_BYTES;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int t = 0; t < DV / 32; ++t) mma16_acc_retired(dvv[t], tr_frag<T, DV, G::OTR_PITCH>(base + G::OTR_OFF, 0, s2, 32 * t, lane), pf[s2]);
#pragma unroll
      for (int t = 0; t < DQK / 32; ++t) mma16_acc_retired(dk[t], tr_frag<T, DQK, G::QTR_PITCH>(base + G::QTR_OFF, 0, s2, 32 * t, lane), zf[s2]);
    }
  };

  // steady-state iteration j, by hand: C(j-1) with (pf, zf), B(j) on (S, dP) -> (pfn, zfn), A(j+1) -> (Sn, dPn)
  auto steady = [&](int j, f32x16& S, f32x16& dP, f32x16& Sn, f32x16& dPn, frag_t<T>* pf, frag_t<T>* zf, frag_t<T>* pfn, frag_t<T>* zfn) {
#ifdef DCC_ATTN_PIPE_PLAIN   // the same dataflow, compiler-scheduled (debugging aid)
    phaseC(j - 1, pf, zf);
    phaseB(j, S, dP, pfn, zfn);
    phaseA(j + 1, Sn, dPn);
    return;
#endif
        // NC + NA slots, one MFMA each: slot k issues the LDS reads of the fragment of slot k + AHEAD, waits for its own, then runs
        // its MFMA and its share of B(j)
        constexpr int NCH = DV / 32 + DQK / 32, NC = 2 * NCH, NA = DQK / 16 + DV / 16, NS = NC + NA, AHEAD = PIPE_AHEAD;
        const uint32_t lbase = (uint32_t)(uintptr_t)LDS_PTR(const char, psmem);
        const uint32_t bC = lbase + ((j - 1) & (PNBUF - 1)) * G::TILE_BYTES, bA = lbase + ((j + 1) & (PNBUF - 1)) * G::TILE_BYTES;
        const uint32_t bB = lbase + (j & (PNBUF - 1)) * G::TILE_BYTES + G::LD_OFF;
        const int i16 = lane & 15, tq = i16 >> 2, tpp = i16 & 3, tg = (lane >> 4) & 1;
        const uint32_t trQ = bC + G::QTR_OFF + (4 * h + tq) * G::QTR_PITCH + (16 * tg + 4 * tpp) * 2;
        const uint32_t trO = bC + G::OTR_OFF + (4 * h + tq) * G::OTR_PITCH + (16 * tg + 4 * tpp) * 2;
        const uint32_t rwQ = bA + (lane & 31) * QT::PITCH + 16 * h;
        const uint32_t rwO = bA + G::Q_BYTES + (lane & 31) * OT::PITCH + 16 * h;
        const uint32_t ldL = bB + 16 * h;
        u32x4 fr[AHEAD + 1];
        u32x4 l4[4], d4[4];
        auto issue = [&](auto kc) {   // LDS reads of slot k's fragment
          constexpr int k = decltype(kc)::value;
          if constexpr (k < NC) {
            constexpr int s2 = k / NCH, r = k % NCH;
            if constexpr (r < DV / 32) {
              const u32x2 lo = hlds_tr64<(16 * s2) * G::OTR_PITCH + 64 * r>(trO), hi = hlds_tr64<(16 * s2 + 8) * G::OTR_PITCH + 64 * r>(trO);
              fr[k % (AHEAD + 1)] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            } else {
              constexpr int t = r - DV / 32;
              const u32x2 lo = hlds_tr64<(16 * s2) * G::QTR_PITCH + 64 * t>(trQ), hi = hlds_tr64<(16 * s2 + 8) * G::QTR_PITCH + 64 * t>(trQ);
              fr[k % (AHEAD + 1)] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
          } else if constexpr (k < NS) {
            constexpr int a = k - NC;
            if constexpr (a < DQK / 16) fr[k % (AHEAD + 1)] = hlds_b128<32 * a>(rwQ);
            else fr[k % (AHEAD + 1)] = hlds_b128<32 * (a - DQK / 16)>(rwO);
          }
        };
        // LDS operations issued for slot k's fragment (what a wait on an older fragment must leave outstanding)
        auto nops = [](int k) constexpr { return k < NC ? 2 : (k < NS ? 1 : 0); };
        auto younger = [nops](int k) constexpr { int t = 0; for (int i = 1; i < AHEAD; ++i) t += nops(k + i); return t; };
        // S and dP are results of the previous iteration's hand-written MFMAs, whose latency the compiler does not know: nothing may
        // read S before this point (its chain ended six MFMAs before the end of that iteration) nor dP before slot 16 (below)
        asm volatile("" : "+v"(S));
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto gc) { constexpr int g = decltype(gc)::value; l4[g] = hlds_b128<32 * g>(ldL); d4[g] = hlds_b128<PT * 4 + 32 * g>(ldL); });
        static_for<AHEAD>([&](auto kc) { issue(kc); });
        static_for<NS>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          // wait first, then issue: at most 8 (L / D) + 2 * AHEAD operations are ever outstanding, within the 4-bit counter
          hwait<younger(k)>(fr[k % (AHEAD + 1)]);
          issue(std::integral_constant<int, k + AHEAD>{});
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 a = __builtin_bit_cast(bf16x8, fr[k % (AHEAD + 1)]);
          if constexpr (k < NC) {
            constexpr int s2 = k / NCH, r = k % NCH;
            if constexpr (r < DV / 32) mma16_acc(dvv[r], a, pf[s2]);
            else mma16_acc(dk[r - DV / 32], a, zf[s2]);
          } else {
            constexpr int q = k - NC;
            // by hand as well (a compiler-placed MFMA whose result is needed an iteration later sinks to the end of the block); the
            // vector reads of Sn / dPn are an iteration's worth of instructions away, far beyond the MFMA's result latency
            if constexpr (q == 0) mma16_vinit(Sn, a, kf[0]);
            else if constexpr (q < DQK / 16) mma16_vacc(Sn, a, kf[q]);
            else if constexpr (q == DQK / 16) mma16_vinit(dPn, a, vf[0]);
            else mma16_vacc(dPn, a, vf[q - DQK / 16]);
          }
          // B(j): P[k] in slot k < 16; dz pairs (and the bf16 fragment of P) in the eight slots from ZS; the bf16 fragments of dz in the
          // four slots from CS.  With 28+ slots (DQK = 128 / 160) the three groups follow each other; with 24 (DQK = 96) the dz pairs
          // start at slot 12, behind the P they need (pair i needs P[2i], P[2i+1], done by slot 2i + 1 <= 12 + i)
          constexpr bool VAL = DCC_PIPE_DBG != 1;
          constexpr int ZS = NS >= 28 ? 16 : NS - 12, CS = NS >= 28 ? 24 : NS - 4;
          if constexpr (VAL && k < 16) {
            if constexpr (k == 0) {   // the L / D reads are older than every fragment read: they have landed with fragment 0
#pragma unroll
              for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(l4[g]), "+v"(d4[g]));
            }
            S[k] = fast_exp2(S[k] * scale2 + kb2 - u2f(l4[k >> 2][k & 3]));
            asm volatile("" : "+v"(S[k]));
          }
          if constexpr (VAL && k >= ZS && k < ZS + 8) {
            constexpr int e = 2 * (k - ZS);
            if constexpr (k == ZS) asm volatile("" : "+v"(dP));
            dP[e] = S[e] * (dP[e] - u2f(d4[e >> 2][e & 3]));
            dP[e + 1] = S[e + 1] * (dP[e + 1] - u2f(d4[(e + 1) >> 2][(e + 1) & 3]));
            bf16x2 c; c[0] = (bf16_t)S[e]; c[1] = (bf16_t)S[e + 1];
            pfn[e >> 3][e & 7] = c[0]; pfn[e >> 3][(e & 7) + 1] = c[1];
            // pinned here: a conversion that sinks to just before the (inline-asm) MFMA reading it would need wait states the compiler
            // does not know to insert
            asm volatile("" : "+v"(dP[e]), "+v"(dP[e + 1]), "+v"(pfn[e >> 3]));
          }
          if constexpr (VAL && k >= CS && k < CS + 4) {
            constexpr int e0 = 4 * (k - CS);
#pragma unroll
            for (int i = 0; i < 4; ++i) zfn[(e0 + i) >> 3][(e0 + i) & 7] = (bf16_t)dP[e0 + i];
            asm volatile("" : "+v"(zfn[e0 >> 3]));
          }
          __builtin_amdgcn_sched_barrier(0);
        });
#if DCC_PIPE_DBG == 1
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        phaseB(j, S, dP, pfn, zfn);
#endif
        static_assert(NS >= 24, "B(j) is spread over at least 24 slots");
  };

  if (n > 0) {
    // prologue: tiles 0 and 1 into LDS, A(0); the loads of tile t are issued at the end of iteration t - 4 (set t & 1), its LDS writes
    // happen at the end of iteration t - 2
    constexpr std::integral_constant<int, 0> SET0{};
    constexpr std::integral_constant<int, 1> SET1{};
#ifdef DCC_PIPE_TILE_BARRIER
    gload(SET0, 0);
    gload(SET1, 1);
    lstore(SET0, 0);
    lstore(SET1, 1);
    gload(SET0, 2);
    gload(SET1, 3);
    __syncthreads();
    f32x16 S, dP, Sn, dPn;
    frag_t<T> pf[2], zf[2], pfn[2], zfn[2];
    // the tail of iteration j (set j & 1): tile j + 2 into LDS, loads of tile j + 4, barrier
    auto tail = [&](auto setc, int j) {
      lstore(setc, j + 2);
      gload(setc, j + 4);
      __syncthreads();
    };
    // j = 0: B(0), A(1)
    if (active) {
      phaseA(0, S, dP);
      phaseB(0, S, dP, pf, zf);
      if (n > 1) phaseA(1, S, dP);
    }
    tail(SET0, 0);
    // pairs (odd j first): the second iteration of a pair runs with the roles of (S, dP, pf, zf) and (Sn, dPn, pfn, zfn) swapped, so
    // nothing is copied
    int j = 1;
    for (; j + 2 < n; j += 2) {
      if (active) steady(j, S, dP, Sn, dPn, pf, zf, pfn, zfn);
      tail(SET1, j);
      if (active) steady(j + 1, Sn, dPn, S, dP, pfn, zfn, pf, zf);
      tail(SET0, j + 1);
    }
    constexpr bool LEFTOVER_TAIL = true;
#else
    // tiles 0 .. 3 into LDS (even tiles through set 0, odd through set 1), loads of tiles 4 and 5 in flight; then one barrier per PAIR
    // of iterations: the pair (j, j+1) reads tiles j-1 .. j+2, its tail writes tiles j+3, j+4 (loaded during the previous pair) and
    // issues the loads of j+5, j+6.  Eight buffers, so a wave still in iteration j never sees its tiles overwritten.
    gload(SET0, 0);
    gload(SET1, 1);
    lstore(SET0, 0);
    lstore(SET1, 1);
    gload(SET0, 2);
    gload(SET1, 3);
    lstore(SET0, 2);
    lstore(SET1, 3);
    gload(SET0, 4);
    gload(SET1, 5);
    __syncthreads();
    f32x16 S, dP, Sn, dPn;
    frag_t<T> pf[2], zf[2], pfn[2], zfn[2];
    if (active) {  // j = 0: B(0), A(1)
      phaseA(0, S, dP);
      phaseB(0, S, dP, pf, zf);
      if (n > 1) phaseA(1, S, dP);
    }
    // pairs (odd j first): the second iteration of a pair runs with the roles of (S, dP, pf, zf) and (Sn, dPn, pfn, zfn) swapped, so
    // nothing is copied
    int j = 1;
    for (; j + 2 < n; j += 2) {
      if (active) {
        steady(j, S, dP, Sn, dPn, pf, zf, pfn, zfn);
        steady(j + 1, Sn, dPn, S, dP, pfn, zfn, pf, zf);
      }
      lstore(SET0, j + 3);
      lstore(SET1, j + 4);
      gload(SET0, j + 5);
      gload(SET1, j + 6);
      __syncthreads();
    }
    constexpr bool LEFTOVER_TAIL = false;
    auto tail = [&](auto, int) {};
#endif
    if (j + 1 < n && active) {
      // one steady iteration left over (odd count): the same hand-placed iteration; its results land in the second register set
      // and are moved once (behind wait states: the compiler does not know the latency of the MFMAs that wrote them)
      steady(j, S, dP, Sn, dPn, pf, zf, pfn, zfn);
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(Sn), "+v"(dPn));
      S = Sn; dP = dPn;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) { pf[s2] = pfn[s2]; zf[s2] = zfn[s2]; }
    }
    if constexpr (LEFTOVER_TAIL) { if (j + 1 < n) tail(SET1, j); }
    // the hand-placed MFMAs' results are invisible to the compiler's hazard tracking: let the last ones retire before compiler-scheduled
    // vector code reads them
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (active) {
      if (n > 1) {  // j = n - 1: C(n-2), B(n-1)
        phaseC(n - 2, pf, zf);
        phaseB(n - 1, S, dP, pf, zf);
      }
      phaseC(n - 1, pf, zf);
    }
    // dK / dV come out of hand-written MFMAs: their last results must have retired before the stores below read the accumulators
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }

  if (p.qsplit == 1) {
    if (mykey < p.Nk) {
      TO* dkrow = reinterpret_cast<TO*>(p.dK) + b * p.sdkb + hd * p.sdkh + (int64_t)mykey * p.sdkn;
      TO* dvrow = reinterpret_cast<TO*>(p.dV) + b * p.sdvb + hd * p.sdvh + (int64_t)mykey * p.sdvn;
#pragma unroll
      for (int t = 0; t < DQK / 32; ++t) store_tile_T<TO, (DQK <= 128)>(dkrow, t, h, dk[t], p.scale, nullptr);
#pragma unroll
      for (int t = 0; t < DV / 32; ++t) store_tile_T<TO, (DQK <= 128)>(dvrow, t, h, dvv[t], 1.0f, nullptr);
    }
  } else if constexpr (sizeof(TO) == 4) {
    constexpr int TPITCH = G::TPITCH;
    __syncthreads();  // all waves are done with the staged tiles
    float* tw = reinterpret_cast<float*>(psmem) + wave * (32 * TPITCH);
#pragma unroll
    for (int t = 0; t < DQK / 32; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) tw[(lane & 31) * TPITCH + 32 * t + acc_row(e, h)] = dk[t][e] * p.scale;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int r = 0; r < 32; ++r) {
      const int key = k0 + r;
      if (key >= p.Nk) break;
      float* dst = reinterpret_cast<float*>(p.dK) + b * p.sdkb + hd * p.sdkh + (int64_t)key * p.sdkn;
      for (int c = lane; c < DQK; c += 64) atomicAdd(dst + c, tw[r * TPITCH + c]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int t = 0; t < DV / 32; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) tw[(lane & 31) * TPITCH + 32 * t + acc_row(e, h)] = dvv[t][e];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int r = 0; r < 32; ++r) {
      const int key = k0 + r;
      if (key >= p.Nk) break;
      float* dst = reinterpret_cast<float*>(p.dV) + b * p.sdvb + hd * p.sdvh + (int64_t)key * p.sdvn;
      for (int c = lane; c < DV; c += 64) atomicAdd(dst + c, tw[r * TPITCH + c]);
    }
  }
}

// The pipelined dK/dV kernel pays off once a block streams enough query tiles to amortise its compiler-scheduled first and last
// iterations (measured: 6 273 and 1 569 queries -8 .. -15 %, 393 queries +-1 %).  DCC_ATTN_PIPE = 0 | 1 forces the choice (A/B runs).
// DQK = 96 (the RoPE configuration: no rel-pos columns) is built but not selected: its compiler-scheduled kernel needs 248 registers,
// i.e. already runs two waves per SIMD, and the hand-placed single wave loses to that (32-frame shapes: 15.0 vs 15.7 ms per step,
// tools/attn_bench_rope.py) — the pipelined form pays exactly where the compiler's form is alone on its SIMD (296 / 324 registers).
inline bool attn_pipe_enabled(const AttnParams& p, int dqk) {
  static const int forced = [] { const char* e = getenv("DCC_ATTN_PIPE"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  if (forced >= 0) return forced == 1;
  return dqk >= 128 && p.Nq / p.qsplit >= 768;
}

template <typename TO, int DQK, int DV>
int launch_dkv_pipe(const AttnParams& p, hipStream_t st) {
  using G = PipeGeo<DQK, DV>;
  static unsigned long long configured = 0;
  const unsigned long long bit = dcc_dev_bit();
  if (!(configured & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_pipe_kernel<TO, DQK, DV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS_BYTES) != hipSuccess)
      return DCC_ERR_LAUNCH;
    configured |= bit;
  }
  dim3 grid(((p.Nk + 127) / 128) * p.qsplit, p.H, p.B);
  hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<TO, DQK, DV>), grid, dim3(ANT), G::LDS_BYTES, st, p);
  return dcc_check_launch();
}

// key tiles per wave of the dK/dV kernel: DCC_ATTN_KW = 1 | 2 forces it (A/B runs), default by shape
inline int attn_kw(const AttnParams& p, int dqk) {
  static const int forced = [] { const char* e = getenv("DCC_ATTN_KW"); return e ? atoi(e) : 0; }();
  if (forced == 1 || forced == 2) return forced;
  (void)p; (void)dqk;
  return 1;
}

template <typename T, int DQK, int DV, bool DROP>
int launch_attn(int which, const AttnParams& p, int out_f32, hipStream_t st) {
  if (which == 0) {
    dim3 grid((p.Nq + 127) / 128, p.H, p.B);
    hipLaunchKernelGGL((attn_fwd_kernel<T, DQK, DV, DROP>), grid, dim3(ANT), 0, st, p);
  } else if (which == 1) {
    dim3 grid((p.Nq + 127) / 128, p.H, p.B);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DQK, DV, DROP>), grid, dim3(ANT), 0, st, p);
  } else {
    if constexpr (sizeof(T) == 2 && !DROP && DV == 96 && (DQK == 96 || DQK == 128 || DQK == 160)) {
      if (attn_pipe_enabled(p, DQK)) return out_f32 ? launch_dkv_pipe<float, DQK, DV>(p, st) : launch_dkv_pipe<bf16_t, DQK, DV>(p, st);
    }
#ifdef DCC_ATTN_KW2   // needs AGPR accumulators: build attention.hip without -amdgpu-mfma-vgpr-form
    if constexpr (sizeof(T) == 2 && !DROP) {
      if (attn_kw(p, DQK) == 2) {
        dim3 grid(((p.Nk + 255) / 256) * p.qsplit, p.H, p.B);
        if (out_f32) hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, float, DQK, DV, DROP, 2>), grid, dim3(ANT), 0, st, p);
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, T, DQK, DV, DROP, 2>), grid, dim3(ANT), 0, st, p);
        return dcc_check_launch();
      }
    }
#endif
    dim3 grid(((p.Nk + 127) / 128) * p.qsplit, p.H, p.B);
    if (out_f32) hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, float, DQK, DV, DROP, 1>), grid, dim3(ANT), 0, st, p);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, T, DQK, DV, DROP, 1>), grid, dim3(ANT), 0, st, p);
  }
  return dcc_check_launch();
}

template <typename T>
int dispatch_dims(int which, const AttnParams& p, int dqk, int dv, int out_f32, hipStream_t st) {
  if (p.drop_p > 0.0f) {
    // attention-probability dropout exists where nn.MultiheadAttention / HF BERT have it: 64-wide heads (BERT-base,
    // AttentionPool 512/8) and 128-wide heads (study aggregator 512/4); MViT's pooled attention has none
    if (dqk == 64 && dv == 64) return launch_attn<T, 64, 64, true>(which, p, out_f32, st);
    if (dqk == 128 && dv == 128) return launch_attn<T, 128, 128, true>(which, p, out_f32, st);
    return DCC_ERR_UNSUPPORTED;
  }
  if (dqk == 64 && dv == 64) return launch_attn<T, 64, 64, false>(which, p, out_f32, st);
  if (dqk == 96 && dv == 96) return launch_attn<T, 96, 96, false>(which, p, out_f32, st);
  if (dqk == 128 && dv == 96) return launch_attn<T, 128, 96, false>(which, p, out_f32, st);
  if (dqk == 160 && dv == 96) return launch_attn<T, 160, 96, false>(which, p, out_f32, st);
  if (dqk == 128 && dv == 128) return launch_attn<T, 128, 128, false>(which, p, out_f32, st);
  return DCC_ERR_UNSUPPORTED;
}

int check_common(const AttnParams& p, int dtype) {
  if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0) return DCC_ERR_ARG;
  if (p.H > 65535 || p.B > 65535) return DCC_ERR_ARG;
  const int ch = dtype == DCC_BF16 ? 8 : 4;
  const int64_t strides[] = {p.sqb, p.sqn, p.sqh, p.skb, p.skn, p.skh, p.svb, p.svn, p.svh, p.sob, p.son, p.soh};
  for (int64_t s : strides)
    if (s % ch) return DCC_ERR_UNSUPPORTED;
  if (((uintptr_t)p.Q & 15) || ((uintptr_t)p.K & 15) || ((uintptr_t)p.V & 15) || ((uintptr_t)p.O & 15))
    return DCC_ERR_UNSUPPORTED;
  return DCC_OK;
}

}  // namespace

extern "C" int dcc_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, const float* kbias,
                            const void* Qres, int res_start, const int64_t* strides /*12: q,k,v,o x (b,n,h)*/, int B,
                            int H, int Nq, int Nk, int DQK, int DV, float scale, float drop_p, uint64_t drop_seed,
                            const int64_t* drop_step, int dtype, void* stream) {
  if (!Q || !K || !V || !O || !strides || !(drop_p >= 0.0f && drop_p < 1.0f)) return DCC_ERR_ARG;
  AttnParams p{};
  p.Q = Q; p.K = K; p.V = V; p.O = O; p.LSE = LSE; p.kbias = kbias; p.Qres = Qres; p.res_start = res_start;
  p.sqb = strides[0]; p.sqn = strides[1]; p.sqh = strides[2];
  p.skb = strides[3]; p.skn = strides[4]; p.skh = strides[5];
  p.svb = strides[6]; p.svn = strides[7]; p.svh = strides[8];
  p.sob = strides[9]; p.son = strides[10]; p.soh = strides[11];
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale; p.qsplit = 1;
  p.drop_p = drop_p; p.drop_seed = drop_seed; p.drop_step = drop_step;
  int rc = check_common(p, dtype);
  if (rc) return rc;
  if (dtype == DCC_BF16) return dispatch_dims<bf16_t>(0, p, DQK, DV, 0, (hipStream_t)stream);
  if (dtype == DCC_F32) return dispatch_dims<float>(0, p, DQK, DV, 0, (hipStream_t)stream);
  return DCC_ERR_UNSUPPORTED;
}

// Backward. Dv is workspace [B,H,Nq] fp32 (written here).  dK/dV are T when qsplit == 1, else fp32
// buffers that the caller has zeroed (atomic accumulation over query splits).
extern "C" int dcc_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                            const float* LSE, float* Dv, const float* kbias, const void* Qres, int res_start, void* dQ,
                            void* dK, void* dV, const int64_t* strides /*21: q,k,v,o,dq,dk,dv x (b,n,h)*/, int B, int H,
                            int Nq, int Nk, int DQK, int DV, float scale, int qsplit, float drop_p, uint64_t drop_seed,
                            const int64_t* drop_step, int dtype, void* stream) {
  if (!Q || !K || !V || !O || !dO || !LSE || !Dv || !dQ || !dK || !dV || !strides) return DCC_ERR_ARG;
  if (!(drop_p >= 0.0f && drop_p < 1.0f)) return DCC_ERR_ARG;
  if (qsplit < 1) return DCC_ERR_ARG;
  AttnParams p{};
  p.Q = Q; p.K = K; p.V = V; p.O = const_cast<void*>(O); p.dO = dO; p.LSE = const_cast<float*>(LSE); p.Dv = Dv;
  p.kbias = kbias; p.Qres = Qres; p.res_start = res_start; p.dQ = dQ; p.dK = dK; p.dV = dV;
  p.sqb = strides[0]; p.sqn = strides[1]; p.sqh = strides[2];
  p.skb = strides[3]; p.skn = strides[4]; p.skh = strides[5];
  p.svb = strides[6]; p.svn = strides[7]; p.svh = strides[8];
  p.sob = strides[9]; p.son = strides[10]; p.soh = strides[11];
  p.sdqb = strides[12]; p.sdqn = strides[13]; p.sdqh = strides[14];
  p.sdkb = strides[15]; p.sdkn = strides[16]; p.sdkh = strides[17];
  p.sdvb = strides[18]; p.sdvn = strides[19]; p.sdvh = strides[20];
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale; p.qsplit = qsplit;
  p.drop_p = drop_p; p.drop_seed = drop_seed; p.drop_step = drop_step;
  int rc = check_common(p, dtype);
  if (rc) return rc;
  for (int i = 12; i < 21; ++i)
    if (strides[i] % (dtype == DCC_BF16 ? 8 : 4)) return DCC_ERR_UNSUPPORTED;   // 16-byte output pieces
  if (((uintptr_t)dQ & 15) || ((uintptr_t)dK & 15) || ((uintptr_t)dV & 15)) return DCC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype != DCC_BF16 && dtype != DCC_F32) return DCC_ERR_UNSUPPORTED;
  const int out_f32 = (qsplit > 1) || dtype == DCC_F32;
  for (int which = 1; which <= 2; ++which) {
    if (dtype == DCC_BF16) rc = dispatch_dims<bf16_t>(which, p, DQK, DV, out_f32, st);
    else rc = dispatch_dims<float>(which, p, DQK, DV, out_f32, st);
    if (rc) return rc;
  }
  return DCC_OK;
}
