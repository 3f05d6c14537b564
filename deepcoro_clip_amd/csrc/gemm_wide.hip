// Wide-tile bf16 GEMM for gfx950: 256 x 256 (or 256 x 128) output tile per workgroup, 8 waves that ALL load and compute.
//   C[m][n] = alpha * sum_k A(m,k) * B(n,k)  (+ bias[n]) (gelu / * gelu'(aux) / + R[m][n]),  same contract as gemm.hip.
//
// Why a third kernel: the ring kernels (gemm_ring.hip) stream 128 x 128 tiles, which costs 32 KB of L2 -> LDS traffic per
// 2.1 MFLOP; on deep reductions (the weight gradients: K = tokens of the batch) that feed, not the matrix cores, sets the
// pace.  A 256 x 256 tile halves the operand bytes per FLOP and the LDS fragment bytes per MFMA (a wave owns 128 x 64:
// 24 fragment reads per 32 MFMAs instead of 16 per 16), and with two MFMA waves per SIMD one wave's LDS / barrier stalls
// are covered by its partner instead of by dedicated loader waves.
//   * operands: global -> LDS by LDS-DMA (16 B per lane, zero-fill through the buffer descriptor) into a ring of four
//     32-deep stages that runs three steps ahead; every wave issues its share of a stage (3-4 KB) BETWEEN its MFMAs of
//     the current step; counted vmcnt, one raw barrier per step;
//   * LDS images are stacks of 128-wide sub-tiles in the ring kernel's two layouts (k-contiguous rows with the chunk
//     XOR-swizzled by the row, k-major rows read with ds_read_tr16_b64), so NN / NT / TN / TT are one kernel;
//   * fp32 output (weight gradients): split-K over workgroups + fp32 atomics, the Linear layer's bias gradient
//     (column sums of the k-major A operand) accumulated on the side from the A fragments the waves already hold;
//   * bf16 output: the MFMA runs with swapped operands (a lane owns 4 consecutive columns of one row), the tile goes
//     through LDS once and leaves in 16-byte stores of complete 128-byte row segments with the fused epilogue.
#include "gemm_params.h"
#include <stdlib.h>

namespace {

constexpr int WNT = 512;
constexpr int WBM = 256, WBK = 32;     // K elements per ring stage
constexpr int SUB_B = 128 * WBK * 2;   // one 128-wide sub-tile of a stage (8 KB)

__device__ __forceinline__ int wswz_km128(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }
template <int N> __device__ __forceinline__ void wwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 16 k of the 32 rows starting at t0 (< 128) of a k-contiguous sub-tile [128][32] bf16: 64-byte rows, the 16-byte chunk c
// of row r stored at c ^ ((r >> 2) & 3) (four rows share a 256-byte bank row; the 16-lane groups of ds_read_b128 then
// touch 16 different slots)
__device__ __forceinline__ bf16x8 wfrag_kc(const char* sub, int t0, int s, int lane) {
  const int row = t0 + (lane & 31), chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(sub + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}
// k-major sub-tile [32 k][128] bf16 (256-byte rows), transposing reads
__device__ __forceinline__ bf16x8 wfrag_km(const char* sub, int t0, int s, int lane) {
  const int h = lane >> 5, i = lane & 15, q = i >> 2, pp = i & 3, g = (lane >> 4) & 1;
  bf16x8 f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int krow = 16 * s + 8 * h + 4 * half + q;
    const int chunk = ((t0 + 16 * g) >> 3) + (pp >> 1);
    const char* p = sub + krow * 256 + ((chunk ^ wswz_km128(krow)) << 4) + 8 * (pp & 1);
    const bf16x4 tb = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p)));
    f[4 * half + 0] = tb[0]; f[4 * half + 1] = tb[1]; f[4 * half + 2] = tb[2]; f[4 * half + 3] = tb[3];
  }
  return f;
}

struct WGeo { int tiles, tilesN, ksteps, per, items; };

// NS = ring depth (stages; NS - 1 steps in flight).  BN = 256: NS = 4, 128 KB of LDS, one workgroup per CU, two waves per SIMD.
// BN = 128, NS = 3: 72 KB and <= 128 VGPRs, so TWO workgroups share a CU and one's epilogue (LDS transposition + stores,
// as long as its 12-step main loop when K = 384) runs under the other's MFMAs.
template <typename TC, bool A_KM, bool B_KM, int BN, int NS>
__global__ __launch_bounds__(WNT, (BN == 128 && NS == 3) ? 4 : 2) void gemm_wide_kernel(GemmParams p, WGeo geo) {
  constexpr int WNS = NS, WD = NS - 1;   // stages / steps in flight
  constexpr bool OUT16 = sizeof(TC) == 2;
  constexpr int NSA = WBM / 128, NSB = BN / 128;           // sub-tiles per stage
  constexpr int STAGE = (NSA + NSB) * SUB_B;
  constexpr int WM = BN == 256 ? 2 : 4, WN = 8 / WM;      // wave grid
  constexpr int RW = WBM / WM, CW = BN / WN;              // rows / columns of a wave's sub-tile (CW == 64)
  constexpr int TI = RW / 32, TJ = CW / 32;
  constexpr int PPW = (NSA + NSB) * 8 / 8;                // 1-KB DMA pieces per wave per stage
  static_assert(CW == 64, "a wave owns 64 output columns");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const int wr = wave / WN, wc = wave % WN;

  // XCD-aware item order: the workgroups of one XCD walk neighbouring tiles of one A row panel
  const int G = gridDim.x, bid = blockIdx.x;
  const int item = (bid & 7) * (G >> 3) + (bid >> 3);
  if (item >= geo.items) return;
  const int z = item / geo.tiles, t = item - z * geo.tiles;
  const int tm = t / geo.tilesN, tn = t - tm * geo.tilesN;
  const int m0 = tm * WBM, n0 = tn * BN;
  const int bz = z / p.splitk, sk = z - bz * p.splitk;
  const int ks0 = sk * geo.per;
  int ks1 = ks0 + geo.per; if (ks1 > geo.ksteps) ks1 = geo.ksteps;
  const int nk = ks1 - ks0;
  if (nk <= 0) return;

  // ------------------------------------------------------------------------------------------------ loads
  auto bytes_left = [](bool km, int64_t ld, int rem, int K) -> unsigned {
    const int64_t elems = rem <= 0 ? 0 : (km ? (int64_t)(K - 1) * ld + rem : (int64_t)(rem - 1) * ld + K);
    return (unsigned)(elems * 2);
  };
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(p.A) + (int64_t)bz * p.sA + (A_KM ? (int64_t)m0 : (int64_t)m0 * p.lda);
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(p.B) + (int64_t)bz * p.sB + (B_KM ? (int64_t)n0 : (int64_t)n0 * p.ldb);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, bytes_left(A_KM, p.lda, p.M - m0, p.K), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb), 0, bytes_left(B_KM, p.ldb, p.N - n0, p.K), 0x00020000);
  constexpr int OOB = 0x7fffff00;
  int voff[PPW], kc[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int g = wave * PPW + j;                  // piece of the stage: A sub-tiles first, then B (8 pieces of 1 KB each)
    const bool isA = g < NSA * 8;
    const int gg = isA ? g : g - NSA * 8, sub = gg >> 3, pin = gg & 7;
    const bool km = isA ? A_KM : B_KM;
    const int ld = (int)(isA ? p.lda : p.ldb);
    kc[j] = 0;
    if (km) { const int R = pin * 4 + (lane >> 4), c = (lane & 15) ^ wswz_km128(R); voff[j] = (R * ld + sub * 128 + c * 8) * 2; }
    else { const int R = pin * 16 + (lane >> 2), c = (lane & 3) ^ ((R >> 2) & 3); voff[j] = ((sub * 128 + R) * ld + c * 8) * 2; kc[j] = c * 8; }
  }
  const int kstrideA = A_KM ? WBK * (int)p.lda * 2 : WBK * 2, kstrideB = B_KM ? WBK * (int)p.ldb * 2 : WBK * 2;
  // `live` false: the step lies past this item's K range; the piece is still issued (out of range = zeros into a stage
  // nobody reads) so that every step has the same number of loads in flight and the waits are compile-time constants
  auto issue_piece = [&](int j, int stage, int kb, bool live) {
    const int g = wave * PPW + j;
    const bool isA = g < NSA * 8;
    const bool km = isA ? A_KM : B_KM;
    const int vo = (!live || (!km && kb * WBK + kc[j] >= p.K)) ? OOB : voff[j] + kb * (isA ? kstrideA : kstrideB);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsA : rsB, LDS_PTR(void, smem + stage * STAGE + g * 1024), 16, vo, 0, 0, 0);
  };

  // ---------------------------------------------------------------------------------------------- main loop
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  const bool do_colsum = A_KM && (p.flags & GEMM_COLSUM_A) && n0 == 0 && wc == 0;
  float csum[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) csum[i] = 0.0f;

#pragma unroll
  for (int d = 0; d < WD; ++d) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) issue_piece(j, d, ks0 + d, d < nk);
  }
  for (int q = 0; q < nk; ++q) {
    wwait_vm<(WD - 1) * PPW>();    // this wave's pieces of step q have landed: only the two younger steps' may be out
    __builtin_amdgcn_s_barrier();  // step q landed for everyone; everyone is done reading step q - 1's stage
    const char* sa = smem + (q % WNS) * STAGE;
    const char* sb = sa + NSA * SUB_B;
    const int nq = q + WD;          // the step whose loads go out underneath this step's MFMAs, into step q - 1's stage
    const bool more = nq < nk;
    bf16x8 fa[2][TI], fb[2][TJ];
    auto fetch = [&](int buf, int s) {
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int r0 = wr * RW + 32 * i;
        fa[buf][i] = A_KM ? wfrag_km(sa + (r0 >> 7) * SUB_B, r0 & 127, s, lane) : wfrag_kc(sa + (r0 >> 7) * SUB_B, r0 & 127, s, lane);
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int c0 = wc * CW + 32 * j;
        fb[buf][j] = B_KM ? wfrag_km(sb + (c0 >> 7) * SUB_B, c0 & 127, s, lane) : wfrag_kc(sb + (c0 >> 7) * SUB_B, c0 & 127, s, lane);
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < WBK / 16; ++s) {
      const int cur = s & 1;
      if (s + 1 < WBK / 16) fetch(cur ^ 1, s + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          if constexpr (OUT16) acc[i][j] = mma16(fb[cur][j], fa[cur][i], acc[i][j]);   // C^T: lanes = rows m
          else acc[i][j] = mma16(fa[cur][i], fb[cur][j], acc[i][j]);
        }
        // one DMA piece behind every TJ MFMAs (the matrix pipe keeps draining while the wave issues it)
        constexpr int STR = (2 * TI) / PPW >= 1 ? (2 * TI) / PPW : 1;   // spread the pieces over the step's MFMA groups
        const int pos = s * TI + i;
        if (pos % STR == 0 && pos / STR < PPW) issue_piece(pos / STR, nq % WNS, ks0 + nq, more);
      }
      if (A_KM && do_colsum) {  // bias gradient: sum over k of this lane's 8 A elements (row = lane & 31 of tile i)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[i] += (float)fa[cur][i][e];
      }
    }
  }

  wwait_vm<0>();  // the trailing dummy pieces too: the epilogue reuses the stages

  // ------------------------------------------------------------------------------------------------ epilogue
  if (A_KM && do_colsum) {
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const float v = csum[i] + __shfl_xor(csum[i], 32);   // the two k halves of a row
      const int row = m0 + wr * RW + 32 * i + (lane & 31);
      if (h == 0 && row < p.M) atomicAdd(p.colsum + row, v);
    }
  }
  if constexpr (!OUT16) {
    float* C = reinterpret_cast<float*>(p.C) + (int64_t)bz * p.sC;
    const bool atomic = (p.splitk > 1) || (p.flags & GEMM_ACCUM);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = n0 + wc * CW + 32 * j + (lane & 31);
        if (col >= p.N) continue;
        const float bv = ((p.flags & GEMM_BIAS) && sk == 0) ? p.bias[col] : 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wr * RW + 32 * i + acc_row(e, h);
          if (row >= p.M) continue;
          const float v = acc[i][j][e] * p.alpha + bv;
          if (atomic) atomicAdd(C + (int64_t)row * p.ldc + col, v);
          else C[(int64_t)row * p.ldc + col] = v;
        }
      }
  } else {
    // bf16: through this wave's private LDS strip [RW rows][64 cols] (pitch 144 B), then whole 128-byte row segments
    constexpr int PITCH = 2 * CW + 16;
    __builtin_amdgcn_s_barrier();  // the stages are free: no wave reads operands any more
    char* strip = smem + wave * (RW * PITCH);
    // alpha * acc + bias rounds to bf16 ONCE here (what an unfused bf16 Linear hands to GELU / the residual add: the
    // ring kernel's contract); this lane's columns are n = 32 j + 8 g4 + 4 h + e
    f32x4 bias4[TJ][4];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + wc * CW + 32 * j + 8 * g4 + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) bias4[j][g4][e] = ((p.flags & GEMM_BIAS) && n + e < p.N) ? p.bias[n + e] : 0.0f;
      }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        char* rowp = strip + (32 * i + (lane & 31)) * PITCH + (32 * j + 4 * h) * 2;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[i][j][4 * g4 + e] * p.alpha + bias4[j][g4][e]);
          *reinterpret_cast<bf16x4*>(rowp + 16 * g4) = o;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // a wave's own LDS writes before its reads of the strip
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + (int64_t)bz * p.sC;
    const int c8 = (lane & 7) * 8, col = n0 + wc * CW + c8;
    const bf16_t* X = reinterpret_cast<const bf16_t*>((p.flags & GEMM_RESID) ? p.R : p.aux);
    const int64_t ldxx = (p.flags & GEMM_RESID) ? p.ldr : p.ldaux;
#pragma unroll 4
    for (int it = 0; it < RW / 8; ++it) {
      const int r = 8 * it + (lane >> 3);
      const int row = m0 + wr * RW + r;
      const bf16x8 mv = *reinterpret_cast<const bf16x8*>(strip + r * PITCH + c8 * 2);
      if (row >= p.M || col >= p.N) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)mv[e];
      if (p.flags & GEMM_SAVE_PRE) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
        STREAM_STORE(reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.aux) + (int64_t)row * p.ldaux + col), o);
      }
      if (p.flags & GEMM_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_for<bf16_t>(v[e]);
      }
      if (p.flags & GEMM_DGELU) {
        const bf16x8 a = STREAM_LOAD(reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.aux) + (int64_t)row * p.ldaux + col));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dgelu_for<bf16_t>((float)a[e]);
      }
      if (p.flags & GEMM_RESID) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(X + (int64_t)row * ldxx + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      *reinterpret_cast<bf16x8*>(C + (int64_t)row * p.ldc + col) = o;
    }
  }
}

template <typename TC, bool A_KM, bool B_KM, int BN, int NS>
int wide_launch(const GemmParams& p, const WGeo& geo, hipStream_t st) {
  constexpr int WNS = NS;
  constexpr int NSA = WBM / 128, NSB = BN / 128;
  constexpr int STAGE = (NSA + NSB) * SUB_B;
  constexpr int WM = BN == 256 ? 2 : 4;
  constexpr int STRIP = 8 * (WBM / WM) * (2 * 64 + 16);
  constexpr int LDS = (WNS * STAGE > STRIP) ? WNS * STAGE : STRIP;
  auto kern = gemm_wide_kernel<TC, A_KM, B_KM, BN, NS>;
  static unsigned long long configured = 0;
  const unsigned long long devbit = dcc_dev_bit();
  if (!(configured & devbit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return DCC_ERR_LAUNCH;
    configured |= devbit;
  }
  const int G = (geo.items + 7) & ~7;
  hipLaunchKernelGGL(kern, dim3(G), dim3(WNT), LDS, st, p, geo);
  return dcc_check_launch();
}

}  // namespace

// Returns DCC_ERR_UNSUPPORTED for shapes / layouts outside the kernel's domain (the caller falls back to the ring kernel).
int dcc_gemm_wide_launch(const GemmParams& p, int a_km, int b_km, int out_dtype, int bn, int ns, hipStream_t st) {
  if ((p.lda % 8) || (p.ldb % 8) || (p.sA % 8) || (p.sB % 8) || (p.ldc % 8) || (p.N % 8) || (p.sC % 8)) return DCC_ERR_UNSUPPORTED;
  if ((a_km && (p.M % 8)) || (b_km && (p.N % 8))) return DCC_ERR_UNSUPPORTED;
  if ((!a_km || !b_km) && (p.K % 8)) return DCC_ERR_UNSUPPORTED;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) || ((uintptr_t)p.C & 15)) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & GEMM_RESID) && ((p.ldr % 8) || ((uintptr_t)p.R & 15))) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & (GEMM_SAVE_PRE | GEMM_DGELU)) && ((p.ldaux % 8) || ((uintptr_t)p.aux & 15))) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & GEMM_BIAS) && ((uintptr_t)p.bias & 3)) return DCC_ERR_UNSUPPORTED;
  const int64_t extA = (a_km ? (int64_t)p.K * p.lda : (int64_t)p.M * p.lda) * 2;
  const int64_t extB = (b_km ? (int64_t)p.K * p.ldb : (int64_t)p.N * p.ldb) * 2;
  if (extA >= (1LL << 31) || extB >= (1LL << 31)) return DCC_ERR_UNSUPPORTED;
  if (out_dtype == DCC_BF16 && (p.splitk > 1 || (p.flags & (GEMM_ACCUM | GEMM_COLSUM_A)))) return DCC_ERR_UNSUPPORTED;
  if (out_dtype == DCC_F32 && (p.flags & (GEMM_GELU | GEMM_DGELU | GEMM_SAVE_PRE | GEMM_RESID))) return DCC_ERR_UNSUPPORTED;
  if ((bn != 128 && bn != 256) || (ns != 3 && ns != 4) || (ns == 3 && bn != 128)) return DCC_ERR_ARG;
  WGeo geo;
  const int tilesM = (p.M + WBM - 1) / WBM;
  geo.tilesN = (p.N + bn - 1) / bn;
  geo.tiles = tilesM * geo.tilesN;
  const int64_t items = (int64_t)geo.tiles * p.batch * p.splitk;
  if (items > 0x3fffffff) return DCC_ERR_UNSUPPORTED;
  geo.items = (int)items;
  geo.ksteps = (p.K + WBK - 1) / WBK;
  geo.per = (geo.ksteps + p.splitk - 1) / p.splitk;
#define WIDE(TC_, BN_, NS_)                                                            \
  do {                                                                                 \
    if (!a_km && !b_km) return wide_launch<TC_, false, false, BN_, NS_>(p, geo, st);   \
    if (!a_km && b_km) return wide_launch<TC_, false, true, BN_, NS_>(p, geo, st);     \
    if (a_km && !b_km) return wide_launch<TC_, true, false, BN_, NS_>(p, geo, st);     \
    return wide_launch<TC_, true, true, BN_, NS_>(p, geo, st);                         \
  } while (0)
  if (out_dtype == DCC_BF16) { if (bn == 256) WIDE(bf16_t, 256, 4); if (ns == 3) WIDE(bf16_t, 128, 3); WIDE(bf16_t, 128, 4); }
  if (out_dtype == DCC_F32) { if (bn == 256) WIDE(float, 256, 4); if (ns == 3) WIDE(float, 128, 3); WIDE(float, 128, 4); }
#undef WIDE
  return DCC_ERR_UNSUPPORTED;
}
