// Persistent LDS-DMA ring GEMM for gfx950 (bf16 operands, fp32 accumulate) — the large GEMMs of both towers.
//   C[m][n] = alpha * sum_k A(m,k) * B(n,k)  (+ bias[n]) (gelu / * gelu'(aux) / + R[m][n]),  same contract as gemm.hip.
//
// Why a second kernel: the towers' GEMMs have SHORT reductions (K = 384 ... 1536) and fat outputs, so a tile spends
// as long in its prologue (first loads) and epilogue as in its MFMA loop; the register-staged kernel of gemm.hip also
// drains its prefetch every K-step (hipcc waits vmcnt(0) across the loop back-edge).  Here
//   * one workgroup per CU walks a list of (tile, split) work items, XCD-aware (the CUs of one XCD work on
//     neighbouring tiles of the same A panel, so the panel is read from HBM once and then from that XCD's L2);
//   * operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, no VGPRs, zero-fill past the end of the
//     operand via the buffer descriptor) into a ring of NS stages that runs D = NS-1 K-steps ahead of the MFMAs and
//     ACROSS work items, so the next tile's operands land while this tile's epilogue runs;
//   * waits are counted (`s_waitcnt vmcnt(N)`, never 0 in steady state) and barriers are raw `s_barrier`s: one per K-step;
//   * LDS images are lane-linear as LDS-DMA requires; bank-conflict-free fragment reads come from XOR-swizzling the
//     16-byte chunk a lane FETCHES (k-contiguous tiles: chunk ^ ((row >> 1) & 7); k-major tiles: gemm.hip's swz_km);
//   * the bias row rides in the ring (512 B per stage) so that the epilogue has no ordinary load to wait for.
// gfx9 counts loads AND stores on vmcnt, in issue order: the stores an epilogue issues sit between ring loads, so
// the first D K-steps after an epilogue may leave them outstanding (+E on the immediate; E is a lower bound of the
// stores issued, rounded down to the immediates compiled in).
#include "gemm_params.h"
#include <stdlib.h>

namespace {

constexpr int NL = 8;                          // loader (LDS-DMA) waves: an LDS-DMA costs its wave ~100-180 issue cycles, so the
                                               // 32 KB of a K-step are spread over many waves that do nothing else
constexpr int NT = 64 * (4 + NL);              // 4 consumer (MFMA + epilogue) waves, one per SIMD, + the loaders
constexpr int BMN = 128, BK = 64;
constexpr int TILE_B = BMN * BK * 2;           // one operand tile, bf16
constexpr int BIAS_B = BMN * 4;                // fp32 bias slice of the tile's columns
constexpr int STAGE_B = 2 * TILE_B + BIAS_B;
constexpr int CWPITCH = 40;                    // fp32 pitch of a consumer wave's private 32x32 epilogue tile (rows 4 apart: 32 banks apart)
constexpr int CWAVE_B = 32 * CWPITCH * 4;
constexpr int CSTAGE_B = 4 * CWAVE_B;
constexpr int PPL = 16 / NL;                   // 1-KiB pieces per operand per loader wave per K-step

__device__ __forceinline__ int swz_km128(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// 16 k of the 32 rows/cols starting at t0: k-contiguous image [128][64] bf16, 128-byte rows, chunk ^ ((row>>1)&7)
__device__ __forceinline__ bf16x8 frag_kc(const char* tile, int t0, int s, int lane) {
  const int row = t0 + (lane & 31), chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
// k-major image [64 k][128] bf16, 256-byte rows, transposing reads (same geometry as gemm.hip's load_frag)
__device__ __forceinline__ bf16x8 frag_km(const char* tile, int t0, int s, int lane) {
  const int h = lane >> 5, i = lane & 15, q = i >> 2, pp = i & 3, g = (lane >> 4) & 1;
  bf16x8 f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int krow = 16 * s + 8 * h + 4 * half + q;
    const int chunk = ((t0 + 16 * g) >> 3) + (pp >> 1);
    const char* p = tile + krow * 256 + ((chunk ^ swz_km128(krow)) << 4) + 8 * (pp & 1);
    const bf16x4 tb = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p)));
    f[4 * half + 0] = tb[0]; f[4 * half + 1] = tb[1]; f[4 * half + 2] = tb[2]; f[4 * half + 3] = tb[3];
  }
  return f;
}

struct Geo {  // launch-constant geometry
  int items, tiles, tilesN, ksteps, per;
};

#ifdef DCC_RING_PROFILE
__device__ unsigned long long g_prof[8];  // [0] consumer barrier wait, [1] MFMA section, [2] epilogue, [3] steps, [4] loader vm wait, [5] loader barrier, [6] loader issue
#define PROF_T() __builtin_amdgcn_s_memtime()
#define PROF_DECL unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_ADD(i, v) prof_acc[i] += (unsigned long long)(v)
#define PROF_FLUSH() do { if (lane == 0 && (wave == 0 || wave == 4)) { for (int i_ = 0; i_ < 8; ++i_) if (prof_acc[i_]) atomicAdd(&g_prof[i_], prof_acc[i_]); } } while (0)
#else
#define PROF_T() 0ull
#define PROF_DECL
#define PROF_ADD(i, v) do { } while (0)
#define PROF_FLUSH() do { } while (0)
#endif

template <typename TC, bool A_KM, bool B_KM, int NS>
__global__ __launch_bounds__(NT, 1) void gemm_ring_kernel(GemmParams p, Geo geo) {
  constexpr int D = NS - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  PROF_DECL;
  const bool loader = wave >= 4;           // waves 4.. only move data; waves 0..3 only do MFMA + epilogue
  const int w4 = wave & 3;                 // consumer index
  const int lw = wave - 4;                 // loader index
  const int wr = w4 >> 1, wc = w4 & 1, h = lane >> 5;

  // work items of this workgroup: item(i) = i * G + xcd * (G / 8) + (bid / 8)  (G is a multiple of 8)
  const int G = gridDim.x, bid = blockIdx.x;
  const int item0 = (bid & 7) * (G >> 3) + (bid >> 3);

  struct Item {
    int idx;       // global item index, >= geo.items when exhausted
    int m0, n0, bz, sk, ks0, nk;
  };
  auto locate = [&](int idx) {
    Item it;
    for (;; idx += G) {  // skip splits without K-steps
      it.idx = idx;
      if (idx >= geo.items) { it.m0 = it.n0 = it.bz = it.sk = it.ks0 = 0; it.nk = 0; return it; }
      const int z = idx / geo.tiles, t = idx - z * geo.tiles;
      const int tm = t / geo.tilesN, tn = t - tm * geo.tilesN;
      it.m0 = tm * BMN; it.n0 = tn * BMN;
      it.bz = z / p.splitk; it.sk = z - it.bz * p.splitk;
      it.ks0 = it.sk * geo.per;
      int ks1 = it.ks0 + geo.per; if (ks1 > geo.ksteps) ks1 = geo.ksteps;
      it.nk = ks1 - it.ks0;
      if (it.nk > 0) return it;
    }
  };

  if (loader) {
    // =============================================================================================== loader waves
    // per-lane source offsets (bytes from the tile origin at k = 0) of this wave's four DMA pieces per operand
    int voffA[PPL], voffB[PPL];
    int kcA[PPL], kcB[PPL];  // k offset (elements) of this lane's chunk inside a K-step, k-contiguous operands only
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const int g = lw * PPL + j;
      kcA[j] = kcB[j] = 0;
      if (A_KM) { const int R = g * 4 + (lane >> 4), c = (lane & 15) ^ swz_km128(R); voffA[j] = (R * (int)p.lda + c * 8) * 2; }
      else { const int R = g * 8 + (lane >> 3), c = (lane & 7) ^ ((R >> 1) & 7); voffA[j] = (R * (int)p.lda + c * 8) * 2; kcA[j] = c * 8; }
      if (B_KM) { const int R = g * 4 + (lane >> 4), c = (lane & 15) ^ swz_km128(R); voffB[j] = (R * (int)p.ldb + c * 8) * 2; }
      else { const int R = g * 8 + (lane >> 3), c = (lane & 7) ^ ((R >> 1) & 7); voffB[j] = (R * (int)p.ldb + c * 8) * 2; kcB[j] = c * 8; }
    }
    constexpr int OOB = 0x7fffff00;  // beyond any descriptor range (operands are < 2 GiB): reads as zero
    const bool bias_wave = lw < 2;  // two 256-byte pieces carry the tile's bias slice
    const int voffBias = ((lw & 1) * 64 + lane) * 4;
    const int kstrideA = A_KM ? BK * (int)p.lda * 2 : BK * 2;  // bytes per K-step
    const int kstrideB = B_KM ? BK * (int)p.ldb * 2 : BK * 2;
    auto bytes_left = [](bool km, int64_t ld, int rem, int K) -> unsigned {
      const int64_t elems = rem <= 0 ? 0 : (km ? (int64_t)(K - 1) * ld + rem : (int64_t)(rem - 1) * ld + K);
      return (unsigned)(elems * 2);  // the launcher keeps operands below 2 GiB
    };
    Item L = locate(item0);
    int l_kt = 0, loads_issued = 0;
    __amdgpu_buffer_rsrc_t rsA, rsB, rsBias;
    auto bind_loads = [&]() {
      if (L.idx >= geo.items) return;
      const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (int64_t)L.bz * p.sA + (A_KM ? (int64_t)L.m0 : (int64_t)L.m0 * p.lda);
      const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + (int64_t)L.bz * p.sB + (B_KM ? (int64_t)L.n0 : (int64_t)L.n0 * p.ldb);
      rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, bytes_left(A_KM, p.lda, p.M - L.m0, p.K), 0x00020000);
      rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, bytes_left(B_KM, p.ldb, p.N - L.n0, p.K), 0x00020000);
      const bool hb = (p.flags & GEMM_BIAS) != 0;
      rsBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hb ? p.bias + L.n0 : (const float*)p.A), 0,
                                                 hb ? (unsigned)((p.N - L.n0) * 4) : 0u, 0x00020000);
    };
    bind_loads();
    auto issue = [&]() {
      if (L.idx >= geo.items) return;
      char* st = smem + (loads_issued % NS) * STAGE_B;
      const int kb = L.ks0 + l_kt;
#ifdef DCC_RING_PROFILE
      if (p.flags & 0x1000) {  // experiment: no DMA at all (results are garbage) — isolates LDS write contention
        ++loads_issued;
        if (++l_kt == L.nk) { L = locate(L.idx + G); l_kt = 0; bind_loads(); }
        return;
      }
#endif
      // the K offset goes into the VGPR offset: the descriptor's range check does not see the scalar offset
      const int soA = kb * kstrideA, soB = kb * kstrideB;
      // a k-contiguous operand's K tail (K % 64 != 0) would run into the next row: those chunks are sent out of range
      const int k0 = kb * BK;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int vo = (!A_KM && k0 + kcA[j] >= p.K) ? OOB : voffA[j] + soA;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, st + (lw * PPL + j) * 1024), 16, vo, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int vo = (!B_KM && k0 + kcB[j] >= p.K) ? OOB : voffB[j] + soB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, st + TILE_B + (lw * PPL + j) * 1024), 16, vo, 0, 0, 0);
      }
      if (bias_wave)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LDS_PTR(void, st + 2 * TILE_B + (lw & 1) * 256), 4, voffBias, 0, 0, 0);
      ++loads_issued;
      if (++l_kt == L.nk) { L = locate(L.idx + G); l_kt = 0; bind_loads(); }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue();
    for (int q = 0; q < loads_issued; ++q) {
      // step q's pieces were issued D steps ago; the D-1 younger steps may stay in flight (only DMAs are on this
      // wave's vmcnt, so the count is exact)
      const unsigned long long t0 = PROF_T();
      if (loads_issued - (q + 1) >= D - 1) {
        if (bias_wave) wait_vm<(D - 1) * (2 * PPL + 1)>();
        else wait_vm<(D - 1) * 2 * PPL>();
      } else {
        wait_vm<0>();
      }
      const unsigned long long t1 = PROF_T();
      __builtin_amdgcn_s_barrier();  // consumers: step q landed; loaders: step q-1's stage has been read by every consumer
      const unsigned long long t2 = PROF_T();
      issue();
      PROF_ADD(4, t1 - t0); PROF_ADD(5, t2 - t1); PROF_ADD(6, PROF_T() - t2);
    }
    PROF_FLUSH();
    return;
  }

  // ================================================================================================ consumer waves
  float* ct = reinterpret_cast<float*>(smem + NS * STAGE_B + w4 * CWAVE_B);  // this wave's private 32x32 staging tile
  Item Cc = locate(item0);
  int c_kt = 0, q = 0;
  f32x16 acc[2][2], cs[2];
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) cs[i][e] = 0.0f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }
  };
  zero_acc();
  const bool atomic = (p.splitk > 1) || (p.flags & GEMM_ACCUM);

  while (Cc.idx < geo.items) {
    const unsigned long long tb0 = PROF_T();
    __builtin_amdgcn_s_barrier();
    const unsigned long long tb1 = PROF_T();
    PROF_ADD(0, tb1 - tb0); PROF_ADD(3, 1);
    const char* sa = smem + (q % NS) * STAGE_B;
    const char* sb = sa + TILE_B;
    const bool do_colsum = A_KM && (p.flags & GEMM_COLSUM_A) && Cc.n0 == 0 && wc == 0;
    // fragments of K-substep s+1 are fetched before the MFMAs of substep s issue (one consumer wave per SIMD: nothing
    // else would cover the LDS latency)
    bf16x8 fa[2][2], fb[2][2];
    auto fetch = [&](int buf, int s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[buf][i] = A_KM ? frag_km(sa, wr * 64 + 32 * i, s, lane) : frag_kc(sa, wr * 64 + 32 * i, s, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[buf][j] = B_KM ? frag_km(sb, wc * 64 + 32 * j, s, lane) : frag_kc(sb, wc * 64 + 32 * j, s, lane);
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int cur = s & 1;
      if (s + 1 < BK / 16) fetch(cur ^ 1, s + 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this substep's MFMAs (hipcc would sink it)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma16(fa[cur][i], fb[cur][j], acc[i][j]);
      if (A_KM && do_colsum) {  // row sums of A on the matrix core: A x ones, every output column holds the sum
#pragma unroll
        for (int i = 0; i < 2; ++i) cs[i] = mma16(fa[cur][i], ones, cs[i]);
      }
    }

    const unsigned long long tb2 = PROF_T();
    PROF_ADD(1, tb2 - tb1);
    if (c_kt + 1 == Cc.nk) {
      // ------------------------------------------------------------------------------ epilogue of this item
      // (no workgroup barrier in here: the loaders keep the ring D steps ahead, into the next item, meanwhile)
      const int m0 = Cc.m0, n0 = Cc.n0;
      TC* C = reinterpret_cast<TC*>(p.C) + (int64_t)Cc.bz * p.sC;
      const float* biasl = reinterpret_cast<const float*>(sa + 2 * TILE_B);
      const bool full = (m0 + BMN <= p.M) && (n0 + BMN <= p.N);
      if (A_KM && do_colsum) {
        if ((lane & 31) == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = m0 + wr * 64 + 32 * i + acc_row(e, h);
              if (row < p.M) atomicAdd(p.colsum + row, cs[i][e]);
            }
        }
      }
      if (atomic) {
        if constexpr (sizeof(TC) == 4) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int lc = wc * 64 + 32 * j + (lane & 31), col = n0 + lc;
              if (col >= p.N) continue;
              const float bv = ((p.flags & GEMM_BIAS) && Cc.sk == 0) ? biasl[lc] : 0.0f;
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int row = m0 + wr * 64 + 32 * i + acc_row(e, h);
                if (row < p.M) atomicAdd(reinterpret_cast<float*>(C) + (int64_t)row * p.ldc + col, acc[i][j][e] * p.alpha + bv);
              }
            }
        }
      } else {
        const TC* R = reinterpret_cast<const TC*>(p.R);
        const int cgl = lane & 3;  // this lane's 8-column group inside a 32-column MFMA tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int lc = wc * 64 + 32 * j + cgl * 8;
          float bias8[8];
          {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(biasl + lc);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(biasl + lc + 4);
            bias8[0] = lo[0]; bias8[1] = lo[1]; bias8[2] = lo[2]; bias8[3] = lo[3];
            bias8[4] = hi[0]; bias8[5] = hi[1]; bias8[6] = hi[2]; bias8[7] = hi[3];
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            // one 32x32 accumulator tile through this wave's private LDS tile (a wave's LDS ops execute in order)
#pragma unroll
            for (int e = 0; e < 16; ++e) ct[acc_row(e, h) * CWPITCH + (lane & 31)] = acc[i][j][e];
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
              const int r = (lane >> 2) + 16 * g2;
              const int row = m0 + wr * 64 + 32 * i + r, col = n0 + lc;
              float v[8];
              {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(ct + r * CWPITCH + cgl * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(ct + r * CWPITCH + cgl * 8 + 4);
                v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
              }
              if (!full && (row >= p.M || col >= p.N)) continue;
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias8[e];
              if (p.flags & GEMM_SAVE_PRE) {
                TC* ap = reinterpret_cast<TC*>(p.aux) + (int64_t)row * p.ldaux + col;
                if constexpr (sizeof(TC) == 2) {
                  bf16x8 o;
#pragma unroll
                  for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                  STREAM_STORE(reinterpret_cast<bf16x8*>(ap), o);
                } else {
                  *reinterpret_cast<f32x4*>(ap) = f32x4{v[0], v[1], v[2], v[3]};
                  *reinterpret_cast<f32x4*>(ap + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
              }
              if (p.flags & GEMM_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_for<TC>(v[e]);
              }
              if (p.flags & GEMM_DGELU) {
                const bf16x8 a = STREAM_LOAD(reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.aux) + (int64_t)row * p.ldaux + col));
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dgelu_for<bf16_t>((float)a[e]);
              }
              if (p.flags & GEMM_RESID) {
                const TC* rp = R + (int64_t)row * p.ldr + col;
                if constexpr (sizeof(TC) == 2) {
                  const bf16x8 a = *reinterpret_cast<const bf16x8*>(rp);
#pragma unroll
                  for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
                } else {
                  const f32x4 lo = *reinterpret_cast<const f32x4*>(rp), hi = *reinterpret_cast<const f32x4*>(rp + 4);
                  v[0] += lo[0]; v[1] += lo[1]; v[2] += lo[2]; v[3] += lo[3]; v[4] += hi[0]; v[5] += hi[1]; v[6] += hi[2]; v[7] += hi[3];
                }
              }
              TC* dst = C + (int64_t)row * p.ldc + col;
              if constexpr (sizeof(TC) == 2) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                *reinterpret_cast<bf16x8*>(dst) = o;
              } else {
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
              }
            }
          }
        }
      }
      zero_acc();
      Cc = locate(Cc.idx + G);
      c_kt = 0;
      PROF_ADD(2, PROF_T() - tb2);
    } else {
      ++c_kt;
    }
    ++q;
  }
  PROF_FLUSH();
}

// =====================================================================================================
// Ring kernel with an epilogue crew (bf16 output, no split-K): the towers' K = 96..1536 GEMMs spend as long in the
// epilogue (bias / GELU / residual / 32 KB of stores per tile) as in their 2..24 K-steps, and 4 MFMA waves doing that
// serially left the matrix cores idle half of the time.  Here a workgroup has three crews of 4 waves:
//   consumers  MFMA only.  One barrier after a tile's last K-step they add the bias, round to bf16 and drop the tile
//              into an LDS mailbox (C^T accumulators: the MFMA runs with swapped operands so that a lane holds 4
//              consecutive columns of one row -> 8-byte LDS writes), then carry on with the next tile;
//   loaders    LDS-DMA ring as above (NS = 3 stages);
//   storers    pick the mailbox up, in row slices spread over the NEXT tile's K-steps: saved pre-activation, GELU,
//              GELU', residual and full 256-byte-row stores all happen here, under the consumers' MFMAs.
// All 12 waves meet at ONE barrier per K-step (+2 at the end); the mailbox hand-offs ride on those barriers.
// Non-linear epilogues see the bf16-rounded linear output (what an unfused bf16 Linear -> GELU / + residual sees).
constexpr int MB_PITCH = 2 * BMN + 16;         // mailbox row pitch (bytes): 16-byte aligned rows, 8-byte writes 2-way at worst
constexpr int MB_B = BMN * MB_PITCH;
constexpr int NS2 = 3;
constexpr int NT2 = 64 * 12;

// F8: operands are fp8 bytes (k-contiguous only): a ring stage then holds 128 k per row instead of 64 — the same 128-byte
// rows, pieces and swizzle — and the consumers run 8 block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales, the
// per-tensor scales ride in alpha) instead of 16 bf16 MFMAs per K-step: half the K-steps per tile at the same step cost.
typedef int v8i32 __attribute__((ext_vector_type(8)));
// FMT (F8 only): element formats as compile-time constants, bit 0 = A (0 e4m3, 1 e5m2), bit 1 = B.  r02 branched on p.fmt_a /
// p.fmt_b around every MFMA: hipcc then kept the accumulators in scratch (a 16-register spill + 19 wait states behind each
// v_mfma_scale) and selected the zero C-input with 16 v_cndmask per MFMA — the fp8 GEMM ran at HALF the bf16 rate
// (profiles/r03_a_kernel_bench_fp8_before_quantiser_fix.txt: 8192^3 458 vs 846 TFLOP/s).
template <bool A_KM, bool B_KM, bool F8, int FMT = 0>
__global__ __launch_bounds__(NT2, 1) void gemm_ring2_kernel(GemmParams p, Geo geo) {
  constexpr int NS = NS2, D = NS - 1, PPL = 4;
  constexpr int ES = F8 ? 1 : 2;          // operand element size
  constexpr int BKE = BK * 2 / ES;        // k elements per ring stage (128-byte rows)
  static_assert(!F8 || (!A_KM && !B_KM), "fp8 operands are k-contiguous");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* mbox = smem + NS * STAGE_B;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2;  // 0 consumers, 1 loaders, 2 storers
  PROF_DECL;
  const int w4 = wave & 3;
  const int G = gridDim.x, bid = blockIdx.x;
  const int item0 = (bid & 7) * (G >> 3) + (bid >> 3);

  struct Item { int idx, m0, n0, bz; };
  auto locate = [&](int idx) {
    Item it;
    it.idx = idx;
    it.m0 = it.n0 = it.bz = 0;
    if (idx < geo.items) {
      const int z = idx / geo.tiles, t = idx - z * geo.tiles;
      const int tm = t / geo.tilesN, tn = t - tm * geo.tilesN;
      it.m0 = tm * BMN; it.n0 = tn * BMN; it.bz = z;
    }
    return it;
  };
  const int nk = geo.ksteps;  // no split-K: every item walks all K-steps (>= 2)
  int nitems = 0;
  for (int i = item0; i < geo.items; i += G) ++nitems;
  const int S = nitems * nk;  // K-steps of this workgroup; every wave passes S + 2 barriers

  if (role == 1) {
    // ================================================================================================ loaders
    int voffA[PPL], voffB[PPL], kcA[PPL], kcB[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const int g = w4 * PPL + j;
      kcA[j] = kcB[j] = 0;
      if (A_KM) { const int R = g * 4 + (lane >> 4), c = (lane & 15) ^ swz_km128(R); voffA[j] = (R * (int)p.lda + c * 8) * 2; }
      else { const int R = g * 8 + (lane >> 3), c = (lane & 7) ^ ((R >> 1) & 7); voffA[j] = R * (int)p.lda * ES + c * 16; kcA[j] = c * (16 / ES); }
      if (B_KM) { const int R = g * 4 + (lane >> 4), c = (lane & 15) ^ swz_km128(R); voffB[j] = (R * (int)p.ldb + c * 8) * 2; }
      else { const int R = g * 8 + (lane >> 3), c = (lane & 7) ^ ((R >> 1) & 7); voffB[j] = R * (int)p.ldb * ES + c * 16; kcB[j] = c * (16 / ES); }
    }
    constexpr int OOB = 0x7fffff00;
    const int voffBias = ((w4 & 1) * 64 + lane) * 4;
    const int kstrideA = A_KM ? BK * (int)p.lda * 2 : BK * 2;
    const int kstrideB = B_KM ? BK * (int)p.ldb * 2 : BK * 2;
    auto bytes_left = [](bool km, int64_t ld, int rem, int K) -> unsigned {
      const int64_t elems = rem <= 0 ? 0 : (km ? (int64_t)(K - 1) * ld + rem : (int64_t)(rem - 1) * ld + K);
      return (unsigned)(elems * ES);
    };
    Item L = locate(item0);
    int l_kt = 0, issued = 0;
    __amdgpu_buffer_rsrc_t rsA, rsB, rsBias;
    auto bind = [&]() {
      if (L.idx >= geo.items) return;
      const char* A = reinterpret_cast<const char*>(p.A) + ((int64_t)L.bz * p.sA + (A_KM ? (int64_t)L.m0 : (int64_t)L.m0 * p.lda)) * ES;
      const char* B = reinterpret_cast<const char*>(p.B) + ((int64_t)L.bz * p.sB + (B_KM ? (int64_t)L.n0 : (int64_t)L.n0 * p.ldb)) * ES;
      rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A), 0, bytes_left(A_KM, p.lda, p.M - L.m0, p.K), 0x00020000);
      rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B), 0, bytes_left(B_KM, p.ldb, p.N - L.n0, p.K), 0x00020000);
      const bool hb = (p.flags & GEMM_BIAS) != 0;
      rsBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hb ? p.bias + L.n0 : (const float*)p.A), 0,
                                                 hb ? (unsigned)((p.N - L.n0) * 4) : 0u, 0x00020000);
    };
    bind();
    auto issue = [&]() {
      if (L.idx >= geo.items) return;
      char* st = smem + (issued % NS) * STAGE_B;
      const int k0 = l_kt * BKE;
      const int soA = l_kt * kstrideA, soB = l_kt * kstrideB;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int vo = (!A_KM && k0 + kcA[j] >= p.K) ? OOB : voffA[j] + soA;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, st + (w4 * PPL + j) * 1024), 16, vo, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int vo = (!B_KM && k0 + kcB[j] >= p.K) ? OOB : voffB[j] + soB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, st + TILE_B + (w4 * PPL + j) * 1024), 16, vo, 0, 0, 0);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LDS_PTR(void, st + 2 * TILE_B + (w4 & 1) * 256), 4, voffBias, 0, 0, 0);
      ++issued;
      if (++l_kt == nk) { L = locate(L.idx + G); l_kt = 0; bind(); }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue();
    for (int q = 0; q < S + 2; ++q) {
      const unsigned long long t0 = PROF_T();
      if (q < S) {
        if (issued - (q + 1) >= D - 1) wait_vm<(D - 1) * (2 * PPL + 1)>();
        else wait_vm<0>();
      }
      const unsigned long long t1 = PROF_T();
      __builtin_amdgcn_s_barrier();
      const unsigned long long t2 = PROF_T();
      if (q < S) issue();
      PROF_ADD(4, t1 - t0); PROF_ADD(5, t2 - t1); PROF_ADD(6, PROF_T() - t2);
    }
    PROF_FLUSH();
    return;
  }

  if (role == 0) {
    // ============================================================================================== consumers
    const int wr = w4 >> 1, wc = w4 & 1, h = lane >> 5;
    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    };
    zero_acc();
    f32x4 bias4[2][4];  // bias of this lane's columns, fetched during the tile's last K-step (its ring stage is recycled at the next barrier)
    bool pending = false;
    int kt = 0;
    float alpha = p.alpha;
    if constexpr (F8) {  // per-tensor dequantisation scales (device scalars written by the quantisation kernels)
      if (p.scale_a) alpha *= *p.scale_a;
      if (p.scale_b) alpha *= *p.scale_b;
    }
    for (int q = 0; q < S + 2; ++q) {
      const unsigned long long tb0 = PROF_T();
      __builtin_amdgcn_s_barrier();
      const unsigned long long tb1 = PROF_T();
      PROF_ADD(0, tb1 - tb0); PROF_ADD(3, 1);
      if (pending) {
        // tile finished one step ago: alpha * acc + bias -> bf16 -> mailbox (row m, 4 consecutive columns per write)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            char* rowp = mbox + (wr * 64 + 32 * i + (lane & 31)) * MB_PITCH + (wc * 64 + 32 * j + 4 * h) * 2;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[i][j][4 * g4 + e] * alpha + bias4[j][g4][e]);
              *reinterpret_cast<bf16x4*>(rowp + 16 * g4) = o;
            }
          }
        pending = false;
        PROF_ADD(2, PROF_T() - tb1);
      }
      if (q >= S) continue;
      const unsigned long long tb2 = PROF_T();
      const char* sa = smem + (q % NS) * STAGE_B;
      const char* sb = sa + TILE_B;
      if constexpr (F8) {
        // 32 k bytes per lane and MFMA: lane (r, h) of K-substep s holds k = 64 s + 32 h .. + 31 of its row = the two
        // 16-byte chunks 4 s + 2 h, 4 s + 2 h + 1 of the 128-byte row (chunk ^ ((row >> 1) & 7) as stored by the loaders)
        v8i32 ga[2][2], gb[2][2];
        auto fetch8 = [&](int buf, int s) {
          const int r = lane & 31, c0 = 4 * s + 2 * h;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int row = wr * 64 + 32 * i + r, sw = (row >> 1) & 7;
            const u32x4 lo = *reinterpret_cast<const u32x4*>(sa + row * 128 + ((c0 ^ sw) << 4));
            const u32x4 hi = *reinterpret_cast<const u32x4*>(sa + row * 128 + (((c0 + 1) ^ sw) << 4));
            ga[buf][i] = v8i32{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = wc * 64 + 32 * j + r, sw = (row >> 1) & 7;
            const u32x4 lo = *reinterpret_cast<const u32x4*>(sb + row * 128 + ((c0 ^ sw) << 4));
            const u32x4 hi = *reinterpret_cast<const u32x4*>(sb + row * 128 + (((c0 + 1) ^ sw) << 4));
            gb[buf][j] = v8i32{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
          }
        };
        fetch8(0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int cur = s & 1;
          if (s + 1 < 2) fetch8(cur ^ 1, s + 1);
          __builtin_amdgcn_sched_barrier(0);
          constexpr int ONE = 0x7f7f7f7f;  // E8M0 block scale 2^0 in every byte
          // swapped operands (C^T): source A of the instruction is this kernel's B fragment (cbsz = its format), source B the A
          // fragment (blgp); the first MFMA of a tile starts from C = 0 (inline constant), like the bf16 path
          constexpr int FA = FMT & 1, FB = (FMT >> 1) & 1;
          if (s == 0 && kt == 0) {
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(gb[cur][j], ga[cur][i], zero, FB, FA, 0, ONE, 0, ONE);
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(gb[cur][j], ga[cur][i], acc[i][j], FB, FA, 0, ONE, 0, ONE);
          }
        }
      } else {
        bf16x8 fa[2][2], fb[2][2];
        auto fetch = [&](int buf, int s) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[buf][i] = A_KM ? frag_km(sa, wr * 64 + 32 * i, s, lane) : frag_kc(sa, wr * 64 + 32 * i, s, lane);
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[buf][j] = B_KM ? frag_km(sb, wc * 64 + 32 * j, s, lane) : frag_kc(sb, wc * 64 + 32 * j, s, lane);
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
          const int cur = s & 1;
          if (s + 1 < BK / 16) fetch(cur ^ 1, s + 1);
          __builtin_amdgcn_sched_barrier(0);
          // swapped operands: the accumulator tile is C^T (rows = n on registers, columns = m on lanes)
          if (s == 0 && kt == 0) {  // first MFMA of a tile starts from C = 0 (inline constant): no 64 v_mov to clear acc
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = mma16(fb[cur][j], fa[cur][i], zero);
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = mma16(fb[cur][j], fa[cur][i], acc[i][j]);
          }
        }
      }
      if (++kt == nk) {
        kt = 0;
        pending = true;
        const float* biasl = reinterpret_cast<const float*>(sa + 2 * TILE_B);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) bias4[j][g4] = *reinterpret_cast<const f32x4*>(biasl + wc * 64 + 32 * j + 8 * g4 + 4 * h);
      }
      PROF_ADD(1, PROF_T() - tb2);
    }
    PROF_FLUSH();
    return;
  }

  // ================================================================================================== storers
  // tile dumped after barrier q_d is readable after barrier q_d + 1 and must be drained before the next dump at
  // q_d + nk: slices over the nk - 1 steps in between (the last tile: everything in the one remaining step).
  // A wave owns the 8 row groups {16 k + 4 w4 .. + 3}; the residual / GELU' operand of all 8 is requested at step q_d
  // (while the tile is being dumped), one K-step before the first slice needs it.
  {
    Item cur = locate(item0);
    int done_items = 0;
    const int c8 = (lane & 15) * 8;
    const bool want_x = (p.flags & (GEMM_RESID | GEMM_DGELU)) != 0;
    const bf16_t* X = reinterpret_cast<const bf16_t*>((p.flags & GEMM_RESID) ? p.R : p.aux);
    const int64_t ldxx = (p.flags & GEMM_RESID) ? p.ldr : p.ldaux;
    bf16x8 pre[8];
    for (int q = 0; q < S + 2; ++q) {
      const unsigned long long ts0 = PROF_T();
      __builtin_amdgcn_s_barrier();
      PROF_ADD(7, PROF_T() - ts0);
      const int t = done_items;
      if (t >= nitems) continue;
      const int qd = (t + 1) * nk;
      if (q == qd && want_x) {
        const int col = cur.n0 + c8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int row = cur.m0 + 16 * k + 4 * w4 + (lane >> 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) pre[k][e] = (bf16_t)0.0f;
          if (row < p.M && col < p.N) {
            const bf16x8* xp = reinterpret_cast<const bf16x8*>(X + (int64_t)row * ldxx + col);
            pre[k] = (p.flags & GEMM_RESID) ? *xp : STREAM_LOAD(xp);   // the saved pre-activation is read exactly once
          }
        }
      }
      const bool last = (t + 1 == nitems);
      const int avail = last ? 1 : (nk - 1 < 8 ? nk - 1 : 8);
      const int sl = q - (qd + 1);
      if (sl < 0 || sl >= avail) continue;
      const int k0 = (8 * sl) / avail, k1 = (8 * (sl + 1)) / avail;
      bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + (int64_t)cur.bz * p.sC;
      const int col = cur.n0 + c8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < k0 || k >= k1) continue;
        const int r = 16 * k + 4 * w4 + (lane >> 4);
        const int row = cur.m0 + r;
        const bf16x8 mv = *reinterpret_cast<const bf16x8*>(mbox + r * MB_PITCH + c8 * 2);
        if (row >= p.M || col >= p.N) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)mv[e];
        if (p.flags & GEMM_SAVE_PRE) STREAM_STORE(reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.aux) + (int64_t)row * p.ldaux + col), mv);
        if (p.flags & GEMM_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_for<bf16_t>(v[e]);
        }
        if (p.flags & GEMM_DGELU) {
          bf16x8 a = pre[k];
          if (p.flags & GEMM_RESID) a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.aux) + (int64_t)row * p.ldaux + col);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= dgelu_for<bf16_t>((float)a[e]);
        }
        if (p.flags & GEMM_RESID) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)pre[k][e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
#ifdef DCC_RING_PROFILE
        if (p.flags & 0x2000) continue;  // experiment: the epilogue crew computes but does not store
#endif
        *reinterpret_cast<bf16x8*>(C + (int64_t)row * p.ldc + col) = o;
      }
      if (sl + 1 == avail) { ++done_items; cur = locate(cur.idx + G); }
    }
#ifdef DCC_RING_PROFILE
    if (lane == 0 && wave == 8) atomicAdd(&g_prof[7], prof_acc[7]);
#endif
  }
}

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}

template <typename TC, bool A_KM, bool B_KM>
int launch(const GemmParams& p, const Geo& geo, hipStream_t st) {
  constexpr int NS = 4;
  constexpr int LDS = NS * STAGE_B + CSTAGE_B;
  auto kern = gemm_ring_kernel<TC, A_KM, B_KM, NS>;
  static unsigned long long configured = 0;
  const unsigned long long devbit = dcc_dev_bit();
  if (!(configured & devbit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return DCC_ERR_LAUNCH;
    configured |= devbit;
  }
  int G = num_cus() & ~7;
  if (G < 8) G = 8;
  const int need = (geo.items + 7) & ~7;
  if (G > need) G = need;
  hipLaunchKernelGGL(kern, dim3(G), dim3(NT), LDS, st, p, geo);
  return dcc_check_launch();
}

}  // namespace

#ifdef DCC_RING_PROFILE
extern "C" int dcc_gemm_ring_profile(unsigned long long* out, int reset) {  // debug builds only (-DDCC_RING_PROFILE)
  if (reset) { unsigned long long z[8] = {0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) == hipSuccess ? 0 : -1; }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

int dcc_gemm_ring_launch(const GemmParams& p, int a_km, int b_km, int out_dtype, hipStream_t st) {
  // shapes the ring handles: 16-byte chunked operands/outputs (K a multiple of 8 for k-contiguous operands), 32-bit
  // byte offsets
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
  Geo geo;
  const int tilesM = (p.M + BMN - 1) / BMN;
  geo.tilesN = (p.N + BMN - 1) / BMN;
  geo.tiles = tilesM * geo.tilesN;
  const int64_t items = (int64_t)geo.tiles * p.batch * p.splitk;
  if (items > 0x3fffffff) return DCC_ERR_UNSUPPORTED;
  geo.items = (int)items;
  geo.ksteps = (p.K + BK - 1) / BK;
  geo.per = (geo.ksteps + p.splitk - 1) / p.splitk;
  static const bool ring_v1 = getenv("DCC_RING_V1") != nullptr;   // A/B knob: read once
  if (out_dtype == DCC_BF16 && p.splitk == 1 && !(p.flags & (GEMM_ACCUM | GEMM_COLSUM_A)) && geo.ksteps >= 2 &&
      !ring_v1) {
    constexpr int LDS2 = NS2 * STAGE_B + MB_B;
#define RING2(AK, BK_)                                                                                              \
  do {                                                                                                              \
    auto kern = gemm_ring2_kernel<AK, BK_, false>;                                                                  \
    static unsigned long long cfg = 0;                                                                              \
    if (!(cfg & dcc_dev_bit())) {                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2) != hipSuccess) return DCC_ERR_LAUNCH; \
      cfg |= dcc_dev_bit();                                                                                         \
    }                                                                                                               \
    int G = num_cus() & ~7;                                                                                         \
    if (G < 8) G = 8;                                                                                               \
    const int need = (geo.items + 7) & ~7;                                                                          \
    if (G > need) G = need;                                                                                         \
    hipLaunchKernelGGL(kern, dim3(G), dim3(NT2), LDS2, st, p, geo);                                                 \
    return dcc_check_launch();                                                                                      \
  } while (0)
    if (!a_km && !b_km) RING2(false, false);
    if (!a_km && b_km) RING2(false, true);
    if (a_km && !b_km) RING2(true, false);
    RING2(true, true);
#undef RING2
  }
#define RING(TC_)                                                                   \
  do {                                                                              \
    if (!a_km && !b_km) return launch<TC_, false, false>(p, geo, st);               \
    if (!a_km && b_km) return launch<TC_, false, true>(p, geo, st);                 \
    if (a_km && !b_km) return launch<TC_, true, false>(p, geo, st);                 \
    return launch<TC_, true, true>(p, geo, st);                                     \
  } while (0)
  if (out_dtype == DCC_BF16) RING(bf16_t);
  if (out_dtype == DCC_F32) RING(float);
#undef RING
  return DCC_ERR_UNSUPPORTED;
}

int dcc_gemm_ring_launch_fp8(const GemmParams& p, hipStream_t st) {
  // byte operands, k-contiguous, 16-byte chunks; at least 2 K-steps of 128
  if ((p.lda % 16) || (p.ldb % 16) || (p.sA % 16) || (p.sB % 16) || (p.K % 16) || (p.ldc % 8) || (p.N % 8) || (p.sC % 8)) return DCC_ERR_UNSUPPORTED;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) || ((uintptr_t)p.C & 15)) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & GEMM_RESID) && ((p.ldr % 8) || ((uintptr_t)p.R & 15))) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & (GEMM_SAVE_PRE | GEMM_DGELU)) && ((p.ldaux % 8) || ((uintptr_t)p.aux & 15))) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & GEMM_BIAS) && ((uintptr_t)p.bias & 3)) return DCC_ERR_UNSUPPORTED;
  if (p.splitk != 1 || (p.flags & (GEMM_ACCUM | GEMM_COLSUM_A))) return DCC_ERR_UNSUPPORTED;
  if ((int64_t)p.M * p.lda >= (1LL << 31) || (int64_t)p.N * p.ldb >= (1LL << 31)) return DCC_ERR_UNSUPPORTED;
  Geo geo;
  const int tilesM = (p.M + BMN - 1) / BMN;
  geo.tilesN = (p.N + BMN - 1) / BMN;
  geo.tiles = tilesM * geo.tilesN;
  const int64_t items = (int64_t)geo.tiles * p.batch;
  if (items > 0x3fffffff) return DCC_ERR_UNSUPPORTED;
  geo.items = (int)items;
  geo.ksteps = (p.K + 2 * BK - 1) / (2 * BK);
  geo.per = geo.ksteps;
  if (geo.ksteps < 2) return DCC_ERR_UNSUPPORTED;
  constexpr int LDS2 = NS2 * STAGE_B + MB_B;
  if ((p.fmt_a != 0 && p.fmt_a != 1) || (p.fmt_b != 0 && p.fmt_b != 1)) return DCC_ERR_ARG;
  int G = num_cus() & ~7;
  if (G < 8) G = 8;
  const int need = (geo.items + 7) & ~7;
  if (G > need) G = need;
#define F8_LAUNCH(FMT_)                                                                                                          \
  do {                                                                                                                           \
    auto kern = gemm_ring2_kernel<false, false, true, FMT_>;                                                                     \
    static unsigned long long cfg = 0;                                                                                        \
    if (!(cfg & dcc_dev_bit())) {                                                                                               \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2) != hipSuccess) return DCC_ERR_LAUNCH; \
      cfg |= dcc_dev_bit();                                                                                                      \
    }                                                                                                                            \
    hipLaunchKernelGGL(kern, dim3(G), dim3(NT2), LDS2, st, p, geo);                                                              \
  } while (0)
  switch (p.fmt_a | (p.fmt_b << 1)) {
    case 0: F8_LAUNCH(0); break;
    case 1: F8_LAUNCH(1); break;
    case 2: F8_LAUNCH(2); break;
    default: F8_LAUNCH(3); break;
  }
#undef F8_LAUNCH
  return dcc_check_launch();
}
