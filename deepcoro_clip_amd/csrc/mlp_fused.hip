// Fused MLP forward of an MViT block for gfx950:   y = x + fc2(gelu(fc1(LayerNorm(x))))      (hidden = 4 C)
// (torchvision MultiscaleBlock: `x = x + stochastic_depth(mlp(norm2(x)))`, reached through models/video_encoder.py:232-253.)
//
// One launch replaces LayerNorm + fc1 (+GELU, saved pre-activation) + fc2 (+residual).  The [rows, 4C] hidden tensor never
// comes back from HBM: it is produced and consumed in registers, and only streams OUT (non-temporal) for the backward.
//   * a wave owns 32 token rows.  Their LayerNorm'ed operand lives in registers as MFMA B fragments (C / 4 VGPRs), the
//     [C x 32] output tile as accumulators (C / 2 VGPRs) whose initial value is the residual x + b2;
//   * products are computed transposed (weights = MFMA A operand): H^T[32 hidden x 32 rows] = W1 tile . xn^T leaves a lane with
//     ONE row and 16 hidden units — which is, after GELU and bf16 packing, exactly the B fragment of the second product
//     Y^T += W2[:, 32 hidden] . act^T (cdna guide: accumulator tile as the next MFMA's operand; the k order inside a step is
//     permuted, and the packed W2 image is permuted the same way), so nothing crosses lanes or LDS between the two GEMMs;
//   * the weights are pre-packed ONCE per optimizer step (dcc_mlp_pack_weights) into MFMA-fragment order: every operand
//     read is one lane-linear ds_read_b128 (conflict-free, immediate offsets), and the image streams global -> LDS by
//     LDS-DMA through a ring of three stages shared by the workgroup's waves (counted vmcnt + one raw barrier per stage;
//     the stream runs on across row tiles, the ring never drains);
//   * software pipeline: sub-slot i of the stream holds [W2 tile i-1 | W1 tile i], so iteration i issues GEMM1(i), then GEMM2(i-1)
//     whose MFMAs cover the bias/GELU/packing VALU work of tile i;
//   * pre-activation / activation / LayerNorm output leave as complete 128-byte lines through wave-private LDS strips
//     (a lane owns one row: direct stores would be 32 partial lines per instruction).
// Per 32 rows: 2 x 2 x 32 x C x 4C FLOP for 128 C^2 x ... bytes of L2 -> LDS weight traffic; HBM bytes: x in, y / xn / pre / act out.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

#ifdef DCC_MLP_PROFILE   // phase timers (tools/mlp_profile.py; never in the shipped library)
__device__ unsigned long long g_mprof[8];  // [0] vmcnt wait [1] barrier [2] GEMM1 + DMA issue [3] GEMM2 || epilogue [4] swaps + stores [5] iterations [6] tile head [7] y store
#define MPROF_T() __builtin_amdgcn_s_memtime()
#define MPROF_ADD(i, v) mprof[i] += (unsigned long long)(v)
#else
#define MPROF_T() 0ull
#define MPROF_ADD(i, v) do { } while (0)
#endif

struct MlpParams {
  const bf16_t* x;
  int64_t ldx;
  const float* gamma;
  const float* beta;
  float eps;
  const bf16_t* wpk;   // packed image (dcc_mlp_pack_weights)
  const float* b1;
  const float* b2;
  bf16_t* y;
  int64_t ldy;
  bf16_t* xn;          // [M, C] LayerNorm output (the weight-gradient GEMM of fc1 reads it), may be null
  float* mean;         // [M] statistics for the LayerNorm backward, may be null
  float* rstd;
  bf16_t* pre;         // [M, 4C] pre-activation (bf16(fc1 + b1)), may be null
  bf16_t* act;         // [M, 4C] gelu(pre), may be null
  int M, tiles;
  int residual;
};

template <int N> __device__ __forceinline__ void mwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ uint32_t mpack2(float a, float b) {
  bf16x2 t;
  t[0] = (bf16_t)a;
  t[1] = (bf16_t)b;
  return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

constexpr int MSTG_PITCH = 144;                // strip rows: 128 bytes + 16

template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}
// An MFMA A fragment read into ACCUMULATOR registers by hand (the compiler neither counts nor waits for it: mlds_wait below).  Why: an
// in-flight MFMA keeps reading its A / B registers, and hipcc hands a fragment's VGPRs to the next vector instruction's result as soon
// as the MFMA has issued — the write then stalls until the MFMA is done (measured: +47 cycles per MFMA, nothing of the epilogue's
// vector work overlapped).  Vector instructions cannot target AGPRs, so fragments kept there are never in their way.
template <int OFF>
__device__ __forceinline__ bf16x8 mlds_read_a(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void mlds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// C: channels (96 | 192 | 384); NW: waves per workgroup; HS: hidden units per ring stage (32 | 64); MINW: waves per SIMD;
// SR: rows of a wave's store-staging strip (32: a whole row tile per LDS round trip; 16 where the ring leaves no room)
// HAND: GEMM2 || epilogue placed instruction group by instruction group (one wave per SIMD: nothing else hides the vector work);
// else eight compiler-scheduled groups (two waves per SIMD cover each other: measured faster at C = 96)
template <int C, int NW, int HS, int MINW, int SR, bool HAND>
__global__ __launch_bounds__(NW * 64, MINW) void ln_mlp_kernel(MlpParams p) {
  constexpr int HID = 4 * C, KS = C / 16, MT = C / 32, U = HS / 32, NC = HID / 32;
  constexpr int HALF = 64 * C;                 // bytes of one packed W1 (or W2) tile of 32 hidden units
  constexpr int SUBB = 2 * HALF;               // sub-slot i = [W2 tile i-1 | W1 tile i]
  constexpr int STAGE = U * SUBB, NPIECE = STAGE / 1024, PW = NPIECE / NW;
  constexpr int NSTG = 3;
  constexpr int NSTT = (NC + U) / U;           // stages per row tile (sub-slots 0 .. NC)
  constexpr int NT = NW * 64;
  static_assert(NPIECE % NW == 0 && (NC % 2) == 0 && 2 * PW == KS, "geometry");
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  float* b1_l = reinterpret_cast<float*>(lds + NSTG * STAGE);
  constexpr bool CONST_LDS = SR == 32;   // b2 / gamma / beta in LDS (else straight from global: once per row tile)
  float* b2_l = b1_l + HID;
  float* gam_l = b2_l + C;
  float* bet_l = gam_l + C;
  constexpr int STG_WAVE = SR * MSTG_PITCH;
  char* stg = reinterpret_cast<char*>(CONST_LDS ? bet_l + C : b1_l + HID) + wave * STG_WAVE;
  for (int i = tid; i < HID; i += NT) b1_l[i] = p.b1[i];
  if constexpr (CONST_LDS) {
    for (int i = tid; i < C; i += NT) {
      b2_l[i] = p.b2[i];
      gam_l[i] = p.gamma[i];
      bet_l[i] = p.beta[i];
    }
  }
  const float* b2_s = CONST_LDS ? b2_l : p.b2;
  const float* gam_s = CONST_LDS ? gam_l : p.gamma;
  const float* bet_s = CONST_LDS ? bet_l : p.beta;
  __syncthreads();

  // ------------------------------------------------------------------------------------------- weight stream
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.wpk), 0, (unsigned)(NSTT * STAGE), 0x00020000);
  int src_stage = 0, dst_slot = 0;   // next stage to issue: source stage (mod NSTT) and ring slot (mod NSTG)
  auto issue_piece = [&](int j) {
    const int g = wave + NW * j;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, lds + dst_slot * STAGE + g * 1024), 16,
                                             src_stage * STAGE + g * 1024 + lane * 16, 0, 0, 0);
  };
  auto issue_done = [&]() {
    src_stage = src_stage + 1 == NSTT ? 0 : src_stage + 1;
    dst_slot = dst_slot + 1 == NSTG ? 0 : dst_slot + 1;
  };
#pragma unroll
  for (int d = 0; d < 2; ++d) {
#pragma unroll
    for (int j = 0; j < PW; ++j) issue_piece(j);
    issue_done();
  }
#ifdef DCC_MLP_PROFILE
  unsigned long long mprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  int cur_slot = NSTG - 1;   // advanced before use
  int issued = 0;            // stage advances so far (0, 1: the ring is still filling, fewer stores are in flight)

  // staging addresses: this lane WRITES row r % SR (in pass r / SR) at byte 64 j + 32 P + 16 h, READS rows lane / 8 + 8 k at byte 16 (lane & 7)
  char* stg_w = stg + (r % SR) * MSTG_PITCH + 16 * h;
  const char* stg_r = stg + (lane >> 3) * MSTG_PITCH + 16 * (lane & 7);

  for (int t0 = blockIdx.x * NW; t0 < p.tiles; t0 += gridDim.x * NW) {
    const int t = t0 + wave;
    const bool tile_live = t < p.tiles;                     // wave-uniform
    const bool exact = tile_live && 32 * t + 32 <= p.M;     // every store of this wave executes: the counted waits hold
    const int row = 32 * t + r;
    const bool live = row < p.M;
    const int64_t lrow = live ? row : p.M - 1;
    const int srow0 = 32 * t + (lane >> 3);

    // a pair of 32-column tiles (64 columns = one 128-byte line per row) leaves through the strip: c[j][P] = this lane's 16
    // bytes at column 32 j + 16 P + 8 h of its row
    auto store_pair = [&](bf16_t* base, int64_t ld, int col0, const u32x4 (&c)[2][2], bool nt) {
      const int scol = col0 + 8 * (lane & 7);
#pragma unroll
      for (int q = 0; q < 32 / SR; ++q) {   // one LDS round trip per SR rows: all their lanes write, then SR / 8 reads + stores
        if (SR == 32 || (r / SR) == q) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int P = 0; P < 2; ++P) *reinterpret_cast<u32x4*>(stg_w + 64 * j + 32 * P) = c[j][P];
        }
        // lanes exchange data through the strip: the compiler must not reason about these bytes per thread (it otherwise keeps a
        // lane's EARLIER read of the same address when that lane's own writes were masked off); the hardware runs a wave's LDS
        // operations in order, so a compiler-level barrier is all the ordering this needs
        asm volatile("" ::: "memory");
        u32x4 o[SR / 8];
#pragma unroll
        for (int k = 0; k < SR / 8; ++k) o[k] = *reinterpret_cast<const u32x4*>(stg_r + 8 * k * MSTG_PITCH);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < SR / 8; ++k) {
          const int srow = srow0 + SR * q + 8 * k;
          if (srow < p.M) {
            u32x4* dst = reinterpret_cast<u32x4*>(base + (int64_t)srow * ld + scol);
            if (nt) __builtin_nontemporal_store(o[k], dst);
            else *dst = o[k];
          }
        }
      }
    };
    auto store_single = [&](bf16_t* base, int64_t ld, int col0, const u32x4 (&c)[2], bool nt) {
      if (!live) return;
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        u32x4* dst = reinterpret_cast<u32x4*>(base + (int64_t)row * ld + col0 + 16 * P + 8 * h);
        if (nt) __builtin_nontemporal_store(c[P], dst);
        else *dst = c[P];
      }
    };

    // ----------------------------------------------------------------------------------- x, LayerNorm, Y init
    const unsigned long long tp0 = MPROF_T();
    u32x4 xf[KS];   // raw x, then LayerNorm(x): this lane's row, channels 16 s + 8 h .. + 7
    {
      const bf16_t* xr = p.x + lrow * p.ldx + 8 * h;
#pragma unroll
      for (int s = 0; s < KS; ++s) xf[s] = *reinterpret_cast<const u32x4*>(xr + 16 * s);
    }
    f32x16 Y[MT];
    // accumulator register 4 g + i of tile mt <-> channel 32 mt + 8 g + 4 h + i.  The fragment piece s = 2 mt + g' holds channels
    // 16 s + 8 h + 0..7: the lower lane keeps its dwords 0, 1 (g = 2 g') and takes the upper lane's dwords 0, 1 (g = 2 g' + 1), the
    // upper lane takes the lower lane's dwords 2, 3 (g = 2 g') and keeps its own (g = 2 g' + 1): two half swaps per piece.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const u32x4 d = xf[2 * mt + gp];
        const auto e = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
        const auto f = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int g = 2 * gp + o;
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(b2_s + 32 * mt + 8 * g + 4 * h);
          const uint32_t e_ = o ? e[1] : e[0], f_ = o ? f[1] : f[0];
          const float rs_ = p.residual ? 1.0f : 0.0f;
          Y[mt][4 * g + 0] = b4[0] + rs_ * lo_f(e_);
          Y[mt][4 * g + 1] = b4[1] + rs_ * hi_f(e_);
          Y[mt][4 * g + 2] = b4[2] + rs_ * lo_f(f_);
          Y[mt][4 * g + 3] = b4[3] + rs_ * hi_f(f_);
        }
      }
    }
    {
      float s1 = 0.0f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) s1 += lo_f(xf[s][q]) + hi_f(xf[s][q]);
      s1 += __shfl_xor(s1, 32);
      const float mean = s1 * (1.0f / C);
      float s2 = 0.0f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a = lo_f(xf[s][q]) - mean, b = hi_f(xf[s][q]) - mean;
          s2 += a * a + b * b;
        }
      s2 += __shfl_xor(s2, 32);
      const float rstd = __builtin_amdgcn_rsqf(s2 * (1.0f / C) + p.eps);
      if (p.mean != nullptr && live && h == 0) {
        p.mean[row] = mean;
        p.rstd[row] = rstd;
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam_s + 16 * s + 8 * h), g1 = *reinterpret_cast<const f32x4*>(gam_s + 16 * s + 8 * h + 4);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(bet_s + 16 * s + 8 * h), c1 = *reinterpret_cast<const f32x4*>(bet_s + 16 * s + 8 * h + 4);
        u32x4 o;
        o[0] = mpack2((lo_f(xf[s][0]) - mean) * rstd * g0[0] + c0[0], (hi_f(xf[s][0]) - mean) * rstd * g0[1] + c0[1]);
        o[1] = mpack2((lo_f(xf[s][1]) - mean) * rstd * g0[2] + c0[2], (hi_f(xf[s][1]) - mean) * rstd * g0[3] + c0[3]);
        o[2] = mpack2((lo_f(xf[s][2]) - mean) * rstd * g1[0] + c1[0], (hi_f(xf[s][2]) - mean) * rstd * g1[1] + c1[1]);
        o[3] = mpack2((lo_f(xf[s][3]) - mean) * rstd * g1[2] + c1[2], (hi_f(xf[s][3]) - mean) * rstd * g1[3] + c1[3]);
        xf[s] = o;
      }
    }
    if (p.xn != nullptr && tile_live) {   // LayerNorm output: pieces s = 4 a .. 4 a + 3 are 64 channels = one line per row
#pragma unroll
      for (int a = 0; a < KS / 4; ++a) {
        const u32x4 c[2][2] = {{xf[4 * a], xf[4 * a + 1]}, {xf[4 * a + 2], xf[4 * a + 3]}};
        store_pair(p.xn, C, 64 * a, c, true);
      }
      if constexpr ((KS % 4) != 0) {      // C = 96: the last 32 channels
        const u32x4 c[2] = {xf[KS - 2], xf[KS - 1]};
        store_single(p.xn, C, 16 * (KS - 2), c, true);
      }
    }

    MPROF_ADD(6, MPROF_T() - tp0);
    // --------------------------------------------------------------------------------------------- main loop
    f32x16 acc;
    u32x4 actf[2][2];     // [parity of the hidden tile][k-step]: gelu(pre) packed = B fragments of GEMM2
    u32x4 cpre[2][2], cact[2][2];   // [parity][P]: 16-byte store pieces (columns 16 P + 8 h of the tile)
    const char* sub = nullptr;

    // the next stage of the stream becomes readable (its refill pieces go out between the MFMAs of GEMM1 below)
    auto advance = [&]() {
      const unsigned long long ta0 = MPROF_T();
      if (issued >= 2 && exact) {
        // younger than this stage's DMA: the next stage's PW pieces + the output stores of the two iterations in between
        if (p.act != nullptr) mwait_vm<PW + (U == 2 ? 16 : 8)>();
        else if (p.pre != nullptr) mwait_vm<PW + (U == 2 ? 8 : 4)>();
        else mwait_vm<PW>();
      } else {
        mwait_vm<PW>();
      }
      const unsigned long long ta1 = MPROF_T();
      __builtin_amdgcn_s_barrier();
      MPROF_ADD(0, ta1 - ta0);
      MPROF_ADD(1, MPROF_T() - ta1);
      cur_slot = cur_slot + 1 == NSTG ? 0 : cur_slot + 1;
      ++issued;
    };
    // iteration i: sub-slot i = [W2 tile i-1 | W1 tile i]; E = parity of i.  GEMM1(i), then GEMM2(i-1) in eight groups of MFMAs with
    // an eighth of tile i's epilogue (round, GELU, pack) behind each group: the matrix pipe drains under the vector work.
    auto body = [&](int i, auto Ec, auto G1c, auto G2c) {
      constexpr int E = decltype(Ec)::value;
      constexpr bool G1 = decltype(G1c)::value, G2 = decltype(G2c)::value;
      constexpr bool ADV = (U == 1 || E == 0);
      if (ADV) {
        advance();
        sub = lds + cur_slot * STAGE;
      }
      const char* ss = sub + ((U == 2 && E == 1) ? SUBB : 0) + lane * 16;
      const unsigned long long tb0 = MPROF_T();
      // Both GEMM phases are written plainly and ORDERED by scheduling groups (cdna guide T19): one MFMA per group, and in its shadow
      // the operand read of a later MFMA plus a share of the phase's other work (LDS-DMA issues in GEMM1, the epilogue's vector
      // instructions in GEMM2).  An in-order wave that issues MFMAs back to back stalls on the second one: the work must sit BETWEEN them.
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (G1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(b1_l + 32 * i + 8 * g + 4 * h);
          acc[4 * g] = b4[0]; acc[4 * g + 1] = b4[1]; acc[4 * g + 2] = b4[2]; acc[4 * g + 3] = b4[3];
        }
        // source order = issue order: an LDS-DMA writes LDS, so the compiler keeps it between the operand reads it was written between
        constexpr int AHEAD = 3;                                // operand reads in flight ahead of their MFMA
        bf16x8 wf[AHEAD + 1];
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) wf[s] = *reinterpret_cast<const bf16x8*>(ss + HALF + s * 1024);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (s + AHEAD < KS) wf[(s + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(ss + HALF + (s + AHEAD) * 1024);
          if (ADV && (s & 1)) issue_piece(s >> 1);               // PW == KS / 2: a DMA piece behind every other MFMA
          acc = mma16(wf[s % (AHEAD + 1)], __builtin_bit_cast(bf16x8, xf[s]), acc);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + AHEAD, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (s + AHEAD < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (ADV && (s & 1)) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      } else if constexpr (ADV) {
#pragma unroll
        for (int j = 0; j < PW; ++j) issue_piece(j);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ADV) issue_done();
      const unsigned long long tb1 = MPROF_T();
      MPROF_ADD(2, tb1 - tb0);
      MPROF_ADD(5, 1);
      uint32_t pk[8], ak[8];
      if constexpr (HAND) {
        // GEMM2(i-1) with the epilogue of tile i cut into 24 parts (8 register pairs x {round + polynomial, exp / rcp, pack}) placed
        // by hand between the MFMAs; the W2 fragments come through AGPRs (mlds_read_a), AHEAD reads in flight
        constexpr int NM2 = 2 * MT, AHEAD = 3, NPART = 24;
        float gx[2], gu[2];
        bf16x8 w2f[AHEAD + 1];
        const uint32_t a2 = (uint32_t)(uintptr_t)LDS_PTR(const char, ss);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (G2) {
          static_for<AHEAD>([&](auto mc) { constexpr int m = decltype(mc)::value; w2f[m] = mlds_read_a<m * 1024>(a2); });
        }
        static_for<NPART>([&](auto pc) {
          constexpr int P = decltype(pc)::value;
          if constexpr (G2) {
            static_for<NM2>([&](auto mc) {
              constexpr int m = decltype(mc)::value;
              if constexpr (m * NPART / NM2 == P) {
                if constexpr (m + AHEAD < NM2) w2f[(m + AHEAD) % (AHEAD + 1)] = mlds_read_a<(m + AHEAD) * 1024>(a2);
                mlds_wait<(m + AHEAD < NM2) ? AHEAD : NM2 - 1 - m>();
                __builtin_amdgcn_sched_barrier(0);
                Y[m >> 1] = mma16(w2f[m % (AHEAD + 1)], __builtin_bit_cast(bf16x8, actf[E ^ 1][m & 1]), Y[m >> 1]);
              }
            });
          }
          if constexpr (G1) {
            // round the pre-activation once (what the unfused bf16 Linear hands to GELU), GELU on the rounded value:
            // gelu(x) = x / (1 + exp2(x (c1 + c2 x^2)))   (gelu_bf16 of common.h with the exponent's scale folded into c1, c2)
            constexpr int q = P / 3, k = P % 3;
            if constexpr (k == 0) {
              pk[q] = mpack2(acc[2 * q], acc[2 * q + 1]);
              gx[0] = lo_f(pk[q]);
              gx[1] = hi_f(pk[q]);
#ifdef DCC_EXACT_GELU
              gu[0] = gelu_f(gx[0]);
              gu[1] = gelu_f(gx[1]);
#else
              gu[0] = gx[0] * (-2.30876515f - 0.10012561f * gx[0] * gx[0]);
              gu[1] = gx[1] * (-2.30876515f - 0.10012561f * gx[1] * gx[1]);
#endif
            } else if constexpr (k == 1) {
#ifndef DCC_EXACT_GELU
              gu[0] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gu[0]));
              gu[1] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gu[1]));
#endif
            } else {
#ifdef DCC_EXACT_GELU
              ak[q] = mpack2(gu[0], gu[1]);
#else
              ak[q] = mpack2(gx[0] * gu[0], gx[1] * gu[1]);
#endif
              asm volatile("" : "+v"(ak[q]));   // needed only an iteration later: without a use here the compiler sinks the work out
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        constexpr int NM2 = 2 * MT, PER = (NM2 + 7) / 8;
        bf16x8 fb[2][PER];
        auto lo = [](int q) { return q * NM2 / 8; };
        if constexpr (G2) {
#pragma unroll
          for (int m = lo(0); m < lo(1); ++m) fb[0][m - lo(0)] = *reinterpret_cast<const bf16x8*>(ss + m * 1024);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if constexpr (G2) {
            if (q + 1 < 8) {
#pragma unroll
              for (int m = lo(q + 1); m < lo(q + 2); ++m) fb[(q + 1) & 1][m - lo(q + 1)] = *reinterpret_cast<const bf16x8*>(ss + m * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = lo(q); m < lo(q + 1); ++m)
              Y[m >> 1] = mma16(fb[q & 1][m - lo(q)], __builtin_bit_cast(bf16x8, actf[E ^ 1][m & 1]), Y[m >> 1]);
          }
          if constexpr (G1) {
            pk[q] = mpack2(acc[2 * q], acc[2 * q + 1]);
            ak[q] = mpack2(gelu_for<bf16_t>(lo_f(pk[q])), gelu_for<bf16_t>(hi_f(pk[q])));
            asm volatile("" : "+v"(ak[q]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      {
        const unsigned long long tb2 = MPROF_T();
        MPROF_ADD(3, tb2 - tb1);
        if constexpr (G1) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            actf[E][s2][0] = ak[4 * s2]; actf[E][s2][1] = ak[4 * s2 + 1]; actf[E][s2][2] = ak[4 * s2 + 2]; actf[E][s2][3] = ak[4 * s2 + 3];
          }
          if (p.pre != nullptr) {
#pragma unroll
            for (int P = 0; P < 2; ++P) {
              const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * P], pk[4 * P + 2], false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * P + 1], pk[4 * P + 3], false, false);
              cpre[E][P][0] = s0[0]; cpre[E][P][1] = s1[0]; cpre[E][P][2] = s0[1]; cpre[E][P][3] = s1[1];
            }
          }
          if (p.act != nullptr) {
#pragma unroll
            for (int P = 0; P < 2; ++P) {
              const auto s0 = __builtin_amdgcn_permlane32_swap(ak[4 * P], ak[4 * P + 2], false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(ak[4 * P + 1], ak[4 * P + 3], false, false);
              cact[E][P][0] = s0[0]; cact[E][P][1] = s1[0]; cact[E][P][2] = s0[1]; cact[E][P][3] = s1[1];
            }
          }
          if constexpr (E == 1) {
            if (tile_live) {
              if (p.pre != nullptr) store_pair(p.pre, HID, 32 * (i - 1), cpre, true);
              if (p.act != nullptr) store_pair(p.act, HID, 32 * (i - 1), cact, true);
            }
          }
        }
        MPROF_ADD(4, MPROF_T() - tb2);
      }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using Yes = std::true_type;
    using No = std::false_type;
    body(0, T0{}, Yes{}, No{});
    for (int i = 1; i + 1 < NC; i += 2) {
      body(i, T1{}, Yes{}, Yes{});
      body(i + 1, T0{}, Yes{}, Yes{});
    }
    body(NC - 1, T1{}, Yes{}, Yes{});
    body(NC, T0{}, No{}, Yes{});

    // ------------------------------------------------------------------------------------------------ y
    const unsigned long long ty0 = MPROF_T();
    if (tile_live) {
      u32x4 cy[2][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        uint32_t pk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pk[q] = mpack2(Y[mt][2 * q], Y[mt][2 * q + 1]);
#pragma unroll
        for (int P = 0; P < 2; ++P) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * P], pk[4 * P + 2], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * P + 1], pk[4 * P + 3], false, false);
          cy[mt & 1][P][0] = s0[0]; cy[mt & 1][P][1] = s1[0]; cy[mt & 1][P][2] = s0[1]; cy[mt & 1][P][3] = s1[1];
        }
        if (mt & 1) store_pair(p.y, p.ldy, 32 * (mt - 1), cy, false);
        else if (mt == MT - 1) store_single(p.y, p.ldy, 32 * mt, cy[0], false);
      }
    }
    MPROF_ADD(7, MPROF_T() - ty0);
  }
  mwait_vm<0>();   // LDS-DMA pieces still in flight must land before the workgroup's LDS is handed on
#ifdef DCC_MLP_PROFILE
  if (lane == 0 && wave == 0) {
    for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_mprof[i_], mprof[i_]);
  }
#endif
}

// w1 [4C, C], w2 [C, 4C] (row-major bf16) -> the stream image: NSTT * STAGE bytes; sub-slot i (128 C bytes) = [W2 tile i-1 | W1 tile i],
// each tile as 1-KB MFMA fragments in lane order (lane l = (r, h), 8 bf16 per lane):
//   W1 tile c, k-step s:          element j = w1[32 c + r][16 s + 8 h + j]
//   W2 tile c, (mt, k-step s2):   element j = w2[32 mt + r][32 c + 16 s2 + 8 (j >> 2) + 4 h + (j & 3)]   (accumulator-as-operand k order)
__global__ void mlp_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, bf16_t* __restrict__ out, int C, int64_t total8) {
  const int HID = 4 * C, half8 = 4 * C;   // 16-byte units per packed tile (64 C bytes)
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total8; u += (int64_t)gridDim.x * blockDim.x) {
    const int64_t slot = u / (2 * half8);
    const int in = (int)(u - slot * 2 * half8);
    const bool is_w1 = in >= half8;
    const int v = is_w1 ? in - half8 : in;
    const int piece = v >> 6, lane = v & 63, r = lane & 31, h = lane >> 5;
    bf16x8 o;
    if (is_w1) {
      const int c = (int)slot;
      if (c * 32 < HID) {
        const bf16_t* src = w1 + (int64_t)(32 * c + r) * C + 16 * piece + 8 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = src[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.0f;
      }
    } else {
      const int c = (int)slot - 1;
      if (c >= 0 && c * 32 < HID) {
        const int mt = piece >> 1, s2 = piece & 1;
        const bf16_t* src = w2 + (int64_t)(32 * mt + r) * HID + 32 * c + 16 * s2 + 4 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = src[8 * (j >> 2) + (j & 3)];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.0f;
      }
    }
    *reinterpret_cast<bf16x8*>(out + 8 * u) = o;
  }
}

template <int C> struct MlpCfg;
template <> struct MlpCfg<96> { static constexpr int NW = 8, HS = 64, MINW = 2, SR = 32; static constexpr bool HAND = false; };
template <> struct MlpCfg<192> { static constexpr int NW = 4, HS = 32, MINW = 1, SR = 32; static constexpr bool HAND = true; };
template <> struct MlpCfg<384> { static constexpr int NW = 4, HS = 32, MINW = 1, SR = 16; static constexpr bool HAND = true; };

template <int C> constexpr int mlp_stages() { return (4 * C / 32 + MlpCfg<C>::HS / 32) / (MlpCfg<C>::HS / 32); }
template <int C> constexpr int mlp_stage_bytes() { return (MlpCfg<C>::HS / 32) * 128 * C; }

template <int C>
int mlp_launch(const MlpParams& p, hipStream_t st) {
  using Cfg = MlpCfg<C>;
  constexpr int LDS = 3 * mlp_stage_bytes<C>() + (4 * C + (Cfg::SR == 32 ? 3 * C : 0)) * 4 + Cfg::NW * Cfg::SR * MSTG_PITCH;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = ln_mlp_kernel<C, Cfg::NW, Cfg::HS, Cfg::MINW, Cfg::SR, Cfg::HAND>;
  static unsigned long long configured = 0;
  const unsigned long long devbit = dcc_dev_bit();
  if (!(configured & devbit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return DCC_ERR_LAUNCH;
    configured |= devbit;
  }
  int per_cu = (160 * 1024) / LDS;
  if (per_cu > 1) per_cu = 1;   // one ring per CU: the workgroup's waves already fill the SIMDs to MINW
  int gx = 256 * per_cu;
  const int need = (p.tiles + Cfg::NW - 1) / Cfg::NW;
  if (gx > need) gx = need;
  hipLaunchKernelGGL(kern, dim3(gx), dim3(Cfg::NW * 64), LDS, st, p);
  return dcc_check_launch();
}

}  // namespace

#ifdef DCC_MLP_PROFILE
extern "C" int dcc_mlp_profile(unsigned long long* out, int reset) {  // debug builds only (-DDCC_MLP_PROFILE)
  if (reset) { unsigned long long z[8] = {0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_mprof), z, sizeof(z)) == hipSuccess ? 0 : -1; }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mprof), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// Size in bytes of the packed image for `C` channels (0: C outside the kernel's domain).
extern "C" int dcc_mlp_packed_bytes(int C) {
  if (C == 96) return mlp_stages<96>() * mlp_stage_bytes<96>();
  if (C == 192) return mlp_stages<192>() * mlp_stage_bytes<192>();
  if (C == 384) return mlp_stages<384>() * mlp_stage_bytes<384>();
  return 0;
}

extern "C" int dcc_mlp_pack_weights(const void* w1, const void* w2, void* packed, int C, void* stream) {
  if (!w1 || !w2 || !packed) return DCC_ERR_ARG;
  const int64_t bytes = dcc_mlp_packed_bytes(C);
  if (bytes == 0) return DCC_ERR_UNSUPPORTED;
  if (((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)packed & 15)) return DCC_ERR_UNSUPPORTED;
  const int64_t total8 = bytes / 16;
  int64_t blocks = (total8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w1, (const bf16_t*)w2, (bf16_t*)packed, C, total8);
  return dcc_check_launch();
}

extern "C" int dcc_ln_mlp_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* packed,
                              const float* b1, const float* b2, void* y, int64_t ldy, void* xn, float* mean, float* rstd, void* pre,
                              void* act, int64_t M, int C, int residual, void* stream) {
  if (!x || !gamma || !beta || !packed || !b1 || !b2 || !y || M <= 0) return DCC_ERR_ARG;
  if ((mean == nullptr) != (rstd == nullptr)) return DCC_ERR_ARG;
  if (act != nullptr && pre == nullptr) return DCC_ERR_ARG;
  if ((ldx % 8) || (ldy % 8) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)packed & 15)) return DCC_ERR_UNSUPPORTED;
  if (((uintptr_t)xn & 15) || ((uintptr_t)pre & 15) || ((uintptr_t)act & 15)) return DCC_ERR_UNSUPPORTED;
  if (M >= (1LL << 31) - 64) return DCC_ERR_UNSUPPORTED;
  MlpParams p;
  p.x = (const bf16_t*)x; p.ldx = ldx; p.gamma = gamma; p.beta = beta; p.eps = eps; p.wpk = (const bf16_t*)packed;
  p.b1 = b1; p.b2 = b2; p.y = (bf16_t*)y; p.ldy = ldy; p.xn = (bf16_t*)xn; p.mean = mean; p.rstd = rstd;
  p.pre = (bf16_t*)pre; p.act = (bf16_t*)act; p.M = (int)M; p.tiles = (int)((M + 31) / 32); p.residual = residual;
  if (C == 96) return mlp_launch<96>(p, (hipStream_t)stream);
  if (C == 192) return mlp_launch<192>(p, (hipStream_t)stream);
  if (C == 384) return mlp_launch<384>(p, (hipStream_t)stream);
  return DCC_ERR_UNSUPPORTED;
}
