// Fused LayerNorm + MLP forward at C = 384 (the ten stage-2 MViT blocks + block 3's output width), 16-row tiles:
//   y = [x +] fc2(gelu(fc1(LayerNorm(x))))       (same contract and outputs as mlp_fused.hip; dcc_ln_mlp_fwd dispatches here)
//
// Why a second kernel: with 32-row tiles a wave holds 96 (operand) + 192 (output accumulators) registers per lane, i.e. ONE wave per
// SIMD, and that single in-order wave cannot overlap its own epilogue with its own MFMAs (measured: 243-258 us against 195-212 for
// LayerNorm + two GEMMs; profiles/r04_mlp_phase_timers_v1.txt).  Here a wave owns 16 rows on v_mfma_f32_16x16x32_bf16:
// 48 + 96 registers, TWO waves per SIMD, so one wave's bias / GELU / packing / stores run beside its partner's matrix work.
//   * lane (n = lane & 15, g = lane >> 4): row n of the tile; B fragments of fc1 = LayerNorm(x)[n][32 s + 8 g ..] (12 k-steps),
//     H^T tiles [16 hidden x 16 rows] come out as 4 values per lane (hidden 4 g + r), two tiles (32 hidden) = the 8 k-values of ONE
//     B fragment of fc2 in the order {4 g + 0..3, 16 + 4 g + 0..3} — the packed W2 image uses the same order;
//   * per 32 hidden units: 24 + 24 MFMAs (16 cycles each), 48 weight fragments of 1 KB from the LDS ring (lane-linear reads): twice
//     the LDS bytes per FLOP of the 32-row kernel, i.e. ~50 % of the LDS read rate at the MFMA rate this reaches;
//   * 8 waves share one 3-stage ring of 48-KB sub-slots [W2 tile i-1 | W1 tile i] filled by LDS-DMA (6 pieces per wave and stage),
//     counted vmcnt + one raw barrier per stage, the stream runs on across row tiles;
//   * software pipeline inside a wave as in mlp_fused.hip: GEMM1(i), GEMM2(i-1), epilogue(i);
//   * outputs leave as complete 128-byte lines through wave-private 8-row LDS strips.
#include "mlp_common.h"
#include <stdlib.h>

namespace {

#ifdef DCC_MLP_PROFILE   // phase timers (tools/mlp_profile.py; never in the shipped library): [16 * late + phase]
__device__ unsigned long long g_mprof16[32];  // phase: 0 vmcnt wait, 1 barrier, 2 epilogue, 3 GEMM1 + DMA issue, 4 GEMM2, 5 iterations, 6 tile head, 7 y store
#define MP16_T() __builtin_amdgcn_s_memtime()
#define MP16_ADD(i, v) mprof[i] += (unsigned long long)(v)
#else
#define MP16_T() 0ull
#define MP16_ADD(i, v) do { } while (0)
#endif

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4_t mma32(const bf16x8& a, const bf16x8& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int C16 = 384, HID16 = 4 * C16, NC16 = HID16 / 32;
constexpr int KS16 = C16 / 32;            // k-steps of fc1 (12)
constexpr int MT16 = C16 / 16;            // 16-channel output tiles of fc2 (24)
constexpr int HALF16 = 64 * C16;          // bytes of a packed W1 (or W2) tile set of 32 hidden units (24 fragments)
constexpr int SUBB16 = 2 * HALF16;        // sub-slot i = [W2 set i-1 | W1 set i]
constexpr int NW16 = 8;
constexpr int NSLOT16 = 5;                // ring of FIVE half sub-slots (24 KB): two in use, one landed, two being refilled
constexpr int PH16 = HALF16 / 1024 / NW16;   // 3 LDS-DMA pieces per wave and half
constexpr int NSTT16 = NC16 + 1;          // sub-slots per row tile
constexpr int NHALF16 = 2 * NSTT16;       // halves of the stream per row tile
constexpr int SROWS16 = 16;               // the wave's store-staging strip holds the whole 16-row tile
constexpr int STRIP16 = SROWS16 * MSTG_PITCH;
constexpr int LDS16 = NSLOT16 * HALF16 + (HID16 + 3 * C16) * 4 + NW16 * STRIP16;
static_assert(LDS16 <= 160 * 1024, "LDS budget");

__global__ __launch_bounds__(NW16 * 64, 2) void ln_mlp16_kernel(MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  float* b1_l = reinterpret_cast<float*>(lds + NSLOT16 * HALF16);
  float* b2_l = b1_l + HID16;
  float* gam_l = b2_l + C16;
  float* bet_l = gam_l + C16;
  char* stg = reinterpret_cast<char*>(bet_l + C16) + wave * STRIP16;
  for (int i = tid; i < HID16; i += NW16 * 64) b1_l[i] = p.b1[i];
  for (int i = tid; i < C16; i += NW16 * 64) {
    b2_l[i] = p.b2[i];
    gam_l[i] = p.gamma[i];
    bet_l[i] = p.beta[i];
  }
  __syncthreads();

  // ------------------------------------------------------------------------------------------- weight stream
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.wpk), 0, (unsigned)(NHALF16 * HALF16), 0x00020000);
  // halves of the stream: half H of a row tile = [W2 set H / 2 - 1] (H even) or [W1 set H / 2] (H odd); it lives in ring slot (global
  // half count) % 5.  Iteration i reads halves 2 i and 2 i + 1 and, behind its barrier, refills the two slots iteration i - 1 read
  // with halves 2 i + 3 (needed next iteration: one iteration of lead) and 2 i + 4 (two).
  int src_half = 0, dst_slot = 0;
  auto issue_piece = [&](int j) {   // piece j of this wave's three of the half being issued
    const int pc = wave + NW16 * j;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, lds + dst_slot * HALF16 + pc * 1024), 16,
                                             src_half * HALF16 + pc * 1024 + lane * 16, 0, 0, 0);
  };
  auto issue_done = [&]() {
    src_half = src_half + 1 == NHALF16 ? 0 : src_half + 1;
    dst_slot = dst_slot + 1 == NSLOT16 ? 0 : dst_slot + 1;
  };
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int j = 0; j < PH16; ++j) issue_piece(j);
    issue_done();
  }
  int rd_slot = NSLOT16 - 2, issued = 0;   // slot of the W2 half of the current iteration (advanced by two before use)
  // a lane's x row comes through a buffer descriptor (base in SGPRs, one lane offset, immediates): no 64-bit address registers
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (unsigned)(((int64_t)(p.M - 1) * p.ldx + C16) * 2), 0x00020000);
#ifdef DCC_MLP_PROFILE
  unsigned long long mprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

  // strip: the whole 16-row tile; a lane WRITES row n, READS rows lane >> 3 and 8 + (lane >> 3) at byte 16 (lane & 7)
  // the WRITE address (row n, + 8 g / 16 g) is recomputed from the lane id where it is needed (five vector instructions): kept in a
  // register across the main loop it is the one value hipcc spills there, and every scratch reload beside LDS-DMA and stores in
  // flight is a vmcnt(0), i.e. a wait for ALL of them (measured: the stores' cost doubled)
  auto strip_w = [&](int per_g) -> char* {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return stg + (l & 15) * MSTG_PITCH + (l >> 4) * per_g;
  };
  const char* stg_r = stg + (lane >> 3) * MSTG_PITCH + 16 * (lane & 7);

  // The NEXT row tile's x (operand layout, and once more in the accumulator layout for the residual) is loaded during the current
  // tile's last iteration and consumed BEFORE its y stores go out: a wait on a load issued behind a burst of stores sits out the
  // stores' whole round trip (vmcnt retires in order) — the first version's head cost 36 000-50 000 cycles per 16-row tile that way.
  u32x4 xf[KS16];      // raw x of the tile to come, then LayerNorm(x) of the current one: row n, channels 32 s + 8 g .. + 7
  u32x2 dres[MT16];    // raw x, channels 16 mt + 4 g .. + 3 (the accumulators' layout)
  auto load_tile = [&](int tt) {
    int rw = 16 * tt + n;
    if (rw >= p.M) rw = p.M - 1;
    const int xo = rw * (int)p.ldx * 2;   // byte offset of this lane's row (M * ldx * 2 < 2^31: checked by the launcher)
#pragma unroll
    for (int s = 0; s < KS16; ++s) xf[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, xo + 16 * g, 64 * s, 0));
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt) dres[mt] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsX, xo + 8 * g, 32 * mt, 0));
  };
  if ((int)(blockIdx.x * NW16) < p.tiles) load_tile(blockIdx.x * NW16 + wave);

  for (int t0 = blockIdx.x * NW16; t0 < p.tiles; t0 += gridDim.x * NW16) {
    const int t = t0 + wave;
    const bool tile_live = t < p.tiles;                     // wave-uniform
    const bool exact = tile_live && 16 * t + 16 <= p.M;     // every store of this wave executes: the counted waits hold
    const int row = 16 * t + n;
    const bool live = row < p.M;
    const int64_t lrow = live ? row : p.M - 1;
    const int srow0 = 16 * t + (lane >> 3);

    // 64 columns (one 128-byte line per row) of the 16-row tile leave through the strip; `put(pass)` writes this lane's pieces of
    // row n into its strip row (only the lanes of rows 8 pass .. 8 pass + 7 execute it)
    auto store_lines = [&](bf16_t* base, int64_t ld, int col0, auto&& put, bool nt) {
      const int scol = col0 + 8 * (lane & 7);
      put();
      asm volatile("" ::: "memory");   // cross-lane exchange through LDS: no per-thread reasoning about these bytes (mlp_fused.hip)
      u32x4 o[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) o[q] = *reinterpret_cast<const u32x4*>(stg_r + 8 * q * MSTG_PITCH);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int srow = srow0 + 8 * q;
        if (srow < p.M) {
          u32x4* dst = reinterpret_cast<u32x4*>(base + (int64_t)srow * ld + scol);
          if (nt) __builtin_nontemporal_store(o[q], dst);
          else *dst = o[q];
        }
      }
    };

    // ----------------------------------------------------------------------------------- x, LayerNorm, Y init
    const unsigned long long th0 = MP16_T();
    // output accumulators: register r of tile mt <-> channel 16 mt + 4 g + r; initial value = b2 (+ the residual x)
    f32x4_t Y[MT16];
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(b2_l + 16 * mt + 4 * g);
      const float rs_ = p.residual ? 1.0f : 0.0f;
      Y[mt][0] = b4[0] + rs_ * lo_f(dres[mt][0]);
      Y[mt][1] = b4[1] + rs_ * hi_f(dres[mt][0]);
      Y[mt][2] = b4[2] + rs_ * lo_f(dres[mt][1]);
      Y[mt][3] = b4[3] + rs_ * hi_f(dres[mt][1]);
    }
    {
      float s1 = 0.0f;
#pragma unroll
      for (int s = 0; s < KS16; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) s1 += lo_f(xf[s][q]) + hi_f(xf[s][q]);
      s1 += __shfl_xor(s1, 16);
      s1 += __shfl_xor(s1, 32);
      const float mean = s1 * (1.0f / C16);
      float s2 = 0.0f;
#pragma unroll
      for (int s = 0; s < KS16; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a = lo_f(xf[s][q]) - mean, b = hi_f(xf[s][q]) - mean;
          s2 += a * a + b * b;
        }
      s2 += __shfl_xor(s2, 16);
      s2 += __shfl_xor(s2, 32);
      const float rstd = __builtin_amdgcn_rsqf(s2 * (1.0f / C16) + p.eps);
      if (p.mean != nullptr && live && g == 0) {
        p.mean[row] = mean;
        p.rstd[row] = rstd;
      }
#pragma unroll
      for (int s = 0; s < KS16; ++s) {
        const float* gp = gam_l + 32 * s + 8 * g;
        const float* bp = bet_l + 32 * s + 8 * g;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(bp), c1 = *reinterpret_cast<const f32x4*>(bp + 4);
        u32x4 o;
        o[0] = mpack2((lo_f(xf[s][0]) - mean) * rstd * g0[0] + c0[0], (hi_f(xf[s][0]) - mean) * rstd * g0[1] + c0[1]);
        o[1] = mpack2((lo_f(xf[s][1]) - mean) * rstd * g0[2] + c0[2], (hi_f(xf[s][1]) - mean) * rstd * g0[3] + c0[3]);
        o[2] = mpack2((lo_f(xf[s][2]) - mean) * rstd * g1[0] + c1[0], (hi_f(xf[s][2]) - mean) * rstd * g1[1] + c1[1]);
        o[3] = mpack2((lo_f(xf[s][3]) - mean) * rstd * g1[2] + c1[2], (hi_f(xf[s][3]) - mean) * rstd * g1[3] + c1[3]);
        xf[s] = o;
      }
    }
    if (p.xn != nullptr && tile_live) {   // pieces s = 2 a, 2 a + 1 are 64 channels = one line per row
#pragma unroll
      for (int a = 0; a < KS16 / 2; ++a)
        store_lines(p.xn, C16, 64 * a, [&]() {
          char* w = strip_w(16);
          *reinterpret_cast<u32x4*>(w) = xf[2 * a];
          *reinterpret_cast<u32x4*>(w + 64) = xf[2 * a + 1];
        }, true);
    }

    MP16_ADD(6, MP16_T() - th0);
    // --------------------------------------------------------------------------------------------- main loop
    f32x4_t acc0, acc1;             // H^T of the 32 hidden units of a tile: hidden 4 g + r and 16 + 4 g + r
    u32x4 actf[2];                  // [parity of the hidden tile]: gelu(pre) packed = the B fragment of GEMM2
    u32x2 cpre[2][2], cact[2][2];   // [parity][half]: 8-byte store pieces (hidden 16 half + 4 g .. + 3 of the tile)

    // The W1 half this iteration reads was issued FIRST in the previous iteration; younger than its pieces are the three pieces of
    // the half issued behind it and — for the early waves, whose epilogue (an odd tile's four line stores) sits between that issue and
    // this wait — those stores.  Universal safe count: PH16 (the younger half's pieces always exist).
    auto advance = [&](bool stores_younger) {
      const unsigned long long ta0 = MP16_T();
      if (issued >= 1 && exact && stores_younger) {
        if (p.act != nullptr) mwait_vm<PH16 + 4>();
        else if (p.pre != nullptr) mwait_vm<PH16 + 2>();
        else mwait_vm<PH16>();
      } else {
        mwait_vm<PH16>();
      }
      const unsigned long long ta1 = MP16_T();
      __builtin_amdgcn_s_barrier();
      MP16_ADD(0, ta1 - ta0);
      MP16_ADD(1, MP16_T() - ta1);
      MP16_ADD(5, 1);
      rd_slot = rd_slot + 2 >= NSLOT16 ? rd_slot + 2 - NSLOT16 : rd_slot + 2;
      ++issued;
    };
    // GEMM1 of hidden tile i (W1 set in the second half of the sub-slot) + this wave's share of the next stage's DMA
    auto gemm1 = [&](int i, const char* ss) {
      const f32x4 ba = *reinterpret_cast<const f32x4*>(b1_l + 32 * i + 4 * g), bb = *reinterpret_cast<const f32x4*>(b1_l + 32 * i + 16 + 4 * g);
      acc0 = f32x4_t{ba[0], ba[1], ba[2], ba[3]};
      acc1 = f32x4_t{bb[0], bb[1], bb[2], bb[3]};
      // fragments through plain loads (VGPRs): forcing them into AGPRs as mlp_fused.hip does makes hipcc split the 256 registers of a
      // two-waves-per-SIMD kernel 128 / 128 and shuttle the output tiles between the halves; with a partner wave on the SIMD the
      // write-after-read stall behind an MFMA is covered anyway.  Source order = issue order (the LDS-DMA pieces pin it).
      const char* w1p = ss;
      constexpr int NM = 2 * KS16, AHEAD = 3;
      bf16x8 wf[AHEAD + 1];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < AHEAD; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(w1p + m * 1024);
#pragma unroll
      for (int m = 0; m < NM; ++m) {   // fragment m = (k-step m >> 1, hidden half m & 1)
        if (m + AHEAD < NM) wf[(m + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(w1p + (m + AHEAD) * 1024);
        if ((m % 4) == 3) {                        // 6 pieces over 24 MFMAs: the half needed next iteration first, then the one after
          issue_piece((m / 4) % PH16);
          if ((m / 4) % PH16 == PH16 - 1) issue_done();
        }
        if (m & 1) acc1 = mma32(wf[m % (AHEAD + 1)], __builtin_bit_cast(bf16x8, xf[m >> 1]), acc1);
        else acc0 = mma32(wf[m % (AHEAD + 1)], __builtin_bit_cast(bf16x8, xf[m >> 1]), acc0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + AHEAD, 0);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (m + AHEAD < NM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if ((m % 4) == 3) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // GEMM2 of hidden tile i - 1 (W2 set in the first half of the sub-slot; B fragment = actf[PAR of tile i - 1])
    auto gemm2 = [&](const char* ss, auto Pc) {
      constexpr int PAR = decltype(Pc)::value;
      constexpr int AHEAD = 3;
      bf16x8 wf[AHEAD + 1];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < AHEAD; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(ss + m * 1024);
#pragma unroll
      for (int m = 0; m < MT16; ++m) {
        if (m + AHEAD < MT16) wf[(m + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(ss + (m + AHEAD) * 1024);
        Y[m] = mma32(wf[m % (AHEAD + 1)], __builtin_bit_cast(bf16x8, actf[PAR]), Y[m]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
      for (int m = 0; m < MT16; ++m) {
        if (m + AHEAD < MT16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // epilogue of hidden tile ti (parity PAR): round once, GELU on the rounded value, pack; an odd tile sends the pair (ti - 1, ti) out
    auto epilogue = [&](int ti, auto Pc) {
      constexpr int PAR = decltype(Pc)::value;
      uint32_t pk[4], ak[4];
      pk[0] = mpack2(acc0[0], acc0[1]); pk[1] = mpack2(acc0[2], acc0[3]);
      pk[2] = mpack2(acc1[0], acc1[1]); pk[3] = mpack2(acc1[2], acc1[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) ak[q] = mpack2(gelu_for<bf16_t>(lo_f(pk[q])), gelu_for<bf16_t>(hi_f(pk[q])));
      actf[PAR] = u32x4{ak[0], ak[1], ak[2], ak[3]};
      cpre[PAR][0] = u32x2{pk[0], pk[1]}; cpre[PAR][1] = u32x2{pk[2], pk[3]};
      cact[PAR][0] = u32x2{ak[0], ak[1]}; cact[PAR][1] = u32x2{ak[2], ak[3]};
      if constexpr (PAR == 1) {
        if (tile_live) {
          if (p.pre != nullptr)
            store_lines(p.pre, HID16, 32 * (ti - 1), [&]() {
              char* w = strip_w(8);
#pragma unroll
              for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) *reinterpret_cast<u32x2*>(w + 64 * e + 32 * hf) = cpre[e][hf];
            }, true);
          if (p.act != nullptr)
            store_lines(p.act, HID16, 32 * (ti - 1), [&]() {
              char* w = strip_w(8);
#pragma unroll
              for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) *reinterpret_cast<u32x2*>(w + 64 * e + 32 * hf) = cact[e][hf];
            }, true);
        }
      }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    // iteration i: sub-slot i = [W2 set i-1 | W1 set i]: epilogue(i-1), GEMM1(i), GEMM2(i-1).  The SIMD partners (waves w, w + 4) sit
    // on opposite sides of the iteration's barrier with their epilogue — waves < 4 run it BEFORE the barrier, waves >= 4 behind it —
    // so that one wave's vector work (GELU, packing, stores) runs beside the other's MFMAs instead of both queueing for the same
    // pipe (MI355X_MICROARCH "two waves that run the same program": stagger by wave number >= half).
    const bool late = wave >= NW16 / 2;
    auto body = [&](int i, auto Ec, bool g1, bool g2, auto&& after_issue) {   // ONE copy of the matrix phases; only the (small) epilogue has two call sites
      constexpr int E = decltype(Ec)::value;
      const unsigned long long tb0 = MP16_T();
      if (!late && g2) epilogue(i - 1, std::integral_constant<int, E ^ 1>{});   // early waves: before the barrier, beside the late waves' GEMM2
      MP16_ADD(2, MP16_T() - tb0);
      advance(!late && E == 0);   // early waves: the epilogue of the ODD tile i - 1 (its stores) ran just before this wait
      const char* w2s = lds + rd_slot * HALF16 + lane * 16;
      const char* w1s = lds + (rd_slot + 1 == NSLOT16 ? 0 : rd_slot + 1) * HALF16 + lane * 16;
      const unsigned long long tb1 = MP16_T();
      if (late && g2) epilogue(i - 1, std::integral_constant<int, E ^ 1>{});    // late waves: behind it, beside the early waves' GEMM1
      const unsigned long long tb2 = MP16_T();
      MP16_ADD(2, tb2 - tb1);
      if (g1) gemm1(i, w1s);
      else {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
          for (int j = 0; j < PH16; ++j) issue_piece(j);
          issue_done();
        }
      }
      after_issue();
      const unsigned long long tb3 = MP16_T();
      MP16_ADD(3, tb3 - tb2);
      if (g2) gemm2(w2s, std::integral_constant<int, E ^ 1>{});
      MP16_ADD(4, MP16_T() - tb3);
    };
    auto nohook = []() {};
    body(0, T0{}, true, false, nohook);
    for (int i = 1; i + 1 < NC16; i += 2) {
      body(i, T1{}, true, true, nohook);
      body(i + 1, T0{}, true, true, nohook);
    }
    body(NC16 - 1, T1{}, true, true, nohook);
    const bool has_next = t0 + (int)gridDim.x * NW16 < p.tiles;   // workgroup-uniform
    body(NC16, T0{}, false, true, [&]() { if (has_next) load_tile(t + (int)gridDim.x * NW16); });
    if (has_next) {   // the loads are complete HERE, in front of the y stores
#pragma unroll
      for (int s = 0; s < KS16; ++s) asm volatile("" ::"v"(xf[s]));
#pragma unroll
      for (int mt = 0; mt < MT16; ++mt) asm volatile("" ::"v"(dres[mt]));
    }

    // ------------------------------------------------------------------------------------------------ y
    const unsigned long long ty0 = MP16_T();
    if (tile_live) {
#pragma unroll
      for (int a = 0; a < MT16 / 4; ++a)   // four 16-channel tiles = 64 channels = one line per row
        store_lines(p.y, p.ldy, 64 * a, [&]() {
          char* w = strip_w(8);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            *reinterpret_cast<u32x2*>(w + 32 * e) = u32x2{mpack2(Y[4 * a + e][0], Y[4 * a + e][1]), mpack2(Y[4 * a + e][2], Y[4 * a + e][3])};
        }, false);
    }
    MP16_ADD(7, MP16_T() - ty0);
  }
  mwait_vm<0>();   // LDS-DMA pieces still in flight must land before the workgroup's LDS is handed on
#ifdef DCC_MLP_PROFILE
  if (lane == 0 && (wave == 0 || wave == 4)) {
    for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_mprof16[16 * (wave >> 2) + i_], mprof[i_]);
  }
#endif
}

// w1 [4C, C], w2 [C, 4C] (row-major bf16) -> the stream image: NSTT16 sub-slots of 48 KB, sub-slot i = [W2 set i-1 | W1 set i],
// every set = 24 fragments of 1 KB in lane order (lane l = (m = l & 15, g = l >> 4), 8 bf16 per lane):
//   W1 set c, fragment f = 2 s + hh (k-step s, hidden half hh):  element j = w1[32 c + 16 hh + m][32 s + 8 g + j]
//   W2 set c, fragment mt:   element j = w2[16 mt + m][32 c + (j < 4 ? 4 g + j : 16 + 4 g + j - 4)]   (accumulator-as-operand k order)
__global__ void mlp16_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, bf16_t* __restrict__ out, int64_t total8) {
  constexpr int half8 = HALF16 / 16;   // 16-byte units per set
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total8; u += (int64_t)gridDim.x * blockDim.x) {
    const int slot = (int)(u / (2 * half8));
    const int in = (int)(u - (int64_t)slot * 2 * half8);
    const bool is_w1 = in >= half8;
    const int v = is_w1 ? in - half8 : in;
    const int frag = v >> 6, lane = v & 63, m = lane & 15, g = lane >> 4;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.0f;
    if (is_w1) {
      const int c = slot;
      if (c < NC16) {
        const int s = frag >> 1, hh = frag & 1;
        const bf16_t* src = w1 + (int64_t)(32 * c + 16 * hh + m) * C16 + 32 * s + 8 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = src[j];
      }
    } else {
      const int c = slot - 1;
      if (c >= 0 && c < NC16) {
        const bf16_t* src = w2 + (int64_t)(16 * frag + m) * HID16 + 32 * c + 4 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = src[(j < 4) ? j : 16 + (j - 4)];
      }
    }
    *reinterpret_cast<bf16x8*>(out + 8 * u) = o;
  }
}

}  // namespace

#ifdef DCC_MLP_PROFILE
extern "C" int dcc_mlp16_profile(unsigned long long* out, int reset) {  // debug builds only (-DDCC_MLP_PROFILE)
  if (reset) { unsigned long long z[32] = {0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_mprof16), z, sizeof(z)) == hipSuccess ? 0 : -1; }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mprof16), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

int mlp16_packed_bytes() { return NHALF16 * HALF16; }

int mlp16_pack(const void* w1, const void* w2, void* packed, hipStream_t st) {
  const int64_t total8 = (int64_t)mlp16_packed_bytes() / 16;
  int64_t blocks = (total8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mlp16_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)w1, (const bf16_t*)w2, (bf16_t*)packed, total8);
  return dcc_check_launch();
}

int mlp16_launch(const MlpParams& p16, hipStream_t st) {
  MlpParams p = p16;
  p.tiles = (p.M + 15) / 16;
  static unsigned long long configured = 0;
  const unsigned long long devbit = dcc_dev_bit();
  if (!(configured & devbit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ln_mlp16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS16) != hipSuccess) return DCC_ERR_LAUNCH;
    configured |= devbit;
  }
  int gx = 256;
  const int need = (p.tiles + NW16 - 1) / NW16;
  if (gx > need) gx = need;
  hipLaunchKernelGGL(ln_mlp16_kernel, dim3(gx), dim3(NW16 * 64), LDS16, st, p);
  return dcc_check_launch();
}
