// 256 x 256 "eight-phase" bf16 GEMM for gfx950 (EXPERIMENTAL: reached only through tuning variant 13, see gemm.hip).
//   C[m][n] = alpha * sum_k A(m,k) * B(n,k)  (+ bias[n]) (gelu / * gelu'(aux) / + R[m][n]),  A [M][K] and B [N][K] k-contiguous bf16,
//   C bf16 — the forward products of the towers' Linear layers (models/video_encoder.py:232-253 via torchvision MultiscaleBlock;
//   same contract as gemm.hip).
//
// Why: the ring kernels spend 600-640 ns per 64-deep step of a 128 x 128 tile against 256 ns of matrix-core time, the wide kernel
// 460-575 ns per equivalent (profiles/r03_gemm_ksweep.txt), and run back to back they sit at the board's power limit
// (profiles/r03_clock_power.txt).  This is the schedule cdna_hip_programming.md section 5 describes for its 256^2 template,
// re-derived for 32 x 32 x 16 MFMAs and this library's C^T epilogue:
//   * 8 waves = 2 wave rows x 4 wave columns, 128 x 64 of the tile per wave (8 accumulator tiles = 128 registers);
//   * a 64-deep K-tile is four 16-KB LDS "pieces" of 128-byte rows: S0 = the A rows phase 1 reads (rows 0..63 of each wave row's
//     128), S1 = the B columns phase 1 reads (columns 0..31 of each wave column's 64), S2 / S3 = the other halves; two K-tile
//     buffers (128 KB).  LDS-DMA writes a piece linearly (1 KB per wave-instruction); the 16-byte chunk a lane FETCHES is
//     XOR-swizzled by its row so that the 32 x 16 fragment reads are conflict-free (the ring kernel's swizzle);
//   * a K-tile is four phases of one accumulator quadrant each (2 x 1 tiles x 4 k-substeps = 8 MFMAs); ALL fragment reads happen
//     in phases 1 (S0, S1) and 2 (S2, S3) — 96 registers — so a piece is dead two barriers after it landed, and phase g can
//     restage the slot of piece g - 1 with piece g + 7: three pieces are in flight across every barrier, waited for once per
//     K-tile with a counted vmcnt(6);
//   * every phase is  [fragment reads, 2 LDS-DMAs, (waits)] s_barrier [8 MFMAs] s_barrier, and the two wave rows run ONE barrier
//     apart, so that on every SIMD one wave issues MFMAs while its partner reads / stages;
//   * persistent workgroups, XCD-aware item order (as the ring); the piece pipeline runs across tiles, so only the epilogue —
//     straight from registers, 16-byte stores after a v_permlane32_swap, both wave rows at the same time — is exposed.
// Hazards (who orders what):
//   RAW  a piece issued at phase g - 7 is read at phase g (S0, S1) or g + 1: every wave's vmcnt(6) in phase 4 of the previous K-tile
//        (the three pieces issued after the awaited ones = 6 loads; loads return in order, stores in between only make the wait
//        stricter) precedes a barrier that every reader passes before its phase-1 reads;
//   WAR  the slot of S0 is restaged ONE phase after its reads: the readers wait lgkmcnt(0) BEFORE their phase's first barrier and
//        the staging waves pass that barrier first; every other slot has two or three phases of slack.
#include "gemm_params.h"
#include <stdlib.h>

namespace {

constexpr int PNT = 512;
constexpr int PIECE_B = 128 * 128;          // 128 rows x 64 k bf16
constexpr int KT_B = 4 * PIECE_B;           // one K-tile: S0 S1 S2 S3
constexpr int PBIAS_OFF = 2 * KT_B;         // two 1-KB bias slots (item parity)
constexpr int PSTG_OFF = PBIAS_OFF + 2 * 1024;   // 8 wave-private store-staging strips: 8 rows x (128 + 16) bytes
constexpr int PSTG_PITCH = 144, PSTG_WAVE = 8 * PSTG_PITCH;
constexpr int P8_LDS = PSTG_OFF + 8 * PSTG_WAVE;

struct P8Geo { int items, tilesN, nk, dbg, stagger; };   // stagger: 2048-cycle sleeps before the first tile of a workgroup that has one item fewer   // dbg (DCC_P8_DEBUG, experiments): 1 = no stores, 2 = no epilogue

template <int N> __device__ __forceinline__ void pwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pwait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// 16 k (substep s) of the 32 rows starting at piece-local row t0: 128-byte rows, chunk c of row r stored at c ^ ((r >> 1) & 7)
__device__ __forceinline__ bf16x8 pfrag(const char* piece, int t0, int s, int lane) {
  const int row = t0 + (lane & 31), chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(piece + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
__device__ __forceinline__ uint32_t ppack2(float a, float b) {
  bf16x2 t;
  t[0] = (bf16_t)a;
  t[1] = (bf16_t)b;
  return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ void punpack8(const u32x4& c, float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = __builtin_bit_cast(float, c[q] << 16);
    v[2 * q + 1] = __builtin_bit_cast(float, c[q] & 0xffff0000u);
  }
}

template <int EP>
__global__ __launch_bounds__(PNT, 1) void gemm_p8_kernel(GemmParams p, P8Geo geo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3, h = lane >> 5;
  const int G = gridDim.x, bid = blockIdx.x;
  const int item0 = (bid & 7) * (G >> 3) + (bid >> 3);   // G is a multiple of 8: the 8 XCDs walk neighbouring tiles
  int nitems = 0;
  for (int i = item0; i < geo.items; i += G) ++nitems;
  if (nitems == 0) return;
  // Tile boundaries of the persistent workgroups coincide (same K, same start): every round ends in a chip-wide store burst that
  // the next K-tile's counted wait has to sit out.  The workgroups with one item fewer have a tile's worth of slack: they start late.
  if (geo.stagger > 0 && (geo.items % G) && item0 >= geo.items % G) {   // floor(items / G) items here, ceil elsewhere
    const int n = geo.stagger / 2 + ((bid >> 3) & 1) * (geo.stagger / 2);   // two start times among them
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
  }
  const int nk = geo.nk;
  const int TT = nitems * nk;   // K-tiles of this workgroup
  const int NP = 4 * TT;        // pieces

  // ---- staging side: this lane's share of a piece is 2 x 16 bytes (rounds j = 0, 1: piece-local rows 64 j + tid / 8)
  const int r8 = tid >> 3;
  const int gch = (tid & 7) ^ ((r8 >> 1) & 7);                       // the chunk of its row this lane fetches
  const int voffA = r8 * (int)p.lda * 2 + gch * 16;                  // + (128 j + 64 sh) rows, + 128 kt bytes
  const int voffB = (64 * (r8 >> 5) + (r8 & 31)) * (int)p.ldb * 2 + gch * 16;   // + (128 j + 32 sh) columns
  const int rstepA = 64 * (int)p.lda * 2, rstepB = 32 * (int)p.ldb * 2;
  int iq = 0, is_item = item0, is_kt = 0, is_par = 0;
  __amdgpu_buffer_rsrc_t rsA, rsB, rsBias;
  auto bind = [&](int idx) {
    const int tm = idx / geo.tilesN, tn = idx - tm * geo.tilesN;
    const int m0 = tm * 256, n0 = tn * 256;
    const int remM = p.M - m0, remN = p.N - n0;   // > 0
    const char* A = reinterpret_cast<const char*>(p.A) + (int64_t)m0 * p.lda * 2;
    const char* B = reinterpret_cast<const char*>(p.B) + (int64_t)n0 * p.ldb * 2;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A), 0, (unsigned)(((int64_t)(remM - 1) * p.lda + p.K) * 2), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B), 0, (unsigned)(((int64_t)(remN - 1) * p.ldb + p.K) * 2), 0x00020000);
    const bool hb = (p.flags & GEMM_BIAS) != 0;
    rsBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hb ? p.bias + n0 : reinterpret_cast<const float*>(p.A)), 0,
                                               hb ? (unsigned)(remN * 4) : 0u, 0x00020000);
  };
  bind(item0);
  auto issue_piece = [&]() {
    if (iq < NP) {
      const int s = iq & 3, sh = s >> 1;
      char* dst = smem + ((iq >> 2) & 1) * KT_B + s * PIECE_B + wid * 1024;
      const int kb = is_kt * 128;
      if (!(s & 1)) {
        if (s == 0 && is_kt == 0)   // the tile's bias row: 4 x 256 B (wave columns); the second wave row repeats the first's writes
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LDS_PTR(void, smem + PBIAS_OFF + is_par * 1024 + wc * 256), 4,
                                                   (wc * 64 + lane) * 4, 0, 0, 0);
        const int vo = voffA + sh * rstepA + kb;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, dst), 16, vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, dst + 8192), 16, vo + 2 * rstepA, 0, 0, 0);
      } else {
        const int vo = voffB + sh * rstepB + kb;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, dst), 16, vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, dst + 8192), 16, vo + 4 * rstepB, 0, 0, 0);
        if (s == 3 && ++is_kt == nk) {
          is_kt = 0;
          is_par ^= 1;
          is_item += G;
          if (is_item < geo.items) bind(is_item);
        }
      }
    }
    ++iq;
  };

  // ---- consumer side
  f32x16 acc[4][2];
  bf16x8 falo[2][4], fahi[2][4], fblo[4], fbhi[4];
  int c_item = item0, kt = 0, c_par = 0;
  const bf16_t* R = reinterpret_cast<const bf16_t*>(p.R);
  bf16_t* AUX = reinterpret_cast<bf16_t*>(p.aux);
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);

  // EP >= 0: the epilogue flags are compile-time (the instances the towers use); EP < 0: read from p.flags
  auto epilogue = [&]() {
    const int F = EP >= 0 ? EP : p.flags;
    const int tm = c_item / geo.tilesN, tn = c_item - tm * geo.tilesN;
    const int m0 = tm * 256 + wr * 128, n0 = tn * 256 + wc * 64;
    // bias of this lane's 32 columns (register 4 g + e of n-tile j <-> column 32 j + 8 g + 4 h + e), once per tile
    f32x4 b4[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        b4[j][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (F & GEMM_BIAS)
          b4[j][g] = *reinterpret_cast<const f32x4*>(smem + PBIAS_OFF + c_par * 1024 + (wc * 64 + 32 * j + 8 * g + 4 * h) * 4);
      }
    const bool full = (m0 + 128 <= p.M) && (n0 + 64 <= p.N);   // wave-uniform: no per-piece bounds inside
    const float alpha = p.alpha;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 32 * i + (lane & 31);
      const bool live = row < p.M;
      const int64_t lrow = live ? row : p.M - 1;
      bf16_t* crow = C + lrow * p.ldc + n0 + 8 * h;
      bf16_t* arow = (F & (GEMM_SAVE_PRE | GEMM_DGELU)) ? AUX + lrow * p.ldaux + n0 + 8 * h : nullptr;
      const bf16_t* rrow = (F & GEMM_RESID) ? R + lrow * p.ldr + n0 + 8 * h : nullptr;
      u32x4 cfin[2][2], cpre[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint32_t pk[8];   // round once after alpha * acc + bias
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          pk[2 * g] = ppack2(acc[i][j][4 * g] * alpha + b4[j][g][0], acc[i][j][4 * g + 1] * alpha + b4[j][g][1]);
          pk[2 * g + 1] = ppack2(acc[i][j][4 * g + 2] * alpha + b4[j][g][2], acc[i][j][4 * g + 3] * alpha + b4[j][g][3]);
        }
#pragma unroll
        for (int P = 0; P < 2; ++P) {
          // column groups g = 2P (this lane: 16P + 4h ..) and 2P + 1 (16P + 8 + 4h ..): after the half exchange the lower lane holds
          // columns 16P .. 16P+7 of its row, the upper lane 16P+8 .. 16P+15
          u32x4 c;
          {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * P], pk[4 * P + 2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * P + 1], pk[4 * P + 3], false, false);
            c[0] = s0[0]; c[1] = s1[0]; c[2] = s0[1]; c[3] = s1[1];
          }
          const int co = 32 * j + 16 * P;                               // element offset from the row pointers (+ 8 h inside them)
          const bool ok = full || (live && n0 + co + 8 * h < p.N);      // N % 8 == 0: a 16-byte piece is inside or outside
          const int lco = ok ? co : -(n0 + 8 * h);                      // dead pieces read column 0 of a valid row
          cpre[j][P] = c;
          if (F & (GEMM_GELU | GEMM_DGELU | GEMM_RESID)) {
            float v[8];
            punpack8(c, v);
            if (F & GEMM_GELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = gelu_for<bf16_t>(v[e]);
            }
            if (F & GEMM_DGELU) {
              const u32x4 a = *reinterpret_cast<const u32x4*>(arow + lco);
              float av[8];
              punpack8(a, av);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= dgelu_for<bf16_t>(av[e]);
            }
            if (F & GEMM_RESID) {
              const u32x4 a = *reinterpret_cast<const u32x4*>(rrow + lco);
              float av[8];
              punpack8(a, av);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += av[e];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = ppack2(v[2 * q], v[2 * q + 1]);
          }
          cfin[j][P] = c;
        }
      }
      if (n0 + 64 <= p.N && !(geo.dbg & 4)) {
        // complete 128-byte lines through the wave's LDS strip (as gemm_stream.hip): pass q = rows 8q .. 8q+7 of this 32-row slab; the
        // 16 lanes that own them write their 4 x 16 bytes, every lane reads 16 bytes back and stores row 8q + lane / 8, bytes 16 (lane & 7)
        char* stg = smem + PSTG_OFF + wid * PSTG_WAVE;
        char* stg_w = stg + (lane & 7) * PSTG_PITCH + 16 * h;
        const char* stg_r = stg + (lane >> 3) * PSTG_PITCH + 16 * (lane & 7);
        const int r = lane & 31;
#pragma unroll
        for (int pass = 0; pass < ((F & GEMM_SAVE_PRE) ? 2 : 1); ++pass) {
          const bool pre = pass == 0 && (F & GEMM_SAVE_PRE);
          bf16_t* base = pre ? AUX : C;
          const int64_t ld = pre ? p.ldaux : p.ldc;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if ((r >> 3) == q) {
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int P = 0; P < 2; ++P) *reinterpret_cast<u32x4*>(stg_w + 64 * j + 32 * P) = pre ? cpre[j][P] : cfin[j][P];
            }
            const u32x4 o = *reinterpret_cast<const u32x4*>(stg_r);
            const int srow = m0 + 32 * i + 8 * q + (lane >> 3);
            if (geo.dbg & 1) asm volatile("" ::"v"(o));
            else if (srow < p.M) *reinterpret_cast<u32x4*>(base + (int64_t)srow * ld + n0 + 8 * (lane & 7)) = o;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int P = 0; P < 2; ++P) {
            const int co = 32 * j + 16 * P;
            const bool ok = full || (live && n0 + co + 8 * h < p.N);
            if ((F & GEMM_SAVE_PRE) && ok) *reinterpret_cast<u32x4*>(arow + co) = cpre[j][P];
            if (geo.dbg & 1) asm volatile("" ::"v"(cfin[j][P]));
            else if (ok) *reinterpret_cast<u32x4*>(crow + co) = cfin[j][P];
          }
      }
    }
    c_item += G;
    c_par ^= 1;
  };

  // ---- prologue: pieces 0..6, the first K-tile (0..3) landed
#pragma unroll 1
  for (int q = 0; q < 7; ++q) issue_piece();
  if (NP > 6) pwait_vm<6>(); else pwait_vm<0>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave row runs one barrier behind

#pragma unroll 1
  for (int T = 0; T < TT; ++T) {
    const char* buf = smem + (T & 1) * KT_B;
    if (kt == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }
    // ---------------------------------------------------------------- phase 1: S0, S1 -> registers; quadrant (A lo, B lo)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) falo[i][s] = pfrag(buf, wr * 64 + 32 * i, s, lane);
      fblo[s] = pfrag(buf + PIECE_B, wc * 32, s, lane);
    }
    issue_piece();
    pwait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][0] = mma16(fblo[s], falo[i][s], acc[i][0]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------------------------------------------------------- phase 2: S2, S3 -> registers; quadrant (A hi, B hi)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fahi[i][s] = pfrag(buf + 2 * PIECE_B, wr * 64 + 32 * i, s, lane);
      fbhi[s] = pfrag(buf + 3 * PIECE_B, wc * 32, s, lane);
    }
    issue_piece();
    pwait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 + i][1] = mma16(fbhi[s], fahi[i][s], acc[2 + i][1]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------------------------------------------------------- phase 3: quadrant (A lo, B hi)
    issue_piece();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][1] = mma16(fbhi[s], falo[i][s], acc[i][1]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------------------------------------------------------- phase 4: next K-tile landed; quadrant (A hi, B lo)
    issue_piece();
    if (iq <= NP) pwait_vm<6>(); else pwait_vm<0>();   // three newer pieces exist (6 loads) / the tail: everything
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 + i][0] = mma16(fblo[s], fahi[i][s], acc[2 + i][0]);
    __builtin_amdgcn_s_setprio(0);
    const bool last = (kt + 1 == nk);
    kt = last ? 0 : kt + 1;
    // both wave rows run their epilogue in the same barrier interval: the row that is one barrier behind before this barrier,
    // the other after it
    if (last && wr == 1 && !(geo.dbg & 2)) epilogue();
    __builtin_amdgcn_s_barrier();
    if (last && wr == 0 && !(geo.dbg & 2)) epilogue();
    if (last && (geo.dbg & 2)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
      c_item += G; c_par ^= 1;
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
}

int p8_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}

}  // namespace

// bf16 x bf16 -> bf16, k-contiguous A [M,K] and B [N,K], K % 64 == 0, K >= 128, N % 8 == 0.  DCC_ERR_UNSUPPORTED outside that domain.
int dcc_gemm_p8_launch(const GemmParams& p, hipStream_t st) {
  if (p.batch != 1 || p.splitk != 1 || (p.flags & (GEMM_ACCUM | GEMM_COLSUM_A))) return DCC_ERR_UNSUPPORTED;
  if ((p.K % 64) || p.K < 128 || (p.N % 8) || (p.lda % 8) || (p.ldb % 8) || (p.ldc % 8) || p.M < 1) return DCC_ERR_UNSUPPORTED;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) || ((uintptr_t)p.C & 15)) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & GEMM_RESID) && ((p.ldr % 8) || ((uintptr_t)p.R & 15))) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & (GEMM_SAVE_PRE | GEMM_DGELU)) && ((p.ldaux % 8) || ((uintptr_t)p.aux & 15))) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & GEMM_BIAS) && ((uintptr_t)p.bias & 3)) return DCC_ERR_UNSUPPORTED;
  // 32-bit buffer offsets: 256 rows of a tile + the k offset
  if ((int64_t)256 * p.lda * 2 + (int64_t)p.K * 2 >= (1LL << 31) || (int64_t)256 * p.ldb * 2 + (int64_t)p.K * 2 >= (1LL << 31)) return DCC_ERR_UNSUPPORTED;
  if ((int64_t)p.M * p.lda * 2 > 0xffffffffLL || (int64_t)p.N * p.ldb * 2 > 0xffffffffLL) return DCC_ERR_UNSUPPORTED;   // buffer descriptor sizes
  P8Geo geo;
  const int64_t tm = (p.M + 255) / 256, tn = (p.N + 255) / 256;
  if (tm * tn > 0x7fffffffLL) return DCC_ERR_UNSUPPORTED;
  geo.items = (int)(tm * tn);
  geo.tilesN = (int)tn;
  geo.nk = p.K / 64;
  static const int dbg = getenv("DCC_P8_DEBUG") ? atoi(getenv("DCC_P8_DEBUG")) : 0;
  geo.dbg = dbg;
  // DCC_P8_STAGGER (experiment): start delay of the lighter workgroups in per-mille of a tile's time (~ nk * 3300 + 6000 cycles)
  static const int stag = getenv("DCC_P8_STAGGER") ? atoi(getenv("DCC_P8_STAGGER")) : 0;
  geo.stagger = (int)((int64_t)stag * (geo.nk * 3300 + 6000) / 1000 / 2048);
  const int ef = p.flags & (GEMM_BIAS | GEMM_GELU | GEMM_SAVE_PRE | GEMM_RESID | GEMM_DGELU);
  void (*kern)(GemmParams, P8Geo) = gemm_p8_kernel<-1>;
  if (ef == 0) kern = gemm_p8_kernel<0>;
  else if (ef == GEMM_BIAS) kern = gemm_p8_kernel<GEMM_BIAS>;
  else if (ef == (GEMM_BIAS | GEMM_GELU | GEMM_SAVE_PRE)) kern = gemm_p8_kernel<GEMM_BIAS | GEMM_GELU | GEMM_SAVE_PRE>;
  else if (ef == (GEMM_BIAS | GEMM_RESID)) kern = gemm_p8_kernel<GEMM_BIAS | GEMM_RESID>;
  else if (ef == GEMM_DGELU) kern = gemm_p8_kernel<GEMM_DGELU>;
  static unsigned long long configured = 0;
  const unsigned long long devbit = dcc_dev_bit();
  if (!(configured & devbit)) {
    for (const void* f : {reinterpret_cast<const void*>(gemm_p8_kernel<-1>), reinterpret_cast<const void*>(gemm_p8_kernel<0>),
                          reinterpret_cast<const void*>(gemm_p8_kernel<GEMM_BIAS>),
                          reinterpret_cast<const void*>(gemm_p8_kernel<GEMM_BIAS | GEMM_GELU | GEMM_SAVE_PRE>),
                          reinterpret_cast<const void*>(gemm_p8_kernel<GEMM_BIAS | GEMM_RESID>),
                          reinterpret_cast<const void*>(gemm_p8_kernel<GEMM_DGELU>)})
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS) != hipSuccess) return DCC_ERR_LAUNCH;
    configured |= devbit;
  }
  int G = p8_cus() & ~7;
  if (G < 8) G = 8;
  const int need = (geo.items + 7) & ~7;
  if (G > need) G = need;
  hipLaunchKernelGGL(kern, dim3(G), dim3(PNT), P8_LDS, st, p, geo);
  return dcc_check_launch();
}
