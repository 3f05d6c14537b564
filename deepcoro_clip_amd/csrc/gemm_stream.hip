// Streaming panel GEMM for the EARLY MViT blocks (C = 96 / 192, 25 089 - 6 273 tokens per clip): C[M,N] = A[M,K] W[N,K]^T with a
// short reduction (K <= 576) and M in the hundreds of thousands.  These GEMMs are HBM-bound by a wide margin (blk0 qkv:
// 154 MB in, 462 MB out, 44 GFLOP = 71 FLOP/B against a 312 FLOP/B ridge), and the tiled kernels (gemm_ring / gemm_wide) run
// them at 2.6-3.1 TB/s: with K = 96 a 256 x 128 tile is three k-steps, so every tile is prologue + epilogue, N = 288 quantises to
// three 128-column tiles (25 % wasted) and the A panel is fetched once per column tile.  Here (torchvision MultiscaleBlock's
// attn.qkv / attn.project / mlp / project Linear layers via models/video_encoder.py:232-253):
//   * the whole weight (or a column chunk of it: <= 150 KB) sits in LDS for the life of the workgroup, rows padded by 16 B so a
//     wave's ds_read_b128 of 32 rows is conflict-free (pitch = odd multiple of 16 B);
//   * every wave owns 32-row tiles of A outright: its 32 x K operand goes global -> VGPRs once (lane = row, 16-byte k-pieces),
//     then the wave walks all column tiles: K/16 MFMAs (W fragment from LDS x its rows), epilogue, store.  No barrier after the
//     weight staging, no LDS traffic for A, no cross-wave dependency: 8-16 independent waves per CU keep 100-200 KB in flight;
//   * the product is computed transposed (W tile as the MFMA A operand, rows as B) so a lane holds ONE output row and 4 x 4
//     contiguous columns; a v_permlane32_swap per dword (cdna guide T21) turns them into two 16-byte pieces per lane;
//   * epilogue contract = the ring / wide kernels': round to bf16 once after alpha * acc + bias, then (save pre) / GELU /
//     x gelu'(aux) / + residual on the rounded value, round again.
// Each A row is read once per column chunk (one chunk for every forward shape of blocks 0-2 except fc1 of blocks 1-2 and qkv of
// block 2-3: two to three chunks, the re-read served by L2 / MALL), each output written once.
#include "gemm_params.h"
#include <stdlib.h>

namespace {

constexpr int SNT = 512;  // 8 waves

struct StreamGeo {
  int dbg;     // DCC_STREAM_DEBUG (experiments): 1 = compute everything, store nothing
  int tiles;   // 32-row tiles
  int nc;      // columns per chunk (multiple of 32)
  int pitch;   // bytes per weight row in LDS: 2 K + 16
};

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2 t;
  t[0] = (bf16_t)a;
  t[1] = (bf16_t)b;
  return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ void unpack8(const u32x4& c, float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = __builtin_bit_cast(float, c[q] << 16);
    v[2 * q + 1] = __builtin_bit_cast(float, c[q] & 0xffff0000u);
  }
}

constexpr int STG_PITCH = 144;                 // staging rows: 128 bytes + 16 (8 rows land on 16 different 16-byte bank slots)
constexpr int STG_WAVE = 8 * STG_PITCH;        // per wave: 8 rows x 64 columns
constexpr int STG_B = 8 * STG_WAVE;            // per workgroup

// EP >= 0: epilogue flags known at compile time (the instances the towers use); EP < 0: read from p.flags.
// Stores: a lane owns ONE output row, so a direct 16-byte store per lane makes a wave-instruction touch 32 rows x 32 bytes — 4 096
// partial-line write requests per 32 x 64 outputs, which is what bounds this kernel (DCC_STREAM_DEBUG=1: 1.5-3x faster without its
// stores).  Column tiles therefore go out in PAIRS through a wave-private 8-row LDS strip: four passes of (the 16 lanes that own rows
// 8q..8q+7 write their 4 x 16 bytes, every lane reads 16 bytes back, ONE store of 8 complete 128-byte lines).  No barrier: a wave's
// LDS operations execute in order.
template <int KK, int MINW, int EP>
__global__ __launch_bounds__(SNT, MINW) void gemm_stream_kernel(GemmParams p, StreamGeo geo) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int F = EP >= 0 ? EP : p.flags;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int n0 = blockIdx.y * geo.nc;
  const int ncols = (p.N - n0) < geo.nc ? (p.N - n0) : geo.nc;   // multiple of 32
  constexpr int CH = KK / 8;   // 16-byte pieces per weight row
  {
    const bf16_t* W = reinterpret_cast<const bf16_t*>(p.B) + (int64_t)n0 * p.ldb;
    for (int i = tid; i < ncols * CH; i += SNT) {
      const int n = i / CH, c = i - n * CH;
      const u32x4 v = *reinterpret_cast<const u32x4*>(W + (int64_t)n * p.ldb + c * 8);
      *reinterpret_cast<u32x4*>(lds + n * geo.pitch + c * 16) = v;
    }
  }
  float* bias_l = reinterpret_cast<float*>(lds + geo.nc * geo.pitch);
  for (int i = tid; i < ncols; i += SNT) bias_l[i] = (F & GEMM_BIAS) ? p.bias[n0 + i] : 0.0f;
  char* stg = lds + geo.nc * (geo.pitch + 4) + wave * STG_WAVE;
  __syncthreads();
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
  const bf16_t* R = reinterpret_cast<const bf16_t*>(p.R);
  bf16_t* AUX = reinterpret_cast<bf16_t*>(p.aux);
  const char* wrow = lds + r * geo.pitch + h * 16;   // this lane's row of a 32-column weight tile, k-half h
  const int ntile = ncols >> 5;
  const float alpha = p.alpha;
  // staging addresses: this lane WRITES row r & 7 (in pass r >> 3) at byte 64 j + 32 P + 16 h, READS row lane >> 3 at byte 16 (lane & 7)
  char* stg_w = stg + (r & 7) * STG_PITCH + 16 * h;
  const char* stg_r = stg + (lane >> 3) * STG_PITCH + 16 * (lane & 7);
  for (int t = blockIdx.x * 8 + wave; t < geo.tiles; t += gridDim.x * 8) {
    const int row = 32 * t + r;
    const bool live = row < p.M;
    const int64_t lrow = live ? row : p.M - 1;   // tail lanes read a valid row; nothing of theirs is stored
    bf16x8 xf[KK / 16];
    const bf16_t* xr = A + lrow * p.lda + 8 * h;
#pragma unroll
    for (int s = 0; s < KK / 16; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xr + 16 * s);
    // direct-path pointers (this lane's own row) and staged-path pointers (pass q: row 32 t + 8 q + lane / 8, column 8 (lane & 7))
    bf16_t* crow = C + lrow * p.ldc + n0 + 8 * h;
    bf16_t* arow = (F & (GEMM_SAVE_PRE | GEMM_DGELU)) ? AUX + lrow * p.ldaux + n0 + 8 * h : nullptr;
    const bf16_t* rrow = (F & GEMM_RESID) ? R + lrow * p.ldr + n0 + 8 * h : nullptr;
    const int srow0 = 32 * t + (lane >> 3);
    for (int np = 0; np < ntile; np += 2) {
      const bool two = np + 1 < ntile;   // wave-uniform
      u32x4 cfin[2][2], cpre[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j == 1 && !two) break;
        const int nt = np + j;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        const char* wt = wrow + nt * 32 * geo.pitch;
#pragma unroll
        for (int s = 0; s < KK / 16; ++s) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wt + 32 * s);
          acc = mma16(wf, xf[s], acc);   // D[n][row]: lane = row, registers = 16 output columns
        }
        // register 4g + i <-> column 32 nt + 8 g + 4 h + i; round once after alpha * acc + bias
        uint32_t pk[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_l + 32 * nt + 8 * g + 4 * h);
          pk[2 * g] = pack2(acc[4 * g] * alpha + b4[0], acc[4 * g + 1] * alpha + b4[1]);
          pk[2 * g + 1] = pack2(acc[4 * g + 2] * alpha + b4[2], acc[4 * g + 3] * alpha + b4[3]);
        }
#pragma unroll
        for (int P = 0; P < 2; ++P) {
          // column groups g = 2P (this lane: 4 columns at 16P + 4h) and 2P + 1 (at 16P + 8 + 4h): after the half exchange the
          // lower lane holds columns 16P .. 16P+7 of its row, the upper lane 16P+8 .. 16P+15
          u32x4 c;
          {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * P], pk[4 * P + 2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * P + 1], pk[4 * P + 3], false, false);
            c[0] = s0[0]; c[1] = s1[0]; c[2] = s0[1]; c[3] = s1[1];
          }
          const int co = 32 * nt + 16 * P;   // element offset from the row pointers
          cpre[j][P] = c;
          if (F & (GEMM_GELU | GEMM_DGELU | GEMM_RESID)) {
            float v[8];
            unpack8(c, v);
            if (F & GEMM_GELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = gelu_for<bf16_t>(v[e]);
            }
            if (F & GEMM_DGELU) {
              const u32x4 a = *reinterpret_cast<const u32x4*>(arow + co);
              float av[8];
              unpack8(a, av);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= dgelu_for<bf16_t>(av[e]);
            }
            if (F & GEMM_RESID) {
              const u32x4 a = *reinterpret_cast<const u32x4*>(rrow + co);
              float av[8];
              unpack8(a, av);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += av[e];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = pack2(v[2 * q], v[2 * q + 1]);
          }
          cfin[j][P] = c;
        }
      }
      if (!two || (geo.dbg & 2)) {   // a last single column tile (N / 32 odd), or DCC_STREAM_DEBUG=2 (A/B: the direct path everywhere)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j == 1 && !two) break;
#pragma unroll
          for (int P = 0; P < 2; ++P) {
            const int co = 32 * (np + j) + 16 * P;
            if ((F & GEMM_SAVE_PRE) && live) *reinterpret_cast<u32x4*>(arow + co) = cpre[j][P];
            if (geo.dbg & 1) asm volatile("" ::"v"(cfin[j][P]));
            else if (live) *reinterpret_cast<u32x4*>(crow + co) = cfin[j][P];
          }
        }
      } else {
        const int scol = n0 + 32 * np + 8 * (lane & 7);
#pragma unroll
        for (int pass = 0; pass < ((F & GEMM_SAVE_PRE) ? 2 : 1); ++pass) {
          bf16_t* base = (pass == 0 && (F & GEMM_SAVE_PRE)) ? AUX : C;
          const int64_t ld = (pass == 0 && (F & GEMM_SAVE_PRE)) ? p.ldaux : p.ldc;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if ((r >> 3) == q) {
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int P = 0; P < 2; ++P)
                  *reinterpret_cast<u32x4*>(stg_w + 64 * j + 32 * P) = (pass == 0 && (F & GEMM_SAVE_PRE)) ? cpre[j][P] : cfin[j][P];
            }
            asm volatile("" ::: "memory");   // cross-lane exchange through LDS: no per-thread reasoning about these bytes (csrc/mlp_fused.hip)
            const u32x4 o = *reinterpret_cast<const u32x4*>(stg_r);
            asm volatile("" ::: "memory");
            const int srow = srow0 + 8 * q;
            if (geo.dbg & 1) asm volatile("" ::"v"(o));
            else if (srow < p.M) *reinterpret_cast<u32x4*>(base + (int64_t)srow * ld + scol) = o;
          }
        }
      }
    }
  }
}

template <int KK, int MINW, int EP>
int stream_launch_ep(const GemmParams& p, const StreamGeo& geo, int lds, int chunks, hipStream_t st) {
  auto kern = gemm_stream_kernel<KK, MINW, EP>;
  static unsigned long long configured = 0;
  const unsigned long long devbit = dcc_dev_bit();
  if (!(configured & devbit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return DCC_ERR_LAUNCH;
    configured |= devbit;
  }
  // persistent waves: enough workgroups to fill every CU at the occupancy LDS and registers allow, never more than the work
  int per_cu = (160 * 1024) / (lds + 512);
  if (per_cu > MINW / 2) per_cu = MINW / 2;   // MINW waves per SIMD = MINW / 2 workgroups of 8 waves per CU
  if (per_cu < 1) per_cu = 1;
  int gx = 256 * per_cu / chunks;
  if (gx < 1) gx = 1;
  const int need = (geo.tiles + 7) / 8;
  if (gx > need) gx = need;
  hipLaunchKernelGGL(kern, dim3(gx, chunks), dim3(SNT), lds, st, p, geo);
  return dcc_check_launch();
}

template <int KK, int MINW>
int stream_launch(const GemmParams& p, const StreamGeo& geo, int lds, int chunks, hipStream_t st) {
  const int ef = p.flags & (GEMM_BIAS | GEMM_GELU | GEMM_SAVE_PRE | GEMM_RESID | GEMM_DGELU);
  if (ef == 0) return stream_launch_ep<KK, MINW, 0>(p, geo, lds, chunks, st);
  if (ef == GEMM_BIAS) return stream_launch_ep<KK, MINW, GEMM_BIAS>(p, geo, lds, chunks, st);
  if (ef == (GEMM_BIAS | GEMM_GELU | GEMM_SAVE_PRE)) return stream_launch_ep<KK, MINW, GEMM_BIAS | GEMM_GELU | GEMM_SAVE_PRE>(p, geo, lds, chunks, st);
  if (ef == (GEMM_BIAS | GEMM_RESID)) return stream_launch_ep<KK, MINW, GEMM_BIAS | GEMM_RESID>(p, geo, lds, chunks, st);
  if (ef == GEMM_DGELU) return stream_launch_ep<KK, MINW, GEMM_DGELU>(p, geo, lds, chunks, st);
  return stream_launch_ep<KK, MINW, -1>(p, geo, lds, chunks, st);
}

}  // namespace

// bf16 x bf16 -> bf16, row-major A [M,K] and W [N,K], K in {96, 192, 288, 384, 576}, N % 32 == 0, 16-byte aligned rows.
// DCC_ERR_UNSUPPORTED outside that domain (dcc_gemm then runs the tiled kernels).
int dcc_gemm_stream_launch(const GemmParams& p, hipStream_t st) {
  if (p.batch != 1 || p.splitk != 1 || (p.flags & (GEMM_ACCUM | GEMM_COLSUM_A))) return DCC_ERR_UNSUPPORTED;
  if ((p.N % 32) || (p.lda % 8) || (p.ldb % 8) || (p.ldc % 8) || p.M < 32) return DCC_ERR_UNSUPPORTED;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15) || ((uintptr_t)p.C & 15)) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & GEMM_RESID) && ((p.ldr % 8) || ((uintptr_t)p.R & 15))) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & (GEMM_SAVE_PRE | GEMM_DGELU)) && ((p.ldaux % 8) || ((uintptr_t)p.aux & 15))) return DCC_ERR_UNSUPPORTED;
  if ((p.flags & GEMM_BIAS) && ((uintptr_t)p.bias & 3)) return DCC_ERR_UNSUPPORTED;
  if (p.K != 96 && p.K != 192 && p.K != 288 && p.K != 384 && p.K != 576) return DCC_ERR_UNSUPPORTED;
  StreamGeo geo;
  static const int sdbg = getenv("DCC_STREAM_DEBUG") ? atoi(getenv("DCC_STREAM_DEBUG")) : 0;
  geo.dbg = sdbg;
  geo.tiles = (p.M + 31) / 32;
  geo.pitch = 2 * p.K + 16;
  const int budget = 150 * 1024 - STG_B;   // + the waves' store-staging strips
  int nc = (budget / (geo.pitch + 4)) & ~31;
  if (nc < 32) return DCC_ERR_UNSUPPORTED;
  int chunks = (p.N + nc - 1) / nc;
  if (chunks > 4) return DCC_ERR_UNSUPPORTED;
  nc = (((p.N + chunks - 1) / chunks) + 31) & ~31;   // equal chunks
  chunks = (p.N + nc - 1) / nc;
  geo.nc = nc;
  const int lds = nc * (geo.pitch + 4) + STG_B;
  // registers: the 32 x K operand is K / 4 VGPRs; K <= 192 leaves room for four waves per SIMD (two workgroups per CU)
  if (p.K == 96) return stream_launch<96, 4>(p, geo, lds, chunks, st);
  if (p.K == 192) return stream_launch<192, 4>(p, geo, lds, chunks, st);
  if (p.K == 288) return stream_launch<288, 2>(p, geo, lds, chunks, st);
  if (p.K == 384) return stream_launch<384, 2>(p, geo, lds, chunks, st);
  return stream_launch<576, 2>(p, geo, lds, chunks, st);
}
