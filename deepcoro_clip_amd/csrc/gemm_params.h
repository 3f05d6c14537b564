// Parameter block and epilogue flags shared by the two GEMM kernels (gemm.hip: register-staged tiles for fp32 /
// small / odd shapes; gemm_ring.hip: persistent LDS-DMA ring kernel for the large bf16 GEMMs of the towers).
#pragma once
#include "common.h"

#define GEMM_BIAS 1
#define GEMM_GELU 2
#define GEMM_SAVE_PRE 4
#define GEMM_RESID 8
#define GEMM_DGELU 16
#define GEMM_ACCUM 32
#define GEMM_COLSUM_A 64

struct GemmParams {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const void* R;
  void* aux;
  float* colsum;  // GEMM_COLSUM_A: colsum[m] += sum_k A(m,k)   (bias gradient fused into the weight-gradient GEMM)
  int64_t lda, ldb, ldc, ldr, ldaux;
  int64_t sA, sB, sC;
  int M, N, K;
  int batch, splitk;
  float alpha;
  int flags;
  // fp8 operands (gemm_ring.hip, F8 variant): per-tensor dequantisation scales on the device (alpha *= *scale_a * *scale_b)
  // and the element formats of A / B (0 = e4m3, 1 = e5m2)
  const float* scale_a;
  const float* scale_b;
  int fmt_a, fmt_b;
};

// Persistent ring kernel (gemm_ring.hip). Returns DCC_OK, or DCC_ERR_UNSUPPORTED when the shape/layout is outside
// what it handles (the caller then runs the register-staged kernel).
int dcc_gemm_ring_launch(const GemmParams& p, int a_km, int b_km, int out_dtype, hipStream_t stream);
// fp8 x fp8 -> bf16 with the block-scaled MFMA (k-contiguous operands only)
int dcc_gemm_ring_launch_fp8(const GemmParams& p, hipStream_t stream);
// Wide-tile kernel (gemm_wide.hip): 256 x bn tiles (bn = 256 | 128), 8 waves.  DCC_ERR_UNSUPPORTED outside its domain.
int dcc_gemm_wide_launch(const GemmParams& p, int a_km, int b_km, int out_dtype, int bn, int ns, hipStream_t stream);
// Streaming panel kernel (gemm_stream.hip): row-major bf16 A [M,K] x W [N,K]^T -> bf16 with the weight resident in LDS, for the
// early MViT blocks' short-reduction GEMMs (K in {96, 192, 288, 384, 576}, N % 32 == 0).  DCC_ERR_UNSUPPORTED outside its domain.
int dcc_gemm_stream_launch(const GemmParams& p, hipStream_t stream);
// Eight-phase 256 x 256 kernel (gemm_p8.hip; dispatched for the qkv projections of stages 2-3, variant 13 forces it): k-contiguous bf16 A [M,K], B [N,K] -> bf16,
// K % 64 == 0, K >= 128.  DCC_ERR_UNSUPPORTED outside its domain.
int dcc_gemm_p8_launch(const GemmParams& p, hipStream_t stream);
