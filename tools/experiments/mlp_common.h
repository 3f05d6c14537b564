// Shared between the fused LayerNorm + MLP kernels (mlp_fused.hip: 32-row tiles; mlp_fused16.hip: 16-row tiles, C = 384).
#pragma once
#include "common.h"
#include <type_traits>

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


// 16-row-tile kernel (mlp_fused16.hip): C = 384
int mlp16_launch(const MlpParams& p, hipStream_t st);
int mlp16_packed_bytes();
int mlp16_pack(const void* w1, const void* w2, void* packed, hipStream_t st);
