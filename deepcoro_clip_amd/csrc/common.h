// Shared device helpers for the gfx950 (CDNA4) kernels of the DeepCORO_CLIP hot path.
// Wave = 64 lanes. All matrix work uses v_mfma_f32_32x32x16_bf16 (bf16 operands) or
// v_mfma_f32_32x32x2_f32 (exact fp32 operands) behind one 8-element fragment abstraction:
//   lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j], B[k = 8h + j][col r], j = 0..7
// for one K=16 step.  For bf16 that is one MFMA instruction, for fp32 eight 32x32x2
// instructions (instruction j pairs k = j of half 0 with k = 8 + j of half 1), so the
// fragment images in LDS / registers are identical for both element types.
// C/D layout of a 32x32 tile: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DCC_OK 0
#define DCC_ERR_ARG -1
#define DCC_ERR_LAUNCH -2
#define DCC_ERR_UNSUPPORTED -3

// dtype codes of the C ABI
#define DCC_F32 0
#define DCC_BF16 1
#define DCC_F16 2   // dcc_cast only: the fp16 wire format of the SigLIP embedding gather

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// Store of a streaming output that nothing reads soon (the pre-activation a forward GEMM saves for its backward, 154 MB per
// MLP): the non-temporal hint keeps it from displacing the kernel's operands — and the tensor the NEXT kernel is about to read —
// in L2 / MALL.  Measured: with every epilogue store non-temporal the isolated fc1-forward GEMM goes 141 -> 122 us and dgelu
// 147 -> 114, but the step does not move (44.3 vs 44.1 ms): the consumer then fetches the output from HBM.  So only stores
// whose consumer is far away use it.  DCC_NO_STREAM_STORES restores ordinary stores for A/B builds.
#ifdef DCC_NO_STREAM_STORES
#define STREAM_STORE(ptr, val) (*(ptr) = (val))
#define STREAM_LOAD(ptr) (*(ptr))
#else
#define STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define STREAM_LOAD(ptr) __builtin_nontemporal_load(ptr)   // read-once operand of an epilogue (the saved pre-activation)
#endif

template <typename T> struct Frag;  // 8-element MFMA operand fragment
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x8 type; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// One K=16 MFMA step on 32x32 tiles.
__device__ __forceinline__ f32x16 mma16(const bf16x8& a, const bf16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma16(const f32x8& a, const f32x8& b, f32x16 c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
  return c;
}

// Accumulator registers 8*s .. 8*s+7 of a 32x32 tile as the fragment of K-step s of a following
// MFMA that sums over the tile's ROW index (cdna guide: accumulator tile as next operand).
// Element j of lane half h is row 16*s + 8*(j>>2) + 4*h + (j&3) of the tile.
template <typename T> __device__ __forceinline__ typename Frag<T>::type acc_as_frag(const f32x16& x, int s);
template <> __device__ __forceinline__ bf16x8 acc_as_frag<bf16_t>(const f32x16& x, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16_t)x[8 * s + j];
  return r;
}
template <> __device__ __forceinline__ f32x8 acc_as_frag<float>(const f32x16& x, int s) {
  f32x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = x[8 * s + j];
  return r;
}

// Row of accumulator register `reg` for lane half h.
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// 16-byte vector helpers -------------------------------------------------------------------
template <typename T> struct Vec16;  // the 16-byte global/LDS access unit
template <> struct Vec16<bf16_t> { typedef bf16x8 type; static constexpr int N = 8; };
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };

template <typename T> using vec16_t = typename Vec16<T>::type;
template <typename T> using frag_t = typename Frag<T>::type;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 round-off): one v_exp + one v_rcp + 6 FMAs
// instead of libm erff's ~40 instructions; GELU sits in GEMM epilogues where VALU time is on the critical path.
__device__ __forceinline__ float erf_fast(float x, float* exp_mx2) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);  // bare v_rcp_f32 (1 ulp); __frcp_rn expands to a full IEEE division
  const float e = __expf(-ax * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * e;
  if (exp_mx2) *exp_mx2 = e;
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) {  // exact-erf GELU (nn.GELU default)
  return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f, nullptr));
}
__device__ __forceinline__ float dgelu_f(float x) {
  float e;  // exp(-(x/sqrt2)^2) = exp(-x^2/2), shared with the pdf term
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f, &e));
  return cdf + x * 0.39894228040143267794f * e;
}

// GELU for bf16 OUTPUTS (the GEMM epilogues, where the exp/rcp/FMA count of the erf form sets the pace of the whole kernel):
// x * sigmoid(x (a + b x^2)) with (a, b) fitted to x Phi(x), in 6 VALU + 2 transcendental instructions instead of ~17 + 2.
// Error against the erf form: |gelu_bf16 - gelu| <= 2.8e-4 absolute (worst near x = -1.74), |dgelu_bf16 - gelu'| <= 8.8e-4
// (near x = -1.24).  That is below the bf16 rounding of the value only where |gelu| > ~0.15 (2^-9 relative); for small
// |gelu| it is a SYSTEMATIC bias (mean +4e-5 over [-8, 8]), not rounding noise — bounded by tests/test_kernels_gpu.py::
// test_bf16_gelu_pair_against_erf, and switchable: -DDCC_EXACT_GELU (DCC_CFLAGS) builds every bf16 epilogue with the erf form
// for parity runs.  dgelu_bf16 is the exact derivative of gelu_bf16, so forward and backward stay consistent.  fp32 compute
// always uses the erf form above.
__device__ __forceinline__ float gelu_bf16(float x) {
  const float u = x * (1.60031417f + 0.06940178f * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u));
}
__device__ __forceinline__ float dgelu_bf16(float x) {
  const float x2 = x * x;
  const float u = x * (1.60031417f + 0.06940178f * x2);
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u));
  return s * (1.0f + x * (1.60031417f + 0.20820534f * x2) * (1.0f - s));
}

// element-type dispatch: bf16 tensors take the cheap pair everywhere (one definition of GELU per compute dtype)
#ifdef DCC_EXACT_GELU   // parity builds: erf form for every dtype
template <typename T> __device__ __forceinline__ float gelu_for(float x) { return gelu_f(x); }
template <typename T> __device__ __forceinline__ float dgelu_for(float x) { return dgelu_f(x); }
#else
template <typename T> __device__ __forceinline__ float gelu_for(float x) {
  if constexpr (sizeof(T) == 2) return gelu_bf16(x); else return gelu_f(x);
}
template <typename T> __device__ __forceinline__ float dgelu_for(float x) {
  if constexpr (sizeof(T) == 2) return dgelu_bf16(x); else return dgelu_f(x);
}
#endif

// Counter-based keep masks (dropout): splitmix64 finaliser of (seed, element index); a mask is regenerated from the
// same (seed, index) wherever it is needed again (backward) instead of being stored.
__device__ __forceinline__ uint32_t hash32(uint64_t k) {
  k += 0x9e3779b97f4a7c15ULL;
  k = (k ^ (k >> 30)) * 0xbf58476d1ce4e5b9ULL;
  k = (k ^ (k >> 27)) * 0x94d049bb133111ebULL;
  k ^= k >> 31;
  return (uint32_t)(k >> 32);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) holds per DEVICE: the launchers remember it in a bit per device, not in one
// process-wide flag (a second GPU driven by the same process would otherwise launch its 150 KB kernels unconfigured).
static inline unsigned long long dcc_dev_bit() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return 1ull << (dev & 63);
}

static inline int dcc_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DCC_OK : DCC_ERR_LAUNCH;
}
