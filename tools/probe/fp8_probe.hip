// Probe of the gfx950 fp8 MFMA operand layouts with exact integer data (tools/fp8_probe.py drives it).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
__global__ void probe_scaled(const int* a, const int* b, float* out, int scale_a, int scale_b, int fmt_a, int fmt_b) {
  const int lane = threadIdx.x;
  v8i va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = a[lane * 8 + i]; vb[i] = b[lane * 8 + i]; }
  f32x16 c = {0};
  if (fmt_a == 0 && fmt_b == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, 0, scale_a, 0, scale_b);
  else if (fmt_a == 1 && fmt_b == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 1, 0, 0, scale_a, 0, scale_b);
  else if (fmt_a == 0 && fmt_b == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 1, 0, scale_a, 0, scale_b);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 1, 1, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
__global__ void probe_plain(const long* a, const long* b, float* out) {
  const int lane = threadIdx.x;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[lane], b[lane], c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
// fp32 -> packed fp8 (e4m3 / e5m2) with the hardware conversion, 2 values per call
__global__ void probe_cvt(const float* x, int* out, int n2, int bf8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  int v = 0;
  if (bf8) v = __builtin_amdgcn_cvt_pk_bf8_f32(x[2 * i], x[2 * i + 1], v, false);
  else v = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], v, false);
  out[i] = v & 0xffff;
}
extern "C" int fp8_probe_scaled(const int* a, const int* b, float* out, int sa, int sb, int fa, int fb) {
  hipLaunchKernelGGL(probe_scaled, dim3(1), dim3(64), 0, 0, a, b, out, sa, sb, fa, fb);
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
extern "C" int fp8_probe_plain(const long* a, const long* b, float* out) {
  hipLaunchKernelGGL(probe_plain, dim3(1), dim3(64), 0, 0, a, b, out);
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
extern "C" int fp8_probe_cvt(const float* x, int* out, int n2, int bf8) {
  hipLaunchKernelGGL(probe_cvt, dim3((n2 + 255) / 256), dim3(256), 0, 0, x, out, n2, bf8);
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
