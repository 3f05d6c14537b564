#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 rbf2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, const unsigned* b, float* c) {
  float acc = 0.5f;
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rbf2, a[threadIdx.x]), __builtin_bit_cast(rbf2, b[threadIdx.x]), acc, false);
  c[threadIdx.x] = acc;
}
static unsigned pk(float lo, float hi) { unsigned l, h; memcpy(&l, &lo, 4); memcpy(&h, &hi, 4); return (l >> 16) | (h & 0xffff0000u); }
int main() {
  unsigned ha[4] = {pk(1, 2), pk(3, 0), pk(0, 5), pk(-1.5f, 0.25f)}, hb[4] = {pk(10, 100), pk(7, 9), pk(9, 7), pk(2, 4)};
  unsigned *a, *b; float* c; float hc[4];
  hipMalloc(&a, 16); hipMalloc(&b, 16); hipMalloc(&c, 16);
  hipMemcpy(a, ha, 16, hipMemcpyHostToDevice); hipMemcpy(b, hb, 16, hipMemcpyHostToDevice);
  k<<<1, 4>>>(a, b, c);
  hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  printf("%g %g %g %g (expect 210.5 21.5 35.5 -1.5)\n", hc[0], hc[1], hc[2], hc[3]);
  return 0;
}
