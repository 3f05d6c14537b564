// Step tail (reference: runners/video_constrative_learning_runner.py:1398-1421): global gradient L2 norm,
// clip coefficient, AdamW.  Multi-tensor: one launch walks a device table of tensor descriptors,
// a (tensor, chunk) work list maps blocks to 16 KiB chunks, so ~500 parameter tensors cost one
// launch each for the norm and the update (HBM-bound: 4 B/param for the norm, 28 B/param for AdamW,
// + 2 B/param when the bf16 shadow weight is refreshed in the same pass).
#include "common.h"

struct DccTensorDesc {   // mirrored by ctypes in optim.py
  float* p;              // parameter (fp32 master)
  float* g;              // gradient (fp32)
  float* m;              // exp_avg
  float* v;              // exp_avg_sq
  void* shadow;          // optional bf16 copy refreshed by the update (may be null)
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay, bc1, bc2;  // bc = 1 - beta^step
  int flags;             // bit0: adamw (decoupled decay) else adam (L2 into grad)
};

namespace {

constexpr int CHUNK = 4096;  // elements per work item; DCC_OPT_CHUNK in include/dcc.h

// grid-stride over the (tensor, chunk) work list: a block folds many chunks in registers and ends with ONE fp64 atomic
// (one atomic per 16-KiB chunk on a single address was the whole cost of the first version)
__global__ __launch_bounds__(256) void sumsq_kernel(const DccTensorDesc* __restrict__ tab, const int2* __restrict__ work,
                                                    int nwork, double* __restrict__ out) {
  float acc = 0.0f;
  for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const int2 w = work[wi];
    const DccTensorDesc t = tab[w.x];
    if (!t.g) continue;
    const int64_t base = (int64_t)w.y * CHUNK;
    if (base + CHUNK <= t.n && (((uintptr_t)t.g) & 15) == 0) {
      const f32x4* g4 = reinterpret_cast<const f32x4*>(t.g + base);
#pragma unroll
      for (int i = 0; i < CHUNK / 4 / 256; ++i) {
        const f32x4 g = g4[threadIdx.x + 256 * i];
        acc += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
      }
    } else {
      for (int i = threadIdx.x; i < CHUNK; i += 256) {
        const int64_t j = base + i;
        if (j < t.n) { const float g = t.g[j]; acc += g * g; }
      }
    }
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (double)part[0] + (double)part[1] + (double)part[2] + (double)part[3]);
}

// coef = min(1, max_norm / (norm + 1e-6)); norm_out = norm   (torch.nn.utils.clip_grad_norm_ semantics)
__global__ void clip_coef_kernel(const double* __restrict__ sumsq, float max_norm, float* __restrict__ coef,
                                 float* __restrict__ norm_out) {
  const float norm = (float)sqrt(*sumsq);
  float c = max_norm / (norm + 1e-6f);
  if (c > 1.0f) c = 1.0f;
  *coef = c;
  if (norm_out) *norm_out = norm;
}

__global__ __launch_bounds__(256) void scale_grads_kernel(const DccTensorDesc* __restrict__ tab,
                                                          const int2* __restrict__ work, const float* __restrict__ coef) {
  const float c = *coef;
  if (c == 1.0f) return;
  const int2 w = work[blockIdx.x];
  const DccTensorDesc t = tab[w.x];
  if (!t.g) return;
  const int64_t base = (int64_t)w.y * CHUNK;
  for (int i = threadIdx.x; i < CHUNK; i += 256) {
    const int64_t j = base + i;
    if (j < t.n) t.g[j] *= c;
  }
}

// torch.optim.AdamW / Adam (amsgrad=False, maximize=False) element update
__global__ __launch_bounds__(256) void adam_kernel(const DccTensorDesc* __restrict__ tab, const int2* __restrict__ work,
                                                   const float* __restrict__ grad_scale) {
  const int2 w = work[blockIdx.x];
  const DccTensorDesc t = tab[w.x];
  if (!t.g) return;
  const float gs = (grad_scale && (t.flags & 2)) ? *grad_scale : 1.0f;  // flag 2: tensor is inside the clipped set
  const int64_t base = (int64_t)w.y * CHUNK;
  const float step_size = t.lr / t.bc1;
  const float bc2s = sqrtf(t.bc2);
  for (int i = threadIdx.x; i < CHUNK; i += 256) {
    const int64_t j = base + i;
    if (j >= t.n) break;
    float p = t.p[j];
    float g = t.g[j] * gs;
    if (t.flags & 1) p *= 1.0f - t.lr * t.weight_decay;
    else g += t.weight_decay * p;
    const float m = t.beta1 * t.m[j] + (1.0f - t.beta1) * g;
    const float v = t.beta2 * t.v[j] + (1.0f - t.beta2) * g * g;
    const float denom = sqrtf(v) / bc2s + t.eps;
    p -= step_size * (m / denom);
    t.p[j] = p; t.m[j] = m; t.v[j] = v;
    if (t.shadow) reinterpret_cast<bf16_t*>(t.shadow)[j] = (bf16_t)p;
  }
}

}  // namespace

// tab: device array of DccTensorDesc; work: device int2[nblocks] = (tensor index, chunk index)
extern "C" int dcc_grad_sumsq(const void* tab, const void* work, int nblocks, double* out, void* stream) {
  if (!tab || !work || !out || nblocks <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblocks < 2048 ? nblocks : 2048), dim3(256), 0, (hipStream_t)stream,
                     (const DccTensorDesc*)tab, (const int2*)work, nblocks, out);
  return dcc_check_launch();
}
extern "C" int dcc_clip_coef(const double* sumsq, float max_norm, float* coef, float* norm_out, void* stream) {
  if (!sumsq || !coef) return DCC_ERR_ARG;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, coef, norm_out);
  return dcc_check_launch();
}
extern "C" int dcc_scale_grads(const void* tab, const void* work, int nblocks, const float* coef, void* stream) {
  if (!tab || !work || !coef || nblocks <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(scale_grads_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const DccTensorDesc*)tab,
                     (const int2*)work, coef);
  return dcc_check_launch();
}
extern "C" int dcc_adam_step(const void* tab, const void* work, int nblocks, const float* grad_scale, void* stream) {
  if (!tab || !work || nblocks <= 0) return DCC_ERR_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const DccTensorDesc*)tab,
                     (const int2*)work, grad_scale);
  return dcc_check_launch();
}
