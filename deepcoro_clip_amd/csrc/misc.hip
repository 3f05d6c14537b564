// Library identity + tiny utilities.
#include "common.h"

extern "C" int dcc_version() { return 0; }  // returns 0 (DCC_OK) so that the generic int-rc check applies


// Epoch-end retrieval metrics (utils/retrieval_metrics.py:65-324): every metric the reference derives from a full
// descending sort of each similarity row (Recall@k, MRR, MAP, NDCG@k, median rank) only needs the RANK of the ground-
// truth candidates.  rank(i, j) = 1 + #{c : sim[i][c] > sim[i][j]} + #{c < j : sim[i][c] == sim[i][j]} is exactly the
// position of j in a stable descending argsort; one wave counts it for one (query, ground-truth slot) pair.
// gt: int32 [Nq, G], -1 = empty slot (rank 0 is written for empty / out-of-range slots).  Integer-exact.
namespace {
__global__ __launch_bounds__(256) void retrieval_rank_kernel(const float* __restrict__ sim, int64_t lds_, const int* __restrict__ gt,
                                                             int* __restrict__ ranks, int Nq, int Nc, int G) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (int64_t)Nq * G) return;
  const int i = pair / G;
  const int j = gt[pair];
  if (j < 0 || j >= Nc) {
    if (lane == 0) ranks[pair] = 0;
    return;
  }
  const float* row = sim + (int64_t)i * lds_;
  const float ref = row[j];
  int cnt = 0;
  for (int c = lane; c < Nc; c += 64) {
    const float v = row[c];
    cnt += (v > ref) || (v == ref && c < j);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) ranks[pair] = cnt + 1;
}
}  // namespace

extern "C" int dcc_retrieval_ranks(const float* sim, int64_t ld, const int* gt, int* ranks, int Nq, int Nc, int G, void* stream) {
  if (!sim || !gt || !ranks || Nq <= 0 || Nc <= 0 || G <= 0) return DCC_ERR_ARG;
  const int64_t pairs = (int64_t)Nq * G;
  hipLaunchKernelGGL(retrieval_rank_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sim, ld, gt, ranks, Nq, Nc, G);
  return dcc_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Row-wise top-k of the [B, M] similarity matrix (inference(): runners/video_constrative_learning_runner.py:1731-1758,
// torch.topk(sim, k, dim=1)).  Order: value descending, NaN above +inf (torch's rule), equal values by ascending column —
// the first k entries of a stable descending sort.  One block per row; pass p finds the largest (value, column) key that is
// strictly below the key picked in pass p-1, so nothing is written back or masked (k passes over an L2-resident row).
namespace {
__device__ __forceinline__ unsigned long long topk_key(float v, int c) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotonic in the float order, NaN (positive payload) on top
  return ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)c);
}
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ sim, int64_t ld, int M, int k,
                                                        int64_t* __restrict__ idx, float* __restrict__ val) {
  __shared__ unsigned long long part[4];
  const float* row = sim + (int64_t)blockIdx.x * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long last = ~0ull;  // above every key
  for (int p = 0; p < k; ++p) {
    unsigned long long best = 0ull;  // below every key (a real key has bit 63 or a non-zero low word... 0 only for -NaN)
    for (int c = threadIdx.x; c < M; c += 256) {
      const unsigned long long key = topk_key(row[c], c);
      if (key < last && key > best) best = key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)best, o), hi = __shfl_xor((unsigned)(best >> 32), o);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      best = other > best ? other : best;
    }
    if (lane == 0) part[wave] = best;
    __syncthreads();
    unsigned long long b = part[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) b = part[w] > b ? part[w] : b;
    __syncthreads();
    last = b;
    if (threadIdx.x == 0) {
      const int c = (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFull));
      idx[(int64_t)blockIdx.x * k + p] = c;
      val[(int64_t)blockIdx.x * k + p] = row[c];
    }
  }
}
}  // namespace

extern "C" int dcc_topk_rows(const float* sim, int64_t ld, int rows, int M, int k, int64_t* idx, float* val, void* stream) {
  if (!sim || !idx || !val || rows <= 0 || M <= 0 || k <= 0 || k > M) return DCC_ERR_ARG;
  hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, sim, ld, M, k, idx, val);
  return dcc_check_launch();
}
