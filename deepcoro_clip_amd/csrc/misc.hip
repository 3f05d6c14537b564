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
