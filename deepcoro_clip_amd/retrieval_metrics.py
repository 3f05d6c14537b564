"""Retrieval metrics with the reference's function names and return types (utils/retrieval_metrics.py:65-324).

These run once per epoch on rank 0 over [Nv, Mt] similarities.  The similarity GEMM and the
rank of every ground-truth candidate are computed on the GPU (`dcc_gemm`, `dcc_retrieval_ranks`: integer-exact
positions in the stable descending order the reference obtains by sorting); the host only combines those few integers
(`*_from_ranks`, no `.item()` loops over the matrix as in the reference).  The per-step diagnostics (`compute_embedding_norms`, alignment) come out of the fused loss kernels
instead (runner.py)."""
from __future__ import annotations

import math
from typing import Iterable, List, Sequence, Set, Union

import torch


def _normalize_ground_truth_sets(ground_truth_indices, num_queries: int) -> List[Set[int]]:
    gt_sets: List[Set[int]] = []
    if isinstance(ground_truth_indices, torch.Tensor):
        if ground_truth_indices.ndim == 1:
            gt_sets = [{int(i)} for i in ground_truth_indices.tolist()]
        elif ground_truth_indices.ndim == 2:
            gt_sets = [{int(x) for x in row if x is not None and int(x) >= 0} for row in ground_truth_indices.tolist()]
        else:
            raise ValueError("ground_truth_indices tensor must be 1D or 2D for multi-label support")
    elif isinstance(ground_truth_indices, (list, tuple)):
        for entry in ground_truth_indices:
            if isinstance(entry, (list, tuple, set)):
                gt_sets.append({int(x) for x in entry if x is not None and int(x) >= 0})
            elif entry is None:
                gt_sets.append(set())
            else:
                gt_sets.append({int(entry)})
    else:
        raise TypeError(f"Unsupported ground_truth_indices type: {type(ground_truth_indices)}")
    if len(gt_sets) < num_queries:
        gt_sets.extend([set() for _ in range(num_queries - len(gt_sets))])
    elif len(gt_sets) > num_queries:
        gt_sets = gt_sets[:num_queries]
    cleaned = [{i for i in gt if i is not None and i >= 0} for gt in gt_sets]
    return cleaned if cleaned else [set() for _ in range(num_queries)]


def gt_ranks(similarity_matrix: torch.Tensor, gt_sets: List[Set[int]]) -> List[List[int]]:
    """Sorted 1-based ranks of every ground-truth candidate of every query in the stable descending order of its
    similarity row, computed on the GPU (`dcc_retrieval_ranks`): one small [Nq, G] int tensor comes back instead of
    the [Nq, Mt] matrix.  Everything below is integer / float arithmetic on these few numbers."""
    from . import kernels as K

    n_q, n_c = similarity_matrix.shape
    G = max(1, max((len(s) for s in gt_sets), default=1))
    table = torch.full((n_q, G), -1, dtype=torch.int32)
    for i, s in enumerate(gt_sets):
        valid = sorted(g for g in s if 0 <= g < n_c)
        if valid:
            table[i, :len(valid)] = torch.tensor(valid, dtype=torch.int32)
    sim = similarity_matrix.detach()
    if sim.dtype != torch.float32:
        sim = K.cast(sim.contiguous(), torch.float32)
    if sim.stride(1) != 1:
        sim = sim.contiguous()
    ranks = K.retrieval_ranks(sim, table.to(sim.device)).cpu()
    return [sorted(int(r) for r in row if r > 0) for row in ranks.tolist()]


def recall_from_ranks(ranks: List[List[int]], k: int) -> float:
    hits = [1.0 if (r and r[0] <= k) else 0.0 for r in ranks]
    return float(sum(hits) / len(hits)) if hits else 0.0


def mrr_from_ranks(ranks: List[List[int]]) -> float:
    vals = [1.0 / r[0] if r else 0.0 for r in ranks]
    return sum(vals) / len(vals) if vals else 0.0


def ndcg_from_ranks(ranks: List[List[int]], gt_sizes: List[int], k: int) -> List[float]:
    vals = []
    for r, n in zip(ranks, gt_sizes):
        if n == 0:
            vals.append(0.0)
            continue
        dcg = sum(1.0 / math.log2(x + 1) for x in r if x <= k)  # position x-1 in the reference's 0-based loop
        idcg = sum(1.0 / math.log2(j + 2) for j in range(min(n, k)))
        vals.append(dcg / idcg if idcg > 0 else 0.0)
    return vals


def ap_from_ranks(ranks: List[List[int]]) -> List[float]:
    return [sum((h + 1) / x for h, x in enumerate(r)) / len(r) if r else 0.0 for r in ranks]


def compute_recall_at_k(similarity_matrix: torch.Tensor, global_gt_indices, k_values: List[int] = [1, 5]) -> dict:
    gt_sets = _normalize_ground_truth_sets(global_gt_indices, similarity_matrix.size(0))
    ranks = gt_ranks(similarity_matrix, gt_sets)
    n_cand = similarity_matrix.size(1)
    metrics = {}
    for k in k_values:
        if n_cand < k:
            print(f"Warning: similarity matrix has only {n_cand} candidates; adjusting Recall@{k} to Recall@{n_cand}.")
        metrics[f"Recall@{k}"] = recall_from_ranks(ranks, min(k, n_cand))
    return metrics


def compute_mrr(similarity_matrix: torch.Tensor, global_gt_indices) -> dict:
    from . import kernels as K

    if similarity_matrix.dim() != 2:
        print(f"Warning: similarity_matrix has {similarity_matrix.dim()} dimensions, expected 2")
        return {"MRR_V2T": 0.0}
    n_v, n_t = similarity_matrix.shape
    if n_t == 1:
        return {"MRR_V2T": 1.0}
    sim = similarity_matrix.detach().float().contiguous().clone()
    K.nan_to_num_(sim, 0.0, 1e4, -1e4)  # utils/retrieval_metrics.py:117-121
    gt_sets = _normalize_ground_truth_sets(global_gt_indices, n_v)
    return {"MRR_V2T": mrr_from_ranks(gt_ranks(sim, gt_sets))}


def compute_similarity_matrix(video_features: torch.Tensor, text_features: torch.Tensor) -> torch.Tensor:
    from . import kernels as K

    v, _ = K.l2norm_fwd(video_features.float().contiguous())
    t, _ = K.l2norm_fwd(text_features.float().contiguous())
    return K.gemm(v, t)


def compute_embedding_norms(video_features: torch.Tensor, text_features: torch.Tensor) -> dict:
    from . import kernels as K

    s = torch.zeros(2, device=video_features.device, dtype=torch.float32)
    K.l2norm_fwd(video_features.float().contiguous(), norm_sum=s[0:1])
    K.l2norm_fwd(text_features.float().contiguous(), norm_sum=s[1:2])
    a, b = s.tolist()
    return {"video_norm": a / video_features.shape[0], "text_norm": b / text_features.shape[0]}


def compute_alignment_score(video_features, text_features, all_video_embeddings=None, all_text_embeddings=None,
                            global_ground_truth_indices_tensor=None) -> float:
    from . import kernels as K

    if all_video_embeddings is not None and all_text_embeddings is not None and global_ground_truth_indices_tensor is not None:
        video_features = all_video_embeddings
        text_features = all_text_embeddings[global_ground_truth_indices_tensor]
    v, _ = K.l2norm_fwd(video_features.float().contiguous())
    t, _ = K.l2norm_fwd(text_features.float().contiguous())
    sim = K.gemm(v, t)
    return float(torch.diagonal(sim).sum().item() / v.shape[0])


def compute_ndcg_at_k(similarity_matrix: torch.Tensor, global_gt_indices, k_values: List[int]) -> dict:
    n_q, n_c = similarity_matrix.shape
    if n_q == 0:
        return {}
    gt_sets = _normalize_ground_truth_sets(global_gt_indices, n_q)
    ranks = gt_ranks(similarity_matrix, gt_sets)
    sizes = [len(s) for s in gt_sets]
    return {f"NDCG@{k}_V2T": float(torch.tensor(ndcg_from_ranks(ranks, sizes, min(k, n_c))).mean().item()) for k in k_values}


def compute_median_rank(similarity_matrix: torch.Tensor, global_gt_indices) -> int:
    n_q, n_c = similarity_matrix.shape
    if n_q == 0:
        return 0
    gt_sets = _normalize_ground_truth_sets(global_gt_indices, n_q)
    best = [r[0] if r else n_c for r in gt_ranks(similarity_matrix, gt_sets)]
    return int(torch.tensor(best, dtype=torch.float).median().item())


def compute_map(similarity_matrix: torch.Tensor, global_gt_indices) -> float:
    n_q = similarity_matrix.size(0)
    if n_q == 0:
        return 0.0
    gt_sets = _normalize_ground_truth_sets(global_gt_indices, n_q)
    return float(torch.tensor(ap_from_ranks(gt_ranks(similarity_matrix, gt_sets))).mean().item())
