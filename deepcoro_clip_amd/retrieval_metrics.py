"""Retrieval metrics with the reference's function names and return types (utils/retrieval_metrics.py:65-324).

These run once per epoch on rank 0 over [Nv, Mt] similarities ("next" row); the rank computation here
is host-side integer logic on the similarity matrix copied back once (no per-element `.item()` loops as in the
reference).  The per-step diagnostics (`compute_embedding_norms`, alignment) come out of the fused loss kernels
instead (runner.py)."""
from __future__ import annotations

import math
from typing import Iterable, List, Sequence, Set, Union

import numpy as np
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


def _order(similarity_matrix: torch.Tensor) -> np.ndarray:
    """Descending candidate order per query (ties by index, like a stable descending argsort)."""
    sim = similarity_matrix.detach().float().cpu().numpy().astype(np.float64)
    return np.argsort(-sim, axis=1, kind="stable")


def _best_ranks(order: np.ndarray, gt_sets, num_candidates):
    ranks = []
    for i, s in enumerate(gt_sets):
        best = None
        if s:
            inv = np.empty(order.shape[1], dtype=np.int64)
            inv[order[i]] = np.arange(order.shape[1])
            pos = [int(inv[g]) + 1 for g in s if 0 <= g < num_candidates]
            best = min(pos) if pos else None
        ranks.append(best)
    return ranks


def compute_recall_at_k(similarity_matrix: torch.Tensor, global_gt_indices, k_values: List[int] = [1, 5]) -> dict:
    gt_sets = _normalize_ground_truth_sets(global_gt_indices, similarity_matrix.size(0))
    order = _order(similarity_matrix)
    n_cand = similarity_matrix.size(1)
    metrics = {}
    for k in k_values:
        k_use = min(k, n_cand)
        if n_cand < k:
            print(f"Warning: similarity matrix has only {n_cand} candidates; adjusting Recall@{k} to Recall@{n_cand}.")
        hits = [1.0 if (s and any(int(j) in s for j in order[i, :k_use])) else 0.0 for i, s in enumerate(gt_sets)]
        metrics[f"Recall@{k}"] = float(sum(hits) / len(hits)) if hits else 0.0
    return metrics


def compute_mrr(similarity_matrix: torch.Tensor, global_gt_indices) -> dict:
    if similarity_matrix.dim() != 2:
        print(f"Warning: similarity_matrix has {similarity_matrix.dim()} dimensions, expected 2")
        return {"MRR_V2T": 0.0}
    n_v, n_t = similarity_matrix.shape
    sim = torch.nan_to_num(similarity_matrix.detach().float().cpu(), nan=0.0, posinf=1e4, neginf=-1e4)
    if n_t == 1:
        return {"MRR_V2T": 1.0}
    gt_sets = _normalize_ground_truth_sets(global_gt_indices, n_v)
    ranks = _best_ranks(_order(sim), gt_sets, n_t)
    vals = [0.0 if r is None else 1.0 / r for r in ranks]
    return {"MRR_V2T": sum(vals) / len(vals) if vals else 0.0}


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
    order = _order(similarity_matrix)
    metrics = {}
    for k in k_values:
        ek = min(k, n_c)
        vals = []
        for i, s in enumerate(gt_sets):
            if not s:
                vals.append(0.0)
                continue
            dcg = sum(1.0 / math.log2(r + 2) for r in range(ek) if int(order[i, r]) in s)
            ideal = min(len(s), ek)
            idcg = sum(1.0 / math.log2(r + 2) for r in range(ideal))
            vals.append(dcg / idcg if idcg > 0 else 0.0)
        metrics[f"NDCG@{k}_V2T"] = float(torch.tensor(vals).mean().item())
    return metrics


def compute_median_rank(similarity_matrix: torch.Tensor, global_gt_indices) -> int:
    n_q, n_c = similarity_matrix.shape
    if n_q == 0:
        return 0
    gt_sets = _normalize_ground_truth_sets(global_gt_indices, n_q)
    ranks = [r if r is not None else n_c for r in _best_ranks(_order(similarity_matrix), gt_sets, n_c)]
    return int(torch.tensor(ranks, dtype=torch.float).median().item())


def compute_map(similarity_matrix: torch.Tensor, global_gt_indices) -> float:
    n_q = similarity_matrix.size(0)
    if n_q == 0:
        return 0.0
    gt_sets = _normalize_ground_truth_sets(global_gt_indices, n_q)
    order = _order(similarity_matrix)
    aps = []
    for i, s in enumerate(gt_sets):
        hits, ps = 0, 0.0
        if s:
            for rank, j in enumerate(order[i], start=1):
                if int(j) in s:
                    hits += 1
                    ps += hits / rank
        aps.append(ps / hits if hits else 0.0)
    return float(torch.tensor(aps).mean().item())
