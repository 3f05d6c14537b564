"""ORACLE (test infrastructure, CPU only) — retrieval metrics restated from utils/retrieval_metrics.py:65-324.

Integer ranks are the parity target (bit-exact); floats to 1e-6.  Pinned by the reference's own known answers
(tests/test_retrieval_metrics.py:11-150) and by tests/golden/retrieval.npz.
"""
from __future__ import annotations

import math

import numpy as np


def _gt_sets(gt, n):
    out = []
    for i in range(n):
        if i >= len(gt):
            out.append(set())
            continue
        e = gt[i]
        if isinstance(e, (list, tuple, set, np.ndarray)):
            out.append({int(x) for x in e if x is not None and int(x) >= 0})
        elif e is None:
            out.append(set())
        else:
            out.append({int(e)} if int(e) >= 0 else set())
    return out


def _order(sim):
    # descending, stable w.r.t. index like torch.argsort(descending=True) on distinct values
    return np.argsort(-np.asarray(sim, dtype=np.float64), axis=1, kind="stable")


def best_ranks(sim, gt):
    sim = np.asarray(sim)
    order = _order(sim)
    sets = _gt_sets(gt, sim.shape[0])
    ranks = []
    for i, s in enumerate(sets):
        pos = [int(np.nonzero(order[i] == g)[0][0]) + 1 for g in s if 0 <= g < sim.shape[1]]
        ranks.append(min(pos) if pos else None)
    return ranks


def recall_at_k(sim, gt, ks):
    sim = np.asarray(sim)
    order = _order(sim)
    sets = _gt_sets(gt, sim.shape[0])
    out = {}
    for k in ks:
        ku = min(k, sim.shape[1])
        hits = [1.0 if (s and any(int(j) in s for j in order[i, :ku])) else 0.0 for i, s in enumerate(sets)]
        out[f"Recall@{k}"] = float(sum(hits) / len(hits)) if hits else 0.0
    return out


def mrr(sim, gt):
    sim = np.nan_to_num(np.asarray(sim, dtype=np.float64), nan=0.0, posinf=1e4, neginf=-1e4)
    if sim.shape[1] == 1:
        return {"MRR_V2T": 1.0}
    r = best_ranks(sim, gt)
    vals = [0.0 if x is None else 1.0 / x for x in r]
    return {"MRR_V2T": sum(vals) / len(vals) if vals else 0.0}


def median_rank(sim, gt):
    sim = np.asarray(sim)
    if sim.shape[0] == 0:
        return 0
    r = [x if x is not None else sim.shape[1] for x in best_ranks(sim, gt)]
    r = sorted(float(x) for x in r)
    return int(r[(len(r) - 1) // 2])  # torch.median returns the lower of the two middle values


def mean_ap(sim, gt):
    sim = np.asarray(sim)
    order = _order(sim)
    sets = _gt_sets(gt, sim.shape[0])
    aps = []
    for i, s in enumerate(sets):
        hits, ps = 0, 0.0
        for rank, j in enumerate(order[i], start=1):
            if int(j) in s:
                hits += 1
                ps += hits / rank
        aps.append(ps / hits if hits else 0.0)
    return float(np.float32(np.mean(np.asarray(aps, dtype=np.float32)))) if aps else 0.0


def ndcg_at_k(sim, gt, ks):
    sim = np.asarray(sim)
    order = _order(sim)
    sets = _gt_sets(gt, sim.shape[0])
    out = {}
    for k in ks:
        ek = min(k, sim.shape[1])
        vals = []
        for i, s in enumerate(sets):
            if not s:
                vals.append(0.0)
                continue
            dcg = sum(1.0 / math.log2(r + 2) for r in range(ek) if int(order[i, r]) in s)
            ideal = min(len(s), ek)
            idcg = sum(1.0 / math.log2(r + 2) for r in range(ideal))
            vals.append(dcg / idcg if idcg > 0 else 0.0)
        out[f"NDCG@{k}_V2T"] = float(np.float32(np.mean(np.asarray(vals, dtype=np.float32))))
    return out
