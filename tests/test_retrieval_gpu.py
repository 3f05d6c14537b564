"""Epoch-end retrieval metrics on the GPU (dcc_retrieval_ranks + host arithmetic) against the values produced by the
reference's utils/retrieval_metrics.py (tests/golden/retrieval.npz) and against the CPU oracle; ranks are integers,
so parity is exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _np_ranks(sim, gt):
    order = np.argsort(-sim.astype(np.float64), axis=1, kind="stable")
    out = np.zeros_like(gt)
    for i in range(sim.shape[0]):
        inv = np.empty(sim.shape[1], dtype=np.int64)
        inv[order[i]] = np.arange(sim.shape[1])
        for s in range(gt.shape[1]):
            j = gt[i, s]
            out[i, s] = inv[j] + 1 if 0 <= j < sim.shape[1] else 0
    return out


@pytest.mark.parametrize("Nq,Nc,G_", [(100, 100, 1), (37, 1000, 3), (5, 7, 2), (1, 1, 1), (300, 4096, 4)])
def test_rank_kernel_exact_with_ties(Nq, Nc, G_):
    from deepcoro_clip_amd import kernels as K

    g = torch.Generator().manual_seed(Nq + Nc)
    sim = torch.randint(-5, 6, (Nq, Nc), generator=g).float() * 0.25  # heavy ties: the stable order matters
    gt = torch.randint(-1, Nc, (Nq, G_), generator=g).to(torch.int32)
    got = K.retrieval_ranks(sim.cuda(), gt.cuda()).cpu().numpy()
    assert np.array_equal(got, _np_ranks(sim.numpy(), gt.numpy()))
    wide = torch.zeros(Nq, Nc + 9)
    wide[:, :Nc] = sim  # a row-strided view
    got2 = K.retrieval_ranks(wide.cuda()[:, :Nc], gt.cuda()).cpu().numpy()
    assert np.array_equal(got2, got)


def test_metrics_match_reference_goldens():
    from deepcoro_clip_amd import retrieval_metrics as RM

    g = np.load(os.path.join(G, "retrieval.npz"))
    sim, gt = torch.from_numpy(g["sim"]).cuda(), torch.arange(100)
    rec = RM.compute_recall_at_k(sim, gt, [1, 5, 10])
    assert [rec["Recall@1"], rec["Recall@5"], rec["Recall@10"]] == pytest.approx(list(g["recall"]))
    assert RM.compute_mrr(sim, gt)["MRR_V2T"] == pytest.approx(float(g["mrr"]), abs=1e-9)
    assert RM.compute_map(sim, gt) == pytest.approx(float(g["map"]), abs=1e-7)
    assert RM.compute_median_rank(sim, gt) == int(g["median"])
    nd = RM.compute_ndcg_at_k(sim, gt, [5, 10])
    assert [nd["NDCG@5_V2T"], nd["NDCG@10_V2T"]] == pytest.approx(list(g["ndcg"]), abs=1e-7)
    gt2 = [[int(x) for x in row if x >= 0] for row in g["gt2"]]
    sim2 = torch.from_numpy(g["sim2"]).cuda()
    rec2 = RM.compute_recall_at_k(sim2, gt2, [1, 3, 10])
    assert [rec2["Recall@1"], rec2["Recall@3"], rec2["Recall@10"]] == pytest.approx(list(g["recall2"]))
    assert RM.compute_mrr(sim2, gt2)["MRR_V2T"] == pytest.approx(float(g["mrr2"]), abs=1e-9)
    assert RM.compute_map(sim2, gt2) == pytest.approx(float(g["map2"]), abs=1e-7)
    assert RM.compute_median_rank(sim2, gt2) == int(g["median2"])
    assert RM.compute_ndcg_at_k(sim2, gt2, [3])["NDCG@3_V2T"] == pytest.approx(float(g["ndcg2"][0]), abs=1e-7)
    # embedding-level helpers
    ev, et = torch.from_numpy(g["ev"]).cuda(), torch.from_numpy(g["et"]).cuda()
    norms = RM.compute_embedding_norms(ev, et)
    assert [norms["video_norm"], norms["text_norm"]] == pytest.approx(list(g["norms"]), rel=1e-5)
    assert RM.compute_alignment_score(ev, et) == pytest.approx(float(g["align"]), abs=1e-5)


def test_metrics_match_oracle_multi_positive_and_edge_cases():
    from deepcoro_clip_amd import retrieval_metrics as RM
    from oracle import metrics as OM

    g = torch.Generator().manual_seed(5)
    v, t = torch.randn(40, 512, generator=g), torch.randn(55, 512, generator=g)
    sim = RM.compute_similarity_matrix(v.cuda(), t.cuda())
    ref_sim = torch.nn.functional.normalize(v, dim=1) @ torch.nn.functional.normalize(t, dim=1).t()
    assert (sim.cpu() - ref_sim).abs().max() < 1e-5
    gt = [[int(x) for x in torch.randint(0, 55, (int(torch.randint(0, 4, (1,), generator=g)),), generator=g)] for _ in range(40)]
    gt[3] = []  # a query without ground truth
    gt[7] = [54, 54, 0]
    simc = sim.cpu()
    simn = simc.numpy()
    assert RM.compute_recall_at_k(sim, gt, [1, 5, 100]) == pytest.approx(OM.recall_at_k(simn, gt, [1, 5, 100]))
    assert RM.compute_mrr(sim, gt)["MRR_V2T"] == pytest.approx(OM.mrr(simn, gt)["MRR_V2T"], abs=1e-9)
    assert RM.compute_map(sim, gt) == pytest.approx(OM.mean_ap(simn, gt), abs=1e-6)
    assert RM.compute_median_rank(sim, gt) == OM.median_rank(simn, gt)
    assert RM.compute_ndcg_at_k(sim, gt, [1, 5]) == pytest.approx(OM.ndcg_at_k(simn, gt, [1, 5]), abs=1e-6)
    one = torch.randn(6, 1).cuda()
    assert RM.compute_mrr(one, torch.zeros(6, dtype=torch.long))["MRR_V2T"] == 1.0  # reference: single text -> 1.0
