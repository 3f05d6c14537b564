"""Epoch-end retrieval metrics on the GPU (dcc_retrieval_ranks + host arithmetic) against the values produced by the
reference's utils/retrieval_metrics.py (tests/golden/retrieval.npz) and against the CPU oracle; ranks are integers,
so parity is exact."""
import math
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


@pytest.mark.parametrize("rows,M,k", [(1, 1, 1), (7, 5, 5), (24, 1000, 5), (3, 40000, 10), (64, 257, 32)])
def test_topk_rows_is_the_head_of_a_stable_descending_sort(rows, M, k):
    """inference()'s torch.topk (runner :1731-1758): values descending, ties by column, NaN first; exact."""
    from deepcoro_clip_amd import kernels as K

    g = torch.Generator().manual_seed(rows * 31 + M)
    for sim in (torch.randn(rows, M, generator=g), torch.randint(-3, 4, (rows, M), generator=g).float() * 0.5):
        if M > 3:
            sim[0, 2] = float("inf"); sim[0, 1] = float("-inf"); sim[rows - 1, M - 1] = float("nan")
        wide = torch.zeros(rows, M + 3)
        wide[:, :M] = sim
        val, idx = K.topk_rows(wide.cuda()[:, :M], k)
        key = torch.where(torch.isnan(sim), torch.full_like(sim, float("inf")), sim).double() + torch.isnan(sim).double()
        ref = torch.sort(key, dim=1, descending=True, stable=True)[1][:, :k]   # NaN above +inf, ties by column
        assert torch.equal(idx.cpu(), ref)
        got, want = val.cpu(), torch.gather(sim, 1, ref)
        assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(got[~torch.isnan(got)], want[~torch.isnan(want)])
        if not torch.isnan(sim).any() and sim.unique().numel() == sim.numel():
            assert torch.equal(idx.cpu(), torch.topk(sim, k, dim=1)[1])


class _ToyTokenizer:
    """word -> id hashing with padding/truncation, the HF call signature the runner uses."""

    def __call__(self, texts, padding=True, truncation=True, return_tensors="pt"):
        rows = [[2] + [3 + (sum(map(ord, w)) % 500) for w in t.split()][:14] + [1] for t in texts]
        n = max(len(r) for r in rows)
        ids = torch.tensor([r + [0] * (n - len(r)) for r in rows])
        return {"input_ids": ids, "attention_mask": (ids != 0).long()}


def test_val_epoch_reencodes_unique_texts_and_ranks_index_sets(tmp_path):
    """`_run_epoch('val')` follows the reference's epoch end (runner :853-1021): de-duplicated positive texts re-encoded by
    the text tower, [Nv, Mt] similarities, ground-truth index SETS; checked against the CPU oracle's metric functions on
    the same embeddings."""
    import __graft_entry__ as ge
    from oracle import metrics as OM

    runner, _, (videos, ids, mask) = ge.make_tiny_setup(torch.float32, B=2)
    tok = _ToyTokenizer()
    runner.text_encoder.tokenizer = tok
    runner.output_dir = str(tmp_path)
    reports = [["normal coronary arteries", "severe stenosis of the lad"], ["severe stenosis of the lad", "mild disease rca"],
               ["normal coronary arteries", "occluded graft"]]
    g = torch.Generator().manual_seed(5)
    batches = []
    for rep in reports:
        batches.append({"videos": torch.randn(2, *videos.shape[1:], generator=g), "encoded_texts": tok(rep), "reports": rep,
                        "paths": [f"clip{len(batches)}a.mp4", f"clip{len(batches)}b.mp4"]})
    runner.val_loader = batches
    out = runner._run_epoch("val", 0)
    uniq = sorted({t for rep in reports for t in rep})
    assert len(uniq) == 4 and os.path.exists(tmp_path / "val_unique_texts.csv")
    saved = torch.load(tmp_path / "val_text_embeddings_epoch_0.pt")
    assert saved["unique_texts"] == uniq and saved["text_embeddings"].shape == (len(uniq), 512)
    # expected: same video embeddings (eval mode), bank = the saved text embeddings, the oracle's metric functions
    runner.video_encoder.eval()
    with torch.no_grad():
        v = torch.cat([runner.video_encoder(b["videos"].cuda()).float().cpu() for b in batches])
    sim = torch.nn.functional.normalize(v, dim=1) @ torch.nn.functional.normalize(saved["text_embeddings"], dim=1).t()
    gt = [[saved["text_to_index"][t]] for rep in reports for t in rep]
    exp = dict(OM.recall_at_k(sim, gt, [1, 5]))
    exp.update(OM.mrr(sim, gt)); exp.update(OM.ndcg_at_k(sim, gt, [5]))
    exp["MAP"] = OM.mean_ap(sim, gt); exp["MedianRank_V2T"] = OM.median_rank(sim, gt)
    for k_, v_ in exp.items():
        assert abs(out[f"val/{k_}"] - float(v_)) < 1e-9, (k_, out[f"val/{k_}"], v_)
    assert math.isfinite(out["val/loss"])
    # a video whose report is missing from the text set is an error, as in the reference (:877-884)
    batches[0]["reports"] = ["", "severe stenosis of the lad"]
    with pytest.raises(ValueError):
        runner._run_epoch("val", 0)


def test_inference_topk_uses_raw_inner_products_like_the_reference():
    """inference(): torch.topk(video_embeddings @ text_embeddings.T, k) — unnormalised (reference runner :1757-1758)."""
    import __graft_entry__ as ge

    runner, _, (videos, ids, mask) = ge.make_tiny_setup(torch.float32, B=2)
    g = torch.Generator().manual_seed(9)
    bank = torch.randn(37, 512, generator=g) * torch.linspace(0.2, 3.0, 37)[:, None]  # very different text norms
    runner.val_loader = [{"videos": torch.randn(2, *videos.shape[1:], generator=g)} for _ in range(2)]
    got = runner.inference(text_embeddings=bank, topk=5)
    runner.video_encoder.eval()
    with torch.no_grad():
        v = torch.cat([runner.video_encoder(b["videos"].cuda()).float().cpu() for b in runner.val_loader])
    assert torch.equal(got, torch.topk(v @ bank.t(), 5, dim=1)[1])
