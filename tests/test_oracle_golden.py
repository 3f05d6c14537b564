"""CPU: the oracle (oracle/) against golden vectors produced by the REFERENCE's own modules
(tests/golden/make_golden.py) and against the reference's known-answer tests (tests/test_retrieval_metrics.py)."""
import math
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("B", [8, 24])
@pytest.mark.parametrize("name,kw", [("contrastive", {}), ("clip", {"clamp_temp": True}),
                                     ("clip_ls", {"clamp_temp": True, "label_smoothing": 0.1}),
                                     ("siglip_gated", {"gated": True})])
def test_ce_losses_match_reference(B, name, kw):
    from oracle.losses import contrastive_loss

    g = _load("losses.npz")
    v = _t(g[f"v{B}"]).requires_grad_(True)
    t = _t(g[f"t{B}"]).requires_grad_(True)
    lt = torch.log(torch.tensor(0.07)).requires_grad_(True)
    loss = contrastive_loss(v, t, lt, **kw)
    loss.backward()
    assert abs(loss.item() - float(g[f"{name}{B}_loss"])) < 1e-6
    assert np.allclose(v.grad.numpy(), g[f"{name}{B}_dv"], atol=1e-7)
    assert np.allclose(t.grad.numpy(), g[f"{name}{B}_dt"], atol=1e-7)
    assert abs(lt.grad.item() - float(g[f"{name}{B}_dlt"])) < 1e-5


def test_known_answers_from_survey():
    """Generator(0) randn(8,512) pairs, temperature 0.07."""
    g = _load("losses.npz")
    assert abs(float(g["contrastive8_loss"]) - 2.27005911) < 1e-6
    assert abs(float(g["clip8_loss"]) - 2.27005911) < 1e-6
    assert abs(float(g["siglip_gated8_loss"]) - 2.12487268) < 1e-6
    assert abs(float(g["siglip_bce8_loss"]) - 1.25289905) < 1e-6
    v, t = _t(g["v8"]), _t(g["t8"])
    sim = torch.nn.functional.normalize(v, dim=1) @ torch.nn.functional.normalize(t, dim=1).t()
    assert sim.argmax(1).tolist() == [4, 0, 6, 2, 6, 3, 4, 1]


def test_bce_and_multipositive_match_reference():
    from oracle.losses import runner_multipositive_loss, siglip_bce_loss

    g = _load("losses.npz")
    for B in (8, 24):
        v, t = _t(g[f"v{B}"]).requires_grad_(True), _t(g[f"t{B}"]).requires_grad_(True)
        lt = torch.log(torch.tensor(0.07)).requires_grad_(True)
        bias = torch.tensor(-10.0, requires_grad=True)
        loss = siglip_bce_loss(v, t, lt, bias)
        loss.backward()
        assert abs(loss.item() - float(g[f"siglip_bce{B}_loss"])) < 1e-6
        assert np.allclose(v.grad.numpy(), g[f"siglip_bce{B}_dv"], atol=1e-8)
        assert abs(bias.grad.item() - float(g[f"siglip_bce{B}_dbias"])) < 1e-6
    v, t = _t(g["mp_v"]).requires_grad_(True), _t(g["mp_t"]).requires_grad_(True)
    lt = torch.log(torch.tensor(0.07)).requires_grad_(True)
    mask, w = _t(g["mp_mask"]), _t(g["mp_w"])
    bias = torch.tensor(-10.0, requires_grad=True)
    loss = siglip_bce_loss(v, t, lt, bias, pos_mask=mask, pos_weights=w)
    assert abs(loss.item() - float(g["mp_bce_loss"])) < 1e-6
    loss, _, _ = runner_multipositive_loss(v, t, lt, mask, w, weighted=True)
    assert abs(loss.item() - float(g["mp_wce_loss"])) < 1e-6
    loss, _, _ = runner_multipositive_loss(v, t, lt, mask, w, weighted=False)
    assert abs(loss.item() - float(g["mp_rbce_loss"])) < 1e-6


def test_siglip_gating_inequality():
    """tests/test_losses.py:78-94 of the reference: x * sigmoid(x) <= x for the cosine range."""
    x = torch.linspace(-1, 1, 101)
    assert torch.all(x * torch.sigmoid(x) <= x.clamp(min=0) + 1e-7)


def test_rope_matches_reference():
    from oracle.mvit import Rope3DOracle, rope_tables

    g = _load("rope.npz")
    cos, sin = rope_tables(8, 7, 7, 96, n_special=1)
    assert np.allclose(cos.numpy(), g["cos"], atol=1e-6) and np.allclose(sin.numpy(), g["sin"], atol=1e-6)
    tok = _t(g["tok"]).long()
    q, k = _t(g["q"]), _t(g["k"])
    from oracle.mvit import rotate_half

    qr = q * cos[tok] + rotate_half(q) * sin[tok]
    kr = k * cos[tok] + rotate_half(k) * sin[tok]
    assert np.allclose(qr.numpy(), g["q_rot"], atol=1e-5) and np.allclose(kr.numpy(), g["k_rot"], atol=1e-5)
    c2, s2 = rope_tables(8, 14, 14, 96, n_special=1)
    sel = _t(g["sel"]).long()
    q2 = _t(g["q2_sel"])
    assert np.allclose((q2 * c2[sel] + rotate_half(q2) * s2[sel]).numpy(), g["q2_rot_sel"], atol=1e-5)
    k2 = _t(g["k2_sel"])
    assert np.allclose((k2 * cos[:8] + rotate_half(k2) * sin[:8]).numpy(), g["k2_rot_sel"], atol=1e-5)
    assert Rope3DOracle(96).one(torch.zeros(1, 1, 393, 96), (8, 7, 7)).shape == (1, 1, 393, 96)


def test_attention_pool_and_aggregator_match_reference():
    from oracle.model import AggregatorOracle, AttentionPoolOracle, init_by_name

    g = _load("pools.npz")
    ap = init_by_name(AttentionPoolOracle(512, 8))
    with torch.no_grad():
        assert np.allclose(ap(_t(g["ap_x"])).numpy(), g["ap_y"], atol=2e-5)
        ag = init_by_name(AggregatorOracle(512, 4, 2))
        for n in (1, 4, 8):
            assert np.allclose(ag(_t(g[f"ag_x{n}"])).numpy(), g[f"ag_y{n}"], atol=2e-5), n


def test_bert_matches_hf():
    from oracle.model import BertOracle, init_by_name

    g = _load("bert.npz")
    m = init_by_name(BertOracle(vocab=1000, layers=2, max_pos=64))
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    with torch.no_grad():
        h = m(_t(g["ids"]), _t(g["mask"]))
    assert np.allclose(h[:, 0].numpy(), g["cls"], atol=3e-5)
    assert np.allclose(h[[0, 2], 5].numpy(), g["tok5"][[0, 2]], atol=3e-5)  # row 1, token 5 is valid too
    assert np.allclose(h[1, 5].numpy(), g["tok5"][1], atol=3e-5)


def test_retrieval_metrics_match_reference():
    from oracle import metrics as M

    g = _load("retrieval.npz")
    sim, gt = g["sim"], list(range(100))
    rec = M.recall_at_k(sim, gt, [1, 5, 10])
    assert [rec["Recall@1"], rec["Recall@5"], rec["Recall@10"]] == pytest.approx(list(g["recall"]), abs=1e-12)
    assert M.mrr(sim, gt)["MRR_V2T"] == pytest.approx(float(g["mrr"]), abs=1e-9)
    assert M.mean_ap(sim, gt) == pytest.approx(float(g["map"]), abs=1e-7)
    assert M.median_rank(sim, gt) == int(g["median"])
    nd = M.ndcg_at_k(sim, gt, [5, 10])
    assert [nd["NDCG@5_V2T"], nd["NDCG@10_V2T"]] == pytest.approx(list(g["ndcg"]), abs=1e-7)
    # SURVEY §8c known answers for this seed
    assert list(g["recall"]) == pytest.approx([0.0, 0.03, 0.07]) and int(g["median"]) == 50
    assert float(g["mrr"]) == pytest.approx(0.0377008, abs=1e-6)
    sim2 = g["sim2"]
    gt2 = [[int(x) for x in row if x >= 0] for row in g["gt2"]]
    rec2 = M.recall_at_k(sim2, gt2, [1, 3, 10])
    assert [rec2["Recall@1"], rec2["Recall@3"], rec2["Recall@10"]] == pytest.approx(list(g["recall2"]), abs=1e-12)
    assert M.mrr(sim2, gt2)["MRR_V2T"] == pytest.approx(float(g["mrr2"]), abs=1e-9)
    assert M.mean_ap(sim2, gt2) == pytest.approx(float(g["map2"]), abs=1e-7)
    assert M.median_rank(sim2, gt2) == int(g["median2"])
    assert M.ndcg_at_k(sim2, gt2, [3])["NDCG@3_V2T"] == pytest.approx(float(g["ndcg2"][0]), abs=1e-7)


def test_retrieval_known_answers_of_reference_tests():
    """tests/test_retrieval_metrics.py:11-43,82-103 of the reference."""
    from oracle import metrics as M

    n = 5
    perfect = np.eye(n) * 2 - 1
    gt = list(range(n))
    assert M.recall_at_k(perfect, gt, [1, 5])["Recall@1"] == 1.0
    assert M.mrr(perfect, gt)["MRR_V2T"] == 1.0 and M.mean_ap(perfect, gt) == 1.0 and M.median_rank(perfect, gt) == 1
    rev = np.fliplr(np.eye(n)) * 2 - 1
    assert M.recall_at_k(rev, gt, [1])["Recall@1"] == pytest.approx(0.2)
    assert M.mrr(rev, gt)["MRR_V2T"] < 0.5 and M.median_rank(rev, gt) > 2.5


def test_mvit_structural_known_answers():
    """torchvision's published MViT-v2-S figures: 34 537 744 params with the 400-way head, i.e. 34 230 144 without."""
    from oracle.mvit import MViTv2S

    m = MViTv2S(16, (224, 224))
    assert sum(p.numel() for p in m.parameters()) == 34_230_144
    assert sum(p.numel() for p in m.parameters()) + 768 * 400 + 400 == 34_537_744
    keys = list(m.state_dict().keys())
    for k in ("conv_proj.weight", "pos_encoding.class_token", "blocks.0.attn.rel_pos_h", "blocks.0.attn.pool_q.pool.weight",
              "blocks.0.attn.pool_k.norm_act.0.weight", "blocks.1.project.weight", "blocks.15.mlp.3.bias", "norm.weight"):
        assert k in keys, k
    small = MViTv2S(4, (112, 112))
    with torch.no_grad():
        y = small.forward_features(torch.randn(1, 3, 4, 112, 112))
    assert y.shape == (1, 33, 768)  # BASELINE config 1: L = 2*4*4 + 1


def test_loss_zoo_oracle_matches_reference():
    """tests/golden/loss_zoo.npz (generated from utils/loss/contrastive.py, siglip_pairwise.py, multi_positive_infonce.py,
    weighted_siglip.py): auto-balance, entropy regulariser (active and inactive), SiglipPairwiseFeatureLoss variants, the
    abnormal-margin column offset, multi-positive InfoNCE reductions."""
    import os

    import numpy as np
    from oracle import losses as OL

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_zoo.npz"))
    T = lambda k: torch.from_numpy(g[k])  # noqa: E731
    mask, w = T("mask"), T("w")
    lt0 = torch.log(torch.tensor(0.07))

    def check(name, fn, with_bias=False):
        v, t = T("v").clone().requires_grad_(True), T("t").clone().requires_grad_(True)
        lt = lt0.clone().requires_grad_(True)
        bias = torch.tensor(-10.0, requires_grad=True) if with_bias else torch.tensor(0.0)
        loss = fn(v, t, lt, bias)
        loss.backward()
        assert abs(loss.item() - float(g[f"{name}_loss"])) < 1e-5 * max(1.0, abs(float(g[f"{name}_loss"]))), name
        for got, key in ((v.grad, "dv"), (t.grad, "dt")):
            ref = T(f"{name}_{key}")
            assert (got - ref).norm() <= 2e-4 * ref.norm().clamp_min(1e-12), (name, key)
        assert abs(lt.grad.item() - float(g[f"{name}_dlt"])) < 2e-4 * max(1e-3, abs(float(g[f"{name}_dlt"]))), name
        if with_bias:
            assert abs(bias.grad.item() - float(g[f"{name}_dbias"])) < 2e-4 * max(1e-6, abs(float(g[f"{name}_dbias"]))), name

    check("bce_auto", lambda v, t, lt, b: OL.siglip_bce_loss(v, t, lt, b, mask, w, auto_balance=True), True)
    check("bce_ent", lambda v, t, lt, b: OL.siglip_bce_loss(v, t, lt, b, mask, w, entropy_weight=0.1, min_entropy_threshold=3.0), True)
    check("bce_ent_off", lambda v, t, lt, b: OL.siglip_bce_loss(v, t, lt, b, mask, w, entropy_weight=0.1, min_entropy_threshold=0.01), True)
    pw = dict(clamp_temp=False, pos_thr=0.0)
    check("spf", lambda v, t, lt, b: OL.siglip_bce_loss(v, t, lt, b, mask, w, **pw))
    check("spf_noweights", lambda v, t, lt, b: OL.siglip_bce_loss(v, t, lt, b, mask, None, positive_weight=2.0, negative_weight=0.5, **pw))
    check("spf_auto", lambda v, t, lt, b: OL.siglip_bce_loss(v, t, lt, b, mask, w, auto_balance=True, **pw))
    check("spf_ent", lambda v, t, lt, b: OL.siglip_bce_loss(v, t, lt, b, mask, w, entropy_weight=0.2, min_entropy_threshold=3.0, **pw))
    check("margin_wce", lambda v, t, lt, b: OL.runner_multipositive_loss(v, t, lt, mask, w, weighted=True, col_margin=T("abn") * 0.35)[0])
    for name, kw, use_w in (("nce_mean", {}, True), ("nce_sum", {"reduction": "sum"}, True), ("nce_nopw", {}, False),
                            ("nce_imp", {"use_importance_weighting": True}, True),
                            ("nce_imp_sum", {"use_importance_weighting": True, "reduction": "sum"}, False)):
        z = T("nce_logits").clone().requires_grad_(True)
        loss = OL.multi_positive_infonce(z, T("nce_mask"), T("nce_w") if use_w else None, **kw)
        loss.backward()
        assert abs(loss.item() - float(g[f"{name}_loss"])) < 1e-5 * max(1.0, abs(float(g[f"{name}_loss"]))), name
        assert (z.grad - T(f"{name}_dz")).norm() <= 1e-4 * T(f"{name}_dz").norm(), name
    assert OL.multi_positive_infonce(T("nce_logits"), torch.zeros(8, 13)).item() == float(g["nce_empty_loss"]) == 0.0


def test_attention_pool_with_cls_oracle_matches_reference():
    """tests/golden/pools2.npz: models/attention_pool.py AttentionPoolWithCLS (nn.TransformerEncoder inside) with
    name-seeded weights; the oracle restates it with explicit ops under the same state_dict names."""
    import os

    import numpy as np
    from oracle.model import AttentionPoolWithCLSOracle, init_by_name

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pools2.npz"))
    m = init_by_name(AttentionPoolWithCLSOracle(512, 8)).eval()
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    with torch.no_grad():
        y = m(torch.from_numpy(g["apcls_x"]))
    assert (y - torch.from_numpy(g["apcls_y"])).abs().max() < 2e-5


def test_mil_oracle_matches_reference():
    """tests/golden/mil.npz (models/multi_instance_linear_probing.py, name-seeded weights): every pooling mode incl. the
    hierarchical 4-D paths, ragged instance mask."""
    import os

    import numpy as np
    from oracle.mil import MILOracle
    from oracle.model import init_by_name

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mil.npz"))
    x3, x4, mask = torch.from_numpy(g["x3"]), torch.from_numpy(g["x4"]), torch.from_numpy(g["mask"])
    from tests.golden.make_golden import MIL_CASES
    for name, kw, four in MIL_CASES:
        m = init_by_name(MILOracle(512, {"a": 3, "b": 1}, mode=kw["pooling_mode"], separate=kw.get("separate_video_attention", True),
                                   strategy=kw.get("normalization_strategy", "post_norm"))).eval()
        assert sorted(m.state_dict().keys()) == list(g[f"{name}_keys"]), name
        with torch.no_grad():
            y = m(x4 if four else x3, mask)
        for h in ("a", "b"):
            assert (y[h] - torch.from_numpy(g[f"{name}_{h}"])).abs().max() < 2e-5, (name, h)


def test_mil_view_embedding_oracle_matches_reference():
    """tests/golden/mil_view.npz: the view-angle embedding add (models/multi_instance_linear_probing.py:142-145,265-269) in
    front of the pooling rules, 3-D and hierarchical 4-D inputs, PAD ids included."""
    import os

    import numpy as np
    from oracle.mil import MILOracle
    from oracle.model import init_by_name
    from tests.golden.make_golden import MIL_VIEW_CASES

    g0 = np.load(os.path.join(os.path.dirname(__file__), "golden", "mil.npz"))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mil_view.npz"))
    x3, x4, mask = torch.from_numpy(g0["x3"]), torch.from_numpy(g0["x4"]), torch.from_numpy(g0["mask"])
    vid = torch.from_numpy(g["view_ids"])
    for name, kw, four in MIL_VIEW_CASES:
        m = init_by_name(MILOracle(512, {"a": 3, "b": 1}, mode=kw["pooling_mode"], num_view_classes=kw["num_view_classes"])).eval()
        assert sorted(m.state_dict().keys()) == list(g[f"{name}_keys"]), name
        with torch.no_grad():
            y = m(x4 if four else x3, mask, vid)
        for h in ("a", "b"):
            assert (y[h] - torch.from_numpy(g[f"{name}_{h}"])).abs().max() < 2e-5, (name, h)


def test_masked_pools_oracle_matches_reference():
    """tests/golden/pools_masked.npz: key_padding_mask variants of AttentionPool / AttentionPoolWithCLS
    (models/attention_pool.py:61-101,162-197) and the masked EnhancedVideoAggregator (models/video_aggregator.py:105-159)."""
    from oracle.model import AggregatorOracle, AttentionPoolOracle, AttentionPoolWithCLSOracle, init_by_name

    g = _load("pools_masked.npz")
    x, mask = _t(g["x"]), torch.from_numpy(g["mask"])
    with torch.no_grad():
        ap = init_by_name(AttentionPoolOracle(512, 8))
        assert np.allclose(ap(x, mask).numpy(), g["ap_y"], atol=2e-5)
        apc = init_by_name(AttentionPoolWithCLSOracle(512, 8)).eval()
        assert np.allclose(apc(x, mask).numpy(), g["apcls_y"], atol=5e-5)
        ag = init_by_name(AggregatorOracle(512, 4, 2))
        for n in (4, 8):
            y = ag(_t(g[f"ag_x{n}"]), torch.from_numpy(g[f"ag_m{n}"]))
            assert np.allclose(y.numpy(), g[f"ag_y{n}"], atol=2e-5), n


@pytest.mark.parametrize("stride", [(1, 1, 1), (1, 2, 2), (1, 4, 4), (1, 8, 8), (2, 2, 2)])
def test_oracle_depthwise_pool_equals_aten_conv3d(stride):
    """oracle/mvit.py restates the q / k / v pools' nn.Conv3d(groups=C, k=3, p=1, bias=False) as 27 strided-slice multiply-adds
    with a hand-written backward (ATen's CPU depthwise paths took 40 % of the oracle's time): forward, input gradient and weight
    gradient against F.conv3d in fp64, and the Pool module's two code paths against each other."""
    import torch.nn.functional as F

    from oracle.mvit import Pool, depthwise_conv3d_taps

    g = torch.Generator().manual_seed(sum(stride))
    B, Hh, C, thw = 2, 3, 8, (4, 16, 12)
    x = torch.randn(B, Hh, *thw, C, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(C, 1, 3, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    y = depthwise_conv3d_taps(x, w, stride)
    xr, wr = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    yr = F.conv3d(xr.permute(0, 1, 5, 2, 3, 4).reshape(B * Hh, C, *thw), wr, None, stride, (1, 1, 1), groups=C)
    yr = yr.reshape(B, Hh, C, *yr.shape[2:]).permute(0, 1, 3, 4, 5, 2)
    assert y.shape == yr.shape and (y - yr).abs().max() < 1e-12
    up = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(up)
    yr.backward(up)
    assert (x.grad - xr.grad).abs().max() < 1e-12 and (w.grad - wr.grad).abs().max() < 1e-11
    # the module: depthwise conv -> tap path; anything else (the skip path's MaxPool3d) -> the module's own forward
    conv = torch.nn.Conv3d(C, C, (3, 3, 3), stride=stride, padding=(1, 1, 1), groups=C, bias=False).double()
    tok = torch.randn(B, Hh, 1 + thw[0] * thw[1] * thw[2], C, generator=g, dtype=torch.float64)
    out, out_thw = Pool(conv, torch.nn.LayerNorm(C).double())(tok, thw)
    ref = conv(tok[:, :, 1:].transpose(2, 3).reshape(B * Hh, C, *thw))
    assert tuple(ref.shape[2:]) == out_thw
    ref = torch.cat((tok[:, :, :1], ref.reshape(B, Hh, C, -1).transpose(2, 3)), dim=2)
    assert (out - torch.nn.functional.layer_norm(ref, (C,))).abs().max() < 1e-10


@pytest.mark.parametrize("B", [8, 24, 16])
@pytest.mark.parametrize("name", ["contrastive_ddp", "siglip_ddp"])
def test_ddp_loss_classes_match_reference(B, name):
    """ContrastiveLossDDP / SiglipLossDDP (utils/loss/losses.py:104-158, 213-276) imported at world size 1 by make_golden.py.
    SiglipLossDDP's fp16 wire format shows in the gradients: they are fp16-representable numbers."""
    from oracle.losses import contrastive_loss_ddp, siglip_loss_ddp

    g = _load("losses_ddp.npz")
    v = _t(g[f"v{B}"]).requires_grad_(True)
    t = _t(g[f"t{B}"]).requires_grad_(True)
    lt = torch.log(torch.tensor(0.07)).requires_grad_(True)
    loss = (contrastive_loss_ddp if name == "contrastive_ddp" else siglip_loss_ddp)(v, t, lt)
    loss.backward()
    assert abs(loss.item() - float(g[f"{name}{B}_loss"])) < 1e-6
    for got, key in ((v.grad, "dv"), (t.grad, "dt")):
        ref = g[f"{name}{B}_{key}"]
        assert np.allclose(got.numpy(), ref, atol=1e-7, rtol=1e-6)
        if name == "siglip_ddp":
            assert np.array_equal(ref, ref.astype(np.float16).astype(np.float32))  # the reference's gradient came back as halves
    assert abs(lt.grad.item() - float(g[f"{name}{B}_dlt"])) < 1e-5
