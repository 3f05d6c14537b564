"""CPU: C-ABI library loads and exports every symbol of include/dcc.h; host logic (registry, geometry, freezing,
state_dict names, optimizer adoption, fail-loud without GPU)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from deepcoro_clip_amd import _lib as L

    text = open(os.path.join(ROOT, "include", "dcc.h")).read()
    declared = set(re.findall(r"\bint\s+(dcc_\w+)\s*\(", text))
    assert len(declared) >= 40 and declared == set(L.SIGNATURES)
    handle = L.lib()
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.dcc_version() == 0


def test_product_fails_loudly_on_cpu_tensors():
    from deepcoro_clip_amd import _lib as L
    from deepcoro_clip_amd import kernels as K
    from deepcoro_clip_amd.losses import ContrastiveLoss

    with pytest.raises(L.DccError):
        K.gemm(torch.randn(8, 8), torch.randn(8, 8))
    with pytest.raises(L.DccError):
        ContrastiveLoss()(torch.randn(4, 512), torch.randn(4, 512), torch.log(torch.tensor(0.07)))


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "deepcoro_clip_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_registry_names():
    from deepcoro_clip_amd.registry import LossRegistry, ModelRegistry, RunnerRegistry, register_submodules

    register_submodules("deepcoro_clip_amd")
    for n in ("contrastive", "contrastive_ddp", "clip", "siglip", "siglip_ddp", "InfoNCE", "siglip_pairwise", "siglip2_bce",
              "siglip2_bce_ddp", "siglip2_multi_positive"):
        assert LossRegistry.get(n) is not None
        LossRegistry.get(n)()  # no-arg constructible (projects/contrastive_pretraining_project.py:204-206)
    assert ModelRegistry.get("video_encoder").__name__ == "VideoEncoder"
    assert ModelRegistry.get("text_encoder").__name__ == "TextEncoder"
    assert RunnerRegistry.get("DeepCORO_clip") is RunnerRegistry.get("DeepCORO_clip_test")
    with pytest.raises(ValueError):
        LossRegistry.get("nope")


def test_block_plan_geometry():
    from deepcoro_clip_amd.mvit import block_plan

    plan = block_plan(16, (224, 224))
    assert plan[0]["thw"] == (8, 56, 56) and plan[0]["k_thw"] == (8, 7, 7) and plan[0]["rel_sp"] == 111 and plan[0]["rel_t"] == 15
    assert [p["q_thw"] for p in plan][-3:] == [(8, 14, 14), (8, 7, 7), (8, 7, 7)]
    tokens = [1 + p["thw"][0] * p["thw"][1] * p["thw"][2] for p in plan]
    assert tokens[0] == 25089 and tokens[2] == 6273 and tokens[4] == 1569 and tokens[15] == 393
    assert block_plan(4, (112, 112))[15]["q_thw"] == (2, 4, 4)


def test_forward_macs_known_answers():
    """SURVEY §8d / BASELINE.md §2 known answers: torchvision publishes 64.224 GMACs for mvit_v2_s at 16 x 224 x 224;
    the per-piece split and the two size-parameterised totals (4 x 112^2: 6.60 GFLOP, 32 x 224^2: 314.80 GFLOP)."""
    from deepcoro_clip_amd.mvit import block_plan, forward_macs

    m = forward_macs(block_plan(16, (224, 224)))
    assert abs(m["total"] / 1e9 - 64.224) < 1e-3, m["total"]
    gflop = {k: round(2 * v / 1e9, 2) for k, v in m.items()}
    assert gflop == {"conv_proj": 2.12, "qkv": 26.37, "pool": 1.09, "attn": 28.86, "relpos": 0.59, "proj": 7.40,
                     "skip": 2.78, "mlp": 59.24, "total": 128.45}, gflop
    assert round(2 * forward_macs(block_plan(4, (112, 112)))["total"] / 1e9, 2) == 6.60
    assert round(2 * forward_macs(block_plan(32, (224, 224)))["total"] / 1e9, 2) == 314.80
    # backbone parameter count with the Identity head (torchvision: 34 537 744 with the 400-way head)
    from deepcoro_clip_amd.mvit import HipMViT
    n = sum(p.numel() for p in HipMViT().parameters())
    assert n == 34_230_144 and n + 768 * 400 + 400 == 34_537_744


def test_video_encoder_contract_and_freezing():
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle

    with pytest.raises(ValueError):
        VideoEncoder(backbone="resnet", pretrained=False)
    with pytest.raises(ValueError):
        VideoEncoder(pretrained=False, output_dim=510, num_heads=4, num_frames=4, spatial_size=(64, 64))
    v = VideoEncoder(pretrained=False, num_frames=4, spatial_size=(64, 64), freeze_ratio=0.25, token_pooling_mode="attention")
    names = [n for n, _ in v.model.named_parameters()]
    flags = [p.requires_grad for _, p in v.model.named_parameters()]
    k = int(0.25 * len(names))
    assert flags == [False] * (len(names) - k) + [True] * k  # trainable fraction counted from the top (quirk 1)
    v.update_freeze_ratio(1.0)
    assert all(p.requires_grad for p in v.model.parameters())
    assert v.embedding_dim == 512 and v.attention_pool is not None and hasattr(v, "aggregator") and hasattr(v, "proj")
    ref_keys = list(VideoEncoderOracle(4, (64, 64), token_pooling_mode="attention").state_dict().keys())
    assert list(v.state_dict().keys()) == ref_keys
    with pytest.raises(Exception):  # CPU tensors are refused: no fallback
        v(torch.randn(1, 4, 64, 64, 3))


def test_text_encoder_contract():
    from deepcoro_clip_amd.text_encoder import TextEncoder

    t = TextEncoder(freeze_ratio=0.5, bert_config=dict(num_hidden_layers=2, vocab_size=100))
    ps = list(t.bert.named_parameters())
    k = int(0.5 * len(ps))
    assert [p.requires_grad for _, p in ps] == [False] * (len(ps) - k) + [True] * k
    t.update_freeze_ratio(0.0)
    assert all(p.requires_grad for p in t.bert.parameters())
    assert any(n == "bert.encoder.layer.0.attention.self.query.weight" for n, _ in t.named_parameters())
    full = TextEncoder(freeze_ratio=0.0)
    assert sum(p.numel() for p in full.bert.parameters()) == 108_891_648 and len(list(full.bert.parameters())) == 197


def test_relpos_index_matches_oracle():
    from deepcoro_clip_amd.functional import relpos_index

    for q, k in ((56, 7), (28, 14), (7, 7), (14, 7), (8, 8), (2, 2)):
        qr, kr = max(k / q, 1.0), max(q / k, 1.0)
        ref = (torch.arange(q)[:, None] * qr - (torch.arange(k)[None, :] + (1.0 - k)) * kr).long()
        assert torch.equal(relpos_index(q, k).long(), ref)
        assert ref.min() >= 0 and ref.max() <= 2 * max(q, k) - 2


def test_runner_validates_config():
    from deepcoro_clip_amd.runner import VideoContrastiveLearningRunner

    class C:
        recall_k, ndcg_k, device, world_size, temperature = 5, [5], "cpu", 1, 0.07

    with pytest.raises(ValueError):
        VideoContrastiveLearningRunner(config=C(), log_temp=torch.tensor(0.0))
    r = VideoContrastiveLearningRunner.__new__(VideoContrastiveLearningRunner)
    with pytest.raises(NotImplementedError):
        r.validate()


def _stable_ranks(sim, gt_sets):
    """1-based positions of the ground-truth candidates in a stable descending argsort (what the GPU kernel returns)."""
    import numpy as np

    order = np.argsort(-np.asarray(sim, dtype=np.float64), axis=1, kind="stable")
    out = []
    for i, s in enumerate(gt_sets):
        inv = np.empty(order.shape[1], dtype=np.int64)
        inv[order[i]] = np.arange(order.shape[1])
        out.append(sorted(int(inv[j]) + 1 for j in s if 0 <= j < order.shape[1]))
    return out


def test_retrieval_metrics_host_part_matches_golden():
    """The host half of the retrieval metrics (arithmetic on ground-truth ranks) against values produced by the
    reference's utils/retrieval_metrics.py; the ranks themselves come from the GPU kernel (tests/test_retrieval_gpu.py)."""
    import numpy as np

    from deepcoro_clip_amd import retrieval_metrics as RM

    g = np.load(os.path.join(ROOT, "tests", "golden", "retrieval.npz"))
    r = _stable_ranks(g["sim"], [{i} for i in range(100)])
    assert [RM.recall_from_ranks(r, k) for k in (1, 5, 10)] == pytest.approx(list(g["recall"]))
    assert RM.mrr_from_ranks(r) == pytest.approx(float(g["mrr"]), abs=1e-9)
    assert float(np.mean(RM.ap_from_ranks(r))) == pytest.approx(float(g["map"]), abs=1e-7)
    assert float(np.mean(RM.ndcg_from_ranks(r, [1] * 100, 5))) == pytest.approx(float(g["ndcg"][0]), abs=1e-7)
    assert float(np.mean(RM.ndcg_from_ranks(r, [1] * 100, 10))) == pytest.approx(float(g["ndcg"][1]), abs=1e-7)
    gt2 = [{int(x) for x in row if x >= 0} for row in g["gt2"]]
    r2 = _stable_ranks(g["sim2"], gt2)
    assert float(np.mean(RM.ap_from_ranks(r2))) == pytest.approx(float(g["map2"]), abs=1e-7)
    assert RM.mrr_from_ranks(r2) == pytest.approx(float(g["mrr2"]), abs=1e-9)
    assert float(np.mean(RM.ndcg_from_ranks(r2, [len(x) for x in gt2], 3))) == pytest.approx(float(g["ndcg2"][0]), abs=1e-7)
    assert [RM.recall_from_ranks(r2, k) for k in (1, 3, 5)] == pytest.approx(list(g["recall2"]))
    with pytest.raises(Exception):  # the full functions need the GPU rank kernel: no CPU path
        RM.compute_recall_at_k(torch.from_numpy(g["sim"]), torch.arange(100), [1])


def test_encoder_checkpoint_round_trip(tmp_path):
    """`encoder_path` loading (models/video_encoder.py:390-440): checkpoints wrapped in `model_state_dict` / `state_dict`,
    `module.` / `model.` prefixes stripped, decoder / mvm / mask_token / head tensors skipped, strict=False semantics;
    and the runner's checkpoint file is loadable back into fresh encoders (reference layout, :1071-1115)."""
    import torch

    from deepcoro_clip_amd.video_encoder import VideoEncoder

    src = VideoEncoder(num_frames=4, spatial_size=(64, 64), pretrained=False, freeze_ratio=1.0, dropout=0.0)
    sd = {}
    for k, v in src.model.state_dict().items():
        sd["module.model." + k] = v.clone() + 0.25           # as saved from a DDP-wrapped pre-training model
    sd["module.decoder.weight"] = torch.zeros(3)             # pre-training-only tensors must be ignored
    sd["module.mask_token"] = torch.zeros(1, 1, 96)
    sd["module.model.head.1.weight"] = torch.zeros(400, 768)
    for wrap in ("model_state_dict", "state_dict", None):
        path = tmp_path / f"enc_{wrap}.pt"
        torch.save({wrap: sd, "epoch": 3} if wrap else sd, path)
        dst = VideoEncoder(num_frames=4, spatial_size=(64, 64), pretrained=False, freeze_ratio=1.0, dropout=0.0, encoder_path=str(path))
        for (k, a), (_, b) in zip(dst.model.state_dict().items(), src.model.state_dict().items()):
            assert torch.equal(a, b + 0.25), (wrap, k)
    # a missing file is reported, not fatal (reference prints and continues)
    VideoEncoder(num_frames=4, spatial_size=(64, 64), pretrained=False, encoder_path=str(tmp_path / "nope.pt"))
    # the full-encoder state dict (what _save_checkpoint writes) round-trips, proj / aggregator / attention_pool included
    full = VideoEncoder(num_frames=4, spatial_size=(64, 64), pretrained=False, token_pooling_mode="cls_token", dropout=0.0)
    ck = tmp_path / "checkpoint.pt"
    torch.save({"video_encoder": full.state_dict(), "epoch": 0, "best_val_loss": 1.0, "best_epoch": 0,
                "highest_alignment_score": 0.5, "highest_alignment_epoch": 0}, ck)
    other = VideoEncoder(num_frames=4, spatial_size=(64, 64), pretrained=False, token_pooling_mode="cls_token", dropout=0.0)
    missing, unexpected = other.load_state_dict(torch.load(ck)["video_encoder"], strict=True)
    assert not missing and not unexpected
    assert any(k.startswith("attention_pool.transformer.layers.0.self_attn.in_proj_weight") for k in full.state_dict())


def test_collate_layouts_and_text_bucketing():
    """custom_collate_fn (dataloaders/video_clip_dataset.py:843-969): single-positive and multi-positive batch dicts, uint8
    frames kept as bytes, right-padding trimmed to the batch's longest text in buckets of 16."""
    import numpy as np
    import torch

    from deepcoro_clip_amd.collate import custom_collate_fn, trim_padding

    def enc(n, S=64):
        ids = torch.zeros(S, dtype=torch.int64)
        ids[:n] = torch.arange(1, n + 1)
        m = torch.zeros(S, dtype=torch.int64)
        m[:n] = 1
        return {"input_ids": ids, "attention_mask": m}

    vids = [np.random.default_rng(i).standard_normal((4, 8, 8, 3)).astype(np.float64) for i in range(3)]
    single = [(vids[i], enc(5 + 9 * i), f"p{i}", f"report {i}") for i in range(3)]
    out = custom_collate_fn(single)
    assert set(out) == {"videos", "encoded_texts", "paths", "reports"}
    assert out["videos"].shape == (3, 4, 8, 8, 3) and out["videos"].dtype == torch.float32
    assert out["encoded_texts"]["input_ids"].shape == (3, 64) and out["paths"] == ["p0", "p1", "p2"]
    trimmed = custom_collate_fn(single, trim_text_padding=True)
    assert trimmed["encoded_texts"]["input_ids"].shape == (3, 32)  # longest = 23 tokens -> bucket of 32
    assert torch.equal(trimmed["encoded_texts"]["attention_mask"].sum(1), out["encoded_texts"]["attention_mask"].sum(1))
    ids, m = trim_padding(out["encoded_texts"]["input_ids"], out["encoded_texts"]["attention_mask"], bucket=8)
    assert ids.shape == (3, 24)
    # uint8 frames stay bytes (normalised on the GPU by the runner), multi-video studies keep their [N, T, H, W, C] block
    u8 = [(np.zeros((2, 4, 8, 8, 3), np.uint8), enc(3), "p", "r")] * 2
    assert custom_collate_fn(u8)["videos"].dtype == torch.uint8 and custom_collate_fn(u8)["videos"].shape == (2, 2, 4, 8, 8, 3)
    # no texts
    assert custom_collate_fn([(vids[0], None, "p", "r")])["encoded_texts"] is None
    # multi-positive: unique prompts become columns, mask / weights rows
    def entry(tid, n, ew=1.0, sw=1.0, abn=False):
        return {"text_id": tid, "encoding": enc(n), "prompt_text": f"prompt {tid}", "prompt_type": "t", "edge_weight": ew,
                "soft_weight": sw, "is_abnormal": abn, "tags": None}

    batch = [(vids[0], {"positive_entries": [entry("a", 4, 2.0, 0.5), entry("b", 6)]}, "p0", "raw0"),
             (vids[1], {"positive_entries": [entry("b", 6), entry("c", 9, 3.0, 1.0, True)]}, "p1", "raw1"),
             (vids[2], {"positive_entries": []}, "p2", "raw2")]
    mp = custom_collate_fn(batch, trim_text_padding=True)
    assert mp["encoded_texts"]["input_ids"].shape == (3, 16) and mp["text_ids"] == ["a", "b", "c"]
    assert mp["positive_mask"].tolist() == [[1, 1, 0], [0, 1, 1], [0, 0, 0]]
    assert mp["positive_weights"].tolist() == [[1.0, 1.0, 0.0], [0.0, 1.0, 3.0], [0.0, 0.0, 0.0]]
    assert [m_["is_abnormal"] for m_ in mp["text_metadata"]] == [False, False, True]
    assert mp["reports"] == ["prompt a", "prompt b", "raw2"]
    assert set(mp) == {"videos", "encoded_texts", "positive_mask", "positive_weights", "text_ids", "text_metadata", "paths", "reports"}


def test_frame_index_selection_matches_reference_rule():
    """utils/video.py:243-251: fewer frames than expected -> repeat the last one; more -> torch.linspace(0, t-1, n).long()."""
    from deepcoro_clip_amd.video import frame_indices
    from oracle.video import load_video_frames
    import numpy as np

    assert frame_indices(3, 5) == [0, 1, 2, 2, 2]
    assert frame_indices(10, 4) == [0, 3, 6, 9]
    assert frame_indices(33, 16) == torch.linspace(0, 32, 16).long().tolist()
    # the oracle applies the same rule: frame k of the output is frame idx[k] of the input (no resize, no normalisation)
    clip = (np.arange(7, dtype=np.uint8)[:, None, None, None] * np.ones((1, 2, 2, 3), np.uint8))
    out = load_video_frames(clip, n_frames=4, resize=None, normalize=False)
    assert out[:, 0, 0, 0].tolist() == [float(i) for i in frame_indices(7, 4)]
