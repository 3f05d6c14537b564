"""The drop-in boundary exercised the way the reference's project code uses it
(projects/contrastive_pretraining_project.py:75-237 with config/clip/base_config.yaml's defaults: dropout 0.158,
num_heads 8, attention_pool_dropout 0.1, HF BERT's default hidden / attention-probability dropout 0.1), plus the
attention-probability dropout inside the flash kernels that those defaults need (nn.MultiheadAttention(dropout=) in
models/attention_pool.py:45-50 and models/video_aggregator.py:16-21, BertSelfAttention in models/text_encoder.py:52)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask_from_kernel(K, B, H, Nq, Nk, D, p, seed, step, dtype):
    """The keep mask the kernels draw for (seed, step): run the forward with uniform scores and V = identity rows, so
    O[q] = dropout(P)[q, :] directly (needs Nk <= D)."""
    assert Nk <= D
    q = torch.zeros(B, Nq, H, D, device="cuda", dtype=dtype)
    k = torch.zeros(B, Nk, H, D, device="cuda", dtype=dtype)
    v = torch.zeros(B, Nk, H, D, device="cuda", dtype=dtype)
    v[:, torch.arange(Nk), :, torch.arange(Nk)] = 1.0
    o, _ = K.attn_fwd(q, k, v, scale=1.0, drop_p=p, drop_seed=seed, drop_step=step)
    pm = o[..., :Nk].float() * Nk  # P = 1/Nk everywhere -> entries are 0 or 1/(1-p)
    return pm.permute(0, 2, 1, 3).contiguous()  # [B, H, Nq, Nk] multipliers


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("D,Nq,Nk", [(64, 37, 50), (128, 5, 100), (64, 1, 64)])
def test_attention_dropout_forward_backward_vs_autograd(D, Nq, Nk, dtype, tol):
    from deepcoro_clip_amd import kernels as K

    B, H, p, seed = 2, 3, 0.3, 12345
    step = torch.tensor([7], device="cuda", dtype=torch.int64)
    mult = _mask_from_kernel(K, B, H, Nq, Nk, D, p, seed, step, dtype)
    keep = (mult > 0).float()
    assert torch.allclose(mult[mult > 0], torch.full_like(mult[mult > 0], 1.0 / (1.0 - p)), rtol=1e-2)
    rate = keep.mean().item()
    n = keep.numel()
    assert abs(rate - (1 - p)) < 5 * math.sqrt(p * (1 - p) / n) + 1e-3, rate
    # a different step counter or seed draws a different mask; the same pair reproduces it
    assert torch.equal(mult, _mask_from_kernel(K, B, H, Nq, Nk, D, p, seed, step, dtype))
    assert not torch.equal(mult, _mask_from_kernel(K, B, H, Nq, Nk, D, p, seed, step + 1, dtype))
    assert not torch.equal(mult, _mask_from_kernel(K, B, H, Nq, Nk, D, p, seed + 1, step, dtype))

    g = torch.Generator().manual_seed(0)
    q0, k0, v0 = (torch.randn(B, n_, H, D, generator=g) for n_ in (Nq, Nk, Nk))
    kb = torch.zeros(B, Nk)
    kb[1, Nk - 7:] = -1e30  # padded keys on one sample
    scale = 1.0 / math.sqrt(D)
    q, k, v = (t.to(dtype).cuda() for t in (q0, k0, v0))
    o, lse = K.attn_fwd(q, k, v, scale=scale, kbias=kb.cuda(), drop_p=p, drop_seed=seed, drop_step=step)
    do0 = torch.randn(B, Nq, H, D, generator=g)
    dq, dk, dv = K.attn_bwd(q, k, v, o, do0.to(dtype).cuda(), lse, scale=scale, kbias=kb.cuda(), drop_p=p, drop_seed=seed,
                            drop_step=step)
    # reference: plain autograd with the extracted mask, on the rounded inputs
    qr, kr, vr = (t.float().cpu().double().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * scale + kb.double()[:, None, None, :]
    pr = s.softmax(-1) * mult.cpu().double()
    oref = torch.einsum("bhqk,bkhd->bqhd", pr, vr)
    oref.backward(do0.to(dtype).double())
    rel = lambda a, b: ((a.float().cpu().double() - b).norm() / b.norm().clamp_min(1e-12)).item()  # noqa: E731
    assert rel(o, oref.detach()) < tol, rel(o, oref.detach())
    assert rel(dq, qr.grad) < tol * 2 and rel(dk, kr.grad) < tol * 2 and rel(dv, vr.grad) < tol * 2, \
        (rel(dq, qr.grad), rel(dk, kr.grad), rel(dv, vr.grad))


def _project_style_setup(pooling="attention", videos_per_study=2, T=4, S=64, B=2, loss_name="contrastive"):
    """Encoders, optimizer and runner built with the reference project's calls and config defaults."""
    from deepcoro_clip_amd.losses import Loss
    from deepcoro_clip_amd.registry import LossRegistry, ModelRegistry, RunnerRegistry, register_submodules

    register_submodules("deepcoro_clip_amd")
    cfg = type("Cfg", (), {})()
    cfg.__dict__.update(model_name="mvit", frames=T, pretrained=False, video_freeze_ratio=1.0, dropout=0.15848373640344304,
                        num_heads=8, aggregator_depth=2, video_pooling_mode=pooling, attention_pool_heads=8,
                        attention_pool_dropout=0.1, text_freeze_ratio=0.0, lr=1e-4, video_weight_decay=1e-6,
                        text_weight_decay=1e-6, temperature=0.07, optimizer="AdamW", recall_k=[1, 5], ndcg_k=[5],
                        device="cuda:0", world_size=1, gradient_accumulation_steps=1, max_grad_norm=5.0, is_ref_device=True,
                        loss_name=loss_name, scheduler_name="", use_amp=True)
    video_encoder = ModelRegistry.get(name="video_encoder")(
        backbone=cfg.model_name, input_channels=3, num_frames=cfg.frames, pretrained=cfg.pretrained, output_dim=512,
        freeze_ratio=cfg.video_freeze_ratio, dropout=cfg.dropout, num_heads=cfg.num_heads, aggregator_depth=cfg.aggregator_depth,
        token_pooling_mode=cfg.video_pooling_mode, attention_pool_heads=cfg.attention_pool_heads,
        attention_pool_dropout=cfg.attention_pool_dropout, use_cls_token=False, use_rope=False, rope_base=10000.0,
        rope_temporal_scale=1.0, rope_normalize_mode="separate",
        spatial_size=(S, S))  # the one extra kwarg: test-sized clips (the reference hard-wires 224 x 224)
    video_encoder = video_encoder.to(cfg.device).float()
    # TextEncoder(freeze_ratio=, dropout=) exactly as the project does: default model name (no hub access here =>
    # randomly initialised BERT-base shape), HF default hidden / attention dropout 0.1
    text_encoder = ModelRegistry.get(name="text_encoder")(freeze_ratio=cfg.text_freeze_ratio, dropout=cfg.dropout)
    text_encoder = text_encoder.to(cfg.device).float()
    log_temperature = torch.nn.Parameter(torch.log(torch.tensor([cfg.temperature], dtype=torch.float32, device=cfg.device)))
    param_groups = [
        {"params": video_encoder.model.parameters(), "lr": cfg.lr, "name": "video_backbone", "weight_decay": cfg.video_weight_decay},
        {"params": text_encoder.parameters(), "lr": 0.00002, "name": "text_encoder", "weight_decay": cfg.text_weight_decay},
        {"params": [log_temperature], "lr": cfg.lr, "name": "temperature"}]
    idx = 1
    ap = [p for p in getattr(video_encoder, "attention_pool", None).parameters() if p.requires_grad] if video_encoder.attention_pool is not None else []
    if ap:
        param_groups.insert(idx, {"params": ap, "lr": cfg.lr * 2.0, "name": "video_attention_pool", "weight_decay": cfg.video_weight_decay})
        idx += 1
    agg = [p for p in video_encoder.aggregator.parameters() if p.requires_grad]
    if agg:
        param_groups.insert(idx, {"params": agg, "lr": cfg.lr * 2.0, "name": "video_aggregator", "weight_decay": cfg.video_weight_decay})
    optimizer = getattr(torch.optim, cfg.optimizer)(param_groups, lr=cfg.lr)
    loss_fn = Loss(loss_type=LossRegistry.get(cfg.loss_name)())
    runner = RunnerRegistry.get("DeepCORO_clip")(config=cfg, wandb_wrapper=None, train_loader=None, val_loader=None,
                                                  video_encoder=video_encoder, text_encoder=text_encoder, optimizer=optimizer,
                                                  scaler=None, log_temp=log_temperature, lr_scheduler=None, loss_fn=loss_fn,
                                                  output_dir=None)
    g = torch.Generator().manual_seed(11)
    shape = (B, videos_per_study, T, S, S, 3) if videos_per_study else (B, T, S, S, 3)
    batch = dict(videos=torch.randn(shape, generator=g).cuda(), input_ids=torch.randint(0, 30522, (B, 24), generator=g).cuda(),
                 attention_mask=torch.ones(B, 24, dtype=torch.int64).cuda())
    return runner, video_encoder, text_encoder, batch


@pytest.mark.parametrize("pooling,N", [("attention", 2), ("mean", 0)])
def test_project_default_construction_trains(pooling, N):
    """No NotImplementedError is reachable from the project's default kwargs: .train(), three optimizer steps with every
    dropout active (proj 0.158, BERT hidden + attention 0.1, AttentionPool attention 0.1, aggregator MHA + MLP 0.158,
    stochastic depth 0.2), finite losses, parameters move, masks change from step to step."""
    runner, ve, te, batch = _project_style_setup(pooling, N)
    ve.train()
    te.train()
    assert te.bert.config.attention_probs_dropout_prob == 0.1 and te.bert.config.hidden_dropout_prob == 0.1
    if pooling == "attention":
        assert ve.attention_pool.attn_dropout == 0.1
    assert ve.aggregator.blocks[0].attn.dropout == pytest.approx(0.15848373640344304)
    w0 = te.bert.encoder.layer[0].attention.self.query.weight.detach().clone()
    losses = []
    for _ in range(3):
        m, emb = runner._train_step(**batch)
        losses.append(m["loss"])
        assert math.isfinite(m["loss"]) and emb["video_embeddings"].shape == (2, 512)
    assert not torch.equal(te.bert.encoder.layer[0].attention.self.query.weight.detach(), w0)
    # same inputs, training mode: the dropout masks differ between calls (device step counter) ...
    with torch.no_grad():
        a = te(batch["input_ids"], batch["attention_mask"])
        b = te(batch["input_ids"], batch["attention_mask"])
        assert not torch.equal(a, b)
        # ... and vanish in eval mode
        ve.eval()
        te.eval()
        assert torch.equal(te(batch["input_ids"], batch["attention_mask"]), te(batch["input_ids"], batch["attention_mask"]))
        assert torch.equal(ve(batch["videos"]), ve(batch["videos"]))


def test_project_default_construction_eval_equals_oracle():
    """Eval mode of the project-style encoders (dropout off) against the CPU oracle on the same weights."""
    from oracle.model import TextEncoderOracle, VideoEncoderOracle

    runner, ve, te, batch = _project_style_setup("attention", 2)
    vo = VideoEncoderOracle(4, (64, 64), num_heads=8, token_pooling_mode="attention")
    to = TextEncoderOracle()
    vo.load_state_dict(ve.state_dict())
    to.load_state_dict(te.state_dict())
    ve.eval()
    te.eval()
    with torch.no_grad():
        v = ve(batch["videos"]).float().cpu()
        t = te(batch["input_ids"], batch["attention_mask"]).float().cpu()
        vr = vo(batch["videos"].cpu())
        tr = to(batch["input_ids"].cpu(), batch["attention_mask"].cpu())
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()  # noqa: E731
    assert rel(v, vr) < 5e-2 and rel(t, tr) < 5e-2, (rel(v, vr), rel(t, tr))  # bf16 compute (the encoders' default)


def test_attention_pool_dropout_statistics():
    """AttentionPool in training mode: E[output before the LayerNorm] is unbiased — checked through the module by
    averaging many draws against the eval-mode output (loose statistical bound)."""
    from deepcoro_clip_amd.attention_pool import AttentionPool

    torch.manual_seed(0)
    pool = AttentionPool(512, num_heads=8, dropout=0.5).cuda()
    pool.norm = torch.nn.LayerNorm(512, elementwise_affine=True).cuda()
    x = torch.randn(4, 50, 512, device="cuda")
    from deepcoro_clip_amd import functional as Fn
    a = pool.attn

    def pre_norm(training):
        q = pool.query.expand(4, 1, 512)
        return Fn.multihead_attention(q, x, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, 8,
                                      self_attn=False, dropout_p=0.5, training=training)

    with torch.no_grad():
        ref = pre_norm(False)
        acc = torch.zeros_like(ref)
        n = 400
        draws = []
        for _ in range(n):
            d = pre_norm(True)
            acc += d
            draws.append(d)
        mean = acc / n
    assert not torch.equal(draws[0], draws[1])
    spread = torch.stack(draws).std(0).mean().item()
    assert (mean - ref).abs().mean().item() < 4 * spread / math.sqrt(n) + 1e-4
