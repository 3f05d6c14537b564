"""Oracle parity at the BENCHMARK geometries (BASELINE.json configs[1]/[2]/[4]): 16 x 224 x 224 and 32 x 224 x 224 clips,
MViT-v2-S with 393 / 785 tokens per clip, the 12-layer BERT-base text tower — forward AND every parameter gradient of
the HIP path against the CPU oracle in fp64 on the same name-seeded weights and seeded inputs.  These are the shapes
bench.py spends its time in (56/28/14/7 pooling grids, Nq = 25 089 attention, M = 25 089-row GEMMs).

The fp64 oracle is evaluated once per configuration (module-level cache) and shared by the fp32 / bf16 variants.
Tolerances: fp32 1e-4-class on outputs (north_star), bf16 1e-2-class; gradients as relative Frobenius error per tensor.
"""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

_ORACLE_CACHE: dict = {}

# Parameters whose gradient is ANALYTICALLY zero without RoPE: the bias of the key pool's LayerNorm is added to every
# key of a softmax row alike (class token included: torchvision's Pool normalises after re-attaching it), which shifts
# all logits of the row by q.beta and leaves the softmax unchanged; rel-pos and the residual pooling only see q.
ZERO_GRAD_KEYS = ("pool_k.norm_act.0.bias",)


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-6)).item()


def _video(B, N, T, S, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((B, N, T, S, S, 3) if N else (B, T, S, S, 3), generator=g)


def _oracle_run(key, build, x, gseed=1, coherent=False):
    """fp64 oracle forward + backward with a fixed upstream gradient; cached per configuration.
    coherent: token outputs [B, tokens, D] get an upstream gradient that is mostly shared by all tokens of a sample (what
    any pooling head sends back) — a white-noise gradient over 1 500 tokens makes the parameter gradients sums of
    incoherent terms whose bf16 rounding noise is as large as the signal, which tests nothing."""
    if key not in _ORACLE_CACHE:
        from oracle.model import init_by_name

        orc = init_by_name(build()).double()
        ref = orc(x.double())
        gen = torch.Generator().manual_seed(gseed)
        g = torch.randn(ref.shape, generator=gen).double()
        if coherent and ref.dim() == 3:
            g = torch.randn(ref.shape[0], 1, ref.shape[2], generator=gen).double() + 0.25 * g
        ref.backward(g)
        grads = {n: (None if p.grad is None else p.grad.clone()) for n, p in orc.named_parameters()}
        state = {k: v.float() for k, v in orc.state_dict().items()}
        _ORACLE_CACHE[key] = (ref.detach(), g, grads, state)
        del orc
    return _ORACLE_CACHE[key]


def _param_class(n: str) -> str:
    if "rel_pos" in n:
        return "rel_pos"
    if ".pool." in n:
        return "pool_conv"
    if "norm" in n or "final_ln" in n:
        return "norm"
    if n.endswith("bias"):
        return "bias"
    return "weight"


def _check_grads(prod, ref_grads, tol, relpos_tol=None, rope=False, floor_frac=1e-3, label=""):
    """Relative Frobenius error per parameter tensor; the denominator is floored at `floor_frac` x the median gradient
    norm (bf16 rounding noise on a tensor whose own gradient is tiny scales with the activations, not with it).
    `tol`: one bar, or a dict per parameter class (rel_pos / pool_conv / norm / bias / weight) — set to about twice the
    worst error MEASURED for that class on an MI355X (the measured worst cases are printed on every run)."""
    norms = sorted(float(g.norm()) for g in ref_grads.values() if g is not None)
    median = norms[len(norms) // 2]
    bad = []
    worst: dict = {}
    for n, p in prod.named_parameters():
        ref = ref_grads[n]
        if ref is None:
            assert p.grad is None or p.grad.abs().max() == 0, n
            continue
        assert p.grad is not None, f"missing grad for {n}"
        got = p.grad.detach().cpu().double()
        # with RoPE the keys of blocks 3-15 are rotated after the pool (position dependent), so only blocks 0-2 keep the zero
        zero = any(k in n for k in ZERO_GRAD_KEYS) and (not rope or any(f"blocks.{i}." in n for i in (0, 1, 2)))
        if zero:
            # analytically zero: the oracle must agree (fp64 rounding only) and the HIP gradient must be noise-sized
            assert float(ref.norm()) < 1e-9 * median, (n, float(ref.norm()), median)
            assert float(got.norm()) < 0.1 * median, (n, float(got.norm()), median)
            continue
        cls = _param_class(n)
        if isinstance(tol, dict):
            t = tol[cls]
        else:
            t = (relpos_tol if relpos_tol is not None else tol) if cls == "rel_pos" else tol
        err = ((got - ref).norm() / ref.norm().clamp_min(floor_frac * median)).item()
        if err > worst.get(cls, ("", 0.0))[1]:
            worst[cls] = (n, err)
        if not err < t:
            bad.append((n, round(err, 4)))
    print(f"[grad check {label}] worst relative Frobenius error per class: "
          + "; ".join(f"{c} {e:.3e} ({n})" for c, (n, e) in sorted(worst.items())))
    assert not bad, bad[:20]


# Gradient bars per parameter class = about twice the worst relative Frobenius error MEASURED on an MI355X (r03 run, printed by
# every run as "[grad check ...]"): fp32 worst cases 5.5e-5 (bias) ... 2.0e-3 (rel_pos: sums of ~1e5 incoherent terms against the
# fp64 oracle); bf16 at 16 x 224^2: bias 1.7e-2, norm 8.3e-2, pool_conv 8.2e-2, rel_pos 1.15e-1, weight 8.4e-2; the 4 x 64^2 RoPE case
# is noisier (few tokens per table entry): bias 7.7e-2, norm 1.8e-1, pool_conv 1.25e-1, rel_pos 1.95e-1, weight 1.4e-1.
GTOL_F32 = {"bias": 2e-4, "norm": 2.5e-3, "pool_conv": 3e-3, "rel_pos": 4e-3, "weight": 2.6e-3}
GTOL_BF16 = {"bias": 4e-2, "norm": 1.6e-1, "pool_conv": 1.6e-1, "rel_pos": 2.2e-1, "weight": 1.6e-1}
GTOL_BF16_TINY = {"bias": 1.5e-1, "norm": 3.4e-1, "pool_conv": 2.5e-1, "rel_pos": 3.8e-1, "weight": 2.7e-1}

CASES = {
    # key: (T, S, B, N, pooling, oracle kwargs / product kwargs)
    "c1_single": dict(T=16, S=224, B=2, N=0, pooling="mean"),
    "c2_multi_attnpool": dict(T=16, S=224, B=1, N=4, pooling="attention"),
}


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, GTOL_F32), (torch.bfloat16, 5e-2, GTOL_BF16)])
@pytest.mark.parametrize("case", list(CASES))
def test_video_encoder_full_geometry_fwd_and_all_grads(case, dtype, tol, gtol):
    """configs[1] (single video) and configs[2] (N = 4 videos per study, AttentionPool + aggregator) at 16 x 224 x 224."""
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle

    c = CASES[case]
    x = _video(c["B"], c["N"], c["T"], c["S"])
    ref, g, ref_grads, state = _oracle_run(case, lambda: VideoEncoderOracle(c["T"], (c["S"], c["S"]), token_pooling_mode=c["pooling"]), x)
    prod = VideoEncoder(backbone="mvit", num_frames=c["T"], spatial_size=(c["S"], c["S"]), pretrained=False, freeze_ratio=1.0,
                        dropout=0.0, token_pooling_mode=c["pooling"], attention_pool_dropout=0.0, compute_dtype=dtype,
                        stochastic_depth_prob=0.0)
    prod.load_state_dict(state)
    prod = prod.cuda()
    out = prod(x.cuda())
    assert out.shape == ref.shape == (c["B"], 512) and out.dtype == torch.float32
    assert _rel(out.cpu(), ref) < tol, _rel(out.cpu(), ref)
    out.backward(g.float().cuda())
    _check_grads(prod, ref_grads, gtol, floor_frac=1e-3 if dtype == torch.float32 else 2e-2, label=f"{case} {dtype}")


def test_video_encoder_fp8_gemms_vs_oracle():
    """BASELINE configs[4] numerics: gemm_dtype="fp8" (e4m3 activations / weights, e5m2 gradients on the block-scaled MFMA,
    bf16 weight gradients) on the 16 x 224 x 224 trunk against the fp64 oracle of the c1 case: output within fp8-class
    tolerance, every gradient finite and pointing the oracle's way (cosine); the bf16 encoder is the yardstick."""
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle

    c = CASES["c1_single"]
    x = _video(c["B"], c["N"], c["T"], c["S"])
    ref, g, ref_grads, state = _oracle_run("c1_single", lambda: VideoEncoderOracle(c["T"], (c["S"], c["S"]), token_pooling_mode=c["pooling"]), x)
    outs = {}
    for mode in ("bf16", "fp8"):
        prod = VideoEncoder(num_frames=c["T"], spatial_size=(c["S"], c["S"]), pretrained=False, freeze_ratio=1.0, dropout=0.0,
                            compute_dtype=torch.bfloat16, stochastic_depth_prob=0.0, gemm_dtype=mode)
        prod.load_state_dict(state)
        prod = prod.cuda()
        out = prod(x.cuda())
        out.backward(g.float().cuda())
        err = _rel(out.detach().cpu(), ref)
        cos = []
        for n, p in prod.named_parameters():
            r = ref_grads[n]
            if r is None or p.grad is None or float(r.norm()) < 1e-12 or any(k in n for k in ZERO_GRAD_KEYS):
                continue
            assert torch.isfinite(p.grad).all(), n
            cos.append(torch.nn.functional.cosine_similarity(p.grad.detach().cpu().double().flatten(), r.flatten(), dim=0).item())
        cos = torch.tensor(cos)
        outs[mode] = (err, cos.median().item(), cos.min().item())
        print(f"[fp8 encoder] {mode}: output rel err {err:.3e}, gradient cosine median {cos.median():.4f} min {cos.min():.4f}")
        del prod
    assert outs["fp8"][0] < 1.5e-1 and outs["fp8"][1] > 0.97, outs
    assert outs["bf16"][0] < 5e-2 and outs["bf16"][1] > 0.995, outs


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, GTOL_F32), (torch.bfloat16, 5e-2, GTOL_BF16)])
@pytest.mark.parametrize("T,S,N", [(4, 64, 2), (16, 224, 1)])
def test_rope_token_output_with_gradients(T, S, N, dtype, tol, gtol):
    """use_rope=True, per_video_pool=False (all patch tokens), forward AND gradients: the rotation after pooling, its
    inverse in backward, and the missing residual-pool shortcut of the patched attention (models/video_encoder.py:268-378)."""
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle

    B = 1
    x = _video(B, N, T, S, seed=3)
    ref, g, ref_grads, state = _oracle_run(("rope", T, S, N), lambda: VideoEncoderOracle(T, (S, S), use_rope=True, aggregate_videos_tokens=False), x, gseed=4,
                                           coherent=True)
    prod = VideoEncoder(num_frames=T, spatial_size=(S, S), pretrained=False, freeze_ratio=1.0, dropout=0.0, use_rope=True,
                        aggregate_videos_tokens=False, per_video_pool=False, compute_dtype=dtype, stochastic_depth_prob=0.0)
    prod.load_state_dict(state)
    prod = prod.cuda()
    out = prod(x.cuda())
    assert out.shape == ref.shape and out.shape[1] == N * (T // 2 * (S // 32) ** 2 + 1)
    assert _rel(out.float().cpu(), ref) < tol, _rel(out.float().cpu(), ref)
    out.backward(g.to(out.dtype).cuda())
    # the aggregator is unused on this path: no gradients there on either side
    if dtype == torch.bfloat16 and S < 224:
        gtol = GTOL_BF16_TINY
    _check_grads(prod, ref_grads, gtol, rope=True, floor_frac=1e-3 if dtype == torch.float32 else 2e-2,
                 label=f"rope {T}x{S} N{N} {dtype}")


def test_config4_geometry_32_frames_rope_tokens_bf16():
    """BASELINE configs[4] geometry: 32 x 224 x 224 clips (785 tokens each), RoPE, per_video_pool=False — forward against
    the fp64 oracle in bf16 compute (the fp8 GEMM variant of that config is compared with this same oracle elsewhere)."""
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle, init_by_name

    T, S = 32, 224
    x = _video(1, 2, T, S, seed=5)
    orc = init_by_name(VideoEncoderOracle(T, (S, S), use_rope=True, aggregate_videos_tokens=False))
    with torch.no_grad():
        ref = orc(x)
    prod = VideoEncoder(num_frames=T, spatial_size=(S, S), pretrained=False, freeze_ratio=1.0, dropout=0.0, use_rope=True,
                        aggregate_videos_tokens=False, per_video_pool=False, compute_dtype=torch.bfloat16,
                        stochastic_depth_prob=0.0)
    prod.load_state_dict(orc.state_dict())
    prod = prod.cuda().eval()
    with torch.no_grad():
        out = prod(x.cuda())
    assert out.shape == ref.shape == (1, 2 * 785, 512)
    assert _rel(out.float().cpu(), ref) < 5e-2, _rel(out.float().cpu(), ref)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 1e-2)])
def test_train_step_full_geometry_12_layer_bert(dtype, tol):
    """One `_train_step` + the step after it at 16 x 224 x 224 with the 12-layer BERT-base tower (vocab 30 522): the first
    loss checks forward + loss, the second one checks backward + clip + AdamW through the updated weights."""
    from __graft_entry__ import make_tiny_setup
    from oracle.losses import contrastive_loss

    B = 2
    runner, (vo, to), (videos, ids, mask) = make_tiny_setup(dtype, B=B, T=16, S=224, seq=32, full_bert=True)
    lt = torch.nn.Parameter(torch.log(torch.tensor([0.07])))
    groups = [{"params": list(vo.model.parameters()), "lr": 1e-4, "weight_decay": 0.01},
              {"params": list(vo.aggregator.parameters()), "lr": 2e-4, "weight_decay": 0.01},
              {"params": list(to.parameters()), "lr": 2e-5, "weight_decay": 0.01}, {"params": [lt], "lr": 1e-4}]
    opt = torch.optim.AdamW(groups, lr=1e-4)
    mask[1, 20:] = 0
    for step in range(2 if dtype == torch.float32 else 1):  # the second step checks backward + clip + AdamW (fp32 only: time)
        opt.zero_grad(set_to_none=True)
        ref = contrastive_loss(vo(videos), to(ids, mask), lt)
        ref.backward()
        torch.nn.utils.clip_grad_norm_(list(vo.parameters()) + list(to.parameters()), 5.0)
        opt.step()
        m, emb = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
        assert abs(m["loss"] - ref.item()) < tol * max(1.0, abs(ref.item())), (step, m["loss"], ref.item())
    assert emb["video_embeddings"].shape == (B, 512)
    if dtype == torch.float32:
        assert abs(runner.log_temp.item() - lt.item()) < 1e-5
        w_ref = vo.model.blocks[0].attn.qkv.weight.detach()
        w_hip = runner.video_encoder.model.blocks[0].attn.qkv.weight.detach().cpu()
        # AdamW's first updates are lr * g / (|g| + eps): an element whose gradient is rounding noise may flip sign, so
        # single elements can differ by up to 2 steps x lr; the typical element must agree to fp32 round-off
        d = (w_hip - w_ref).abs()
        assert d.mean() < 2e-6 and d.max() <= 2.5e-4 and (d > 1e-5).float().mean() < 0.02, (d.mean(), d.max())


@pytest.mark.parametrize("kind", ["ce", "gated", "bce"])
def test_loss_at_global_batch_4096_vs_fp64(kind):
    """BASELINE configs[3] size: the [4096, 4096] similarity matrix of a global batch of 4096 (8 ranks x 512), InfoNCE,
    gated SigLIP and the BCE SigLIP loss with bias, value + all gradients against fp64 on the host."""
    import torch.nn.functional as F

    from deepcoro_clip_amd.losses import ContrastiveLoss, SigLIPLoss, SiglipLoss

    B, D = 4096, 512
    gen = torch.Generator().manual_seed(4096)
    v0 = torch.randn(B, D, generator=gen)
    t0 = torch.randn(B, D, generator=gen) + 0.7 * v0
    vr, tr = v0.double().requires_grad_(True), t0.double().requires_grad_(True)
    ltr = torch.log(torch.tensor(0.07, dtype=torch.float64)).requires_grad_(True)
    sim = F.normalize(vr, dim=-1) @ F.normalize(tr, dim=-1).t()
    v, t = v0.cuda().requires_grad_(True), t0.cuda().requires_grad_(True)
    lt = torch.nn.Parameter(torch.log(torch.tensor([0.07], device="cuda")))
    if kind == "bce":
        crit = SigLIPLoss().cuda()
        bias_r = torch.tensor(-10.0, dtype=torch.float64, requires_grad=True)
        logits = torch.clamp(sim / torch.clamp(torch.exp(ltr), min=1e-4) + bias_r, -30.0, 30.0)
        ref = F.binary_cross_entropy_with_logits(logits, torch.eye(B, dtype=torch.float64), reduction="mean")
        loss = crit(v, t, lt)
    else:
        if kind == "gated":
            sim = sim * torch.sigmoid(sim)
        logits = sim / torch.exp(ltr)
        tgt = torch.arange(B)
        ref = 0.5 * (F.cross_entropy(logits, tgt) + F.cross_entropy(logits.t(), tgt))
        loss = (SiglipLoss() if kind == "gated" else ContrastiveLoss())(v, t, lt)
    ref.backward()
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    relf = lambda a, b: ((a.detach().cpu().double() - b).norm() / b.norm().clamp_min(1e-30)).item()  # noqa: E731
    assert relf(v.grad, vr.grad) < 5e-4 and relf(t.grad, tr.grad) < 5e-4, (relf(v.grad, vr.grad), relf(t.grad, tr.grad))
    assert abs(lt.grad.item() - ltr.grad.item()) < 5e-4 * max(1.0, abs(ltr.grad.item()))
    if kind == "bce":
        assert abs(crit.bias.grad.item() - bias_r.grad.item()) < 5e-4 * max(1e-3, abs(bias_r.grad.item()))
    assert math.isfinite(loss.item())


@pytest.mark.parametrize("dtype,steps,tol", [(torch.float32, 3, 1e-4), (torch.bfloat16, 6, 1e-2)])
def test_headline_config_loss_curve_vs_oracle(dtype, steps, tol):
    """north_star: "loss curve matching CPU reference within 1e-2" on the 16-frame 224^2 multi-video config — the benchmark's own
    workload (BASELINE configs[2] shard: N = 4 videos per study, AttentionPool + aggregator, 12-layer BERT-base, InfoNCE, clip,
    AdamW), 2 studies, fresh random batches every step, the CPU oracle trained beside it with torch.optim.AdamW on the same
    name-seeded weights.  fp32: 1e-4 per step; bf16: 1e-2 per step (the tiny-geometry curves are in tests/test_step_gpu.py)."""
    from __graft_entry__ import make_tiny_setup
    from oracle.losses import contrastive_loss

    B, N = 2, 4
    runner, (vo, to), _ = make_tiny_setup(dtype, B=B, T=16, S=224, seq=32, full_bert=True, pooling="attention", videos_per_study=N)
    lt = torch.nn.Parameter(torch.log(torch.tensor([0.07])))
    groups = [{"params": list(vo.model.parameters()), "lr": 1e-4, "weight_decay": 0.01},
              {"params": list(vo.attention_pool.parameters()), "lr": 2e-4, "weight_decay": 0.01},
              {"params": list(vo.aggregator.parameters()), "lr": 2e-4, "weight_decay": 0.01},
              {"params": list(to.parameters()), "lr": 2e-5, "weight_decay": 0.01}, {"params": [lt], "lr": 1e-4}]
    opt = torch.optim.AdamW(groups, lr=1e-4)
    allp = list(vo.parameters()) + list(to.parameters())
    g = torch.Generator().manual_seed(2024)
    worst, curve = 0.0, []
    for step in range(steps):
        videos = torch.randn(B, N, 16, 224, 224, 3, generator=g)
        ids = torch.randint(0, 30522, (B, 32), generator=g)
        mask = torch.ones(B, 32, dtype=torch.int64)
        mask[1, 24:] = 0
        opt.zero_grad(set_to_none=True)
        ref = contrastive_loss(vo(videos), to(ids, mask), lt)
        ref.backward()
        torch.nn.utils.clip_grad_norm_(allp, 5.0)
        opt.step()
        m, _ = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
        err = abs(m["loss"] - ref.item()) / max(1.0, abs(ref.item()))
        curve.append((round(m["loss"], 5), round(ref.item(), 5)))
        worst = max(worst, err)
        assert err < tol, (step, curve)
    print(f"[headline loss curve] {dtype}: (HIP, oracle) per step {curve}; worst relative deviation {worst:.2e}")
