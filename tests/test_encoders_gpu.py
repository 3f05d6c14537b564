"""HIP VideoEncoder / TextEncoder vs the CPU oracle (oracle/model.py) on the same name-seeded weights and inputs.

fp32 compute: 1e-4 relative (north_star); bf16 compute: 1e-2-class tolerances, embedding argmax exact.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-6)).item()


def _make_video(B, N, T, S, seed=0):
    g = torch.Generator().manual_seed(seed)
    shape = (B, N, T, S, S, 3) if N else (B, T, S, S, 3)
    return torch.randn(shape, generator=g)


def _grad_check(prod, orc, names_tol, default_tol, floor_frac=1e-2):
    """Relative Frobenius error per parameter.  Some gradients are analytically zero (a key bias / key-LayerNorm bias
    shifts every logit of a row equally), so the denominator is floored by a fraction of the typical gradient norm
    (bf16 runs: rounding noise on such a parameter scales with the activations, not with its own tiny gradient)."""
    bad = []
    op = dict(orc.named_parameters())
    norms = sorted(float(q.grad.double().norm()) for q in op.values() if q.grad is not None)
    floor = floor_frac * norms[len(norms) // 2]
    for n, p in prod.named_parameters():
        if op[n].grad is None:
            assert p.grad is None or p.grad.abs().max() == 0, n
            continue
        assert p.grad is not None, f"missing grad for {n}"
        tol = default_tol
        for key, t in names_tol.items():
            if key in n:
                tol = t
        ref = op[n].grad.double()
        err = ((p.grad.cpu().double() - ref).norm() / ref.norm().clamp_min(floor)).item()
        if not err < tol:
            bad.append((n, err))
    assert not bad, bad[:20]


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 2e-3), (torch.bfloat16, 5e-2, 2e-1)])
@pytest.mark.parametrize("pooling,N", [("mean", 0), ("attention", 2), ("cls_token", 2)])
def test_video_encoder_vs_oracle(dtype, tol, gtol, pooling, N):
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle, init_by_name

    T, S, B = 4, 64, 2
    orc = init_by_name(VideoEncoderOracle(T, (S, S), token_pooling_mode=pooling)).double()
    prod = VideoEncoder(backbone="mvit", num_frames=T, spatial_size=(S, S), pretrained=False, freeze_ratio=1.0,
                        dropout=0.0, token_pooling_mode=pooling, attention_pool_dropout=0.0, compute_dtype=dtype,
                        stochastic_depth_prob=0.0)
    prod.load_state_dict({k: v.float() for k, v in orc.state_dict().items()})
    prod = prod.cuda().float()
    x = _make_video(B, N, T, S)
    ref = orc(x.double())
    out = prod(x.cuda())
    assert out.shape == ref.shape == (B, 512) and out.dtype == torch.float32
    assert _rel(out.cpu(), ref) < tol, _rel(out.cpu(), ref)
    # backward: same upstream gradient on both sides
    g = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1)).double()
    ref.backward(g)
    out.backward(g.float().cuda())
    _grad_check(prod, orc, {"rel_pos": gtol * 4}, gtol, 1e-2 if dtype == torch.float32 else 5e-2)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 5e-2)])
def test_video_encoder_rope_tokens(dtype, tol):
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle, init_by_name

    T, S, B, N = 4, 64, 1, 2
    orc = init_by_name(VideoEncoderOracle(T, (S, S), use_rope=True, aggregate_videos_tokens=False)).double()
    prod = VideoEncoder(num_frames=T, spatial_size=(S, S), pretrained=False, freeze_ratio=1.0, dropout=0.0, use_rope=True,
                        aggregate_videos_tokens=False, per_video_pool=False, compute_dtype=dtype, stochastic_depth_prob=0.0)
    prod.load_state_dict({k: v.float() for k, v in orc.state_dict().items()})
    prod = prod.cuda()
    x = _make_video(B, N, T, S, seed=3)
    ref = orc(x.double())
    out = prod(x.cuda())
    assert out.shape == ref.shape  # [B, N*L, 512] patch tokens (tests/test_video_encoder_tokens.py:37-54)
    assert _rel(out.float().cpu(), ref) < tol


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 2e-3), (torch.bfloat16, 5e-2, 2e-1)])
def test_text_encoder_vs_oracle(dtype, tol, gtol):
    from deepcoro_clip_amd.text_encoder import TextEncoder
    from oracle.model import TextEncoderOracle, init_by_name

    small = dict(vocab=1000, hidden=768, layers=2, heads=12, ffn=3072, max_pos=64)
    orc = init_by_name(TextEncoderOracle(**small)).double()
    prod = TextEncoder(dropout=0.0, freeze_ratio=0.0, compute_dtype=dtype,
                       bert_config=dict(vocab_size=1000, num_hidden_layers=2, max_position_embeddings=64,
                                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    prod.load_state_dict({k: v.float() for k, v in orc.state_dict().items()})
    prod = prod.cuda()
    g = torch.Generator().manual_seed(2)
    B, S = 3, 32
    ids = torch.randint(0, 1000, (B, S), generator=g)
    mask = torch.ones(B, S, dtype=torch.int64)
    mask[1, 20:] = 0  # right padding
    ref = orc(ids, mask)
    out = prod(ids.cuda(), mask.cuda())
    assert out.shape == (B, 512)
    assert _rel(out.float().cpu(), ref) < tol
    gg = torch.randn(ref.shape, generator=g).double()
    ref.backward(gg)
    out.backward(gg.to(out.dtype).cuda())
    _grad_check(prod, orc, {}, gtol)


def test_video_encoder_c1_size_relpos_interpolation():
    """BASELINE configs[0] geometry (4 frames, 112 x 112): the 7 -> 4 pooled grid is not what torchvision's floor
    bookkeeping (7 // 2 = 3) sized the relative-position tables for, so they are linearly resampled
    (torchvision mvit._interpolate) — forward and every parameter gradient against the oracle, fp32."""
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle, init_by_name

    T, S, B = 4, 112, 2
    orc = init_by_name(VideoEncoderOracle(T, (S, S))).double()
    prod = VideoEncoder(num_frames=T, spatial_size=(S, S), pretrained=False, freeze_ratio=1.0, dropout=0.0,
                        compute_dtype=torch.float32, stochastic_depth_prob=0.0)
    prod.load_state_dict({k: v.float() for k, v in orc.state_dict().items()})
    prod = prod.cuda()
    assert any(w is not None for blk in prod.model.blocks for w in blk._idx(torch.device("cuda"))[2]), \
        "this geometry must exercise the interpolation path"
    x = _make_video(B, 0, T, S)
    ref = orc(x.double())
    out = prod(x.cuda())
    assert _rel(out.cpu(), ref) < 2e-4, _rel(out.cpu(), ref)
    g = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1)).double()
    ref.backward(g)
    out.backward(g.float().cuda())
    _grad_check(prod, orc, {"rel_pos": 8e-3}, 2e-3)


def test_stochastic_depth_semantics():
    """torchvision StochasticDepth(p, "row") on both residual branches: identity in eval mode (equals the oracle);
    in train mode whole clips keep or drop a branch, survivors are scaled by 1/(1-p), gradients follow the mask."""
    from deepcoro_clip_amd import functional as Fn
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle, init_by_name

    # the fused kernel against its definition
    gen = torch.Generator().manual_seed(3)
    B, N, C = 6, 5, 64
    resid = torch.randn(B * N, C, generator=gen).cuda().requires_grad_(True)
    branch = torch.randn(B * N, C, generator=gen).cuda().requires_grad_(True)
    scale = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25, 1.25], device="cuda")
    out = Fn.StochasticDepthFn.apply(resid, branch, scale, B)
    ref = resid.detach() + branch.detach() * scale.repeat_interleave(N)[:, None]
    assert torch.allclose(out, ref, atol=1e-6)
    out.sum().backward()
    assert torch.equal(resid.grad, torch.ones_like(resid))
    assert torch.allclose(branch.grad, scale.repeat_interleave(N)[:, None].expand(-1, C))
    # model level
    T, S = 4, 64
    orc = init_by_name(VideoEncoderOracle(T, (S, S))).double()
    prod = VideoEncoder(num_frames=T, spatial_size=(S, S), pretrained=False, freeze_ratio=1.0, dropout=0.0,
                        compute_dtype=torch.float32)  # default stochastic_depth_prob = 0.2 like mvit_v2_s
    prod.load_state_dict({k: v.float() for k, v in orc.state_dict().items()})
    prod = prod.cuda()
    assert [round(b.sd_prob, 4) for b in prod.model.blocks][::5] == [0.0, round(0.2 * 5 / 15, 4), round(0.2 * 10 / 15, 4), 0.2]
    x = _make_video(8, 0, T, S)
    ref = orc(x.double())
    prod.eval()
    assert _rel(prod(x.cuda()).cpu(), ref) < 2e-4
    prod.train()
    torch.manual_seed(0)
    a = prod(x.cuda())
    torch.manual_seed(0)
    b = prod(x.cuda())
    torch.manual_seed(1)
    c = prod(x.cuda())
    assert torch.equal(a, b) and not torch.equal(a, c), "masks come from torch's device generator"
    assert _rel(a.cpu(), ref) > 1e-3, "training mode must actually drop branches"
    a.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in prod.parameters() if p.grad is not None)


def test_full_size_properties_16x224():
    """BASELINE configs[1] geometry (16 x 224 x 224, MViT-v2-S, 393 tokens per clip) is too big for the CPU oracle to
    be a quick checker, so the full-size path is pinned by size-independent properties instead:
    per-clip independence (a clip's embedding does not depend on its batch mates), permutation equivariance over the
    batch, run-to-run determinism of the forward, symmetry of the InfoNCE loss under exchanging the two towers, and
    the loss of perfectly aligned unit embeddings at temperature tau against its closed form."""
    import math

    from deepcoro_clip_amd.losses import ContrastiveLoss
    from deepcoro_clip_amd.video_encoder import VideoEncoder

    torch.manual_seed(0)
    enc = VideoEncoder(num_frames=16, spatial_size=(224, 224), pretrained=False, freeze_ratio=1.0, dropout=0.0,
                       compute_dtype=torch.bfloat16, stochastic_depth_prob=0.0).cuda().eval()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 16, 224, 224, 3, generator=g).cuda()
    with torch.no_grad():
        tok = enc.model.forward_features(x, torch.bfloat16)
        assert tok.shape == (3, 393, 768)
        y = enc(x)
        assert y.shape == (3, 512) and torch.isfinite(y).all()
        assert torch.equal(y, enc(x)), "forward must be deterministic"
        perm = torch.tensor([2, 0, 1], device="cuda")
        yp = enc(x[perm])
        assert (yp - y[perm]).abs().max() <= 2e-2 * y.abs().max()
        y1 = enc(x[1:2])
        assert (y1[0] - y[1]).abs().max() <= 2e-2 * y.abs().max()
    # loss properties at the runner's batch size
    v = torch.randn(24, 512, generator=g).cuda()
    t = torch.randn(24, 512, generator=g).cuda()
    lt = torch.log(torch.tensor(0.07)).cuda()
    crit = ContrastiveLoss()
    assert abs(crit(v, t, lt).item() - crit(t, v, lt).item()) < 1e-5
    e = torch.nn.functional.normalize(torch.randn(24, 512, generator=g), dim=1).cuda()
    # identical towers: logits = cos/tau with a unit diagonal; closed form via fp64 on the host
    sim = (e.double().cpu() @ e.double().cpu().t()) / 0.07
    ref = -(torch.diagonal(torch.log_softmax(sim, dim=1)).mean() + torch.diagonal(torch.log_softmax(sim.t(), dim=1)).mean()) / 2
    assert abs(crit(e, e, lt).item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item()))
    assert math.isfinite(ref.item())


def test_attention_pool_with_cls_matches_reference_golden():
    """AttentionPoolWithCLS (token_pooling_mode='cls_token') on the HIP path vs the golden produced by the reference module
    (tests/golden/pools2.npz), fp32 compute; state_dict names are the reference's (nn.TransformerEncoder inside)."""
    import os

    import numpy as np
    from deepcoro_clip_amd.attention_pool import AttentionPoolWithCLS
    from oracle.model import AttentionPoolWithCLSOracle, init_by_name

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pools2.npz"))
    orc = init_by_name(AttentionPoolWithCLSOracle(512, 8))
    m = AttentionPoolWithCLS(512, num_heads=8, dropout=0.0)
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    m.load_state_dict(orc.state_dict())
    m = m.cuda().eval()
    y = m(torch.from_numpy(g["apcls_x"]).cuda())
    assert _rel(y.cpu(), torch.from_numpy(g["apcls_y"])) < 2e-4


def test_masked_pools_and_aggregator_match_reference_golden_and_oracle_gradients():
    """key_padding_mask (True = ignored) on the HIP path: AttentionPool, AttentionPoolWithCLS and the masked
    EnhancedVideoAggregator against goldens produced by the reference modules (tests/golden/pools_masked.npz: ragged masks,
    a row with a single valid token / video), and every gradient against the fp64 oracle."""
    import os

    import numpy as np
    from deepcoro_clip_amd.attention_pool import AttentionPool, AttentionPoolWithCLS
    from deepcoro_clip_amd.video_aggregator import EnhancedVideoAggregator
    from oracle.model import AggregatorOracle, AttentionPoolOracle, AttentionPoolWithCLSOracle, init_by_name

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pools_masked.npz"))

    def check(prod, orc, x, mask, ref, tol):
        prod.load_state_dict({k: v.float() for k, v in orc.state_dict().items()})
        prod = prod.cuda().eval()
        xg = x.cuda().requires_grad_(True)
        y = prod(xg, mask=mask.cuda())
        assert _rel(y.detach().cpu(), ref) < tol, (type(prod).__name__, _rel(y.detach().cpu(), ref))
        up = torch.randn(y.shape, generator=torch.Generator().manual_seed(3))
        y.backward(up.cuda())
        orc = orc.double()
        xo = x.double().requires_grad_(True)
        orc(xo, mask).backward(up.double())
        relf = lambda a, b: float((a.detach().cpu().double() - b).norm() / b.norm().clamp_min(1e-12))  # noqa: E731
        assert relf(xg.grad, xo.grad) < 2e-3, (type(prod).__name__, relf(xg.grad, xo.grad))
        # masked tokens / videos receive exactly no gradient
        assert float(xg.grad.cpu()[mask].abs().max()) == 0.0
        op = dict(orc.named_parameters())
        norms = sorted(float(q.grad.norm()) for q in op.values() if q.grad is not None)
        floor = 1e-3 * norms[len(norms) // 2]
        for n, p in prod.named_parameters():
            if op[n].grad is None:
                continue
            err = float((p.grad.detach().cpu().double() - op[n].grad).norm() / op[n].grad.norm().clamp_min(floor))
            assert err < 3e-3, (type(prod).__name__, n, err)

    x, mask = torch.from_numpy(g["x"]), torch.from_numpy(g["mask"])
    check(AttentionPool(512, num_heads=8, dropout=0.0), init_by_name(AttentionPoolOracle(512, 8)), x, mask, torch.from_numpy(g["ap_y"]), 2e-4)
    check(AttentionPoolWithCLS(512, num_heads=8, dropout=0.0), init_by_name(AttentionPoolWithCLSOracle(512, 8)), x, mask,
          torch.from_numpy(g["apcls_y"]), 3e-4)
    for n in (4, 8):
        check(EnhancedVideoAggregator(512, num_heads=4, dropout=0.0, aggregator_depth=2), init_by_name(AggregatorOracle(512, 4, 2)),
              torch.from_numpy(g[f"ag_x{n}"]), torch.from_numpy(g[f"ag_m{n}"]), torch.from_numpy(g[f"ag_y{n}"]), 2e-4)
    # a study without any valid video pools to zeros (reference: fallback weights valid / sum(valid) = 0)
    agg = EnhancedVideoAggregator(512, num_heads=4, dropout=0.0).cuda().eval()
    allm = torch.ones(2, 4, dtype=torch.bool); allm[0, 0] = False
    y = agg(torch.randn(2, 4, 512).cuda(), mask=allm.cuda())
    assert torch.all(y[1] == 0) and torch.isfinite(y).all() and y[0].abs().max() > 0


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 2e-3), (torch.bfloat16, 5e-2, 2e-1)])
def test_video_encoder_mvit_rope_backbone_takes_the_pooled_cls_fallback(dtype, tol, gtol):
    """backbone="mvit_rope" (models/video_encoder.py:84-90): same trunk, use_rope forced on — but the blocks are only patched
    for backbone == "mvit" (:256-257) and the forward goes through torchvision's MViT.forward (:544, :552-561): the class token
    behind the final norm as the clip's ONE token (L = 1).  Output + every gradient against the oracle."""
    from deepcoro_clip_amd.video_encoder import VideoEncoder
    from oracle.model import VideoEncoderOracle, init_by_name

    T, S, B, N = 4, 64, 2, 2
    orc = init_by_name(VideoEncoderOracle(T, (S, S), backbone="mvit_rope", use_rope=True)).double()
    prod = VideoEncoder(backbone="mvit_rope", num_frames=T, spatial_size=(S, S), pretrained=False, freeze_ratio=1.0, dropout=0.0,
                        compute_dtype=dtype, stochastic_depth_prob=0.0)
    assert prod.use_rope and prod.model.rope_cfg is None
    prod.load_state_dict({k: v.float() for k, v in orc.state_dict().items()})
    prod = prod.cuda().float()
    x = _make_video(B, N, T, S, seed=5)
    ref = orc(x.double())
    out, tokens = prod(x.cuda(), return_tokens=True)
    assert tokens.shape == (B, N, 512)  # N clips x L = 1 token each
    assert out.shape == ref.shape == (B, 512)
    assert _rel(out.cpu(), ref) < tol, _rel(out.cpu(), ref)
    g = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1)).double()
    ref.backward(g)
    out.backward(g.float().cuda())
    _grad_check(prod, orc, {"rel_pos": gtol * 4}, gtol, 1e-2 if dtype == torch.float32 else 5e-2)
