"""HIP loss path (csrc/loss.hip through the C ABI) against the golden vectors produced by the REFERENCE's own loss
modules (tests/golden/losses.npz, see make_golden.py) and against the CPU oracle on fresh seeded inputs.

Tolerances: north_star's 1e-4 for fp32 (values are O(1)); gradients 1e-4 relative in Frobenius norm;
embedding argmax bit-exact.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _g():
    return np.load(os.path.join(G, "losses.npz"), allow_pickle=False)


def _dev(a, grad=False):
    t = torch.from_numpy(np.asarray(a)).float().cuda()
    return t.requires_grad_(True) if grad else t


def _relf(a, b):
    a = a.detach().cpu().double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def _lt():
    return torch.log(torch.tensor(0.07)).cuda().requires_grad_(True)


@pytest.mark.parametrize("B", [8, 24])
@pytest.mark.parametrize("name", ["contrastive", "clip", "clip_ls", "siglip_gated"])
def test_ce_losses_match_reference_goldens(B, name):
    from deepcoro_clip_amd import losses as Ls

    g = _g()
    mod = {"contrastive": Ls.ContrastiveLoss(), "clip": Ls.CLIPLoss(), "clip_ls": Ls.CLIPLoss(label_smoothing=0.1),
           "siglip_gated": Ls.SiglipLoss()}[name]
    v, t, lt = _dev(g[f"v{B}"], True), _dev(g[f"t{B}"], True), _lt()
    loss = mod(v, t, lt)
    loss.backward()
    assert abs(loss.item() - float(g[f"{name}{B}_loss"])) < 1e-4
    assert _relf(v.grad, g[f"{name}{B}_dv"]) < 1e-4
    assert _relf(t.grad, g[f"{name}{B}_dt"]) < 1e-4
    assert abs(lt.grad.item() - float(g[f"{name}{B}_dlt"])) < 1e-4 * max(1.0, abs(float(g[f"{name}{B}_dlt"])))


@pytest.mark.parametrize("B", [8, 24])
def test_siglip_bce_matches_reference_goldens(B):
    from deepcoro_clip_amd.losses import SigLIPLoss

    g = _g()
    mod = SigLIPLoss().cuda()
    v, t, lt = _dev(g[f"v{B}"], True), _dev(g[f"t{B}"], True), _lt()
    loss = mod(v, t, lt)
    loss.backward()
    assert abs(loss.item() - float(g[f"siglip_bce{B}_loss"])) < 1e-4
    assert _relf(v.grad, g[f"siglip_bce{B}_dv"]) < 1e-4
    assert _relf(t.grad, g[f"siglip_bce{B}_dt"]) < 1e-4
    assert abs(lt.grad.item() - float(g[f"siglip_bce{B}_dlt"])) < 1e-4 * max(1.0, abs(float(g[f"siglip_bce{B}_dlt"])))
    assert abs(mod.bias.grad.item() - float(g[f"siglip_bce{B}_dbias"])) < 1e-5


def test_multipositive_branches_match_reference_goldens():
    """[8 videos x 13 texts] mask + severity weights: SigLIPLoss(pos_mask, pos_weights), WeightedSigLIPLoss and the
    runner's inline BCE(sum)/max(1, sum targets) branch (runner.py:1256-1296 of the reference)."""
    from deepcoro_clip_amd.losses import KIND_BCE, MODE_GATED, SigLIPLoss, WeightedSigLIPLoss, pair_loss
    from deepcoro_clip_amd import kernels as K

    g = _g()
    mask, w = _dev(g["mp_mask"]), _dev(g["mp_w"])
    # 1) sigmoid BCE with bias
    v, t, lt = _dev(g["mp_v"], True), _dev(g["mp_t"], True), _lt()
    mod = SigLIPLoss().cuda()
    loss = mod(v, t, lt, pos_mask=mask, pos_weights=w)
    loss.backward()
    assert abs(loss.item() - float(g["mp_bce_loss"])) < 1e-4
    assert _relf(v.grad, g["mp_bce_dv"]) < 1e-4 and _relf(t.grad, g["mp_bce_dt"]) < 1e-4
    assert abs(mod.bias.grad.item() - float(g["mp_bce_dbias"])) < 1e-5
    # 2) weighted bidirectional CE on gated logits
    v, t, lt = _dev(g["mp_v"], True), _dev(g["mp_t"], True), _lt()
    loss = WeightedSigLIPLoss()(v, t, lt, mask * w)
    loss.backward()
    assert abs(loss.item() - float(g["mp_wce_loss"])) < 1e-4
    assert _relf(v.grad, g["mp_wce_dv"]) < 1e-4 and _relf(t.grad, g["mp_wce_dt"]) < 1e-4
    assert abs(lt.grad.item() - float(g["mp_wce_dlt"])) < 1e-4 * max(1.0, abs(float(g["mp_wce_dlt"])))
    # 3) runner inline BCE: weights = positive_weights on positives, 1 on negatives; denominator on the device
    v, t, lt = _dev(g["mp_v"], True), _dev(g["mp_t"], True), _lt()
    denom = K.colsum(mask.contiguous().view(-1, 1))
    loss = pair_loss(v, t, lt, mode=MODE_GATED, kind=KIND_BCE, targets=mask, pos_weights=w, positive_weight=1.0,
                     negative_weight=1.0, thr=0.0, denom_dev=denom)
    loss.backward()
    assert abs(loss.item() - float(g["mp_rbce_loss"])) < 1e-4 * max(1.0, abs(float(g["mp_rbce_loss"])))
    assert _relf(v.grad, g["mp_rbce_dv"]) < 1e-4 and _relf(t.grad, g["mp_rbce_dt"]) < 1e-4
    assert abs(lt.grad.item() - float(g["mp_rbce_dlt"])) < 1e-4 * max(1.0, abs(float(g["mp_rbce_dlt"])))


@pytest.mark.parametrize("B,D", [(1, 512), (3, 64), (129, 512), (512, 512), (1000, 256)])
@pytest.mark.parametrize("gated", [False, True])
def test_ce_loss_vs_oracle_ragged_sizes(B, D, gated):
    """Sizes off the kernels' tile grid (1 row, odd rows, > one tile) + the SigLIP-scale matrix."""
    from deepcoro_clip_amd.losses import ContrastiveLoss, SiglipLoss
    from oracle.losses import contrastive_loss

    gen = torch.Generator().manual_seed(100 + B)
    v0, t0 = torch.randn(B, D, generator=gen), torch.randn(B, D, generator=gen)
    t0 = t0 + 0.5 * v0  # correlated pairs, like a partly trained model
    vr, tr = v0.clone().double().requires_grad_(True), t0.clone().double().requires_grad_(True)
    ltr = torch.log(torch.tensor(0.07, dtype=torch.float64)).requires_grad_(True)
    ref = contrastive_loss_double(vr, tr, ltr, gated)
    ref.backward()
    v, t, lt = v0.cuda().requires_grad_(True), t0.cuda().requires_grad_(True), _lt()
    loss = (SiglipLoss() if gated else ContrastiveLoss())(v, t, lt)
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item()))
    assert _relf(v.grad, vr.grad.numpy()) < 2e-4 and _relf(t.grad, tr.grad.numpy()) < 2e-4
    assert abs(lt.grad.item() - ltr.grad.item()) < 2e-4 * max(1.0, abs(ltr.grad.item()))
    # the fp32 oracle agrees with the double restatement too (sanity of the comparison itself)
    assert abs(contrastive_loss(v0, t0, torch.log(torch.tensor(0.07)), gated=gated).item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item()))


def contrastive_loss_double(v, t, log_temp, gated):
    import torch.nn.functional as F

    sim = F.normalize(v, dim=-1) @ F.normalize(t, dim=-1).t()
    if gated:
        sim = sim * torch.sigmoid(sim)
    logits = sim / torch.exp(log_temp)
    tgt = torch.arange(logits.size(0))
    return 0.5 * (F.cross_entropy(logits, tgt) + F.cross_entropy(logits.t(), tgt))


def test_bf16_embeddings_and_argmax():
    """bf16 embeddings in (the towers' bf16 mode casts back to fp32 before the loss, but the loss accepts bf16):
    loss within 1e-2, similarity argmax identical to the fp32 oracle's on the same rounded inputs."""
    from deepcoro_clip_amd.losses import ContrastiveLoss
    from oracle.losses import contrastive_loss

    g = _g()
    v, t = _dev(g["v24"]).bfloat16(), _dev(g["t24"]).bfloat16()
    diag = {}
    from deepcoro_clip_amd.losses import pair_loss
    loss = pair_loss(v.requires_grad_(True), t.requires_grad_(True), _lt(), mode=0, kind=0, diag_out=diag)
    ref = contrastive_loss(v.float().cpu(), t.float().cpu(), torch.log(torch.tensor(0.07)))
    assert abs(loss.item() - ref.item()) < 1e-2
    vr, tr = v.float().cpu(), t.float().cpu()
    sim_ref = torch.nn.functional.normalize(vr, dim=1) @ torch.nn.functional.normalize(tr, dim=1).t()
    assert torch.equal(diag["sim"].argmax(1).cpu(), sim_ref.argmax(1))
    assert torch.equal(diag["sim"].argmax(0).cpu(), sim_ref.argmax(0))
    assert ContrastiveLoss()(v, t, _lt()).item() == pytest.approx(loss.item(), abs=1e-6)


def test_step_metrics_match_oracle():
    """scal[] diagnostics of the fused loss node = the reference's per-step metrics (runner.py:1322-1335, 1470-1530;
    utils/retrieval_metrics.py:167-204)."""
    from deepcoro_clip_amd.losses import pair_loss
    from oracle.losses import step_metrics

    g = _g()
    for gated in (False, True):
        v, t = _dev(g["v24"]), _dev(g["t24"])
        diag = {}
        pair_loss(v, t, _lt().detach(), mode=1 if gated else 0, kind=0, diag_out=diag)
        s = diag["scal"].tolist()
        ref = step_metrics(v.cpu(), t.cpu(), torch.log(torch.tensor(0.07)), use_siglip=gated)
        assert abs(s[3] - ref["alignment_logprob"]) < 1e-4
        assert abs(s[4] - ref["alignment_cosine"]) < 1e-5
        assert abs(s[5] / 24 - ref["video_norm"]) < 1e-3 and abs(s[6] / 24 - ref["text_norm"]) < 1e-3


def test_upstream_gradient_scaling_and_shape_errors():
    from deepcoro_clip_amd.losses import ContrastiveLoss, WeightedSigLIPLoss

    g = _g()
    v, t, lt = _dev(g["v8"], True), _dev(g["t8"], True), _lt()
    (ContrastiveLoss()(v, t, lt) * 3.0).backward()
    assert _relf(v.grad, 3.0 * g["contrastive8_dv"]) < 1e-4
    assert abs(lt.grad.item() - 3.0 * float(g["contrastive8_dlt"])) < 3e-4
    with pytest.raises(ValueError):
        ContrastiveLoss()(v[:5], t, lt)
    with pytest.raises(ValueError):
        WeightedSigLIPLoss()(v, t, lt, torch.ones(3, 3, device="cuda"))


def _z():
    return np.load(os.path.join(G, "loss_zoo.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", ["bce_auto", "bce_ent", "bce_ent_off", "spf", "spf_noweights", "spf_auto", "spf_ent", "margin_wce"])
def test_loss_zoo_matches_reference_goldens(name):
    """N4 loss zoo on the HIP path vs tests/golden/loss_zoo.npz (generated from the reference's own modules): auto-balance,
    entropy regulariser (active / inactive), SiglipPairwiseFeatureLoss variants, abnormal-margin column offset."""
    from deepcoro_clip_amd import losses as Ls

    g = _z()
    v, t, lt = _dev(g["v"], True), _dev(g["t"], True), _lt()
    mask, w = _dev(g["mask"]), _dev(g["w"])
    mod = None
    if name == "bce_auto":
        mod = Ls.SigLIPLoss(auto_balance=True).cuda()
        loss = mod(v, t, lt, pos_mask=mask, pos_weights=w)
    elif name in ("bce_ent", "bce_ent_off"):
        mod = Ls.SigLIPLoss(entropy_regularization=True, entropy_weight=0.1, min_entropy_threshold=3.0 if name == "bce_ent" else 0.01).cuda()
        loss = mod(v, t, lt, pos_mask=mask, pos_weights=w)
    elif name == "spf":
        loss = Ls.SiglipPairwiseFeatureLoss()(v, t, lt, mask, w)
    elif name == "spf_noweights":
        loss = Ls.SiglipPairwiseFeatureLoss(use_positive_weights=False, positive_weight=2.0, negative_weight=0.5)(v, t, lt, mask, w)
    elif name == "spf_auto":
        loss = Ls.SiglipPairwiseFeatureLoss(auto_positive_weight=True)(v, t, lt, mask, w)
    elif name == "spf_ent":
        m2 = Ls.SiglipPairwiseFeatureLoss(entropy_regularization=True, entropy_weight=0.2, min_entropy_threshold=3.0)
        loss = m2(v, t, lt, mask, w)
    else:
        loss = Ls.WeightedSigLIPLoss()(v, t, lt, _dev(g["mask"] * g["w"]), col_margin=_dev(g["abn"]) * 0.35)
    loss.backward()
    assert abs(loss.item() - float(g[f"{name}_loss"])) < 1e-4 * max(1.0, abs(float(g[f"{name}_loss"]))), (loss.item(), float(g[f"{name}_loss"]))
    assert _relf(v.grad, g[f"{name}_dv"]) < 2e-4 and _relf(t.grad, g[f"{name}_dt"]) < 2e-4, (_relf(v.grad, g[f"{name}_dv"]), _relf(t.grad, g[f"{name}_dt"]))
    assert abs(lt.grad.item() - float(g[f"{name}_dlt"])) < 2e-4 * max(1e-3, abs(float(g[f"{name}_dlt"])))
    if mod is not None:
        assert abs(mod.bias.grad.item() - float(g[f"{name}_dbias"])) < 2e-4 * max(1e-6, abs(float(g[f"{name}_dbias"])))
    if name == "bce_ent":
        d = mod.get_entropy_diagnostics()
        ref = g["bce_ent_diag"]
        for k, r in zip(("entropy_mean", "entropy_min", "entropy_max", "entropy_normalized", "entropy_deficit", "bce_loss", "entropy_loss"), ref):
            assert abs(d[k] - float(r)) < 1e-4 * max(1.0, abs(float(r))), (k, d[k], float(r))


@pytest.mark.parametrize("name,kw,use_w", [("nce_mean", {}, True), ("nce_sum", {"reduction": "sum"}, True), ("nce_nopw", {}, False),
                                           ("nce_imp", {"use_importance_weighting": True}, True),
                                           ("nce_imp_sum", {"use_importance_weighting": True, "reduction": "sum"}, False)])
def test_multi_positive_infonce_matches_reference_goldens(name, kw, use_w):
    from deepcoro_clip_amd.losses import MultiPositiveInfoNCELoss

    g = _z()
    z = _dev(g["nce_logits"], True)
    loss = MultiPositiveInfoNCELoss(**kw)(z, _dev(g["nce_mask"]), _dev(g["nce_w"]) if use_w else None)
    (loss * 2.0).backward()
    assert abs(loss.item() - float(g[f"{name}_loss"])) < 1e-4 * max(1.0, abs(float(g[f"{name}_loss"])))
    assert _relf(z.grad, 2.0 * g[f"{name}_dz"]) < 2e-4
    if name == "nce_mean":  # nothing selected anywhere: zero loss, zero gradient
        z2 = _dev(g["nce_logits"], True)
        l0 = MultiPositiveInfoNCELoss()(z2, torch.zeros(8, 13, device="cuda"))
        l0.backward()
        assert l0.item() == 0.0 and float(z2.grad.abs().max()) == 0.0
        with pytest.raises(ValueError):
            MultiPositiveInfoNCELoss()(z2, torch.zeros(8, 12, device="cuda"))
        with pytest.raises(ValueError):
            MultiPositiveInfoNCELoss(reduction="max")


@pytest.mark.parametrize("B", [8, 24, 16])
@pytest.mark.parametrize("name", ["contrastive_ddp", "siglip_ddp"])
def test_ddp_loss_classes_match_reference_goldens(B, name):
    """The DDP loss classes against the reference's own (utils/loss/losses.py:104-158, 213-276; world size 1, golden
    `losses_ddp.npz`).  SiglipLossDDP: embeddings cross the gather as fp16 and the gradient returns as fp16 — every gradient
    element must be an fp16-representable number, and the value must match the reference's (which the fp32-gather variant
    of round 3 missed by 3-6e-4 relative in the gradients)."""
    from deepcoro_clip_amd import losses as Ls

    g = np.load(os.path.join(G, "losses_ddp.npz"), allow_pickle=False)
    mod = Ls.ContrastiveLossDDP() if name == "contrastive_ddp" else Ls.SiglipLossDDP()
    v, t, lt = _dev(g[f"v{B}"], True), _dev(g[f"t{B}"], True), _lt()
    loss = mod(v, t, lt)
    loss.backward()
    assert abs(loss.item() - float(g[f"{name}{B}_loss"])) < 1e-4 * max(1.0, abs(float(g[f"{name}{B}_loss"])))
    assert _relf(v.grad, g[f"{name}{B}_dv"]) < 1e-4
    assert _relf(t.grad, g[f"{name}{B}_dt"]) < 1e-4
    assert abs(lt.grad.item() - float(g[f"{name}{B}_dlt"])) < 1e-4 * max(1.0, abs(float(g[f"{name}{B}_dlt"])))
    if name == "siglip_ddp":
        for x in (v.grad, t.grad):
            assert torch.equal(x, x.half().float())
        # and the plain (fp32) gated loss on the same inputs is measurably different: the wire format is not a no-op
        v2, t2, lt2 = _dev(g[f"v{B}"], True), _dev(g[f"t{B}"], True), _lt()
        Ls.SiglipLoss()(v2, t2, lt2).backward()
        assert _relf(v2.grad, g[f"{name}{B}_dv"]) > 5e-5
