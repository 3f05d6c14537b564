"""Token-level consumers on the HIP path: MultiInstanceLinearProbing vs the goldens generated from the
reference module (tests/golden/mil.npz) and vs the CPU oracle's gradients; VideoMILWrapper over the HIP VideoEncoder;
the patch pooling of the multitask patch-contrastive term."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
    from tests.golden.make_golden import MIL_CASES
    return MIL_CASES


@pytest.mark.parametrize("idx", range(10))
def test_mil_matches_reference_golden_and_oracle_gradients(idx):
    from deepcoro_clip_amd.multi_instance_linear_probing import MultiInstanceLinearProbing
    from oracle.mil import MILOracle
    from oracle.model import init_by_name

    name, kw, four = _cases()[idx]
    g = np.load(os.path.join(G, "mil.npz"))
    x = torch.from_numpy(g["x4"] if four else g["x3"])
    mask = torch.from_numpy(g["mask"])
    orc = init_by_name(MILOracle(512, {"a": 3, "b": 1}, mode=kw["pooling_mode"], separate=kw.get("separate_video_attention", True),
                                 strategy=kw.get("normalization_strategy", "post_norm"))).double()
    m = MultiInstanceLinearProbing(512, {"a": 3, "b": 1}, **kw)
    assert sorted(m.state_dict().keys()) == list(g[f"{name}_keys"])
    m.load_state_dict({k: v.float() for k, v in orc.state_dict().items()})
    m = m.cuda().eval()
    xg = x.cuda().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = m(xg, mask=mask.cuda())
    for h in ("a", "b"):
        ref = torch.from_numpy(g[f"{name}_{h}"])
        assert (y[h].float().cpu() - ref).abs().max() < 2e-4 * ref.abs().max().clamp_min(1.0), (name, h)
    # gradients: same scalar on both sides
    xo = x.double().requires_grad_(True)
    yo = orc(xo, mask)
    (yo["a"].sum() + 2.0 * yo["b"].sum()).backward()
    (y["a"].sum() + 2.0 * y["b"].sum()).backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b).norm() / b.norm().clamp_min(1e-9))  # noqa: E731
    assert rel(xg.grad, xo.grad) < 1e-3, (name, rel(xg.grad, xo.grad))
    op = dict(orc.named_parameters())
    norms = sorted(float(q.grad.norm()) for q in op.values() if q.grad is not None)
    floor = 1e-3 * norms[len(norms) // 2]  # attention_w.bias shifts every score of a softmax alike: its gradient is exactly 0
    for n, p in m.named_parameters():
        if op[n].grad is None:
            continue
        err = float((p.grad.detach().cpu().double() - op[n].grad).norm() / op[n].grad.norm().clamp_min(floor))
        assert p.grad is not None and err < 2e-3, (name, n, err)


@pytest.mark.parametrize("idx", range(5))
def test_mil_view_angle_embeddings_match_reference_golden_and_oracle_gradients(idx):
    """num_view_classes > 0 (models/multi_instance_linear_probing.py:142-145,265-269): x + view_embedding(view_ids) broadcast
    over the patch axis, in front of every pooling rule — outputs against goldens produced by the reference module
    (tests/golden/mil_view.npz), gradients (input, embedding table, everything else) against the fp64 oracle."""
    from deepcoro_clip_amd.multi_instance_linear_probing import MultiInstanceLinearProbing
    from oracle.mil import MILOracle
    from oracle.model import init_by_name
    from tests.golden.make_golden import MIL_VIEW_CASES

    name, kw, four = MIL_VIEW_CASES[idx]
    g0 = np.load(os.path.join(G, "mil.npz"))
    g = np.load(os.path.join(G, "mil_view.npz"))
    x = torch.from_numpy(g0["x4"] if four else g0["x3"])
    mask, vid = torch.from_numpy(g0["mask"]), torch.from_numpy(g["view_ids"])
    orc = init_by_name(MILOracle(512, {"a": 3, "b": 1}, mode=kw["pooling_mode"], num_view_classes=kw["num_view_classes"])).double()
    m = MultiInstanceLinearProbing(512, {"a": 3, "b": 1}, **kw)
    assert sorted(m.state_dict().keys()) == list(g[f"{name}_keys"]) and m.view_pad_id == 4
    m.load_state_dict({k: v.float() for k, v in orc.state_dict().items()})
    m = m.cuda().eval()
    xg = x.cuda().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = m(xg, mask=mask.cuda(), view_ids=vid.cuda())
    for h in ("a", "b"):
        ref = torch.from_numpy(g[f"{name}_{h}"])
        assert (y[h].float().cpu() - ref).abs().max() < 2e-4 * ref.abs().max().clamp_min(1.0), (name, h)
    xo = x.double().requires_grad_(True)
    yo = orc(xo, mask, vid)
    (yo["a"].sum() + 2.0 * yo["b"].sum()).backward()
    (y["a"].sum() + 2.0 * y["b"].sum()).backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b).norm() / b.norm().clamp_min(1e-9))  # noqa: E731
    assert rel(xg.grad, xo.grad) < 1e-3, (name, rel(xg.grad, xo.grad))
    assert rel(m.view_embedding.weight.grad, orc.view_embedding.weight.grad) < 2e-3, name
    with pytest.raises(IndexError):
        m(xg.detach(), mask=mask.cuda(), view_ids=torch.full_like(vid, 5).cuda())
    # view ids without view classes configured are ignored, like the reference (:265)
    m0 = MultiInstanceLinearProbing(512, {"a": 3}, pooling_mode="mean").cuda().eval()
    x3 = torch.from_numpy(g0["x3"]).cuda()
    assert torch.equal(m0(x3, view_ids=vid.cuda())["a"], m0(x3)["a"])


def test_mil_edge_cases_and_errors():
    from deepcoro_clip_amd.multi_instance_linear_probing import MultiInstanceLinearProbing

    m = MultiInstanceLinearProbing(64, {"a": 2}, pooling_mode="mean+cls_token").cuda()
    x = torch.randn(2, 3, 64, device="cuda")
    out = m(x, mask=torch.zeros(2, 3, dtype=torch.bool, device="cuda"))  # no valid instance anywhere -> heads of zeros
    assert torch.equal(out["a"], m.heads["a"].bias.detach().expand(2, -1))
    with pytest.raises(ValueError):
        MultiInstanceLinearProbing(64, {"a": 2}, pooling_mode="median")
    with pytest.raises(ValueError):
        MultiInstanceLinearProbing(64, {})
    with pytest.raises(ValueError):
        m(torch.randn(2, 3, 32, device="cuda"))
    with pytest.raises(ValueError):
        m(x, mask=torch.ones(2, 4, device="cuda"))


def test_video_mil_wrapper_over_hip_encoder_and_patch_pooling():
    """VideoEncoder patch tokens [B, N*L, 512] -> [B, N, L, 512] -> hierarchical attention pooling -> task heads, end to end
    with gradients down to the patch embedding; the multitask patch-contrastive pooling equals tokens.mean(1)."""
    from deepcoro_clip_amd.losses import ContrastiveLoss
    from deepcoro_clip_amd.multi_instance_linear_probing import MultiInstanceLinearProbing, VideoMILWrapper, patch_contrastive_features
    from deepcoro_clip_amd.video_encoder import VideoEncoder

    torch.manual_seed(0)
    enc = VideoEncoder(num_frames=4, spatial_size=(64, 64), pretrained=False, freeze_ratio=1.0, dropout=0.0, compute_dtype=torch.float32,
                       aggregate_videos_tokens=False, per_video_pool=False, stochastic_depth_prob=0.0).cuda()
    mil = MultiInstanceLinearProbing(512, {"stenosis": 1, "view": 4}, pooling_mode="attention").cuda()
    wrap = VideoMILWrapper(enc, mil, num_videos=2)
    x = torch.randn(3, 2, 4, 64, 64, 3, device="cuda")
    out = wrap(x)
    assert out["stenosis"].shape == (3, 1) and out["view"].shape == (3, 4)
    (out["stenosis"].sum() + out["view"].sum()).backward()
    assert torch.isfinite(enc.model.conv_proj.weight.grad).all() and enc.model.conv_proj.weight.grad.abs().sum() > 0
    with torch.no_grad():
        tokens = enc.get_tokens(x, mode="patch")
        pooled = patch_contrastive_features(tokens)
        assert torch.allclose(pooled, tokens.mean(1), atol=1e-5)
        t = torch.randn(3, 512, device="cuda")
        lt = torch.log(torch.tensor(0.07)).cuda()
        assert torch.isfinite(ContrastiveLoss()(pooled, t, lt))
    with pytest.raises(ValueError):
        VideoMILWrapper(enc, mil, num_videos=7)(x)
