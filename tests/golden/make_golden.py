#!/usr/bin/env python
"""Generate the golden vectors that pin oracle/ to the REFERENCE's own arithmetic.

Runs only in the build container (needs /root/reference, read-only): imports the reference's importable hot-path
modules — utils/loss/{losses,contrastive,weighted_siglip}.py, utils/retrieval_metrics.py and, by file
path, models/{rope_3d,attention_pool,video_aggregator}.py — plus HF `BertModel(BertConfig(...))` (the arithmetic
behind models/text_encoder.py:52,131), evaluates them on seeded inputs and stores inputs + outputs as small .npz
fixtures.  Nothing of the reference's source travels: fixtures are plain arrays.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import zlib

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(1, ROOT)


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def init_by_name(module, seed=0):
    # must stay identical to oracle.model.init_by_name
    with torch.no_grad():
        for name, p in module.named_parameters():
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7FFFFFFF)
            r = torch.randn(p.shape, generator=g)
            lname = name.lower()
            is_norm = p.dim() == 1 and any(t in lname for t in ("norm", "final_ln"))
            if is_norm and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * r)
            elif p.dim() == 1:
                p.copy_(0.02 * r)
            elif "conv_proj" in name or ".pool." in name or "rel_pos" in name:
                p.copy_(0.1 * r)
            else:
                p.copy_(0.02 * r)
    return module


def np_(t):
    return t.detach().cpu().numpy()


def losses():
    from utils.loss import contrastive as C
    from utils.loss import losses as Ls
    from utils.loss.weighted_siglip import WeightedSigLIPLoss

    out = {}
    for B in (8, 24):
        g = torch.Generator().manual_seed(0)
        v = torch.randn(B, 512, generator=g).requires_grad_(True)
        t = torch.randn(B, 512, generator=g).requires_grad_(True)
        lt = torch.log(torch.tensor(0.07)).requires_grad_(True)
        out[f"v{B}"], out[f"t{B}"] = np_(v), np_(t)
        for name, mod in (("contrastive", Ls.ContrastiveLoss()), ("clip", C.CLIPLoss()), ("clip_ls", C.CLIPLoss(label_smoothing=0.1)),
                          ("siglip_gated", Ls.SiglipLoss()), ("siglip_bce", C.SigLIPLoss())):
            for x in (v, t, lt):
                x.grad = None
            params = list(mod.parameters())
            loss = mod(v, t, lt)
            loss.backward()
            out[f"{name}{B}_loss"] = np_(loss)
            out[f"{name}{B}_dv"], out[f"{name}{B}_dt"], out[f"{name}{B}_dlt"] = np_(v.grad), np_(t.grad), np_(lt.grad)
            if params:
                out[f"{name}{B}_dbias"] = np_(params[0].grad)
    # multi-positive: 8 videos x 13 texts
    g = torch.Generator().manual_seed(7)
    v = torch.randn(8, 512, generator=g).requires_grad_(True)
    t = torch.randn(13, 512, generator=g).requires_grad_(True)
    lt = torch.log(torch.tensor(0.07)).requires_grad_(True)
    mask = torch.zeros(8, 13)
    for i in range(8):
        for j in torch.randperm(13, generator=g)[: 1 + i % 3].tolist():
            mask[i, j] = 1.0
    w = mask * (1.0 + 9.0 * torch.rand(8, 13, generator=g))
    out.update(mp_v=np_(v), mp_t=np_(t), mp_mask=np_(mask), mp_w=np_(w))
    # contrastive.SigLIPLoss with pos_mask / pos_weights
    mod = C.SigLIPLoss()
    loss = mod(v, t, lt, pos_mask=mask, pos_weights=w)
    loss.backward()
    out.update(mp_bce_loss=np_(loss), mp_bce_dv=np_(v.grad), mp_bce_dt=np_(t.grad), mp_bce_dlt=np_(lt.grad),
               mp_bce_dbias=np_(mod.bias.grad))
    # runner inline branch (runners/video_constrative_learning_runner.py:1256-1296) evaluated with the reference's
    # own WeightedSigLIPLoss / F.binary_cross_entropy_with_logits on the reference's gated logits
    import torch.nn.functional as F

    for x in (v, t, lt):
        x.grad = None
    sim = F.normalize(v, dim=1) @ F.normalize(t, dim=1).t()
    logits = sim * torch.sigmoid(sim) / torch.exp(lt.float())
    loss = WeightedSigLIPLoss()(logits=logits, positive_weights=mask * w)
    loss.backward()
    out.update(mp_wce_loss=np_(loss), mp_wce_dv=np_(v.grad), mp_wce_dt=np_(t.grad), mp_wce_dlt=np_(lt.grad))
    for x in (v, t, lt):
        x.grad = None
    sim = F.normalize(v, dim=1) @ F.normalize(t, dim=1).t()
    logits = sim * torch.sigmoid(sim) / torch.exp(lt.float())
    wm = torch.where(mask > 0, w, torch.full_like(mask, 1.0))
    loss = F.binary_cross_entropy_with_logits(logits, mask, weight=wm, reduction="sum") / max(1.0, float(mask.sum()))
    loss.backward()
    out.update(mp_rbce_loss=np_(loss), mp_rbce_dv=np_(v.grad), mp_rbce_dt=np_(t.grad), mp_rbce_dlt=np_(lt.grad))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)


def losses_ddp():
    """ContrastiveLossDDP / SiglipLossDDP (utils/loss/losses.py:104-158, 213-276) at world size 1 — `gather_all` returns its
    input when no process group exists, but SiglipLossDDP's fp16 round trip (`.half()` before the gather, `.float()` after)
    still happens, in the value and in the gradient."""
    from utils.loss import losses as Ls

    out = {}
    for B, scale in ((8, 1.0), (24, 1.0), (16, 37.0)):  # the third case: large-norm embeddings (fp16 rounding is relative)
        g = torch.Generator().manual_seed(11 + B)
        v = (scale * torch.randn(B, 512, generator=g)).requires_grad_(True)
        t = (scale * torch.randn(B, 512, generator=g)).requires_grad_(True)
        lt = torch.log(torch.tensor(0.07)).requires_grad_(True)
        out[f"v{B}"], out[f"t{B}"] = np_(v), np_(t)
        for name, mod in (("contrastive_ddp", Ls.ContrastiveLossDDP()), ("siglip_ddp", Ls.SiglipLossDDP())):
            for x in (v, t, lt):
                x.grad = None
            loss = mod(v, t, lt)
            loss.backward()
            out[f"{name}{B}_loss"] = np_(loss)
            out[f"{name}{B}_dv"], out[f"{name}{B}_dt"], out[f"{name}{B}_dlt"] = np_(v.grad), np_(t.grad), np_(lt.grad)
    np.savez_compressed(os.path.join(OUT, "losses_ddp.npz"), **out)


def rope():
    R = load_by_path("ref_rope_3d", "models/rope_3d.py")
    out = {}
    g = torch.Generator().manual_seed(3)
    m = R.Rope3D(embed_dim=4 * 96, num_heads=4).eval()
    q = torch.randn(2, 4, 393, 96, generator=g)
    k = torch.randn(2, 4, 393, 96, generator=g)
    qr, kr = m(q, k, 8, 7, 7, n_special=1)
    tok = torch.tensor(list(range(0, 60)) + list(range(200, 212)) + list(range(385, 393)))
    out.update(tok=np_(tok), q=np_(q[:, :, tok]), k=np_(k[:, :, tok]), q_rot=np_(qr[:, :, tok]), k_rot=np_(kr[:, :, tok]))
    # asymmetric grids as in the patched attention (q grid 8x14x14, k grid 8x7x7), small batch
    q2 = torch.randn(1, 4, 1 + 8 * 14 * 14, 96, generator=g)
    k2 = torch.randn(1, 4, 1 + 8 * 7 * 7, 96, generator=g)
    q2r, _ = m(q2, q2, 8, 14, 14, n_special=1)
    _, k2r = m(k2, k2, 8, 7, 7, n_special=1)
    sel = torch.tensor([0, 1, 2, 15, 197, 198, 1568])
    out.update(q2_sel=np_(q2[:, :, sel]), q2_rot_sel=np_(q2r[:, :, sel]), sel=np_(sel), k2_sel=np_(k2[:, :, :8]),
               k2_rot_sel=np_(k2r[:, :, :8]))
    sin, cos = m._get_cached_freqs(8, 7, 7, torch.device("cpu"), torch.float32, 1)
    out.update(sin=np_(sin), cos=np_(cos))
    np.savez_compressed(os.path.join(OUT, "rope.npz"), **out)


def pools():
    AP = load_by_path("ref_attention_pool", "models/attention_pool.py")
    AG = load_by_path("ref_video_aggregator", "models/video_aggregator.py")
    out = {}
    g = torch.Generator().manual_seed(5)
    ap = init_by_name(AP.AttentionPool(512, num_heads=8, dropout=0.0)).eval()
    x = torch.randn(2, 50, 512, generator=g)
    out.update(ap_x=np_(x), ap_y=np_(ap(x)))
    ag = init_by_name(AG.EnhancedVideoAggregator(512, num_heads=4, dropout=0.0, aggregator_depth=2)).eval()
    for n in (1, 4, 8):
        xn = torch.randn(2, n, 512, generator=g)
        out[f"ag_x{n}"], out[f"ag_y{n}"] = np_(xn), np_(ag(xn))
    np.savez_compressed(os.path.join(OUT, "pools.npz"), **out)


def bert():
    from transformers import BertConfig, BertModel

    cfg = BertConfig(vocab_size=1000, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = init_by_name(BertModel(cfg, add_pooling_layer=False)).eval()
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 1000, (3, 32), generator=g)
    mask = torch.ones(3, 32, dtype=torch.int64)
    mask[1, 20:] = 0
    with torch.no_grad():
        h = m(input_ids=ids, attention_mask=mask).last_hidden_state
    np.savez_compressed(os.path.join(OUT, "bert.npz"), ids=np_(ids), mask=np_(mask), cls=np_(h[:, 0]), tok5=np_(h[:, 5]),
                        keys=np.array(sorted(m.state_dict().keys())))


def retrieval():
    import utils.retrieval_metrics as RM

    out = {}
    torch.manual_seed(42)
    sim = torch.randn(100, 100)
    gt = torch.arange(100)
    out["sim"] = np_(sim)
    rec = RM.compute_recall_at_k(sim, gt, [1, 5, 10])
    out.update(recall=np.array([rec["Recall@1"], rec["Recall@5"], rec["Recall@10"]]), mrr=np.array(RM.compute_mrr(sim, gt)["MRR_V2T"]),
               map=np.array(RM.compute_map(sim, gt)), median=np.array(RM.compute_median_rank(sim, gt)),
               ndcg=np.array([RM.compute_ndcg_at_k(sim, gt, [5, 10])[k] for k in ("NDCG@5_V2T", "NDCG@10_V2T")]))
    # multi-label, non-square (10 queries x 5 candidates)
    sim2 = torch.randn(10, 5)
    gt2 = [[0, 2], [1], [4, 3], [], [2], [0], [1, 4], [3], [2, 0], [4]]
    rec2 = RM.compute_recall_at_k(sim2, gt2, [1, 3, 10])
    out.update(sim2=np_(sim2), gt2=np.array([g + [-1] * (2 - len(g)) for g in gt2]),
               recall2=np.array([rec2["Recall@1"], rec2["Recall@3"], rec2["Recall@10"]]),
               mrr2=np.array(RM.compute_mrr(sim2, gt2)["MRR_V2T"]), map2=np.array(RM.compute_map(sim2, gt2)),
               median2=np.array(RM.compute_median_rank(sim2, gt2)),
               ndcg2=np.array([RM.compute_ndcg_at_k(sim2, gt2, [3])["NDCG@3_V2T"]]))
    v = torch.randn(16, 512)
    t = torch.randn(16, 512)
    out.update(ev=np_(v), et=np_(t), norms=np.array(list(RM.compute_embedding_norms(v, t).values())),
               align=np.array(RM.compute_alignment_score(v, t)))
    np.savez_compressed(os.path.join(OUT, "retrieval.npz"), **out)


def pools2():
    """AttentionPoolWithCLS (models/attention_pool.py:104-197; token_pooling_mode='cls_token')."""
    AP = load_by_path("ref_attention_pool", "models/attention_pool.py")
    g = torch.Generator().manual_seed(9)
    m = init_by_name(AP.AttentionPoolWithCLS(512, num_heads=8, dropout=0.0)).eval()
    x = torch.randn(2, 50, 512, generator=g)
    with torch.no_grad():
        y = m(x)
    np.savez_compressed(os.path.join(OUT, "pools2.npz"), apcls_x=np_(x), apcls_y=np_(y), keys=np.array(sorted(m.state_dict().keys())))


MIL_CASES = [  # (name, ctor kwargs, 4-D input?)
    ("mean", dict(pooling_mode="mean"), False), ("max", dict(pooling_mode="max"), False),
    ("attention", dict(pooling_mode="attention"), False), ("cls", dict(pooling_mode="cls_token"), False),
    ("cls_shared_pre", dict(pooling_mode="cls_token", separate_video_attention=False, normalization_strategy="pre_norm"), False),
    ("mean_cls", dict(pooling_mode="mean+cls_token"), False), ("attn_cls", dict(pooling_mode="attention+cls_token"), False),
    ("attention4d", dict(pooling_mode="attention"), True), ("cls4d", dict(pooling_mode="cls_token"), True),
    ("mean4d", dict(pooling_mode="mean"), True),
]


def mil():
    """models/multi_instance_linear_probing.py: every pooling mode on [3, 5, 512] instance embeddings with a ragged mask, and
    the hierarchical modes on [3, 5, 7, 512] patch tokens; name-seeded weights."""
    M = load_by_path("ref_mil", "models/multi_instance_linear_probing.py")
    g = torch.Generator().manual_seed(13)
    x3 = torch.randn(3, 5, 512, generator=g)
    x4 = torch.randn(3, 5, 7, 512, generator=g)
    mask = torch.tensor([[1, 1, 1, 1, 1], [1, 1, 0, 1, 0], [1, 0, 0, 0, 0]], dtype=torch.bool)
    out = {"x3": np_(x3), "x4": np_(x4), "mask": np_(mask)}
    import warnings
    for name, kw, four in MIL_CASES:
        m = init_by_name(M.MultiInstanceLinearProbing(512, {"a": 3, "b": 1}, **kw)).eval()
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = m(x4 if four else x3, mask=mask)
        out[f"{name}_a"], out[f"{name}_b"] = np_(y["a"]), np_(y["b"])
        out[f"{name}_keys"] = np.array(sorted(m.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "mil.npz"), **out)


MIL_VIEW_CASES = [  # view-angle embeddings (num_view_classes = 4; id 4 = PAD), 3-D and 4-D inputs
    ("view_mean", dict(pooling_mode="mean", num_view_classes=4), False),
    ("view_attention", dict(pooling_mode="attention", num_view_classes=4), False),
    ("view_cls4d", dict(pooling_mode="cls_token", num_view_classes=4), True),
    ("view_attention4d", dict(pooling_mode="attention", num_view_classes=4), True),
    ("view_mean4d", dict(pooling_mode="mean", num_view_classes=4), True),
]


def mil_view():
    """models/multi_instance_linear_probing.py:142-145,265-269: the view-angle embedding add in front of every pooling rule."""
    M = load_by_path("ref_mil", "models/multi_instance_linear_probing.py")
    g0 = np.load(os.path.join(OUT, "mil.npz"))
    x3, x4, mask = torch.from_numpy(g0["x3"]), torch.from_numpy(g0["x4"]), torch.from_numpy(g0["mask"])
    view_ids = torch.tensor([[0, 1, 2, 3, 4], [4, 4, 1, 0, 2], [3, 4, 4, 4, 4]], dtype=torch.long)
    out = {"view_ids": np_(view_ids)}
    import warnings
    for name, kw, four in MIL_VIEW_CASES:
        m = init_by_name(M.MultiInstanceLinearProbing(512, {"a": 3, "b": 1}, **kw)).eval()
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = m(x4 if four else x3, mask=mask, view_ids=view_ids)
        out[f"{name}_a"], out[f"{name}_b"] = np_(y["a"]), np_(y["b"])
        out[f"{name}_keys"] = np.array(sorted(m.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "mil_view.npz"), **out)


def pools_masked():
    """key_padding_mask variants (True = ignored): models/attention_pool.py:61-101 (AttentionPool), :162-197
    (AttentionPoolWithCLS, mask extended by an unmasked CLS column) and the masked EnhancedVideoAggregator
    (models/video_aggregator.py:105-159) incl. a study with a single valid video."""
    AP = load_by_path("ref_attention_pool", "models/attention_pool.py")
    AG = load_by_path("ref_video_aggregator", "models/video_aggregator.py")
    out = {}
    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 50, 512, generator=g)
    mask = torch.zeros(3, 50, dtype=torch.bool)
    mask[0, 30:] = True
    mask[1, ::3] = True
    mask[2, 1:] = True   # one valid token
    ap = init_by_name(AP.AttentionPool(512, num_heads=8, dropout=0.0)).eval()
    apc = init_by_name(AP.AttentionPoolWithCLS(512, num_heads=8, num_layers=1, dropout=0.0)).eval()
    with torch.no_grad():
        out.update(x=np_(x), mask=np_(mask), ap_y=np_(ap(x, mask=mask)), apcls_y=np_(apc(x, mask=mask)))
    ag = init_by_name(AG.EnhancedVideoAggregator(512, num_heads=4, dropout=0.0, aggregator_depth=2)).eval()
    for n in (4, 8):
        xn = torch.randn(3, n, 512, generator=g)
        mn = torch.zeros(3, n, dtype=torch.bool)
        mn[0, n // 2:] = True
        mn[2, 1:] = True
        with torch.no_grad():
            out[f"ag_x{n}"], out[f"ag_m{n}"], out[f"ag_y{n}"] = np_(xn), np_(mn), np_(ag(xn, mask=mn))
    np.savez_compressed(os.path.join(OUT, "pools_masked.npz"), **out)


def loss_zoo():
    """Remaining loss zoo: auto-balance / entropy-regularised contrastive.SigLIPLoss,
    siglip_pairwise.SiglipPairwiseFeatureLoss, multi_positive_infonce.MultiPositiveInfoNCELoss, and the runner's
    abnormal-margin column offset (runner.py:1265-1274) through WeightedSigLIPLoss — on the [8 x 13] multi-positive case."""
    import torch.nn.functional as F
    from utils.loss import contrastive as C
    from utils.loss import siglip_pairwise as SP
    from utils.loss.multi_positive_infonce import MultiPositiveInfoNCELoss
    from utils.loss.weighted_siglip import WeightedSigLIPLoss

    out = {}
    g = torch.Generator().manual_seed(7)
    v = torch.randn(8, 512, generator=g).requires_grad_(True)
    t = torch.randn(13, 512, generator=g).requires_grad_(True)
    lt = torch.log(torch.tensor(0.07)).requires_grad_(True)
    mask = torch.zeros(8, 13)
    for i in range(8):
        for j in torch.randperm(13, generator=g)[: 1 + i % 3].tolist():
            mask[i, j] = 1.0
    w = mask * (1.0 + 9.0 * torch.rand(8, 13, generator=g))
    out.update(v=np_(v), t=np_(t), mask=np_(mask), w=np_(w))

    def run(name, fn, bias_of=None):
        for x in (v, t, lt):
            x.grad = None
        loss = fn()
        loss.backward()
        out[f"{name}_loss"], out[f"{name}_dv"], out[f"{name}_dt"], out[f"{name}_dlt"] = np_(loss), np_(v.grad), np_(t.grad), np_(lt.grad)
        if bias_of is not None:
            out[f"{name}_dbias"] = np_(bias_of.bias.grad)

    m = C.SigLIPLoss(auto_balance=True)
    run("bce_auto", lambda: m(v, t, lt, pos_mask=mask, pos_weights=w), m)
    m2 = C.SigLIPLoss(entropy_regularization=True, entropy_weight=0.1, min_entropy_threshold=3.0)  # ln 13 < 3: always active
    run("bce_ent", lambda: m2(v, t, lt, pos_mask=mask, pos_weights=w), m2)
    d = m2.get_entropy_diagnostics()
    out["bce_ent_diag"] = np.array([d[k] for k in ("entropy_mean", "entropy_min", "entropy_max", "entropy_normalized",
                                                    "entropy_deficit", "bce_loss", "entropy_loss")])
    m3 = C.SigLIPLoss(entropy_regularization=True, entropy_weight=0.1, min_entropy_threshold=0.01)  # inactive branch
    run("bce_ent_off", lambda: m3(v, t, lt, pos_mask=mask, pos_weights=w), m3)
    run("spf", lambda: SP.SiglipPairwiseFeatureLoss()(v, t, lt, mask, w))
    run("spf_noweights", lambda: SP.SiglipPairwiseFeatureLoss(use_positive_weights=False, positive_weight=2.0, negative_weight=0.5)(v, t, lt, mask, w))
    run("spf_auto", lambda: SP.SiglipPairwiseFeatureLoss(auto_positive_weight=True)(v, t, lt, mask, w))
    run("spf_ent", lambda: SP.SiglipPairwiseFeatureLoss(entropy_regularization=True, entropy_weight=0.2, min_entropy_threshold=3.0)(v, t, lt, mask, w))
    # runner inline branch with the abnormal margin on some text columns
    abn = torch.zeros(13)
    abn[[1, 4, 9]] = 1.0

    def margin_loss():
        sim = F.normalize(v, dim=1) @ F.normalize(t, dim=1).t()
        logits = sim * torch.sigmoid(sim) / torch.exp(lt.float()) + abn.unsqueeze(0) * 0.35
        return WeightedSigLIPLoss()(logits=logits, positive_weights=mask * w)

    run("margin_wce", margin_loss)
    out["abn"] = np_(abn)
    # multi-positive InfoNCE on logits; row 5 and column 2 lose their positives
    mask2 = mask.clone()
    mask2[5] = 0.0
    mask2[:, 2] = 0.0
    w2 = 1.0 + 9.0 * torch.rand(8, 13, generator=g)
    logits0 = (F.normalize(v.detach(), dim=1) @ F.normalize(t.detach(), dim=1).t() / 0.07)
    out.update(nce_logits=np_(logits0), nce_mask=np_(mask2), nce_w=np_(w2))
    for name, kw, pw in (("nce_mean", {}, w2), ("nce_sum", {"reduction": "sum"}, w2), ("nce_nopw", {}, None),
                         ("nce_imp", {"use_importance_weighting": True}, w2),
                         ("nce_imp_sum", {"use_importance_weighting": True, "reduction": "sum"}, None)):
        z = logits0.clone().requires_grad_(True)
        loss = MultiPositiveInfoNCELoss(**kw)(z, mask2, pw)
        loss.backward()
        out[f"{name}_loss"], out[f"{name}_dz"] = np_(loss), np_(z.grad)
    z = logits0.clone().requires_grad_(True)
    out["nce_empty_loss"] = np_(MultiPositiveInfoNCELoss()(z, torch.zeros_like(mask2), None))
    np.savez_compressed(os.path.join(OUT, "loss_zoo.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = sys.argv[1:]
    for fn in (losses, rope, pools, bert, retrieval, loss_zoo, pools2, mil, mil_view, pools_masked, losses_ddp):
        if only and fn.__name__ not in only:
            continue
        fn()
        print("wrote", fn.__name__)
