"""Step tail and whole-step parity: HIP AdamW / gradient clipping against torch.optim / clip_grad_norm_, and
`_train_step` loss curves against the CPU oracle trained with torch.optim.AdamW on the same seeded batches
(reference ordering: runners/video_constrative_learning_runner.py:1387-1424)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (7,), (96, 3, 3, 7, 7), (4096,), (4097,), (768, 768), (30522, 8), (3, 5, 1)]


def _params(seed, device):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(device)) for s in SHAPES]


def _set_grads(ps, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    for p in ps:
        p.grad = (torch.randn(p.shape, generator=g) * scale).to(p.device)


@pytest.mark.parametrize("adamw", [True, False])
def test_adam_matches_torch(adamw):
    from deepcoro_clip_amd.optim import HipAdamW

    ref_p, hip_p = _params(0, "cpu"), _params(0, "cuda")
    mk = lambda ps: [{"params": ps[:3], "lr": 1e-2, "weight_decay": 0.05}, {"params": ps[3:], "lr": 3e-3, "weight_decay": 0.0}]  # noqa: E731
    ref = (torch.optim.AdamW if adamw else torch.optim.Adam)(mk(ref_p), lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    hip = HipAdamW(mk(hip_p), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, adamw=adamw)
    for step in range(6):
        _set_grads(ref_p, 10 + step)
        _set_grads(hip_p, 10 + step)
        if step == 3:  # a parameter without gradient is skipped, like torch
            ref_p[1].grad = None
            hip_p[1].grad = None
        ref.step()
        hip.step()
    for a, b in zip(hip_p, ref_p):
        assert torch.allclose(a.detach().cpu(), b.detach(), rtol=2e-5, atol=2e-6), (a.shape, (a.cpu() - b).abs().max())
    st_h, st_r = hip.state[hip_p[5]], ref.state[ref_p[5]]
    assert float(st_h["step"]) == float(st_r["step"]) == 6
    assert torch.allclose(st_h["exp_avg"].cpu(), st_r["exp_avg"], rtol=1e-5, atol=1e-7)
    assert torch.allclose(st_h["exp_avg_sq"].cpu(), st_r["exp_avg_sq"], rtol=1e-5, atol=1e-9)


def test_adopt_keeps_groups_and_state():
    from deepcoro_clip_amd.optim import HipAdamW

    ref_p, hip_p = _params(1, "cpu"), _params(1, "cuda")
    ref = torch.optim.AdamW(ref_p, lr=1e-2)
    base = torch.optim.AdamW(hip_p, lr=1e-2)
    _set_grads(ref_p, 5)
    _set_grads(hip_p, 5)
    ref.step()
    base.step()  # torch's own CUDA step creates the state ...
    hip = HipAdamW.adopt(base)  # ... which the HIP optimizer continues from
    assert all(a is b for a, b in zip(hip.param_groups, base.param_groups)) and hip.state is base.state
    _set_grads(ref_p, 6)
    _set_grads(hip_p, 6)
    ref.step()
    hip.step()
    for a, b in zip(hip_p, ref_p):
        assert torch.allclose(a.detach().cpu(), b.detach(), rtol=2e-5, atol=2e-6)
    with pytest.raises(Exception):
        HipAdamW.adopt(torch.optim.SGD(hip_p, lr=0.1))


@pytest.mark.parametrize("scale,max_norm", [(1.0, 5.0), (1e-4, 5.0), (30.0, 1.0)])
def test_grad_clip_matches_torch(scale, max_norm):
    from deepcoro_clip_amd.optim import GradClipper

    ref_p, hip_p = _params(2, "cpu"), _params(2, "cuda")
    _set_grads(ref_p, 3, scale)
    _set_grads(hip_p, 3, scale)
    hip_p[2].grad = None
    ref_p[2].grad = None
    total = torch.nn.utils.clip_grad_norm_(ref_p, max_norm)
    norm = GradClipper(hip_p).clip_(max_norm)
    assert norm.is_cuda and abs(norm.item() - total.item()) < 1e-5 * max(1.0, total.item())
    for a, b in zip(hip_p, ref_p):
        if b.grad is not None:
            assert torch.allclose(a.grad.cpu(), b.grad, rtol=1e-5, atol=1e-9)


def _oracle_trainer(vo, to, log_temp0):
    lt = torch.nn.Parameter(torch.log(torch.tensor([log_temp0])))
    groups = [{"params": list(vo.model.parameters()), "lr": 1e-4, "weight_decay": 0.01},
              {"params": list(vo.aggregator.parameters()), "lr": 2e-4, "weight_decay": 0.01},
              {"params": list(to.parameters()), "lr": 2e-5, "weight_decay": 0.01},
              {"params": [lt], "lr": 1e-4}]
    return lt, torch.optim.AdamW(groups, lr=1e-4)


@pytest.mark.parametrize("dtype,steps,tol", [(torch.float32, 8, 1e-4), (torch.bfloat16, 20, 1e-2)])
def test_train_step_loss_curve_vs_oracle(dtype, steps, tol):
    """Same weights, same batches, same optimizer on both sides: loss per step within north_star's tolerance
    (1e-4 fp32 / 1e-2 bf16), metric keys of the reference, embeddings' argmax identical at step 0."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import make_tiny_setup
    from oracle.losses import contrastive_loss, step_metrics

    B = 4
    runner, (vo, to), _ = make_tiny_setup(dtype, B=B)
    lt, opt = _oracle_trainer(vo, to, 0.07)
    all_params = list(vo.parameters()) + list(to.parameters())
    g = torch.Generator().manual_seed(99)
    worst, n_decided, cos_dev = 0.0, 0, 0.0
    for step in range(steps):
        videos = torch.randn(B, 4, 64, 64, 3, generator=g)
        ids = torch.randint(0, 512, (B, 16), generator=g)
        mask = torch.ones(B, 16, dtype=torch.int64)
        mask[1, 11:] = 0  # one padded sequence
        opt.zero_grad(set_to_none=True)
        v, t = vo(videos), to(ids, mask)
        ref = contrastive_loss(v, t, lt)
        ref_m = step_metrics(v.detach(), t.detach(), lt.detach())
        ref.backward()
        torch.nn.utils.clip_grad_norm_(all_params, 5.0)
        opt.step()
        m, emb = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
        err = abs(m["loss"] - ref.item()) / max(1.0, abs(ref.item()))
        worst = max(worst, err)
        assert err < tol, (step, m["loss"], ref.item())
        if step == 0:
            for key in ("loss", "temperature", "contrastive/clip_loss", "video_norm", "text_norm", "alignment_score",
                        "alignment_logprob", "alignment_cosine", "lr/video_backbone", "lr/text_encoder", "lr/temperature"):
                assert key in m, key
            assert abs(m["temperature"] - math.exp(lt.item())) < 1e-6  # reported after the update, like the reference
            mt = 1e-3 if dtype == torch.float32 else 3e-2
            assert abs(m["video_norm"] - ref_m["video_norm"]) < mt * max(1, ref_m["video_norm"])
            assert abs(m["text_norm"] - ref_m["text_norm"]) < mt * max(1, ref_m["text_norm"])
            assert abs(m["alignment_cosine"] - ref_m["alignment_cosine"]) < mt
            assert abs(m["alignment_score"] - ref_m["alignment_score"]) < mt
            assert emb["video_embeddings"].shape == (B, 512) and emb["video_embeddings"].dtype == torch.float32
            if dtype == torch.float32:
                sim = torch.nn.functional.normalize(emb["video_embeddings"], dim=1) @ torch.nn.functional.normalize(emb["text_embeddings"], dim=1).t()
                sim_ref = torch.nn.functional.normalize(v.detach(), dim=1) @ torch.nn.functional.normalize(t.detach(), dim=1).t()
                assert torch.equal(sim.argmax(1).cpu(), sim_ref.argmax(1))
        if dtype == torch.bfloat16:
            # north_star: embedding argmax bit-exact.  In bf16 a row whose best and second-best cosine are closer than bf16's
            # own noise on a cosine (1e-2 class; measured max |cosine - oracle| 1.9e-3 at step 0) has no defined argmax;
            # every row with a clear winner must agree, at every step of the curve (the embeddings separate as training goes on)
            sim = torch.nn.functional.normalize(emb["video_embeddings"], dim=1) @ torch.nn.functional.normalize(emb["text_embeddings"], dim=1).t()
            sim_ref = torch.nn.functional.normalize(v.detach(), dim=1) @ torch.nn.functional.normalize(t.detach(), dim=1).t()
            top2 = sim_ref.topk(2, dim=1).values
            decided = (top2[:, 0] - top2[:, 1]) > 1e-2
            assert torch.equal(sim.argmax(1).cpu()[decided], sim_ref.argmax(1)[decided]), (step, sim.cpu(), sim_ref)
            n_decided += int(decided.sum())
            cos_dev = max(cos_dev, float((sim.cpu() - sim_ref).abs().max()))
    # parameters stay together too (fp32): temperature after `steps` updates
    if dtype == torch.float32:
        assert abs(runner.log_temp.item() - lt.item()) < 1e-5
    print(f"[loss-curve] {dtype} worst relative deviation over {steps} steps: {worst:.2e}")
    if dtype == torch.bfloat16:
        print(f"[loss-curve] bf16 argmax: {n_decided}/{B * steps} rows with a margin above 1e-2, all equal to the oracle's; "
              f"max |cosine - oracle| over the curve {cos_dev:.2e}")



def test_gradient_accumulation_and_val_step():
    """accum=2: the optimizer moves only on the boundary step and sees the mean gradient (runner.py:1223-1224,
    1387-1424 of the reference); _val_step leaves parameters and grads alone and drops the lr/ keys."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import make_tiny_setup
    from oracle.losses import contrastive_loss

    runner, (vo, to), (videos, ids, mask) = make_tiny_setup(torch.float32, B=2)
    runner.config.gradient_accumulation_steps = 2
    lt, opt = _oracle_trainer(vo, to, 0.07)
    g = torch.Generator().manual_seed(5)
    batches = [(torch.randn(2, 4, 64, 64, 3, generator=g), torch.randint(0, 512, (2, 16), generator=g)) for _ in range(2)]
    p0 = runner.video_encoder.model.blocks[0].attn.qkv.weight.detach().clone()
    opt.zero_grad()
    for i, (vid, tok) in enumerate(batches):
        (contrastive_loss(vo(vid), to(tok, mask), lt) / 2).backward()
        m, _ = runner._train_step(videos=vid.cuda(), input_ids=tok.cuda(), attention_mask=mask.cuda())
        changed = not torch.equal(runner.video_encoder.model.blocks[0].attn.qkv.weight.detach(), p0)
        assert changed == (i == 1)
    torch.nn.utils.clip_grad_norm_(list(vo.parameters()) + list(to.parameters()), 5.0)
    opt.step()
    w_ref = vo.model.blocks[0].attn.qkv.weight.detach()
    w_hip = runner.video_encoder.model.blocks[0].attn.qkv.weight.detach().cpu()
    assert torch.allclose(w_hip, w_ref, atol=2e-6), (w_hip - w_ref).abs().max()
    before = w_hip.clone()
    vm, emb = runner._val_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
    assert not any(k.startswith("lr/") for k in vm) and math.isfinite(vm["loss"])
    with torch.no_grad():
        ref = contrastive_loss(vo(videos), to(ids, mask), lt).item()
    assert abs(vm["loss"] - ref) < 1e-4 * max(1.0, abs(ref))
    assert torch.equal(runner.video_encoder.model.blocks[0].attn.qkv.weight.detach().cpu(), before)


def test_baseline_config0_geometry_steps():
    """BASELINE configs[0]: single-video CLIP, 4 frames 112 x 112, 32-token text, batch 8, fp32 — three optimizer steps
    against the CPU oracle (text tower shortened to 2 layers to keep the CPU side in seconds)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import make_tiny_setup
    from oracle.losses import contrastive_loss

    B = 8
    runner, (vo, to), _ = make_tiny_setup(torch.float32, B=B, T=4, S=112, seq=32)
    lt, opt = _oracle_trainer(vo, to, 0.07)
    all_params = list(vo.parameters()) + list(to.parameters())
    g = torch.Generator().manual_seed(7)
    for step in range(3):
        videos = torch.randn(B, 4, 112, 112, 3, generator=g)
        ids = torch.randint(0, 512, (B, 32), generator=g)
        mask = torch.ones(B, 32, dtype=torch.int64)
        opt.zero_grad(set_to_none=True)
        ref = contrastive_loss(vo(videos), to(ids, mask), lt)
        ref.backward()
        torch.nn.utils.clip_grad_norm_(all_params, 5.0)
        opt.step()
        m, emb = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
        assert abs(m["loss"] - ref.item()) < 1e-4 * max(1.0, abs(ref.item())), (step, m["loss"], ref.item())
    assert emb["video_embeddings"].shape == (B, 512)


def test_step_metrics_resolve_lazily_with_reference_keys():
    """`_train_step` returns the reference's metric dict (same keys, float values), filled on first access."""
    import json

    import __graft_entry__ as ge
    from deepcoro_clip_amd.runner import StepMetrics

    runner, _, (videos, ids, mask) = ge.make_tiny_setup(torch.float32)
    m, emb = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
    assert isinstance(m, dict) and isinstance(m, StepMetrics)
    torch.cuda.synchronize()
    assert m.ready()
    keys = set(m)
    assert {"loss", "temperature", "contrastive/clip_loss", "video_norm", "text_norm", "alignment_score", "alignment_logprob",
            "alignment_cosine", "lr/video_backbone", "lr/text_encoder", "lr/temperature"} <= keys
    assert all(isinstance(v, float) for v in m.values())
    assert m["loss"] == m["contrastive/clip_loss"] and math.isfinite(m["loss"])
    plain = dict(m)
    assert plain == m and {**m} == plain and json.loads(json.dumps(m)) == plain and m.copy() == plain
    assert m.get("nope", 7) == 7 and "loss" in m and len(m) == len(plain)
    assert emb["video_embeddings"].shape == (2, 512) and emb["text_embeddings"].shape == (2, 512)


@pytest.mark.parametrize("p_drop", [0.0, 0.25])
def test_hip_graph_step_matches_eager_steps(p_drop):
    """Opt-in HIP-graph replay of forward + loss + backward + clip (runner._graph_train_step): same losses and parameters
    as the eager step on changing batches; warm-up calls run eagerly, the 4th call captures.  With dropout the masks come
    from (host seed of the call site, DEVICE step counter), so a replay draws the masks the eager step would draw."""
    import __graft_entry__ as ge

    def run(use_graph):
        runner, _, (videos, ids, mask) = ge.make_tiny_setup(torch.float32, seed=3)
        runner._use_step_graph = use_graph
        runner.video_encoder.dropout = p_drop
        g = torch.Generator().manual_seed(11)
        losses = []
        for i in range(8):
            v = (videos + 0.05 * torch.randn(videos.shape, generator=g)).cuda()
            idb = torch.randint(0, 512, ids.shape, generator=g).cuda()
            m, emb = runner._train_step(videos=v, input_ids=idb, attention_mask=mask.cuda())
            losses.append(m["loss"])
            assert emb["video_embeddings"].shape == (2, 512)
        torch.cuda.synchronize()
        return runner, losses

    eager, le = run(False)
    graph, lg = run(True)
    assert any(st["graph"] is not None for st in graph._graphs.values()), "the step was never captured"
    # split-K atomics make two runs of the SAME path differ in the last bits; after 8 optimizer steps that is ~1e-5
    assert max(abs(a - b) for a, b in zip(le, lg)) < 2e-4 * max(1.0, max(abs(x) for x in le)), (le, lg)
    for (n, p), (_, q) in zip(eager.text_encoder.named_parameters(), graph.text_encoder.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-3, atol=1e-5), n
    assert eager.step == graph.step == 8


def test_ddp_wrapped_step_writes_gradients_into_the_arena_and_matches_plain_steps():
    """HipDDP (world size 1 here): weight / bias gradients are produced straight in the flat arena (_grad_slots), the rest
    is moved there by the post-accumulate hook; losses and parameters track the unwrapped runner."""
    import __graft_entry__ as ge
    from deepcoro_clip_amd.ddp import HipDDP

    def run(wrap):
        runner, _, (videos, ids, mask) = ge.make_tiny_setup(torch.float32, seed=5)
        if wrap:
            runner.video_encoder, runner.text_encoder = HipDDP(runner.video_encoder), HipDDP(runner.text_encoder)
        losses = []
        for _ in range(4):
            m, _ = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
            losses.append(m["loss"])
        torch.cuda.synchronize()
        return runner, losses

    plain, lp = run(False)
    ddp, ld = run(True)
    assert max(abs(a - b) for a, b in zip(lp, ld)) < 2e-4 * max(1.0, max(abs(x) for x in lp)), (lp, ld)
    te = ddp.text_encoder
    lo, hi = te._arena.data_ptr(), te._arena.data_ptr() + te._arena.numel() * 4
    n_in = sum(1 for p in te.module.parameters() if p.grad is not None and lo <= p.grad.data_ptr() < hi)
    n_all = sum(1 for p in te.module.parameters() if p.requires_grad)
    assert n_in == n_all, (n_in, n_all)   # every gradient ended up as a view into the flat arena
    from deepcoro_clip_amd import _grad_slots
    n_lin = sum(2 for m in te.module.modules() if isinstance(m, torch.nn.Linear))
    assert len(_grad_slots._TAKEN) >= n_lin, (len(_grad_slots._TAKEN), n_lin)  # Linear weights + biases were written in place
    for (n, p), (_, q) in zip(plain.text_encoder.named_parameters(), te.module.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-3, atol=1e-5), n


@pytest.mark.parametrize("loss_name", ["contrastive", "siglip"])
def test_embedding_cache_two_pass_equals_one_shot(loss_name):
    """BASELINE configs[3] mechanism (global batch 4096 = 512 studies per rank through an embedding cache): B = 32 in 4
    micro-batches of 8 against the one-shot step — same loss (<= 1e-6) and the same parameter gradients (fp32 compute)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import make_tiny_setup

    B = 32
    ra, _, (videos, ids, mask) = make_tiny_setup(torch.float32, B=B, loss_name=loss_name)
    rb, _, _ = make_tiny_setup(torch.float32, B=B, loss_name=loss_name)
    mask[3, 9:] = 0
    args = (videos.cuda(), ids.cuda(), mask.cuda(), None, None, None, 1)
    la, va, ta, _ = ra._forward_backward(*args)
    lb, vb, tb, _ = rb._forward_backward_cached(*args, 8)
    assert abs(la.item() - lb.item()) <= 1e-6 * max(1.0, abs(la.item())), (la.item(), lb.item())
    assert torch.allclose(va.detach(), vb, atol=1e-6) and torch.allclose(ta.detach(), tb, atol=1e-6)
    worst = 0.0
    pa = dict(ra.video_encoder.named_parameters()) | {"text." + k: v for k, v in ra.text_encoder.named_parameters()}
    pb = dict(rb.video_encoder.named_parameters()) | {"text." + k: v for k, v in rb.text_encoder.named_parameters()}
    norms = sorted(float(p.grad.norm()) for p in pa.values() if p.grad is not None)
    floor = 1e-3 * norms[len(norms) // 2]
    for n, p in pa.items():
        if p.grad is None:
            assert pb[n].grad is None or float(pb[n].grad.abs().max()) == 0.0, n
            continue
        if "pool_k.norm_act.0.bias" in n:
            continue  # analytically zero gradient (a uniform shift of a softmax row): both sides hold summation noise
        err = float((p.grad - pb[n].grad).norm() / p.grad.norm().clamp_min(floor))
        worst = max(worst, err)
        assert err < 2e-4, (n, err)
    assert abs(ra.log_temp.grad.item() - rb.log_temp.grad.item()) < 1e-5 * max(1.0, abs(ra.log_temp.grad.item()))
    print(f"[embedding cache] worst relative gradient deviation {worst:.2e}")
    # through the public entry point: config switch, optimizer step, metric keys
    rb.config.embedding_cache_micro_batch = 8
    m, emb = rb._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
    assert math.isfinite(m["loss"]) and emb["video_embeddings"].shape == (B, 512) and "alignment_score" in m
    # the step graph never replaces the two-pass step (it would run the whole batch at once: the memory bound is the point)
    rb._use_step_graph = True
    calls = []
    orig = rb._forward_backward_cached
    rb._forward_backward_cached = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    for _ in range(5):   # more calls than the graph path's eager warm-up
        rb._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
    assert len(calls) == 5 and not any(st.get("graph") is not None for st in rb._graphs.values())


def test_embedding_cache_redraws_the_same_masks():
    """With every stochastic layer on (dropout 0.2, stochastic depth 0.2) the re-forward of pass 3 must reproduce the
    cached embeddings of pass 1 exactly (same dropout call-site seeds + step counter, rewound device generator)."""
    from deepcoro_clip_amd.losses import Loss
    from deepcoro_clip_amd.optim import HipAdamW
    from deepcoro_clip_amd.registry import LossRegistry, register_submodules
    from deepcoro_clip_amd.runner import VideoContrastiveLearningRunner
    from deepcoro_clip_amd.text_encoder import TextEncoder
    from deepcoro_clip_amd.video_encoder import VideoEncoder

    register_submodules("deepcoro_clip_amd")
    torch.manual_seed(0)
    ve = VideoEncoder(num_frames=4, spatial_size=(64, 64), pretrained=False, freeze_ratio=1.0, dropout=0.2,
                      token_pooling_mode="attention", attention_pool_dropout=0.1, compute_dtype=torch.float32).cuda().train()
    te = TextEncoder(dropout=0.2, freeze_ratio=0.0, compute_dtype=torch.float32,
                     bert_config=dict(vocab_size=512, num_hidden_layers=2, max_position_embeddings=64)).cuda().train()
    lt = torch.nn.Parameter(torch.log(torch.tensor([0.07], device="cuda")))
    opt = HipAdamW([{"params": list(ve.model.parameters()) + list(ve.attention_pool.parameters()) + list(ve.aggregator.parameters())
                     + list(te.parameters()) + [lt], "lr": 1e-4}], lr=1e-4)
    cfg = type("C", (), dict(recall_k=[1], ndcg_k=[5], device="cuda:0", world_size=1, gradient_accumulation_steps=1,
                             max_grad_norm=5.0, is_ref_device=True, loss_name="siglip", scheduler_name="", temperature=0.07,
                             embedding_cache_micro_batch=4))()
    r = VideoContrastiveLearningRunner(config=cfg, video_encoder=ve, text_encoder=te, optimizer=opt, log_temp=lt,
                                       loss_fn=Loss(loss_type=LossRegistry.get("siglip")()))
    r._cache_check = True
    g = torch.Generator().manual_seed(3)
    videos = torch.randn(12, 2, 4, 64, 64, 3, generator=g).cuda()
    ids = torch.randint(0, 512, (12, 16), generator=g).cuda()
    mask = torch.ones(12, 16, dtype=torch.int64).cuda()
    for _ in range(2):
        m, _ = r._train_step(videos=videos, input_ids=ids, attention_mask=mask)
        assert math.isfinite(m["loss"])
    assert r._cache_maxdiff == 0.0, r._cache_maxdiff


@pytest.mark.parametrize("weighted,zero_weights", [(False, False), (True, False), (True, True)])
def test_multi_positive_alignment_metrics_match_reference_formulas(weighted, zero_weights):
    """The runner's inline multi-positive SigLIP branch (reference runner.py:1256-1313): alignment_logprob = mean over
    rows with positives of sum_j log_softmax(logits)_ij * w_ij / sum_j w_ij with w = positive_weights * mask (the plain
    mask when every weight is zero), alignment_cosine = mean cosine over the positives; M != B texts."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import make_tiny_setup
    import torch.nn.functional as F

    B, M = 4, 6
    runner, _, (videos, _, _) = make_tiny_setup(torch.float32, B=B, loss_name="siglip")
    runner.use_weighted_siglip = weighted
    if weighted:
        from deepcoro_clip_amd.losses import WeightedSigLIPLoss
        runner.weighted_siglip_loss = WeightedSigLIPLoss()
    g = torch.Generator().manual_seed(21)
    ids = torch.randint(0, 512, (M, 16), generator=g)
    mask = torch.ones(M, 16, dtype=torch.int64)
    pm = torch.zeros(B, M)
    pm[0, 0] = pm[0, 4] = pm[1, 1] = pm[2, 2] = pm[2, 3] = pm[2, 5] = 1.0  # row 3 has no positive
    pw = torch.zeros(B, M) if zero_weights else (torch.rand(B, M, generator=g) * 9 + 1)
    m, emb = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda(),
                                positive_mask=pm.cuda(), positive_weights=pw.cuda())
    v, t = emb["video_embeddings"].cpu().double(), emb["text_embeddings"].cpu().double()
    sim = F.normalize(v, dim=1) @ F.normalize(t, dim=1).t()
    logits = sim * torch.sigmoid(sim) / 0.07
    logp = F.log_softmax(logits, dim=1)
    wmask = (pw * pm).double() if torch.count_nonzero(pw) > 0 else pm.double()
    rs = wmask.sum(1)
    valid = rs > 0
    ref_lp = ((logp * (wmask / rs.clamp_min(1e-30)[:, None])).sum(1)[valid]).mean().item()
    ref_cos = sim[pm.bool()].mean().item()
    assert abs(m["alignment_logprob"] - ref_lp) < 1e-4 * max(1.0, abs(ref_lp)), (m["alignment_logprob"], ref_lp)
    assert abs(m["alignment_score"] - math.exp(ref_lp)) < 1e-4
    assert abs(m["alignment_cosine"] - ref_cos) < 1e-5
    # loss value of the branch against the same formulas
    if weighted:
        posw = wmask
        row = -(logp * posw).sum(1) / posw.sum(1).clamp_min(1e-6)
        logp_c = F.log_softmax(logits.t(), dim=1)
        col = -(logp_c * posw.t()).sum(1) / posw.t().sum(1).clamp_min(1e-6)
        ref_loss = 0.5 * (row.mean() + col.mean()).item()  # utils/loss/weighted_siglip.py:41-51: plain means
        assert abs(m["loss"] - ref_loss) < 2e-4 * max(1.0, abs(ref_loss)), (m["loss"], ref_loss)
