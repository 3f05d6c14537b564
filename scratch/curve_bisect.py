import sys, os, math; sys.path.insert(0, '.')
import torch
from __graft_entry__ import make_tiny_setup
from oracle.losses import contrastive_loss
from deepcoro_clip_amd.optim import HipAdamW
def run(tag, fuse=True, steps=6):
    B = 4
    runner, (vo, to), _ = make_tiny_setup(torch.float32, B=B)
    if not fuse:
        import deepcoro_clip_amd.runner as R
        R_isinstance = None
    lt = torch.nn.Parameter(torch.log(torch.tensor([0.07])))
    groups = [{"params": list(vo.model.parameters()), "lr": 1e-4, "weight_decay": 0.01},
              {"params": list(vo.aggregator.parameters()), "lr": 2e-4, "weight_decay": 0.01},
              {"params": list(to.parameters()), "lr": 2e-5, "weight_decay": 0.01}, {"params": [lt], "lr": 1e-4}]
    opt = torch.optim.AdamW(groups, lr=1e-4)
    allp = list(vo.parameters()) + list(to.parameters())
    g = torch.Generator().manual_seed(99)
    out = []
    for step in range(steps):
        videos = torch.randn(B, 4, 64, 64, 3, generator=g); ids = torch.randint(0, 512, (B, 16), generator=g)
        mask = torch.ones(B, 16, dtype=torch.int64); mask[1, 11:] = 0
        opt.zero_grad(set_to_none=True)
        ref = contrastive_loss(vo(videos), to(ids, mask), lt); ref.backward()
        n = torch.nn.utils.clip_grad_norm_(allp, 5.0); opt.step()
        if not fuse:
            runner.optimizer.__class__ = type("X", (HipAdamW,), {})  # isinstance still true; so monkeypatch clip instead
        m, _ = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
        out.append(f"{abs(m['loss']-ref.item())/max(1,abs(ref.item())):.1e}(n={n.item():.2f},hipn={runner._clipper.norm.item():.2f})")
    print(tag, ' '.join(out), flush=True)
run("fused")
import deepcoro_clip_amd.optim as O
orig = O.GradClipper.clip_
def clip_always_scale(self, max_norm, scale=True):
    r = orig(self, max_norm, True)
    self.coef.fill_(1.0)
    return r
O.GradClipper.clip_ = clip_always_scale
run("unfused")
