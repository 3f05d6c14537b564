import sys, os, math; sys.path.insert(0, '.')
import torch
from __graft_entry__ import make_tiny_setup
from oracle.losses import contrastive_loss
B = 4
runner, (vo, to), _ = make_tiny_setup(torch.float32, B=B)
lt = torch.nn.Parameter(torch.log(torch.tensor([0.07])))
groups = [{"params": list(vo.model.parameters()), "lr": 1e-4, "weight_decay": 0.01},
          {"params": list(vo.aggregator.parameters()), "lr": 2e-4, "weight_decay": 0.01},
          {"params": list(to.parameters()), "lr": 2e-5, "weight_decay": 0.01}, {"params": [lt], "lr": 1e-4}]
opt = torch.optim.AdamW(groups, lr=1e-4)
allp = list(vo.parameters()) + list(to.parameters())
g = torch.Generator().manual_seed(99)
for step in range(2):
    videos = torch.randn(B, 4, 64, 64, 3, generator=g); ids = torch.randint(0, 512, (B, 16), generator=g)
    mask = torch.ones(B, 16, dtype=torch.int64); mask[1, 11:] = 0
    opt.zero_grad(set_to_none=True)
    ref = contrastive_loss(vo(videos), to(ids, mask), lt); ref.backward()
    gref = {n: p.grad.clone() for n, p in list(vo.named_parameters()) + [("T." + n, p) for n, p in to.named_parameters()] if p.grad is not None}
    n = torch.nn.utils.clip_grad_norm_(allp, 5.0); opt.step()
    m, _ = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
    rows = []
    hp = dict(list(runner.video_encoder.named_parameters()) + [("T." + n, p) for n, p in runner.text_encoder.named_parameters()])
    op = dict(list(vo.named_parameters()) + [("T." + n, p) for n, p in to.named_parameters()])
    for name, p in hp.items():
        d = (p.detach().cpu() - op[name].detach()).abs().max().item()
        gd = (p.grad.cpu() - gref[name]).abs().max().item() if (p.grad is not None and name in gref) else -1
        rows.append((d, gd, name, gref[name].abs().max().item() if name in gref else -1))
    rows.sort(reverse=True)
    print("step", step, "norm", n.item(), runner._clipper.norm.item(), "coef", runner._clipper.coef.item())
    for r in rows[:6]: print("   param diff %.2e grad diff(unscaled hip vs raw ref) %.2e  %s  |gref|max %.2e" % r)
