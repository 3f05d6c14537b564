import sys; sys.path.insert(0, '.')
import torch
from __graft_entry__ import make_tiny_setup
import deepcoro_clip_amd.optim as O
def run(fuse):
    runner, _, (videos, ids, mask) = make_tiny_setup(torch.float32, B=2)
    if not fuse:
        orig = O.GradClipper.clip_
        def clip_always_scale(self, max_norm, scale=True):
            r = orig(self, max_norm, True); self.coef.fill_(1.0); return r
        O.GradClipper.clip_ = clip_always_scale
    outs = []
    for s in range(2):
        m, _ = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
        outs.append({n: p.detach().clone() for n, p in list(runner.video_encoder.named_parameters()) + list(runner.text_encoder.named_parameters())})
        print("fuse", fuse, "step", s, "loss", m["loss"], "norm", runner._clipper.norm.item())
    if not fuse: O.GradClipper.clip_ = orig
    return outs
a = run(True); b = run(False)
for s in range(2):
    rows = sorted(((a[s][n] - b[s][n]).abs().max().item(), n) for n in a[s])[::-1][:5]
    print("step", s, rows)
