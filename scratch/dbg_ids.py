import sys; sys.path.insert(0, '.')
import torch
from __graft_entry__ import make_tiny_setup
runner, (vo, to), (videos, ids, mask) = make_tiny_setup(torch.float32, B=2)
m, _ = runner._train_step(videos=videos.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda())
opt = runner.optimizer
plist = [p for g in opt.param_groups for p in g["params"]]
print("clipped", len(runner._clipped_ids), "opt params", len(plist), "in set", sum(id(p) in runner._clipped_ids for p in plist))
print("grads None", sum(p.grad is None for p in plist), "requires_grad False", sum(not p.requires_grad for p in plist))
import numpy as np
from deepcoro_clip_amd.optim import _DESC
d = opt._table.hosts[opt._table.turn ^ 1].numpy().view(_DESC)
print("flags histogram", np.unique(d["flags"], return_counts=True))
