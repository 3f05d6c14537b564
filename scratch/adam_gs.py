import sys; sys.path.insert(0, '.')
import torch
from deepcoro_clip_amd.optim import HipAdamW, GradClipper
torch.manual_seed(0)
def mk(): 
    torch.manual_seed(0)
    return [torch.nn.Parameter(torch.randn(5000, device='cuda')), torch.nn.Parameter(torch.randn(300, 7, device='cuda'))]
a, b = mk(), mk()
oa, ob = HipAdamW(a, lr=1e-2), HipAdamW(b, lr=1e-2)
ca, cb = GradClipper(a), GradClipper(b)
for step in range(3):
    g = [torch.randn_like(p) * 3 for p in a]
    for p, q, gg in zip(a, b, g): p.grad = gg.clone(); q.grad = gg.clone()
    ca.clip_(1.0, scale=True); oa.step()
    cb.clip_(1.0, scale=False); ob.step(grad_scale=cb.coef, scaled={id(p) for p in b})
    print(step, ca.coef.item(), cb.coef.item(), [(p - q).abs().max().item() for p, q in zip(a, b)])
