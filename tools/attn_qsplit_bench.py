"""dK/dV kernel launch geometry: time attn_bwd with the query range split 1..4 ways for the MViT shapes of the bench step."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
B = 32
for (H, Nq, Nk, DQK, DV) in ((1, 25089, 393, 128, 96), (2, 6273, 1569, 160, 96), (2, 6273, 393, 128, 96), (4, 1569, 1569, 160, 96), (4, 1569, 393, 128, 96), (8, 393, 1569, 160, 96), (8, 393, 393, 128, 96)):
    q = torch.randn(B, Nq, H, DQK, device='cuda').bfloat16() * 0.3
    k = torch.randn(B, Nk, H, DQK, device='cuda').bfloat16() * 0.3
    v = torch.randn(B, Nk, H, DV, device='cuda').bfloat16()
    o, lse = K.attn_fwd(q, k, v, scale=0.1, residual=True, res_start=1)
    do = torch.randn_like(o)
    tf = t(lambda: K.attn_fwd(q, k, v, scale=0.1, residual=True, res_start=1))
    line = f"H{H} Nq{Nq} Nk{Nk} D{DQK}/{DV}: fwd {tf:7.1f}  bwd"
    for qs in (1, 2, 3, 4, 6):
        tb = t(lambda: K.attn_bwd(q, k, v, o, do, lse, scale=0.1, residual=True, res_start=1, qsplit=qs))
        line += f"  qs{qs} {tb:7.1f}"
    print(line, flush=True)
