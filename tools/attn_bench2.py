"""fwd / bwd times of the MViT attention shapes of the bench step (B = 32) for the library selected by DCC_LIB."""
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
tot_f = tot_b = 0.0
# (H, Nq, Nk, DQK, DV, launches per step)
for (H, Nq, Nk, DQK, DV, n) in ((1, 25089, 393, 128, 96, 1), (2, 6273, 1569, 160, 96, 1), (2, 6273, 393, 128, 96, 1), (4, 1569, 1569, 160, 96, 1), (4, 1569, 393, 128, 96, 10), (8, 393, 1569, 160, 96, 1), (8, 393, 393, 128, 96, 1)):
    q = torch.randn(B, Nq, H, DQK, device='cuda').bfloat16() * 0.3
    k = torch.randn(B, Nk, H, DQK, device='cuda').bfloat16() * 0.3
    v = torch.randn(B, Nk, H, DV, device='cuda').bfloat16()
    o, lse = K.attn_fwd(q, k, v, scale=0.1, residual=True, res_start=1)
    do = torch.randn_like(o)
    tf = t(lambda: K.attn_fwd(q, k, v, scale=0.1, residual=True, res_start=1))
    tb = t(lambda: K.attn_bwd(q, k, v, o, do, lse, scale=0.1, residual=True, res_start=1))
    tot_f += n * tf; tot_b += n * tb
    print(f"H{H} Nq{Nq} Nk{Nk} D{DQK}/{DV} x{n}: fwd {tf:7.1f}  bwd {tb:7.1f}", flush=True)
print(f"per step: fwd {tot_f/1e3:.2f} ms  bwd {tot_b/1e3:.2f} ms")
