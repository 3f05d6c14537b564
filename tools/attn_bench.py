import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for (B, H, Nq, Nk, DQK, DV) in ((24, 4, 1569, 393, 128, 96), (24, 2, 6273, 1569, 160, 96), (24, 1, 25089, 393, 128, 96), (24, 4, 1569, 1569, 160, 96), (24, 8, 393, 1569, 160, 96), (24, 8, 393, 393, 128, 96)):
    q = torch.randn(B, Nq, H, DQK, device='cuda').bfloat16() * 0.3
    k = torch.randn(B, Nk, H, DQK, device='cuda').bfloat16() * 0.3
    v = torch.randn(B, Nk, H, DV, device='cuda').bfloat16()
    o, lse = K.attn_fwd(q, k, v, scale=0.1, residual=True, res_start=1)
    do = torch.randn_like(o)
    tf = t(lambda: K.attn_fwd(q, k, v, scale=0.1, residual=True, res_start=1))
    tb = t(lambda: K.attn_bwd(q, k, v, o, do, lse, scale=0.1, residual=True, res_start=1))
    ff = 2.0 * B * H * Nq * Nk * (DQK + DV)
    print(f"B{B} H{H} Nq{Nq} Nk{Nk} D{DQK}/{DV}: fwd {tf:7.1f} us ({ff/tf/1e6:6.1f} TF/s)  bwd {tb:7.1f} us ({2.5*ff/tb/1e6:6.1f} TF/s)", flush=True)
