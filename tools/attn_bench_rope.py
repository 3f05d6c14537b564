"""Attention backward at the head width of the RoPE configuration (DQK = DV = 96, 32 frames, 32 clips): tools/attn_bench2.py's
loop over configs[4]'s MViT shapes.  DCC_ATTN_PIPE=0|1 selects the dK/dV kernel."""
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
B = 32
tot = 0.0
for (H, Nq, Nk, n) in ((1, 50177, 785, 1), (2, 12545, 3137, 1), (2, 12545, 785, 1), (4, 3137, 3137, 1), (4, 3137, 785, 10), (8, 785, 3137, 1), (8, 785, 785, 1)):
    q = torch.randn(B, Nq, H, 96, device='cuda').bfloat16() * 0.3
    k = torch.randn(B, Nk, H, 96, device='cuda').bfloat16() * 0.3
    v = torch.randn(B, Nk, H, 96, device='cuda').bfloat16()
    o, lse = K.attn_fwd(q, k, v, scale=0.1, residual=True, res_start=1)
    do = torch.randn_like(o)
    tb = t(lambda: K.attn_bwd(q, k, v, o, do, lse, scale=0.1, residual=True, res_start=1))
    tot += n * tb
    print(f"H{H} Nq{Nq} Nk{Nk} D96/96 x{n}: bwd {tb:8.1f} us", flush=True)
    del q, k, v, o, do
print(f"per step: bwd {tot/1e3:.2f} ms")
