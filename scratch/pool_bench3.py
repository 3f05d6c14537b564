import sys; sys.path.insert(0, '.')
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
# the pools of the headline workload: (heads, C, thw, stride)
cases = [("b0 q", 1, 96, (8, 56, 56), (1, 1, 1)), ("b0 kv", 1, 96, (8, 56, 56), (1, 8, 8)),
         ("b1 q", 2, 96, (8, 56, 56), (1, 2, 2)), ("b2 kv", 2, 96, (8, 28, 28), (1, 4, 4)),
         ("b3 q", 4, 96, (8, 28, 28), (1, 2, 2)), ("b4 q", 4, 96, (8, 14, 14), (1, 1, 1)),
         ("b4 kv", 4, 96, (8, 14, 14), (1, 2, 2)), ("b14 q", 8, 96, (8, 14, 14), (1, 2, 2)), ("b15 q", 8, 96, (8, 7, 7), (1, 1, 1))]
for B in (24,):
    for name, H, C, thw, stride in cases:
        g = K.PoolGeom(B, H, C, thw, (3, 3, 3), stride, (1, 1, 1), True)
        qkv = torch.randn(B, g.n_in, 3, H, C, device='cuda').bfloat16()
        x = qkv[:, :, 0]; w = torch.randn(C, 27, device='cuda')
        y, _ = K.pool_fwd(x, g, w); dy = torch.randn_like(y); dw = torch.zeros(C, 27, device='cuda')
        dx = torch.empty(B, g.n_in, H, C, device='cuda', dtype=torch.bfloat16)
        mb_in = x.numel() * 2 / 1e6; mb_out = y.numel() * 2 / 1e6
        tw = t(lambda: K.dwconv_bwd_weight(dy, x, g, dw)); tf = t(lambda: K.pool_fwd(x, g, w, out=y))
        td = t(lambda: K.pool_bwd_data(dy, dx, g, w))
        print(f"B{B} {name:6s} in {mb_in:6.1f} MB out {mb_out:6.1f} MB | wgrad {tw:7.1f} us ({(mb_in+mb_out)/tw*1e-3*1e3:6.0f} GB/s) fwd {tf:7.1f} us ({(mb_in+mb_out)/tf*1e3:6.0f} GB/s) bwd_data {td:7.1f} us ({(mb_in+mb_out)/td*1e3:6.0f} GB/s)", flush=True)
