"""Pool kernels at the shapes of the B=24 step (blocks 4-13 q and k/v paths, block 0)."""
import sys; sys.path.insert(0, '.')
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
B = 24
for name, H, thw, stride in (("b4-13 q", 4, (8, 14, 14), (1, 1, 1)), ("b4-13 kv", 4, (8, 14, 14), (1, 2, 2)), ("b14 kv", 8, (8, 14, 14), (1, 1, 1)),
                             ("b0 q", 1, (8, 56, 56), (1, 1, 1)), ("b0 kv", 1, (8, 56, 56), (1, 8, 8)), ("b2 kv", 2, (8, 28, 28), (1, 4, 4))):
    g = K.PoolGeom(B, H, 96, thw, (3, 3, 3), stride, (1, 1, 1), True)
    qkv = torch.randn(B, g.n_in, 3, H, 96, device='cuda').bfloat16()
    x = qkv[:, :, 0]
    w = torch.randn(96, 27, device='cuda')
    y, _ = K.pool_fwd(x, g, w)
    dy = torch.randn_like(y)
    dqkv = torch.empty_like(qkv)
    dw = torch.zeros(96, 27, device='cuda')
    mb_in, mb_out = x.numel() * 2 / 1e6, y.numel() * 2 / 1e6
    tf = t(lambda: K.pool_fwd(x, g, w, out=y))
    tb = t(lambda: K.pool_bwd_data(dy, dqkv[:, :, 0], g, w))
    tw = t(lambda: K.dwconv_bwd_weight(dy, x, g, dw))
    print(f"{name:9s} in {mb_in:6.1f} MB out {mb_out:6.1f} MB | fwd {tf:6.1f} us ({(mb_in+mb_out)/tf:5.2f} TB/s) | bwd_data {tb:6.1f} us ({(mb_in+mb_out)/tb:5.2f} TB/s) | bwd_weight {tw:6.1f} us ({(mb_in+mb_out)/tw:5.2f} TB/s)", flush=True)
