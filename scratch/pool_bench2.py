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
for B in (1, 2, 4, 8, 24, 48):
    H, thw, stride = 4, (8, 14, 14), (1, 2, 2)
    g = K.PoolGeom(B, H, 96, thw, (3, 3, 3), stride, (1, 1, 1), True)
    qkv = torch.randn(B, g.n_in, 3, H, 96, device='cuda').bfloat16()
    x = qkv[:, :, 0]; w = torch.randn(96, 27, device='cuda')
    y, _ = K.pool_fwd(x, g, w); dy = torch.randn_like(y); dw = torch.zeros(96, 27, device='cuda')
    print(B, f"bwd_weight {t(lambda: K.dwconv_bwd_weight(dy, x, g, dw)):.1f} us  fwd {t(lambda: K.pool_fwd(x, g, w, out=y)):.1f} us")
