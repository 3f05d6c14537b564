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
for rows, C in ((24 * 1569 * 4, 96), (24 * 393 * 4, 96), (24 * 25089, 96), (37656, 384), (37656, 768)):
    x = torch.randn(rows, C, device='cuda').bfloat16(); dy = torch.randn_like(x)
    g = torch.randn(C, device='cuda'); b = torch.randn(C, device='cuda')
    y, m, r = K.layernorm_fwd(x, g, b, 1e-6)
    dg = torch.zeros(C, device='cuda'); db = torch.zeros(C, device='cuda')
    tf = t(lambda: K.layernorm_fwd(x, g, b, 1e-6, out=y)); tb = t(lambda: K.layernorm_bwd(dy, x, m, r, g, dg, db))
    mb = rows * C * 2 / 1e6
    print(f"rows {rows} C {C}: fwd {tf:.1f} us ({2*mb/tf:.2f} TB/s) bwd {tb:.1f} us ({3*mb/tb:.2f} TB/s)", flush=True)
