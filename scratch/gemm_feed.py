"""Is the ring GEMM bound by the L2->LDS operand feed?  Same launch with operands whose rows all alias one row
(stride 0: every DMA request hits the same few cache lines) vs real operands."""
import sys; sys.path.insert(0, '.')
import torch
from deepcoro_clip_amd import kernels as K
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps * 1e3
for M, N, Kd in ((37656, 1536, 384), (37656, 384, 1536), (37656, 1152, 384), (37656, 384, 384), (8192, 8192, 8192)):
    a = torch.randn(M, Kd, device='cuda').bfloat16(); b = torch.randn(N, Kd, device='cuda').bfloat16()
    o = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    a0 = a[:1].expand(M, Kd); b0 = b[:1].expand(N, Kd)
    r = [t(lambda: K.gemm(a, b, out=o)), t(lambda: K.gemm(a0, b, out=o)), t(lambda: K.gemm(a, b0, out=o)), t(lambda: K.gemm(a0, b0, out=o))]
    fl = 2.0 * M * N * Kd
    print(f"M{M} N{N} K{Kd}: real {r[0]:7.1f} us ({fl/r[0]/1e6:5.0f} TF/s) | A aliased {r[1]:7.1f} | B aliased {r[2]:7.1f} | both {r[3]:7.1f} ({fl/r[3]/1e6:5.0f} TF/s)", flush=True)
