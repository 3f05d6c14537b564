"""conv_proj operand gather at the headline shape (32 clips of 16 x 224 x 224): fp32 and uint8 frames."""
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
v = torch.randn(32, 16, 224, 224, 3, device="cuda")
us = t(lambda: K.im2col3d(v, (3, 7, 7), (2, 4, 4), (1, 3, 3), 448, torch.bfloat16))
mb = v.numel() * 4 / 1e6 + 32 * 8 * 56 * 56 * 448 * 2 / 1e6
print(f"im2col fp32 -> bf16: {us:7.1f} us  {mb / us / 1e6 * 1e6 / 1e6:6.2f} TB/s ({mb:.0f} MB)")
