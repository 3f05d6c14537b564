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
print(f"im2col fp32 -> bf16: {us:7.1f} us  {mb / us / 1e6 * 1e6:6.2f} TB/s ({mb:.0f} MB)")
u8 = torch.randint(0, 256, (32, 16, 224, 224, 3), device="cuda", dtype=torch.uint8)
norm = (torch.tensor([110.0, 100.0, 120.0], device="cuda"), torch.tensor([58.0, 61.0, 57.0], device="cuda"))
us = t(lambda: K.im2col3d(u8, (3, 7, 7), (2, 4, 4), (1, 3, 3), 448, torch.bfloat16, norm=norm))
print(f"im2col uint8 -> bf16: {us:7.1f} us")
