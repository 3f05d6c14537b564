import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K
x = torch.randn(128 * 1024 * 1024, device='cuda'); y = torch.empty(128 * 1024 * 1024, device='cuda', dtype=torch.bfloat16)
for _ in range(3): K.cast(x, torch.bfloat16, out=y)   # reads 512 MiB fp32, writes 256 MiB bf16 per launch
torch.cuda.synchronize()
