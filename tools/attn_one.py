"""One attention shape, backward only, a few launches (for rocprofv3 --pmc): attn_one.py H Nq Nk DQK DV [reps]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K
H, Nq, Nk, DQK, DV = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
B = 32
q = torch.randn(B, Nq, H, DQK, device='cuda').bfloat16() * 0.3
k = torch.randn(B, Nk, H, DQK, device='cuda').bfloat16() * 0.3
v = torch.randn(B, Nk, H, DV, device='cuda').bfloat16()
o, lse = K.attn_fwd(q, k, v, scale=0.1, residual=True, res_start=1)
do = torch.randn_like(o)
for _ in range(reps):
    K.attn_bwd(q, k, v, o, do, lse, scale=0.1, residual=True, res_start=1)
torch.cuda.synchronize()
