import sys, os; sys.path.insert(0, "/root/repo")
import torch
from deepcoro_clip_amd import kernels as K
from tools.p8_bench import t
K.GEMM_VARIANT = 13 << 8
for (M,N,Kd) in [(50208,1536,384),(50208,1152,384),(50208,1536,128),(8192,8192,8192)]:
    a = (torch.rand(M, Kd, device="cuda") * 2 - 1).bfloat16(); b = (torch.rand(N, Kd, device="cuda") * 2 - 1).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: K.gemm(a, b, out=c))
    print(f"DCC_P8_DEBUG={os.environ.get('DCC_P8_DEBUG','0')} M{M} N{N} K{Kd}: {us:.1f} us {2*M*N*Kd/us/1e6:.0f} TF", flush=True)
