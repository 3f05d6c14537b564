import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K
def bench(M, N, Kd, sk, reps=10):
    NB = 3
    dy = [torch.randn(Kd, M, device='cuda').bfloat16() for _ in range(NB)]
    x = [torch.randn(Kd, N, device='cuda').bfloat16() for _ in range(NB)]
    out = torch.zeros(M, N, device='cuda'); cs = torch.zeros(M, device='cuda')
    def call(i): K.gemm(dy[i % NB], x[i % NB], a_km=True, b_km=True, out=out, accumulate=True, splitk=sk, colsum=cs)
    for i in range(2): call(i)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps): call(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for M, N, Kd, cands in ((384, 1536, 37656, (5, 6, 7)), (1152, 384, 37656, (7, 8, 9)), (384, 384, 37656, (24, 28)), (768, 3072, 9432, (1, 2, 3, 4)), (768, 768, 9432, (5, 6, 7)),
                        (2304, 768, 9432, (2,)), (768, 192, 150552, (16, 20, 21)), (576, 192, 150552, (24, 25)), (96, 288, 602136, (64, 85))):
    print(f"M{M} N{N} K{Kd}: policy sk{K.wgrad_splitk(Kd, M, N)} | " + " ".join(f"{sk}:{bench(M, N, Kd, sk):.0f}" for sk in cands), flush=True)
