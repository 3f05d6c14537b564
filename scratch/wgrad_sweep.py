"""Sweep split-K for the weight-gradient (TT) GEMM shapes of the B=24 step."""
import sys; sys.path.insert(0, '.')
import torch
from deepcoro_clip_amd import kernels as K
shapes = [(384, 1536, 37656), (1536, 384, 37656), (1152, 384, 37656), (384, 384, 37656), (768, 768, 768), (3072, 768, 768),
          (768, 3072, 768), (768, 192, 150552), (192, 768, 150552), (576,192,150552), (768, 3072, 9432), (3072,768,9432), (768,768,9432), (2304,768,9432), (72, 96, 150624), (96, 288, 602136), (384, 96, 602136)]
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
    return s.elapsed_time(e) / reps
for M, N, Kd in shapes:
    cur = K.wgrad_splitk(Kd, M, N)
    res = []
    cands = sorted(set([1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 512, cur]))
    for sk in cands:
        if sk > max(1, Kd // 64): continue
        ms = bench(M, N, Kd, sk)
        res.append((ms, sk))
    best = min(res)
    curms = [m for m, s in res if s == cur][0]
    print(f"M{M} N{N} K{Kd}: policy sk{cur} {curms*1e3:.1f} us ({2*M*N*Kd/curms/1e9:.0f} TF/s); best sk{best[1]} {best[0]*1e3:.1f} us ({2*M*N*Kd/best[0]/1e9:.0f} TF/s) | " + " ".join(f"{s}:{m*1e3:.0f}" for m, s in res), flush=True)
