"""In-situ GEMM shapes of the B=24 step, timed in isolation (rotating buffers to defeat cache reuse across calls)."""
import sys; sys.path.insert(0, '.')
import torch
from deepcoro_clip_amd import kernels as K
def bench(M, N, Kd, lay, reps=20, **kw):
    NB = 4
    a_km, b_km = lay[0] == 'T', lay[1] == 'T'
    a = [torch.randn((Kd, M) if a_km else (M, Kd), device='cuda').bfloat16() for _ in range(NB)]
    b = [torch.randn((Kd, N) if b_km else (N, Kd), device='cuda').bfloat16() for _ in range(NB)]
    out = [torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(NB)]
    def call(i): K.gemm(a[i % NB], b[i % NB], a_km=a_km, b_km=b_km, out=out[i % NB], **kw)
    for i in range(3): call(i)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps): call(i)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return ms * 1e3, 2.0 * M * N * Kd / ms / 1e9
shapes = [(37656, 1536, 384, 'NN'), (37656, 1536, 384, 'NT'), (37656, 384, 1536, 'NN'), (37656, 384, 1536, 'NT'),
          (37656, 1152, 384, 'NN'), (37656, 384, 1152, 'NT'), (37656, 384, 384, 'NN'), (37656, 384, 384, 'NT'),
          (150552, 768, 192, 'NN'), (150552, 192, 768, 'NN'), (150552, 576, 192, 'NN'), (9432, 3072, 768, 'NN'), (9432, 768, 3072, 'NN'),
          (768, 768, 768, 'NN'), (768, 3072, 768, 'NN'), (768, 768, 3072, 'NN'), (4096, 4096, 4096, 'NN'), (8192, 8192, 8192, 'NT')]
tot = 0
for M, N, Kd, lay in shapes:
    us, tf = bench(M, N, Kd, lay)
    print(f"M{M} N{N} K{Kd} {lay}: {us:8.1f} us {tf:7.1f} TF/s", flush=True)
