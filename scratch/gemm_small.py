"""Small / skinny GEMM shapes of the step (rel-pos products, BERT projections): old tiled kernel vs ring kernel."""
import sys, os; sys.path.insert(0, '.')
import torch
from deepcoro_clip_amd import kernels as K
def bench(M, N, Kd, lay, reps=30, out_dtype=torch.bfloat16, **kw):
    NB = 4
    a_km, b_km = lay[0] == 'T', lay[1] == 'T'
    a = [torch.randn((Kd, M) if a_km else (M, Kd), device='cuda').bfloat16() for _ in range(NB)]
    b = [torch.randn((Kd, N) if b_km else (N, Kd), device='cuda').bfloat16() for _ in range(NB)]
    out = [torch.zeros(M, N, device='cuda', dtype=out_dtype) for _ in range(NB)]
    def call(i): K.gemm(a[i % NB], b[i % NB], a_km=a_km, b_km=b_km, out=out[i % NB], **kw)
    for i in range(3): call(i)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps): call(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
cases = [("relpos P   ", 150624, 72, 96, 'NN', torch.bfloat16, {}),
         ("relpos dQ  ", 150624, 96, 72, 'NT', torch.bfloat16, {}),
         ("relpos dR 128", 72, 96, 150624, 'TT', torch.float32, dict(splitk=128, accumulate=True)),
         ("relpos dR 256", 72, 96, 150624, 'TT', torch.float32, dict(splitk=256, accumulate=True)),
         ("relpos dR 512", 72, 96, 150624, 'TT', torch.float32, dict(splitk=512, accumulate=True)),
         ("bert 768^3 NN", 768, 768, 768, 'NN', torch.bfloat16, {}), ("bert 768^3 NT", 768, 768, 768, 'NT', torch.bfloat16, {}),
         ("bert fc1 NN", 768, 3072, 768, 'NN', torch.bfloat16, {}), ("bert fc2 NN", 768, 768, 3072, 'NN', torch.bfloat16, {}),
         ("bert wgrad sk3", 768, 768, 768, 'TT', torch.float32, dict(splitk=3, accumulate=True)),
         ("bert wgrad sk6", 768, 768, 768, 'TT', torch.float32, dict(splitk=6, accumulate=True)),
         ("bert wgrad fc sk1", 3072, 768, 768, 'TT', torch.float32, dict(splitk=1, accumulate=True)),
         ("bert wgrad fc sk2", 3072, 768, 768, 'TT', torch.float32, dict(splitk=2, accumulate=True))]
for name, M, N, Kd, lay, od, kw in cases:
    r = []
    for v in (0, 5):
        K.GEMM_VARIANT = v << 8
        r.append(bench(M, N, Kd, lay, out_dtype=od, **kw))
    print(f"{name:18s} M{M} N{N} K{Kd} {lay}: tiled {r[0]:7.1f} us   ring {r[1]:7.1f} us", flush=True)
print("fp32 skinny rows (tiled = variant 3):")
def bench32(M, N, Kd, b_km, reps=30):
    a = torch.randn(M, Kd, device='cuda'); b = torch.randn((Kd, N) if b_km else (N, Kd), device='cuda'); o = torch.empty(M, N, device='cuda')
    for _ in range(3): K.gemm(a, b, b_km=b_km, out=o)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(reps): K.gemm(a, b, b_km=b_km, out=o)
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps * 1e3
for M, N, Kd, b_km in [(24, 512, 2048, False), (24, 512, 2048, True), (24, 512, 1536, False), (24, 2048, 512, False), (24, 2048, 512, True), (24, 512, 512, False), (24, 24, 512, True), (24, 1536, 512, True)]:
    r = []
    for v in (3, 0):
        K.GEMM_VARIANT = v << 8
        r.append(bench32(M, N, Kd, b_km))
    print(f"M{M} N{N} K{Kd} {'NN' if b_km else 'NT'}: tiled {r[0]:7.1f} us  skinny {r[1]:7.1f} us", flush=True)
