import sys, os; sys.path.insert(0, '.')
import torch
from deepcoro_clip_amd import kernels as K
def bench(M,N,Kd,a_km=False,b_km=False,reps=20, **kw):
    a = torch.randn((Kd,M) if a_km else (M,Kd), device='cuda').bfloat16()
    b = torch.randn((Kd,N) if b_km else (N,Kd), device='cuda').bfloat16()
    out_dtype = kw.pop('out_dtype', torch.bfloat16)
    for _ in range(3): K.gemm(a,b,a_km=a_km,b_km=b_km,out_dtype=out_dtype,**kw)
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): K.gemm(a,b,a_km=a_km,b_km=b_km,out_dtype=out_dtype,**kw)
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/reps
    print(f"M{M} N{N} K{Kd} {'T' if a_km else 'N'}{'T' if b_km else 'N'} {kw}: {ms*1e3:8.1f} us  {2*M*N*Kd/ms/1e9:7.1f} TFLOP/s")
for shape in [(4096,4096,4096),(8192,8192,8192),(37656,1536,384),(37656,384,1536),(37656,384,384),(602136,288,96),(602136,96,384)]:
    bench(*shape)
bench(4096,4096,4096,b_km=True)
bench(1536,384,37656,a_km=True,b_km=True,out_dtype=torch.float32,splitk=28)
bench(4096,4096,4096,a_km=True,b_km=True,out_dtype=torch.float32)
