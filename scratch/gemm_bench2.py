import sys, os; sys.path.insert(0, '.')
import torch
from deepcoro_clip_amd import kernels as K
M,N,Kd=37656,1536,384
NB=6
a=[torch.randn(M,Kd,device='cuda').bfloat16() for _ in range(NB)]
w=torch.randn(N,Kd,device='cuda').bfloat16()
bias=torch.randn(N,device='cuda')
outs=[torch.empty(M,N,device='cuda',dtype=torch.bfloat16) for _ in range(NB)]
pres=[torch.empty(M,N,device='cuda',dtype=torch.bfloat16) for _ in range(NB)]
res=[torch.randn(M,N,device='cuda').bfloat16() for _ in range(NB)]
def run(name, reps=24, **kw):
    def call(i):
        kk=dict(kw)
        if kk.pop('pre',False): kk.update(save_pre=True, aux=pres[i%NB])
        if kk.pop('dg',False): kk.update(dgelu=True, aux=pres[i%NB])
        if kk.pop('rs',False): kk.update(resid=res[i%NB])
        K.gemm(a[i%NB], w, out=outs[i%NB], **kk)
    for i in range(3): call(i)
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps): call(i)
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/reps
    print(f"{name:28s} {ms*1e3:8.1f} us  {2*M*N*Kd/ms/1e9:7.1f} TFLOP/s")
run("plain")
run("bias", bias=bias)
run("gelu", gelu=True)
run("pre", pre=True)
run("bias+gelu+pre", bias=bias, gelu=True, pre=True)
run("dgelu", dg=True)
run("resid", rs=True)
# copy bandwidth reference
x=torch.empty(256*1024*1024//2,device='cuda',dtype=torch.bfloat16); y=torch.empty_like(x)
for _ in range(2): K.cast(x, torch.bfloat16, out=y)
torch.cuda.synchronize()
s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): K.cast(x, torch.bfloat16, out=y)
e.record(); torch.cuda.synchronize()
ms=s.elapsed_time(e)/10
print(f"copy 256MB->256MB (cast kernel): {ms*1e3:.1f} us  {2*x.numel()*2/ms/1e9:.2f} TB/s")
y.copy_(x); torch.cuda.synchronize(); s.record()
for _ in range(10): y.copy_(x)
e.record(); torch.cuda.synchronize(); ms=s.elapsed_time(e)/10
print(f"torch copy_: {ms*1e3:.1f} us {2*x.numel()*2/ms/1e9:.2f} TB/s")
