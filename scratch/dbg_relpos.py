import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import functional as Fn, kernels as K
sys.path.insert(0, 'tests')
from test_relpos_fused_gpu import _tables
B,H,q_thw,k_thw=1,1,(8,14,14),(8,7,7)
D=96; g=torch.Generator().manual_seed(5); A=Fn.relpos_aug_width(k_thw); DQK=D+A
Nq=q_thw[0]*q_thw[1]*q_thw[2]+1
idx,inv,tabs=_tables(q_thw,k_thw,D,g)
lens=tuple(t.shape[0] for t in tabs); Lp=8*((sum(lens)+7)//8)
Rcat=torch.zeros(Lp,D,dtype=torch.bfloat16); Rcat[:sum(lens)]=torch.cat(tabs,0); Rcat=Rcat.cuda()
scale=1.0/D**0.5
dims=K.relpos_dims(B,H,D,A,Lp,q_thw,k_thw,lens)
q=(torch.randn(B,Nq,H,D,generator=g)*0.5).bfloat16()
Qa=torch.full((B,Nq,H,DQK),3.0,dtype=torch.bfloat16).cuda(); Qa[...,:D]=q.cuda(); Qb=Qa.clone()
K.relpos_aug_fwd(Qa,Rcat,idx,dims,scale)
P=K.gemm(Qb.view(-1,DQK)[:,:D],Rcat); K.relpos_gather(Qb,P,idx,dims,scale)
torch.cuda.synchronize()
for n in (0,1,2,200):
    print(n, 'fused', Qa[0,n,0,D:D+24].float().cpu().numpy().round(2))
    print(n, 'old  ', Qb[0,n,0,D:D+24].float().cpu().numpy().round(2))
# direct check of a single dot
l=idx[0][0,0].item()
print('l',l, (q[0,1,0].float()@Rcat[l].float().cpu()/scale).item())
print("---- unit test: q = one-hot channel ch")
for ch in (0, 1, 8, 50, 95):
    Qa=torch.zeros((B,Nq,H,DQK),dtype=torch.bfloat16).cuda(); Qa[...,ch]=1.0
    K.relpos_aug_fwd(Qa,Rcat,idx,dims,scale); torch.cuda.synchronize()
    n=5  # token 5: t=0,h=0,w=4
    got=Qa[0,n,0,D:D+22].float().cpu()*scale
    ih,iw,it=[i.cpu().long() for i in idx]
    exp=torch.cat([Rcat[ih[0]][:,ch].float().cpu(), Rcat[lens[0]+iw[4]][:,ch].float().cpu(), Rcat[lens[0]+lens[1]+it[0]][:,ch].float().cpu()])
    print(ch, 'got', got.numpy().round(3)[:10]); print(ch, 'exp', exp.numpy().round(3)[:10])
