import sys, os, subprocess; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
def run():
    from deepcoro_clip_amd import kernels as K
    torch.manual_seed(0)
    B, H, Nq, Nk, DQK, DV = 1, 1, 128, 32, 128, 96
    q = torch.randn(B, Nq, H, DQK, device='cuda').bfloat16()
    k = torch.randn(B, Nk, H, DQK, device='cuda').bfloat16()
    v = torch.randn(B, Nk, H, DV, device='cuda').bfloat16()
    o, lse = K.attn_fwd(q, k, v, scale=0.1)
    do = torch.randn_like(o)
    m = torch.zeros(Nq, device='cuda')
    for t in (1, 3): m[32 * t:32 * t + 32] = 1
    do = do * m[None, :, None, None].to(do.dtype)
    dq, dk, dv = K.attn_bwd(q, k, v, o, do, lse, scale=0.1, qsplit=1)
    torch.save((dq.float().cpu(), dk.float().cpu(), dv.float().cpu(), do.float().cpu(), q.float().cpu()), sys.argv[1])
if len(sys.argv) > 1:
    run()
else:
    for tag, env in (("pipe", "1"), ("base", "0")):
        subprocess.run([sys.executable, __file__, f"/tmp/attn2_{tag}.pt"], env={**os.environ, "DCC_ATTN_PIPE": env}, check=True)
    a = torch.load("/tmp/attn2_pipe.pt"); b = torch.load("/tmp/attn2_base.pt")
    do = a[3][0, :, 0]  # [Nq, DV]
    q = a[4][0, :, 0]
    for name, idx, basis in (("dv", 2, do), ("dk", 1, q)):
        err = (a[idx] - b[idx])[0, :, 0]  # [Nk, D]
        print(name, "max err", err.abs().max().item())
        for t in (1, 3):
            Bm = basis[32 * t:32 * t + 32]  # [32, D]
            sol = torch.linalg.lstsq(Bm.t(), err.t()).solution  # [32, Nk]: delta[q, k]
            res = (Bm.t() @ sol - err.t()).abs().max().item()
            print(f"  fit with tile {t} rows: residual {res:.4f}")
            if res < 1e-2:
                nz = (sol.abs() > 1e-3).nonzero()
                print("   nonzero delta (row q, key k):", nz.tolist()[:40])
                print("   values:", [round(sol[i, j].item(), 4) for i, j in nz.tolist()[:12]])
