import sys, os, subprocess; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
def run():
    from deepcoro_clip_amd import kernels as K
    torch.manual_seed(0)
    B, H, Nq, Nk, DQK, DV = (int(x) for x in sys.argv[2:8])
    q = torch.randn(B, Nq, H, DQK, device='cuda').bfloat16()
    k = torch.randn(B, Nk, H, DQK, device='cuda').bfloat16()
    v = torch.randn(B, Nk, H, DV, device='cuda').bfloat16()
    o, lse = K.attn_fwd(q, k, v, scale=0.1)
    do = torch.randn_like(o)
    tsel = os.environ.get('DBG_TILE', '')
    if tsel:
        m = torch.zeros(Nq, device='cuda')
        for t in tsel.split(','): m[32 * int(t):32 * int(t) + 32] = 1
        do = do * m[None, :, None, None].to(do.dtype)
    dq, dk, dv = K.attn_bwd(q, k, v, o, do, lse, scale=0.1, qsplit=1)
    torch.save((dq.float().cpu(), dk.float().cpu(), dv.float().cpu()), sys.argv[1])
if len(sys.argv) > 2:
    run()
else:
    for tile, shape in (('', ["1", "1", str(nq), "32", "128", "96"]) for nq in (128, 160, 192, 224, 256, 288, 300, 320)):
      os.environ["DBG_TILE"] = str(tile); print("tile", tile)
      if True:
        for tag, env in (("pipe", "1"), ("base", "0")):
            subprocess.run([sys.executable, __file__, f"/tmp/attn_{tag}.pt", *shape], env={**os.environ, "DCC_ATTN_PIPE": env}, check=True)
        a = torch.load("/tmp/attn_pipe.pt"); b = torch.load("/tmp/attn_base.pt")
        for name, x, y in zip(("dq", "dk", "dv"), a, b):
            d = (x - y).abs()
            print(f"  {name}: max diff {d.max().item():.4f} ref max {y.abs().max().item():.3f}")
            continue
            if d.max() > 0.01 * y.abs().max():
                bad = (d > 0.01 * y.abs().max()).nonzero()
                print("   bad count", len(bad), "of", d.numel(), "first", bad[:6].tolist(), "last", bad[-3:].tolist())
                # per key summary
                perkey = (d > 0.01 * y.abs().max()).any(-1).any(-1)  # [B, N]
                print("   bad keys (batch 0):", perkey[0].nonzero().flatten().tolist()[:40])
                cols = (d > 0.01 * y.abs().max()).any(0).any(0).any(0)
                print("   bad cols:", cols.nonzero().flatten().tolist())
