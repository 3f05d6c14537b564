"""Pipelined vs unpipelined dK/dV kernel on random shapes: `attn_pipe_fuzz.py OUT.pt` runs the shapes with the kernel choice of the
current DCC_ATTN_PIPE and saves the gradients; without arguments it runs itself twice (forced on / off) and compares — bit-exact for
qsplit == 1 (same summation order), to fp32 atomics' rounding otherwise."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def shapes(n, seed=1234):
    g = torch.Generator().manual_seed(seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    out = []
    for i in range(n):
        dqk = (96, 128, 160)[r(0, 2)]
        out.append(dict(B=r(1, 2), H=r(1, 3), Nq=r(1, 1400) if i % 3 else r(1, 140), Nk=r(1, 450), DQK=dqk, DV=96, qsplit=(1, 1, 2, 3, 5)[r(0, 4)],
                        residual=bool(r(0, 1))))
    return out

def run(path, n):
    from deepcoro_clip_amd import kernels as K
    res = []
    for i, s in enumerate(shapes(n)):
        torch.manual_seed(i)
        q = torch.randn(s["B"], s["Nq"], s["H"], s["DQK"], device="cuda").bfloat16()
        k = torch.randn(s["B"], s["Nk"], s["H"], s["DQK"], device="cuda").bfloat16()
        v = torch.randn(s["B"], s["Nk"], s["H"], s["DV"], device="cuda").bfloat16()
        o, lse = K.attn_fwd(q, k, v, scale=0.1, residual=s["residual"], res_start=1)
        do = torch.randn_like(o)
        dq, dk, dv = K.attn_bwd(q, k, v, o, do, lse, scale=0.1, residual=s["residual"], res_start=1, qsplit=s["qsplit"])
        res.append((dq.float().cpu(), dk.float().cpu(), dv.float().cpu()))
    torch.save(res, path)

def compare(n=36, tmp="/tmp"):
    for tag, env in (("pipe", "1"), ("base", "0")):
        subprocess.run([sys.executable, os.path.abspath(__file__), f"{tmp}/attn_fuzz_{tag}.pt", str(n)], env={**os.environ, "DCC_ATTN_PIPE": env}, check=True)
    a, b = torch.load(f"{tmp}/attn_fuzz_pipe.pt"), torch.load(f"{tmp}/attn_fuzz_base.pt")
    bad = []
    for s, x, y in zip(shapes(n), a, b):
        for name, u, w in zip(("dq", "dk", "dv"), x, y):
            d = (u - w).abs().max().item() if u.numel() else 0.0
            tol = 0.0 if s["qsplit"] == 1 or name == "dq" else 2e-2 * max(1.0, w.abs().max().item())
            if not (d <= tol):
                bad.append((s, name, d))
    return bad

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1], int(sys.argv[2]))
    else:
        bad = compare()
        print("mismatches:", bad if bad else "none")
        sys.exit(1 if bad else 0)
