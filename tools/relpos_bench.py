"""Rel-pos path of one MViT block at the headline shapes: fused kernels against the stacked-table GEMM + gather / scatter path.
usage: python tools/relpos_bench.py [blk ...]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import functional as Fn, kernels as K

CLIPS = 32
BLOCKS = {  # heads, q grid, k grid
    "blk0": (1, (8, 56, 56), (8, 7, 7)), "blk1": (2, (8, 28, 28), (8, 14, 14)), "blk2": (2, (8, 28, 28), (8, 7, 7)),
    "blk3": (4, (8, 14, 14), (8, 14, 14)), "blk4-13": (4, (8, 14, 14), (8, 7, 7)), "blk14": (8, (8, 7, 7), (8, 14, 14)),
    "blk15": (8, (8, 7, 7), (8, 7, 7)),
}


def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    D = 96
    for name in (sys.argv[1:] or list(BLOCKS)):
        H, q_thw, k_thw = BLOCKS[name]
        B = CLIPS
        A = Fn.relpos_aug_width(k_thw); DQK = D + A
        Nq = q_thw[0] * q_thw[1] * q_thw[2] + 1
        idx, inv, lens = [], [], []
        for a in (1, 2, 0):
            qs, ks = q_thw[a], k_thw[a]
            d = 2 * max(qs, ks) - 1
            ix = Fn.relpos_index(qs, ks)
            iv = torch.full((qs, d), -1, dtype=torch.int32)
            iv.scatter_(1, ix.long(), torch.arange(ks, dtype=torch.int32).expand(qs, ks).contiguous())
            idx.append(ix.cuda()); inv.append(iv.contiguous().cuda()); lens.append(d)
        Lp = 8 * ((sum(lens) + 7) // 8)
        Rcat = (torch.randn(Lp, D, device="cuda") * 0.5).bfloat16()
        dims = K.relpos_dims(B, H, D, A, Lp, q_thw, k_thw, lens)
        Qf = (torch.randn(B, Nq, H, DQK, device="cuda") * 0.5).bfloat16()
        dQf = torch.randn_like(Qf)
        scale = D ** -0.5
        rows = B * Nq * H

        def old_f():
            P = K.gemm(Qf.view(-1, DQK)[:, :D], Rcat)
            K.relpos_gather(Qf, P, idx, dims, scale)

        def old_b():
            dP = torch.empty(rows, Lp, device="cuda", dtype=torch.bfloat16)
            K.relpos_scatter(dQf, dP, inv, dims, scale)
            Fn.wgrad(dP, Qf.view(-1, DQK)[:, :D])
            K.gemm(dP, Rcat, b_km=True, resid=dQf.view(-1, DQK)[:, :D])

        fused = ""
        mf = f"{t(lambda: K.relpos_aug_fwd_mfma(Qf, Rcat, idx, dims, scale)):7.1f} us" if K.relpos_aug_fwd_mfma(Qf, Rcat, idx, dims, scale) else "   n/a"
        fused = f"MFMA tile kernel {mf}"
        dR = torch.zeros(Lp, D, device="cuda")
        fb = K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, scale, dR)
        fusedb = f"{t(lambda: K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, scale, dR)):7.1f} us" if fb is not None else "   n/a"
        # the query pool's LayerNorm backward: as the next launch pair (ln_bwd + ln_reduce) or in the rel-pos kernel's epilogue
        conv = torch.randn(B, Nq, H, D, device="cuda").bfloat16()
        gamma = torch.ones(D, device="cuda"); beta = torch.zeros(D, device="cuda")
        _, mean, rstd = K.layernorm_fwd(conv.view(rows, D), gamma, beta, 1e-6)
        dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
        lnb = ""
        if fb is not None:
            def two():
                dq = K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, scale, dR)
                K.layernorm_bwd(dq.view(rows, D), conv.view(rows, D), mean, rstd, gamma, dg, db)
            one = lambda: K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, scale, dR, ln=(conv, mean, rstd, gamma, dg, db))
            lnb = f";  bwd + q-pool LayerNorm bwd: separate {t(two):7.1f} us, epilogue {t(one):7.1f} us"
        print(f"{name}: rows {rows}, L {Lp}, A {A}:  fwd gemm+gather {t(old_f):7.1f} us, {fused};"
              f"  bwd scatter+2 gemm {t(old_b):7.1f} us, fused (MFMA) {fusedb}{lnb}", flush=True)


if __name__ == "__main__":
    main()
