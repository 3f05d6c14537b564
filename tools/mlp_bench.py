"""Fused LayerNorm + MLP forward (dcc_ln_mlp_fwd) against the unfused launches (LayerNorm + fc1/GELU + fc2/residual) at the
tower's shapes (32 clips), variants interleaved in one process (cdna guide rule 24).

usage: python tools/mlp_bench.py [rounds]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K

SHAPES = [("blk0", 802848, 96), ("blk1-2", 200736, 192), ("stage2", 50208, 384)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = "cuda:0"
    for name, M, C in SHAPES:
        H = 4 * C
        x = (torch.randn(M, C, device=dev) * 1.2).bfloat16()
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        w1 = (torch.randn(H, C, device=dev) * C ** -0.5).bfloat16()
        w2 = (torch.randn(C, H, device=dev) * H ** -0.5).bfloat16()
        b1, b2 = torch.zeros(H, device=dev), torch.zeros(C, device=dev)
        pk = K.mlp_pack(w1, w2)
        pre = torch.empty((M, H), device=dev, dtype=torch.bfloat16)

        def unfused():
            xn, _, _ = K.layernorm_fwd(x, g, b, 1e-6)
            act = K.gemm(xn, w1, bias=b1, gelu=True, save_pre=True, aux=pre)
            return K.gemm(act, w2, bias=b2, resid=x)

        variants = {"unfused": unfused,
                    "fused": lambda: K.ln_mlp_fwd(x, g, b, 1e-6, pk, b1, b2),
                    "fused(no act)": lambda: K.ln_mlp_fwd(x, g, b, 1e-6, pk, b1, b2, save_act=False),
                    "fused(infer)": lambda: K.ln_mlp_fwd(x, g, b, 1e-6, pk, b1, b2, save=False)}
        times = {k: [] for k in variants}
        for fn in variants.values():
            fn()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for k, fn in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) * 200.0)
        flops = 16.0 * M * C * C
        print(f"{name}: M {M} C {C}  ({flops / 1e9:.1f} GFLOP)")
        for k, v in times.items():
            v = sorted(v)
            med = v[len(v) // 2]
            print(f"   {k:14s} median {med:8.1f} us  min {v[0]:8.1f} us   {flops / med / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
