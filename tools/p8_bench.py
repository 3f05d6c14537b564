"""Eight-phase kernel (variant 13) against the shipped dispatch (0), the ring (10) and the wide kernels (8, 11) on the stage-2 forward
shapes and two cubes; random uniform [-1, 1) operands.  usage: python tools/p8_bench.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    shapes = [("plain", 50208, 1536, 384, {}), ("qkv bias", 50208, 1152, 384, {"bias": 1}), ("fc1 bias+gelu+pre", 50208, 1536, 384, {"bias": 1, "gelu": 1}),
              ("fc2 bias+resid", 50208, 384, 1536, {"bias": 1, "resid": 1}), ("plain", 50208, 384, 1536, {}), ("plain", 50208, 384, 384, {}),
              ("cube", 4096, 4096, 4096, {}), ("cube", 8192, 8192, 8192, {})]
    for name, M, N, Kd, ep in shapes:
        a = (torch.rand(M, Kd, device="cuda") * 2 - 1).bfloat16()
        b = ((torch.rand(N, Kd, device="cuda") * 2 - 1) * (0.05 if ep else 1.0)).bfloat16()
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        kw = {}
        if ep.get("bias"): kw["bias"] = torch.randn(N, device="cuda")
        if ep.get("resid"): kw["resid"] = torch.randn(M, N, device="cuda").bfloat16()
        if ep.get("gelu"): kw.update(gelu=True, save_pre=True, aux=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
        row = []
        for v in (0, 10, 8, 11, 13):
            K.GEMM_VARIANT = v << 8
            us = t(lambda: K.gemm(a, b, out=c, **kw))
            row.append(f"v{v} {us:7.1f} us {2 * M * N * Kd / us / 1e6:7.1f} TF")
        K.GEMM_VARIANT = 0
        print(f"{name:18s} M{M} N{N} K{Kd}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
