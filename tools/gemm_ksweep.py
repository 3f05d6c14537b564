"""Per-tile fixed cost and per-K-step cost of the bf16 GEMM kernels: time(M, N, K) over a K sweep at the stage-2 row count, fitted as
tiles_per_CU * (a + b * K / 64).  usage: python tools/gemm_ksweep.py [variant ...]   (0 dispatch, 10 ring only, 8 wide 256x256, 11 wide 256x128)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
    variants = [int(v) for v in sys.argv[1:]] or [10]
    M = 50208
    for variant in variants:
        K.GEMM_VARIANT = variant << 8
        for N in (1536, 1152, 384):
            rows = []
            for Kd in (128, 256, 384, 768, 1536, 3072):
                a = (torch.rand(M, Kd, device="cuda") * 2 - 1).bfloat16()
                b = (torch.rand(N, Kd, device="cuda") * 2 - 1).bfloat16()
                c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                us = t(lambda: K.gemm(a, b, out=c))
                rows.append((Kd, us))
            ks = np.array([r[0] / 64 for r in rows]); us = np.array([r[1] for r in rows])
            A = np.stack([np.ones_like(ks), ks], 1)
            (a0, b0), *_ = np.linalg.lstsq(A, us, rcond=None)
            tiles128 = ((M + 127) // 128) * ((N + 127) // 128) / 256.0
            print(f"variant {variant} M {M} N {N}: " + ", ".join(f"K{k} {u:.1f}us" for k, u in rows)
                  + f";  fit {a0:.1f} us + {b0:.2f} us per 64-deep step;  per 128x128 tile on 256 CUs: fixed {a0 / tiles128 * 1e3:.0f} ns, "
                  f"step {b0 / tiles128 * 1e3:.0f} ns (512 MFMA cycles = {512 / 2.0:.0f} ns at 2.0 GHz)", flush=True)


if __name__ == "__main__":
    main()
