"""LayerNorm backward (csrc/norm.hip) at the step's shapes: time and bytes per second (dy + x read, dx written).  The launch knobs are
read once per process: usage  DCC_LNB_U=4 DCC_LNB_CAP_WS=1024 python tools/ln_bench.py"""
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
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("DCC_LN"))
    out = []
    for name, rows, C in (("blk0 q", 802848, 96), ("s2 q", 200832, 96), ("s2 k/v", 50304, 96), ("s2 LN1/LN2", 50208, 384), ("blk0 LN", 802848, 96), ("s3 LN", 12576, 768)):
        x = torch.randn(rows, C, device="cuda").bfloat16()
        dy = torch.randn(rows, C, device="cuda").bfloat16()
        g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        _, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6)
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        us = t(lambda: K.layernorm_bwd(dy, x, mean, rstd, g, dg, db))
        out.append(f"{name} [{rows}x{C}] {us:6.1f} us {3 * rows * C * 2 / us / 1e6:4.2f} TB/s")
    print(f"[{tag}] " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
