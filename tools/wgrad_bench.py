"""Weight-gradient products dW[n, k] = sum_m dY[m, n] X[m, k] of the MViT blocks at the headline shard (fp32 output, split-K inside
dcc_gemm): time and operand bytes per second.  usage: python tools/wgrad_bench.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import functional as Fn


def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    shapes = [("blk0 qkv", 802848, 288, 96), ("blk0 proj", 802848, 96, 96), ("blk0 fc1", 802848, 384, 96), ("blk0 fc2", 802848, 96, 384),
              ("blk1 qkv", 802848, 576, 96), ("blk1 fc1", 200736, 768, 192), ("blk1 fc2", 200736, 192, 768), ("blk2 qkv", 200736, 576, 192),
              ("blk3 qkv", 200832, 1152, 192), ("s2 qkv", 50208, 1152, 384), ("s2 fc1", 50208, 1536, 384), ("s2 fc2", 50208, 384, 1536),
              ("s2 proj", 50208, 384, 384)]
    for name, M, N, Kd in shapes:
        dy = torch.randn(M, N, device="cuda").bfloat16()
        x = torch.randn(M, Kd, device="cuda").bfloat16()
        db = torch.zeros(N, device="cuda")
        us = t(lambda: Fn.wgrad(dy, x, db))
        nbytes = 2 * M * (N + Kd)
        print(f"{name:10s} dW[{N} x {Kd}] over {M} rows: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s  {2.0 * M * N * Kd / us / 1e6:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
