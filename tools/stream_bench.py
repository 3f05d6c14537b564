"""Early-block GEMM shapes (headline shard: 32 clips of 16 x 224 x 224) on the streaming panel kernel (csrc/gemm_stream.hip) against
the tiled kernels it replaces there (variant 10: ring kernel), interleaved in one process.  Algorithmic bytes = every operand and
output once.  usage: python tools/stream_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [  # tag, M, N, K, epilogue
    ("blk0 qkv", 802848, 288, 96, dict(bias=True)),
    ("blk0 proj+res", 802848, 96, 96, dict(bias=True, resid=True)),
    ("blk0 fc1 gelu+pre", 802848, 384, 96, dict(bias=True, gelu=True, save_pre=True)),
    ("blk0 fc2+res", 802848, 96, 384, dict(bias=True, resid=True)),
    ("blk0 dpre (dgelu)", 802848, 384, 96, dict(dgelu=True)),
    ("blk0 dx fc1", 802848, 96, 384, dict()),
    ("blk0 dx qkv", 802848, 96, 288, dict()),
    ("blk1 qkv", 802848, 576, 96, dict(bias=True)),
    ("blk1 project", 802848, 192, 96, dict(bias=True)),
    ("blk1 dx qkv", 802848, 96, 576, dict()),
    ("blk1 proj+res", 200736, 192, 192, dict(bias=True, resid=True)),
    ("blk1 fc1 gelu+pre", 200736, 768, 192, dict(bias=True, gelu=True, save_pre=True)),
    ("blk1 dpre (dgelu)", 200736, 768, 192, dict(dgelu=True)),
    ("blk2 qkv", 200736, 576, 192, dict(bias=True)),
    ("blk3 qkv", 200832, 1152, 192, dict(bias=True)),
    ("blk3 project", 200832, 384, 192, dict(bias=True)),
]


def run(M, N, Kd, kw, variant):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = (torch.rand((M, Kd), device="cuda", generator=g) * 2 - 1).bfloat16()
    w = (torch.rand((N, Kd), device="cuda", generator=g) * 2 - 1).bfloat16()
    args = dict(kw)
    if args.pop("bias", False):
        args["bias"] = torch.randn(N, device="cuda")
    if args.get("save_pre") or args.get("dgelu"):
        args["aux"] = torch.randn(M, N, device="cuda").bfloat16()
    if args.pop("resid", False):
        args["resid"] = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    old = K.GEMM_VARIANT
    K.GEMM_VARIANT = variant << 8
    try:
        for _ in range(2):
            K.gemm(a, w, out=out, **args)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            K.gemm(a, w, out=out, **args)
        e.record()
        torch.cuda.synchronize()
    finally:
        K.GEMM_VARIANT = old
    return s.elapsed_time(e) / reps * 1e3


tot = [0.0, 0.0]
for tag, M, N, Kd, kw in SHAPES:
    nbytes = 2 * (M * Kd + N * Kd + M * N * (1 + int(bool(kw.get("resid"))) + int(bool(kw.get("save_pre"))) + int(bool(kw.get("dgelu")))))
    t_ring = run(M, N, Kd, kw, 10)
    t_stream = run(M, N, Kd, kw, 12)
    tot[0] += t_ring; tot[1] += t_stream
    print(f"{tag:20s} M{M} N{N} K{Kd}: tiled {t_ring:7.1f} us ({nbytes / t_ring / 1e6:5.2f} TB/s)   streaming {t_stream:7.1f} us "
          f"({nbytes / t_stream / 1e6:5.2f} TB/s, {2.0 * M * N * Kd / t_stream / 1e6:6.1f} TFLOP/s)   x{t_ring / t_stream:4.2f}", flush=True)
print(f"sum: tiled {tot[0]:.0f} us, streaming {tot[1]:.0f} us")
