"""Isolated timings of the towers' bf16 GEMM shapes (headline config: 32 clips -> M = 50 208 rows in blocks 4-13).
usage: [DCC_LIB=path/to/variant.so] python tools/gemm_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def bench(M, N, Kd, a_km=False, b_km=False, tag="", **kw):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = (torch.rand((Kd, M) if a_km else (M, Kd), device="cuda", generator=g) * 2 - 1).bfloat16()
    b = (torch.rand((Kd, N) if b_km else (N, Kd), device="cuda", generator=g) * 2 - 1).bfloat16()
    od = kw.pop("out_dtype", torch.bfloat16)
    args = dict(kw)
    if args.pop("bias", False):
        args["bias"] = torch.randn(N, device="cuda")
    if args.get("save_pre") or args.get("dgelu"):
        args["aux"] = torch.randn(M, N, device="cuda").bfloat16()
    if args.pop("resid", False):
        args["resid"] = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty((M, N), device="cuda", dtype=od) if not (args.get("splitk", 1) > 1) else torch.zeros((M, N), device="cuda", dtype=od)
    if args.get("splitk", 1) > 1:
        args["accumulate"] = True
    for _ in range(3):
        K.gemm(a, b, a_km=a_km, b_km=b_km, out=out, **args)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        K.gemm(a, b, a_km=a_km, b_km=b_km, out=out, **args)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"{tag:28s} M{M} N{N} K{Kd} {'T' if a_km else 'N'}{'T' if b_km else 'N'}: {ms * 1e3:8.1f} us  {2 * M * N * Kd / ms / 1e9:7.1f} TFLOP/s", flush=True)


M = 50208
VAR = int(os.environ.get("DCC_GEMM_VARIANT", "0"))


def wsplit(rows, n, k):
    if VAR in (8, 9, 11):  # 256 x (256 | 128) tiles: about one item per CU
        bn = 256 if VAR == 8 else 128
        tiles = ((n + 255) // 256) * ((k + bn - 1) // bn)
        return max(1, min(256 // tiles, rows // 256))
    return K.wgrad_splitk(rows, n, k)


bench(M, 1536, 384, tag="fc1 fwd bias+gelu+pre", bias=True, gelu=True, save_pre=True)
bench(M, 384, 1536, tag="fc2 fwd bias+resid", bias=True, resid=True)
bench(M, 1152, 384, tag="qkv fwd bias", bias=True)
bench(M, 384, 384, tag="proj fwd bias+resid", bias=True, resid=True)
bench(M, 1536, 384, b_km=True, tag="dpre = dy W2 * gelu'", dgelu=True)
bench(M, 384, 1536, b_km=True, tag="dx = dpre W1")
bench(M, 384, 1152, b_km=True, tag="dx = dqkv Wqkv")
bench(M, 1536, 384, tag="plain K384")
bench(1536, 384, M, a_km=True, b_km=True, tag="wgrad fc2", out_dtype=torch.float32, splitk=wsplit(M, 1536, 384))
bench(384, 1536, M, a_km=True, b_km=True, tag="wgrad fc1", out_dtype=torch.float32, splitk=wsplit(M, 384, 1536))
bench(1152, 384, M, a_km=True, b_km=True, tag="wgrad qkv", out_dtype=torch.float32, splitk=wsplit(M, 1152, 384))
bench(200832, 576, 192, tag="blk2 qkv", bias=True)
bench(802848, 288, 96, tag="blk0 qkv", bias=True)
bench(8192, 8192, 8192, tag="8192^3")
bench(4096, 4096, 4096, tag="4096^3")


def bench_fp8(M, N, Kd, tag="", fmt_a=0, **kw):
    """fp8 column (BASELINE configs[4]): the block-scaled-MFMA GEMM alone on pre-quantised operands, and the same GEMM
    with the per-tensor quantisation of its activation operand (absmax + cast: two passes over A) that a layer pays in situ."""
    g = torch.Generator(device="cuda").manual_seed(0)
    a = (torch.rand((M, Kd), device="cuda", generator=g) * 2 - 1).bfloat16()
    b = (torch.rand((N, Kd), device="cuda", generator=g) * 2 - 1).bfloat16()
    args = dict(kw)
    if args.pop("bias", False):
        args["bias"] = torch.randn(N, device="cuda")
    if args.get("save_pre") or args.get("dgelu"):
        args["aux"] = torch.randn(M, N, device="cuda").bfloat16()
    if args.pop("resid", False):
        args["resid"] = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    a8, sa = K.fp8_quantize(a, fmt_a)
    b8, sb = K.fp8_quantize(b, 0)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps

    ms = timed(lambda: K.gemm_fp8(a8, b8, sa, sb, fmt_a=fmt_a, out=out, **args))

    def with_quant():
        q, s_ = K.fp8_quantize(a, fmt_a)
        K.gemm_fp8(q, b8, s_, sb, fmt_a=fmt_a, out=out, **args)

    msq = timed(with_quant)
    print(f"fp8 {tag:24s} M{M} N{N} K{Kd}: {ms * 1e3:8.1f} us  {2 * M * N * Kd / ms / 1e9:7.1f} TFLOP/s   "
          f"with A quantised in line {msq * 1e3:8.1f} us  {2 * M * N * Kd / msq / 1e9:7.1f} TFLOP/s", flush=True)


if os.environ.get("DCC_GEMM_VARIANT", "0") == "0":
    print("== fp8 (block-scaled MFMA, per-tensor scales; e4m3 activations / weights, e5m2 gradients)")
    bench_fp8(M, 1536, 384, tag="fc1 fwd bias+gelu+pre", bias=True, gelu=True, save_pre=True)
    bench_fp8(M, 384, 1536, tag="fc2 fwd bias+resid", bias=True, resid=True)
    bench_fp8(M, 1152, 384, tag="qkv fwd bias", bias=True)
    bench_fp8(M, 384, 384, tag="proj fwd bias+resid", bias=True, resid=True)
    bench_fp8(M, 1536, 384, tag="dpre = dy W2 * gelu'", fmt_a=1, dgelu=True)
    bench_fp8(M, 384, 1536, tag="dx = dpre W1", fmt_a=1)
    bench_fp8(M, 384, 1152, tag="dx = dqkv Wqkv", fmt_a=1)
    M32 = 8 * 4 * 3137   # configs[4] shard: 4 studies x 8 clips of 32 frames -> 3137 tokens per clip in blocks 4-13
    bench_fp8(M32, 1536, 384, tag="cfg4 fc1 fwd", bias=True, gelu=True, save_pre=True)
    bench_fp8(M32, 384, 1536, tag="cfg4 fc2 fwd", bias=True, resid=True)
    bench_fp8(8192, 8192, 8192, tag="8192^3")
    bench_fp8(4096, 4096, 4096, tag="4096^3")
