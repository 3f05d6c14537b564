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
