"""Weight-gradient products of a stage-2 MViT block, dW[n, k] = sum_m dY[m, n] X[m, k] (fp32 output, split-K + fp32 atomics), on the
three tiled kernels at explicit split factors — what the split-K atomics cost and whether a 256-wide tile (the eight-phase kernel's)
could pay.  Variants interleaved in one process.
usage: python tools/wgrad_split_bench.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K

SHAPES = [("s2 fc1", 50208, 1536, 384), ("s2 fc2", 50208, 384, 1536), ("s2 qkv", 50208, 1152, 384), ("s2 proj", 50208, 384, 384)]
VARIANTS = [("ring 128x128", 10), ("wide 256x128", 9), ("wide 256x256", 8)]


def main():
    for name, M, N, Kd in SHAPES:
        dy = torch.randn(M, N, device="cuda").bfloat16()
        x = torch.randn(M, Kd, device="cuda").bfloat16()
        out = torch.zeros(N, Kd, device="cuda")
        ref = None
        print(f"{name}: dW[{N} x {Kd}] over {M} rows, {2.0 * M * N * Kd / 1e9:.1f} GFLOP; output {N * Kd * 4 / 1e6:.2f} MB fp32")
        cases = []
        for vname, v in VARIANTS:
            tile = (128, 128) if v == 10 else ((256, 128) if v == 9 else (256, 256))
            tiles = -(-N // tile[0]) * -(-Kd // tile[1])
            for sk in sorted({1, 2, 4, max(1, 128 // tiles), max(1, 256 // tiles), max(1, 512 // tiles)}):
                cases.append((vname, v, sk, tiles))
        times = {c: [] for c in cases}
        for rnd in range(6):
            for c in cases:
                vname, v, sk, tiles = c
                K.GEMM_VARIANT = v << 8
                fn = lambda: K.gemm(dy, x, a_km=True, b_km=True, out=out, accumulate=sk > 1, splitk=sk)  # noqa: E731
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[c].append(e0.elapsed_time(e1) * 200.0)
        K.GEMM_VARIANT = 0
        for c in cases:
            vname, v, sk, tiles = c
            t = sorted(times[c])[len(times[c]) // 2]
            atom_mb = (sk * N * Kd * 4 / 1e6) if sk > 1 else 0.0
            print(f"   {vname:13s} tiles {tiles:3d} x split {sk:3d} = {tiles * sk:4d} items: {t:7.1f} us  {2.0 * M * N * Kd / t / 1e6:6.1f} TFLOP/s   atomics {atom_mb:6.1f} MB")


if __name__ == "__main__":
    main()
