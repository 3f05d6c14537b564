"""Isolated timings of the q/k/v pooling branch kernels of one MViT block at the headline shapes (32 clips):
depthwise 3x3x3 conv pool forward, per-head LayerNorm forward / backward, pool data gradient, pool weight gradient.
Prints microseconds and the rate at ALGORITHMIC bytes (each tensor read / written once).

usage: python tools/pool_bench.py [stage ...]   (default: all block geometries of MViT-v2-S at 16 x 224 x 224)
"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import kernels as K

CLIPS = int(os.environ.get("POOL_BENCH_CLIPS", "32"))
# name: (heads, thw, q stride, kv stride)
BLOCKS = {
    "blk0": (1, (8, 56, 56), 1, 8),
    "blk1": (2, (8, 56, 56), 2, 4),
    "blk2": (2, (8, 28, 28), 1, 4),
    "blk3": (4, (8, 28, 28), 2, 2),
    "blk4-13": (4, (8, 14, 14), 1, 2),
    "blk14": (8, (8, 14, 14), 2, 1),
    "blk15": (8, (8, 7, 7), 1, 1),
}


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def line(label, us, nbytes):
    # the loop re-runs one launch on the same buffers: a working set below the 256 MiB Infinity Cache is served from it, so
    # such a line's TB/s is a cache rate (it can exceed the 8 TB/s HBM peak), not HBM evidence
    note = "  [< 256 MiB: Infinity-Cache resident, not an HBM rate]" if nbytes < 256 * 2**20 else ""
    print(f"  {label:34s} {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s  ({nbytes / 1e6:7.1f} MB){note}", flush=True)


def main():
    names = sys.argv[1:] or list(BLOCKS)
    dev = "cuda"
    D = 96
    for name in names:
        H, thw, sq, skv = BLOCKS[name]
        B = CLIPS
        N = thw[0] * thw[1] * thw[2] + 1
        qkv = (torch.randn(B * N, 3 * H * D, device=dev) * 0.5).bfloat16()
        qkv5 = qkv.view(B, N, 3, H, D)
        print(f"{name}: {B} clips, N {N}, heads {H}, grid {thw}, q stride {sq}, kv stride {skv}")
        for br, (i, s) in (("q", (0, sq)), ("kv", (1, skv))):
            g = K.PoolGeom(B, H, D, thw, (3, 3, 3), (1, s, s), (1, 1, 1), True)
            w = torch.randn(D, 27, device=dev) * 0.2
            gam = torch.rand(D, device=dev) + 0.5
            bet = torch.randn(D, device=dev) * 0.1
            x = qkv5[:, :, i]
            conv, _ = K.pool_fwd(x, g, w)
            rows = B * g.n_out * H
            e_in, e_out = B * N * H * D * 2, rows * D * 2
            line(f"{br} pool fwd (s{s})", t(lambda: K.pool_fwd(x, g, w, out=conv)), e_in + e_out)
            y, mean, rstd = K.layernorm_fwd(conv.view(rows, D), gam, bet, 1e-6)
            if K.pool_ln_supported(x, g):
                y2 = torch.empty_like(y)
                conv2, mean2, rstd2 = K.pool_ln_fwd(x, g, w, gam, bet, 1e-6, y2)
                torch.cuda.synchronize()
                dc = (conv2.float() - conv.float()).abs().max().item()
                dyv = (y2.float() - y.float()).abs().max().item()
                dm = (mean2 - mean).abs().max().item()
                line(f"{br} pool+LN fused (s{s})", t(lambda: K.pool_ln_fwd(x, g, w, gam, bet, 1e-6, y2)), e_in + 2 * e_out)
                print(f"      fused vs separate: conv max diff {dc:.3g}, LN out max diff {dyv:.3g}, mean max diff {dm:.3g}", flush=True)
            line(f"{br} LayerNorm(96) fwd", t(lambda: K.layernorm_fwd(conv.view(rows, D), gam, bet, 1e-6, out=y)), 2 * e_out)
            dy = torch.randn_like(y)
            dgam = torch.zeros(D, device=dev); dbet = torch.zeros(D, device=dev)
            dconv = K.layernorm_bwd(dy, conv.view(rows, D), mean, rstd, gam, dgam, dbet)
            line(f"{br} LayerNorm(96) bwd", t(lambda: K.layernorm_bwd(dy, conv.view(rows, D), mean, rstd, gam, dgam, dbet, dx=dconv)), 3 * e_out)
            dqkv = torch.empty_like(qkv5)
            dconv4 = dconv.view(B, g.n_out, H, D)
            line(f"{br} pool bwd data", t(lambda: K.pool_bwd_data(dconv4, dqkv[:, :, i], g, w)), e_in + e_out)
            dw = torch.zeros(D, 27, device=dev)
            line(f"{br} pool bwd weight", t(lambda: K.dwconv_bwd_weight(dconv4, x, g, dw)), e_in + e_out)
            del conv, y, dy, dconv, dqkv
        del qkv


if __name__ == "__main__":
    main()
