"""Per-block GPU time of the MViT-v2-S tower at the headline shape (32 clips of 16 x 224 x 224, bf16):
forward from events recorded between blocks, backward from events recorded when each block's input gradient arrives.
Tells which stage to work on (the stages have equal FLOPs per block but very different arithmetic intensity).

usage: python tools/block_times.py [clips]
       python tools/block_times.py clips --only I [reps]   forward + backward of block I alone on a random input
       (under rocprofv3 --kernel-trace --stats: the per-kernel breakdown of that block)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import functional as Fn
from deepcoro_clip_amd.mvit import forward_macs
from deepcoro_clip_amd.video_encoder import VideoEncoder


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = "cuda:0"
    ve = VideoEncoder(num_frames=16, spatial_size=(224, 224), pretrained=False, freeze_ratio=1.0, dropout=0.0,
                      compute_dtype=torch.bfloat16, stochastic_depth_prob=0.0, token_pooling_mode="mean").to(dev)
    ve.train()
    m = ve.model
    if "--only" in sys.argv:
        i = int(sys.argv[sys.argv.index("--only") + 1])
        reps = int(sys.argv[sys.argv.index("--only") + 2]) if len(sys.argv) > sys.argv.index("--only") + 2 else 3
        blk = m.blocks[i]
        t, h, w = blk.p["thw"]
        x = (torch.randn(clips * (t * h * w + 1), blk.p["in_ch"], device=dev) * 0.5).bfloat16().requires_grad_(True)
        for _ in range(reps):
            y = blk.forward_tokens(x, clips, rope=m._rope(blk, x.device))
            y.backward(torch.randn_like(y))
        torch.cuda.synchronize()
        print(f"block {i}: {reps} x (fwd + bwd) done")
        return
    video = torch.randn(clips, 16, 224, 224, 3, device=dev)
    macs = forward_macs(m.plan) if hasattr(m, "plan") else None

    def run(record):
        ev = lambda: torch.cuda.Event(enable_timing=True)
        fwd = [ev()]
        bwd = []
        fwd[0].record()
        x = Fn.patch_embed(video, m.conv_proj.weight, m.conv_proj.bias, m.pos_encoding.class_token, torch.bfloat16, None)
        e = ev(); e.record(); fwd.append(e)
        for blk in m.blocks:
            if record:
                def hook(g, _b=bwd):
                    e2 = ev(); e2.record(); _b.append(e2)
                    return g
                x.register_hook(hook)
            x = blk.forward_tokens(x, clips, rope=m._rope(blk, video.device))
            e = ev(); e.record(); fwd.append(e)
        x = Fn.layernorm(x, m.norm.weight, m.norm.bias, m.eps)
        loss_grad = torch.randn_like(x)
        b0 = ev(); b0.record()
        x.backward(loss_grad)
        b1 = ev(); b1.record()
        torch.cuda.synchronize()
        return fwd, [b0] + bwd + [b1]

    for _ in range(2):
        run(False)
        for p in ve.parameters():
            p.grad = None
    fwd, bwd = run(True)
    print(f"{clips} clips; patch embed fwd {fwd[0].elapsed_time(fwd[1]) * 1e3:8.1f} us")
    nb = len(m.blocks)
    tf = tb = 0.0
    for i, blk in enumerate(m.blocks):
        f = fwd[i + 1].elapsed_time(fwd[i + 2]) * 1e3
        # bwd events: b0, then hooks fire in reverse block order (input grad of block nb-1 first), then b1
        b = bwd[nb - 1 - i].elapsed_time(bwd[nb - i]) * 1e3
        p = blk.p
        tf += f; tb += b
        print(f"blk{i:2d} C {p['in_ch']:3d}->{p['out_ch']:3d} thw {p['thw']} q{p['stride_q'][1]} kv{p['stride_kv'][1]}: "
              f"fwd {f:8.1f} us   bwd {b:8.1f} us")
    print(f"blocks fwd {tf / 1e3:.2f} ms, bwd {tb / 1e3:.2f} ms (bwd of blk0 also holds the patch-embed backward)")


if __name__ == "__main__":
    main()
