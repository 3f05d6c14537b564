import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, argparse
import bench
from deepcoro_clip_amd import functional as Fn, kernels as K
args = argparse.Namespace(gpus=1, steps=5, warmup=2, batch=24, videos=1, frames=16, size=224, seq=32, dtype="bf16", pooling="mean")
runner, batch = bench.build_job(args, torch.device("cuda", 0), 1, 0)
ve = runner.video_encoder; m = ve.model
x = batch["videos"]
def run():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(19)]
    with torch.no_grad():
        ev[0].record()
        t = Fn.patch_embed(x, m.conv_proj.weight, m.conv_proj.bias, m.pos_encoding.class_token, torch.bfloat16)
        ev[1].record()
        for i, blk in enumerate(m.blocks):
            t = blk.forward_tokens(t, 24, rope=None)
            ev[i + 2].record()
        torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(17)]
for _ in range(2): run()
r = run()
print("patch_embed %.2f ms" % r[0]); print("blocks fwd ms:", " ".join(f"{v:.2f}" for v in r[1:]), " total %.2f" % sum(r))
