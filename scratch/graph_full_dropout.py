"""Full-size step with dropout 0.1 + stochastic depth 0.2 replayed from the HIP graph: runs, losses finite and moving."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, argparse
import bench
args = argparse.Namespace(gpus=1, steps=5, warmup=5, batch=24, videos=1, frames=16, size=224, seq=32, dtype="bf16", pooling="mean", graph="on")
runner, batch = bench.build_job(args, torch.device("cuda", 0), 1, 0)
ve = runner.video_encoder
ve.dropout = 0.1
blocks = ve.model.blocks
for i, b in enumerate(blocks): b.sd_prob = 0.2 * i / (len(blocks) - 1)
losses = []
for i in range(10):
    m, _ = runner._train_step(**batch)
    losses.append(m["loss"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): runner._train_step(**batch)
torch.cuda.synchronize()
print("losses", [round(x, 4) for x in losses], "captured", any(s["graph"] is not None for s in runner._graphs.values()), f"{1e3*(time.perf_counter()-t0)/10:.2f} ms/step")
