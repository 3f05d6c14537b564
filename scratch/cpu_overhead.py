"""How long does the host need to ENQUEUE one training step (no waiting for the GPU)?"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, argparse
import bench
args = argparse.Namespace(gpus=1, steps=5, warmup=2, batch=24, videos=1, frames=16, size=224, seq=32, dtype="bf16", pooling="mean")
dev = torch.device("cuda", 0)
runner, batch = bench.build_job(args, dev, 1, 0)
import deepcoro_clip_amd.runner as R
orig = R.VideoContrastiveLearningRunner._step_metrics
stamps = []
def patched(self, *a, **k):
    stamps.append(time.perf_counter())   # everything of the step has been enqueued here; the .tolist() below syncs
    return orig(self, *a, **k)
R.VideoContrastiveLearningRunner._step_metrics = patched
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    runner._train_step(**batch)
    t1 = time.perf_counter()
    print(f"step {i}: enqueue {1e3*(stamps[-1]-t0):.1f} ms, total {1e3*(t1-t0):.1f} ms", flush=True)
