"""GPU and host time per phase of the step (forward video / text, loss, backward, step tail)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, argparse
import bench
args = argparse.Namespace(gpus=1, steps=5, warmup=2, batch=24, videos=1, frames=16, size=224, seq=32, dtype="bf16", pooling="mean")
dev = torch.device("cuda", 0)
runner, batch = bench.build_job(args, dev, 1, 0)
for _ in range(3): runner._train_step(**batch)
torch.cuda.synchronize()
import deepcoro_clip_amd.runner as R
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, time.perf_counter(), e))
orig_encode, orig_loss = runner._encode, runner._loss
def enc(*a, **k):
    mark("start"); r = orig_encode(*a, **k); mark("fwd_enqueued"); return r
def loss(*a, **k):
    r = orig_loss(*a, **k); mark("loss_enqueued"); return r
runner._encode, runner._loss = enc, loss
oc = R.GradClipper.clip_
def clip(self, *a, **k):
    mark("bwd_enqueued"); return oc(self, *a, **k)
R.GradClipper.clip_ = clip
osm = runner._step_metrics
def sm(*a, **k):
    mark("tail_enqueued"); return osm(*a, **k)
runner._step_metrics = sm
for it in range(3):
    marks.clear()
    runner._train_step(**batch); torch.cuda.synchronize()
    t0 = marks[0][1]; e0 = marks[0][2]
    print(" | ".join(f"{n}: host {1e3*(t-t0):.1f} gpu {e0.elapsed_time(e):.1f}" for n, t, e in marks[1:]), flush=True)
