"""How many CPU threads the oracle should use on this box: torch's default pool size, the cgroup share, and the time of one tiny
oracle training step (4 x 64 x 64 clips) and one 16 x 224 x 224 forward+backward at several pool sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.losses import contrastive_loss
from oracle.model import TextEncoderOracle, VideoEncoderOracle, init_by_name

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch default threads", torch.get_num_threads(), flush=True)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
small = dict(vocab=512, hidden=768, layers=2, heads=12, ffn=3072, max_pos=64)
vo = init_by_name(VideoEncoderOracle(4, (64, 64))); to = init_by_name(TextEncoderOracle(**small))
lt = torch.nn.Parameter(torch.log(torch.tensor([0.07])))
g = torch.Generator().manual_seed(0)


def step():
    videos = torch.randn(4, 4, 64, 64, 3, generator=g); ids = torch.randint(0, 512, (4, 16), generator=g); mask = torch.ones(4, 16, dtype=torch.int64)
    loss = contrastive_loss(vo(videos), to(ids, mask), lt); loss.backward()


big = init_by_name(VideoEncoderOracle(16, (224, 224)))
xb = torch.randn(1, 16, 224, 224, 3)
default = torch.get_num_threads()
for nt in (default, 32, 16, 8):
    torch.set_num_threads(nt)
    step()
    t = time.time(); [step() for _ in range(3)]; t1 = (time.time() - t) / 3
    t = time.time(); big(xb).sum().backward(); t2 = time.time() - t
    print(f"{nt:4d} threads: tiny step {t1:6.2f} s, 16x224x224 fwd+bwd (1 clip, fp32) {t2:6.2f} s", flush=True)
