import sys, os, time, cProfile, pstats; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, argparse
import bench
args = argparse.Namespace(gpus=1, steps=5, warmup=2, batch=24, videos=1, frames=16, size=224, seq=32, dtype="bf16", pooling="mean")
runner, batch = bench.build_job(args, torch.device("cuda", 0), 1, 0)
for _ in range(3): runner._train_step(**batch)
torch.cuda.synchronize()
ve = runner.video_encoder
pr = cProfile.Profile()
with torch.no_grad():
    for _ in range(2): ve(batch["videos"])
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(3): ve(batch["videos"])
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
