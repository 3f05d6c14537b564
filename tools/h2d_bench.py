"""PCIe-inclusive step rate of the headline shard: `_run_epoch('train')` over host-resident batches (pinned memory, as
DataLoader(pin_memory=True) delivers them), with the one-batch-lookahead prefetch on its side stream and without it, fp32 frames
(308 MB per batch) and raw uint8 frames (77 MB, normalised inside the patch gather).  bench.py's `value` keeps the batch resident
(its contract); this is the number with the host -> device copy in the loop.  usage: python tools/h2d_bench.py [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
args = type("A", (), dict(batch=8, videos=4, frames=16, size=224, seq=32, dtype="bf16", pooling="attention", graph="off", warmup=3,
                          loss="contrastive", rope=False, gemm_dtype="bf16", micro_batch=0, config=""))()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
runner, batch = bench.build_job(args, dev, 1, 0)
runner.config.video_mean, runner.config.video_std = [110.2, 101.5, 95.25], [60.0, 58.5, 62.25]
for _ in range(4):
    runner._train_step(**batch)
torch.cuda.synchronize()
g = torch.Generator().manual_seed(0)
ids = torch.randint(0, 30522, (8, 32), generator=g).pin_memory()
mask = torch.ones(8, 32, dtype=torch.int64).pin_memory()
for kind in ("fp32", "uint8"):
    if kind == "fp32":
        vids = [torch.randn(8, 4, 16, 224, 224, 3, generator=g).pin_memory() for _ in range(2)]
    else:
        vids = [torch.randint(0, 256, (8, 4, 16, 224, 224, 3), generator=g, dtype=torch.uint8).pin_memory() for _ in range(2)]
    loader = [{"videos": vids[i % 2], "encoded_texts": {"input_ids": ids, "attention_mask": mask}} for i in range(nb)]
    for prefetch in (False, True, False, True):
        runner._prefetch = prefetch
        runner.train_loader = loader
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner._run_epoch("train", 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nb
        print(f"{kind:5s} frames ({vids[0].numel() * vids[0].element_size() / 1e6:5.0f} MB per batch), prefetch {'on ' if prefetch else 'off'}: "
              f"{1e3 * dt:6.2f} ms per step incl. host->device = {32 / dt:6.1f} clips/s", flush=True)
