"""Where the HOST time of a training step goes (headline shard), and which torch (ATen) ops still launch kernels beside the C-ABI
kernels: (1) a census of torch-level tensor ops per call site over one step (each is an ATen launch or a host-side allocation),
(2) cProfile of four steps (top functions by own time).  usage: python tools/host_census.py"""
import cProfile, collections, io, os, pstats, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sys.argv = [sys.argv[0]]
ap_args = type("A", (), dict(batch=8, videos=4, frames=16, size=224, seq=32, dtype="bf16", pooling="attention", graph="off", warmup=3,
                             loss="contrastive", rope=False, gemm_dtype="bf16", micro_batch=0, config=""))()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
runner, batch = bench.build_job(ap_args, dev, 1, 0)
for _ in range(3):
    runner._train_step(**batch)
torch.cuda.synchronize()

COUNTS = collections.Counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def site():
    for fr in reversed(traceback.extract_stack(limit=12)[:-2]):
        if fr.filename.startswith(ROOT) and "host_census" not in fr.filename:
            return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
    return "?"


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        COUNTS[(name, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)
    return orig


saved = []
for owner, names in ((torch, ["zeros", "zeros_like", "ones", "full", "cat", "stack", "tensor", "bernoulli", "where"]),
                     (torch.Tensor, ["zero_", "fill_", "add_", "mul_", "div_", "copy_", "clone", "contiguous", "float", "to", "__setitem__",
                                     "__getitem__", "sum", "expand", "reshape"])):
    for n in names:
        saved.append((owner, n, wrap(owner, n)))
runner._train_step(**batch)
torch.cuda.synchronize()
for owner, n, orig in saved:
    setattr(owner, n, orig)
print("== torch-level ops per step by call site (views like __getitem__/expand/reshape launch nothing; listed for host cost)")
for (name, where), c in COUNTS.most_common(70):
    print(f"{c:5d}  {name:12s} {where}")

pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    runner._train_step(**batch)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print("== cProfile, 4 steps, by own time")
print(s.getvalue()[:9000])
