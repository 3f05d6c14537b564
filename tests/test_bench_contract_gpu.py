"""bench.py's output contract (one JSON line on stdout with the fields the driver reads), exercised on a small geometry."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "2",
           "--frames", "4", "--size", "64", "--seq", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["vs_baseline"] is None and out["dtype"] == "bf16" and out["data"] == "synthetic"
    # default workload = the headline multi-video config (N = 4 videos per study, attention pooling)
    assert out["config"]["clips_per_step"] == 8 and "multi-video" in out["config"]["workload"]
    assert out["value"] > 0 and abs(out["value"] - 8 / (out["ms_per_step"] * 1e-3)) < 1e-2 * out["value"]
    assert "workload" in out["config"] and "model" not in out["config"]
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and "traffic" in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # bf16: the headline fraction is the MFMA one (SURVEY §8d classifies the GEMM family as MFMA-bound); the HBM view sits beside it
    assert rf["bound"] == "mfma" and rf["frac"] == rf["mfma"]["frac"] and rf["binding_by_bytes_moved"] in ("hbm", "mfma")
    assert "traffic_source" in rf and "instrumented" in rf["roofline_pass"] and 0 < rf["video_fwd_frac"] < 1
    cb = out["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"] and cb["unit"]


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """The N > 1 launch path (torch.distributed.run, one process per rank, HipDDP arena + bucketed all-reduce, global
    loss over gathered embeddings) rehearsed with two ranks sharing this GPU over gloo (RCCL needs one GPU per rank)."""
    env = dict(os.environ, DCC_BENCH_BACKEND="gloo", DCC_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2",
           "--frames", "4", "--size", "64", "--seq", "16", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["parallelism"] == "dp2"
    assert out["config"]["global_batch"] == 4 and out["value"] > 0
    import math
    assert abs(out["config"]["loss_first"] - math.log(4)) < 0.2  # InfoNCE over the GLOBAL 4 x 4 similarity matrix at init
