import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_share() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup quota (a GPU box hands 16 cores of a much larger
    host to one GPU; torch sizes its intra-op pool by the HOST's core count, and an oversubscribed pool makes the CPU oracle
    several times slower — r03: 3.9 s per tiny oracle training step on the box against 0.43 s in the 8-CPU build container)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                cores = min(cores, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except Exception:
            pass
    return max(1, min(cores, 32))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    want = _cpu_share()
    had = torch.get_num_threads()
    if had != want:
        torch.set_num_threads(want)
    print(f"[conftest] torch intra-op threads {had} -> {want} (cpu_count {os.cpu_count()})", file=sys.stderr)


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
