"""The software-pipelined dK/dV kernel (csrc/attention.hip, attn_bwd_dkv_pipe_kernel) forced on for every bf16 case of
test_attention_gpu.py — short query ranges, odd tile counts, query splits — in a child process (the choice is read once per process
from DCC_ATTN_PIPE).  The default heuristic only selects it for blocks streaming >= 768 queries."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("force", ["1", "0"])
def test_attention_cases_with_pipelined_kernel_forced(force):
    env = dict(os.environ, DCC_ATTN_PIPE=force)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_attention_gpu.py"), "-q", "-m", "gpu", "-x",
                        "-k", "test_attention_fwd_bwd", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_pipelined_kernel_matches_unpipelined_on_random_shapes(tmp_path):
    """Same summation order, so bit-exact when the query range is not split (tools/attn_pipe_fuzz.py: 24 random shapes, D = 128 / 160,
    1 .. 1400 queries, 1 .. 450 keys, query splits 1 / 2 / 3 / 5, with and without the residual-pooling shortcut)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import attn_pipe_fuzz

    assert attn_pipe_fuzz.compare(n=24, tmp=str(tmp_path)) == []
