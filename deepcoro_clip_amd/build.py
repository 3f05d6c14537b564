"""In-tree build of the HIP C-ABI library (`libdcc_hip.so`) for gfx950.

`hipcc` cross-compiles without a GPU, so this runs in the build container; the resulting
`.so` lives next to this file (git-ignored, but shipped to the GPU box with the tree).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
OBJ_DIR = CSRC / "build"
LIB_PATH = PKG_DIR / "libdcc_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wno-unused-result", "-ffp-contract=fast",
          *os.environ.get("DCC_CFLAGS", "").split()]  # e.g. DCC_CFLAGS=-DDCC_RING_PROFILE for scratch/ring_prof.py


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _needs(obj: Path, deps: list[Path]) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


# per-file code-generation options
FILE_FLAGS = {
    # softmax runs on the MFMA results: with AGPR accumulators hipcc copies every score AGPR->VGPR->AGPR (a third of
    # the kernels' VALU instructions); VGPR-form MFMA removes the copies at the price of occupancy (measured faster)
    "attention.hip": [*os.environ.get("DCC_ATTN_FLAGS", "-mllvm -amdgpu-mfma-vgpr-form").split()],
}


def _compile(src: Path, headers: list[Path], force: bool) -> Path:
    obj = OBJ_DIR / (src.stem + ".o")
    if force or _needs(obj, [src] + headers):
        cmd = [HIPCC, *CFLAGS, *FILE_FLAGS.get(src.name, []), "-I", str(CSRC), "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = True) -> Path:
    """Compile every `csrc/*.hip` for gfx950 and link `libdcc_hip.so`. Returns the library path."""
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    headers = sorted(CSRC.glob("*.h"))
    srcs = _sources()
    if not srcs:
        raise RuntimeError("no HIP sources found")
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, headers, force), srcs))
    if force or _needs(LIB_PATH, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB_PATH), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB_PATH} from {len(objs)} objects", file=sys.stderr)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
