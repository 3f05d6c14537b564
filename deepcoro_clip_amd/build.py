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
          *os.environ.get("DCC_CFLAGS", "").split()]  # e.g. DCC_CFLAGS=-DDCC_RING_PROFILE (crew timestamps of the ring GEMM)


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


def _compile(src: Path, headers: list[Path], force: bool, obj_dir: Path = None, extra: dict | None = None) -> Path:
    obj = (obj_dir or OBJ_DIR) / (src.stem + ".o")
    if force or _needs(obj, [src] + headers):
        cmd = [HIPCC, *CFLAGS, *FILE_FLAGS.get(src.name, []), *(extra or {}).get(src.name, []), "-I", str(CSRC), "-c", str(src),
               "-o", str(obj)]
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


def build_variant(name: str, file_flags: dict[str, list[str]]) -> Path:
    """A second copy of the library with extra per-file compiler flags (e.g. {"gemm_ring.hip": ["-DRING_X=1"]}) as
    `libdcc_hip_<name>.so`, for A/B runs of one kernel from tools/ (select it with DCC_LIB=<path>).  Files without extra
    flags reuse the main build's objects."""
    build(verbose=False)
    vdir = CSRC / f"build_{name}"
    vdir.mkdir(parents=True, exist_ok=True)
    headers = sorted(CSRC.glob("*.h"))
    objs = []
    for src in _sources():
        if src.name in file_flags:
            objs.append(_compile(src, headers, True, vdir, file_flags))
        else:
            objs.append(OBJ_DIR / (src.stem + ".o"))
    lib = vdir / f"libdcc_hip_{name}.so"  # outside the package directory proper: not picked up as a module
    r = subprocess.run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(lib), *map(str, objs)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--variant":  # build.py --variant NAME file.hip=-DFOO=1,-mllvm,-x ...
        flags = {}
        for spec in sys.argv[3:]:
            f, v = spec.split("=", 1)
            flags[f] = [x for x in v.split(",") if x]
        print(build_variant(sys.argv[2], flags))
    else:
        build(force="--force" in sys.argv)
