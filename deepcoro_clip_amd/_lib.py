"""ctypes binding of the C-ABI HIP library (`include/dcc.h`).

The product path has no CPU fallback: if `libdcc_hip.so` is missing or a kernel returns an
error code, this raises. Tensors cross the boundary as raw device pointers + sizes + the
current HIP stream; nothing in the signatures is a torch type.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

# DCC_LIB: alternative build of the same library (kernel A/B runs from tools/, see build.build_variant)
LIB_PATH = Path(os.environ.get("DCC_LIB") or (Path(__file__).resolve().parent / "libdcc_hip.so"))

DCC_F32, DCC_BF16, DCC_F16 = 0, 1, 2
_ERR = {-1: "DCC_ERR_ARG", -2: "DCC_ERR_LAUNCH", -3: "DCC_ERR_UNSUPPORTED"}


class DccError(RuntimeError):
    pass


_LIB = None

HEADER_PATH = Path(__file__).resolve().parent.parent / "include" / "dcc.h"


def _parse_header(path: Path = HEADER_PATH) -> dict[str, list]:
    """Derive ctypes argtypes from the prototypes in include/dcc.h (single source of truth)."""
    import re

    text = path.read_text()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    sigs: dict[str, list] = {}
    for m in re.finditer(r"\bint\s+(dcc_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                elif a.startswith("int64_t"):
                    argtypes.append(ctypes.c_int64)
                elif a.startswith("uint64_t"):
                    argtypes.append(ctypes.c_uint64)
                elif a.startswith("float"):
                    argtypes.append(ctypes.c_float)
                elif a.startswith("int"):
                    argtypes.append(ctypes.c_int)
                else:
                    raise DccError(f"cannot map C type in '{a}' of {name}")
        sigs[name] = argtypes
    return sigs


SIGNATURES: dict[str, list] = _parse_header()


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise DccError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (there is no CPU fallback on the product path)."
            )
        _LIB = ctypes.CDLL(str(LIB_PATH))
        for name, argtypes in SIGNATURES.items():
            fn = getattr(_LIB, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
    return _LIB


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return DCC_F32
    if d == torch.bfloat16:
        return DCC_BF16
    if d == torch.float16:  # dcc_cast only: fp16 wire format of the SigLIP embedding gather
        return DCC_F16
    raise DccError(f"unsupported dtype {d}")


def ptr(t):
    """Device address as a plain int (argtypes are c_void_p, ctypes converts; no c_void_p object per argument)."""
    return None if t is None else t.data_ptr()


def stream() -> int:
    """Raw hipStream_t of torch's current stream (the C getters: `torch.cuda.current_stream()` builds a Stream object
    and resolves the device through several Python layers, ~9 us x several hundred launches per step on the host)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DccError("deepcoro_clip_amd ops run on the MI355X HIP path only: got a CPU tensor "
                           "(there is no CPU fallback; the CPU oracle lives under oracle/ for tests)")


def call(name: str, *args) -> None:
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise DccError(f"{name} failed with {_ERR.get(rc, rc)}")


def try_call(name: str, *args) -> bool:
    """For SPECIALISED entry points that have a general HIP counterpart (fused pool + LayerNorm, fused rel-pos forward): False
    when the kernel reports DCC_ERR_UNSUPPORTED for this shape, so that the caller launches the general HIP kernels instead;
    every other failure raises.  (Still no CPU fallback: both alternatives are entries of the same library.)"""
    rc = getattr(lib(), name)(*args)
    if rc == -3:
        return False
    if rc != 0:
        raise DccError(f"{name} failed with {_ERR.get(rc, rc)}")
    return True
