"""MI355X-native (gfx950) video-text contrastive training path for DeepCORO_CLIP.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed/RCCL).
Arithmetic: hand-written HIP kernels behind the C ABI of `include/dcc.h`.
"""
__all__ = ["build"]


def build(force: bool = False):
    from .build import build as _b

    return _b(force=force)
