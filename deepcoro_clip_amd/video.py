"""Input side on the GPU: the frame pipeline of the reference's `utils/video.py:load_video` (:142-300)
from DECODED uint8 frames onward — frame count handling (:243-251), `v2.Resize((s, s), antialias=True)` (:253-255),
grayscale expansion (:213-225), `v2.Normalize` (:272-284), [F, H, W, C] output (:293-295) — as one HIP kernel
(`dcc_resize_frames_u8`) instead of four fp32 passes on the host.  Decoding itself (cv2 / ffmpeg) is out of scope: `load_video`
here takes the reference's `.npy` branch (:183-190); other containers raise.

The clip crosses PCIe as raw bytes at its native size (a 512 x 512 x 16 clip: 12.6 MB instead of 9.6 MB of fp32 per 224^2
clip AFTER the host did the work), and either comes out as the fp32 NDHWC tensor `VideoEncoder` takes, or stays uint8
(`out_uint8=True`) for the bytes-in path of `dcc_im2col3d_u8`, which normalises inside the patch gather.
Random augmentations (`video_transforms`, `rand_augment`) are host-side torchvision code in the reference and not offered."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import kernels as K


def frame_indices(t: int, expected: int) -> list[int]:
    """Source frame of every output frame (utils/video.py:243-251): pad by repeating the last frame, or sub-sample with
    torch.linspace(0, t - 1, expected).long()."""
    if t <= 0 or expected <= 0:
        raise ValueError(f"need at least one frame (got {t} -> {expected})")
    if t < expected:
        return list(range(t)) + [t - 1] * (expected - t)
    if t > expected:
        return torch.linspace(0, t - 1, expected).long().tolist()
    return list(range(t))


def _vec3(v, c: int, device) -> torch.Tensor:
    vals = [float(v)] * 3 if isinstance(v, (int, float)) else [float(x) for x in v][:c]
    vals = (vals + [vals[-1]] * 3)[:3]
    return torch.tensor(vals, dtype=torch.float32, device=device)


def preprocess_frames(frames, n_frames: int = 32, resize: int | None = 224, normalize: bool = True, mean=0.0, std=1.0,
                      backbone: str = "default", device="cuda", out_uint8: bool = False) -> torch.Tensor:
    """frames: uint8 [F, H, W], [F, H, W, 1] or [F, H, W, 3] (numpy or torch, host or device) -> device tensor
    [expected_frames, resize, resize, 3]: fp32, normalised like the reference, or uint8 (`out_uint8`, rounded)."""
    x = torch.from_numpy(np.ascontiguousarray(frames)) if isinstance(frames, np.ndarray) else frames
    if x.dtype != torch.uint8:
        raise TypeError(f"preprocess_frames takes decoded uint8 frames, got {x.dtype} (float clips go to VideoEncoder as they are)")
    if x.dim() == 3:
        x = x.unsqueeze(-1)
    if x.dim() != 4 or x.shape[-1] not in (1, 3):
        raise ValueError(f"Invalid video shape after loading: Expected 1 or 3 channels, got {tuple(x.shape)}")
    expected = 16 if backbone.lower() == "mvit" else n_frames   # utils/video.py:233-241 (x3d side sizes: not on this path)
    x = x.to(device, non_blocking=True).contiguous()
    T, H, W, C = x.shape
    size = (H, W) if resize is None else (int(resize), int(resize))
    idx = torch.tensor(frame_indices(T, expected), dtype=torch.int32).to(x.device, non_blocking=True)
    mv = sv = None
    if normalize and not out_uint8:
        if mean is None or std is None:
            raise ValueError("Mean and std must be provided for normalization.")
        mv, sv = _vec3(mean, 3, x.device), _vec3(std, 3, x.device)
    return K.resize_frames_u8(x, size, idx, mean=mv, std=sv, antialias=True, out_uint8=out_uint8)


def load_video(video_path: str, n_frames: int = 32, resize: int = 224, normalize: bool = True, mean=0.0, std=1.0,
               video_transforms=None, rand_augment: bool = False, backbone: str = "default", stride: int = 1, device="cuda",
               out_uint8: bool = False, **_timeouts) -> torch.Tensor:
    """utils/video.py:142-300 for pre-decoded `.npy` clips of uint8 frames; returns a DEVICE tensor [F, H, W, 3]."""
    if video_transforms is not None or rand_augment:
        raise NotImplementedError("random augmentations are host-side torchvision code in the reference; not on the HIP path")
    if not str(video_path).endswith(".npy"):
        raise NotImplementedError("only pre-decoded .npy clips: container decoding (cv2) is outside the accelerated path")
    video = np.load(video_path, mmap_mode="r")
    if not isinstance(video, np.ndarray) or video.ndim < 3 or video.ndim > 4:
        raise ValueError(f"Invalid .npy file: Expected 3D or 4D array: {video_path}")
    return preprocess_frames(np.ascontiguousarray(video), n_frames, resize, normalize, mean, std, backbone, device, out_uint8)
