"""ORACLE (test infrastructure, CPU only) — restatement of the frame pipeline of utils/video.py:load_video (:206-295) from
decoded frames onward, in plain torch: float32 conversion, grayscale -> 3 channels, frame count handling
(last-frame padding / torch.linspace sub-sampling), `v2.Resize((s, s), antialias=True)` — which for float tensors is
`F.interpolate(mode="bilinear", antialias=True, align_corners=False)` (torchvision 0.25 `_geometry.resize_image`; torchvision is
absent from this image, the interpolate call it makes is not) — and `v2.Normalize`.  Parity note: the resize itself is ATen's,
i.e. the very operator the reference executes, so this part of the oracle is pinned by construction.

Only tests/ may import this module."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def load_video_frames(video: np.ndarray, n_frames=32, resize=224, normalize=True, mean=0.0, std=1.0, backbone="default") -> np.ndarray:
    video = video.astype(np.float32)
    if video.ndim == 3:
        video = np.repeat(np.expand_dims(video, axis=-1), 3, axis=-1)
    elif video.ndim == 4 and video.shape[-1] == 1:
        video = np.repeat(video, 3, axis=-1)
    elif video.ndim != 4 or video.shape[-1] != 3:
        raise ValueError(f"Invalid video shape after loading: {video.shape}")
    v = torch.from_numpy(video).permute(0, 3, 1, 2)
    expected = 16 if backbone.lower() == "mvit" else n_frames
    t = v.shape[0]
    if t < expected:
        v = torch.cat([v, v[-1:].repeat(expected - t, 1, 1, 1)], dim=0)
    elif t > expected:
        v = v[torch.linspace(0, t - 1, expected).long()]
    if resize is not None:
        v = F.interpolate(v, size=(resize, resize), mode="bilinear", antialias=True, align_corners=False)
    if normalize:
        c = v.shape[1]
        m = [float(mean)] * c if isinstance(mean, (int, float)) else list(mean)[:c]
        s = [float(std)] * c if isinstance(std, (int, float)) else list(std)[:c]
        v = (v - torch.tensor(m).view(1, c, 1, 1)) / torch.tensor(s).view(1, c, 1, 1)
    return v.permute(0, 2, 3, 1).contiguous().numpy()
