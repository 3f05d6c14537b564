"""VideoEncoder with the reference's constructor, attributes and forward contract (models/video_encoder.py:17-637),
computed by HIP kernels: MViT-v2-S trunk (mvit.py) -> proj (Dropout-Linear-GELU-Dropout on every token) ->
token pooling (mean | AttentionPool per video) -> EnhancedVideoAggregator (fp32).

Differences that are deliberate and stated: only the `mvit` / `mvit_rope` backbones are implemented (r3d/x3d need
torchvision / torch.hub and are outside the hot path); `pretrained=True` cannot download weights
(no network) — load a converted checkpoint with `load_state_dict` / `encoder_path` instead; the spatial size can be
set with `spatial_size` (the reference hard-wires 16x224x224 through torchvision's defaults).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import functional as Fn
from .attention_pool import AttentionPool, AttentionPoolWithCLS
from .mvit import HipMViT
from .registry import ModelRegistry
from .video_aggregator import EnhancedVideoAggregator


@ModelRegistry.register("video_encoder")
class VideoEncoder(nn.Module):
    def __init__(
        self,
        backbone: str = "mvit",
        input_channels: int = 3,
        num_frames: int = 16,
        pretrained: bool = True,
        output_dim: int = 512,
        dropout: float = 0.2,
        num_heads: int = 4,
        freeze_ratio: float = 0.8,
        aggregator_depth: int = 2,
        aggregate_videos_tokens: bool = True,
        per_video_pool: bool = False,
        token_pooling_mode: str = "mean",
        attention_pool_heads: int = 8,
        attention_pool_dropout: float = 0.1,
        use_rope: bool = False,
        rope_base: float = 10000.0,
        rope_temporal_scale: float = 1.0,
        rope_normalize_mode: str = "separate",
        use_cls_token: bool = False,
        multi_video_cls_aggregation: str = "mean",
        encoder_path: Optional[str] = None,
        spatial_size: tuple[int, int] = (224, 224),
        compute_dtype: torch.dtype = torch.bfloat16,
        stochastic_depth_prob: float = 0.2,
        gemm_dtype: str = "bf16",
    ):
        super().__init__()
        self.backbone = backbone
        self.input_channels = input_channels
        self.num_frames = num_frames
        self.output_dim = output_dim
        self.dropout = dropout
        self.freeze_ratio = freeze_ratio
        self._embedding_dim = output_dim
        self.use_rope = use_rope
        self.rope_base, self.rope_temporal_scale, self.rope_normalize_mode = rope_base, rope_temporal_scale, rope_normalize_mode
        self.use_cls_token = use_cls_token or token_pooling_mode == "cls_token"
        self.multi_video_cls_aggregation = multi_video_cls_aggregation
        self.encoder_path = encoder_path
        self.compute_dtype = compute_dtype
        if gemm_dtype not in ("bf16", "fp8"):
            raise ValueError(f"gemm_dtype must be 'bf16' or 'fp8', got {gemm_dtype!r}")
        if gemm_dtype == "fp8" and compute_dtype != torch.bfloat16:
            raise ValueError("gemm_dtype='fp8' runs on bf16 activations: use compute_dtype=torch.bfloat16")
        self.input_norm = None  # (mean[3], std[3]) device tensors: set_input_normalization(); lets forward() take uint8 frames
        self.gemm_dtype = gemm_dtype  # "fp8": forward / data-gradient GEMMs of the trunk on the fp8 MFMA (BASELINE configs[4])

        if backbone in ("mvit", "mvit_rope"):
            if input_channels != 3:
                raise ValueError("the MViT patch embedding takes 3 input channels")
            # the reference ignores num_frames for mvit (torchvision default 16); here it sizes the trunk
            # stochastic_depth_prob 0.2 = torchvision's mvit_v2_s default, which the reference trains with
            self.model: nn.Module = HipMViT(temporal_size=num_frames, spatial_size=spatial_size,
                                            stochastic_depth_prob=stochastic_depth_prob)
            if backbone == "mvit_rope":
                self.use_rope = True
            in_features = 768
            # models/video_encoder.py:256-257: attention blocks are only patched for backbone == "mvit"
            if backbone == "mvit" and self.use_rope:
                self.model.rope_cfg = (self.rope_base, self.rope_temporal_scale)
            if self.use_cls_token:  # allocated but unused, as in the reference (:147-153)
                self.cls_token = nn.Parameter(torch.randn(1, 1, 768) * 0.02)
            else:
                self.cls_token = None
        elif backbone in ("r3d", "x3d_s", "x3d_m"):
            raise NotImplementedError(f"backbone '{backbone}' is outside the MI355X hot path (needs torchvision / torch.hub)")
        else:
            raise ValueError(f"Unsupported backbone: {backbone}")
        # backbone="mvit_rope" (models/video_encoder.py:84-90): the reference builds the same trunk, sets use_rope, but patches the
        # attention blocks only for backbone == "mvit" (:256-257) and routes the forward through torchvision's MViT.forward
        # (:544, :552-561): norm -> x[:, 0] -> Identity head, i.e. ONE token per clip (L = 1) and no RoPE in the blocks.
        self._pooled_cls_fallback = backbone == "mvit_rope"

        self.proj = nn.Sequential(nn.Dropout(self.dropout), nn.Linear(in_features, output_dim), nn.GELU(), nn.Dropout(self.dropout))
        if output_dim % num_heads != 0:
            raise ValueError(f"Output dimension ({output_dim}) must be divisible by number of heads ({num_heads})")
        self.aggregator = EnhancedVideoAggregator(embedding_dim=output_dim, num_heads=num_heads, dropout=self.dropout,
                                                  use_positional_encoding=True, aggregator_depth=aggregator_depth)
        if self.encoder_path:
            self._load_encoder_checkpoint()
        self._freeze_partial_layers()
        self._apply_aggregator = aggregate_videos_tokens
        self._per_video_pool = per_video_pool
        self.token_pooling_mode = token_pooling_mode
        self.attention_pool = None
        if token_pooling_mode == "attention":
            self.attention_pool = AttentionPool(embed_dim=output_dim, num_heads=attention_pool_heads, dropout=attention_pool_dropout)
        elif token_pooling_mode == "cls_token":
            self.attention_pool = AttentionPoolWithCLS(embed_dim=output_dim, num_heads=attention_pool_heads, dropout=attention_pool_dropout)

    # ------------------------------------------------------------------ checkpoint / freezing (host logic)
    def _load_encoder_checkpoint(self):
        if not self.encoder_path or not os.path.exists(self.encoder_path):
            print(f"[VideoEncoder] encoder checkpoint not found: {self.encoder_path}")
            return
        checkpoint = torch.load(self.encoder_path, map_location="cpu")
        state_dict = checkpoint.get("model_state_dict", checkpoint.get("state_dict", checkpoint))
        enc = {}
        for k, v in state_dict.items():
            if not any(skip in k for skip in ["decoder", "mvm", "mask_token", "head"]):
                enc[k.replace("module.", "").replace("model.", "")] = v
        missing, unexpected = self.model.load_state_dict(enc, strict=False)
        print(f"[VideoEncoder] loaded {len(enc)} tensors (missing {len(missing)}, unexpected {len(unexpected)})")

    def _freeze_partial_layers(self):
        """`freeze_ratio` is the TRAINABLE fraction counted over named_parameters from the top (reference :442-454)."""
        params = list(self.model.named_parameters())
        train_count = int(self.freeze_ratio * len(params))
        for i, (_, p) in enumerate(params):
            if i < (len(params) - train_count):
                p.requires_grad = False

    def update_freeze_ratio(self, new_freeze_ratio: float):
        if abs(self.freeze_ratio - new_freeze_ratio) < 1e-6:
            return
        old = self.freeze_ratio
        self.freeze_ratio = new_freeze_ratio
        for p in self.model.parameters():
            p.requires_grad = True
        self._freeze_partial_layers()
        trainable = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        total = sum(p.numel() for p in self.model.parameters())
        print(f"[VideoEncoder] Updated freeze_ratio from {old:.2f} to {new_freeze_ratio:.2f}")
        print(f"[VideoEncoder] Trainable parameters: {trainable:,} / {total:,} ({100.0 * trainable / max(total, 1):.1f}%)")

    def set_input_normalization(self, mean, std) -> None:
        """Per-channel mean / std in pixel units (the dataset's, utils/video.py:278-288).  With them set, `forward` accepts
        raw uint8 frames: the normalisation happens inside the patch-embedding gather (dcc_im2col3d_u8), the clip crosses
        PCIe as bytes and is never materialised in fp32."""
        dev = self.proj[1].weight.device
        as_vec = lambda v: torch.tensor([float(v)] * 3 if isinstance(v, (int, float)) else [float(x) for x in v][:3],  # noqa: E731
                                        dtype=torch.float32, device=dev)
        self.input_norm = (as_vec(mean), as_vec(std))

    @property
    def embedding_dim(self) -> int:
        return self._embedding_dim

    # ------------------------------------------------------------------------------------------ forward
    def get_tokens(self, x: torch.Tensor, mode: str = "patch", return_dict: bool = False):
        prev_apply, prev_pv = self._apply_aggregator, self._per_video_pool
        self._apply_aggregator = False
        self._per_video_pool = mode == "video"
        features = self._compute_feature_dict(x, compute_aggregated=True)
        self._apply_aggregator, self._per_video_pool = prev_apply, prev_pv
        if return_dict:
            return {k: features[k] for k in ("patch_tokens", "per_video_tokens", "token_grid", "study_features")}
        if mode == "video":
            pv = features["per_video_tokens"]
            return pv.squeeze(1) if pv.dim() == 3 and pv.shape[1] == 1 else pv
        return features["patch_tokens"]

    def _extract_backbone_features(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B*N, T, H, W, C] fp32 channels-last -> [B*N, L, 768] (class token included, as in the reference)."""
        norm = None
        if x.dtype == torch.uint8:
            if self.input_norm is None:
                raise ValueError("uint8 frames need VideoEncoder.set_input_normalization(mean, std) (pixel units)")
            norm = tuple(t.to(x.device) for t in self.input_norm)
        with Fn.fp8_gemms(self.gemm_dtype == "fp8"):
            tok = self.model.forward_features(x, self.compute_dtype, norm)
        return Fn.class_token(tok) if self._pooled_cls_fallback else tok

    def _compute_feature_dict(self, x: torch.Tensor, compute_aggregated: bool) -> Dict[str, torch.Tensor]:
        if x.ndim == 5:
            x = x.unsqueeze(1)
        if x.ndim == 7:
            s = x.shape
            x = x.view(s[0], s[1] * s[2], s[3], s[4], s[5], s[6])
        B, N, T, H, W, C = x.shape
        if x.dtype not in (torch.float32, torch.uint8):
            x = Fn.cast(x.contiguous(), torch.float32)
        clips = x.reshape(B * N, T, H, W, C)  # the reference permutes to NCTHW as a VIEW; memory stays channels-last
        tok = self._extract_backbone_features(clips)
        L = tok.shape[1]
        p = self.dropout
        tok = Fn.dropout(tok, p, self.training)
        tok = Fn.linear(tok, self.proj[1].weight, self.proj[1].bias, gelu=True)
        tok = Fn.dropout(tok, p, self.training)
        D = tok.shape[-1]
        token_feats = tok.view(B, N, L, D)
        per_video = self._pool_video_tokens(token_feats)
        patch_tokens = token_feats.reshape(B, N * L, D)
        study = None
        if compute_aggregated and self.aggregator is not None:
            study = self._aggregate_video_features(per_video)
        return {"token_grid": token_feats, "per_video_tokens": per_video, "patch_tokens": patch_tokens,
                "study_features": study}

    def _pool_video_tokens(self, token_feats: torch.Tensor) -> torch.Tensor:
        B, N, L, D = token_feats.shape
        flat = token_feats.reshape(B * N, L, D)
        if self.attention_pool is not None:
            return self.attention_pool(flat).view(B, N, D)  # one batched launch instead of the reference's loop over N
        return Fn.token_mean(flat).view(B, N, D)

    def _aggregate_video_features(self, per_video: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.aggregator(Fn.cast(per_video, torch.float32), mask=mask)

    def forward(self, x: torch.Tensor, return_tokens: bool = False, token_mode: str = "patch"):
        features = self._compute_feature_dict(x, compute_aggregated=self._apply_aggregator)
        if self._apply_aggregator and features["study_features"] is not None:
            primary = features["study_features"]
        elif self._per_video_pool:
            primary = features["per_video_tokens"]
            if primary.dim() == 3 and primary.shape[1] == 1:
                primary = primary.squeeze(1)
        else:
            primary = features["patch_tokens"]
        if not return_tokens:
            return primary
        tokens = features["per_video_tokens"] if token_mode == "video" else (
            features["token_grid"] if token_mode == "grid" else features["patch_tokens"])
        return primary, tokens
