"""Token-level consumers of the accelerated encoder on the HIP path:

  * `MultiInstanceLinearProbing` — models/multi_instance_linear_probing.py:81-544: per-instance embeddings [B, N, D] (or
    patch tokens [B, N, L, D]) -> masked mean / max / gated-attention / cls_token pooling (+ the two hybrids) -> one linear
    head per task.  Same constructor, parameter names (state_dict compatible) and forward contract.
  * `VideoMILWrapper` — projects/linear_probing_project.py:29-114: VideoEncoder output -> [B, N, D] / [B, N, L, D] -> MIL.
  * `patch_contrastive_features` — the patch pooling of utils/loss/multitask_loss.py:279-285 (study-level mean of the patch
    tokens fed to the same contrastive loss).

The head runs in fp32 (its tensors are tiny next to the trunk's; the reference runs it outside autocast-sensitive code too):
linears through the fp32 MFMA GEMM, pooling through csrc/mil.hip, attention through the flash kernel with the instance mask as
an additive key bias.  View-angle embeddings (`num_view_classes > 0`, reference :142-145,265-269): `dcc_view_embed_add`.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import functional as Fn
from .registry import ModelRegistry

_MODES = {"mean", "max", "attention", "cls_token", "mean+cls_token", "attention+cls_token"}


def _mask_u8(mask: torch.Tensor | None, B: int, N: int, device) -> torch.Tensor | None:
    if mask is None:
        return None
    if tuple(mask.shape) != (B, N):
        raise ValueError(f"Mask shape {tuple(mask.shape)} does not match input shape {(B, N)}")
    return (mask != 0).to(device=device, dtype=torch.uint8).contiguous()


@ModelRegistry.register("multi_instance_linear_probing")
class MultiInstanceLinearProbing(nn.Module):
    def __init__(self, embedding_dim: int, head_structure: Dict[str, int], pooling_mode: str = "mean", attention_hidden: int = 128,
                 dropout: float = 0.0, use_cls_token: bool = False, num_attention_heads: int = 8,
                 separate_video_attention: bool = True, normalization_strategy: str = "post_norm", num_view_classes: int = 0) -> None:
        super().__init__()
        if pooling_mode not in _MODES:
            raise ValueError(f"pooling_mode must be one of {_MODES}")
        if not head_structure:
            raise ValueError("head_structure cannot be empty")
        if normalization_strategy not in {"pre_norm", "post_norm"}:
            raise ValueError("normalization_strategy must be 'pre_norm' or 'post_norm'")
        self.embedding_dim, self.pooling_mode, self.head_structure = embedding_dim, pooling_mode, head_structure
        self.use_cls_token = use_cls_token or "cls_token" in pooling_mode
        self.num_attention_heads, self.separate_video_attention = num_attention_heads, separate_video_attention
        self.normalization_strategy, self.num_view_classes, self.p = normalization_strategy, num_view_classes, dropout
        if num_view_classes > 0:  # EchoJEPA-style angle embeddings; +1 = the PAD class of missing / padded views (reference :142-145)
            self.view_embedding = nn.Embedding(num_view_classes + 1, embedding_dim)
            self.view_pad_id = num_view_classes
        if self.use_cls_token:
            self.cls_token = nn.Parameter(torch.randn(1, 1, embedding_dim))
            mk = lambda: nn.MultiheadAttention(embed_dim=embedding_dim, num_heads=num_attention_heads, dropout=dropout, batch_first=True)  # noqa: E731
            if separate_video_attention:
                self.cls_attention_within, self.cls_attention_across = mk(), mk()
                self.cls_norm_within, self.cls_norm_across = nn.LayerNorm(embedding_dim), nn.LayerNorm(embedding_dim)
            else:
                self.cls_attention = mk()
                self.cls_norm = nn.LayerNorm(embedding_dim)
            self.cls_dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        if "attention" in pooling_mode:
            self.attention_V = nn.Linear(embedding_dim, attention_hidden)
            self.attention_U = nn.Linear(embedding_dim, attention_hidden)
            self.attention_w = nn.Linear(attention_hidden, 1)
            self.attn_dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        head_in = 2 * embedding_dim if "+" in pooling_mode else embedding_dim
        self.heads = nn.ModuleDict({name: nn.Linear(head_in, n) for name, n in head_structure.items()})
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, mask: torch.Tensor | None = None, view_ids: torch.Tensor | None = None) -> Dict[str, torch.Tensor]:
        if x.ndim not in (3, 4):
            raise ValueError(f"Unsupported input shape {x.shape}; expected 3-D or 4-D tensor.")
        if x.shape[-1] != self.embedding_dim:
            raise ValueError(f"Expected embedding_dim={self.embedding_dim} but got {x.shape[-1]}")
        x = Fn.cast(x.contiguous(), torch.float32)
        if x.ndim == 4 and (self.pooling_mode in {"mean", "max"} or ("mean" in self.pooling_mode and "+" in self.pooling_mode)):
            B, N, L, D = x.shape  # patch tokens with a patch-agnostic pooling rule: average the patches first (reference :246-262)
            x = Fn.masked_mean(x.view(B * N, L, D)).view(B, N, D)
        if view_ids is not None and self.num_view_classes > 0:  # reference :265-269 (after the patch averaging, broadcast over L)
            B, N = x.shape[0], x.shape[1]
            if tuple(view_ids.shape) != (B, N):
                raise ValueError(f"view_ids shape {tuple(view_ids.shape)} does not match {(B, N)}")
            ids = view_ids.to(device=x.device, dtype=torch.int64).contiguous()
            lo, hi = int(ids.min()), int(ids.max())   # nn.Embedding's index check (the kernel trusts the ids)
            if lo < 0 or hi > self.num_view_classes:
                raise IndexError(f"view id out of range [0, {self.num_view_classes}]: min {lo}, max {hi}")
            Lt = x.shape[2] if x.ndim == 4 else 1
            x = Fn.view_embed_add(x.reshape(B * N, Lt, x.shape[-1]), ids.view(-1), self.view_embedding.weight).view(x.shape)
        pooled = self._pool_instances(x, mask)
        return {name: Fn.linear(pooled, head.weight, head.bias) for name, head in self.heads.items()}

    def _pool_instances(self, x, mask):
        B, N = x.shape[0], x.shape[1]
        host_mask = None if mask is None else (mask != 0).cpu()
        if host_mask is not None and not bool(host_mask.any()):  # no valid instance anywhere (reference :302-308)
            return torch.zeros(B, (2 if "+" in self.pooling_mode else 1) * x.shape[-1], dtype=x.dtype, device=x.device)
        m8 = _mask_u8(mask, B, N, x.device)
        if "+" in self.pooling_mode:
            first, _ = self.pooling_mode.split("+")
            f1 = self._mean(x, m8) if first == "mean" else self._attention(x, m8)
            f2 = self._cls(x, m8)
            return _ConcatFn.apply(f1, f2)
        if self.pooling_mode == "cls_token":
            return self._cls(x, m8)
        if self.pooling_mode == "mean":
            return self._mean(x, m8)
        if self.pooling_mode == "max":
            return Fn.masked_max(x, m8)
        return self._attention(x, m8)

    @staticmethod
    def _mean(x, m8):
        return Fn.masked_mean(x, m8)

    def _scores(self, x3):
        B, N, D = x3.shape
        v = Fn.linear(x3, self.attention_V.weight, self.attention_V.bias)
        u = Fn.linear(x3, self.attention_U.weight, self.attention_U.bias)
        return Fn.linear(Fn.gate(v, u), self.attention_w.weight, self.attention_w.bias).view(B, N)

    def _attn_pool(self, x3, m8):
        w = Fn.masked_softmax(self._scores(x3), m8)
        w = Fn.dropout(w, self.p, self.training)
        return Fn.weighted_sum(w, x3)

    def _attention(self, x, m8):
        if x.ndim == 4:  # patch-level attention inside every video, then attention across the videos (reference :513-538)
            B, N, L, D = x.shape
            video = self._attn_pool(x.view(B * N, L, D), None).view(B, N, D)
            return self._attn_pool(video, m8)
        return self._attn_pool(x, m8)

    def _cls_layer(self, seq, attn, norm, kbias):
        if self.normalization_strategy == "pre_norm":
            seq = Fn.layernorm(seq, norm.weight, norm.bias, norm.eps)
        out = Fn.multihead_attention(seq, seq, attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias,
                                     self.num_attention_heads, self_attn=True, dropout_p=self.p, training=self.training, kbias=kbias)
        if self.normalization_strategy == "post_norm":
            out = Fn.layernorm(out, norm.weight, norm.bias, norm.eps)
        return Fn.dropout(_FirstTokenFn.apply(out), self.p, self.training)

    def _cls(self, x, m8):
        sep = self.separate_video_attention
        if x.ndim == 4:
            B, N, L, D = x.shape
            seq = Fn.prepend_token(self.cls_token, x.view(B * N, L, D))
            x = self._cls_layer(seq, self.cls_attention_within if sep else self.cls_attention,
                                self.cls_norm_within if sep else self.cls_norm, None).view(B, N, D)
        B, N, D = x.shape
        kbias = None
        if m8 is not None:  # key_padding_mask = ~[cls | mask] as an additive bias
            valid = torch.cat([torch.ones(B, 1, dtype=torch.uint8, device=x.device), m8], dim=1)
            kbias = ((1.0 - valid.float()) * -1e30).contiguous()
        seq = Fn.prepend_token(self.cls_token, x)
        return self._cls_layer(seq, self.cls_attention_across if sep else self.cls_attention,
                               self.cls_norm_across if sep else self.cls_norm, kbias)


class _FirstTokenFn(torch.autograd.Function):
    """seq[:, 0, :] of a contiguous [B, S, D] tensor with HIP copy kernels in both directions."""

    @staticmethod
    def forward(ctx, seq):
        B, S, D = seq.shape
        out = torch.empty((B, D), device=seq.device, dtype=seq.dtype)
        Fn.K.cast2d(Fn._c(seq).view(B, S * D)[:, :D], out)
        ctx.shape = (B, S, D)
        return out

    @staticmethod
    def backward(ctx, d):
        B, S, D = ctx.shape
        dseq = Fn.zeros_f32((B, S * D), d.device) if d.dtype == torch.float32 else torch.zeros((B, S * D), device=d.device, dtype=d.dtype)
        Fn.K.cast2d(Fn._c(d), dseq[:, :D])
        return dseq.view(B, S, D)


class _ConcatFn(torch.autograd.Function):
    """cat([a, b], dim=-1) of two [B, D] tensors (hybrid pooling)."""

    @staticmethod
    def forward(ctx, a, b):
        B, D = a.shape
        out = torch.empty((B, 2 * D), device=a.device, dtype=a.dtype)
        Fn.K.cast2d(Fn._c(a), out[:, :D])
        Fn.K.cast2d(Fn._c(b), out[:, D:])
        ctx.D = D
        return out

    @staticmethod
    def backward(ctx, d):
        D = ctx.D
        d = Fn._c(d)
        da, db = torch.empty((d.shape[0], D), device=d.device, dtype=d.dtype), torch.empty((d.shape[0], D), device=d.device, dtype=d.dtype)
        Fn.K.cast2d(d[:, :D], da)
        Fn.K.cast2d(d[:, D:], db)
        return da, db


class VideoMILWrapper(nn.Module):
    """projects/linear_probing_project.py:29-114."""

    def __init__(self, video_encoder, mil_model, num_videos: int):
        super().__init__()
        self.video_encoder, self.mil_model, self.num_videos = video_encoder, mil_model, num_videos

    def forward(self, x: torch.Tensor, video_indices: Optional[torch.Tensor] = None, view_ids: Optional[torch.Tensor] = None):
        emb = self.video_encoder(x)
        if emb.ndim == 2:
            emb = emb.unsqueeze(1)
        elif emb.ndim == 3 and emb.shape[1] > self.num_videos:
            B, NL, D = emb.shape
            if NL % self.num_videos != 0:
                raise ValueError(f"Number of tokens (NL={NL}) is not divisible by the expected num_videos={self.num_videos}. "
                                 "Cannot infer tokens per video for hierarchical pooling.")
            emb = emb.view(B, self.num_videos, NL // self.num_videos, D)
        B, N = emb.shape[0], emb.shape[1]
        mask = torch.ones((B, N), dtype=torch.bool, device=emb.device)
        return self.mil_model(emb, mask=mask, view_ids=view_ids)


def patch_contrastive_features(patch_features: torch.Tensor) -> torch.Tensor:
    """Study-level pooling of the patch tokens for the patch-contrastive term (utils/loss/multitask_loss.py:279-285:
    `patch_features.mean(dim=1)`, then the SAME contrastive loss as the study embedding, weighted by
    patch_contrastive_weight): [B, tokens, D] -> [B, D]."""
    return Fn.token_mean(patch_features.contiguous())
