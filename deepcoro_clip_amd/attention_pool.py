"""AttentionPool on HIP kernels (reference: models/attention_pool.py:10-101): a learnable query attends over the
tokens of one video through nn.MultiheadAttention(batch_first) semantics, then LayerNorm (+ optional Linear)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn


class AttentionPool(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int = 8, output_dim: int | None = None, dropout: float = 0.0):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.output_dim = output_dim or embed_dim
        self.query = nn.Parameter(torch.randn(1, 1, embed_dim))
        nn.init.trunc_normal_(self.query, std=0.02)
        # parameter container only (in_proj_weight / in_proj_bias / out_proj.*): forward() is never called
        self.attn = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads, dropout=dropout, batch_first=True)
        self.attn_dropout = dropout
        self.norm = nn.LayerNorm(embed_dim)
        self.proj = nn.Linear(embed_dim, self.output_dim) if self.output_dim != embed_dim else nn.Identity()

    def forward(self, x: torch.Tensor, mask: torch.Tensor | None = None) -> torch.Tensor:
        """x: [B, N, D] (compute dtype) -> [B, output_dim]."""
        if mask is not None:
            raise NotImplementedError("key_padding_mask is unused on the contrastive path (models/video_encoder.py:597-602)")
        B, N, D = x.shape
        assert D == self.embed_dim, f"Input dim {D} != expected {self.embed_dim}"
        q = Fn.cast(self.query, x.dtype).expand(B, 1, D)
        a = self.attn
        o = Fn.multihead_attention(q, x, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                   self.num_heads, self_attn=False, dropout_p=self.attn_dropout, training=self.training)
        o = Fn.layernorm(o, self.norm.weight, self.norm.bias, self.norm.eps)
        if isinstance(self.proj, nn.Linear):
            o = Fn.linear(o, self.proj.weight, self.proj.bias)
        return o.reshape(B, -1)


class AttentionPoolWithCLS(nn.Module):
    """models/attention_pool.py:104-197: a learnable CLS token is prepended, `num_layers` post-norm
    nn.TransformerEncoderLayer blocks (self-attention + ReLU feed-forward, dim_feedforward 2048) run over the sequence,
    the CLS output goes through LayerNorm (+ optional Linear).  nn.TransformerEncoder is the parameter container (its
    state_dict names are the reference's); the arithmetic runs on the HIP kernels."""

    def __init__(self, embed_dim: int, num_heads: int = 8, num_layers: int = 1, output_dim: int | None = None, dropout: float = 0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.num_layers = embed_dim, num_heads, num_layers
        self.output_dim = output_dim or embed_dim
        self.p = dropout
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.transformer = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(d_model=embed_dim, nhead=num_heads, dropout=dropout, batch_first=True), num_layers=num_layers)
        self.norm = nn.LayerNorm(embed_dim)
        self.proj = nn.Linear(embed_dim, self.output_dim) if self.output_dim != embed_dim else nn.Identity()

    def forward(self, x: torch.Tensor, mask: torch.Tensor | None = None) -> torch.Tensor:
        if mask is not None:
            raise NotImplementedError("key_padding_mask is unused on the contrastive path (models/video_encoder.py:597-602)")
        B, N, D = x.shape
        assert D == self.embed_dim, f"Input dim {D} != expected {self.embed_dim}"
        x = Fn.prepend_token(self.cls_token, x)
        tr = self.training
        for l in self.transformer.layers:
            a = l.self_attn
            sa = Fn.multihead_attention(x, x, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                        self.num_heads, self_attn=True, dropout_p=self.p, training=tr)
            x = Fn.layernorm(Fn.add(x, Fn.dropout(sa, self.p, tr)), l.norm1.weight, l.norm1.bias, l.norm1.eps)
            h = Fn.dropout(Fn.relu(Fn.linear(x, l.linear1.weight, l.linear1.bias)), self.p, tr)
            y = Fn.linear(h, l.linear2.weight, l.linear2.bias)
            x = Fn.layernorm(Fn.add(x, Fn.dropout(y, self.p, tr)), l.norm2.weight, l.norm2.bias, l.norm2.eps)
        cls = x.view(B, (N + 1) * D)[:, :D]  # strided rows; LayerNorm takes a row pitch
        o = Fn.layernorm(cls, self.norm.weight, self.norm.bias, self.norm.eps)
        if isinstance(self.proj, nn.Linear):
            o = Fn.linear(o, self.proj.weight, self.proj.bias)
        return o.reshape(B, -1)
