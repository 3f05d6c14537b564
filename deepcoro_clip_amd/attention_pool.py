"""AttentionPool on HIP kernels (reference: models/attention_pool.py:10-101): a learnable query attends over the
tokens of one video through nn.MultiheadAttention(batch_first) semantics, then LayerNorm (+ optional Linear)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn


def key_bias(mask: torch.Tensor | None, B: int, N: int, device, prefix: int = 0) -> torch.Tensor | None:
    """nn.MultiheadAttention's key_padding_mask (bool [B, N], True = ignored) as the flash kernels' additive fp32 key bias
    [B, prefix + N]: 0 on valid keys, -1e30 on masked ones (`prefix` unmasked columns in front: a CLS token).  A row whose
    keys are ALL masked yields NaN in the reference (softmax over -inf only) and a uniform average here; callers that can
    meet such rows guard them themselves (the MIL head and the aggregator's pooling do)."""
    if mask is None:
        return None
    if tuple(mask.shape) != (B, N):
        raise ValueError(f"mask shape {tuple(mask.shape)} does not match tokens {(B, N)}")
    kb = torch.zeros((B, prefix + N), device=device, dtype=torch.float32)
    kb[:, prefix:] = mask.to(device=device, dtype=torch.bool).to(torch.float32) * -1e30
    return kb


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
        """x: [B, N, D] (compute dtype) -> [B, output_dim]; mask: bool [B, N], True = ignored (reference :61-101)."""
        B, N, D = x.shape
        assert D == self.embed_dim, f"Input dim {D} != expected {self.embed_dim}"
        q = Fn.cast(self.query, x.dtype).expand(B, 1, D)
        a = self.attn
        o = Fn.multihead_attention(q, x, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                   self.num_heads, self_attn=False, dropout_p=self.attn_dropout, training=self.training,
                                   kbias=key_bias(mask, B, N, x.device))
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
        B, N, D = x.shape
        assert D == self.embed_dim, f"Input dim {D} != expected {self.embed_dim}"
        kb = key_bias(mask, B, N, x.device, prefix=1)  # the CLS column is never masked (reference :180-184)
        x = Fn.prepend_token(self.cls_token, x)
        tr = self.training
        for l in self.transformer.layers:
            a = l.self_attn
            sa = Fn.multihead_attention(x, x, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                        self.num_heads, self_attn=True, dropout_p=self.p, training=tr, kbias=kb)
            x = Fn.layernorm(Fn.add(x, Fn.dropout(sa, self.p, tr)), l.norm1.weight, l.norm1.bias, l.norm1.eps)
            h = Fn.dropout(Fn.relu(Fn.linear(x, l.linear1.weight, l.linear1.bias)), self.p, tr)
            y = Fn.linear(h, l.linear2.weight, l.linear2.bias)
            x = Fn.layernorm(Fn.add(x, Fn.dropout(y, self.p, tr)), l.norm2.weight, l.norm2.bias, l.norm2.eps)
        cls = x.view(B, (N + 1) * D)[:, :D]  # strided rows; LayerNorm takes a row pitch
        o = Fn.layernorm(cls, self.norm.weight, self.norm.bias, self.norm.eps)
        if isinstance(self.proj, nn.Linear):
            o = Fn.linear(o, self.proj.weight, self.proj.bias)
        return o.reshape(B, -1)
