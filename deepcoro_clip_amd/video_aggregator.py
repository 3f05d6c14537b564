"""EnhancedVideoAggregator on HIP kernels (reference: models/video_aggregator.py:7-159).

Runs in fp32 like the reference (autocast disabled + .float(), models/video_encoder.py:605-613): positional
encoding add, `depth` pre-LN blocks (MHA + MLP), final LayerNorm, learned-query softmax pooling over N."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from .attention_pool import key_bias


class TransformerBlock(nn.Module):
    def __init__(self, embedding_dim, num_heads, dropout):
        super().__init__()
        self.num_heads = num_heads
        self.p = dropout
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.attn = nn.MultiheadAttention(embed_dim=embedding_dim, num_heads=num_heads, dropout=dropout, batch_first=True)
        self.dropout1 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.mlp = nn.Sequential(nn.Linear(embedding_dim, embedding_dim * 4), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(embedding_dim * 4, embedding_dim))
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor, key_padding_mask=None) -> torch.Tensor:
        """key_padding_mask: bool [B, N], True = padded video (reference :33-54), or the additive fp32 key bias built from it."""
        a = self.attn
        kb = key_padding_mask
        if kb is not None and kb.dtype != torch.float32:
            kb = key_bias(kb, x.shape[0], x.shape[1], x.device)
        h = Fn.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        att = Fn.multihead_attention(h, h, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                     self.num_heads, self_attn=True, dropout_p=self.p, training=self.training, kbias=kb)
        x = Fn.add(x, Fn.dropout(att, self.p, self.training))
        h = Fn.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        if self.training and self.p > 0:
            h1 = Fn.dropout(Fn.linear(h, self.mlp[0].weight, self.mlp[0].bias, gelu=True), self.p, True)
            m = Fn.dropout(Fn.linear(h1, self.mlp[3].weight, self.mlp[3].bias), self.p, True)
            return Fn.add(x, m)
        return Fn.mlp(h, self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight, self.mlp[3].bias, residual=x)


class EnhancedVideoAggregator(nn.Module):
    def __init__(self, embedding_dim: int, num_heads: int = 4, dropout: float = 0.1, use_positional_encoding: bool = True,
                 aggregator_depth: int = 2, max_segments: int = 1024):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.use_positional_encoding = use_positional_encoding
        self.aggregator_depth = aggregator_depth
        if use_positional_encoding:
            self.pos_encoding = nn.Parameter(torch.zeros(1, max_segments, embedding_dim))
            nn.init.trunc_normal_(self.pos_encoding, std=0.02)
        else:
            self.pos_encoding = None
        self.blocks = nn.ModuleList([TransformerBlock(embedding_dim, num_heads, dropout) for _ in range(aggregator_depth)])
        self.final_ln = nn.LayerNorm(embedding_dim)
        self.attn_query = nn.Parameter(torch.randn(1, 1, embedding_dim))
        nn.init.normal_(self.attn_query, std=0.02)

    def forward(self, x: torch.Tensor, mask=None) -> torch.Tensor:
        """x: [B, N, D] fp32 -> [B, D] fp32.  mask: bool [B, N], True = padded video (reference :105-159): masked keys in
        every block's self-attention, then a softmax over the VALID videos only; a study without any valid video pools to
        zeros (the reference's fallback weights valid / sum(valid) are all zero there)."""
        B, N, D = x.shape
        x = Fn.cast(x, torch.float32)
        if self.pos_encoding is not None:
            x = Fn.add(x, self.pos_encoding[:, :N, :].expand(B, N, D))
        kb = key_bias(mask, B, N, x.device)
        for blk in self.blocks:
            x = blk(x, kb)
        x = Fn.layernorm(x, self.final_ln.weight, self.final_ln.bias, self.final_ln.eps)
        if mask is None:
            return Fn.qpool(x, self.attn_query)
        # masked rows get weight 0 (forward and backward), so zero-filling them first (reference :131-132) changes nothing
        valid = (~mask.to(device=x.device, dtype=torch.bool)).to(torch.uint8).contiguous()
        scores = Fn.linear(x, self.attn_query.view(1, D)).view(B, N)
        return Fn.weighted_sum(Fn.masked_softmax(scores, valid), x)
