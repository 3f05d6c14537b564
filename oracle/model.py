"""ORACLE (test infrastructure, CPU only) — plain-PyTorch restatement of the encoders around the backbones.

Follows (file:line in /root/reference):
  * VideoEncoder._compute_feature_dict / forward          models/video_encoder.py:565-637
  * proj head Dropout-Linear-GELU-Dropout                  models/video_encoder.py:175-180, models/text_encoder.py:63-68
  * AttentionPool.forward (nn.MultiheadAttention, 1 query) models/attention_pool.py:61-101
  * EnhancedVideoAggregator / TransformerBlock             models/video_aggregator.py:33-54,105-159
  * TextEncoder.forward over HF BertModel                  models/text_encoder.py:120-137  (BERT arithmetic: HF
    transformers `BertModel`, pinned 5.2.0 in uv.lock, 5.15.0 installed; restated here, pinned by golden vectors
    generated from `BertModel(BertConfig())` with name-seeded weights — tests/golden/make_golden.py)
Parameter names equal the reference's state_dict keys so weights can be exchanged with the HIP modules.
Dropout is restated as identity (parity runs use p = 0 / eval, like the reference's own tests).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
import zlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from .mvit import MViTv2S, enable_rope


def init_by_name(module: nn.Module, seed: int = 0) -> nn.Module:
    """Deterministic name-seeded initialisation (same bytes for the oracle and the HIP modules)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7FFFFFFF)
            r = torch.randn(p.shape, generator=g)
            lname = name.lower()
            is_norm = p.dim() == 1 and any(t in lname for t in ("norm", "final_ln"))
            if is_norm and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * r)
            elif p.dim() == 1:
                p.copy_(0.02 * r)
            elif "conv_proj" in name or ".pool." in name or "rel_pos" in name:
                p.copy_(0.1 * r)
            else:
                p.copy_(0.02 * r)
    return module


def mha_forward(query, key, in_w, in_b, out_w, out_b, num_heads, key_padding_mask=None):
    """batch_first nn.MultiheadAttention(query, key, key) without dropout; key_padding_mask: bool [B, Lk], True = ignored
    (nn.MultiheadAttention's key_padding_mask: -inf on the masked keys' scores)."""
    B, Lq, E = query.shape
    hd = E // num_heads
    q = F.linear(query, in_w[:E], in_b[:E]).view(B, Lq, num_heads, hd).transpose(1, 2)
    k = F.linear(key, in_w[E:2 * E], in_b[E:2 * E]).view(B, -1, num_heads, hd).transpose(1, 2)
    v = F.linear(key, in_w[2 * E:], in_b[2 * E:]).view(B, -1, num_heads, hd).transpose(1, 2)
    a = (q @ k.transpose(-2, -1)) / math.sqrt(hd)
    if key_padding_mask is not None:
        a = a.masked_fill(key_padding_mask.bool()[:, None, None, :], float("-inf"))
    o = (a.softmax(-1) @ v).transpose(1, 2).reshape(B, Lq, E)
    return F.linear(o, out_w, out_b)


class _MHAParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = nn.Linear(dim, dim)


class AttentionPoolOracle(nn.Module):
    def __init__(self, embed_dim=512, num_heads=8):
        super().__init__()
        self.num_heads = num_heads
        self.query = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.attn = _MHAParams(embed_dim)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x, mask=None):
        """mask: bool [B, N], True = ignored (models/attention_pool.py:61-101)."""
        q = self.query.expand(x.shape[0], -1, -1)
        o = mha_forward(q, x, self.attn.in_proj_weight, self.attn.in_proj_bias, self.attn.out_proj.weight,
                        self.attn.out_proj.bias, self.num_heads, mask)
        return self.norm(o).squeeze(1)


class _EncLayerParams(nn.Module):
    """nn.TransformerEncoderLayer parameter names (post-norm, ReLU feed-forward)."""

    def __init__(self, dim, ffn):
        super().__init__()
        self.self_attn = _MHAParams(dim)
        self.linear1 = nn.Linear(dim, ffn)
        self.linear2 = nn.Linear(ffn, dim)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)


class _EncStack(nn.Module):
    def __init__(self, n, dim, ffn):
        super().__init__()
        self.layers = nn.ModuleList([_EncLayerParams(dim, ffn) for _ in range(n)])


class AttentionPoolWithCLSOracle(nn.Module):
    """models/attention_pool.py:104-197 restated: CLS token + post-norm nn.TransformerEncoderLayer blocks (ReLU FFN 2048)
    + LayerNorm on the CLS output; parameter names of the reference's state_dict."""

    def __init__(self, embed_dim=512, num_heads=8, num_layers=1):
        super().__init__()
        self.num_heads = num_heads
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.transformer = _EncStack(num_layers, embed_dim, 2048)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x, mask=None):
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1)
        if mask is not None:  # the CLS token is never masked (models/attention_pool.py:180-184)
            mask = torch.cat([torch.zeros(x.shape[0], 1, dtype=torch.bool), mask.bool()], dim=1)
        for l in self.transformer.layers:
            a = l.self_attn
            x = l.norm1(x + mha_forward(x, x, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, self.num_heads, mask))
            x = l.norm2(x + l.linear2(F.relu(l.linear1(x))))
        return self.norm(x[:, 0])


class _AggBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _MHAParams(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = nn.Sequential(nn.Linear(dim, 4 * dim), nn.GELU(), nn.Dropout(0.0), nn.Linear(4 * dim, dim))

    def forward(self, x, key_padding_mask=None):
        h = self.norm1(x)
        x = x + mha_forward(h, h, self.attn.in_proj_weight, self.attn.in_proj_bias, self.attn.out_proj.weight,
                            self.attn.out_proj.bias, self.heads, key_padding_mask)
        return x + self.mlp(self.norm2(x))


class AggregatorOracle(nn.Module):
    def __init__(self, dim=512, heads=4, depth=2, max_segments=1024):
        super().__init__()
        self.pos_encoding = nn.Parameter(torch.zeros(1, max_segments, dim))
        self.blocks = nn.ModuleList([_AggBlock(dim, heads) for _ in range(depth)])
        self.final_ln = nn.LayerNorm(dim)
        self.attn_query = nn.Parameter(torch.zeros(1, 1, dim))

    def forward(self, x, mask=None):
        """mask: bool [B, N], True = padded video (models/video_aggregator.py:105-159, the masked variant included)."""
        B, N, D = x.shape
        x = x + self.pos_encoding[:, :N]
        for b in self.blocks:
            x = b(x, mask)
        x = self.final_ln(x)
        s = torch.matmul(self.attn_query.expand(B, -1, -1), x.transpose(1, 2))
        if mask is not None:
            mask = mask.bool()
            x = x.masked_fill(mask.unsqueeze(-1), 0.0)
            w = torch.nan_to_num(s.masked_fill(mask.unsqueeze(1), float("-inf")).softmax(-1), nan=0.0, posinf=0.0, neginf=0.0)
            # rows without a valid video: uniform over the valid ones = all zero (:142-152)
        else:
            w = s.softmax(-1)
        return torch.bmm(w, x).squeeze(1)


class VideoEncoderOracle(nn.Module):
    def __init__(self, num_frames=16, spatial=(224, 224), output_dim=512, num_heads=4, aggregator_depth=2,
                 token_pooling_mode="mean", attention_pool_heads=8, use_rope=False, rope_base=10000.0,
                 rope_temporal_scale=1.0, aggregate_videos_tokens=True, per_video_pool=False, backbone="mvit"):
        super().__init__()
        self.model = MViTv2S(num_frames, spatial)
        # backbone="mvit_rope" (models/video_encoder.py:84-90, :256-257, :544-561): RoPE is never patched in and the forward is
        # torchvision's MViT.forward = norm -> x[:, 0] -> Identity head: one token per clip
        self.pooled_cls_fallback = backbone == "mvit_rope"
        if use_rope and backbone == "mvit":
            enable_rope(self.model, rope_base, rope_temporal_scale)
        self.proj = nn.Sequential(nn.Dropout(0.0), nn.Linear(768, output_dim), nn.GELU(), nn.Dropout(0.0))
        self.aggregator = AggregatorOracle(output_dim, num_heads, aggregator_depth)
        self.attention_pool = AttentionPoolOracle(output_dim, attention_pool_heads) if token_pooling_mode == "attention" else (
            AttentionPoolWithCLSOracle(output_dim, attention_pool_heads) if token_pooling_mode == "cls_token" else None)
        if token_pooling_mode == "cls_token":  # allocated but never used, as in the reference (models/video_encoder.py:147-153)
            self.cls_token = nn.Parameter(torch.zeros(1, 1, 768))
        self._apply_aggregator, self._per_video_pool = aggregate_videos_tokens, per_video_pool

    def features(self, x):
        if x.ndim == 5:
            x = x.unsqueeze(1)
        x = x.permute(0, 1, 5, 2, 3, 4)
        B, N, C, T, H, W = x.shape
        feats = self.model.forward_features(x.reshape(B * N, C, T, H, W).to(self.model.conv_proj.weight.dtype))
        if self.pooled_cls_fallback:
            feats = feats[:, 0].unsqueeze(1)
        tok = self.proj(feats)
        L, D = tok.shape[1], tok.shape[2]
        tok = tok.view(B, N, L, D)
        if self.attention_pool is not None:
            per_video = torch.stack([self.attention_pool(tok[:, i]) for i in range(N)], dim=1)
        else:
            per_video = tok.mean(dim=2)
        return {"token_grid": tok, "per_video_tokens": per_video, "patch_tokens": tok.reshape(B, N * L, D),
                "study_features": self.aggregator(per_video)}

    def forward(self, x):
        f = self.features(x)
        if self._apply_aggregator:
            return f["study_features"]
        if self._per_video_pool:
            p = f["per_video_tokens"]
            return p.squeeze(1) if p.shape[1] == 1 else p
        return f["patch_tokens"]


# --------------------------------------------------------------------------------------------- BERT
class _BertEmbeddings(nn.Module):
    def __init__(self, vocab, hidden, max_pos, eps):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, hidden)
        self.position_embeddings = nn.Embedding(max_pos, hidden)
        self.token_type_embeddings = nn.Embedding(2, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)

    def forward(self, ids):
        S = ids.shape[1]
        x = self.word_embeddings(ids) + self.position_embeddings.weight[:S][None] + self.token_type_embeddings.weight[0]
        return self.LayerNorm(x)


class _Self(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h)


class _SelfOut(nn.Module):
    def __init__(self, hin, h, eps):
        super().__init__()
        self.dense = nn.Linear(hin, h)
        self.LayerNorm = nn.LayerNorm(h, eps=eps)


class _Attn(nn.Module):
    def __init__(self, h, eps):
        super().__init__()
        self.self = _Self(h)
        self.output = _SelfOut(h, h, eps)


class _Inter(nn.Module):
    def __init__(self, h, f):
        super().__init__()
        self.dense = nn.Linear(h, f)


class _BertLayer(nn.Module):
    def __init__(self, h, f, heads, eps):
        super().__init__()
        self.heads = heads
        self.attention = _Attn(h, eps)
        self.intermediate = _Inter(h, f)
        self.output = _SelfOut(f, h, eps)

    def forward(self, x, bias):
        B, S, Hd = x.shape
        hd = Hd // self.heads
        s = self.attention.self
        q = s.query(x).view(B, S, self.heads, hd).transpose(1, 2)
        k = s.key(x).view(B, S, self.heads, hd).transpose(1, 2)
        v = s.value(x).view(B, S, self.heads, hd).transpose(1, 2)
        a = (q @ k.transpose(-2, -1)) / math.sqrt(hd) + bias[:, None, None, :]
        ctx = (a.softmax(-1) @ v).transpose(1, 2).reshape(B, S, Hd)
        x = self.attention.output.LayerNorm(self.attention.output.dense(ctx) + x)
        h = F.gelu(self.intermediate.dense(x))
        return self.output.LayerNorm(self.output.dense(h) + x)


class _Encoder(nn.Module):
    def __init__(self, n, h, f, heads, eps):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(h, f, heads, eps) for _ in range(n)])


class BertOracle(nn.Module):
    """BERT-base encoder without pooler; HF BertModel state_dict names."""

    def __init__(self, vocab=30522, hidden=768, layers=12, heads=12, ffn=3072, max_pos=512, eps=1e-12):
        super().__init__()
        self.embeddings = _BertEmbeddings(vocab, hidden, max_pos, eps)
        self.encoder = _Encoder(layers, hidden, ffn, heads, eps)

    def forward(self, input_ids, attention_mask):
        x = self.embeddings(input_ids)
        bias = (1.0 - attention_mask.to(x.dtype)) * torch.finfo(x.dtype).min
        for l in self.encoder.layer:
            x = l(x, bias)
        return x


class TextEncoderOracle(nn.Module):
    def __init__(self, output_dim=512, **bert_kw):
        super().__init__()
        self.bert = BertOracle(**bert_kw)
        hidden = self.bert.embeddings.word_embeddings.weight.shape[1]
        self.proj = nn.Sequential(nn.Dropout(0.0), nn.Linear(hidden, output_dim), nn.GELU(), nn.Dropout(0.0))

    def forward(self, input_ids, attention_mask):
        return self.proj(self.bert(input_ids, attention_mask)[:, 0])
