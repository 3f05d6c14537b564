"""TextEncoder with the reference's contract (models/text_encoder.py:26-137) on HIP kernels: BERT-base encoder
(HF `BertModel` parameter names under `.bert`, pooler removed) -> CLS token -> proj (Dropout-Linear-GELU-Dropout).

The reference constructs `AutoModel.from_pretrained(<hub name>)`; there is no network here, so the trunk is built
from a config (BERT-base-uncased shape = PubMedBERT's) with random init, and `model_name` may be a LOCAL directory
holding `config.json` / weights / tokenizer files, in which case they are loaded through transformers.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from . import functional as Fn
from .registry import ModelRegistry


def get_tokenizer(model_name="microsoft/BiomedNLP-PubMedBERT-base-uncased-abstract-fulltext"):
    from transformers import AutoTokenizer

    return AutoTokenizer.from_pretrained(model_name, use_fast=True, model_max_length=1024, padding_side="right",
                                         truncation_side="right")


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Embeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)


class _DenseLN(nn.Module):
    def __init__(self, fin, fout, eps):
        super().__init__()
        self.dense = nn.Linear(fin, fout)
        self.LayerNorm = nn.LayerNorm(fout, eps=eps)


class _AttentionBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _DenseLN(c.hidden_size, c.hidden_size, c.layer_norm_eps)


class _Intermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = _AttentionBlock(c)
        self.intermediate = _Intermediate(c)
        self.output = _DenseLN(c.intermediate_size, c.hidden_size, c.layer_norm_eps)


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])


class HipBertModel(nn.Module):
    """BERT encoder (no pooler) with HF `BertModel` state_dict names; forward on HIP kernels."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1):
        super().__init__()
        self.config = _Cfg(vocab_size=vocab_size, hidden_size=hidden_size, num_hidden_layers=num_hidden_layers,
                           num_attention_heads=num_attention_heads, intermediate_size=intermediate_size,
                           max_position_embeddings=max_position_embeddings, type_vocab_size=type_vocab_size,
                           layer_norm_eps=layer_norm_eps, hidden_dropout_prob=hidden_dropout_prob,
                           attention_probs_dropout_prob=attention_probs_dropout_prob)
        c = self.config
        self.embeddings = _Embeddings(c)
        self.encoder = _Encoder(c)
        self.pooler = None
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, std=0.02)
                if isinstance(m, nn.Linear):
                    nn.init.zeros_(m.bias)

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, compute_dtype: torch.dtype):
        """-> last_hidden_state [B, S, hidden] in the compute dtype."""
        c = self.config
        B, S = input_ids.shape
        if S > c.max_position_embeddings:
            raise ValueError(f"sequence length {S} exceeds max_position_embeddings {c.max_position_embeddings}")
        H, hd = c.num_attention_heads, c.hidden_size // c.num_attention_heads
        e = self.embeddings
        x = Fn.embed(input_ids, e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight,
                     compute_dtype)
        x = Fn.layernorm(x, e.LayerNorm.weight, e.LayerNorm.bias, c.layer_norm_eps)
        pdrop = c.hidden_dropout_prob
        x = Fn.dropout(x, pdrop, self.training)
        # additive key bias: 0 for valid tokens, a large negative for padding (HF extended attention mask)
        kbias = None
        if attention_mask is not None:
            m = attention_mask.to(torch.float32)
            kbias = ((1.0 - m) * -1e30).contiguous()
        padrop = c.attention_probs_dropout_prob  # HF BertSelfAttention: dropout on the softmax output (training only)
        for l in self.encoder.layer:
            s = l.attention.self
            # one autograd node for the three projections + the residual use of x: their four gradients meet in GEMM epilogues
            q, k, v, x = Fn.linears_skip(x, (s.query.weight, s.query.bias), (s.key.weight, s.key.bias), (s.value.weight, s.value.bias))
            q, k, v = q.view(B, S, H, hd), k.view(B, S, H, hd), v.view(B, S, H, hd)
            ctx = Fn.attention(q, k, v, 1.0 / math.sqrt(hd), kbias, dropout_p=padrop, training=self.training).view(B * S, c.hidden_size)
            o = l.attention.output
            if self.training and pdrop > 0:
                h = Fn.add(Fn.dropout(Fn.linear(ctx, o.dense.weight, o.dense.bias), pdrop, True), x)
            else:
                h = Fn.linear(ctx, o.dense.weight, o.dense.bias, residual=x)
            x = Fn.layernorm(h, o.LayerNorm.weight, o.LayerNorm.bias, c.layer_norm_eps)
            f = l.output
            if self.training and pdrop > 0:
                inter = Fn.linear(x, l.intermediate.dense.weight, l.intermediate.dense.bias, gelu=True)
                h = Fn.add(Fn.dropout(Fn.linear(inter, f.dense.weight, f.dense.bias), pdrop, True), x)
            else:
                h = Fn.mlp(x, l.intermediate.dense.weight, l.intermediate.dense.bias, f.dense.weight, f.dense.bias, residual=x)
            x = Fn.layernorm(h, f.LayerNorm.weight, f.LayerNorm.bias, c.layer_norm_eps)
        return x.view(B, S, c.hidden_size)


@ModelRegistry.register("text_encoder")
class TextEncoder(nn.Module):
    def __init__(self, model_name="microsoft/BiomedNLP-PubMedBERT-base-uncased-abstract-fulltext", output_dim=512,
                 dropout=0.2, freeze_ratio=0.5, compute_dtype: torch.dtype = torch.bfloat16, bert_config: dict | None = None):
        super().__init__()
        self.model_name = model_name
        self.output_dim = output_dim
        self.freeze_ratio = freeze_ratio
        self.dropout = dropout
        self.compute_dtype = compute_dtype
        self.tokenizer = None
        cfg = dict(bert_config or {})
        local = isinstance(model_name, str) and os.path.isdir(model_name)
        if local:
            import json

            with open(os.path.join(model_name, "config.json")) as f:
                hf = json.load(f)
            for k in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
                      "max_position_embeddings", "type_vocab_size", "layer_norm_eps", "hidden_dropout_prob",
                      "attention_probs_dropout_prob"):
                if k in hf:
                    cfg.setdefault(k, hf[k])
            try:
                self.tokenizer = get_tokenizer(model_name)
            except Exception as e:  # tokenizer files are optional for the training step itself
                print(f"[TextEncoder] tokenizer not loaded from {model_name}: {e}")
        self.bert = HipBertModel(**cfg)
        if local:
            self._load_local_weights(model_name)
        hidden_size = self.bert.config.hidden_size
        self._freeze_partial_bert()
        self.proj = nn.Sequential(nn.Dropout(self.dropout), nn.Linear(hidden_size, output_dim), nn.GELU(), nn.Dropout(self.dropout))

    def _load_local_weights(self, path):
        sd = None
        st = os.path.join(path, "model.safetensors")
        pt = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file

            sd = load_file(st)
        elif os.path.exists(pt):
            sd = torch.load(pt, map_location="cpu")
        if sd is None:
            return
        sd = {k[len("bert."):] if k.startswith("bert.") else k: v for k, v in sd.items()}
        own = self.bert.state_dict()
        sd = {k: v for k, v in sd.items() if k in own}
        missing, unexpected = self.bert.load_state_dict(sd, strict=False)
        print(f"[TextEncoder] loaded {len(sd)} tensors from {path} (missing {len(missing)})")

    def _freeze_partial_bert(self):
        if self.freeze_ratio == 0.0:
            return
        params = list(self.bert.named_parameters())
        train_count = int(self.freeze_ratio * len(params))
        for i, (_, p) in enumerate(params):
            if i < (len(params) - train_count):
                p.requires_grad = False

    def update_freeze_ratio(self, new_freeze_ratio: float):
        self.freeze_ratio = new_freeze_ratio
        for p in self.bert.parameters():
            p.requires_grad = True
        self._freeze_partial_bert()
        trainable = sum(p.numel() for p in self.bert.parameters() if p.requires_grad)
        total = sum(p.numel() for p in self.bert.parameters())
        print(f"[TextEncoder] Updated freeze_ratio to {new_freeze_ratio:.2f}")
        print(f"[TextEncoder] Trainable parameters: {trainable:,} / {total:,} ({100.0 * trainable / max(total, 1):.1f}%)")

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        hidden = self.bert(input_ids, attention_mask, self.compute_dtype)
        B, S, Hd = hidden.shape
        pooled = hidden.view(B, S * Hd)[:, :Hd]  # CLS token rows (strided view, no copy)
        x = Fn.dropout(pooled, self.dropout, self.training)
        x = Fn.linear(x, self.proj[1].weight, self.proj[1].bias, gelu=True)
        return Fn.dropout(x, self.dropout, self.training)
