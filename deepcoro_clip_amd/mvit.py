"""MViT-v2-S trunk on HIP kernels, with torchvision's parameter names (state_dict compatible).

Replaces `torchvision.models.video.mvit_v2_s` + the monkey-patched `forward_features` / RoPE attention of
models/video_encoder.py:85-107,232-253,259-388.  torch.nn modules below are PARAMETER CONTAINERS only
(names, default init, state_dict); their forward() is never called — all arithmetic goes through
`functional.py` -> `libdcc_hip.so`.

Token layout: [B' * N, C] row-major (class token first in every clip), heads interleaved inside C exactly
like the qkv GEMM writes them, so no transposes happen between GEMMs, pooling and attention.
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K

MVIT_V2_S = {
    "num_heads": [1, 2, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 8, 8],
    "input_channels": [96, 96, 192, 192, 384, 384, 384, 384, 384, 384, 384, 384, 384, 384, 384, 768],
    "output_channels": [96, 192, 192, 384, 384, 384, 384, 384, 384, 384, 384, 384, 384, 384, 768, 768],
    "stride_q": [(1, 1, 1), (1, 2, 2), (1, 1, 1), (1, 2, 2)] + [(1, 1, 1)] * 10 + [(1, 2, 2), (1, 1, 1)],
    "stride_kv": [(1, 8, 8), (1, 4, 4), (1, 4, 4)] + [(1, 2, 2)] * 11 + [(1, 1, 1), (1, 1, 1)],
}


def _conv_out(size, stride, k=3, p=1):
    return tuple((s + 2 * p - k) // st + 1 for s, st in zip(size, stride))


def block_plan(temporal_size: int, spatial_size: Sequence[int]):
    """Static geometry of the 16 blocks: bookkeeping sizes as torchvision computes them (floor division, used for
    the rel-pos table lengths) next to the actual conv-arithmetic grids."""
    book = [temporal_size // 2, spatial_size[0] // 4, spatial_size[1] // 4]
    actual = ((temporal_size + 2 - 3) // 2 + 1, (spatial_size[0] + 6 - 7) // 4 + 1, (spatial_size[1] + 6 - 7) // 4 + 1)
    plan = []
    for i in range(16):
        sq, skv = MVIT_V2_S["stride_q"][i], MVIT_V2_S["stride_kv"][i]
        smax = max(book[1:])
        q_size, kv_size = smax // sq[1], smax // skv[1]
        plan.append(dict(index=i, in_ch=MVIT_V2_S["input_channels"][i], out_ch=MVIT_V2_S["output_channels"][i],
                         heads=MVIT_V2_S["num_heads"][i], stride_q=sq, stride_kv=skv, thw=tuple(actual),
                         q_thw=_conv_out(actual, sq), k_thw=_conv_out(actual, skv),
                         rel_sp=2 * max(q_size, kv_size) - 1, rel_t=2 * book[0] - 1))
        book = [s // st for s, st in zip(book, sq)]
        actual = _conv_out(actual, sq)
    return plan


def forward_macs(plan, patch_tokens: int | None = None) -> dict:
    """Multiply-accumulates of one clip's forward pass, per piece (torchvision publishes 64.224 GMACs for
    16 x 224 x 224).  Used by bench.py's roofline accounting and pinned by tests/test_host_cpu.py."""
    macs = {k: 0 for k in ("conv_proj", "qkv", "pool", "attn", "relpos", "proj", "skip", "mlp")}
    l0 = patch_tokens if patch_tokens is not None else math.prod(plan[0]["thw"])
    macs["conv_proj"] = l0 * 96 * 3 * 3 * 7 * 7
    for p in plan:
        cin, cout, h = p["in_ch"], p["out_ch"], p["heads"]
        hd = cout // h
        nin, nq, nk = math.prod(p["thw"]) + 1, math.prod(p["q_thw"]) + 1, math.prod(p["k_thw"]) + 1
        macs["qkv"] += nin * cin * 3 * cout
        macs["pool"] += ((nq - 1) + 2 * (nk - 1)) * cout * 27
        macs["attn"] += 2 * h * nq * nk * hd
        macs["relpos"] += h * (nq - 1) * hd * sum(p["k_thw"])
        macs["proj"] += nq * cout * cout
        if cin != cout:
            macs["skip"] += nin * cin * cout
        macs["mlp"] += 2 * nq * cout * 4 * cout
    macs["total"] = sum(macs.values())
    return macs


class _Pool(nn.Module):
    """container: torchvision mvit.Pool(Conv3d depthwise, LayerNorm)"""

    def __init__(self, hd, stride, eps):
        super().__init__()
        self.pool = nn.Conv3d(hd, hd, (3, 3, 3), stride=stride, padding=(1, 1, 1), groups=hd, bias=False)
        self.norm_act = nn.Sequential(nn.LayerNorm(hd, eps=eps))


class _Attention(nn.Module):
    def __init__(self, p, eps):
        super().__init__()
        cin, cout, heads = p["in_ch"], p["out_ch"], p["heads"]
        hd = cout // heads
        self.num_heads, self.head_dim, self.output_dim = heads, hd, cout
        self.qkv = nn.Linear(cin, 3 * cout)
        self.project = nn.Sequential(nn.Linear(cout, cout))
        self.pool_q = _Pool(hd, p["stride_q"], eps)
        self.pool_k = _Pool(hd, p["stride_kv"], eps)
        self.pool_v = _Pool(hd, p["stride_kv"], eps)
        self.rel_pos_h = nn.Parameter(torch.zeros(p["rel_sp"], hd))
        self.rel_pos_w = nn.Parameter(torch.zeros(p["rel_sp"], hd))
        self.rel_pos_t = nn.Parameter(torch.zeros(p["rel_t"], hd))
        for t in (self.rel_pos_h, self.rel_pos_w, self.rel_pos_t):
            nn.init.trunc_normal_(t, std=0.02)


class _Block(nn.Module):
    def __init__(self, p, eps, sd_prob: float = 0.0):
        super().__init__()
        cin, cout = p["in_ch"], p["out_ch"]
        self.p = p
        self.eps = eps
        self.sd_prob = float(sd_prob)  # torchvision StochasticDepth(sd_prob, "row") on both residual branches
        self.norm1 = nn.LayerNorm(cin, eps=eps)
        self.norm2 = nn.LayerNorm(cout, eps=eps)
        self.attn = _Attention(p, eps)
        self.mlp = nn.Sequential(nn.Linear(cout, 4 * cout), nn.GELU(), nn.Dropout(0.0), nn.Linear(4 * cout, cout),
                                 nn.Dropout(0.0))
        self.project = nn.Linear(cin, cout) if cin != cout else None
        self._idx_cache: dict = {}

    def _idx(self, device):
        """(idx, inv, interp) device tables: idx_a[q][k] = table row of a (query coord, key coord) pair, inv_a[q][l] =
        the key coord whose pair uses row l (or -1), interp_a = [d, L] resampling matrix when the stored table length L
        differs from d = 2 max(q, k) - 1 (torchvision mvit._interpolate: grids that are not multiples of their
        strides, e.g. 112 x 112 inputs), else None."""
        key = str(device)
        if key not in self._idx_cache:
            p = self.p
            idx, inv, interp = [], [], []
            for a, table in ((1, self.attn.rel_pos_h), (2, self.attn.rel_pos_w), (0, self.attn.rel_pos_t)):
                qs, ks = p["q_thw"][a], p["k_thw"][a]
                d = 2 * max(qs, ks) - 1
                interp.append(None if table.shape[0] == d else Fn.interp_matrix(table.shape[0], d).to(device))
                ix = Fn.relpos_index(qs, ks)
                iv = torch.full((qs, d), -1, dtype=torch.int32)
                iv.scatter_(1, ix.long(), torch.arange(ks, dtype=torch.int32).expand(qs, ks).contiguous())
                idx.append(ix.to(device))
                inv.append(iv.contiguous().to(device))
            self._idx_cache[key] = (tuple(idx), tuple(inv), tuple(interp))
        return self._idx_cache[key]

    def forward_tokens(self, x, Bp, rope=None):
        """x: [B'*N, Cin] in the compute dtype -> [B'*Nq, Cout]."""
        p, a = self.p, self.attn
        hd = a.head_dim
        N = x.shape[0] // Bp
        if self.project is None:  # x feeds norm1 and, unchanged, the skip branch: their gradients meet in one kernel
            xn1, x = Fn.layernorm_skip(x, self.norm1.weight, self.norm1.bias, self.eps)
        else:
            xn1 = Fn.layernorm(x, self.norm1.weight, self.norm1.bias, self.eps)
        if self.project is None:
            qkv, xproj = Fn.linear(xn1, a.qkv.weight, a.qkv.bias), None
        else:  # both read norm1(x): one autograd node, the data gradients chained through a residual epilogue
            qkv, xproj = Fn.linear2(xn1, a.qkv.weight, a.qkv.bias, self.project.weight, self.project.bias)
        idx, inv, interp = self._idx(x.device)
        cfg = dict(B=Bp, N=N, H=a.num_heads, D=hd, thw=p["thw"], eps=self.eps, stride_q=p["stride_q"],
                   stride_kv=p["stride_kv"], scale=1.0 / math.sqrt(hd), idx=idx, inv=inv, rope=rope)
        rel = [t if w is None else Fn.interp_rows(t, w) for t, w in zip((a.rel_pos_h, a.rel_pos_w, a.rel_pos_t), interp)]
        att = Fn.mvit_attention_core(
            qkv, a.pool_q.pool.weight, a.pool_k.pool.weight, a.pool_v.pool.weight,
            a.pool_q.norm_act[0].weight, a.pool_q.norm_act[0].bias, a.pool_k.norm_act[0].weight,
            a.pool_k.norm_act[0].bias, a.pool_v.norm_act[0].weight, a.pool_v.norm_act[0].bias,
            rel[0], rel[1], rel[2], cfg)
        xs = x if self.project is None else xproj
        if math.prod(p["stride_q"]) > 1:
            ks = tuple(s + 1 if s > 1 else s for s in p["stride_q"])
            xs = Fn.pool_skip(xs, Bp, p["thw"], ks, p["stride_q"], tuple(k // 2 for k in ks))
        drop = self.training and self.sd_prob > 0.0
        if drop:  # x = skip + sd(attn); x = x + sd(mlp): the residual add moves out of the GEMM epilogue
            x = Fn.stochastic_depth_add(xs, Fn.linear(att, a.project[0].weight, a.project[0].bias), self.sd_prob, Bp)
        else:
            x = Fn.linear(att, a.project[0].weight, a.project[0].bias, residual=xs)
        if Fn.ln_mlp_ok(x, self.mlp[0].weight, self.mlp[3].weight):   # norm2 + fc1 + GELU + fc2 (+ residual) as one launch
            y = Fn.ln_mlp(x, self.norm2.weight, self.norm2.bias, self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight,
                          self.mlp[3].bias, self.eps, residual=not drop)
            return Fn.stochastic_depth_add(x, y, self.sd_prob, Bp) if drop else y
        xn2, x = Fn.layernorm_skip(x, self.norm2.weight, self.norm2.bias, self.eps)  # x also is the MLP's residual
        if drop:
            return Fn.stochastic_depth_add(x, Fn.mlp(xn2, self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight,
                                                     self.mlp[3].bias), self.sd_prob, Bp)
        return Fn.mlp(xn2, self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight, self.mlp[3].bias, residual=x)


class _PosEnc(nn.Module):
    def __init__(self, c, temporal_size, spatial_size):
        super().__init__()
        self.class_token = nn.Parameter(torch.zeros(c))
        nn.init.trunc_normal_(self.class_token, std=0.02)
        self.temporal_size, self.spatial_size = temporal_size, tuple(spatial_size)


def rope_tables(T, H, W, head_dim, base=10000.0, temporal_scale=1.0, n_special=1):
    """cos/sin [n_special + T*H*W, head_dim], built on the host exactly as models/rope_3d.py:107-184."""
    t_dim = h_dim = head_dim // 3
    w_dim = head_dim - t_dim - h_dim

    def freqs(dim, length, b):
        inv = 1.0 / (b ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
        f = torch.einsum("n,d->nd", torch.arange(length, dtype=torch.float32), inv)
        return torch.stack((f, f), dim=-1).reshape(length, dim)

    tf, hf, wf = freqs(t_dim, T, base * temporal_scale), freqs(h_dim, H, base), freqs(w_dim, W, base)
    ang = torch.cat([tf.view(T, 1, 1, -1).expand(T, H, W, -1), hf.view(1, H, 1, -1).expand(T, H, W, -1),
                     wf.view(1, 1, W, -1).expand(T, H, W, -1)], dim=-1).reshape(T * H * W, head_dim)
    cos, sin = ang.cos(), ang.sin()
    if n_special:
        cos = torch.cat([torch.ones(n_special, head_dim), cos], 0)
        sin = torch.cat([torch.zeros(n_special, head_dim), sin], 0)
    return cos.contiguous(), sin.contiguous()


class HipMViT(nn.Module):
    """Size-parameterised MViT-v2-S (head = Identity): `forward_features` returns all tokens incl. the class token."""

    def __init__(self, temporal_size: int = 16, spatial_size: Sequence[int] = (224, 224), eps: float = 1e-6,
                 stochastic_depth_prob: float = 0.2):
        super().__init__()
        self.plan = block_plan(temporal_size, spatial_size)
        self.conv_proj = nn.Conv3d(3, 96, kernel_size=(3, 7, 7), stride=(2, 4, 4), padding=(1, 3, 3))
        self.pos_encoding = _PosEnc(96, temporal_size // 2, (spatial_size[0] // 4, spatial_size[1] // 4))
        # torchvision: sd_prob grows linearly over the 16 blocks, 0 -> stochastic_depth_prob (mvit_v2_s: 0.2)
        self.blocks = nn.ModuleList([_Block(p, eps, stochastic_depth_prob * i / 15.0) for i, p in enumerate(self.plan)])
        self.norm = nn.LayerNorm(768, eps=eps)
        self.head = nn.Identity()
        self.eps = eps
        self.temporal_size, self.spatial_size = temporal_size, tuple(spatial_size)
        self.rope_cfg = None  # (base, temporal_scale) when 3-D RoPE is enabled
        self._rope_cache: dict = {}
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def _rope(self, blk: _Block, device):
        if self.rope_cfg is None:
            return None
        a = blk.attn
        if not (a.num_heads in (4, 8) and a.head_dim == 96):  # models/video_encoder.py:118-121
            return None
        key = (blk.p["index"], str(device))
        if key not in self._rope_cache:
            base, ts = self.rope_cfg
            cq, sq = rope_tables(*blk.p["q_thw"], a.head_dim, base, ts)
            ck, sk = rope_tables(*blk.p["k_thw"], a.head_dim, base, ts)
            self._rope_cache[key] = dict(cos_q=cq.to(device), sin_q=sq.to(device), cos_k=ck.to(device), sin_k=sk.to(device))
        return self._rope_cache[key]

    def forward_features(self, video: torch.Tensor, compute_dtype: torch.dtype, norm=None) -> torch.Tensor:
        """video: [B', T, H, W, 3] fp32 (channels-last, as the dataset collates it), or uint8 raw pixels with
        norm = (mean[3], std[3]) applied inside the patch gather -> [B', L, 768]."""
        K.L.require_cuda(video)
        Bp, T, H, W, C = video.shape
        if (T, H, W) != (self.temporal_size, *self.spatial_size):
            raise ValueError(f"MViT built for {(self.temporal_size, *self.spatial_size)} got {(T, H, W)}")
        x = Fn.patch_embed(video, self.conv_proj.weight, self.conv_proj.bias, self.pos_encoding.class_token, compute_dtype, norm)
        for blk in self.blocks:
            x = blk.forward_tokens(x, Bp, rope=self._rope(blk, video.device))
        x = Fn.layernorm(x, self.norm.weight, self.norm.bias, self.eps)
        return x.view(Bp, -1, x.shape[-1])
