"""ORACLE (test infrastructure, CPU only) — plain-PyTorch restatement of torchvision's MViT-v2-S token path.

The reference builds `torchvision.models.video.mvit_v2_s` (models/video_encoder.py:85,93), replaces the
head by Identity (:107) and monkey-patches `forward_features` (:232-253) and, with `use_rope`, the attention
forward (:268-378).  torchvision 0.25.0 (uv.lock) is NOT installed in the build container and its source is
not under /root/reference, so this file restates upstream `torchvision/models/video/mvit.py` from its
published algorithm.  PARITY UNPINNED against torchvision itself; pinned only by structural known answers
(34 230 144 backbone params without the head, torchvision state_dict key names, token counts 25089 -> 393)
and, for the RoPE pieces, by golden vectors from the importable `models/rope_3d.py`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

MVIT_V2_S = {
    "num_heads": [1, 2, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 8, 8],
    "input_channels": [96, 96, 192, 192, 384, 384, 384, 384, 384, 384, 384, 384, 384, 384, 384, 768],
    "output_channels": [96, 192, 192, 384, 384, 384, 384, 384, 384, 384, 384, 384, 384, 384, 768, 768],
    "stride_q": [[1, 1, 1], [1, 2, 2], [1, 1, 1], [1, 2, 2], [1, 1, 1], [1, 1, 1], [1, 1, 1], [1, 1, 1], [1, 1, 1],
                 [1, 1, 1], [1, 1, 1], [1, 1, 1], [1, 1, 1], [1, 1, 1], [1, 2, 2], [1, 1, 1]],
    "stride_kv": [[1, 8, 8], [1, 4, 4], [1, 4, 4], [1, 2, 2], [1, 2, 2], [1, 2, 2], [1, 2, 2], [1, 2, 2], [1, 2, 2],
                  [1, 2, 2], [1, 2, 2], [1, 2, 2], [1, 2, 2], [1, 2, 2], [1, 1, 1], [1, 1, 1]],
}


def block_plan(temporal_size: int, spatial_size: Sequence[int]):
    """Per-block geometry shared by the oracle and the HIP encoder (host logic, no arithmetic)."""
    t0 = temporal_size // 2
    s0 = (spatial_size[0] // 4, spatial_size[1] // 4)
    size = [t0, s0[0], s0[1]]
    plan = []
    for i in range(16):
        sq, skv = MVIT_V2_S["stride_q"][i], MVIT_V2_S["stride_kv"][i]
        smax = max(size[1:])
        q_size, kv_size = smax // sq[1], smax // skv[1]
        plan.append({
            "index": i,
            "in_ch": MVIT_V2_S["input_channels"][i],
            "out_ch": MVIT_V2_S["output_channels"][i],
            "heads": MVIT_V2_S["num_heads"][i],
            "stride_q": tuple(sq),
            "stride_kv": tuple(skv),
            "thw": tuple(size),
            "rel_sp": 2 * max(q_size, kv_size) - 1,
            "rel_t": 2 * size[0] - 1,
        })
        size = [s // st for s, st in zip(size, sq)]
    return plan


class _DepthwiseConv3dTaps(torch.autograd.Function):
    """Depthwise 3 x 3 x 3 convolution (padding 1, stride (st, sh, sw), no bias) on channels-LAST grids [..., T, H, W, C]:
    out[..., t, h, w, c] = sum over the 27 taps of weight[c, 0, dt, dh, dw] * x_padded[..., t*st + dt, h*sh + dh, w*sw + dw, c]
    — the arithmetic of nn.Conv3d(C, C, 3, stride, padding=1, groups=C, bias=False) (torchvision mvit.Pool's `pool`), written as
    27 strided-slice multiply-adds with a hand-written backward of the same form.  ATen's CPU paths for depthwise 3-D
    convolutions (slow_conv3d_forward in fp64, convolution_backward in both precisions) took 40 % of the oracle's time and need
    a channels-first copy of every q / k / v tensor; equality with F.conv3d is checked in tests/test_oracle_golden.py."""

    @staticmethod
    def forward(ctx, x, weight, stride):
        st, sh, sw = stride
        T, H, W, C = x.shape[-4:]
        To, Ho, Wo = (T - 1) // st + 1, (H - 1) // sh + 1, (W - 1) // sw + 1
        xp = F.pad(x, (0, 0, 1, 1, 1, 1, 1, 1))
        out = torch.zeros(*x.shape[:-4], To, Ho, Wo, C, dtype=x.dtype)
        for dt in range(3):
            for dh in range(3):
                for dw in range(3):
                    sl = xp[..., dt:dt + st * (To - 1) + 1:st, dh:dh + sh * (Ho - 1) + 1:sh, dw:dw + sw * (Wo - 1) + 1:sw, :]
                    out.addcmul_(sl, weight[:, 0, dt, dh, dw])
        ctx.save_for_backward(xp, weight)
        ctx.stride, ctx.out_thw = stride, (To, Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, dy):
        xp, weight = ctx.saved_tensors
        st, sh, sw = ctx.stride
        To, Ho, Wo = ctx.out_thw
        dxp = torch.zeros_like(xp)
        dw = torch.zeros_like(weight)
        red = tuple(range(dy.dim() - 1))
        for dt in range(3):
            for dh in range(3):
                for dw_ in range(3):
                    idx = (Ellipsis, slice(dt, dt + st * (To - 1) + 1, st), slice(dh, dh + sh * (Ho - 1) + 1, sh),
                           slice(dw_, dw_ + sw * (Wo - 1) + 1, sw), slice(None))
                    dxp[idx].addcmul_(dy, weight[:, 0, dt, dh, dw_])
                    dw[:, 0, dt, dh, dw_] = (xp[idx] * dy).sum(dim=red)
        return dxp[..., 1:-1, 1:-1, 1:-1, :], dw, None


def depthwise_conv3d_taps(x, weight, stride):
    return _DepthwiseConv3dTaps.apply(x, weight, tuple(stride))


class Pool(nn.Module):
    """torchvision mvit.Pool: split class token, pool the (T,H,W) grid, re-attach, then norm (norm covers the class token)."""

    def __init__(self, pool: nn.Module, norm: nn.Module | None):
        super().__init__()
        self.pool = pool
        self.norm_act = nn.Sequential(norm) if norm is not None else None

    def forward(self, x, thw):
        # x: [B, heads, N, C]
        cls, x = x[:, :, :1], x[:, :, 1:]
        B, Hh, N, C = x.shape
        p = self.pool
        if isinstance(p, nn.Conv3d) and p.groups == C and p.kernel_size == (3, 3, 3) and p.padding == (1, 1, 1) and p.bias is None:
            # the q / k / v pools: same arithmetic as `self.pool(x)` on the channels-first grid, without ATen's slow depthwise path
            y = depthwise_conv3d_taps(x.reshape(B, Hh, *thw, C), p.weight, p.stride)
            T, H, W = y.shape[2:5]
            x = torch.cat((cls, y.reshape(B, Hh, T * H * W, C)), dim=2)
            if self.norm_act is not None:
                x = self.norm_act(x)
            return x, (T, H, W)
        x = x.transpose(2, 3).reshape(B * Hh, C, *thw)
        x = self.pool(x)
        T, H, W = x.shape[2:]
        x = x.reshape(B, Hh, C, -1).transpose(2, 3)
        x = torch.cat((cls, x), dim=2)
        if self.norm_act is not None:
            x = self.norm_act(x)
        return x, (T, H, W)


def add_rel_pos(attn, q, q_thw, k_thw, rel_pos_h, rel_pos_w, rel_pos_t):
    """torchvision mvit._add_rel_pos (decomposed relative positional bias on attn[:, :, 1:, 1:])."""
    q_t, q_h, q_w = q_thw
    k_t, k_h, k_w = k_thw

    def dist(qs, ks):
        qr, kr = max(ks / qs, 1.0), max(qs / ks, 1.0)
        return (torch.arange(qs)[:, None] * qr - (torch.arange(ks)[None, :] + (1.0 - ks)) * kr).long()

    def interp(emb, d):  # torchvision mvit._interpolate (only hit when a grid size is not a multiple of its stride)
        if emb.shape[0] == d:
            return emb
        return F.interpolate(emb.permute(1, 0).unsqueeze(0), size=d, mode="linear").squeeze(0).permute(1, 0)

    Rh = interp(rel_pos_h, 2 * max(q_h, k_h) - 1)[dist(q_h, k_h)]
    Rw = interp(rel_pos_w, 2 * max(q_w, k_w) - 1)[dist(q_w, k_w)]
    Rt = interp(rel_pos_t, 2 * max(q_t, k_t) - 1)[dist(q_t, k_t)]
    B, n_head, _, dim = q.shape
    r_q = q[:, :, 1:].reshape(B, n_head, q_t, q_h, q_w, dim)
    rel_h_q = torch.einsum("bythwc,hkc->bythwk", r_q, Rh)
    rel_w_q = torch.einsum("bythwc,wkc->bythwk", r_q, Rw)
    rel_t_q = torch.einsum("bythwc,tkc->bythwk", r_q, Rt)
    rel = (rel_h_q[:, :, :, :, :, None, :, None] + rel_w_q[:, :, :, :, :, None, None, :]
           + rel_t_q[:, :, :, :, :, :, None, None]).reshape(B, n_head, q_t * q_h * q_w, k_t * k_h * k_w)
    attn = attn.clone()
    attn[:, :, 1:, 1:] += rel
    return attn


class MultiscaleAttention(nn.Module):
    def __init__(self, input_size, embed_dim, output_dim, num_heads, stride_q, stride_kv, rel_sp, rel_t, eps):
        super().__init__()
        self.num_heads = num_heads
        self.output_dim = output_dim
        self.head_dim = output_dim // num_heads
        self.scaler = 1.0 / math.sqrt(self.head_dim)
        self.qkv = nn.Linear(embed_dim, 3 * output_dim)
        self.project = nn.Sequential(nn.Linear(output_dim, output_dim))
        hd = self.head_dim

        def mk(stride):
            return Pool(nn.Conv3d(hd, hd, (3, 3, 3), stride=tuple(stride), padding=(1, 1, 1), groups=hd, bias=False),
                        nn.LayerNorm(hd, eps=eps))

        self.pool_q, self.pool_k, self.pool_v = mk(stride_q), mk(stride_kv), mk(stride_kv)
        self.rel_pos_h = nn.Parameter(torch.zeros(rel_sp, hd))
        self.rel_pos_w = nn.Parameter(torch.zeros(rel_sp, hd))
        self.rel_pos_t = nn.Parameter(torch.zeros(rel_t, hd))
        self.rope = None  # set by enable_rope()

    def forward(self, x, thw):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).transpose(1, 3).unbind(dim=2)
        k, k_thw = self.pool_k(k, thw)
        v = self.pool_v(v, thw)[0]
        q, q_thw = self.pool_q(q, thw)
        if self.rope is not None:
            # RoPE-patched forward (models/video_encoder.py:307-378): rotate after pooling, scale after QK^T,
            # rel-pos bias on the rotated q, NO residual pooling.
            q, k = self.rope(q, k, q_thw, k_thw)
            attn = (q @ k.transpose(-2, -1)) * self.scaler
            attn = add_rel_pos(attn, q, q_thw, k_thw, self.rel_pos_h, self.rel_pos_w, self.rel_pos_t)
            x = attn.softmax(dim=-1) @ v
        else:
            attn = torch.matmul(self.scaler * q, k.transpose(2, 3))
            attn = add_rel_pos(attn, q, q_thw, k_thw, self.rel_pos_h, self.rel_pos_w, self.rel_pos_t)
            x = torch.matmul(attn.softmax(dim=-1), v)
            x = torch.cat((x[:, :, :1], x[:, :, 1:] + q[:, :, 1:]), dim=2)  # residual pooling, class token excluded
        x = self.project(x.transpose(1, 2).reshape(B, -1, self.output_dim))
        return x, q_thw


class MultiscaleBlock(nn.Module):
    def __init__(self, p, eps):
        super().__init__()
        cin, cout = p["in_ch"], p["out_ch"]
        self.pool_skip = None
        if math.prod(p["stride_q"]) > 1:
            ks = [s + 1 if s > 1 else s for s in p["stride_q"]]
            self.pool_skip = Pool(nn.MaxPool3d(ks, stride=p["stride_q"], padding=[k // 2 for k in ks]), None)
        self.norm1 = nn.LayerNorm(cin, eps=eps)
        self.norm2 = nn.LayerNorm(cout, eps=eps)
        self.attn = MultiscaleAttention(p["thw"], cin, cout, p["heads"], p["stride_q"], p["stride_kv"], p["rel_sp"],
                                        p["rel_t"], eps)
        self.mlp = nn.Sequential(nn.Linear(cout, 4 * cout), nn.GELU(), nn.Dropout(0.0), nn.Linear(4 * cout, cout),
                                 nn.Dropout(0.0))
        self.project = nn.Linear(cin, cout) if cin != cout else None

    def forward(self, x, thw):
        x_norm1 = self.norm1(x)
        x_attn, thw_new = self.attn(x_norm1, thw)
        x = x if self.project is None else self.project(x_norm1)
        if self.pool_skip is not None:
            x = self.pool_skip(x.unsqueeze(1), thw)[0].squeeze(1)
        x = x + x_attn
        x_norm2 = self.norm2(x)
        return x + self.mlp(x_norm2), thw_new


class PositionalEncoding(nn.Module):
    def __init__(self, embed_size):
        super().__init__()
        self.class_token = nn.Parameter(torch.zeros(embed_size))

    def forward(self, x):
        return torch.cat((self.class_token.expand(x.size(0), -1).unsqueeze(1), x), dim=1)


class MViTv2S(nn.Module):
    """Size-parameterised MViT-v2-S trunk with torchvision's parameter names (head = Identity)."""

    def __init__(self, temporal_size: int = 16, spatial_size: Sequence[int] = (224, 224), eps: float = 1e-6):
        super().__init__()
        self.temporal_size, self.spatial_size = temporal_size, tuple(spatial_size)
        self.plan = block_plan(temporal_size, spatial_size)
        self.conv_proj = nn.Conv3d(3, 96, kernel_size=(3, 7, 7), stride=(2, 4, 4), padding=(1, 3, 3))
        self.pos_encoding = PositionalEncoding(96)
        self.blocks = nn.ModuleList([MultiscaleBlock(p, eps) for p in self.plan])
        self.norm = nn.LayerNorm(768, eps=eps)
        self.head = nn.Identity()

    def forward_features(self, x):
        # x: [B, C, T, H, W]
        x = self.conv_proj(x).flatten(2).transpose(1, 2)
        x = self.pos_encoding(x)
        thw = self.plan[0]["thw"]
        for blk in self.blocks:
            x, thw = blk(x, thw)
        return self.norm(x)


# ------------------------------------------------------------------------------------------- RoPE
def rope_tables(T, H, W, head_dim, base=10000.0, temporal_scale=1.0, n_special=1):
    """cos/sin [n_special + T*H*W, head_dim] as models/rope_3d.py:107-184 builds them (dims split 3 ways)."""
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
    return cos, sin


def rotate_half(x):
    return torch.stack((-x[..., 1::2], x[..., 0::2]), dim=-1).flatten(-2)


class Rope3DOracle:
    def __init__(self, head_dim, base=10000.0, temporal_scale=1.0):
        self.head_dim, self.base, self.ts = head_dim, base, temporal_scale

    def one(self, x, thw):
        cos, sin = rope_tables(*thw, self.head_dim, self.base, self.ts, n_special=x.shape[2] - math.prod(thw))
        return x * cos + rotate_half(x) * sin

    def __call__(self, q, k, q_thw, k_thw):
        return self.one(q, q_thw), self.one(k, k_thw)


def enable_rope(model: MViTv2S, base=10000.0, temporal_scale=1.0):
    """models/video_encoder.py:118-121,319-321: RoPE only where (heads, head_dim) in {(4,96),(8,96)} = blocks 3..15."""
    for blk in model.blocks:
        if blk.attn.num_heads in (4, 8) and blk.attn.head_dim == 96:
            blk.attn.rope = Rope3DOracle(96, base, temporal_scale)
