"""Thin, autograd-free Python wrappers over the C-ABI kernels (shape checks + launch).

Every function here calls into `libdcc_hip.so`; there is no torch fallback.
"""
from __future__ import annotations

import torch

from . import _lib as L

GEMM_BIAS, GEMM_GELU, GEMM_SAVE_PRE, GEMM_RESID, GEMM_DGELU, GEMM_ACCUM = 1, 2, 4, 8, 16, 32


def gemm(a: torch.Tensor, b: torch.Tensor, *, a_km: bool = False, b_km: bool = False, bias: torch.Tensor | None = None,
         resid: torch.Tensor | None = None, aux: torch.Tensor | None = None, gelu: bool = False,
         save_pre: bool = False, dgelu: bool = False, out: torch.Tensor | None = None, accumulate: bool = False,
         out_dtype: torch.dtype | None = None, splitk: int = 1, alpha: float = 1.0) -> torch.Tensor:
    """C[m,n] = alpha * sum_k A(m,k) B(n,k) with fused epilogue.

    a: [M,K] (or [K,M] if a_km); b: [N,K] (or [K,N] if b_km); 2-D, inner stride 1.
    """
    L.require_cuda(a, b, bias, resid, aux, out)
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    assert a.dtype == b.dtype
    M, K = (a.shape[1], a.shape[0]) if a_km else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if b_km else (b.shape[0], b.shape[1])
    assert K == Kb, (a.shape, b.shape, a_km, b_km)
    if out is None:
        od = out_dtype or a.dtype
        if splitk > 1 or accumulate:
            out = torch.zeros((M, N), device=a.device, dtype=od)
        else:
            out = torch.empty((M, N), device=a.device, dtype=od)
    assert out.shape == (M, N) and out.stride(1) == 1
    flags = 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
        flags |= GEMM_BIAS
    if gelu:
        flags |= GEMM_GELU
    if save_pre:
        assert aux is not None and aux.shape == (M, N) and aux.dtype == out.dtype
        flags |= GEMM_SAVE_PRE
    if dgelu:
        assert aux is not None and aux.shape == (M, N) and aux.dtype == a.dtype
        flags |= GEMM_DGELU
    if resid is not None:
        assert resid.shape == (M, N) and resid.dtype == out.dtype and resid.stride(1) == 1
        flags |= GEMM_RESID
    if accumulate:
        flags |= GEMM_ACCUM
    L.call("dcc_gemm", L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(bias), L.ptr(resid), L.ptr(aux), a.stride(0), b.stride(0),
           out.stride(0), resid.stride(0) if resid is not None else 0, aux.stride(0) if aux is not None else 0, M, N, K,
           1, 0, 0, 0, splitk, float(alpha), flags, int(a_km), int(b_km), L.dt(a), L.dt(out), L.stream())
    return out


def _strides3(t: torch.Tensor) -> list[int]:
    assert t.dim() == 4 and t.stride(3) == 1, "attention operands are [B, N, H, D] views with unit inner stride"
    return [t.stride(0), t.stride(1), t.stride(2)]


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, scale: float, kbias: torch.Tensor | None = None,
             residual: bool = False, res_start: int = 0):
    """O = softmax(scale * q k^T + kbias) v (+ q[..., :DV] for rows >= res_start). Returns (O [B,Nq,H,DV], LSE [B,H,Nq])."""
    import ctypes

    L.require_cuda(q, k, v, kbias)
    B, Nq, H, DQK = q.shape
    Nk, DV = k.shape[1], v.shape[3]
    assert k.shape == (B, Nk, H, DQK) and v.shape == (B, Nk, H, DV) and q.dtype == k.dtype == v.dtype
    o = torch.empty((B, Nq, H, DV), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, H, Nq), device=q.device, dtype=torch.float32)
    if kbias is not None:
        assert kbias.shape == (B, Nk) and kbias.dtype == torch.float32 and kbias.is_contiguous()
    st = (ctypes.c_int64 * 12)(*(_strides3(q) + _strides3(k) + _strides3(v) + _strides3(o)))
    L.call("dcc_attn_fwd", L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(lse), L.ptr(kbias),
           L.ptr(q) if residual else None, int(res_start), st, B, H, Nq, Nk, DQK, DV, float(scale), L.dt(q), L.stream())
    return o, lse


def attn_bwd(q, k, v, o, do, lse, *, scale: float, kbias=None, residual: bool = False, res_start: int = 0,
             qsplit: int | None = None):
    """Gradients of attn_fwd w.r.t. q (attention path only; the residual's dq = do is the caller's), k, v."""
    import ctypes

    L.require_cuda(q, k, v, o, do, lse)
    B, Nq, H, DQK = q.shape
    Nk, DV = k.shape[1], v.shape[3]
    assert do.shape == o.shape and do.stride() == o.stride(), "dO must share O's layout"
    if qsplit is None:
        kblocks = ((Nk + 127) // 128) * H * B
        qsplit = 1 if kblocks >= 512 else max(1, min((Nq + 63) // 64, (1024 + kblocks - 1) // kblocks))
    dq = torch.empty((B, Nq, H, DQK), device=q.device, dtype=q.dtype)
    kv_dtype = torch.float32 if qsplit > 1 else q.dtype
    alloc = torch.zeros if qsplit > 1 else torch.empty
    dk = alloc((B, Nk, H, DQK), device=q.device, dtype=kv_dtype)
    dv = alloc((B, Nk, H, DV), device=q.device, dtype=kv_dtype)
    dvec = torch.empty((B, H, Nq), device=q.device, dtype=torch.float32)
    st = (ctypes.c_int64 * 21)(*(_strides3(q) + _strides3(k) + _strides3(v) + _strides3(o) + _strides3(dq) +
                                 _strides3(dk) + _strides3(dv)))
    L.call("dcc_attn_bwd", L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(do), L.ptr(lse), L.ptr(dvec), L.ptr(kbias),
           L.ptr(q) if residual else None, int(res_start), L.ptr(dq), L.ptr(dk), L.ptr(dv), st, B, H, Nq, Nk, DQK, DV,
           float(scale), int(qsplit), L.dt(q), L.stream())
    return dq, dk, dv
