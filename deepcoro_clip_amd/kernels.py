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
