"""Thin, autograd-free Python wrappers over the C-ABI kernels (shape checks + launch).

Every function here calls into `libdcc_hip.so` (include/dcc.h); there is no torch fallback.
torch is used for device memory (allocation, views) and the current HIP stream only.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from . import _lib as L

GEMM_BIAS, GEMM_GELU, GEMM_SAVE_PRE, GEMM_RESID, GEMM_DGELU, GEMM_ACCUM, GEMM_COLSUM_A = 1, 2, 4, 8, 16, 32, 64
import os as _os

GEMM_VARIANT = int(_os.environ.get("DCC_GEMM_VARIANT", "0")) << 8  # tuning knob for A/B runs (see csrc/gemm.hip)


class KernelTimer:
    """HIP-event timing of individual kernel families on the launch stream (bench.py roofline leg).

    Events are recorded on torch's current stream, which is the stream every dcc_* launch uses."""

    def __init__(self):
        self.records: dict[str, list] = {}

    def time(self, family: str, units: float, label: str = "", nbytes: float = 0.0):
        return _TimerCtx(self, family, units, label, nbytes)

    def by_label(self) -> list:
        """[(family, label, launches, ms, units)] sorted by time (call after summary(), which synchronises)."""
        agg: dict = {}
        for fam, recs in self.records.items():
            for a, b, u, lab, _nb in recs:
                k = (fam, lab)
                c = agg.setdefault(k, [0, 0.0, 0.0])
                c[0] += 1
                c[1] += a.elapsed_time(b)
                c[2] += u
        return sorted(((f, l, c[0], c[1], c[2]) for (f, l), c in agg.items()), key=lambda r: -r[3])

    def summary(self) -> dict:
        torch.cuda.synchronize()
        out = {}
        for fam, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            units = sum(r[2] for r in recs)
            out[fam] = {"launches": len(recs), "ms": ms, "units": units, "bytes": sum(r[4] for r in recs)}
        return out


class _TimerCtx:
    def __init__(self, timer, family, units, label="", nbytes=0.0):
        self.t, self.f, self.u, self.l, self.nb = timer, family, units, label, nbytes

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.t.records.setdefault(self.f, []).append((self.a, self.b, self.u, self.l, self.nb))


TIMER: KernelTimer | None = None  # set by bench.py for the instrumented pass only


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOCTX = _NoCtx()


def _timed(family: str, units: float, label=None, nbytes: float = 0.0):
    """`label` may be a callable: the launch path must not pay for formatting a string nobody reads (~3 us x 500 GEMMs)."""
    if TIMER is None:
        return _NOCTX
    return TIMER.time(family, units, label() if callable(label) else (label or ""), nbytes)


def _i64(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])


def _i32(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


# ----------------------------------------------------------------------------------------- GEMM
def gemm(a: torch.Tensor, b: torch.Tensor, *, a_km: bool = False, b_km: bool = False, bias: torch.Tensor | None = None,
         resid: torch.Tensor | None = None, aux: torch.Tensor | None = None, gelu: bool = False,
         save_pre: bool = False, dgelu: bool = False, out: torch.Tensor | None = None, accumulate: bool = False,
         out_dtype: torch.dtype | None = None, splitk: int = 1, alpha: float = 1.0,
         colsum: torch.Tensor | None = None, stream: int | None = None) -> torch.Tensor:
    """C[m,n] = alpha * sum_k A(m,k) B(n,k) with fused epilogue.  `stream`: raw hipStream_t to launch on (default: torch's
    current stream).  `colsum` (fp32 [M], accumulated into; a_km only)
    receives sum_k A(m,k) — the bias gradient of a Linear layer for free inside its weight-gradient GEMM.

    a: [M,K] (or [K,M] if a_km); b: [N,K] (or [K,N] if b_km); 2-D, inner stride 1.
    """
    L.require_cuda(a, b, bias, resid, aux, out)
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
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
    if colsum is not None:
        assert a_km and colsum.dtype == torch.float32 and colsum.numel() == M
        flags |= GEMM_COLSUM_A
    flags |= GEMM_VARIANT
    # algorithmic HBM bytes: every operand once, the output once (read-modify-write when accumulating)
    osz = out.element_size()
    nbytes = (M * K + N * K) * a.element_size() + M * N * osz * (2 if accumulate else 1)
    nbytes += M * N * osz * (int(resid is not None) + int(save_pre)) + M * N * a.element_size() * int(dgelu)
    with _timed("gemm_bf16" if a.dtype == torch.bfloat16 else "gemm_f32", 2.0 * M * N * K,
                lambda: f"M{M} N{N} K{K} {'T' if a_km else 'N'}{'T' if b_km else 'N'} sk{splitk} f{flags} out{out.dtype}", nbytes):
        L.call("dcc_gemm", L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(bias), L.ptr(resid), L.ptr(aux), L.ptr(colsum),
               a.stride(0), b.stride(0), out.stride(0), resid.stride(0) if resid is not None else 0,
               aux.stride(0) if aux is not None else 0, M, N, K, 1, 0, 0, 0, splitk, float(alpha), flags, int(a_km),
               int(b_km), L.dt(a), L.dt(out), L.stream() if stream is None else stream)
    return out


STREAM_GEMM = int(_os.environ.get("DCC_STREAM_GEMM", "1"))   # 0: off; 1: the shapes it wins (K <= 192, N >= 192); 2: whole domain (tests, A/B); 3: plain-bias wide outputs only


def stream_gemm_shape(M: int, N: int, K: int, dtype: torch.dtype) -> bool:
    """Shapes dcc_gemm hands to the streaming panel GEMM (csrc/gemm_stream.hip): bf16 row-major x row-major with the weight
    resident in LDS — the early MViT blocks' products.  Mirrors the C-side test so that a data-gradient product g W (k-major
    weight) can be handed over with a transposed weight shadow instead (functional._dx_gemm): since the kernel stores complete
    lines through LDS it wins K <= 192 with N >= 192, i.e. the gelu' products of blocks 0-2 (profiles/r03_stream_bench_staged.txt)."""
    whole = STREAM_GEMM == 2 or (GEMM_VARIANT >> 8) == 12   # variant 12: tests / tools force the kernel's whole domain
    if STREAM_GEMM == 0 and not whole:
        return False
    if dtype != torch.bfloat16 or M < 16384 or K not in (96, 192, 288, 384, 576) or N % 32 or (GEMM_VARIANT >> 8) not in (0, 12):
        return False
    if not (K <= 192 or N <= 192):
        return False
    if not whole and (STREAM_GEMM == 3 or not (K <= 192 and N >= 192)):   # mode 3: round 3's first rule never took data gradients
        return False
    nc = ((150 * 1024 - 9216) // (2 * K + 16 + 4)) & ~31
    return nc >= 32 and -(-N // nc) <= 4


def cast_transpose(w: torch.Tensor) -> torch.Tensor:
    """fp32 [R, C] (row stride free) -> contiguous bf16 [C, R]."""
    L.require_cuda(w)
    assert w.dim() == 2 and w.dtype == torch.float32 and w.stride(1) == 1
    R, C = w.shape
    out = torch.empty((C, R), device=w.device, dtype=torch.bfloat16)
    L.call("dcc_cast_transpose", L.ptr(w), w.stride(0), L.ptr(out), R, R, C, L.stream())
    return out


def gemm_batched(a, b, out, *, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, bias=None, a_km=False, b_km=False,
                 alpha: float = 1.0, accumulate: bool = False, splitk: int = 1):
    """Raw batched form (base tensors + explicit strides); `a`/`b`/`out` only provide base pointers and dtypes."""
    L.require_cuda(a, b, out, bias)
    flags = (GEMM_BIAS if bias is not None else 0) | (GEMM_ACCUM if accumulate else 0) | GEMM_VARIANT
    nbytes = batch * ((M * K + N * K) * a.element_size() + M * N * out.element_size() * (2 if accumulate else 1))
    with _timed("gemm_bf16" if a.dtype == torch.bfloat16 else "gemm_f32", 2.0 * M * N * K * batch,
                lambda: f"batched{batch} M{M} N{N} K{K} {'T' if a_km else 'N'}{'T' if b_km else 'N'} sk{splitk}", nbytes):
        L.call("dcc_gemm", L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(bias), None, None, None, lda, ldb, ldc, 0, 0, M, N, K, batch,
               sA, sB, sC, splitk, float(alpha), flags, int(a_km), int(b_km), L.dt(a), L.dt(out), L.stream())
    return out


def wgrad_splitk(rows: int, n: int, k: int) -> int:
    """Split factor for weight-gradient GEMMs (reduction over `rows`, output n x k).

    Measured on MI355X (round 1 sweeps; tools/gemm_bench.py today) with the persistent ring kernel: one work item (128x128 output tile x
    K split) per CU is the sweet spot — fewer leave CUs idle, more only add fp32 atomics (served at the memory side).
    So: the largest split with tiles x split <= 256 CUs; outputs with more tiles than half the CUs pick the split
    (<= 4) that wastes least of the last round of items."""
    tiles = ((n + 127) // 128) * ((k + 127) // 128)
    ksteps = (rows + 63) // 64
    if tiles > 128:
        want = min(range(1, 5), key=lambda s: -(-tiles * s // 256) / s + 0.02 * s) if ksteps >= 64 else 1
    else:
        want = max(1, 256 // tiles)
    return int(max(1, min(want, ksteps // 4, 128)))


# ------------------------------------------------------------------------------------ attention
def _strides3(t: torch.Tensor) -> list[int]:
    assert t.dim() == 4 and t.stride(3) == 1, "attention operands are [B, N, H, D] views with unit inner stride"
    return [t.stride(0), t.stride(1), t.stride(2)]


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, scale: float, kbias: torch.Tensor | None = None,
             residual: bool = False, res_start: int = 0, drop_p: float = 0.0, drop_seed: int = 0,
             drop_step: torch.Tensor | None = None):
    """O = dropout(softmax(scale * q k^T + kbias)) v (+ q[..., :DV] for rows >= res_start). Returns (O [B,Nq,H,DV],
    LSE [B,H,Nq]).  drop_p: attention-probability dropout (mask = hash of (drop_seed, *drop_step, element))."""
    L.require_cuda(q, k, v, kbias, drop_step)
    B, Nq, H, DQK = q.shape
    Nk, DV = k.shape[1], v.shape[3]
    assert k.shape == (B, Nk, H, DQK) and v.shape == (B, Nk, H, DV) and q.dtype == k.dtype == v.dtype
    o = torch.empty((B, Nq, H, DV), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, H, Nq), device=q.device, dtype=torch.float32)
    if kbias is not None:
        assert kbias.shape == (B, Nk) and kbias.dtype == torch.float32 and kbias.is_contiguous()
    st = _i64(_strides3(q) + _strides3(k) + _strides3(v) + _strides3(o))
    with _timed("attn_fwd", 2.0 * B * H * Nq * Nk * (DQK + DV), lambda: f"B{B} H{H} Nq{Nq} Nk{Nk} D{DQK}/{DV}"):
        L.call("dcc_attn_fwd", L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(lse), L.ptr(kbias),
               L.ptr(q) if residual else None, int(res_start), st, B, H, Nq, Nk, DQK, DV, float(scale), float(drop_p),
               ctypes.c_uint64(drop_seed & (2**64 - 1)), L.ptr(drop_step), L.dt(q), L.stream())
    return o, lse


def attn_bwd(q, k, v, o, do, lse, *, scale: float, kbias=None, residual: bool = False, res_start: int = 0,
             qsplit: int | None = None, drop_p: float = 0.0, drop_seed: int = 0, drop_step: torch.Tensor | None = None):
    """Gradients of attn_fwd w.r.t. q (including the residual shortcut's dO when `residual`), k, v.

    dk/dv come back in fp32 when the query range is split (atomic accumulation)."""
    L.require_cuda(q, k, v, o, do, lse)
    B, Nq, H, DQK = q.shape
    Nk, DV = k.shape[1], v.shape[3]
    assert do.shape == o.shape and do.stride() == o.stride(), "dO must share O's layout"
    if qsplit is None:
        # one block per CU (register-bound kernel): split the query range only as far as needed to fill 256 CUs
        kblocks = ((Nk + 127) // 128) * H * B
        qsplit = 1 if kblocks >= 192 else max(1, min((Nq + 127) // 128, (512 + kblocks - 1) // kblocks))
    dq = torch.empty((B, Nq, H, DQK), device=q.device, dtype=q.dtype)
    kv_dtype = torch.float32 if qsplit > 1 else q.dtype
    alloc = torch.zeros if qsplit > 1 else torch.empty
    dk = alloc((B, Nk, H, DQK), device=q.device, dtype=kv_dtype)
    dv = alloc((B, Nk, H, DV), device=q.device, dtype=kv_dtype)
    dvec = torch.empty((B, H, Nq), device=q.device, dtype=torch.float32)
    st = _i64(_strides3(q) + _strides3(k) + _strides3(v) + _strides3(o) + _strides3(dq) + _strides3(dk) + _strides3(dv))
    # dQ kernel: S, dP, dQ products; dK/dV kernel: S, dP, dV, dK products
    with _timed("attn_bwd", 2.0 * B * H * Nq * Nk * (3 * DQK + 4 * DV), lambda: f"B{B} H{H} Nq{Nq} Nk{Nk} D{DQK}/{DV} qs{qsplit}"):
        L.call("dcc_attn_bwd", L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(do), L.ptr(lse), L.ptr(dvec),
               L.ptr(kbias), L.ptr(q) if residual else None, int(res_start), L.ptr(dq), L.ptr(dk), L.ptr(dv), st, B, H,
               Nq, Nk, DQK, DV, float(scale), int(qsplit), float(drop_p), ctypes.c_uint64(drop_seed & (2**64 - 1)),
               L.ptr(drop_step), L.dt(q), L.stream())
    return dq, dk, dv


# ------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x2d: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out: torch.Tensor | None = None):
    """x2d: [rows, C] (row stride free). Returns (y, mean, rstd)."""
    L.require_cuda(x2d, gamma, beta, out)
    rows, C = x2d.shape
    assert x2d.stride(1) == 1 and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    if out is None:
        out = torch.empty((rows, C), device=x2d.device, dtype=x2d.dtype)
    assert out.shape == (rows, C) and out.stride(1) == 1 and out.dtype == x2d.dtype
    mean = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    L.call("dcc_layernorm_fwd", L.ptr(x2d), x2d.stride(0), L.ptr(out), out.stride(0), L.ptr(gamma), L.ptr(beta),
           L.ptr(mean), L.ptr(rstd), rows, C, float(eps), L.dt(x2d), L.stream())
    return out, mean, rstd


# --------------------------------------------------------------------- fused LayerNorm + MLP (csrc/mlp_fused.hip)
def mlp_fused_channels(C: int) -> bool:
    return L.lib().dcc_mlp_packed_bytes(int(C)) > 0


def mlp_pack(w1c: torch.Tensor, w2c: torch.Tensor) -> torch.Tensor:
    """bf16 fc1 weight [4C, C] and fc2 weight [C, 4C] -> the fused kernel's stream image (MFMA-fragment order)."""
    L.require_cuda(w1c, w2c)
    C = w1c.shape[1]
    assert w1c.dtype == torch.bfloat16 and w2c.dtype == torch.bfloat16 and w1c.is_contiguous() and w2c.is_contiguous()
    assert w1c.shape == (4 * C, C) and w2c.shape == (C, 4 * C)
    nbytes = L.lib().dcc_mlp_packed_bytes(C)
    if nbytes <= 0:
        raise L.DccError(f"dcc_mlp_pack_weights: C = {C} is outside the fused MLP kernel's domain")
    out = torch.empty(nbytes // 2, device=w1c.device, dtype=torch.bfloat16)
    L.call("dcc_mlp_pack_weights", L.ptr(w1c), L.ptr(w2c), L.ptr(out), C, L.stream())
    return out


def ln_mlp_fwd(x2d: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, packed: torch.Tensor, b1: torch.Tensor,
               b2: torch.Tensor, residual: bool = True, save: bool = True, save_act: bool = True):
    """y = [x +] fc2(gelu(fc1(LayerNorm(x)))) in one launch.  Returns (y, xn, mean, rstd, pre, act); the last five are None
    without `save` (inference), `act` is None without `save_act`."""
    L.require_cuda(x2d, gamma, beta, packed, b1, b2)
    M, C = x2d.shape
    assert x2d.dtype == torch.bfloat16 and x2d.stride(1) == 1
    dev = x2d.device
    y = torch.empty((M, C), device=dev, dtype=torch.bfloat16)
    xn = mean = rstd = pre = act = None
    if save:
        xn = torch.empty((M, C), device=dev, dtype=torch.bfloat16)
        mean = torch.empty(M, device=dev, dtype=torch.float32)
        rstd = torch.empty(M, device=dev, dtype=torch.float32)
        pre = torch.empty((M, 4 * C), device=dev, dtype=torch.bfloat16)
        if save_act:
            act = torch.empty((M, 4 * C), device=dev, dtype=torch.bfloat16)
    nbytes = 2.0 * M * C * (2 + (1 if save else 0) + (4 if save else 0) + (4 if act is not None else 0)) + packed.numel() * 2.0
    with _timed("gemm_bf16", 16.0 * M * C * C, lambda: f"ln_mlp M{M} C{C}", nbytes):
        L.call("dcc_ln_mlp_fwd", L.ptr(x2d), x2d.stride(0), L.ptr(gamma), L.ptr(beta), float(eps), L.ptr(packed), L.ptr(b1),
               L.ptr(b2), L.ptr(y), y.stride(0), L.ptr(xn), L.ptr(mean), L.ptr(rstd), L.ptr(pre), L.ptr(act), M, C,
               int(bool(residual)), L.stream())
    return y, xn, mean, rstd, pre, act


_LN_WS: dict = {}


def layernorm_bwd(dy2d, x2d, mean, rstd, gamma, dgamma, dbeta, dx: torch.Tensor | None = None,
                  dres: torch.Tensor | None = None):
    """dgamma/dbeta (fp32) are accumulated into. dy2d may be fp32 while x is bf16; dx takes dy's dtype.
    dres: gradient of x's other use (residual branch), added into dx in the same pass."""
    L.require_cuda(dy2d, x2d, dx, dres)
    rows, C = x2d.shape
    assert dy2d.shape == (rows, C) and dy2d.stride(1) == 1 and x2d.stride(1) == 1
    if dx is None:
        dx = torch.empty((rows, C), device=x2d.device, dtype=dy2d.dtype)
    assert dx.dtype == dy2d.dtype and dx.stride(1) == 1
    assert dres is None or (dres.shape == (rows, C) and dres.dtype == dx.dtype and dres.stride() == dx.stride())
    stream = L.stream()
    key = (dy2d.device.index, stream)  # block partials of dgamma/dbeta; one per stream (the branches overlap)
    ws = _LN_WS.get(key)
    if ws is None:
        ws = _LN_WS[key] = torch.empty(1 << 20, device=dy2d.device, dtype=torch.float32)  # DCC_LN_WS_FLOATS
    L.call("dcc_layernorm_bwd", L.ptr(dy2d), dy2d.stride(0), L.ptr(x2d), x2d.stride(0), L.ptr(mean), L.ptr(rstd),
           L.ptr(gamma), L.ptr(dx), dx.stride(0), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dres), L.ptr(ws), rows, C,
           L.dt(x2d), L.dt(dy2d), stream)
    return dx


# ------------------------------------------------------------------------------- pooling / conv
@dataclass(frozen=True)
class PoolGeom:
    B: int
    H: int
    C: int
    thw: tuple[int, int, int]
    kernel: tuple[int, int, int]
    stride: tuple[int, int, int]
    padding: tuple[int, int, int]
    has_cls: bool = True

    @property
    def out_thw(self) -> tuple[int, int, int]:
        return tuple((s + 2 * p - k) // st + 1 for s, p, k, st in zip(self.thw, self.padding, self.kernel, self.stride))

    @property
    def n_in(self) -> int:
        t, h, w = self.thw
        return t * h * w + int(self.has_cls)

    @property
    def n_out(self) -> int:
        t, h, w = self.out_thw
        return t * h * w + int(self.has_cls)

    def dims(self):
        return _i32([self.B, self.H, self.C, *self.thw, *self.kernel, *self.stride, *self.padding, int(self.has_cls)])


def _pool_strides(x: torch.Tensor, y: torch.Tensor):
    return _i64(_strides3(x) + _strides3(y))


def pool_fwd(x: torch.Tensor, g: PoolGeom, weight: torch.Tensor | None, out: torch.Tensor | None = None,
             want_argmax: bool = False):
    """x: [B, N_in, H, C] view. weight: fp32 [C, taps] for the depthwise conv, None for max pooling."""
    L.require_cuda(x, weight, out)
    assert x.shape == (g.B, g.n_in, g.H, g.C), (x.shape, g)
    if out is None:
        out = torch.empty((g.B, g.n_out, g.H, g.C), device=x.device, dtype=x.dtype)
    mode = 0 if weight is not None else 1
    argmax = None
    if mode == 1 and want_argmax:
        argmax = torch.empty((g.B, g.n_out, g.H, g.C), device=x.device, dtype=torch.uint8)
    if weight is not None:
        assert weight.dtype == torch.float32 and weight.is_contiguous() and weight.numel() == g.C * g.kernel[0] * g.kernel[1] * g.kernel[2]
    L.call("dcc_pool_fwd", L.ptr(x), L.ptr(out), L.ptr(weight), L.ptr(argmax), g.dims(), _pool_strides(x, out), mode,
           L.dt(x), L.stream())
    return out, argmax


def pool_ln_supported(x: torch.Tensor, g: PoolGeom) -> bool:
    """Domain of the fused conv-pool + LayerNorm kernel (dcc_pool_ln_fwd)."""
    return (x.dtype == torch.bfloat16 and g.C == 96 and g.kernel == (3, 3, 3) and g.padding == (1, 1, 1)
            and g.stride[0] == 1 and g.stride[1] == g.stride[2] and g.stride[1] in (1, 2, 4, 8))


def pool_ln_fwd(x: torch.Tensor, g: PoolGeom, weight: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                out: torch.Tensor, aug: int = 0):
    """Depthwise-conv pool of x ([B, N_in, H, 96] view) + LayerNorm over the head dimension in one launch.
    out: [B*N_out*H, 96] rows (row stride free, e.g. the first 96 columns of the [.., DQK] attention operand).
    aug = 32 / 64: `out` is the first 96 columns of the key operand [.., 96 + aug]; the one-hot rel-pos key augmentation is written
    behind it (replaces relpos_kaug).  Returns (conv [B, N_out, H, 96], mean, rstd), or None outside the kernel's domain."""
    L.require_cuda(x, weight, gamma, beta, out)
    assert x.shape == (g.B, g.n_in, g.H, g.C) and pool_ln_supported(x, g)
    rows = g.B * g.n_out * g.H
    assert out.shape == (rows, g.C) and out.stride(1) == 1 and out.dtype == x.dtype
    assert weight.dtype == torch.float32 and weight.is_contiguous() and weight.numel() == g.C * 27
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    conv = torch.empty((g.B, g.n_out, g.H, g.C), device=x.device, dtype=x.dtype)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    if not L.try_call("dcc_pool_ln_fwd", L.ptr(x), L.ptr(conv), L.ptr(out), out.stride(0), L.ptr(weight), L.ptr(gamma), L.ptr(beta),
                      L.ptr(mean), L.ptr(rstd), g.dims(), _pool_strides(x, conv), float(eps), int(aug), L.dt(x), L.stream()):
        return None  # outside the fused kernel's index range (very large batches): the caller runs conv + LayerNorm
    return conv, mean, rstd


def pool_ln_fwd_pair(xs, g: PoolGeom, weights, gammas, betas, eps: float, outs, augs=(0, 0)):
    """Two fused pools of identical geometry and input strides (a block's k and v branches) in one launch.
    Returns ((conv, mean, rstd), (conv, mean, rstd)) or None outside the kernel's domain."""
    L.require_cuda(*xs, *weights, *gammas, *betas, *outs)
    assert xs[0].stride() == xs[1].stride() and xs[0].shape == xs[1].shape == (g.B, g.n_in, g.H, g.C)
    rows = g.B * g.n_out * g.H
    res, ptrs = [], []
    for x, w, ga, be, out in zip(xs, weights, gammas, betas, outs):
        assert pool_ln_supported(x, g) and out.shape == (rows, g.C) and out.stride(1) == 1 and out.dtype == x.dtype
        assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == g.C * 27
        conv = torch.empty((g.B, g.n_out, g.H, g.C), device=x.device, dtype=x.dtype)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        res.append((conv, mean, rstd))
        ptrs.append((ctypes.c_void_p * 8)(*[L.ptr(t) for t in (x, conv, out, w, ga, be, mean, rstd)]))
    ok = L.try_call("dcc_pool_ln_fwd2", ptrs[0], ptrs[1], outs[0].stride(0), outs[1].stride(0), int(augs[0]), int(augs[1]),
                    g.dims(), _pool_strides(xs[0], res[0][0]), float(eps), L.dt(xs[0]), L.stream())
    return tuple(res) if ok else None


def pool_bwd_data(dy: torch.Tensor, dx: torch.Tensor, g: PoolGeom, weight: torch.Tensor | None,
                  argmax: torch.Tensor | None = None):
    """dx (a [B,N_in,H,C] view, e.g. a slice of the dqkv buffer) is overwritten."""
    L.require_cuda(dy, dx, weight, argmax)
    assert dy.shape == (g.B, g.n_out, g.H, g.C) and dx.shape == (g.B, g.n_in, g.H, g.C)
    mode = 0 if weight is not None else 1
    L.call("dcc_pool_bwd_data", L.ptr(dy), L.ptr(dx), L.ptr(weight), L.ptr(argmax), g.dims(), _pool_strides(dx, dy),
           mode, L.dt(dx), L.dt(dy), L.stream())
    return dx




def dwconv_bwd_weight(dy: torch.Tensor, x: torch.Tensor, g: PoolGeom, dw: torch.Tensor):
    """dw fp32 [C, taps] accumulated into: every block folds its threads in LDS and adds its partial with contiguous float
    atomics (one launch; r02 wrote per-block slabs to a scratch buffer and folded them with a second kernel)."""
    L.require_cuda(dy, x, dw)
    assert dw.dtype == torch.float32 and dw.is_contiguous()
    L.call("dcc_dwconv_bwd_weight", L.ptr(dy), L.ptr(x), L.ptr(dw), None, g.dims(), _pool_strides(x, dy), L.dt(x),
           L.dt(dy), L.stream())
    return dw


def im2col3d(video: torch.Tensor, kernel, stride, padding, kp: int, dtype: torch.dtype, norm=None) -> torch.Tensor:
    """video: [B, T, H, W, C] contiguous, fp32 (already normalised) or uint8 with norm = (mean[C], std[C]) fp32 device
    tensors in pixel units -> [B*To*Ho*Wo, kp] in `dtype`, column order (kt,kh,kw,c)."""
    L.require_cuda(video)
    assert video.is_contiguous() and video.dim() == 5
    B, T, H, W, C = video.shape
    To, Ho, Wo = [(s + 2 * p - k) // st + 1 for s, p, k, st in zip((T, H, W), padding, kernel, stride)]
    col = torch.empty((B * To * Ho * Wo, kp), device=video.device, dtype=dtype)
    if video.dtype == torch.uint8:
        if norm is None:
            raise L.DccError("uint8 frames need the per-channel (mean, std) of the input normalisation")
        mean, std = norm
        L.require_cuda(mean, std)
        assert mean.dtype == std.dtype == torch.float32 and mean.numel() == C and std.numel() == C
        L.call("dcc_im2col3d_u8", L.ptr(video), L.ptr(mean), L.ptr(std), L.ptr(col), B, T, H, W, C, *kernel, *stride, *padding,
               kp, L.dt(dtype), L.stream())
        return col
    assert video.dtype == torch.float32
    L.call("dcc_im2col3d", L.ptr(video), L.ptr(col), B, T, H, W, C, *kernel, *stride, *padding, kp, L.dt(dtype),
           L.stream())
    return col


# ------------------------------------------------------------------------------------- rel-pos
def relpos_dims(B, H, D, A, L, q_thw, k_thw, table_lens, has_cls=True):
    return _i32([B, H, D, A, L, *q_thw, *k_thw, *table_lens, int(has_cls)])


def relpos_gather(Qf, P, idx, dims, scale):
    L.require_cuda(Qf, P)
    assert P.is_contiguous() and P.dtype == Qf.dtype
    L.call("dcc_relpos_gather", L.ptr(Qf), L.ptr(P), L.ptr(idx[0]), L.ptr(idx[1]), L.ptr(idx[2]), dims,
           _i64(_strides3(Qf)), float(scale), L.dt(Qf), L.stream())


def relpos_scatter(dQf, dP, inv, dims, scale):
    L.require_cuda(dQf, dP)
    assert dP.is_contiguous() and dP.dtype == dQf.dtype
    L.call("dcc_relpos_scatter", L.ptr(dQf), L.ptr(dP), L.ptr(inv[0]), L.ptr(inv[1]), L.ptr(inv[2]), dims,
           _i64(_strides3(dQf)), float(scale), L.dt(dQf), L.stream())


def relpos_aug_fwd_mfma(Qf, Rcat, idx, dims, scale) -> bool:
    """Qf[..., D:] from Qf[..., :D] on the matrix cores, tile-local P in LDS (any augmentation width); False when unsupported."""
    L.require_cuda(Qf, Rcat)
    assert Rcat.is_contiguous() and Rcat.dtype == Qf.dtype
    return L.try_call("dcc_relpos_aug_fwd_mfma", L.ptr(Qf), L.ptr(Rcat), L.ptr(idx[0]), L.ptr(idx[1]), L.ptr(idx[2]), dims,
                      _i64(_strides3(Qf)), float(scale), L.dt(Qf), L.stream())


_RELPOS_WS: dict = {}
_RELPOS_MAX_BLOCKS = 768


def relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, scale, dRcat, ln=None):
    """dq [B, Nq, H, D] = dQf[..., :D] + table path of dQf[..., D:]; dRcat [L, D] fp32 accumulated into.  None when the shape
    is outside the fused kernel's domain (the caller then runs scatter + the two GEMMs).

    ln = (conv [B, Nq, H, D], mean, rstd, gamma, dgamma, dbeta): the LayerNorm backward autograd runs next on dq (Pool.norm_act
    of the pooled queries) happens in the kernel's epilogue and the returned tensor is dconv; dgamma / dbeta accumulated into."""
    L.require_cuda(dQf, Qf, Rcat, dRcat)
    assert Rcat.is_contiguous() and dQf.stride() == Qf.stride() and dRcat.dtype == torch.float32 and dRcat.is_contiguous()
    B, Nq, H, _ = dQf.shape
    D = Rcat.shape[1]
    dq = torch.empty((B, Nq, H, D), device=dQf.device, dtype=dQf.dtype)
    n = _RELPOS_MAX_BLOCKS * (Rcat.shape[0] + 2) * D
    key = (dQf.device.index, n, L.stream())
    ws = _RELPOS_WS.get(key)
    if ws is None:
        ws = _RELPOS_WS[key] = torch.empty(n, device=dQf.device, dtype=torch.float32)
    if ln is None:
        ok = L.try_call("dcc_relpos_aug_bwd", L.ptr(dQf), L.ptr(Qf), L.ptr(dq), L.ptr(Rcat), L.ptr(dRcat), L.ptr(ws),
                        _RELPOS_MAX_BLOCKS, L.ptr(idx[0]), L.ptr(idx[1]), L.ptr(idx[2]), dims, _i64(_strides3(dQf)), float(scale),
                        L.dt(dQf), L.stream())
    else:
        conv, mean, rstd, gamma, dgamma, dbeta = ln
        L.require_cuda(conv, mean, rstd, gamma, dgamma, dbeta)
        rows = B * Nq * H
        assert conv.is_contiguous() and conv.numel() == rows * D and conv.dtype == dQf.dtype
        assert mean.numel() == rows and rstd.numel() == rows and mean.dtype == rstd.dtype == torch.float32
        assert all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == D for t in (gamma, dgamma, dbeta))
        ok = L.try_call("dcc_relpos_aug_bwd_ln", L.ptr(dQf), L.ptr(Qf), L.ptr(dq), L.ptr(Rcat), L.ptr(dRcat), L.ptr(ws),
                        _RELPOS_MAX_BLOCKS, L.ptr(idx[0]), L.ptr(idx[1]), L.ptr(idx[2]), dims, _i64(_strides3(dQf)), float(scale),
                        L.ptr(conv), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dgamma), L.ptr(dbeta), L.dt(dQf), L.stream())
    return dq if ok else None


def relpos_kaug(Kf, dims):
    L.require_cuda(Kf)
    L.call("dcc_relpos_kaug", L.ptr(Kf), dims, _i64(_strides3(Kf)), L.dt(Kf), L.stream())


# ---------------------------------------------------------------------------------- elementwise
def cast(src: torch.Tensor, dtype: torch.dtype, out: torch.Tensor | None = None) -> torch.Tensor:
    L.require_cuda(src, out)
    assert src.is_contiguous()
    if out is None:
        out = torch.empty(src.shape, device=src.device, dtype=dtype)
    assert out.is_contiguous() and out.numel() == src.numel()
    if src.numel():
        L.call("dcc_cast", L.ptr(src), L.ptr(out), src.numel(), L.dt(src), L.dt(out), L.stream())
    return out


def cast2d(src: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """Strided 2-D copy/cast: both [rows, cols] with unit inner stride."""
    L.require_cuda(src, out)
    assert src.dim() == 2 and out.shape == src.shape and src.stride(1) == 1 and out.stride(1) == 1
    L.call("dcc_cast2d", L.ptr(src), src.stride(0), L.ptr(out), out.stride(0), src.shape[0], src.shape[1], L.dt(src),
           L.dt(out), L.stream())
    return out


def axpby(x: torch.Tensor, y: torch.Tensor | None = None, a: float = 1.0, b: float = 1.0,
          out: torch.Tensor | None = None) -> torch.Tensor:
    L.require_cuda(x, y, out)
    assert x.is_contiguous() and (y is None or (y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype))
    if out is None:
        out = torch.empty_like(x)
    L.call("dcc_axpby", L.ptr(out), float(a), L.ptr(x), float(b), L.ptr(y), x.numel(), L.dt(x), L.stream())
    return out


def resize_frames_u8(frames: torch.Tensor, size: tuple[int, int], frame_idx: torch.Tensor | None = None, n_out: int | None = None,
                     mean: torch.Tensor | None = None, std: torch.Tensor | None = None, antialias: bool = True,
                     out_uint8: bool = False) -> torch.Tensor:
    """uint8 frames [T_in, H_in, W_in, C_in] (C_in 1 or 3) -> [T_out, size[0], size[1], 3]: anti-aliased bilinear resize with
    optional frame index select (device int32 [T_out]) and per-channel normalisation (fp32 output), or uint8 output."""
    L.require_cuda(frames, frame_idx, mean, std)
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.is_contiguous() and frames.shape[3] in (1, 3)
    T_in, H_in, W_in, C_in = frames.shape
    T_out = int(frame_idx.numel()) if frame_idx is not None else int(n_out or T_in)
    if frame_idx is not None:
        assert frame_idx.dtype == torch.int32 and frame_idx.is_contiguous()
    if mean is not None:
        assert std is not None and mean.dtype == torch.float32 and std.dtype == torch.float32 and mean.numel() == 3 and std.numel() == 3
    out = torch.empty((T_out, int(size[0]), int(size[1]), 3), device=frames.device, dtype=torch.uint8 if out_uint8 else torch.float32)
    L.call("dcc_resize_frames_u8", L.ptr(frames), L.ptr(frame_idx), L.ptr(out), L.ptr(mean), L.ptr(std), T_in, H_in, W_in, C_in,
           T_out, int(size[0]), int(size[1]), int(antialias), int(out_uint8), L.stream())
    return out


def u8_normalize(frames: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """uint8 [..., C] frames -> fp32 (x - mean[c]) / std[c]."""
    L.require_cuda(frames, mean, std)
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.numel() % 4 == 0
    C = frames.shape[-1]
    assert mean.dtype == std.dtype == torch.float32 and mean.numel() == C and std.numel() == C
    out = torch.empty(frames.shape, device=frames.device, dtype=torch.float32)
    L.call("dcc_u8_normalize", L.ptr(frames), L.ptr(out), frames.numel(), C, L.ptr(mean), L.ptr(std), L.stream())
    return out


def scale_rows(x: torch.Tensor, scale: torch.Tensor, samples: int, resid: torch.Tensor | None = None) -> torch.Tensor:
    """out = resid + x * scale[sample]; x is [samples * rows, C] with equally many rows per sample."""
    L.require_cuda(x, scale, resid)
    assert x.is_contiguous() and scale.dtype == torch.float32 and scale.numel() == samples and x.numel() % samples == 0
    assert resid is None or (resid.is_contiguous() and resid.shape == x.shape and resid.dtype == x.dtype)
    out = torch.empty_like(x)
    L.call("dcc_scale_rows", L.ptr(x), L.ptr(scale), L.ptr(resid), L.ptr(out), samples, x.numel() // samples, L.dt(x),
           L.stream())
    return out


def dgelu_mul(dy: torch.Tensor, pre: torch.Tensor) -> torch.Tensor:
    L.require_cuda(dy, pre)
    assert dy.is_contiguous() and pre.is_contiguous() and dy.shape == pre.shape and dy.dtype == pre.dtype
    out = torch.empty_like(dy)
    L.call("dcc_dgelu_mul", L.ptr(dy), L.ptr(pre), L.ptr(out), dy.numel(), L.dt(dy), L.stream())
    return out


def colsum(x2d: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """out[n] (+)= sum_m x[m,n]; out fp32 (zero-initialised when not given)."""
    L.require_cuda(x2d, out)
    assert x2d.dim() == 2 and x2d.stride(1) == 1
    M, N = x2d.shape
    if out is None:
        out = torch.zeros(N, device=x2d.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.numel() == N
    L.call("dcc_colsum", L.ptr(x2d), x2d.stride(0), L.ptr(out), M, N, L.dt(x2d), L.stream())
    return out


def broadcast_row(src: torch.Tensor, dst: torch.Tensor, ld: int, rows: int):
    L.require_cuda(src, dst)
    assert src.dtype == torch.float32 and src.is_contiguous()
    L.call("dcc_broadcast_row", L.ptr(src), L.ptr(dst), ld, rows, src.numel(), L.dt(dst), L.stream())


def token_mean(x: torch.Tensor, backward: bool = False, n_tokens: int | None = None) -> torch.Tensor:
    """forward: [B,N,D] -> [B,D]; backward: dy [B,D] -> dx [B,n_tokens,D]."""
    L.require_cuda(x)
    assert x.is_contiguous()
    if not backward:
        B, N, D = x.shape
        out = torch.empty((B, D), device=x.device, dtype=x.dtype)
    else:
        B, D = x.shape
        N = int(n_tokens)
        out = torch.empty((B, N, D), device=x.device, dtype=x.dtype)
    L.call("dcc_token_mean", L.ptr(x), L.ptr(out), B, N, D, int(backward), L.dt(x), L.stream())
    return out


def embed_fwd(ids, word, pos, type_, dtype) -> torch.Tensor:
    L.require_cuda(ids, word, pos, type_)
    B, S = ids.shape
    D = word.shape[1]
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    out = torch.empty((B * S, D), device=ids.device, dtype=dtype)
    L.call("dcc_embed_fwd", L.ptr(ids), L.ptr(word), L.ptr(pos), L.ptr(type_), L.ptr(out), B * S, S, D, word.shape[0],
           L.dt(dtype), L.stream())
    return out


def embed_bwd(ids, dy, dword, dpos, dtype0):
    L.require_cuda(ids, dy, dword, dpos, dtype0)
    B, S = ids.shape
    D = dy.shape[-1]
    vocab = dword.shape[0] if dword is not None else 1 << 30
    L.call("dcc_embed_bwd", L.ptr(ids), L.ptr(dy), L.ptr(dword), L.ptr(dpos), L.ptr(dtype0), B * S, S, D, vocab,
           L.dt(dy), L.stream())


def rope(x: torch.Tensor, y: torch.Tensor, cos_t: torch.Tensor, sin_t: torch.Tensor, backward: bool = False):
    """x, y: [B,N,H,D] views (y may alias x); tables [N,D] fp32."""
    L.require_cuda(x, y, cos_t, sin_t)
    B, N, H, D = x.shape
    assert cos_t.shape == (N, D) and sin_t.shape == (N, D) and cos_t.is_contiguous() and sin_t.is_contiguous()
    L.call("dcc_rope", L.ptr(x), L.ptr(y), L.ptr(cos_t), L.ptr(sin_t), B, N, H, D, _i64(_strides3(x) + _strides3(y)),
           int(backward), L.dt(x), L.stream())
    return y


def qpool_fwd(x: torch.Tensor, q: torch.Tensor):
    L.require_cuda(x, q)
    B, N, D = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and q.numel() == D
    out = torch.empty((B, D), device=x.device, dtype=torch.float32)
    w = torch.empty((B, N), device=x.device, dtype=torch.float32)
    L.call("dcc_qpool_fwd", L.ptr(x), L.ptr(q), L.ptr(out), L.ptr(w), B, N, D, L.stream())
    return out, w


def qpool_bwd(x, q, w, dout):
    L.require_cuda(x, q, w, dout)
    B, N, D = x.shape
    dx = torch.empty_like(x)
    dq = torch.zeros(D, device=x.device, dtype=torch.float32)
    L.call("dcc_qpool_bwd", L.ptr(x), L.ptr(q), L.ptr(w), L.ptr(dout.contiguous()), L.ptr(dx), L.ptr(dq), B, N, D,
           L.stream())
    return dx, dq


def nan_to_num_(x: torch.Tensor, nan: float = 0.0, posinf: float = 0.0, neginf: float = 0.0,
                count: torch.Tensor | None = None) -> torch.Tensor:
    L.require_cuda(x, count)
    assert x.is_contiguous()
    if x.numel():
        L.call("dcc_nan_to_num", L.ptr(x), x.numel(), float(nan), float(posinf), float(neginf), L.ptr(count), L.dt(x),
               L.stream())
    return x


def dropout(x: torch.Tensor, p: float, seed: int, step: torch.Tensor | None = None) -> torch.Tensor:
    """`step`: device int64 counter mixed into the seed (a HIP-graph replay then draws fresh masks when it advanced)."""
    L.require_cuda(x, step)
    assert x.is_contiguous() and (step is None or (step.dtype == torch.int64 and step.numel() == 1))
    out = torch.empty_like(x)
    L.call("dcc_dropout", L.ptr(x), L.ptr(out), x.numel(), float(p), ctypes.c_uint64(seed & (2**64 - 1)), L.ptr(step), L.dt(x),
           L.stream())
    return out


def retrieval_ranks(sim: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """sim fp32 [Nq, Nc] (unit inner stride), gt int32 [Nq, G] (-1 = empty) -> int32 [Nq, G] 1-based ranks (0 = empty)."""
    L.require_cuda(sim, gt)
    assert sim.dim() == 2 and sim.dtype == torch.float32 and sim.stride(1) == 1
    assert gt.dim() == 2 and gt.dtype == torch.int32 and gt.is_contiguous() and gt.shape[0] == sim.shape[0]
    ranks = torch.empty_like(gt)
    L.call("dcc_retrieval_ranks", L.ptr(sim), sim.stride(0), L.ptr(gt), L.ptr(ranks), sim.shape[0], sim.shape[1], gt.shape[1],
           L.stream())
    return ranks


def topk_rows(sim: torch.Tensor, k: int):
    """torch.topk(sim, k, dim=1) for fp32 [rows, M]: (values fp32 [rows, k], indices int64 [rows, k]), ties by column."""
    L.require_cuda(sim)
    assert sim.dim() == 2 and sim.dtype == torch.float32 and sim.stride(1) == 1 and 1 <= k <= sim.shape[1]
    idx = torch.empty((sim.shape[0], k), device=sim.device, dtype=torch.int64)
    val = torch.empty((sim.shape[0], k), device=sim.device, dtype=torch.float32)
    if sim.shape[0]:
        L.call("dcc_topk_rows", L.ptr(sim), sim.stride(0), sim.shape[0], sim.shape[1], int(k), L.ptr(idx), L.ptr(val), L.stream())
    return val, idx


# ----------------------------------------------------------------------------------------- loss
def l2norm_fwd(x: torch.Tensor, eps: float = 1e-12, norm_sum: torch.Tensor | None = None):
    L.require_cuda(x, norm_sum)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    R, D = x.shape
    y = torch.empty((R, D), device=x.device, dtype=torch.float32)
    norms = torch.empty(R, device=x.device, dtype=torch.float32)
    L.call("dcc_l2norm_fwd", L.ptr(x), x.stride(0), L.ptr(y), L.ptr(norms), R, D, float(eps), L.ptr(norm_sum), L.stream())
    return y, norms


def l2norm_bwd(dy, y, norms, eps: float = 1e-12):
    L.require_cuda(dy, y, norms)
    R, D = y.shape
    dx = torch.empty_like(y)
    L.call("dcc_l2norm_bwd", L.ptr(dy), L.ptr(y), L.ptr(norms), L.ptr(dx), D, R, D, float(eps), L.stream())
    return dx


def logits_fwd(sim, mode: int, log_temp, clamp_temp: bool, bias=None, col_margin=None):
    L.require_cuda(sim, log_temp, bias, col_margin)
    B, M = sim.shape
    z = torch.empty_like(sim)
    L.call("dcc_logits_fwd", L.ptr(sim), L.ptr(z), B, M, mode, L.ptr(log_temp), int(clamp_temp), L.ptr(bias),
           L.ptr(col_margin), L.stream())
    return z


def logits_bwd(sim, dz, mode: int, log_temp, clamp_temp: bool, bias, scal):
    L.require_cuda(sim, dz, log_temp, scal)
    B, M = sim.shape
    dsim = torch.empty_like(sim)
    L.call("dcc_logits_bwd", L.ptr(sim), L.ptr(dz), L.ptr(dsim), B, M, mode, L.ptr(log_temp), int(clamp_temp),
           L.ptr(bias), L.ptr(scal), L.stream())
    return dsim


def lse(z, rows: bool = True, cols: bool = True):
    L.require_cuda(z)
    B, M = z.shape
    lr = torch.empty(B, device=z.device, dtype=torch.float32) if rows else None
    lc = torch.empty(M, device=z.device, dtype=torch.float32) if cols else None
    L.call("dcc_lse", L.ptr(z), L.ptr(lr), L.ptr(lc), B, M, L.stream())
    return lr, lc


def ce_diag(z, lr, lc, smooth: float, scal, want_grad: bool = True, sim=None):
    L.require_cuda(z, lr, lc, scal, sim)
    B = z.shape[0]
    assert z.shape == (B, B)
    dz = torch.empty_like(z) if want_grad else None
    L.call("dcc_ce_diag", L.ptr(z), L.ptr(lr), L.ptr(lc), L.ptr(dz), B, float(smooth), L.ptr(scal), L.ptr(sim), L.stream())
    return dz


def rowcol_sums(w):
    L.require_cuda(w)
    B, M = w.shape
    rs = torch.zeros(B, device=w.device, dtype=torch.float32)
    cs = torch.zeros(M, device=w.device, dtype=torch.float32)
    L.call("dcc_rowcol_sums", L.ptr(w), L.ptr(rs), L.ptr(cs), B, M, L.stream())
    return rs, cs


def weighted_ce(z, pos, lr, lc, rs, cs, eps: float, scal, want_grad: bool = True):
    L.require_cuda(z, pos, scal)
    B, M = z.shape
    dz = torch.empty_like(z) if want_grad else None
    L.call("dcc_weighted_ce", L.ptr(z), L.ptr(pos), L.ptr(lr), L.ptr(lc), L.ptr(rs), L.ptr(cs), L.ptr(dz), B, M,
           float(eps), L.ptr(scal), L.stream())
    return dz


def bce(z, y, pw, pos_weight: float, neg_weight: float, thr: float, denom_host: float, denom_dev, scal,
        want_grad: bool = True):
    L.require_cuda(z, y, pw, denom_dev, scal)
    dz = torch.empty_like(z) if want_grad else None
    L.call("dcc_bce", L.ptr(z), L.ptr(y), L.ptr(pw), float(pos_weight), float(neg_weight), float(thr),
           float(denom_host), L.ptr(denom_dev), L.ptr(dz), z.numel(), L.ptr(scal), L.stream())
    return dz


def mp_align(z, sim, targets, weights=None) -> torch.Tensor:
    """Multi-positive alignment sums (see include/dcc.h): fp32[8] on the device."""
    L.require_cuda(z, sim, targets, weights)
    B, M = z.shape
    assert sim.shape == (B, M) and targets.shape == (B, M) and z.is_contiguous() and sim.is_contiguous() and targets.is_contiguous()
    assert weights is None or (weights.shape == (B, M) and weights.is_contiguous() and weights.dtype == torch.float32)
    out = torch.zeros(8, device=z.device, dtype=torch.float32)
    L.call("dcc_mp_align", L.ptr(z), L.ptr(sim), L.ptr(targets), L.ptr(weights), B, M, L.ptr(out), L.stream())
    return out


def mp_pos_weights(targets, weights) -> torch.Tensor:
    """targets * weights, or targets when every weight is zero (decided on the device)."""
    L.require_cuda(targets, weights)
    assert targets.shape == weights.shape and targets.is_contiguous() and weights.is_contiguous()
    out = torch.empty_like(targets)
    nz = torch.zeros(1, device=targets.device, dtype=torch.float32)
    L.call("dcc_mp_pos_weights", L.ptr(targets), L.ptr(weights), L.ptr(out), targets.numel(), L.ptr(nz), L.stream())
    return out


def mul_clamp0(a: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
    L.require_cuda(a, b)
    assert a.dtype == torch.float32 and a.is_contiguous() and (b is None or (b.shape == a.shape and b.is_contiguous() and b.dtype == torch.float32))
    out = torch.empty_like(a)
    L.call("dcc_mul_clamp0", L.ptr(a), L.ptr(b), L.ptr(out), a.numel(), L.stream())
    return out


def auto_balance_weights(counts: torch.Tensor) -> torch.Tensor:
    """[B, T] matrix whose row i holds max((T - n_i) / n_i, 1), n_i = max(sum_j counts[i, j], 1)."""
    L.require_cuda(counts)
    assert counts.dim() == 2 and counts.dtype == torch.float32 and counts.is_contiguous()
    out = torch.empty_like(counts)
    L.call("dcc_auto_balance_weights", L.ptr(counts), L.ptr(out), counts.shape[0], counts.shape[1], L.stream())
    return out


_ENT_INIT: dict = {}


def entropy_reg(z: torch.Tensor, dz: torch.Tensor | None, weight: float, thr: float, scal: torch.Tensor) -> torch.Tensor:
    """scal[0] += weight * relu(thr - mean row entropy of softmax(z)); dz += gradient.  Returns ent = {sum H, min H, max H, -}."""
    L.require_cuda(z, dz, scal)
    B, M = z.shape
    # {0, +inf, 0, 0} from a per-device constant by a device-to-device copy: a pageable host-to-device copy per call is
    # illegal inside HIP-graph capture and silently sent SigLIP + entropy regularisation back to eager
    init = _ENT_INIT.get(z.device)
    if init is None:
        init = _ENT_INIT[z.device] = torch.tensor([0.0, float("inf"), 0.0, 0.0], dtype=torch.float32, device=z.device)
    ent = init.clone()
    L.call("dcc_entropy_reg", L.ptr(z), L.ptr(dz), B, M, float(weight), float(thr), L.ptr(ent), L.ptr(scal), L.stream())
    return ent


def mp_infonce(z, w, mask, imp_base, mean: bool, scal, want_grad: bool = True):
    """Symmetric multi-positive InfoNCE on logits (see include/dcc.h).  w: clamped weights; imp_base: None or [N, M]."""
    L.require_cuda(z, w, mask, imp_base, scal)
    N, M = z.shape
    lr, lc = lse(z)
    wr, wc = rowcol_sums(w)
    mr, mc = rowcol_sums(mask)
    ir = ic = None
    if imp_base is not None:
        ir, ic = rowcol_sums(imp_base)
    ws = torch.zeros(2 * (N + M), device=z.device, dtype=torch.float32)
    dz = torch.empty_like(z) if want_grad else None
    L.call("dcc_mp_infonce", L.ptr(z), L.ptr(w), L.ptr(lr), L.ptr(lc), L.ptr(wr), L.ptr(wc), L.ptr(mr), L.ptr(mc), L.ptr(ir),
           L.ptr(ic), L.ptr(ws), L.ptr(dz), N, M, int(mean), float(torch.finfo(torch.float32).eps), L.ptr(scal), L.stream())
    return dz


def clamp01(x: torch.Tensor) -> torch.Tensor:
    """clamp(x, 0, 1) of a small fp32 mask (targets of the BCE losses)."""
    L.require_cuda(x)
    x = x.contiguous()
    assert x.dtype == torch.float32
    out = torch.empty_like(x)
    L.call("dcc_clamp", L.ptr(x), L.ptr(out), x.numel(), 0.0, 1.0, L.stream())
    return out


def view_embed_add(x: torch.Tensor, ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """x [rows, L, D] fp32 + table[ids[row]] broadcast over L (view-angle embeddings of the MIL head)."""
    L.require_cuda(x, ids, table)
    rows, Lt, D = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and table.dtype == torch.float32 and table.is_contiguous()
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == rows and table.shape[1] == D
    out = torch.empty_like(x)
    L.call("dcc_view_embed_add", L.ptr(x), L.ptr(ids), L.ptr(table), L.ptr(out), rows, Lt, D, L.stream())
    return out


def view_embed_bwd(dy: torch.Tensor, ids: torch.Tensor, classes: int) -> torch.Tensor:
    """Table gradient [classes, D] of view_embed_add: rows of dy summed over L, scattered by class."""
    L.require_cuda(dy, ids)
    rows, Lt, D = dy.shape
    assert dy.dtype == torch.float32 and dy.is_contiguous() and ids.numel() == rows
    dtable = torch.zeros((classes, D), device=dy.device, dtype=torch.float32)
    L.call("dcc_view_embed_bwd", L.ptr(dy), L.ptr(ids), L.ptr(dtable), rows, Lt, D, L.stream())
    return dtable


# ------------------------------------------------------------------------------------------ fp8
def fp8_quantize(x2d: torch.Tensor, fmt: int = 0, transpose: bool = False):
    """Per-tensor fp8 quantisation of a [rows, cols] view (unit inner stride): returns (uint8 tensor [rows, cols] or
    [cols, rows], fp32[1] dequantisation scale = amax / FMAX).  fmt 0 = e4m3, 1 = e5m2."""
    L.require_cuda(x2d)
    assert x2d.dim() == 2 and x2d.stride(1) == 1
    rows, cols = x2d.shape
    amax = torch.zeros(1, device=x2d.device, dtype=torch.float32)
    L.call("dcc_absmax", L.ptr(x2d), rows, cols, x2d.stride(0), L.dt(x2d), L.ptr(amax), L.stream())
    out = torch.empty((cols, rows) if transpose else (rows, cols), device=x2d.device, dtype=torch.uint8)
    scale = torch.empty(1, device=x2d.device, dtype=torch.float32)
    L.call("dcc_fp8_quantize", L.ptr(x2d), rows, cols, x2d.stride(0), L.dt(x2d), L.ptr(amax), L.ptr(out), out.stride(0),
           int(transpose), int(fmt), L.ptr(scale), L.stream())
    return out, scale


def gemm_fp8(a8: torch.Tensor, b8: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, *, fmt_a: int = 0, fmt_b: int = 0,
             bias=None, resid=None, aux=None, gelu=False, save_pre=False, dgelu=False, alpha: float = 1.0,
             out: torch.Tensor | None = None) -> torch.Tensor:
    """C[m,n] = alpha * sa * sb * sum_k A8(m,k) B8(n,k) (+ epilogue) in bf16; a8 [M,K], b8 [N,K] uint8 (fp8 bytes)."""
    L.require_cuda(a8, b8, sa, sb, bias, resid, aux, out)
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and a8.dim() == 2 and b8.dim() == 2
    assert a8.stride(1) == 1 and b8.stride(1) == 1 and a8.shape[1] == b8.shape[1]
    M, K = a8.shape
    N = b8.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a8.device, dtype=torch.bfloat16)
    flags = 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
        flags |= GEMM_BIAS
    if gelu:
        flags |= GEMM_GELU
    if save_pre:
        assert aux is not None and aux.shape == (M, N) and aux.dtype == torch.bfloat16
        flags |= GEMM_SAVE_PRE
    if dgelu:
        assert aux is not None and aux.shape == (M, N) and aux.dtype == torch.bfloat16
        flags |= GEMM_DGELU
    if resid is not None:
        assert resid.shape == (M, N) and resid.dtype == torch.bfloat16 and resid.stride(1) == 1
        flags |= GEMM_RESID
    nbytes = (M * K + N * K) + M * N * 2 * (1 + int(resid is not None) + int(save_pre) + int(dgelu))
    with _timed("gemm_fp8", 2.0 * M * N * K, lambda: f"M{M} N{N} K{K} f{flags}", nbytes):
        L.call("dcc_gemm_fp8", L.ptr(a8), L.ptr(b8), L.ptr(out), L.ptr(bias), L.ptr(resid), L.ptr(aux), L.ptr(sa), L.ptr(sb),
               a8.stride(0), b8.stride(0), out.stride(0), resid.stride(0) if resid is not None else 0,
               aux.stride(0) if aux is not None else 0, M, N, K, float(alpha), flags, int(fmt_a), int(fmt_b), L.stream())
    return out


def relu(x: torch.Tensor, gate: torch.Tensor | None = None) -> torch.Tensor:
    """gate None: max(x, 0); else x * [gate > 0] (the backward form, gate = forward output)."""
    L.require_cuda(x, gate)
    assert x.is_contiguous() and (gate is None or (gate.is_contiguous() and gate.shape == x.shape and gate.dtype == x.dtype))
    out = torch.empty_like(x)
    L.call("dcc_relu", L.ptr(x), L.ptr(gate), L.ptr(out), x.numel(), L.dt(x), L.stream())
    return out


# ------------------------------------------------------------------------- multi-instance pooling
def gate_fwd(v, u):
    L.require_cuda(v, u)
    assert v.is_contiguous() and u.is_contiguous() and v.shape == u.shape and v.dtype == u.dtype
    out = torch.empty_like(v)
    L.call("dcc_gate_fwd", L.ptr(v), L.ptr(u), L.ptr(out), v.numel(), L.dt(v), L.stream())
    return out


def gate_bwd(d, v, u):
    L.require_cuda(d, v, u)
    dv, du = torch.empty_like(v), torch.empty_like(u)
    L.call("dcc_gate_bwd", L.ptr(d.contiguous()), L.ptr(v), L.ptr(u), L.ptr(dv), L.ptr(du), v.numel(), L.dt(v), L.stream())
    return dv, du


def masked_softmax_fwd(s, mask):
    L.require_cuda(s, mask)
    B, N = s.shape
    assert s.dtype == torch.float32 and s.is_contiguous() and (mask is None or (mask.dtype == torch.uint8 and mask.shape == (B, N) and mask.is_contiguous()))
    w = torch.empty_like(s)
    L.call("dcc_masked_softmax_fwd", L.ptr(s), L.ptr(mask), L.ptr(w), B, N, L.stream())
    return w


def masked_softmax_bwd(w, dw):
    L.require_cuda(w, dw)
    ds = torch.empty_like(w)
    L.call("dcc_masked_softmax_bwd", L.ptr(w), L.ptr(dw.contiguous()), L.ptr(ds), w.shape[0], w.shape[1], L.stream())
    return ds


def wsum_fwd(w, x):
    L.require_cuda(w, x)
    B, N, D = x.shape
    assert w.shape == (B, N) and w.dtype == torch.float32 and w.is_contiguous() and x.is_contiguous()
    out = torch.empty((B, D), device=x.device, dtype=x.dtype)
    L.call("dcc_wsum_fwd", L.ptr(w), L.ptr(x), L.ptr(out), B, N, D, L.dt(x), L.stream())
    return out


def wsum_bwd(w, x, dout):
    L.require_cuda(w, x, dout)
    B, N, D = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    L.call("dcc_wsum_bwd", L.ptr(w), L.ptr(x), L.ptr(dout.contiguous()), L.ptr(dx), L.ptr(dw), B, N, D, L.dt(x), L.stream())
    return dx, dw


def masked_reduce_fwd(x, mask, mode: int):
    L.require_cuda(x, mask)
    B, N, D = x.shape
    assert x.is_contiguous() and (mask is None or (mask.dtype == torch.uint8 and mask.shape == (B, N) and mask.is_contiguous()))
    out = torch.empty((B, D), device=x.device, dtype=x.dtype)
    arg = torch.empty((B, D), device=x.device, dtype=torch.int32) if mode == 1 else None
    L.call("dcc_masked_reduce_fwd", L.ptr(x), L.ptr(mask), L.ptr(out), L.ptr(arg), B, N, D, int(mode), L.dt(x), L.stream())
    return out, arg


def masked_reduce_bwd(dout, mask, arg, N: int, mode: int):
    L.require_cuda(dout, mask, arg)
    B, D = dout.shape
    dx = torch.empty((B, N, D), device=dout.device, dtype=dout.dtype)
    L.call("dcc_masked_reduce_bwd", L.ptr(dout.contiguous()), L.ptr(mask), L.ptr(arg), L.ptr(dx), B, N, D, int(mode), L.dt(dout), L.stream())
    return dx
