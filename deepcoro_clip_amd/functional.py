"""Autograd glue: `torch.autograd.Function`s whose forward AND backward are HIP kernels.

torch supplies the tape, device memory and views; every arithmetic step calls `kernels.py`
(-> libdcc_hip.so).  Activations live in the compute dtype (bf16 or fp32), parameters stay fp32
masters with cached bf16 shadows, weight/bias gradients come back in fp32.
"""
from __future__ import annotations

import math
import collections
import os

import torch

from . import _grad_slots
from . import kernels as K

# ------------------------------------------------------------------------------ weight shadows
_WEPOCH = 0


def invalidate_weight_cache() -> None:
    """Call after parameters were updated through raw pointers (our fused optimizer)."""
    global _WEPOCH
    _WEPOCH += 1


def wcast(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 parameter (or a slice view of one) -> contiguous tensor in the compute dtype.

    The shadow is cached ON the owning parameter object (never keyed by address: freed parameters' addresses are
    reused by the allocator) and is valid for one (autograd version, optimizer epoch) pair."""
    wd = w.detach()
    if dtype == torch.float32:
        return wd if wd.is_contiguous() else wd.contiguous()
    owner = w._base if w._base is not None else w
    cache = owner.__dict__.get("_dcc_shadows")
    if cache is None:
        cache = {}
        owner.__dict__["_dcc_shadows"] = cache
    key = (w.storage_offset(), tuple(w.shape), tuple(w.stride()), dtype)
    stamp = (owner._version, _WEPOCH, owner.data_ptr())
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    src = wd if wd.is_contiguous() else wd.contiguous()
    t = K.cast(src, dtype)
    cache[key] = (stamp, t)
    return t


def wcast_t(w: torch.Tensor) -> torch.Tensor:
    """Transposed bf16 shadow [K, N] of an fp32 weight [N, K], cached on the parameter like `wcast`'s shadows; the optimizer's
    `invalidate_weight_cache()` retires it once per step (one small transpose launch per weight and step)."""
    owner = w._base if w._base is not None else w
    cache = owner.__dict__.setdefault("_dcc_shadows_t", {})
    key = (w.storage_offset(), tuple(w.shape), tuple(w.stride()))
    stamp = (owner._version, _WEPOCH, owner.data_ptr())
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    t = K.cast_transpose(w.detach())
    cache[key] = (stamp, t)
    return t


def _dx_gemm(g: torch.Tensor, weight: torch.Tensor, wc: torch.Tensor, **kw) -> torch.Tensor:
    """g [rows, N] @ W [N, Kd] (the data gradient of a Linear layer, W k-major for the GEMM).  Where the product is in the
    streaming panel kernel's domain (early MViT blocks) it runs there on a transposed weight shadow; else on the tiled kernels'
    k-major operand path."""
    if weight is not None and weight.dtype == torch.float32 and weight.dim() == 2 and weight.stride(1) == 1 and \
            K.stream_gemm_shape(g.shape[0], wc.shape[1], wc.shape[0], g.dtype):
        return K.gemm(g, wcast_t(weight), **kw)
    return K.gemm(g, wc, b_km=True, **kw)


# ------------------------------------------------------------------------------ fp8 GEMM mode
# BASELINE configs[4]: the towers' forward and data-gradient GEMMs in fp8 (block-scaled MFMA, per-tensor scales: e4m3 for
# activations and weights, e5m2 for gradients), weight gradients stay bf16.  Switched on around the video trunk by
# VideoEncoder(gemm_dtype="fp8"); shapes the fp8 kernel does not take (K <= 128, odd alignments) stay bf16.
_FP8 = {"on": False}


class fp8_gemms:
    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def __enter__(self):
        self.prev = _FP8["on"]
        _FP8["on"] = bool(self.enabled)

    def __exit__(self, *exc):
        _FP8["on"] = self.prev
        return False


def _fp8_shape(rows: int, K_: int, N: int, dtype) -> bool:
    return dtype == torch.bfloat16 and K_ % 16 == 0 and K_ > 128 and N % 8 == 0 and rows >= 128


def _fp8_ok(rows: int, K_: int, N: int, dtype) -> bool:
    return _FP8["on"] and _fp8_shape(rows, K_, N, dtype)


def wquant(w: torch.Tensor, transpose: bool):
    """fp8 (e4m3) copy of an fp32 weight [N, K] (or its transpose [K, N]) + its scale, cached on the parameter like the
    bf16 shadows (valid for one (autograd version, optimizer epoch) pair)."""
    owner = w._base if w._base is not None else w
    cache = owner.__dict__.setdefault("_dcc_fp8", {})
    key = (w.storage_offset(), tuple(w.shape), tuple(w.stride()), transpose)
    stamp = (owner._version, _WEPOCH, owner.data_ptr())
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1], hit[2]
    wd = w.detach()
    q, sc = K.fp8_quantize(wd if wd.is_contiguous() else wd.contiguous(), 0, transpose)
    cache[key] = (stamp, q, sc)
    return q, sc


def stacked_tables(tables, rows: int, width: int, dtype: torch.dtype) -> torch.Tensor:
    """The rel-pos tables stacked row-wise in the compute dtype, zero padded to `rows`.

    When the tables are bf16-shadowed fp32 Parameters the stack is a persistent buffer whose row slices ARE the
    parameters' shadows (same cache `wcast` uses): the AdamW kernel rewrites them in its update pass, so a training step
    spends no cast / fill launches on them (4 per block before).  Anything else (fp32 compute, resampled tables that are
    autograd intermediates) is cast on the spot."""
    if dtype != torch.float32 and all(isinstance(t, torch.nn.Parameter) and t.is_contiguous() for t in tables):
        owner = tables[0]
        hit = owner.__dict__.get("_dcc_rcat")
        if hit is None or hit[0] != (rows, width, dtype, tuple(id(t) for t in tables)) or hit[1].device != owner.device:
            buf = torch.zeros((rows, width), device=owner.device, dtype=dtype)
            hit = ((rows, width, dtype, tuple(id(t) for t in tables)), buf)
            owner.__dict__["_dcc_rcat"] = hit
        buf = hit[1]
        o = 0
        for t in tables:
            cache = t.__dict__.setdefault("_dcc_shadows", {})
            key = (0, tuple(t.shape), tuple(t.stride()), dtype)
            stamp = (t._version, _WEPOCH, t.data_ptr())
            view = buf[o:o + t.shape[0]]
            ent = cache.get(key)
            if ent is None or ent[0] != stamp or ent[1].data_ptr() != view.data_ptr():
                K.cast(t.detach(), dtype, out=view)
                cache[key] = (stamp, view)
            o += t.shape[0]
        return buf
    out = torch.zeros((rows, width), device=tables[0].device, dtype=dtype)
    o = 0
    for t in tables:
        K.cast(t.detach().contiguous(), dtype, out=out[o:o + t.shape[0]])
        o += t.shape[0]
    return out


class ZeroArena:
    """One big pre-zeroed fp32 buffer per training step for every gradient that is produced by accumulation
    (split-K / atomics): one memset instead of ~800 small fill launches.  `begin()` re-zeroes and rewinds; slices
    handed out stay valid until the next `begin()` (the runner calls it where the reference calls zero_grad)."""

    def __init__(self):
        self.buf: torch.Tensor | None = None
        self.off = 0
        self.active = False
        self.high = 0
        self.zeroed = 0

    def begin(self, device, numel_hint: int):
        need = max(numel_hint, self.high)
        if self.buf is None or self.buf.device != device or self.buf.numel() < need:
            self.buf = torch.empty(int(need * 1.05) + 4096, device=device, dtype=torch.float32)
        # zero what a step actually takes (the previous steps' high-water mark + 25 %), not the whole buffer: the headline step
        # uses 141 MB of a 640 MB arena; slices beyond the zeroed prefix fall back to torch.zeros (shape changes)
        self.zeroed = self.buf.numel() if self.high == 0 else min(self.buf.numel(), int(self.high * 1.25) + (1 << 20))
        self.buf[:self.zeroed].zero_()
        self.off = 0
        self.active = True

    def end(self):
        self.active = False

    def zeros(self, shape, device) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        if not self.active:
            return torch.zeros(shape, device=device, dtype=torch.float32)
        n4 = (n + 3) // 4 * 4  # keep every slice 16-byte aligned
        if self.buf is not None and self.buf.device == device and self.off + n4 <= self.zeroed:
            t = self.buf[self.off:self.off + n].view(shape)
            self.off += n4
            self.high = max(self.high, self.off)
            return t
        self.high = max(self.high, self.off + n4)
        return torch.zeros(shape, device=device, dtype=torch.float32)


ARENA = ZeroArena()


def zeros_f32(shape, device) -> torch.Tensor:
    return ARENA.zeros(tuple(shape) if not isinstance(shape, int) else (shape,), device)


# Weight gradients feed nothing until the optimizer, so they can leave the backward's critical chain: with WGRAD_SIDE
# (set by the runner: one rank, no accumulation) a weight-gradient GEMM whose parameter has no gradient yet (AccumulateGrad
# will adopt the buffer without touching it) runs on a side stream, filling the launch gaps and partial last waves of the
# data-gradient chain; the main stream re-joins it in an end-of-backward engine callback.
WGRAD_SIDE = False
_WG = {"stream": None, "queued": False, "held": collections.deque()}


def _join_wgrad():
    _WG["queued"] = False
    if _WG["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_WG["stream"])
    _WG["held"].clear()


def _hold_gradient(g2: torch.Tensor, ws) -> None:
    """The incoming gradient a side-stream weight-gradient GEMM reads stays referenced until that GEMM has run.  Not for its memory
    (record_stream covers that): autograd accumulates gradients that meet at a tensor IN PLACE when nobody else holds the buffer
    (InputBuffer: use_count == 1) — on the main stream, possibly while the side stream is still reading that very gradient, e.g. the
    one an add node hands to both its inputs (round 4: the study aggregator's out_proj weight gradient came out 77 % off).  A second
    reference makes autograd allocate the sum instead.  Released as soon as the GEMM's event has fired (checked at the next launch),
    so only the few gradients the side stream is behind by stay alive; the OUTPUT is never held — AccumulateGrad adopts a gradient
    buffer only while it is the sole owner."""
    held = _WG["held"]
    if torch.cuda.is_current_stream_capturing():  # HIP-graph capture of the step: events cannot be queried; held until the join
        held.append((None, g2))
        return
    while held and held[0][0] is not None and held[0][0].query():
        held.popleft()
    ev = torch.cuda.Event()
    ev.record(ws)
    held.append((ev, g2))


def side_param_work(fn, *tensors) -> bool:
    """Run `fn` (kernels that only produce PARAMETER gradients: nothing on the data path waits for them) on the weight-gradient
    side stream; the main stream re-joins it in the end-of-backward engine callback.  `tensors`: everything `fn` reads or writes
    (kept alive for the side stream).  False (nothing launched) outside a backward pass."""
    if not _WG["queued"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_wgrad)
            _WG["queued"] = True
        except RuntimeError:
            return False
    dev = tensors[0].device
    if _WG["stream"] is None or _WG["stream"].device != dev:
        _WG["stream"] = torch.cuda.Stream(device=dev)
        _WG["event"] = torch.cuda.Event()
    ws, ev = _WG["stream"], _WG["event"]
    ev.record()            # the inputs are complete on the current stream here
    ws.wait_event(ev)
    with torch.cuda.stream(ws):
        fn()
    for t in tensors:
        t.record_stream(ws)
    return True


def grads_unset(*params) -> bool:
    """True when none of the given tensors carries a .grad yet (then autograd adopts a fresh gradient buffer as is)."""
    return all(p is None or (p.is_leaf and p.grad is None) for p in params)  # non-leaf: not ours to reason about


def wgrad(g2: torch.Tensor, x2: torch.Tensor, colsum: torch.Tensor | None = None, side: bool = False,
          out: torch.Tensor | None = None) -> torch.Tensor:
    """dW[n][k] = sum_rows g2[row][n] * x2[row][k] in fp32 (k-major GEMM; split over the row range when the
    n x k output alone cannot fill the chip, accumulating into a pre-zeroed arena slice)."""
    rows, N = g2.shape
    Kd = x2.shape[1]
    sk = K.wgrad_splitk(rows, N, Kd)
    if out is None:  # `out`: a zeroed gradient slot handed out by the owner of the gradient memory (_grad_slots)
        out = zeros_f32((N, Kd), g2.device) if sk > 1 else torch.empty((N, Kd), device=g2.device, dtype=torch.float32)
    side = side and WGRAD_SIDE
    if side and not _WG["queued"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_wgrad)
            _WG["queued"] = True
        except RuntimeError:  # not inside a backward pass: nobody would join the stream
            side = False
    if not side:
        return K.gemm(g2, x2, a_km=True, b_km=True, out=out, accumulate=sk > 1, splitk=sk, colsum=colsum)
    if _WG["stream"] is None or _WG["stream"].device != g2.device:
        _WG["stream"] = torch.cuda.Stream(device=g2.device)
        _WG["event"] = torch.cuda.Event()
    ws, ev = _WG["stream"], _WG["event"]
    ev.record()            # g2 / x2 are complete on the current stream here
    ws.wait_event(ev)
    K.gemm(g2, x2, a_km=True, b_km=True, out=out, accumulate=sk > 1, splitk=sk, colsum=colsum, stream=ws.cuda_stream)
    g2.record_stream(ws)
    x2.record_stream(ws)
    _hold_gradient(g2, ws)
    return out


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _as_dtype(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    return t if t.dtype == dtype else K.cast(_c(t), dtype)


# ------------------------------------------------------------------------------------- Linear
class LinearFn(torch.autograd.Function):
    """y = [gelu](x W^T + b) [+ residual]   (nn.Linear [+ nn.GELU] [+ skip add])."""

    @staticmethod
    def forward(ctx, x, weight, bias, gelu, residual):
        K_ = x.shape[-1]
        N = weight.shape[0]
        x2 = _c(x).reshape(-1, K_)
        wc = wcast(weight, x.dtype)
        res2 = _c(residual).reshape(-1, N) if residual is not None else None
        pre = torch.empty((x2.shape[0], N), device=x.device, dtype=x.dtype) if gelu else None
        ctx.fp8 = _fp8_ok(x2.shape[0], K_, N, x.dtype)
        if ctx.fp8:
            x8, sx = K.fp8_quantize(x2, 0)
            w8, sw = wquant(weight, False)
            y = K.gemm_fp8(x8, w8, sx, sw, bias=bias.detach() if bias is not None else None, gelu=gelu, save_pre=gelu, aux=pre, resid=res2)
        else:
            y = K.gemm(x2, wc, bias=bias.detach() if bias is not None else None, gelu=gelu, save_pre=gelu, aux=pre, resid=res2)
        ctx.save_for_backward(x2, wc, pre)
        ctx.params = (weight, bias)
        ctx.gelu = gelu
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.xshape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, wc, pre = ctx.saved_tensors
        N = wc.shape[0]
        dy2 = _c(dy).reshape(-1, N)
        g = K.dgelu_mul(dy2, pre) if ctx.gelu else dy2
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            if ctx.fp8 and _fp8_shape(g.shape[0], N, wc.shape[1], g.dtype):  # dX = dY W: gradients in e5m2, W^T in e4m3
                g8, sg = K.fp8_quantize(g, 1)
                wt8, swt = wquant(ctx.params[0], True)
                dx = K.gemm_fp8(g8, wt8, sg, swt, fmt_a=1).reshape(ctx.xshape)
            else:
                dx = _dx_gemm(g, ctx.params[0], wc).reshape(ctx.xshape)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_db:  # bias gradient rides in the weight-gradient GEMM (column sums of its k-major A operand)
                db = _grad_slots.take(ctx.params[1])
                if db is None:
                    db = zeros_f32(N, g.device)
            dw = wgrad(g, x2, db, side=WGRAD_SIDE and grads_unset(*ctx.params), out=_grad_slots.take(ctx.params[0]))
        elif want_db:
            db = K.colsum(g)
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = dy
        return dx, dw, db, None, dres


def linear(x, weight, bias=None, gelu: bool = False, residual=None):
    return LinearFn.apply(x, weight, bias, gelu, residual)


class Linear2Fn(torch.autograd.Function):
    """Two Linear layers of ONE input (the MViT blocks that change width: `attn.qkv` and the skip `project` both read
    norm1(x), torchvision MultiscaleBlock.forward via models/video_encoder.py:232-253).  The backward chains the two
    data-gradient GEMMs — the second adds the first through its residual epilogue — instead of leaving a [rows, C] add of
    the two contributions to autograd (77 M elements in block 1)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        Kd = x.shape[-1]
        x2 = _c(x).reshape(-1, Kd)
        wc1, wc2 = wcast(w1, x.dtype), wcast(w2, x.dtype)
        y1 = K.gemm(x2, wc1, bias=b1.detach() if b1 is not None else None)
        y2 = K.gemm(x2, wc2, bias=b2.detach() if b2 is not None else None)
        ctx.save_for_backward(x2, wc1, wc2)
        ctx.params = (w1, b1, w2, b2)
        ctx.xshape = x.shape
        return y1.reshape(*x.shape[:-1], w1.shape[0]), y2.reshape(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy1, dy2):
        x2, wc1, wc2 = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        g1 = _c(dy1).reshape(-1, wc1.shape[0])
        g2 = _c(dy2).reshape(-1, wc2.shape[0])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dx_gemm(g1, w1, wc1)
            dx = _dx_gemm(g2, w2, wc2, resid=dx).reshape(ctx.xshape)
        grads = []
        for g, w, b, iw, ib in ((g1, w1, b1, 1, 2), (g2, w2, b2, 3, 4)):
            dw = db = None
            want_db = b is not None and ctx.needs_input_grad[ib]
            if ctx.needs_input_grad[iw]:
                if want_db:
                    db = _grad_slots.take(b)
                    if db is None:
                        db = zeros_f32(g.shape[1], g.device)
                dw = wgrad(g, x2, db, side=WGRAD_SIDE and grads_unset(w, b), out=_grad_slots.take(w))
            elif want_db:
                db = K.colsum(g)
            grads += [dw, db]
        return (dx, *grads)


class LinearNSkipFn(torch.autograd.Function):
    """N Linear layers of ONE input that also feeds a residual branch (BERT's post-LN layer: the LayerNorm output goes to
    `self.query / key / value` and, unchanged, to the attention output's residual add — HF BertLayer via models/text_encoder.py:52).
    Returns (y_1 .. y_N, x'): x' is an alias of x for the residual consumer, so that ALL gradients of x meet in this node and are
    summed by the data-gradient GEMMs' residual epilogues (autograd otherwise launches N elementwise adds per layer)."""

    @staticmethod
    def forward(ctx, x, *wb):
        Kd = x.shape[-1]
        x2 = _c(x).reshape(-1, Kd)
        ws, bs = wb[0::2], wb[1::2]
        wcs = [wcast(w, x.dtype) for w in ws]
        ys = [K.gemm(x2, wc, bias=b.detach() if b is not None else None) for wc, b in zip(wcs, bs)]
        ctx.save_for_backward(x2, *wcs)
        ctx.params = (ws, bs)
        ctx.xshape = x.shape
        return (*[y.reshape(*x.shape[:-1], w.shape[0]) for y, w in zip(ys, ws)], x.view_as(x))

    @staticmethod
    def backward(ctx, *dys):
        x2, *wcs = ctx.saved_tensors
        ws, bs = ctx.params
        n = len(ws)
        dskip = dys[n]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _as_dtype(_c(dskip).reshape(x2.shape), x2.dtype) if dskip is not None else None
            for dy, w, wc in zip(dys[:n], ws, wcs):
                if dy is None:
                    continue
                g = _c(dy).reshape(-1, wc.shape[0])
                dx = _dx_gemm(g, w, wc, resid=dx) if dx is not None else _dx_gemm(g, w, wc)
            dx = dx.reshape(ctx.xshape) if dx is not None else None
        grads = []
        for i, (dy, w, b) in enumerate(zip(dys[:n], ws, bs)):
            dw = db = None
            if dy is not None:
                g = _c(dy).reshape(-1, w.shape[0])
                want_db = b is not None and ctx.needs_input_grad[2 + 2 * i]
                if ctx.needs_input_grad[1 + 2 * i]:
                    if want_db:
                        db = _grad_slots.take(b)
                        if db is None:
                            db = zeros_f32(g.shape[1], g.device)
                    dw = wgrad(g, x2, db, side=WGRAD_SIDE and grads_unset(w, b), out=_grad_slots.take(w))
                elif want_db:
                    db = K.colsum(g)
            grads += [dw, db]
        return (dx, *grads)


def linears_skip(x, *layers):
    """layers: (weight, bias) pairs -> (y_1, ..., y_N, x') with x' the alias of x for its residual use.  fp8 GEMM mode and fp32
    compute keep the plain nodes."""
    M = x.numel() // x.shape[-1]
    if any(_fp8_ok(M, x.shape[-1], w.shape[0], x.dtype) for w, _ in layers):
        return (*[linear(x, w, b) for w, b in layers], x)
    flat = [t for wb in layers for t in wb]
    return LinearNSkipFn.apply(x, *flat)


def linear2(x, w1, b1, w2, b2):
    """(linear(x, w1, b1), linear(x, w2, b2)) with one fused data gradient; fp8 GEMM mode keeps the two separate layers."""
    M = x.numel() // x.shape[-1]
    if _fp8_ok(M, x.shape[-1], w1.shape[0], x.dtype) or _fp8_ok(M, x.shape[-1], w2.shape[0], x.dtype):
        return linear(x, w1, b1), linear(x, w2, b2)
    return Linear2Fn.apply(x, w1, b1, w2, b2)


class MlpFn(torch.autograd.Function):
    """y = W2 gelu(W1 x + b1) + b2 [+ residual]; backward fuses gelu' into the fc2 data-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual):
        Kd = x.shape[-1]
        x2 = _c(x).reshape(-1, Kd)
        # y = x + mlp(x) (BERT's post-LN feed-forward): the residual's gradient IS dy and joins dx in the last GEMM's epilogue
        ctx.res_is_x = residual is x
        w1c, w2c = wcast(w1, x.dtype), wcast(w2, x.dtype)
        Hd, N = w1.shape[0], w2.shape[0]
        pre = torch.empty((x2.shape[0], Hd), device=x.device, dtype=x.dtype)
        res2 = _c(residual).reshape(-1, N) if residual is not None else None
        ctx.fp8 = _fp8_ok(x2.shape[0], Kd, Hd, x.dtype) and _fp8_ok(x2.shape[0], Hd, N, x.dtype)
        if ctx.fp8:
            x8, sx = K.fp8_quantize(x2, 0)
            w18, s1 = wquant(w1, False)
            act = K.gemm_fp8(x8, w18, sx, s1, bias=b1.detach(), gelu=True, save_pre=True, aux=pre)
            a8, sa = K.fp8_quantize(act, 0)
            w28, s2 = wquant(w2, False)
            y = K.gemm_fp8(a8, w28, sa, s2, bias=b2.detach(), resid=res2)
        else:
            act = K.gemm(x2, w1c, bias=b1.detach(), gelu=True, save_pre=True, aux=pre)
            y = K.gemm(act, w2c, bias=b2.detach(), resid=res2)
        ctx.save_for_backward(x2, w1c, w2c, pre, act)
        ctx.params = (w1, b1, w2, b2)
        ctx.has_res = residual is not None
        ctx.xshape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w1c, w2c, pre, act = ctx.saved_tensors
        Hd, N = w1c.shape[0], w2c.shape[0]
        dy2 = _c(dy).reshape(-1, N)
        rows = dy2.shape[0]
        dw2 = db2 = dw1 = db1 = dx = None
        f32 = dict(device=dy2.device, dtype=torch.float32)
        if ctx.needs_input_grad[3]:
            db2 = (_grad_slots.take(ctx.params[3]) if ctx.needs_input_grad[4] else None)
            if db2 is None and ctx.needs_input_grad[4]:
                db2 = zeros_f32(N, dy2.device)
            dw2 = wgrad(dy2, act, db2, side=WGRAD_SIDE and grads_unset(*ctx.params[2:]), out=_grad_slots.take(ctx.params[2]))
        elif ctx.needs_input_grad[4]:
            db2 = K.colsum(dy2)
        fp8 = ctx.fp8 and _fp8_shape(rows, N, Hd, dy2.dtype) and _fp8_shape(rows, Hd, x2.shape[1], dy2.dtype)
        if fp8:
            g8, sg = K.fp8_quantize(dy2, 1)
            w2t8, s2t = wquant(ctx.params[2], True)
            dpre = K.gemm_fp8(g8, w2t8, sg, s2t, fmt_a=1, dgelu=True, aux=pre)
        else:
            dpre = _dx_gemm(dy2, ctx.params[2], w2c, dgelu=True, aux=pre)
        if ctx.needs_input_grad[1]:
            db1 = (_grad_slots.take(ctx.params[1]) if ctx.needs_input_grad[2] else None)
            if db1 is None and ctx.needs_input_grad[2]:
                db1 = zeros_f32(Hd, dy2.device)
            dw1 = wgrad(dpre, x2, db1, side=WGRAD_SIDE and grads_unset(*ctx.params[:2]), out=_grad_slots.take(ctx.params[0]))
        elif ctx.needs_input_grad[2]:
            db1 = K.colsum(dpre)
        if ctx.needs_input_grad[0]:
            if fp8:
                p8, sp = K.fp8_quantize(dpre, 1)
                w1t8, s1t = wquant(ctx.params[0], True)
                dx = K.gemm_fp8(p8, w1t8, sp, s1t, fmt_a=1).reshape(ctx.xshape)
            elif ctx.res_is_x and dy2.dtype == x2.dtype:
                dx = _dx_gemm(dpre, ctx.params[0], w1c, resid=dy2).reshape(ctx.xshape)
                return dx, dw1, db1, dw2, db2, None
            else:
                dx = _dx_gemm(dpre, ctx.params[0], w1c).reshape(ctx.xshape)
        return dx, dw1, db1, dw2, db2, (dy if ctx.has_res else None)


def mlp(x, w1, b1, w2, b2, residual=None):
    return MlpFn.apply(x, w1, b1, w2, b2, residual)


def wpack_mlp(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """The fused MLP kernel's weight image of (fc1.weight, fc2.weight), cached on fc1's parameter like the bf16 shadows and valid
    for one (autograd versions, optimizer epoch) pair: one small repack launch per MLP and step."""
    cache = w1.__dict__.setdefault("_dcc_mlp_pack", {})
    stamp = (w1._version, w2._version, _WEPOCH, w1.data_ptr(), w2.data_ptr())
    hit = cache.get("p")
    if hit is not None and hit[0] == stamp:
        return hit[1]
    t = K.mlp_pack(wcast(w1, torch.bfloat16), wcast(w2, torch.bfloat16))
    cache["p"] = (stamp, t)
    return t


# Which channel counts take the fused LayerNorm + MLP forward kernel.  Measured at the tower's shapes (tools/mlp_bench.py, 32 clips):
# C = 96 362 vs 584 us (at the HBM roofline of the tensors it must write), C = 192 263 vs 348 us; C = 384 is not there yet (one wave
# per SIMD cannot cover its own epilogue: 243-258 vs 195-212 us) and stays on LayerNorm + two GEMMs.  DCC_FUSED_MLP = 0 | 1 | all.
_FUSED_MLP_ENV = os.environ.get("DCC_FUSED_MLP", "1")
FUSED_MLP_CHANNELS = () if _FUSED_MLP_ENV == "0" else ((96, 192, 384) if _FUSED_MLP_ENV == "all" else (96, 192))


def ln_mlp_ok(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> bool:
    C = x.shape[-1]
    fused = C in FUSED_MLP_CHANNELS
    if not fused and C == 384 and _FUSED_MLP_ENV != "0":
        # without the tensors a backward would need (validation / inference()) the fused kernel wins at C = 384 too: 192 vs 212 us
        fused = not (torch.is_grad_enabled() and (x.requires_grad or w1.requires_grad or w2.requires_grad))
    return (fused and x.dtype == torch.bfloat16 and not _FP8["on"] and w1.shape == (4 * C, C)
            and w2.shape == (C, 4 * C) and w1.dtype == torch.float32 and w1.is_contiguous() and w2.is_contiguous())


class LnMlpFn(torch.autograd.Function):
    """y = [x +] fc2(gelu(fc1(LayerNorm(x)))): the second half of an MViT block (torchvision MultiscaleBlock, reached through
    models/video_encoder.py:232-253) as ONE forward launch (csrc/mlp_fused.hip).  The backward is the unfused one: the kernel
    saved LayerNorm(x), the statistics, the pre-activation and the activation."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, w2, b2, eps, residual):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        save = any(ctx.needs_input_grad)
        y, xn, mean, rstd, pre, act = K.ln_mlp_fwd(x2, gamma.detach(), beta.detach(), eps, wpack_mlp(w1, w2), b1.detach(),
                                                  b2.detach(), residual=residual, save=save)
        if save:
            ctx.save_for_backward(x2, xn, mean, rstd, gamma.detach(), wcast(w1, x.dtype), wcast(w2, x.dtype), pre, act)
        ctx.params = (w1, b1, w2, b2)
        ctx.residual = residual
        ctx.xshape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, xn, mean, rstd, gamma, w1c, w2c, pre, act = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        Hd, C = w1c.shape
        dy2 = _as_dtype(_c(dy).reshape(-1, C), x2.dtype)
        dw1 = db1 = dw2 = db2 = None
        if ctx.needs_input_grad[5]:
            db2 = _grad_slots.take(b2) if ctx.needs_input_grad[6] else None
            if db2 is None and ctx.needs_input_grad[6]:
                db2 = zeros_f32(C, dy2.device)
            dw2 = wgrad(dy2, act, db2, side=WGRAD_SIDE and grads_unset(w2, b2), out=_grad_slots.take(w2))
        elif ctx.needs_input_grad[6]:
            db2 = K.colsum(dy2)
        dpre = _dx_gemm(dy2, w2, w2c, dgelu=True, aux=pre)
        if ctx.needs_input_grad[3]:
            db1 = _grad_slots.take(b1) if ctx.needs_input_grad[4] else None
            if db1 is None and ctx.needs_input_grad[4]:
                db1 = zeros_f32(Hd, dy2.device)
            dw1 = wgrad(dpre, xn, db1, side=WGRAD_SIDE and grads_unset(w1, b1), out=_grad_slots.take(w1))
        elif ctx.needs_input_grad[4]:
            db1 = K.colsum(dpre)
        dx = dg = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dxn = _dx_gemm(dpre, w1, w1c)
            dgb = zeros_f32((2, C), x2.device)
            dx = K.layernorm_bwd(dxn, x2, mean, rstd, gamma, dgb[0], dgb[1], dres=dy2 if ctx.residual else None).reshape(ctx.xshape)
            dg, db = dgb[0], dgb[1]
        return dx, dg, db, dw1, db1, dw2, db2, None, None


def ln_mlp(x, gamma, beta, w1, b1, w2, b2, eps: float, residual: bool = True):
    return LnMlpFn.apply(x, gamma, beta, w1, b1, w2, b2, eps, residual)


# ---------------------------------------------------------------------------------- LayerNorm
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        y, mean, rstd = K.layernorm_fwd(x2, gamma.detach(), beta.detach(), eps)
        ctx.save_for_backward(x2, mean, rstd, gamma.detach())
        ctx.xshape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, gamma = ctx.saved_tensors
        C = x2.shape[1]
        dy2 = _as_dtype(_c(dy).reshape(-1, C), x2.dtype)
        dgb = zeros_f32((2, C), x2.device)
        dx = K.layernorm_bwd(dy2, x2, mean, rstd, gamma, dgb[0], dgb[1])
        return dx.reshape(ctx.xshape), dgb[0], dgb[1], None


def layernorm(x, gamma, beta, eps: float):
    return LayerNormFn.apply(x, gamma, beta, eps)


class LayerNormSkipFn(torch.autograd.Function):
    """(LayerNorm(x), x): the pre-norm transformer pattern where x also feeds the residual branch.  Returning x through
    the node lets backward receive BOTH gradients and add them inside the LayerNorm-backward kernel, instead of
    autograd launching a separate elementwise add (one extra read+write pass per residual) per block."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = x.shape[-1]
        x2 = _c(x).reshape(-1, C)
        y, mean, rstd = K.layernorm_fwd(x2, gamma.detach(), beta.detach(), eps)
        ctx.save_for_backward(x2, mean, rstd, gamma.detach())
        ctx.xshape = x.shape
        return y.reshape(x.shape), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x2, mean, rstd, gamma = ctx.saved_tensors
        C = x2.shape[1]
        dy2 = _as_dtype(_c(dy).reshape(-1, C), x2.dtype)
        dr2 = _as_dtype(_c(dskip).reshape(-1, C), x2.dtype) if dskip is not None else None
        dgb = zeros_f32((2, C), x2.device)
        dx = K.layernorm_bwd(dy2, x2, mean, rstd, gamma, dgb[0], dgb[1], dres=dr2)
        return dx.reshape(ctx.xshape), dgb[0], dgb[1], None


def layernorm_skip(x, gamma, beta, eps: float):
    """-> (LayerNorm(x), x') with x' an alias of x to be used by the residual / skip branch."""
    return LayerNormSkipFn.apply(x, gamma, beta, eps)


# ---------------------------------------------------------------------------------- attention
class AttentionFn(torch.autograd.Function):
    """dropout(softmax(scale q k^T + kbias)) v on [B, N, H, D] views (BERT self-attention, nn.MultiheadAttention core)."""

    @staticmethod
    def forward(ctx, q, k, v, kbias, scale, drop):
        p, seed, step = drop
        o, lse = K.attn_fwd(q, k, v, scale=scale, kbias=kbias, drop_p=p, drop_seed=seed, drop_step=step)
        ctx.save_for_backward(q, k, v, o, lse, kbias)
        ctx.scale, ctx.drop = scale, drop
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, kbias = ctx.saved_tensors
        do = _c(do)
        p, seed, step = ctx.drop
        dq, dk, dv = K.attn_bwd(q, k, v, o, do, lse, scale=ctx.scale, kbias=kbias, drop_p=p, drop_seed=seed, drop_step=step)
        return dq, _as_dtype(dk, q.dtype), _as_dtype(dv, q.dtype), None, None, None


def attention(q, k, v, scale: float, kbias=None, dropout_p: float = 0.0, training: bool = False):
    """`dropout_p`: attention-probability dropout (nn.MultiheadAttention(dropout=), HF attention_probs_dropout_prob),
    active in training mode only; the keep mask is drawn inside the flash kernels from the same seeding scheme as
    `dropout()` (host call-site seed + device step counter) and regenerated in backward."""
    drop = (0.0, 0, None)
    if training and dropout_p > 0.0:
        seed, step = _dropout_seed(q.device)
        drop = (float(dropout_p), seed, step)
    return AttentionFn.apply(q, k, v, kbias, scale, drop)


# --------------------------------------------------------------------- MViT pooled attention core
def relpos_index(q_size: int, k_size: int) -> torch.Tensor:
    """dist = arange(q)*q_ratio - (arange(k) + (1-k))*k_ratio, truncated (torchvision mvit._add_rel_pos)."""
    q_ratio = max(k_size / q_size, 1.0)
    k_ratio = max(q_size / k_size, 1.0)
    dist = torch.arange(q_size)[:, None] * q_ratio - (torch.arange(k_size)[None, :] + (1.0 - k_size)) * k_ratio
    return dist.long().to(torch.int32).contiguous()


def relpos_aug_width(k_thw) -> int:
    need = k_thw[0] + k_thw[1] + k_thw[2]
    return 32 * ((need + 31) // 32)


_BRANCH_STREAMS: dict = {}
_USE_BRANCH_STREAMS = __import__("os").environ.get("DCC_BRANCH_STREAMS", "1") != "0"
_BRANCH_STREAMS_FWD = __import__("os").environ.get("DCC_BRANCH_STREAMS_FWD", "0") == "1"


class _Branches:
    """Run the independent q / k / v pooling branches of an MViT block on three HIP streams (the current one + two
    cached side streams): their kernels are bound by memory latency at modest occupancy, so they overlap almost for
    free.  Fork: the side streams wait for the current stream; join: the current stream waits for them."""

    def __init__(self, device, enabled: bool = True):
        self.main = torch.cuda.current_stream(device)
        key = str(device)
        if key not in _BRANCH_STREAMS:
            _BRANCH_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(2)]
        self.side = _BRANCH_STREAMS[key] if (_USE_BRANCH_STREAMS and enabled) else []
        for s in self.side:
            s.wait_stream(self.main)

    def stream(self, i: int):
        return torch.cuda.stream(self.side[i - 1]) if (i > 0 and self.side) else torch.cuda.stream(self.main)

    def join(self):
        for s in self.side:
            self.main.wait_stream(s)


# conv pool + LayerNorm of the q/k/v branches in one launch where dcc_pool_ln_fwd applies (bf16, head_dim 96);
# DCC_FUSED_POOL_LN=0 keeps the two-kernel path for A/B runs
FUSED_POOL_LN = os.environ.get("DCC_FUSED_POOL_LN", "1") != "0"
FUSED_KAUG = os.environ.get("DCC_FUSED_KAUG", "1") != "0"   # the k pool kernel writes the rel-pos one-hot key columns too
FUSED_KV_PAIR = os.environ.get("DCC_FUSED_KV_PAIR", "1") != "0"   # k and v pools of a block in one launch (forward, single stream)
# forward rel-pos inner products per 64-row tile on the matrix cores, P never in HBM (dcc_relpos_aug_fwd_mfma, bf16);
# DCC_FUSED_RELPOS=0 keeps the stacked-table GEMM + gather
FUSED_RELPOS = os.environ.get("DCC_FUSED_RELPOS", "1") != "0"
# rel-pos backward as one MFMA kernel that builds dP per 64-row tile in LDS (dcc_relpos_aug_bwd) instead of scatter + two GEMMs
FUSED_RELPOS_BWD = os.environ.get("DCC_FUSED_RELPOS_BWD", "1") != "0"
# ... with the query pool's LayerNorm backward in its epilogue (dcc_relpos_aug_bwd_ln; without RoPE, whose inverse rotation sits
# between the two).  OFF by default — it wins isolated and loses in the step — DCC_FUSED_RELPOS_LN=1 enables
# it; tests/test_relpos_fused_gpu.py exercises the kernel in the default GPU suite either way
FUSED_RELPOS_LN = os.environ.get("DCC_FUSED_RELPOS_LN", "0") != "0"
# depthwise-conv weight gradients of the q/k/v pools on the parameter-gradient side stream (joined at the end of backward)
POOL_WGRAD_SIDE = os.environ.get("DCC_POOL_WGRAD_SIDE", "1") != "0"
PARAM_SIDE_OK = False   # set by the runner around backward: one rank, no accumulation, no gradient hooks that would read early


class MvitAttnCoreFn(torch.autograd.Function):
    """qkv -> Pool(q,k,v) (dw conv3d + LayerNorm) -> [3-D RoPE] -> rel-pos tables -> flash attention (+ residual pool).

    Follows torchvision MultiscaleAttention.forward as called from models/video_encoder.py:232-253 (non-RoPE) and the
    patched forward of models/video_encoder.py:268-378 (RoPE: rotation after pooling, no residual shortcut).
    Input qkv: [B*N, 3*H*D] (output of the qkv GEMM); output: [B*Nq, H*D].
    """

    @staticmethod
    def forward(ctx, qkv, wq, wk, wv, gq, bq, gk, bk, gv, bv, rel_h, rel_w, rel_t, cfg):
        B, N, H, D = cfg["B"], cfg["N"], cfg["H"], cfg["D"]
        thw, eps = cfg["thw"], cfg["eps"]
        dt = qkv.dtype
        qkv5 = qkv.view(B, N, 3, H, D)
        geo = {}
        conv = {}
        for name, stride in (("q", cfg["stride_q"]), ("k", cfg["stride_kv"]), ("v", cfg["stride_kv"])):
            geo[name] = K.PoolGeom(B, H, D, tuple(thw), (3, 3, 3), tuple(stride), (1, 1, 1), True)
        q_thw, k_thw = geo["q"].out_thw, geo["k"].out_thw
        Nq, Nk = geo["q"].n_out, geo["k"].n_out
        A = relpos_aug_width(k_thw)
        if A > 64:  # k_t + k_h + k_w > 64 (e.g. 32 frames x 448^2): the flash kernels are built for D + 32 / D + 64 columns
            raise NotImplementedError(f"key grid {k_thw}: relative-position augmentation of {A} columns exceeds the attention "
                                      "kernels' 64 (head_dim 96 + 32 / 64); reduce the clip size or the key stride")
        DQK = D + A
        Qf = torch.empty((B, Nq, H, DQK), device=qkv.device, dtype=dt)
        Kf = torch.empty((B, Nk, H, DQK), device=qkv.device, dtype=dt)
        Vn = torch.empty((B, Nk, H, D), device=qkv.device, dtype=dt)
        stats = {}
        kaug_done = False
        # forward: the three fused pool + LayerNorm kernels are short now (24-47 us at stage 2) and the fork / join events cost more
        # than the overlap returns (VideoEncoder forward 13.13 -> 12.87 ms without side streams); the backward keeps them
        br = _Branches(qkv.device, enabled=_BRANCH_STREAMS_FWD)
        pair_done = False
        if (FUSED_POOL_LN and FUSED_KV_PAIR and not br.side and K.pool_ln_supported(qkv5[:, :, 1], geo["k"])):
            # k and v pools (same geometry, same strides) as ONE grid: each alone under-fills the chip at stage 2
            pair = K.pool_ln_fwd_pair((qkv5[:, :, 1], qkv5[:, :, 2]), geo["k"], (wk.detach().reshape(D, 27), wv.detach().reshape(D, 27)),
                                      (gk.detach(), gv.detach()), (bk.detach(), bv.detach()), eps,
                                      (Kf.view(-1, DQK)[:, :D], Vn.view(-1, D)), augs=(A if FUSED_KAUG else 0, 0))
            if pair is not None:
                (conv["k"], mk_, rk_), (conv["v"], mv_, rv_) = pair
                stats["k"], stats["v"] = (mk_, rk_), (mv_, rv_)
                kaug_done = FUSED_KAUG
                pair_done = True
        for i, (name, w, gam, bet, dst) in enumerate((("q", wq, gq, bq, Qf.view(-1, DQK)[:, :D]),
                                                      ("k", wk, gk, bk, Kf.view(-1, DQK)[:, :D]),
                                                      ("v", wv, gv, bv, Vn.view(-1, D)))):
            if pair_done and name != "q":
                continue
            with br.stream(i):
                fused = None
                if FUSED_POOL_LN and K.pool_ln_supported(qkv5[:, :, i], geo[name]):
                    fused = K.pool_ln_fwd(qkv5[:, :, i], geo[name], w.detach().reshape(D, 27), gam.detach(), bet.detach(), eps, dst,
                                          aug=A if (name == "k" and FUSED_KAUG) else 0)   # the k pool writes the one-hot key augmentation too
                    kaug_done = kaug_done or (name == "k" and FUSED_KAUG and fused is not None)
                if fused is not None:
                    conv[name], m_, r_ = fused
                else:
                    conv[name], _ = K.pool_fwd(qkv5[:, :, i], geo[name], w.detach().reshape(D, 27))
                    _, m_, r_ = K.layernorm_fwd(conv[name].view(-1, D), gam.detach(), bet.detach(), eps, out=dst)
                stats[name] = (m_, r_)
        br.join()
        (mq, rq), (mk, rk), (mv, rv) = stats["q"], stats["k"], stats["v"]
        rope = cfg.get("rope")
        if rope is not None:
            K.rope(Qf[..., :D], Qf[..., :D], rope["cos_q"], rope["sin_q"])
            K.rope(Kf[..., :D], Kf[..., :D], rope["cos_k"], rope["sin_k"])
        scale = cfg["scale"]
        idx, inv = cfg["idx"], cfg["inv"]
        # rel-pos inner products: ONE GEMM of the (rotated) queries against the three tables stacked
        lens = (rel_h.shape[0], rel_w.shape[0], rel_t.shape[0])
        Lp = 8 * ((sum(lens) + 7) // 8)
        Rcat = stacked_tables((rel_h, rel_w, rel_t), Lp, D, dt)
        dims_q = K.relpos_dims(B, H, D, A, Lp, q_thw, k_thw, lens)
        fused_rel = FUSED_RELPOS and dt == torch.bfloat16 and D == 96
        if fused_rel:  # P of a 64-row tile lives in LDS only; same roundings as GEMM + gather
            fused_rel = K.relpos_aug_fwd_mfma(Qf, Rcat, idx, dims_q, scale)
        if not fused_rel:
            q2d = Qf.view(-1, DQK)[:, :D]
            P = K.gemm(q2d, Rcat)
            K.relpos_gather(Qf, P, idx, dims_q, scale)
            del P
        if not kaug_done:
            K.relpos_kaug(Kf, dims_q)
        residual = rope is None
        O, lse = K.attn_fwd(Qf, Kf, Vn, scale=scale, residual=residual, res_start=1)
        ctx.save_for_backward(qkv, conv["q"], conv["k"], conv["v"], mq, rq, mk, rk, mv, rv, Qf, Kf, Vn, O, lse,
                              wq.detach(), wk.detach(), wv.detach(), gq.detach(), gk.detach(), gv.detach(), Rcat)
        ctx.cfg, ctx.geo, ctx.A, ctx.residual, ctx.lens, ctx.Lp = cfg, geo, A, residual, lens, Lp
        ctx.pool_params = (wq, wk, wv)
        return O.view(B * Nq, H * D)

    @staticmethod
    def backward(ctx, dO):
        (qkv, cq, ck, cv, mq, rq, mk, rk, mv, rv, Qf, Kf, Vn, O, lse, wq, wk, wv, gq, gk, gv, Rcat) = ctx.saved_tensors
        cfg, geo, A = ctx.cfg, ctx.geo, ctx.A
        B, N, H, D = cfg["B"], cfg["N"], cfg["H"], cfg["D"]
        dt = qkv.dtype
        dev = qkv.device
        Nq, Nk = geo["q"].n_out, geo["k"].n_out
        DQK = D + A
        scale = cfg["scale"]
        dO4 = _c(dO).view(B, Nq, H, D)
        dQf, dKf, dVn = K.attn_bwd(Qf, Kf, Vn, O, dO4, lse, scale=scale, residual=ctx.residual, res_start=1)
        f32 = dict(device=dev, dtype=torch.float32)
        lens, Lp = ctx.lens, ctx.Lp
        rope = cfg.get("rope")
        dqkv = torch.empty((B, N, 3, H, D), device=dev, dtype=dt)
        grads = {}
        small = zeros_f32((3, 2, D), dev)  # LayerNorm gamma/beta gradients of the q, k, v pools
        small_w = zeros_f32((3, D, 27), dev)
        qkv5 = qkv.view(B, N, 3, H, D)
        ctx.pool_w_unset = [grads_unset(p_) for p_ in ctx.pool_params]

        def pool_branch(name, i, dy4, conv, mean, rstd, gam, w, dconv=None):
            """LayerNorm backward -> pool data gradient + pool weight gradient of one of the q / k / v branches
            (dconv given: the LayerNorm backward already happened in the producer of dy)."""
            g = geo[name]
            rows = B * g.n_out * H
            dgam, dbet = small[i, 0], small[i, 1]
            dw = small_w[i]
            if dconv is None:
                if dy4.is_contiguous():
                    dy2 = dy4.view(rows, D)
                else:  # dk is a column slice of the [.., DQK] buffer: rows are strided by DQK
                    dy2 = dy4.as_strided((rows, D), (dy4.stride(2), 1), dy4.storage_offset())
                dconv = K.layernorm_bwd(dy2, conv.view(rows, D), mean, rstd, gam, dgam, dbet)
            dconv4 = dconv.view(B, g.n_out, H, D)
            K.pool_bwd_data(dconv4, dqkv[:, :, i], g, w.reshape(D, 27))
            # the conv weight gradient feeds nothing but the optimizer: off the data path when autograd will adopt the buffer
            # as is (no accumulate kernel on another stream), else in line
            if not (POOL_WGRAD_SIDE and PARAM_SIDE_OK and ctx.pool_w_unset[i]
                    and side_param_work(lambda: K.dwconv_bwd_weight(dconv4, qkv5[:, :, i], g, dw), dconv, qkv, small_w)):
                K.dwconv_bwd_weight(dconv4, qkv5[:, :, i], g, dw)
            grads[name] = (dw.view(D, 1, 3, 3, 3), dgam, dbet)

        # the k and v branches only need dKf / dVn: they start on their own streams right away, underneath the
        # query side's rel-pos scatter and its two GEMMs
        br = _Branches(dev)
        with br.stream(1):
            dk = dKf[..., :D]  # the one-hot columns are constants
            if rope is not None:
                dk_d = torch.empty((B, Nk, H, D), device=dev, dtype=dKf.dtype)
                K.rope(dk, dk_d, rope["cos_k"], rope["sin_k"], backward=True)
                dk = dk_d
            pool_branch("k", 1, dk, ck, mk, rk, gk, wk)
        with br.stream(2):
            pool_branch("v", 2, dVn, cv, mv, rv, gv, wv)
        # queries: dQf[:, :D] already holds attention path + residual shortcut; the table path goes through
        # dP (scatter of the augmentation columns), then two GEMMs: d tables = dP^T q, dq = dP Rcat + dQf[:, :D]
        dims_q = K.relpos_dims(B, H, D, A, Lp, geo["q"].out_thw, geo["k"].out_thw, lens)
        rows_q = B * Nq * H
        dq = None
        dconv_q = None
        # dP built and consumed in LDS (no [rows, L] tensor): faster than scatter + two GEMMs at every block geometry
        # (tools/relpos_bench.py: 584 -> 271 us in block 0, 71 -> 43 us in the stage-2 blocks)
        if FUSED_RELPOS_BWD and dt == torch.bfloat16 and D == 96:
            drel_cat = zeros_f32((Lp, D), dev)
            # tools/relpos_bench.py (32 clips): blk1 150 -> 137 us, blk2 134 -> 116, blk3 84 -> 68, stage 2 75 -> 60, blk14 57 -> 45,
            # blk15 52 -> 48; block 0 (240 table rows: one workgroup per CU, the epilogue's VALU work is exposed) 360 -> 376: not there
            if FUSED_RELPOS_LN and rope is None and cq.is_contiguous() and Lp <= 128:
                dq = dconv_q = K.relpos_aug_bwd(dQf, Qf, Rcat, cfg["idx"], dims_q, scale, drel_cat,
                                                ln=(cq, mq, rq, gq, small[0, 0], small[0, 1]))
            else:
                dq = K.relpos_aug_bwd(dQf, Qf, Rcat, cfg["idx"], dims_q, scale, drel_cat)
        if dq is None:
            dP = torch.empty((rows_q, Lp), device=dev, dtype=dt)
            K.relpos_scatter(dQf, dP, cfg["inv"], dims_q, scale)
            q2d = Qf.view(-1, DQK)[:, :D]
            drel_cat = wgrad(dP, q2d)
            dq = K.gemm(dP, Rcat, b_km=True, resid=dQf.view(-1, DQK)[:, :D]).view(B, Nq, H, D)
            del dP
        drel = (drel_cat[:lens[0]], drel_cat[lens[0]:lens[0] + lens[1]], drel_cat[lens[0] + lens[1]:sum(lens)])
        if rope is not None:
            K.rope(dq, dq, rope["cos_q"], rope["sin_q"], backward=True)
        pool_branch("q", 0, dq, cq, mq, rq, gq, wq, dconv=dconv_q)
        br.join()
        return (dqkv.view(B * N, 3 * H * D), grads["q"][0], grads["k"][0], grads["v"][0], grads["q"][1], grads["q"][2],
                grads["k"][1], grads["k"][2], grads["v"][1], grads["v"][2], drel[0], drel[1], drel[2], None)


def mvit_attention_core(qkv, wq, wk, wv, gq, bq, gk, bk, gv, bv, rel_h, rel_w, rel_t, cfg):
    return MvitAttnCoreFn.apply(qkv, wq, wk, wv, gq, bq, gk, bk, gv, bv, rel_h, rel_w, rel_t, cfg)


# ----------------------------------------------------------------------------- skip-path max pool
class PoolSkipFn(torch.autograd.Function):
    """torchvision Pool(MaxPool3d(k=(1,3,3), s=(1,2,2), p=(0,1,1))) on [B*N, C] tokens, class token bypass."""

    @staticmethod
    def forward(ctx, x, B, thw, kernel, stride, padding):
        C = x.shape[1]
        g = K.PoolGeom(B, 1, C, tuple(thw), tuple(kernel), tuple(stride), tuple(padding), True)
        y, am = K.pool_fwd(x.view(B, g.n_in, 1, C), g, None, want_argmax=True)
        ctx.save_for_backward(am)
        ctx.g = g
        return y.view(B * g.n_out, C)

    @staticmethod
    def backward(ctx, dy):
        (am,) = ctx.saved_tensors
        g = ctx.g
        dy4 = _c(dy).view(g.B, g.n_out, 1, g.C)
        dx = torch.empty((g.B, g.n_in, 1, g.C), device=dy.device, dtype=dy.dtype)
        K.pool_bwd_data(dy4, dx, g, None, am)
        return dx.view(g.B * g.n_in, g.C), None, None, None, None, None


def pool_skip(x, B, thw, kernel, stride, padding):
    return PoolSkipFn.apply(x, B, thw, kernel, stride, padding)


# -------------------------------------------------------------------------------- patch embedding
class PatchEmbedFn(torch.autograd.Function):
    """conv_proj (Conv3d 3->C, k(3,7,7) s(2,4,4) p(1,3,3)) + flatten/transpose + class-token concat
    (models/video_encoder.py:237-241) as im2col + batched MFMA GEMM writing rows 1.. of each clip."""

    @staticmethod
    def forward(ctx, video, weight, bias, cls_token, dtype, norm=None):
        # video: [B', T, H, W, Cin] fp32 (NDHWC memory), or uint8 raw pixels with norm = (mean, std)
        Bp = video.shape[0]
        Cout, Cin, kt, kh, kw = weight.shape
        stride, padding = (2, 4, 4), (1, 3, 3)
        Kc = Cin * kt * kh * kw
        Kp = 32 * ((Kc + 31) // 32)
        col = K.im2col3d(_c(video), (kt, kh, kw), stride, padding, Kp, dtype, norm)
        L_ = col.shape[0] // Bp
        # weight in (kt,kh,kw,cin) column order, zero padded (tiny host-side view op on 42k elements)
        w2 = torch.zeros((Cout, Kp), device=video.device, dtype=torch.float32)
        w2[:, :Kc] = weight.detach().permute(0, 2, 3, 4, 1).reshape(Cout, Kc)
        wc = _as_dtype(w2, dtype)
        out = torch.empty((Bp, L_ + 1, Cout), device=video.device, dtype=dtype)
        K.gemm_batched(col, wc, out[:, 1:], M=L_, N=Cout, K=Kp, lda=Kp, ldb=Kp, ldc=Cout, batch=Bp, sA=L_ * Kp, sB=0,
                       sC=(L_ + 1) * Cout, bias=bias.detach())
        K.broadcast_row(_c(cls_token.detach()), out, (L_ + 1) * Cout, Bp)
        ctx.save_for_backward(col)
        ctx.meta = (Bp, L_, Cout, Cin, kt, kh, kw, Kc, Kp)
        return out.view(Bp * (L_ + 1), Cout)

    @staticmethod
    def backward(ctx, dy):
        (col,) = ctx.saved_tensors
        Bp, L_, Cout, Cin, kt, kh, kw, Kc, Kp = ctx.meta
        dy3 = _c(dy).view(Bp, L_ + 1, Cout)
        dw = db = dcls = None
        f32 = dict(device=dy.device, dtype=torch.float32)
        dcls_sum = K.colsum(dy3.view(Bp, (L_ + 1) * Cout)[:, :Cout])
        if ctx.needs_input_grad[3]:
            dcls = dcls_sum
        if ctx.needs_input_grad[1]:
            # dW[c][k] = sum_b sum_l dy[b][1+l][c] * col[b][l][k]: batched k-major GEMM, atomically accumulated
            dw2 = torch.zeros((Cout, Kp), **f32)
            K.gemm_batched(dy3[:, 1:], col, dw2, M=Cout, N=Kp, K=L_, lda=Cout, ldb=Kp, ldc=Kp, batch=Bp,
                           sA=(L_ + 1) * Cout, sB=L_ * Kp, sC=0, a_km=True, b_km=True, accumulate=True,
                           splitk=max(1, min(64, 1024 // max(1, Bp))))
            dw = dw2[:, :Kc].reshape(Cout, kt, kh, kw, Cin).permute(0, 4, 1, 2, 3).contiguous()
        if ctx.needs_input_grad[2]:
            db_all = K.colsum(dy3.view(Bp * (L_ + 1), Cout))
            db = K.axpby(db_all, dcls_sum, 1.0, -1.0)
        return None, dw, db, dcls, None, None


def patch_embed(video, weight, bias, cls_token, dtype, norm=None):
    return PatchEmbedFn.apply(video, weight, bias, cls_token, dtype, norm)


# ----------------------------------------------------------------------------------- small ops
class TokenMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.n = x.shape[1]
        return K.token_mean(_c(x))

    @staticmethod
    def backward(ctx, dy):
        return K.token_mean(_c(dy), backward=True, n_tokens=ctx.n)


def token_mean(x):
    return TokenMeanFn.apply(x)


class ClassTokenFn(torch.autograd.Function):
    """x [B, L, C] -> x[:, 0] as [B, 1, C] (torchvision MViT.forward's `x = x[:, 0]` behind the final norm: the pooled-CLS
    fallback of backbone="mvit_rope", models/video_encoder.py:552-561); backward scatters into a zero tensor."""

    @staticmethod
    def forward(ctx, x):
        B, L, C = x.shape
        ctx.shape = (B, L, C)
        x = _c(x)
        out = torch.empty((B, C), device=x.device, dtype=x.dtype)
        K.cast2d(x.view(B, L * C)[:, :C], out)
        return out.view(B, 1, C)

    @staticmethod
    def backward(ctx, d):
        B, L, C = ctx.shape
        dx = torch.zeros((B, L * C), device=d.device, dtype=d.dtype)
        K.cast2d(_c(d).view(B, C), dx[:, :C])
        return dx.view(B, L, C)


def class_token(x):
    return ClassTokenFn.apply(x)


class QPoolFn(torch.autograd.Function):
    """softmax(<q, x_n>) weighted sum over n (EnhancedVideoAggregator tail), fp32."""

    @staticmethod
    def forward(ctx, x, q):
        x = _c(x)
        out, w = K.qpool_fwd(x, _c(q.detach()).view(-1))
        ctx.save_for_backward(x, q.detach(), w)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, q, w = ctx.saved_tensors
        dx, dq = K.qpool_bwd(x, _c(q).view(-1), w, _c(dout))
        return dx, dq.view(q.shape)


def qpool(x, q):
    return QPoolFn.apply(x, q)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.axpby(_c(a), _c(b))

    @staticmethod
    def backward(ctx, d):
        return d, d


def add(a, b):
    return AddFn.apply(a, b)


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = K.relu(_c(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, d):
        (y,) = ctx.saved_tensors
        return K.relu(_c(d), y)


def relu(x):
    return ReluFn.apply(x)


class PrependTokenFn(torch.autograd.Function):
    """cat([token.expand(B, 1, D), x], dim=1) for x [B, N, D] and a learnable token [1, 1, D] (AttentionPoolWithCLS)."""

    @staticmethod
    def forward(ctx, token, x):
        B, N, D = x.shape
        out = torch.empty((B, N + 1, D), device=x.device, dtype=x.dtype)
        K.broadcast_row(_c(token.detach().float()).view(-1), out, (N + 1) * D, B)
        K.cast2d(_c(x).view(B, N * D), out.view(B, (N + 1) * D)[:, D:])
        ctx.shape = (B, N, D)
        ctx.tshape = token.shape
        return out

    @staticmethod
    def backward(ctx, d):
        B, N, D = ctx.shape
        d = _c(d)
        d2 = d.view(B, (N + 1) * D)
        dx = torch.empty((B, N * D), device=d.device, dtype=d.dtype)
        K.cast2d(d2[:, D:], dx)
        dtok = K.colsum(d2[:, :D]).view(ctx.tshape)
        return dtok, dx.view(B, N, D)


def prepend_token(token, x):
    return PrependTokenFn.apply(token, x)


class GateFn(torch.autograd.Function):
    """tanh(v) * sigmoid(u): gated-attention features of the MIL head."""

    @staticmethod
    def forward(ctx, v, u):
        v, u = _c(v), _c(u)
        ctx.save_for_backward(v, u)
        return K.gate_fwd(v, u)

    @staticmethod
    def backward(ctx, d):
        v, u = ctx.saved_tensors
        return K.gate_bwd(_c(d), v, u)


class MaskedSoftmaxFn(torch.autograd.Function):
    """softmax over dim 1 of fp32 scores [B, N] restricted to mask != 0 (uint8 [B, N] or None)."""

    @staticmethod
    def forward(ctx, s, mask):
        w = K.masked_softmax_fwd(_c(s), mask)
        ctx.save_for_backward(w)
        return w

    @staticmethod
    def backward(ctx, dw):
        (w,) = ctx.saved_tensors
        return K.masked_softmax_bwd(w, _c(dw)), None


class WeightedSumFn(torch.autograd.Function):
    """sum_n w[b, n] x[b, n, :]."""

    @staticmethod
    def forward(ctx, w, x):
        w, x = _c(w), _c(x)
        ctx.save_for_backward(w, x)
        return K.wsum_fwd(w, x)

    @staticmethod
    def backward(ctx, d):
        w, x = ctx.saved_tensors
        dx, dw = K.wsum_bwd(w, x, _c(d))
        return dw, dx


class MaskedReduceFn(torch.autograd.Function):
    """masked mean (mode 0) / masked max (mode 1) over dim 1 of [B, N, D]."""

    @staticmethod
    def forward(ctx, x, mask, mode):
        x = _c(x)
        out, arg = K.masked_reduce_fwd(x, mask, mode)
        ctx.mask, ctx.arg, ctx.N, ctx.mode = mask, arg, x.shape[1], mode
        return out

    @staticmethod
    def backward(ctx, d):
        return K.masked_reduce_bwd(_c(d), ctx.mask, ctx.arg, ctx.N, ctx.mode), None, None


class ViewEmbedAddFn(torch.autograd.Function):
    """x + table[ids] broadcast over the patch axis (models/multi_instance_linear_probing.py:265-269)."""

    @staticmethod
    def forward(ctx, x, ids, table):
        ctx.save_for_backward(ids)
        ctx.classes = table.shape[0]
        return K.view_embed_add(x, ids, _c(table.detach()))

    @staticmethod
    def backward(ctx, d):
        (ids,) = ctx.saved_tensors
        d = _c(d)
        return d, None, (K.view_embed_bwd(d, ids, ctx.classes) if ctx.needs_input_grad[2] else None)


def view_embed_add(x, ids, table):
    """x: [rows, L, D] fp32; ids: int64 [rows]; table: [classes, D] fp32."""
    return ViewEmbedAddFn.apply(x, ids, table)


def gate(v, u):
    return GateFn.apply(v, u)


def masked_softmax(s, mask=None):
    return MaskedSoftmaxFn.apply(s, mask)


def weighted_sum(w, x):
    return WeightedSumFn.apply(w, x)


def masked_mean(x, mask=None):
    return MaskedReduceFn.apply(x, mask, 0)


def masked_max(x, mask=None):
    return MaskedReduceFn.apply(x, mask, 1)


class StochasticDepthFn(torch.autograd.Function):
    """out = resid + branch * scale[sample]  (torchvision StochasticDepth, mode "row", fused with the residual add)."""

    @staticmethod
    def forward(ctx, resid, branch, scale, samples):
        ctx.save_for_backward(scale)
        ctx.samples = samples
        return K.scale_rows(_c(branch), scale, samples, resid=_c(resid))

    @staticmethod
    def backward(ctx, d):
        (scale,) = ctx.saved_tensors
        d = _c(d)
        return d, K.scale_rows(d, scale, ctx.samples), None, None


def stochastic_depth_add(resid, branch, p: float, samples: int):
    """resid + StochasticDepth(p, "row")(branch) in training mode (caller handles eval / p == 0 with the fused
    GEMM residual instead).  One Bernoulli(1-p)/(1-p) factor per clip, drawn from torch's device generator."""
    keep = 1.0 - p
    scale = torch.bernoulli(torch.full((samples,), keep, device=branch.device, dtype=torch.float32)) / keep
    return StochasticDepthFn.apply(resid, branch, scale, samples)


class InterpRowsFn(torch.autograd.Function):
    """table' = W table with a fixed [d, L] interpolation matrix (torchvision mvit._interpolate: F.interpolate(linear,
    align_corners=False) of a relative-position table whose length does not match the token grid)."""

    @staticmethod
    def forward(ctx, table, W):
        ctx.save_for_backward(W)
        return K.gemm(W, _c(table.detach()).float(), b_km=True)  # [d, L] x [L, C]

    @staticmethod
    def backward(ctx, d):
        (W,) = ctx.saved_tensors
        return K.gemm(W, _c(d).float(), a_km=True, b_km=True), None  # W^T d: [L, d] x [d, C]


def interp_matrix(length_in: int, length_out: int) -> torch.Tensor:
    """Rows of F.interpolate(mode="linear", align_corners=False): out[i] = (1-l) in[i0] + l in[i1]."""
    W = torch.zeros(length_out, length_in, dtype=torch.float32)
    scale = length_in / length_out
    for i in range(length_out):
        src = max((i + 0.5) * scale - 0.5, 0.0)
        i0 = min(int(src), length_in - 1)
        i1 = min(i0 + 1, length_in - 1)
        lam = src - i0
        W[i, i0] += 1.0 - lam
        W[i, i1] += lam
    return W


def interp_rows(table, W):
    return InterpRowsFn.apply(table, W)


class CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return _as_dtype(x, dtype)

    @staticmethod
    def backward(ctx, d):
        return _as_dtype(_c(d), ctx.src), None


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    return CastFn.apply(x, dtype)


_DROPOUT_CALLS = 0
# Training-step scope (set by the runner): inside it a dropout call's seed is (base seed, position of the call within the
# step) on the host plus a DEVICE step counter the kernel mixes in — nothing per-step is baked into the launch arguments,
# so a step captured in a HIP graph draws new masks at every replay (the runner advances the counter inside the graph).
_STEP = {"counter": None, "calls": 0}


def begin_step(counter: torch.Tensor | None) -> None:
    """counter: device int64 [1] advanced once per step by the caller, or None to leave the step scope."""
    _STEP["counter"], _STEP["calls"] = counter, 0


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, step):
        ctx.p, ctx.seed, ctx.step = p, seed, step
        return K.dropout(_c(x), p, seed, step)

    @staticmethod
    def backward(ctx, d):
        return K.dropout(_c(d), ctx.p, ctx.seed, ctx.step), None, None, None


def _dropout_seed(device):
    """(host seed, device step counter or None) of the next mask-drawing call site (see `begin_step`)."""
    global _DROPOUT_CALLS
    step = _STEP["counter"]
    if step is not None and step.device == device:
        _STEP["calls"] += 1
        return (torch.initial_seed() * 1000003 + _STEP["calls"]) & (2**63 - 1), step
    _DROPOUT_CALLS += 1
    return (torch.initial_seed() * 1000003 + _DROPOUT_CALLS) & (2**63 - 1), None


def dropout(x, p: float, training: bool):
    """nn.Dropout: identity in eval / p == 0; counter-hash keep mask otherwise (mask regenerated in backward)."""
    if not training or p <= 0.0:
        return x
    seed, step = _dropout_seed(x.device)
    return DropoutFn.apply(x, float(p), seed, step)


class EmbedFn(torch.autograd.Function):
    """BERT embeddings: word[ids] + position[s] + token_type[0] (fp32 tables -> compute dtype)."""

    @staticmethod
    def forward(ctx, ids, word, pos, typ, dtype):
        ids = _c(ids)
        out = K.embed_fwd(ids, word.detach(), pos.detach(), typ.detach(), dtype)
        ctx.save_for_backward(ids)
        ctx.shapes = (word.shape, pos.shape, typ.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        ws, ps, ts = ctx.shapes
        f32 = dict(device=dy.device, dtype=torch.float32)
        dword = torch.zeros(ws, **f32) if ctx.needs_input_grad[1] else None
        dpos = torch.zeros(ps, **f32) if ctx.needs_input_grad[2] else None
        dtyp = torch.zeros(ts, **f32) if ctx.needs_input_grad[3] else None
        K.embed_bwd(ids, _c(dy), dword, dpos, dtyp)
        return None, dword, dpos, dtyp, None


def embed(ids, word, pos, typ, dtype):
    return EmbedFn.apply(ids, word, pos, typ, dtype)


# ------------------------------------------------------------------- nn.MultiheadAttention core
def multihead_attention(query, key_value, in_w, in_b, out_w, out_b, num_heads: int, self_attn: bool,
                        dropout_p: float = 0.0, training: bool = False, kbias=None):
    """batch_first nn.MultiheadAttention forward (no masks) built from our GEMM + attention; `dropout_p` is the module's
    attention-weight dropout (applied to the softmax output in training mode, inside the flash kernels).

    query: [B, Lq, E]; key_value: [B, Lk, E] (key == value).  Weight slices are views of the packed in_proj
    parameters, so gradients flow back to `in_proj_weight` / `in_proj_bias` through autograd's slicing.
    """
    B, Lq, E = query.shape
    Lk = key_value.shape[1]
    hd = E // num_heads
    if self_attn:
        qkv = linear(query, in_w, in_b).view(B, Lq, 3, num_heads, hd)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q = linear(query, in_w[:E], in_b[:E]).view(B, Lq, num_heads, hd)
        kv = linear(key_value, in_w[E:], in_b[E:]).view(B, Lk, 2, num_heads, hd)
        k, v = kv[:, :, 0], kv[:, :, 1]
    # kbias: additive fp32 [B, Lk] key bias (key_padding_mask as 0 / -1e30)
    ctx = attention(q, k, v, 1.0 / math.sqrt(hd), kbias, dropout_p=dropout_p, training=training)
    return linear(ctx.reshape(B, Lq, E), out_w, out_b)
