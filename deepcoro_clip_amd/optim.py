"""Step tail on HIP kernels: global gradient-norm clipping and AdamW (reference: runners/
video_constrative_learning_runner.py:1398-1421 — `clip_grad_norm_` over both encoders, `optimizer.step()`).

`HipAdamW` is a `torch.optim.Optimizer` with AdamW's constructor / param_groups / state layout
(`step`, `exp_avg`, `exp_avg_sq`), so schedulers, checkpoints and the project's per-group learning rates work
unchanged; `adopt(optimizer)` wraps an existing `torch.optim.AdamW/Adam` instance the project already built.
One multi-tensor launch per phase, no host synchronisation (the clip coefficient stays on the device).
"""
from __future__ import annotations

import ctypes
from typing import Iterable

import numpy as np
import torch

from . import _lib as L
from . import functional as Fn

OPT_CHUNK = 4096  # DCC_OPT_CHUNK

_DESC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("shadow", "<u8"), ("n", "<i8"),
                  ("lr", "<f4"), ("beta1", "<f4"), ("beta2", "<f4"), ("eps", "<f4"), ("wd", "<f4"), ("bc1", "<f4"),
                  ("bc2", "<f4"), ("flags", "<i4")])
assert _DESC.itemsize == 80


class _TensorTable:
    """Device-side descriptor table + (tensor, chunk) work list for a fixed list of parameters."""

    def __init__(self, params: list[torch.Tensor]):
        self.params = params
        self.device = params[0].device
        work = []
        for i, p in enumerate(params):
            for c in range((p.numel() + OPT_CHUNK - 1) // OPT_CHUNK):
                work.append((i, c))
        self.nblocks = len(work)
        self.work = torch.tensor(work, dtype=torch.int32).reshape(-1, 2).to(self.device)
        nbytes = len(params) * _DESC.itemsize
        # two pinned staging buffers + events: a buffer is only rewritten after its previous H2D copy completed
        self.hosts = [torch.zeros(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.events = [None, None]
        self.turn = 0
        self.dev = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        # staging buffer of a captured step (HIP graph): the copy node re-reads it at every replay, so a replay driver
        # rewrites it (refill) after the previous replay has finished and before the next launch
        self.cap_host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
        self.captured = False

    def refill(self, desc: np.ndarray):
        self.cap_host.numpy().view(_DESC)[:] = desc

    def upload(self, desc: np.ndarray):
        if torch.cuda.is_current_stream_capturing():
            self.refill(desc)
            self.dev.copy_(self.cap_host, non_blocking=True)
            self.captured = True
            return
        i = self.turn
        self.turn ^= 1
        if self.events[i] is not None:
            self.events[i].synchronize()
        self.hosts[i].numpy().view(_DESC)[:] = desc
        self.dev.copy_(self.hosts[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.events[i] = ev


def _grad_ptr(p: torch.Tensor) -> int:
    g = p.grad
    if g is None:
        return 0
    if g.dtype != torch.float32 or not g.is_contiguous():
        raise L.DccError("HIP optimizer expects contiguous fp32 gradients")
    return g.data_ptr()


class GradClipper:
    """torch.nn.utils.clip_grad_norm_(params, max_norm) semantics: grads scaled in place by
    min(1, max_norm / (total_norm + 1e-6)); returns the total norm as a DEVICE tensor (no sync)."""

    def __init__(self, params: Iterable[torch.Tensor]):
        self.params = [p for p in params]
        L.require_cuda(*self.params[:1])
        self.table = _TensorTable(self.params)
        dev = self.table.device
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self.coef = torch.ones(1, device=dev, dtype=torch.float32)
        self.norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self._base: np.ndarray | None = None

    def _desc(self) -> np.ndarray:
        if self._base is None:
            self._base = np.zeros(len(self.params), dtype=_DESC)
            self._base["n"] = [p.numel() for p in self.params]
        d = self._base.copy()
        d["g"] = np.fromiter((_grad_ptr(p) for p in self.params), dtype=np.uint64, count=len(self.params))
        return d

    def clip_(self, max_norm: float, scale: bool = True) -> torch.Tensor:
        """scale=False only computes `self.coef` (device scalar): HipAdamW.step(grad_scale=coef, scaled=params) then
        applies it while it reads the gradients anyway, saving one read+write pass over all of them."""
        t = self.table
        t.upload(self._desc())
        self.sumsq.zero_()
        L.call("dcc_grad_sumsq", L.ptr(t.dev), L.ptr(t.work), t.nblocks, L.ptr(self.sumsq), L.stream())
        L.call("dcc_clip_coef", L.ptr(self.sumsq), float(max_norm), L.ptr(self.coef), L.ptr(self.norm), L.stream())
        if scale:
            L.call("dcc_scale_grads", L.ptr(t.dev), L.ptr(t.work), t.nblocks, L.ptr(self.coef), L.stream())
        return self.norm

    def scale_with(self, coef: torch.Tensor) -> None:
        """Scale this clipper's gradients in place by a coefficient computed elsewhere (device scalar)."""
        t = self.table
        t.upload(self._desc())
        L.call("dcc_scale_grads", L.ptr(t.dev), L.ptr(t.work), t.nblocks, L.ptr(coef), L.stream())


class HipAdamW(torch.optim.Optimizer):
    """AdamW (decoupled weight decay) / Adam (adamw=False) with torch's state layout, updated by `dcc_adam_step`."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 adamw: bool = True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, adamw=adamw)
        super().__init__(params, defaults)
        self._table: _TensorTable | None = None
        self._plist: list[torch.Tensor] = []
        self._plan: dict | None = None

    @classmethod
    def adopt(cls, optimizer: torch.optim.Optimizer) -> "HipAdamW":
        """Re-home the param_groups / state of an existing torch Adam/AdamW (same hyper-parameters, same state)."""
        if isinstance(optimizer, HipAdamW):
            return optimizer
        if not isinstance(optimizer, (torch.optim.AdamW, torch.optim.Adam)):
            raise L.DccError(f"HipAdamW.adopt supports torch.optim.Adam/AdamW, got {type(optimizer).__name__}")
        adamw = isinstance(optimizer, torch.optim.AdamW) or bool(optimizer.defaults.get("decoupled_weight_decay", False))
        for g in optimizer.param_groups:
            if g.get("amsgrad") or g.get("maximize"):
                raise L.DccError("amsgrad / maximize are not supported by the HIP AdamW")
        self = cls.__new__(cls)
        torch.optim.Optimizer.__init__(self, optimizer.param_groups, dict(optimizer.defaults, adamw=adamw))
        self.state = optimizer.state
        for g in self.param_groups:
            g.setdefault("adamw", adamw)
        self._table, self._plist, self._plan = None, [], None
        return self

    def _build(self):
        self._plist = [p for g in self.param_groups for p in g["params"]]
        L.require_cuda(*self._plist[:1])
        self._table = _TensorTable(self._plist)

    def _plan_build(self):
        """Per-parameter constants of the descriptor table, cached between steps (the table has ~500 rows and filling it
        row by row in Python cost 4 ms of host time per step).  Everything that can change between steps without the
        optimizer noticing is re-validated cheaply in step(): gradient / parameter addresses, state identity, shadows."""
        n = len(self._plist)
        base = np.zeros(n, dtype=_DESC)
        gidx = np.zeros(n, dtype=np.int64)
        steps, states, k0 = [], [], np.zeros(n, dtype=np.float64)
        i = 0
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if not p.is_contiguous():
                    raise L.DccError("HIP optimizer expects contiguous parameters")
                st = self.state.get(p)  # created on the first gradient (step()), like torch.optim.AdamW
                if st is not None and len(st) == 0:
                    st = None
                if st is not None:
                    if not torch.is_tensor(st["step"]):
                        st["step"] = torch.tensor(float(st["step"]))
                    base[i]["m"], base[i]["v"] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    k0[i] = float(st["step"])
                base[i]["p"], base[i]["n"] = p.data_ptr(), p.numel()
                gidx[i] = gi
                steps.append(st["step"] if st is not None else None)
                states.append((st, st["exp_avg"], st["exp_avg_sq"]) if st is not None else (None, None, None))
                i += 1
        self._plan = dict(base=base, gidx=gidx, steps=steps, states=states, k=k0,
                          pptr=[p.data_ptr() for p in self._plist])

    def _plan_valid(self) -> bool:
        pl = self._plan
        if pl is None or len(pl["pptr"]) != len(self._plist):
            return False
        for p, ptr, (st, m, v), stp in zip(self._plist, pl["pptr"], pl["states"], pl["steps"]):
            cur = self.state.get(p)
            if cur is not None and len(cur) == 0:
                cur = None
            if p.data_ptr() != ptr or cur is not st:
                return False
            if st is not None and (st["exp_avg"] is not m or st["exp_avg_sq"] is not v or st["step"] is not stp):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None, grad_scale: torch.Tensor | None = None, scaled: set | None = None):
        """grad_scale: device scalar multiplied into the gradients of the parameters whose id() is in `scaled`
        (all parameters when `scaled` is None) — the clip coefficient of GradClipper.clip_(scale=False)."""
        loss = closure() if closure is not None else None
        if self._table is None or len(self._plist) != sum(len(g["params"]) for g in self.param_groups):
            self._build()
            self._plan = None
        if not self._plan_valid():
            self._plan_build()
        pl = self._plan
        plist = self._plist
        gptr = np.fromiter((_grad_ptr(p) if (p.requires_grad and p.grad is not None) else 0 for p in plist), dtype=np.uint64,
                           count=len(plist))
        live = gptr != 0
        fresh = [i for i in np.nonzero(live)[0] if pl["steps"][i] is None]
        if fresh:  # a parameter got its first gradient: create its state, rebuild the cached rows
            for i in fresh:
                p = plist[i]
                st = self.state[p]
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self._plan_build()
            pl = self._plan
        idx = np.nonzero(live)[0]
        torch._foreach_add_([pl["steps"][i] for i in idx], 1.0)   # torch's per-parameter step tensors (checkpoint layout)
        pl["k"][idx] += 1.0
        d = pl["base"].copy()
        d["g"] = gptr
        groups = self.param_groups
        gi = pl["gidx"]
        b1 = np.array([g["betas"][0] for g in groups], dtype=np.float64)[gi]
        b2 = np.array([g["betas"][1] for g in groups], dtype=np.float64)[gi]
        d["lr"] = np.array([float(g["lr"]) for g in groups], dtype=np.float64)[gi]
        d["beta1"], d["beta2"] = b1, b2
        d["eps"] = np.array([g["eps"] for g in groups], dtype=np.float64)[gi]
        d["wd"] = np.array([g["weight_decay"] for g in groups], dtype=np.float64)[gi]
        k = np.maximum(pl["k"], 1.0)
        d["bc1"], d["bc2"] = 1.0 - b1 ** k, 1.0 - b2 ** k
        adamw = np.array([1 if g.get("adamw", True) else 0 for g in groups], dtype=np.int32)[gi]
        inclip = np.full(len(plist), 2, dtype=np.int32) if scaled is None else \
            np.fromiter((2 if id(p) in scaled else 0 for p in plist), dtype=np.int32, count=len(plist))
        d["flags"] = adamw | inclip
        # a whole-parameter bf16 shadow (functional.wcast) is refreshed by the update kernel itself
        refreshed: list = []
        for i in idx:
            p = plist[i]
            cache = p.__dict__.get("_dcc_shadows")
            if cache:
                skey = (0, tuple(p.shape), tuple(p.stride()), torch.bfloat16)
                hit = cache.get(skey)
                if hit is not None:
                    d[i]["shadow"] = hit[1].data_ptr()
                    refreshed.append((p, skey))
        dead = ~live
        if dead.any():  # rows without a gradient are skipped by the kernel (g == 0)
            for f in ("p", "m", "v", "shadow"):
                d[f][dead] = 0
        t = self._table
        t.upload(d)
        L.call("dcc_adam_step", L.ptr(t.dev), L.ptr(t.work), t.nblocks, L.ptr(grad_scale), L.stream())
        Fn.invalidate_weight_cache()  # parameters changed behind autograd's version counters
        for p, skey in refreshed:    # ... except the shadows the kernel just rewrote
            cache = p.__dict__["_dcc_shadows"]
            cache[skey] = ((p._version, Fn._WEPOCH, p.data_ptr()), cache[skey][1])
        return loss
