"""Where a parameter's gradient should be written when the owner of the gradient memory is not autograd.

Under HipDDP every `param.grad` lives in one flat fp32 arena.  Attaching the arena views before backward makes autograd
ACCUMULATE into them (one add kernel per parameter per step: ~500 launches and 3 passes over 607 MB).  Instead the
wrapper leaves `param.grad` unset and registers the views here; a gradient producer that can write anywhere (the
weight-gradient GEMMs, their fused bias column sums) asks `take(param)` for a fresh alias of the slot, writes into it
and returns it — AccumulateGrad then adopts that tensor as is (no kernel).  Producers that do not know about slots return
ordinary tensors; the wrapper's post-accumulate hook copies those into the arena (ddp.HipDDP._make_hook)."""
from __future__ import annotations

import torch

_SLOTS: dict[int, torch.Tensor] = {}
_TAKEN: set[int] = set()


def register(param: torch.Tensor, view: torch.Tensor) -> None:
    _SLOTS[id(param)] = view
    _TAKEN.discard(id(param))


def unregister(param: torch.Tensor) -> None:
    _SLOTS.pop(id(param), None)
    _TAKEN.discard(id(param))


def take(param) -> torch.Tensor | None:
    """A fresh alias of `param`'s zeroed gradient slot, once per backward, only while `param.grad` is unset (otherwise
    autograd will add the returned gradient to the existing one and must get an ordinary tensor)."""
    if param is None or not _SLOTS:
        return None
    k = id(param)
    v = _SLOTS.get(k)
    if v is None or k in _TAKEN or not param.is_leaf or param.grad is not None:
        return None
    _TAKEN.add(k)
    return v.view(v.shape)
