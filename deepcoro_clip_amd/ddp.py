"""Data-parallel communication for one process per GPU over RCCL/xGMI (reference: utils/ddp.py:11-136 and the two
`DistributedDataParallel` wrappers of projects/contrastive_pretraining_project.py:107-114).

`HipDDP` keeps the reference's wrapper contract (`.module`, `forward` passthrough, `device_ids=` kwarg) but is built
for the MI355X node rather than translating torch's reducer:
  * gradients live in ONE flat fp32 arena per wrapper; `param.grad` are views into it, so autograd accumulates in
    place and buckets need no copy-in / copy-out;
  * buckets are large (default 64 MiB): xGMI is point-to-point (7 links x ~153 GB/s), ring/direct collectives are
    per-link bound, so few large messages beat DDP's 25 MiB NVSwitch-era default;
  * a bucket's all-reduce (AVG) is launched from a post-accumulate-grad hook as soon as its last gradient is
    written — RCCL runs it on its own stream, overlapped with the rest of backward; `finish()` joins them;
  * host cost: the first reduced backward runs one Python hook per parameter and records the order in which gradients
    arrive; from then on ONE hook per bucket (on the parameter whose gradient arrived last) moves the bucket's stray
    gradients into the arena and starts its all-reduce — ~10 hook calls per step instead of ~500 on a step whose host
    enqueue time is the bottleneck.  A bucket whose expected gradients are not all there when its hook fires (the order
    changed: freezing, unused branches) is deferred to `finish()`; accumulation windows keep the per-parameter hooks;
  * gradient accumulation has torch DDP's `no_sync` semantics: `begin_backward(sync=False)` micro-steps only
    accumulate into the arena, the buckets are reduced once, from the hooks of the window's LAST backward
    (`begin_backward(sync=True)`, the default), so no collective ever overlaps a later in-place accumulate.
The same code path runs on gloo/CPU for the world_size-2 tests (SUM + divide, since gloo has no AVG).
"""
from __future__ import annotations

import os
from typing import Any

import torch
import torch.distributed as dist
import torch.multiprocessing as MP
import torch.utils.data.distributed as DS

from . import _grad_slots


class HipDDP(torch.nn.Module):
    def __init__(self, module: torch.nn.Module, device_ids=None, bucket_mb: float = 64.0, grad_wire_dtype=None, **_ignored):
        super().__init__()
        self.module = module
        # optional reduced-precision transport of the gradient buckets (north_star: "reduce-scatter of grads" in reduced
        # precision over xGMI): the fp32 bucket is rounded to bf16, all-reduced in bf16 (half the wire bytes: 306 instead of
        # 612 MB per step at the headline shapes) and written back to the fp32 arena.  Default None = fp32 on the wire.
        if grad_wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("grad_wire_dtype must be None, torch.float32 or torch.bfloat16")
        self._wire = torch.bfloat16 if grad_wire_dtype == torch.bfloat16 else None
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self._params = [p for p in module.parameters()]
        self._bucket_bytes = int(bucket_mb * (1 << 20))
        self._handles: list = []
        self._arena: torch.Tensor | None = None
        self._buckets: list[tuple[int, int]] = []
        self._pending: list[int] = []
        self._bucket_of: dict[int, int] = {}
        self._hooks_installed = False
        self._hook_handles: list = []
        self._mode = "learn"        # "learn": per-parameter hooks, arrival order recorded; "bucket": one hook per bucket;
        #                             "param": per-parameter hooks for good (accumulation windows)
        self._arrivals: list[int] = []
        self._expected: list[list] = []   # per bucket: the parameters whose gradient arrived in the learning backward
        self._deferred: set[int] = set()
        self._arrived: set[int] = set()   # bucket mode: parameters with an arrival hook that fired in this backward
        self._relearn = False             # bucket mode: a gradient outside the recorded order arrived -> learn the order again
        self.hook_calls = 0         # Python hook invocations (tests / diagnostics)
        self._sync = True  # reduce from this backward's hooks (False: accumulate-only micro-step of a window)
        # parameters whose gradient nobody zeroes (outside every optimizer group): their arena slot survives
        # attach_grads() and stays attached, so the history keeps accumulating as it does under torch DDP
        self.persistent_grad_ids: set[int] = set()
        if self.world > 1:
            self._broadcast_parameters()
        self._build_arena()

    # ---------------------------------------------------------------------------------------------- setup
    def _broadcast_parameters(self):
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=0)

    def _build_arena(self):
        """Flat gradient arena in REVERSE parameter order (roughly the order backward produces gradients)."""
        ps = [p for p in reversed(self._params)]
        if not ps:
            return
        dev, total = ps[0].device, 0
        offs = []
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4  # keep every view 16-byte aligned
        self._arena = torch.zeros(total, device=dev, dtype=torch.float32)
        self._views = {}
        start, b = 0, 0
        for p, o in zip(ps, offs):
            self._views[id(p)] = self._arena[o:o + p.numel()].view(p.shape)
            if (o + p.numel() - start) * 4 > self._bucket_bytes and o > start:
                self._buckets.append((start, o))
                start, b = o, b + 1
            self._bucket_of[id(p)] = b
        self._buckets.append((start, total))
        self._order = ps

    def attach_grads(self):
        """Zero the arena and hand its slots out for this step; call instead of optimizer.zero_grad(set_to_none=True).

        `param.grad` is left unset: producers that can write anywhere put the gradient straight into the parameter's slot
        (_grad_slots.take) and autograd adopts it without a kernel; any other gradient is copied into its slot by the
        post-accumulate hook.  Either way `param.grad` is a view into the flat arena when backward returns."""
        if self._arena is None:
            return
        keep = [(p, self._views[id(p)].clone()) for p in self._order
                if id(p) in self.persistent_grad_ids and p.grad is not None]
        self._arena.zero_()
        for p in self._order:
            p.grad = None
            if p.requires_grad and id(p) not in self.persistent_grad_ids:
                _grad_slots.register(p, self._views[id(p)])
            else:
                _grad_slots.unregister(p)
        for p, hist in keep:  # history back into its slot, gradient stays attached: autograd accumulates in place
            self._views[id(p)].copy_(hist)
            p.grad = self._views[id(p)]
        self._handles = []
        if not self._hooks_installed:
            self._install_param_hooks()
        self._fresh = True   # every gradient slot is empty: the precondition of the one-hook-per-bucket mode
        self.begin_backward(sync=True)

    def _install_param_hooks(self):
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = [p.register_post_accumulate_grad_hook(self._make_hook(p)) for p in self._order if p.requires_grad]
        self._req_sig = tuple(p.requires_grad for p in self._order)
        self._hooks_installed = True

    def _install_bucket_hooks(self):
        """One hook per bucket, on the parameter whose gradient arrived last in the recorded backward — plus a cheap arrival
        hook on every trainable parameter whose `.grad` cannot tell whether THIS backward's gradient has arrived:
          * parameters outside the recorded order (no gradient in the learning backward).  Their gradient may later be written
            straight into the arena slot (_grad_slots.take: data_ptr == view), which `finish()` cannot tell from "nothing
            arrived" — without a hook it could land while the bucket's all-reduce is in flight.  An arrival defers the bucket
            to `finish()` and sends the wrapper back to learning the order; an arrival behind the bucket's launch raises;
          * `persistent_grad_ids` (their `.grad` is never None): the bucket is ready only once their hook has fired."""
        by_id = {id(p): p for p in self._order}
        self._expected = [[] for _ in self._buckets]
        last: dict[int, int] = {}
        for pid in self._arrivals:
            b = self._bucket_of[pid]
            self._expected[b].append(by_id[pid])
            last[b] = pid
        for h in self._hook_handles:
            h.remove()
        recorded = set(self._arrivals)
        self._hook_handles = []
        for q in self._order:  # arrival hooks first: on a shared parameter they run before the bucket hook
            if q.requires_grad and (id(q) not in recorded or id(q) in self.persistent_grad_ids):
                self._hook_handles.append(q.register_post_accumulate_grad_hook(self._make_arrival_hook(q, id(q) in recorded)))
        self._hook_handles += [by_id[pid].register_post_accumulate_grad_hook(self._make_bucket_hook(b)) for b, pid in last.items()]
        self._req_sig = tuple(p.requires_grad for p in self._order)
        self._persist_sig = frozenset(self.persistent_grad_ids)
        self._mode = "bucket"

    def begin_backward(self, sync: bool = True):
        """Arm the wrapper for ONE backward pass.  sync=False: a non-final micro-step of a gradient-accumulation window
        (hooks only move / accumulate gradients into the arena); sync=True: the window's last backward — each bucket's
        all-reduce starts when the last of its gradients has been written.  `attach_grads()` arms with sync=True, so
        callers without accumulation never need this."""
        if self._handles:
            raise RuntimeError("begin_backward() while bucket all-reduces of the previous backward are in flight: call finish()")
        if not sync or not getattr(self, "_fresh", False):
            # an accumulation window: gradients of earlier micro-steps sit in the slots, so "the gradient is there" no longer
            # proves that THIS backward's contribution arrived — count arrivals per parameter, for good
            if self._mode != "param":
                self._mode = "param"
                if self._hooks_installed:
                    self._install_param_hooks()
        if self._hooks_installed and (self._req_sig != tuple(p.requires_grad for p in self._order) or (
                self._mode == "bucket" and self._persist_sig != frozenset(self.persistent_grad_ids))):
            # the trainable set changed (update_freeze_ratio), or the set of never-zeroed gradients did: hooks follow it, the
            # arrival order is learned again
            if self._mode == "bucket":
                self._mode = "learn"
            self._install_param_hooks()
        self._fresh = False
        self._arrivals = []
        self._deferred = set()
        self._arrived = set()
        self._relearn = False
        self._sync = bool(sync)
        self._pending = [0] * len(self._buckets)
        for p in self._order:
            if p.requires_grad:
                self._pending[self._bucket_of[id(p)]] += 1

    def _make_arrival_hook(self, p, expected: bool):
        def hook(_param):
            self.hook_calls += 1
            self._arrived.add(id(p))
            if expected:
                return
            # a gradient the recorded backward did not have (a branch unused then, a parameter unfrozen behind our back)
            self._relearn = True
            b = self._bucket_of[id(p)]
            if self.world > 1 and self._sync:
                if self._pending[b] == 0:
                    raise RuntimeError("HipDDP: a gradient outside the recorded backward order arrived after its bucket had been "
                                       "reduced; call begin_backward(sync=False) once (per-parameter hooks) or construct the "
                                       "wrapper again")
                self._deferred.add(b)

        return hook

    def _make_bucket_hook(self, b: int):
        def hook(_param):
            self.hook_calls += 1
            ready = b not in self._deferred
            for q in self._expected[b]:
                g = q.grad
                if g is None or (id(q) in self.persistent_grad_ids and id(q) not in self._arrived):
                    ready = False      # the arrival order changed: leave the bucket to finish()
                    continue
                v = self._views[id(q)]
                if g.data_ptr() != v.data_ptr():
                    v.copy_(g)
                    q.grad = v
            if self.world <= 1 or not self._sync:
                return
            if not ready:
                self._deferred.add(b)
                return
            self._pending[b] = 0
            self._launch(b)

        return hook

    def _make_hook(self, p):
        def hook(param):
            self.hook_calls += 1
            if self._mode == "learn":
                self._arrivals.append(id(p))
            v = self._views[id(p)]
            g = param.grad
            if g is not None and g.data_ptr() != v.data_ptr():  # produced elsewhere: move it into the arena
                v.copy_(g)
                param.grad = v
            if self.world <= 1 or not self._sync:
                return
            b = self._bucket_of[id(p)]
            if b >= len(self._pending):
                return
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)

        return hook

    def _launch(self, b: int):
        lo, hi = self._buckets[b]
        buf = self._arena[lo:hi]
        avg = dist.get_backend() == "nccl"   # gloo has no AVG: SUM, then divide in finish()
        if self._wire is not None:
            wire = buf.to(self._wire)
            h = dist.all_reduce(wire, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
            self._handles.append((h, buf, wire, 1.0 if avg else 1.0 / self.world))
        else:
            h = dist.all_reduce(buf, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
            self._handles.append((h, buf, None, 1.0 if avg else 1.0 / self.world))

    def finish(self):
        """Join the bucket all-reduces (also launches buckets whose hooks never fired, e.g. unused parameters)."""
        if self._arena is None:
            return
        for p in self._order:  # parameters without a gradient this step keep the old contract: a zero view
            if not p.requires_grad:
                continue
            v = self._views[id(p)]
            if p.grad is None:
                p.grad = v
            elif self._mode == "bucket" and p.grad.data_ptr() != v.data_ptr():
                # bucket mode: a gradient that arrived after its bucket's hook (or outside the recorded order) is still
                # outside the arena; its bucket was deferred below, so nothing has been reduced without it
                b = self._bucket_of[id(p)]
                if self.world > 1 and self._sync and self._pending[b] == 0:
                    raise RuntimeError("HipDDP: a gradient arrived after its bucket had been reduced (the backward order changed "
                                       "between steps); construct the wrapper again or call begin_backward(sync=False) once to "
                                       "return to per-parameter hooks")
                self._deferred.add(b)
                v.copy_(p.grad)
                p.grad = v
        learned = self._mode == "learn" and self._sync and bool(self._arrivals)
        if self._mode == "bucket" and self._relearn:  # the backward is not the recorded one any more: per-parameter hooks again
            self._mode = "learn"
            self._install_param_hooks()
        if self.world <= 1:
            if learned:
                self._install_bucket_hooks()
            return
        if not self._sync:
            raise RuntimeError("finish() after an accumulate-only backward: the window's last backward must be armed with "
                               "begin_backward(sync=True)")
        for b, n in enumerate(self._pending):
            if n > 0:
                self._pending[b] = 0
                self._launch(b)
        for h, buf, wire, scale in self._handles:
            h.wait()
            if wire is not None:
                buf.copy_(wire)
            if scale != 1.0:
                buf.mul_(scale)
        self._handles = []
        if learned:
            self._install_bucket_hooks()

    # ------------------------------------------------------------------------------------------- nn.Module
    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def state_dict(self, *args, **kwargs):
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        return self.module.load_state_dict(*args, **kwargs)


class DistributedUtils:
    """utils/ddp.py:11-136 with RCCL ('nccl' on ROCm) on GPUs and gloo on CPU."""

    MP = MP
    DS = DS
    DDP = HipDDP
    dist = dist

    @staticmethod
    def is_initialized():
        return dist.is_initialized()

    @staticmethod
    def ddp_setup(gpu_id: int, world_size: int):
        local_rank = int(os.environ.get("LOCAL_RANK", gpu_id if isinstance(gpu_id, int) else 0))
        rank = int(os.environ.get("RANK", gpu_id if isinstance(gpu_id, int) else 0))
        backend = "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            backend = "nccl"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, init_method="env://", world_size=world_size, rank=rank, **kw)

    @staticmethod
    def ddp_cleanup():
        if dist.is_initialized():
            dist.destroy_process_group()

    @staticmethod
    def sync_process_group(world_size: int, device_ids: int):
        if world_size > 1 and dist.is_initialized():
            dist.barrier()

    @staticmethod
    def gather_loss(loss: list, device: int) -> float:
        t = torch.tensor(loss, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.mean().item() / dist.get_world_size()

    @staticmethod
    def gather_object(obj: Any, world_size: int) -> list:
        if world_size <= 1:
            return [obj]
        out = [None for _ in range(world_size)]
        dist.all_gather_object(out, obj)
        return out

    @staticmethod
    def gather_tensor(tensor: torch.Tensor, world_size: int) -> torch.Tensor:
        if world_size <= 1:
            return tensor
        t = tensor.contiguous()
        out = torch.empty((world_size * t.shape[0], *t.shape[1:]), device=t.device, dtype=t.dtype)
        dist.all_gather_into_tensor(out, t)
        return out
