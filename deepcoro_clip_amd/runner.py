"""VideoContrastiveLearningRunner on the HIP path — same registry names, constructor, `_train_step` / `_val_step` /
`_run_epoch` / `train` / `inference` contract as runners/video_constrative_learning_runner.py:45-1826.

What is kept bit-for-bit in meaning: gradient-accumulation / clip / step ordering (:1223-1224,1387-1424), the
`nan_to_num(…, 1e4, -1e4)` guards (:1236-1249), both loss branches (registry loss :1315-1321, inline multi-positive
SigLIP :1256-1296), the manual `log_temp.grad` all-reduce (:1410-1415), the metric keys (:1516-1530) and the
`(batch_metrics, {"video_embeddings", "text_embeddings"})` return value.

What is re-designed for MI355X: one packed device->host transfer per step instead of >= 8 `.item()` syncs, no
duplicate similarity GEMM for the alignment metrics (they fall out of the fused loss kernels), gradients reduced
in large flat-arena buckets overlapped with backward (ddp.HipDDP), clip + AdamW as multi-tensor HIP kernels.
AMP: compute dtype (bf16 / fp32) is a property of the encoders; a GradScaler is accepted for API compatibility and
ignored (bf16 needs no loss scaling).
"""
from __future__ import annotations

import itertools
import math
import os
import time
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist

from . import functional as Fn
from . import kernels as K
from .ddp import DistributedUtils, HipDDP
from .losses import KIND_BCE, MODE_GATED, WeightedSigLIPLoss, pair_loss
from .optim import GradClipper, HipAdamW
from .registry import RunnerRegistry
from .retrieval_metrics import (compute_map, compute_median_rank, compute_mrr, compute_ndcg_at_k, compute_recall_at_k,
                                compute_similarity_matrix)


def _cfg(config, name, default):
    return getattr(config, name, default) if config is not None else default


class StepMetrics(dict):
    """The per-step metric dict (`_train_step`'s first return value, same keys and float values as the reference's),
    filled on FIRST ACCESS: the scalars travel in one asynchronous device->pinned-host copy and the host only waits for
    it when a value is actually read.  A loop that looks at the metrics every step (`_run_epoch`, the reference's own
    loop) behaves exactly as before; a loop that does not (bench.py, log-every-N training) no longer drains the GPU
    queue once per step, so the next step's launches overlap the current step's backward."""

    def __init__(self, host: torch.Tensor, event, build):
        super().__init__()
        self._pending = (host, event, build)

    def _resolve(self):
        pending, self._pending = self._pending, None
        if pending is not None:
            host, event, build = pending
            event.synchronize()
            dict.update(self, build(host.tolist()))

    def ready(self) -> bool:
        """True once the values are on the host (never blocks)."""
        return self._pending is None or self._pending[1].query()

    def __getitem__(self, k): self._resolve(); return dict.__getitem__(self, k)
    def __setitem__(self, k, v): self._resolve(); dict.__setitem__(self, k, v)
    def __delitem__(self, k): self._resolve(); dict.__delitem__(self, k)
    def __contains__(self, k): self._resolve(); return dict.__contains__(self, k)
    def __iter__(self): self._resolve(); return dict.__iter__(self)
    def __len__(self): self._resolve(); return dict.__len__(self)
    def __repr__(self): self._resolve(); return dict.__repr__(self)
    def __eq__(self, o): self._resolve(); return dict.__eq__(self, o)
    def __ne__(self, o): self._resolve(); return dict.__ne__(self, o)
    def __or__(self, o): self._resolve(); return dict(self) | o
    def __reduce__(self): self._resolve(); return (dict, (dict(self),))
    def keys(self): self._resolve(); return dict.keys(self)
    def values(self): self._resolve(); return dict.values(self)
    def items(self): self._resolve(); return dict.items(self)
    def get(self, k, d=None): self._resolve(); return dict.get(self, k, d)
    def pop(self, *a): self._resolve(); return dict.pop(self, *a)
    def popitem(self): self._resolve(); return dict.popitem(self)
    def setdefault(self, k, d=None): self._resolve(); return dict.setdefault(self, k, d)
    def update(self, *a, **k): self._resolve(); dict.update(self, *a, **k)
    def copy(self): self._resolve(); return dict(self)


@RunnerRegistry.register("DeepCORO_clip")
@RunnerRegistry.register("DeepCORO_clip_test")
class VideoContrastiveLearningRunner:
    def __init__(self, config=None, wandb_wrapper=None, train_loader=None, val_loader=None, video_encoder=None,
                 text_encoder=None, optimizer=None, scaler=None, log_temp: torch.Tensor = None, lr_scheduler=None,
                 loss_fn=None, output_dir: str = None):
        if not isinstance(config.recall_k, list):
            raise ValueError(f"config.recall_k must be a list of ints, got {type(config.recall_k)}")
        if not isinstance(config.ndcg_k, list):
            raise ValueError(f"config.ndcg_k must be a list of ints, got {type(config.ndcg_k)}")
        self.config = config
        self.wandb_wrapper = wandb_wrapper
        self.device = config.device
        self.world_size = config.world_size
        self.train_loader, self.val_loader = train_loader, val_loader
        self.video_encoder, self.text_encoder = video_encoder, text_encoder
        self.optimizer = optimizer
        if isinstance(optimizer, (torch.optim.AdamW, torch.optim.Adam)) and not isinstance(optimizer, HipAdamW):
            self.optimizer = HipAdamW.adopt(optimizer)  # same param_groups / state, HIP update kernel
        self.scaler = scaler
        self.lr_scheduler = lr_scheduler
        self.loss_fn = loss_fn
        self.output_dir = output_dir
        self.best_val_loss = float("inf")
        self.best_epoch = -1
        self.highest_alignment_score = float("-inf")
        self.highest_alignment_epoch = -1
        self.log_temp = log_temp
        self.step = 0
        self.scheduler_per_iteration = self._scheduler_is_per_iteration()
        self.initial_log_temp = log_temp.detach().clone() if log_temp is not None else torch.log(torch.tensor(config.temperature))
        self.use_weighted_siglip = bool(_cfg(config, "siglip_use_weighted_loss", False))
        self.weighted_siglip_loss = WeightedSigLIPLoss() if self.use_weighted_siglip else None
        self.siglip_abnormal_margin = float(_cfg(config, "siglip_abnormal_margin", 0.0))
        self._clipper: GradClipper | None = None
        self._orphan_clipper: GradClipper | None = None
        self._text_stream: torch.cuda.Stream | None = None
        self._accum_grad: torch.Tensor | None = None
        self._use_step_graph = _cfg(config, "use_step_graph", False)  # False | True | "auto"
        if os.environ.get("DCC_STEP_GRAPH") in ("1", "auto"):
            self._use_step_graph = True if os.environ["DCC_STEP_GRAPH"] == "1" else "auto"
        self._graphs: dict = {}
        self._dropout_step: torch.Tensor | None = None
        self._autograd_mt = os.environ.get("DCC_AUTOGRAD_MT", "0") == "1"  # backward on the engine's device thread
        # weight-gradient GEMMs on the parameter-gradient side stream (one rank, no accumulation: functional.wgrad).  Round 4
        # re-measured it (and fixed the in-place-accumulation race it had): -1.4 ms per step on three boxes, +1.3 ms on two others,
        # where the host becomes the limit (every side launch costs two event operations; enqueue time inside the timed region
        # 36.7 -> 40.5 ms against 40.7 ms of GPU time).  Off by default; DCC_WGRAD_STREAM=1 turns it on.
        self._wgrad_side = os.environ.get("DCC_WGRAD_STREAM", "0") == "1"
        self._graph_stream: torch.cuda.Stream | None = None
        self._cache_check = os.environ.get("DCC_CACHE_CHECK", "0") == "1"
        # host -> device copies of batch i+1 on their own stream while step i runs (config.prefetch_batches, default on): the
        # headline shard is 308 MB of fp32 frames (5-6 ms over PCIe; 77 MB as uint8) that would otherwise sit in front of every step
        self._h2d_stream: torch.cuda.Stream | None = None
        self._prefetch = bool(_cfg(config, "prefetch_batches", True)) and os.environ.get("DCC_PREFETCH", "1") != "0"
        self._arena = Fn.ZeroArena()  # this runner's per-step gradient arena (installed as functional.ARENA by _zero_grad)

    # ------------------------------------------------------------------------------------------ helpers
    def _scheduler_is_per_iteration(self) -> bool:
        name = str(_cfg(self.config, "scheduler_name", "") or "").lower()
        return "warmup" in name

    def _ddp_wrappers(self):
        return [m for m in (self.video_encoder, self.text_encoder) if isinstance(m, HipDDP)]

    def _zero_grad(self):
        wrappers = self._ddp_wrappers()
        if self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=True)
        for w in wrappers:  # flat gradient arenas: zero + re-attach views (replaces set_to_none)
            w.attach_grads()
        if int(_cfg(self.config, "gradient_accumulation_steps", 1)) == 1 and self.optimizer is not None:
            # accumulation-produced gradients of this step come out of one pre-zeroed arena (one memset per step)
            dev = self.log_temp.device if self.log_temp is not None else torch.device("cuda")
            hint = sum(p.numel() for g in self.optimizer.param_groups for p in g["params"]) + (1 << 20)
            self._rehome_orphan_grads()
            Fn.ARENA = self._arena  # two runners in one process must not hand out each other's gradient memory
            Fn.ARENA.begin(dev, hint)

    def _rehome_orphan_grads(self):
        """Gradients nobody zeroes (parameters outside every optimizer group: the reference's video `proj`, a loss module's
        own bias — SURVEY §9.11) keep accumulating across steps.  When such a `.grad` is a slice of the per-step zero arena
        it must leave the arena before `begin()` re-zeroes and re-issues that memory: clone it out once; from then on
        autograd accumulates into the private clone."""
        buf = self._arena.buf
        if buf is None:
            return
        if getattr(self, "_orphans", None) is None:
            in_opt = {id(p) for g in self.optimizer.param_groups for p in g["params"]}
            mods = [self.video_encoder, self.text_encoder]
            lt = getattr(self.loss_fn, "loss_type", self.loss_fn)
            if isinstance(lt, torch.nn.Module):
                mods.append(lt)
            self._orphans = [p for m in mods if m is not None for p in m.parameters() if id(p) not in in_opt]
        lo, hi = buf.data_ptr(), buf.data_ptr() + buf.numel() * 4
        for p in self._orphans:
            g = p.grad
            if g is not None and lo <= g.data_ptr() < hi:
                p.grad = g.clone()

    def _all_encoder_params(self):
        return list(itertools.chain(self.video_encoder.parameters(), self.text_encoder.parameters()))

    def _preprocess_inputs(self, batch: dict):
        videos = batch["videos"].to(self.device, non_blocking=True)
        if videos.dtype == torch.uint8:
            # optional fast input path: raw uint8 frames, normalised on the GPU exactly like
            # utils/video.py:278-288 does on the host ((x - mean) / std per channel); 4x less H2D traffic
            mean, std = _cfg(self.config, "video_mean", None), _cfg(self.config, "video_std", None)
            if mean is None or std is None:
                raise ValueError("uint8 videos need config.video_mean / config.video_std (per-channel, pixel units)")
            enc = self.video_encoder.module if hasattr(self.video_encoder, "module") else self.video_encoder
            if hasattr(enc, "set_input_normalization"):
                # bytes all the way into the patch-embedding gather: normalised on the fly, no fp32 copy of the clip
                if getattr(enc, "input_norm", None) is None:
                    enc.set_input_normalization(mean, std)
                videos = videos.contiguous()
            else:
                C = videos.shape[-1]
                as_vec = lambda v: torch.tensor([float(v)] * C if isinstance(v, (int, float)) else [float(x) for x in v][:C],  # noqa: E731
                                                dtype=torch.float32, device=videos.device)
                videos = K.u8_normalize(videos.contiguous(), as_vec(mean), as_vec(std))
        elif videos.dtype != torch.float32:
            videos = K.cast(videos.contiguous(), torch.float32)
        if videos.dtype != torch.uint8:
            videos = K.nan_to_num_(videos.contiguous(), 0.0, 0.0, 0.0)
        step_inputs: Dict[str, Any] = {"videos": videos}
        enc = batch.get("encoded_texts")
        if enc is not None:
            step_inputs["input_ids"] = enc["input_ids"].to(self.device, non_blocking=True)
            step_inputs["attention_mask"] = enc["attention_mask"].to(self.device, non_blocking=True)
        for key in ("positive_mask", "positive_weights"):
            if batch.get(key) is not None:
                step_inputs[key] = batch[key].to(self.device, non_blocking=True)
        for key in ("text_ids", "text_metadata"):
            if key in batch:
                step_inputs[key] = batch[key]
        return step_inputs, batch.get("paths", [])

    def _prefetched_batches(self, loader):
        """Yield `_preprocess_inputs(batch)` for every batch of `loader` with the NEXT batch's host -> device copies already
        issued on a side stream (one batch of lookahead).  Pinned host tensors (DataLoader(pin_memory=True), as the reference
        builds its loaders) make those copies asynchronous, so they overlap the running step; pageable tensors still work
        (the copy then blocks the host inside the side-stream context, as it would have on the main stream).  The consumer's
        stream waits for the copy event, and every device tensor is recorded on it (the caching allocator must not hand the
        memory back while the step still reads it)."""
        use_side = self._prefetch and torch.cuda.is_available() and str(self.device).startswith("cuda")

        def stage(batch):
            if not use_side:
                return self._preprocess_inputs(batch) + (None,)
            if self._h2d_stream is None:
                self._h2d_stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._h2d_stream):
                step_inputs, paths = self._preprocess_inputs(batch)
                ev = torch.cuda.Event()
                ev.record()
            return step_inputs, paths, ev

        def valid(batches):
            for b_ in batches:
                if b_ is not None and b_.get("videos") is not None:
                    yield b_

        it = valid(iter(loader))
        batch = next(it, None)
        staged = stage(batch) if batch is not None else None
        while staged is not None:
            nb = next(it, None)
            nxt = stage(nb) if nb is not None else None   # issued BEFORE the current step is enqueued: the copy runs under it
            step_inputs, paths, ev = staged
            if ev is not None:
                main = torch.cuda.current_stream()
                main.wait_event(ev)
                for v in step_inputs.values():
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(main)
            yield batch, step_inputs, paths
            batch, staged = nb, nxt

    def _encode(self, videos, input_ids, attention_mask):
        # The towers are independent until the loss: the text tower (B x S = a few hundred tokens, ~200 small
        # latency-bound kernels) runs on a side HIP stream underneath the video tower's large kernels, forward and —
        # because autograd replays each node on its forward stream — backward too.
        text_emb = None
        if input_ids is not None and attention_mask is not None:
            if self._text_stream is None:
                self._text_stream = torch.cuda.Stream(device=videos.device)
            main = torch.cuda.current_stream()
            self._text_stream.wait_stream(main)
            # video first: the host runs far ahead of the GPU, so the text kernels still start at once on their own
            # stream, and autograd (reverse creation order) then enqueues the text BACKWARD first, which lets it
            # overlap the video backward as well
            video_emb = self.video_encoder(videos)
            with torch.cuda.stream(self._text_stream):
                text_emb = self.text_encoder(input_ids, attention_mask)
            main.wait_stream(self._text_stream)
            if not torch.cuda.is_current_stream_capturing():
                text_emb.record_stream(main)
        else:
            video_emb = self.video_encoder(videos)
        # torch.nan_to_num(emb, nan=0, posinf=1e4, neginf=-1e4): applied in place on the embedding storage (finite
        # after this, so the reference's isfinite().all() zeroing branch is unreachable and needs no host sync)
        K.nan_to_num_(video_emb.detach(), 0.0, 1e4, -1e4)
        if text_emb is not None:
            K.nan_to_num_(text_emb.detach(), 0.0, 1e4, -1e4)
        return video_emb, text_emb

    def _loss(self, video_emb, text_emb, positive_mask, positive_weights, text_metadata, diag: dict):
        """Returns the loss tensor; fills `diag` with the device-side diagnostics of the fused loss node."""
        if positive_mask is not None and text_emb is not None:
            targets = positive_mask.float().contiguous()
            margin = None
            if self.siglip_abnormal_margin > 0 and text_metadata:
                flags = [1.0 if (m and m.get("is_abnormal", False)) else 0.0 for m in text_metadata]
                if any(flags):
                    margin = torch.tensor(flags, device=targets.device, dtype=torch.float32) * self.siglip_abnormal_margin
            diag["multi_positive"] = True
            diag["mp_targets"] = targets
            diag["mp_weights"] = positive_weights.float().contiguous() if positive_weights is not None else None
            if self.use_weighted_siglip and self.weighted_siglip_loss is not None:
                pw = targets
                if positive_weights is not None:
                    # reference: targets * positive_weights when any weight is non-zero, else targets (:1276-1278);
                    # a product with an all-zero weight matrix is the degenerate case it guards against
                    pw = K.mp_pos_weights(targets, positive_weights.float().contiguous())
                return pair_loss(video_emb, text_emb, self.log_temp, mode=MODE_GATED, kind=1, pos=pw,
                                 eps=self.weighted_siglip_loss.eps, col_margin=margin, diag_out=diag)
            neg_w = float(_cfg(self.config, "siglip_negative_weight", 1.0))
            denom = K.colsum(targets.view(-1, 1))  # sum(targets) stays on the device (reference: .item() sync)
            return pair_loss(video_emb, text_emb, self.log_temp, mode=MODE_GATED, kind=KIND_BCE, targets=targets,
                             pos_weights=positive_weights.float().contiguous() if positive_weights is not None else None,
                             positive_weight=1.0 if positive_weights is not None else neg_w, negative_weight=neg_w, thr=0.0,
                             denom_dev=denom, col_margin=margin, diag_out=diag)
        lt = self.loss_fn.loss_type if hasattr(self.loss_fn, "loss_type") else self.loss_fn
        loss = self.loss_fn.run(video_features=video_emb, text_features=text_emb, log_temp=self.log_temp)
        last = getattr(lt, "last_diag", None)
        if last:
            diag.update(last)
        return loss

    def _step_metrics(self, loss, video_emb, text_emb, diag: dict) -> dict:
        """All per-step scalars in ONE device->host copy."""
        B = video_emb.shape[0]
        use_siglip = "siglip" in str(_cfg(self.config, "loss_name", "")).lower()
        scal = diag.get("scal")
        have_align = scal is not None and diag.get("lse_rows") is not None and not diag.get("multi_positive") \
            and diag.get("logits") is not None and diag["logits"].shape[0] == diag["logits"].shape[1] \
            and not diag.get("bias_logits", False)
        if (scal is None or not have_align) and not diag.get("multi_positive"):
            # foreign loss module or BCE/multi-positive logits: derive the diagonal diagnostics with small kernels
            if text_emb is not None and text_emb.shape[0] == B:
                scal2 = torch.zeros(8, device=video_emb.device, dtype=torch.float32)
                vh, _ = K.l2norm_fwd(video_emb.detach().float().contiguous(), norm_sum=scal2[5:6])
                th, _ = K.l2norm_fwd(text_emb.detach().float().contiguous(), norm_sum=scal2[6:7])
                sim = K.gemm(vh, th)
                z = K.logits_fwd(sim, 1 if use_siglip else 0, self.log_temp.detach().reshape(-1)[:1].float(), False)
                lr, lc = K.lse(z)
                K.ce_diag(z, lr, lc, 0.0, scal2, want_grad=False, sim=sim)
                scal = scal2
                have_align = True
        pack = torch.empty(18, device=video_emb.device, dtype=torch.float32)
        pack[0:1] = loss.detach().reshape(1)
        pack[1:2] = self.log_temp.detach().reshape(-1)[:1]
        if scal is not None:
            pack[2:9] = scal[:7] if scal.numel() >= 7 else 0
        mp = bool(diag.get("multi_positive")) and diag.get("logits") is not None and diag.get("sim") is not None
        if mp:  # weighted log-prob of the positives + mean cosine over the positives (reference :1298-1313)
            pack[10:18] = K.mp_align(diag["logits"], diag["sim"], diag["mp_targets"], diag.get("mp_weights"))
        # never inside a capture: a replayed step runs its optimizer launch and this metric copy eagerly (_graph_train_step)
        host = torch.empty(18, dtype=torch.float32, pin_memory=True)
        host.copy_(pack, non_blocking=True)  # the one device->host transfer of the step; waited for on first read
        event = torch.cuda.Event()
        event.record()
        n_text = text_emb.shape[0] if text_emb is not None else 0
        lrs = {f"lr/{pg.get('name', str(i))}": pg["lr"]
               for i, pg in enumerate(self.optimizer.param_groups if self.optimizer is not None else [])}
        has_scal, has_text = scal is not None, text_emb is not None

        def build(vals):
            m = {"loss": vals[0], "temperature": math.exp(vals[1]), "contrastive/clip_loss": vals[0]}
            if has_scal and has_text:
                m["video_norm"] = vals[2 + 5] / B
                m["text_norm"] = vals[2 + 6] / max(1, n_text)
            m.update(lrs)
            if mp:
                a = vals[10:18]
                use_w = a[6] > 0.0 and a[1] > 0.0   # any non-zero weight (else the reference falls back to the plain mask)
                num, cnt = (a[0], a[1]) if use_w else (a[2], a[3])
                cos = a[4] / a[5] if a[5] > 0 else None
                if cnt > 0:
                    m["alignment_logprob"] = num / cnt
                    m["alignment_score"] = math.exp(num / cnt)
                else:
                    m["alignment_score"] = cos if cos is not None else 0.0
                if cos is not None:
                    m["alignment_cosine"] = cos
            elif have_align:
                m["alignment_logprob"] = vals[2 + 3]
                m["alignment_score"] = math.exp(vals[2 + 3])
                m["alignment_cosine"] = vals[2 + 4]
            else:
                m["alignment_score"] = 0.0
            return m

        return StepMetrics(host, event, build)

    # --------------------------------------------------------------------------------------- train / val step
    def _train_step(self, videos: torch.Tensor, input_ids: Optional[torch.Tensor] = None,
                    attention_mask: Optional[torch.Tensor] = None, positive_mask: Optional[torch.Tensor] = None,
                    positive_weights: Optional[torch.Tensor] = None, text_ids: Optional[List[str]] = None,
                    text_metadata: Optional[List[Dict[str, Any]]] = None,
                    debug_sample_info: Optional[List[Dict[str, Any]]] = None) -> tuple[dict, dict]:
        accum = int(_cfg(self.config, "gradient_accumulation_steps", 1))
        micro = int(_cfg(self.config, "embedding_cache_micro_batch", 0) or 0)
        cached = micro > 0 and videos.shape[0] > micro and input_ids is not None
        # the two-pass embedding-cache step exists to bound activation memory: it is never replaced by a one-shot captured step
        # (r03: `bench.py --config siglip4096` with the step graph on ran all 512 clips at once, 272 GB instead of 24.6)
        if not cached and self._graph_eligible(accum, positive_mask, text_metadata):
            return self._graph_train_step(videos, input_ids, attention_mask)
        if cached:
            total_loss, video_emb, text_emb, diag = self._forward_backward_cached(
                videos, input_ids, attention_mask, positive_mask, positive_weights, text_metadata, accum, micro)
        else:
            total_loss, video_emb, text_emb, diag = self._forward_backward(videos, input_ids, attention_mask, positive_mask,
                                                                           positive_weights, text_metadata, accum)
        boundary = (self.step + 1) % accum == 0
        fused_clip = self._reduce_and_clip() if boundary else None
        return self._finish_step(boundary, fused_clip, total_loss, video_emb, text_emb, diag)

    def _forward_backward(self, videos, input_ids, attention_mask, positive_mask, positive_weights, text_metadata, accum):
        if self.step % accum == 0:
            self._zero_grad()
        # dropout masks: device-side step counter (advanced here, inside a captured step too) + per-call host seeds
        if self._dropout_step is None or self._dropout_step.device != videos.device:
            self._dropout_step = torch.zeros(1, device=videos.device, dtype=torch.int64)
        self._dropout_step.add_(1)
        Fn.begin_step(self._dropout_step)
        try:
            video_emb, text_emb = self._encode(videos, input_ids, attention_mask)
        finally:
            Fn.begin_step(None)
        diag: dict = {}
        clip_loss = self._loss(video_emb, text_emb, positive_mask, positive_weights, text_metadata, diag)
        total_loss = clip_loss
        if self._accum_grad is None or float(self._accum_grad_val) != 1.0 / accum:
            self._accum_grad_val = 1.0 / accum
            self._accum_grad = torch.full((), 1.0 / accum, device=clip_loss.device, dtype=clip_loss.dtype)
        # weight-gradient GEMMs off the critical chain (functional.wgrad): safe when every gradient buffer is adopted as is
        Fn.WGRAD_SIDE = self._wgrad_side and accum == 1 and self.world_size == 1 and not self._ddp_wrappers()
        # parameter-only gradient kernels (the pools' conv weight gradients) may leave the data path under the same conditions
        Fn.PARAM_SIDE_OK = accum == 1 and self.world_size == 1 and not self._ddp_wrappers()
        if accum > 1:  # torch DDP's no_sync(): only the window's last backward reduces the gradient buckets
            for w in self._ddp_wrappers():
                w.begin_backward(sync=(self.step + 1) % accum == 0)
        try:
            # backward on the calling thread: handing ~670 nodes to the engine's device thread one by one costs the host
            # ~13 % of a step's launch time (31.0 -> 26.9 ms on launch-bound 112^2 inputs), and the host is the limit on
            # boxes with slow cores
            with torch.autograd.set_multithreading_enabled(self._autograd_mt):
                total_loss.backward(self._accum_grad)  # == (loss / accum).backward() without an extra kernel
        finally:
            Fn.WGRAD_SIDE = False
            Fn.PARAM_SIDE_OK = False
        return total_loss, video_emb, text_emb, diag

    def _forward_backward_cached(self, videos, input_ids, attention_mask, positive_mask, positive_weights, text_metadata,
                                 accum, micro):
        """Large-batch contrastive step with an embedding cache (BASELINE configs[3]: SigLIP at a global batch of 4096 =
        512 studies per rank, far beyond what one forward's activations allow).  Two passes over micro-batches of `micro`
        studies:
          1. forward WITHOUT autograd, embeddings only -> [B, 512] video / text caches (a few MB);
          2. the loss on the full cached batch (all-gathered to [Bg, 512] by the *_ddp losses: the reference's global
             similarity matrix, utils/loss/losses.py:213-276) and its gradient w.r.t. the cached embeddings;
          3. forward again WITH autograd per micro-batch, back-propagating that micro-batch's rows of dL/dE.
        Parameter gradients are exactly those of the one-shot step (the chain rule through E); every stochastic layer
        redraws the masks of pass 1 in pass 3 (dropout: same call-site seeds + device step counter; stochastic depth:
        the device generator is rewound per micro-batch).  Gradients reach DDP once, from the last micro-batch's hooks."""
        if self.step % accum == 0:
            self._zero_grad()
        B = videos.shape[0]
        if self._dropout_step is None or self._dropout_step.device != videos.device:
            self._dropout_step = torch.zeros(1, device=videos.device, dtype=torch.int64)
        self._dropout_step.add_(1)
        chunks = [(s0, min(B, s0 + micro)) for s0 in range(0, B, micro)]
        rng_states = []

        def encode(lo, hi, i):
            # identical randomness in both passes: rewind the device generator, restart the dropout call-site counter
            if len(rng_states) <= i:
                rng_states.append(torch.cuda.get_rng_state(videos.device))
            else:
                torch.cuda.set_rng_state(rng_states[i], videos.device)
            Fn.begin_step(self._dropout_step)
            Fn._STEP["calls"] = 100003 * (i + 1)  # disjoint call-site seeds per micro-batch
            try:
                return self._encode(videos[lo:hi], input_ids[lo:hi], attention_mask[lo:hi])
            finally:
                Fn.begin_step(None)

        with torch.no_grad():
            cache = [encode(lo, hi, i) for i, (lo, hi) in enumerate(chunks)]
        video_emb = torch.cat([c[0].float() for c in cache], 0).requires_grad_(True)
        text_emb = torch.cat([c[1].float() for c in cache], 0).requires_grad_(True)
        del cache
        diag: dict = {}
        clip_loss = self._loss(video_emb, text_emb, positive_mask, positive_weights, text_metadata, diag)
        if self._accum_grad is None or float(self._accum_grad_val) != 1.0 / accum:
            self._accum_grad_val = 1.0 / accum
            self._accum_grad = torch.full((), 1.0 / accum, device=clip_loss.device, dtype=clip_loss.dtype)
        boundary = (self.step + 1) % accum == 0
        wrappers = self._ddp_wrappers()
        with torch.autograd.set_multithreading_enabled(self._autograd_mt):
            for w in wrappers:  # the loss backward touches no encoder parameter; keep the buckets closed
                w.begin_backward(sync=False)
            clip_loss.backward(self._accum_grad)   # -> video_emb.grad, text_emb.grad, log_temp.grad
            dv, dt = video_emb.grad, text_emb.grad
            for i, (lo, hi) in enumerate(chunks):
                for w in wrappers:
                    w.begin_backward(sync=boundary and i == len(chunks) - 1)
                v_i, t_i = encode(lo, hi, i)
                if self._cache_check:  # debug aid (tests): pass 3 must reproduce pass 1's embeddings, masks included
                    d = max(float((v_i.detach().float() - video_emb.detach()[lo:hi]).abs().max()),
                            float((t_i.detach().float() - text_emb.detach()[lo:hi]).abs().max()))
                    self._cache_maxdiff = max(getattr(self, "_cache_maxdiff", 0.0), d)
                torch.autograd.backward([v_i, t_i], [Fn._as_dtype(dv[lo:hi], v_i.dtype), Fn._as_dtype(dt[lo:hi], t_i.dtype)])
                Fn.ARENA.end()  # later micro-batches' gradient buffers are temporaries (accumulated, then freed)
        return clip_loss, video_emb.detach(), text_emb.detach(), diag

    def _reduce_and_clip(self):
        """Join the gradient all-reduces, compute the clip coefficient; returns it when the optimizer applies it on the fly."""
        max_norm = _cfg(self.config, "max_grad_norm", 5.0)
        for w in self._ddp_wrappers():
            w.finish()  # join the overlapped bucket all-reduces (RCCL)
        fused_clip = None
        if max_norm:
            if self._clipper is None:
                self._clipper = GradClipper(self._all_encoder_params())
                self._clipped_ids = {id(p) for p in self._clipper.params}
                # clipped parameters outside every optimizer group (the reference's video `proj`):
                # nobody zeroes their .grad, so it keeps accumulating and keeps being rescaled in place — that
                # history feeds the norm of every later step and must be reproduced
                in_opt = {id(p) for g in self.optimizer.param_groups for p in g["params"]}
                orphans = [p for p in self._clipper.params if id(p) not in in_opt]
                self._orphan_clipper = GradClipper(orphans) if orphans else None
                for w in self._ddp_wrappers():
                    w.persistent_grad_ids = {id(p) for p in orphans}
            fuse = isinstance(self.optimizer, HipAdamW)  # the update kernel applies the coefficient on the fly
            self._clipper.clip_(float(max_norm), scale=not fuse)
            fused_clip = self._clipper.coef if fuse else None
            if fuse and self._orphan_clipper is not None:
                self._orphan_clipper.scale_with(self._clipper.coef)
        if isinstance(self.log_temp, torch.nn.Parameter) and self.log_temp.grad is not None and dist.is_available() \
                and dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() == "nccl":
                dist.all_reduce(self.log_temp.grad, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(self.log_temp.grad, op=dist.ReduceOp.SUM)
                self.log_temp.grad.div_(dist.get_world_size())
        return fused_clip

    def _finish_step(self, boundary: bool, fused_clip, total_loss, video_emb, text_emb, diag, static_outputs: bool = False):
        if boundary:
            if fused_clip is not None:
                self.optimizer.step(grad_scale=fused_clip, scaled=self._clipped_ids)
            else:
                self.optimizer.step()
            if not isinstance(self.optimizer, HipAdamW):
                Fn.invalidate_weight_cache()
            if self.lr_scheduler and self.scheduler_per_iteration and self.step > 0:
                self.lr_scheduler.step()
        self.step += 1
        with torch.no_grad():
            batch_metrics = self._step_metrics(total_loss, video_emb, text_emb, diag)
            video_fp32 = video_emb.detach().float()
            text_fp32 = text_emb.detach().float() if text_emb is not None else torch.empty(0, device=video_fp32.device)
            if static_outputs:  # replayed graphs rewrite their output buffers: hand out copies
                if video_fp32.data_ptr() == video_emb.data_ptr():
                    video_fp32 = video_fp32.clone()
                if text_emb is not None and text_fp32.data_ptr() == text_emb.data_ptr():
                    text_fp32 = text_fp32.clone()
        return batch_metrics, {"video_embeddings": video_fp32, "text_embeddings": text_fp32}

    # ------------------------------------------------------------------------------------ HIP-graph step (opt-in)
    def _graph_eligible(self, accum: int, positive_mask, text_metadata) -> bool:
        """The step can be replayed from a HIP graph when nothing in it depends on per-step host state: one rank,
        no accumulation, the HIP AdamW (its table upload and launch stay outside the graph), the plain pair loss."""
        if not self._use_step_graph:
            return False
        return (self.world_size == 1 and not self._ddp_wrappers() and accum == 1 and isinstance(self.optimizer, HipAdamW)
                and positive_mask is None and not text_metadata and self.video_encoder.training and torch.is_grad_enabled())

    def _graph_train_step(self, videos, input_ids, attention_mask):
        """forward + loss + backward + clip replayed from a captured graph (one launch instead of ~1400); the optimizer
        launch, the scheduler and the metric copy follow eagerly.  The first calls with a new input signature run eagerly
        on the capture stream (warm-up: allocator, lazy state, autograd's accumulate nodes), then the step is captured;
        a step that draws host-seeded randomness (dropout) is never captured."""
        import gc

        key = (tuple(videos.shape), videos.dtype, None if input_ids is None else tuple(input_ids.shape))
        st = self._graphs.get(key)
        if st is None:
            st = self._graphs[key] = {"calls": 0, "graph": None, "eager_only": False}
        if self._graph_stream is None:
            self._graph_stream = torch.cuda.Stream(device=videos.device)
        side, cur = self._graph_stream, torch.cuda.current_stream()
        if st["graph"] is None:
            if st["calls"] == 3 and not st["eager_only"] and self._use_step_graph == "auto":
                # "auto": replay only when launching the step costs the host about as long as the GPU needs to run it
                # (slow host cores, small inputs); a clearly GPU-bound step stays eager (hipGraphLaunch itself is ~1 ms
                # slower per step there).  Measured on the warm-up calls: host time of the call vs event time on the GPU.
                torch.cuda.synchronize()
                host = st["host_ms"][-1]                       # the last warm-up call is the closest to steady state
                gpu = st["events"][-1][0].elapsed_time(st["events"][-1][1])
                st["eager_only"] = host < 0.95 * gpu           # GPU-bound: ~0.90 on a fast host; host-bound: ~1.0
                st["decision"] = (host, gpu)
            if st["eager_only"] or st["calls"] < 3:
                if st["eager_only"] and st["calls"] >= 3:  # settled on eager: the ordinary path on the caller's stream
                    out = self._forward_backward(videos, input_ids, attention_mask, None, None, None, 1)
                    return self._finish_step(True, self._reduce_and_clip(), *out)
                st["calls"] += 1
                drops = Fn._DROPOUT_CALLS
                side.wait_stream(cur)
                t0 = time.perf_counter()
                with torch.cuda.stream(side):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    out = self._forward_backward(videos, input_ids, attention_mask, None, None, None, 1)
                    clip = self._reduce_and_clip()
                    res = self._finish_step(True, clip, *out)
                    e1.record()
                st.setdefault("host_ms", []).append(1e3 * (time.perf_counter() - t0))
                st.setdefault("events", []).append((e0, e1))
                cur.wait_stream(side)
                for t in (videos, input_ids, attention_mask):
                    if t is not None:
                        t.record_stream(side)
                if Fn._DROPOUT_CALLS != drops:
                    st["eager_only"] = True   # a dropout call outside the step scope: its host seed would be frozen
                return res
            st["inputs"] = [None if t is None else t.clone() for t in (videos, input_ids, attention_mask)]
            gc.collect()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            step0 = self.step
            try:
                with torch.cuda.graph(g, stream=side):
                    st["out"] = self._forward_backward(*st["inputs"], None, None, None, 1)
                    st["clip"] = self._reduce_and_clip()
            except Exception as e:  # a step that cannot be captured keeps running eagerly (nothing was executed yet)
                print(f"[runner] HIP-graph capture of the step failed ({type(e).__name__}: {e}); staying eager")
                torch.cuda.synchronize()
                st["eager_only"], self.step = True, step0
                out = self._forward_backward(videos, input_ids, attention_mask, None, None, None, 1)
                return self._finish_step(True, self._reduce_and_clip(), *out)
            st["graph"] = g
            st["grads"] = [(p, p.grad) for grp in self.optimizer.param_groups for p in grp["params"]] + \
                          [(p, p.grad) for p in (self._clipper.params if self._clipper is not None else [])]
        else:
            for dst, src in zip(st["inputs"], (videos, input_ids, attention_mask)):
                if dst is not None:
                    dst.copy_(src, non_blocking=True)
        st["graph"].replay()
        for p, gr in st["grads"]:   # the replay rewrote these buffers; keep .grad pointing at them (eager steps of another
            if p.grad is not gr:    # signature may have re-homed the gradients in between)
                p.grad = gr
        return self._finish_step(True, st["clip"], *st["out"], static_outputs=True)

    def _val_step(self, videos: torch.Tensor, input_ids: Optional[torch.Tensor] = None,
                  attention_mask: Optional[torch.Tensor] = None, positive_mask: Optional[torch.Tensor] = None,
                  positive_weights: Optional[torch.Tensor] = None, text_ids=None, text_metadata=None,
                  debug_sample_info=None) -> tuple[dict, dict]:
        with torch.no_grad():
            video_emb, text_emb = self._encode(videos, input_ids, attention_mask)
            diag: dict = {}
            loss = self._loss(video_emb, text_emb, positive_mask, positive_weights, text_metadata, diag)
            m = self._step_metrics(loss, video_emb, text_emb, diag)
            m = {k: v for k, v in m.items() if not k.startswith("lr/")}
            video_fp32 = video_emb.float()
            text_fp32 = text_emb.float() if text_emb is not None else torch.empty(0, device=video_fp32.device)
        return m, {"video_embeddings": video_fp32, "text_embeddings": text_fp32}

    # ----------------------------------------------------------------------------------------------- epochs
    def _gather_tensor_along_batch(self, local_tensor: torch.Tensor, world_size: int) -> torch.Tensor:
        """Variable-length all-gather along dim 0 (reference :494-534): sizes, then one padded all_gather_into_tensor."""
        if _cfg(self.config, "world_size", 1) < 2 or not (dist.is_available() and dist.is_initialized()):
            return local_tensor
        dev = local_tensor.device
        sizes = torch.zeros(world_size, device=dev, dtype=torch.long)
        dist.all_gather_into_tensor(sizes, torch.tensor([local_tensor.shape[0]], device=dev, dtype=torch.long))
        sizes = sizes.tolist()
        mx = max(sizes)
        padded = torch.zeros((mx, *local_tensor.shape[1:]), device=dev, dtype=local_tensor.dtype)
        padded[: local_tensor.shape[0]] = local_tensor
        out = torch.empty((world_size * mx, *local_tensor.shape[1:]), device=dev, dtype=local_tensor.dtype)
        dist.all_gather_into_tensor(out, padded)
        return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world_size)], dim=0)

    def _maybe_reduce_metrics(self, metrics: dict, rank0_only=()) -> dict:
        """One packed all-reduce instead of one per key (reference :1056-1069).  Every rank must pass the same keys.
        `rank0_only`: keys computed on the reference rank alone (the retrieval metrics; the other ranks hold 0.0
        placeholders like the reference's :1023-1033): they are SUMMED, so every rank ends up with rank 0's value — the
        reference averages them with the placeholders and so reports value / world_size, and its placeholder keys
        ("MRR", "NDCG@k") do not even match rank 0's ("MRR_V2T", "NDCG@k_V2T"); neither defect is reproduced."""
        if self.world_size < 2 or not (dist.is_available() and dist.is_initialized()):
            return metrics
        keys = sorted(k for k, v in metrics.items() if isinstance(v, (int, float)))
        t = torch.tensor([float(metrics[k]) for k in keys], device=self.device, dtype=torch.float32)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        vals = t.tolist()
        out = dict(metrics)
        out.update({k: (v if k in rank0_only else v / self.world_size) for k, v in zip(keys, vals)})
        return out

    def _run_epoch(self, mode: str, epoch: int) -> dict:
        mode = str(mode)
        training = mode == "train"
        self.video_encoder.train(training)
        self.text_encoder.train(training)
        loader = self.train_loader if training else self.val_loader
        sums: Dict[str, float] = {}
        count = 0
        all_v, all_t = [], []
        all_texts: List[List[str]] = []   # positive texts per validation video (reference :768-802)
        all_paths: List[str] = []
        for batch, step_inputs, paths in self._prefetched_batches(loader):
            metrics, emb = (self._train_step if training else self._val_step)(**step_inputs)
            if not math.isfinite(metrics["loss"]):
                raise RuntimeError(f"NaN loss encountered in {mode} at step {self.step} (epoch {epoch})")
            for k, v in metrics.items():
                sums[k] = sums.get(k, 0.0) + float(v)
            count += 1
            all_v.append(emb["video_embeddings"])
            all_t.append(emb["text_embeddings"])
            if not training:
                rows = emb["video_embeddings"].shape[0]
                texts = self._batch_positive_texts(batch, rows)
                all_texts.extend(texts)
                plist = list(paths) if paths else []
                all_paths.extend(plist + [""] * (rows - len(plist)))
        out = {f"{mode}/{k}": v / max(count, 1) for k, v in sums.items()}
        retrieval_keys: List[str] = []
        if not training and (all_v or self.world_size > 1):
            # every collective below runs on EVERY rank, before the rank-0-only metric computation (a gather inside the
            # is_ref_device branch deadlocks; a packed all-reduce over different key sets corrupts or hangs)
            dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
            enc = self.video_encoder.module if hasattr(self.video_encoder, "module") else self.video_encoder
            width = int(getattr(enc, "embedding_dim", 512))
            v_local = torch.cat(all_v, 0) if all_v else torch.empty((0, width), device=dev)
            t_local = torch.cat(all_t, 0) if all_t else torch.empty((0, width), device=dev)
            if t_local.dim() != 2:   # steps without texts hand out 1-D empties: every rank still joins the gather below
                t_local = torch.empty((0, width), device=dev, dtype=v_local.dtype)
            v = self._gather_tensor_along_batch(v_local, self.world_size)
            t = self._gather_tensor_along_batch(t_local, self.world_size)   # unconditional: a collective must not hang on rank-local state
            texts = self._gather_objects(all_texts)
            paths_g = self._gather_objects(all_paths)
            tok = getattr(self.text_encoder.module if hasattr(self.text_encoder, "module") else self.text_encoder,
                          "tokenizer", None)
            retrieval_keys = self._retrieval_metric_keys()
            ret = {k: 0.0 for k in retrieval_keys}   # non-reference ranks: placeholders under the same keys (reference :1023-1033)
            if _cfg(self.config, "is_ref_device", True):
                computed = False
                if v.shape[0]:
                    if tok is not None and len(texts) == v.shape[0] and any(x for x in texts):
                        ret.update(self._unique_text_retrieval(v, texts, paths_g, tok, epoch))
                        computed = True
                    elif t.numel() and t.shape[0] == v.shape[0]:
                        # no report texts / no tokenizer: the in-batch pairs are the ground truth (diagonal)
                        ret.update(self._retrieval_metrics(compute_similarity_matrix(v, t), torch.arange(v.shape[0])))
                        computed = True
                if not computed:   # nothing to rank: NaN marks "absent" through the packed all-reduce (same keys on every rank)
                    ret = {k: float("nan") for k in retrieval_keys}
            out.update({f"{mode}/{k}": float(x) for k, x in ret.items() if k in retrieval_keys})
        out = self._maybe_reduce_metrics(out, rank0_only={f"{mode}/{k}" for k in retrieval_keys})
        # retrieval keys the reference rank could not compute are dropped, not reported (and checkpointed) as a real 0.0
        return {k: v for k, v in out.items() if not (isinstance(v, float) and math.isnan(v) and k.split("/", 1)[-1] in retrieval_keys)}

    def _retrieval_metric_keys(self) -> List[str]:
        """The retrieval keys every rank reports at the end of a validation epoch (same order on all ranks)."""
        return [f"Recall@{k}" for k in self.config.recall_k] + ["MRR_V2T"] + [f"NDCG@{k}_V2T" for k in self.config.ndcg_k] + \
               ["MAP", "MedianRank_V2T"]

    # ------------------------------------------------------------------------- epoch-end retrieval (val)
    @staticmethod
    def _batch_positive_texts(batch: dict, rows: int) -> List[List[str]]:
        """Positive prompts of every video of a batch: the active columns of `positive_mask` mapped through
        `text_metadata[*]['prompt_text']`, else the sample's own report (reference :772-800)."""
        reports = list(batch.get("reports", []) or [])
        mask, meta = batch.get("positive_mask"), batch.get("text_metadata")
        out: List[List[str]] = []
        if isinstance(mask, torch.Tensor) and isinstance(meta, list) and mask.numel() > 0:
            m = mask.detach().cpu()
            for r in range(m.size(0)):
                prompts = [meta[c].get("prompt_text", "") for c in (m[r] > 0).nonzero(as_tuple=False).flatten().tolist()
                           if c < len(meta)]
                prompts = [p for p in prompts if p]
                if not prompts and r < len(reports) and reports[r]:
                    prompts = [reports[r]]
                out.append(prompts)
        else:
            out = [[t] if t else [] for t in reports]
        out.extend([] for _ in range(rows - len(out)))
        return out[:rows] if rows else out

    def _gather_objects(self, local: list) -> list:
        if self.world_size <= 1 or not (dist.is_available() and dist.is_initialized()):
            return local
        parts: list = [None] * self.world_size
        dist.all_gather_object(parts, local)
        return [x for p in parts for x in p]

    def _retrieval_metrics(self, sim: torch.Tensor, gt) -> dict:
        out = dict(compute_recall_at_k(sim, gt, self.config.recall_k))
        out.update(compute_mrr(sim, gt))
        out.update(compute_ndcg_at_k(sim, gt, self.config.ndcg_k))
        out["MAP"] = compute_map(sim, gt)
        out["MedianRank_V2T"] = compute_median_rank(sim, gt)
        return out

    def _unique_text_retrieval(self, video_feats: torch.Tensor, positive_texts: List[List[str]], paths: List[str], tokenizer,
                               epoch: int) -> dict:
        """Epoch-end validation retrieval (reference :853-1021): de-duplicate the positive texts, re-encode the unique
        ones in batches of 64 with the text tower, score every video against every unique text ([Nv, Mt] cosine
        similarities on the GPU) and rank the ground-truth index SETS (rank kernel: retrieval_metrics.gt_ranks)."""
        unique = sorted({t for ts in positive_texts for t in ts if t}) or [""]
        index = {t: i for i, t in enumerate(unique)}
        gt_sets: List[List[int]] = []
        for vid, ts in enumerate(positive_texts):
            ids: List[int] = []
            for t in ts:
                j = index.get(t)
                if j is not None and j not in ids:
                    ids.append(j)
            if not ids:
                fallback = ts[0] if ts else ""
                if fallback not in index:
                    raise ValueError(f"Missing ground-truth text in mapping: '{fallback}' for identifier '{paths[vid] if vid < len(paths) else vid}'")
                ids.append(index[fallback])
            gt_sets.append(ids)
        enc = self.text_encoder
        was_training = enc.training
        enc.eval()
        embs = []
        with torch.no_grad():
            for s0 in range(0, len(unique), 64):
                tk = tokenizer(unique[s0:s0 + 64], padding=True, truncation=True, return_tensors="pt")
                e = enc(tk["input_ids"].to(self.device), tk["attention_mask"].to(self.device))
                e = K.nan_to_num_(e.float().contiguous(), 0.0, 1e4, -1e4)
                embs.append(e)
        enc.train(was_training)
        bank = torch.cat(embs, 0)
        sim = compute_similarity_matrix(video_feats, bank)   # L2-normalises both sides (reference steps 4-5)
        out = self._retrieval_metrics(sim, gt_sets)
        if self.output_dir:
            import csv
            os.makedirs(self.output_dir, exist_ok=True)
            with open(os.path.join(self.output_dir, "val_unique_texts.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Index", "Text"])
                w.writerows(enumerate(unique))
            torch.save({"text_embeddings": bank.cpu(), "text_to_index": index, "unique_texts": unique},
                       os.path.join(self.output_dir, f"val_text_embeddings_epoch_{epoch}.pt"))
        return out

    def train(self, start_epoch: int, end_epoch: int) -> None:
        for epoch in range(start_epoch, end_epoch):
            DistributedUtils.sync_process_group(self.world_size, self.device)
            for enc, key in ((self.video_encoder, "video_freeze_ratio"), (self.text_encoder, "text_freeze_ratio")):
                mod = enc.module if hasattr(enc, "module") else enc
                ratio = _cfg(self.config, key, None)
                if ratio is not None and hasattr(mod, "update_freeze_ratio"):
                    mod.update_freeze_ratio(ratio)
            for loader in (self.train_loader, self.val_loader):
                sampler = getattr(loader, "sampler", None)
                if sampler is not None and hasattr(sampler, "set_epoch"):
                    sampler.set_epoch(epoch)
            train_metrics = self._run_epoch("train", epoch)
            val_metrics = self._run_epoch("val", epoch) if self.val_loader is not None else {}
            if self.lr_scheduler and not self.scheduler_per_iteration:
                self.lr_scheduler.step()
            if self.wandb_wrapper is not None and getattr(self.wandb_wrapper, "is_initialized", lambda: False)():
                self.wandb_wrapper.log({**train_metrics, **val_metrics, "epoch": epoch})
            # best / highest-alignment bookkeeping on every rank (reference :385-430), files on the reference rank only
            save = _cfg(self.config, "is_ref_device", True) and self.output_dir
            all_metrics = {**train_metrics, **val_metrics}
            val_loss = val_metrics.get("val/loss", float("inf"))
            if val_loss < self.best_val_loss:
                self.best_val_loss, self.best_epoch = val_loss, epoch
                if save:
                    self._save_checkpoint(epoch, all_metrics, is_best=True)
            align = val_metrics.get("val/alignment_score", float("-inf"))
            if align > self.highest_alignment_score:
                self.highest_alignment_score, self.highest_alignment_epoch = align, epoch
                if save:
                    self._save_checkpoint(epoch, all_metrics, is_best=False, is_highest_alignment=True)
            if save:
                self._save_checkpoint(epoch, all_metrics, is_best=False)
            DistributedUtils.sync_process_group(self.world_size, self.device)
            patience = _cfg(self.config, "early_stop_patience", 0)
            if patience and patience > 0 and self.best_epoch >= 0 and epoch - self.best_epoch >= patience:
                break

    def _save_checkpoint(self, epoch: int, metrics: dict, is_best: bool = False, is_highest_alignment: bool = False) -> None:
        """Reference checkpoint layout and file names (:1071-1115): encoders unwrapped from DDP, optimizer, scheduler,
        scaler, epoch, the epoch metrics, best_val_loss / best_epoch / highest_alignment_score / highest_alignment_epoch."""
        ckpt_dir = os.path.join(self.output_dir, "checkpoints")
        os.makedirs(ckpt_dir, exist_ok=True)
        unwrap = lambda m: m.module if hasattr(m, "module") else m  # noqa: E731
        state = {
            "video_encoder": unwrap(self.video_encoder).state_dict(),
            "text_encoder": unwrap(self.text_encoder).state_dict(),
            "optimizer": self.optimizer.state_dict() if self.optimizer else None,
            "scheduler": self.lr_scheduler.state_dict() if self.lr_scheduler else None,
            "epoch": epoch,
            "scaler": self.scaler.state_dict() if self.scaler else None,
            **metrics,
            "best_val_loss": self.best_val_loss, "best_epoch": self.best_epoch,
            "highest_alignment_score": self.highest_alignment_score,
            "highest_alignment_epoch": self.highest_alignment_epoch,
        }
        if is_best:
            name = f"best_model_epoch_{epoch}.pt"
        elif is_highest_alignment:
            name = f"highest_alignment_epoch_{epoch}.pt"
        else:
            name = "checkpoint.pt"
        torch.save(state, os.path.join(ckpt_dir, name))

    def validate(self) -> None:
        raise NotImplementedError("Validation is not implemented for this runner")

    def inference(self, text_embeddings: torch.Tensor | None = None, topk: int = 5):
        """Top-k text retrieval for every validation video (reference :1731-1758): [B,512] x [512,M] raw inner products
        (`torch.matmul(video_embeddings, text_embeddings.t())`, no normalisation) + top-k indices.  The reference then
        averages metadata rows of a parquet file over those indices (pandas, out of scope); the bank is passed in."""
        if text_embeddings is None:
            raise ValueError("inference() needs the [M, 512] text embedding bank")
        self.video_encoder.eval()
        results = []
        with torch.no_grad():
            bank = text_embeddings.to(self.device).float().contiguous()
            for batch in self.val_loader:
                v = self.video_encoder(batch["videos"].to(self.device).float())
                sim = K.gemm(v.float().contiguous(), bank)  # raw inner products, not cosines: reference :1757
                results.append(K.topk_rows(sim.float().contiguous(), min(topk, bank.shape[0]))[1].cpu())
        return torch.cat(results, 0) if results else torch.empty(0, dtype=torch.long)
