#!/usr/bin/env python
"""bench.py — video-text clip-pairs/sec of the contrastive training step on N MI355X GPUs of one node.

A "step" = one `_train_step` (VideoEncoder fwd/bwd + TextEncoder fwd/bwd + InfoNCE loss + gradient reduction +
clip + AdamW) on one synthetic batch that is already resident in HBM.  One process per GPU (torch.distributed /
RCCL).  The default workload is the one BASELINE.json's metric is quoted on — configs[2], "16f x 224^2 multi-video":
per rank B = 8 studies x N = 4 videos (32 clips of 16 x 224 x 224), MViT-v2-S + AttentionPool + study aggregator,
BERT-base S = 32, bf16, global-batch InfoNCE with the embedding all-gather for N > 1.  `--videos 1 --batch 24
--pooling mean` gives configs[1] (single-video CLIP, config/clip/base_config.yaml:23).  Weak scaling: the per-GPU
batch is fixed, `value` = clips (each with its study's text) per second over all ranks.

Prints ONE JSON line on rank 0 (contract in the task statement), with:
  roofline     — the dominant kernel family (bf16 MFMA GEMMs) timed per launch with HIP events on the launch stream
                 in an instrumented pass of the same K steps right after the timed region (events off during it),
                 against BOTH ceilings (MFMA peak = the headline `frac`, as SURVEY §8d classifies the family; algorithmic bytes /
                 8 TB/s beside it in `hbm`, `binding_by_bytes_moved` = the larger of the two), plus
                 `video_fwd_frac`: the VideoEncoder forward alone against the bf16 MFMA peak (north_star's target)
  cpu_baseline — the CPU oracle (oracle/, a port) doing the same training step on a bounded sample, rank 0, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0  # spec; ~6300 achievable


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def build_job(args, device, world, rank):
    from deepcoro_clip_amd.ddp import HipDDP
    from deepcoro_clip_amd.losses import Loss
    from deepcoro_clip_amd.optim import HipAdamW
    from deepcoro_clip_amd.registry import LossRegistry, register_submodules
    from deepcoro_clip_amd.runner import VideoContrastiveLearningRunner
    from deepcoro_clip_amd.text_encoder import TextEncoder
    from deepcoro_clip_amd.video_encoder import VideoEncoder

    register_submodules("deepcoro_clip_amd")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)  # same random-init weights on every rank
    ve = VideoEncoder(backbone="mvit", num_frames=args.frames, spatial_size=(args.size, args.size), pretrained=False,
                      freeze_ratio=1.0, dropout=0.0, token_pooling_mode=args.pooling, attention_pool_dropout=0.0,
                      compute_dtype=dtype, stochastic_depth_prob=0.0,  # dropout 0, stochastic depth 0
                      use_rope=args.rope, per_video_pool=False, gemm_dtype=args.gemm_dtype).to(device)
    te = TextEncoder(dropout=0.0, freeze_ratio=0.0, compute_dtype=dtype,
                     bert_config=dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).to(device)
    ve.train()
    te.train()
    vew, tew = ve, te
    if world > 1:
        vew, tew = HipDDP(ve, device_ids=[device.index]), HipDDP(te, device_ids=[device.index])
    log_temp = torch.nn.Parameter(torch.log(torch.tensor([0.07], device=device)))
    # param groups of projects/contrastive_pretraining_project.py:128-179
    groups = [{"params": list(ve.model.parameters()), "lr": 1e-4, "name": "video_backbone", "weight_decay": 1e-6}]
    if ve.attention_pool is not None:
        groups.append({"params": list(ve.attention_pool.parameters()), "lr": 2e-4, "name": "video_attention_pool", "weight_decay": 1e-6})
    groups += [{"params": list(ve.aggregator.parameters()), "lr": 2e-4, "name": "video_aggregator", "weight_decay": 1e-6},
               {"params": list(te.parameters()), "lr": 2e-5, "name": "text_encoder", "weight_decay": 1e-6},
               {"params": [log_temp], "lr": 1e-4, "name": "temperature"}]
    opt = HipAdamW(groups, lr=1e-4)
    loss_name = args.loss + ("_ddp" if world > 1 else "")
    cfg = _Cfg(recall_k=[1, 5], ndcg_k=[5], device=device, world_size=world, gradient_accumulation_steps=1,
               max_grad_norm=5.0, is_ref_device=rank == 0, loss_name=loss_name, scheduler_name="", temperature=0.07,
               embedding_cache_micro_batch=args.micro_batch)
    runner = VideoContrastiveLearningRunner(config=cfg, wandb_wrapper=None, train_loader=None, val_loader=None,
                                            video_encoder=vew, text_encoder=tew, optimizer=opt, scaler=None,
                                            log_temp=log_temp, lr_scheduler=None,
                                            loss_fn=Loss(loss_type=LossRegistry.get(loss_name)()), output_dir=None)
    g = torch.Generator().manual_seed(1234 + rank)
    shape = (args.batch, args.videos, args.frames, args.size, args.size, 3) if args.videos > 1 else \
        (args.batch, args.frames, args.size, args.size, 3)
    videos = torch.randn(shape, generator=g).to(device)
    ids = torch.randint(0, 30522, (args.batch, args.seq), generator=g).to(device)
    mask = torch.ones(args.batch, args.seq, dtype=torch.int64, device=device)
    mode = getattr(args, "graph", "off")
    # the first three calls of a signature run eagerly and the fourth captures: keep all of that inside the warm-up
    runner._use_step_graph = ({"on": True, "auto": "auto"}.get(mode, False)) if (world == 1 and args.warmup >= 4) else False
    return runner, dict(videos=videos, input_ids=ids, attention_mask=mask)


def _host_cores() -> int:
    """CPU share of this process: affinity mask, capped by the cgroup quota; a GPU box gives 16 cores per GPU, and
    oversubscribing torch's thread pool beyond the quota makes the baseline collapse rather than speed up."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    cores = min(cores, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, min(cores, 16))


def cpu_baseline(args):
    """The CPU oracle (a port of the reference algorithm) timed on this host: fwd + bwd + AdamW, bounded sample."""
    from oracle.losses import contrastive_loss
    from oracle.model import TextEncoderOracle, VideoEncoderOracle, init_by_name

    cores = _host_cores()
    torch.set_num_threads(cores)
    N = max(1, args.videos)
    B = 2 if N == 1 else 1   # bounded sample: 2 single-video studies, or 1 study of N videos (same clip shapes)
    vo = init_by_name(VideoEncoderOracle(args.frames, (args.size, args.size), token_pooling_mode=args.pooling, use_rope=args.rope))
    to = init_by_name(TextEncoderOracle())
    lt = torch.nn.Parameter(torch.log(torch.tensor([0.07])))
    opt = torch.optim.AdamW(list(vo.parameters()) + list(to.parameters()) + [lt], lr=1e-4)
    g = torch.Generator().manual_seed(1234)
    videos = torch.randn(*((B, N) if N > 1 else (B,)), args.frames, args.size, args.size, 3, generator=g)
    ids = torch.randint(0, 30522, (B, args.seq), generator=g)
    mask = torch.ones(B, args.seq, dtype=torch.int64)

    loss_fn = (lambda v, t, l: contrastive_loss(v, t, l, gated=True)) if args.loss == "siglip" else contrastive_loss

    def step():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(vo(videos), to(ids, mask), lt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(vo.parameters()) + list(to.parameters()), 5.0)
        opt.step()

    tw = time.perf_counter()
    step()  # warm-up (allocator, thread pool)
    print(f"[bench] cpu_baseline: warm-up step {time.perf_counter() - tw:.1f} s on {cores} threads", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    times = []
    while len(times) < 1 or (time.perf_counter() - t0 < 20.0 and len(times) < 10):
        ts = time.perf_counter()
        step()
        times.append(time.perf_counter() - ts)
        print(f"[bench] cpu_baseline: {len(times)} timed steps, {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    med = sorted(times)[len(times) // 2]
    return {"value": round(B * N / med, 4), "unit": "clip-pairs/s", "cores": cores, "kind": "port",
            "sample": f"median of {len(times)} fp32 training steps (after 1 warm-up) of {B} stud{'y' if B == 1 else 'ies'} x {N} clip(s) "
                      f"{args.frames}x{args.size}x{args.size} + {args.seq}-token texts, {args.pooling} pooling, oracle/ (plain PyTorch CPU), "
                      f"torch.set_num_threads({cores})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (studies)")
    ap.add_argument("--videos", type=int, default=4, help="videos per study")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--seq", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--pooling", default="attention", choices=["mean", "attention"])
    ap.add_argument("--loss", default="contrastive", choices=["contrastive", "siglip"],
                    help="registry name of the pair loss ('_ddp' is appended for N > 1)")
    ap.add_argument("--rope", action="store_true", help="3-D RoPE attention in blocks 3-15 (models/video_encoder.py:268-378)")
    ap.add_argument("--gemm-dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: the video trunk's forward / data-gradient GEMMs on the block-scaled fp8 MFMA (BASELINE configs[4])")
    ap.add_argument("--micro-batch", type=int, default=0,
                    help="embedding-cache micro-batch (studies): the batch is encoded in slices of this size, the loss sees all of it")
    ap.add_argument("--config", default="", choices=["", "siglip4096", "fp8rope32"],
                    help="presets of BASELINE.json configs[3] / configs[4] per-rank shards (flags given explicitly still win): "
                         "siglip4096 = 512 single-video studies per rank through the embedding cache (micro-batch 32), SigLIP; "
                         "fp8rope32 = 4 studies x 8 videos of 32 x 224 x 224, RoPE tokens, fp8 GEMMs")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="runner.use_step_graph: replay forward+backward+clip from a HIP graph (1 rank). auto = only when the "
                         "warm-up calls show the host needs about as long to launch a step as the GPU to run it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-kernels", type=int, default=0, help="print the N slowest timed kernel shapes (stderr)")
    args = ap.parse_args()
    presets = {"siglip4096": dict(batch=512, videos=1, loss="siglip", micro_batch=32, pooling="mean"),
               "fp8rope32": dict(batch=4, videos=8, frames=32, rope=True, gemm_dtype="fp8")}
    given = {a.lstrip("-").split("=")[0].replace("-", "_") for a in sys.argv[1:] if a.startswith("--")}
    for k, v in presets.get(args.config, {}).items():
        if k not in given:
            setattr(args, k, v)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    # DCC_BENCH_DEVICE / DCC_BENCH_BACKEND: rehearsal knobs (several ranks sharing the one GPU of a test box over
    # gloo); the driver's multi-GPU runs use one GPU per rank and RCCL
    dev_index = int(os.environ.get("DCC_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DCC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from deepcoro_clip_amd import kernels as K

    runner, batch = build_job(args, device, world, rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss0 = None
    for _ in range(args.warmup):
        m, _ = runner._train_step(**batch)
        loss0 = m["loss"] if loss0 is None else loss0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m, _ = runner._train_step(**batch)
    graphs = list(getattr(runner, "_graphs", {}).values())
    step_graph = ("replay" if any(g.get("graph") is not None for g in graphs) else "eager") + \
                 "".join(f" (auto: host {g['decision'][0]:.1f} ms vs gpu {g['decision'][1]:.1f} ms per warm-up step)" for g in graphs if g.get("decision"))
    t_host = time.perf_counter() - t0  # launches enqueued; the metrics are not read, so nothing drained the queue yet
    barrier()
    dt = time.perf_counter() - t0
    # host cost of a step WITHOUT queue back-pressure: in the free-running timed region a GPU-bound step makes the HIP queue fill
    # up and the enqueue calls block, so `t_host` approaches the wall time whatever the host needs; one step launched into an
    # EMPTY queue returns after the host's own work (Python + ctypes + hipLaunchKernel); median of three, outside the timed region
    host_free = []
    for _ in range(3):
        torch.cuda.synchronize()
        th = time.perf_counter()
        runner._train_step(**batch)
        host_free.append(time.perf_counter() - th)
    barrier()
    host_free_ms = 1e3 * sorted(host_free)[1]
    peak_hbm_gb = round(torch.cuda.max_memory_allocated() / 2**30, 1)   # of the product path: before the instrumented pass below
    if rank == 0:
        print(f"[bench] host enqueue {1e3 * t_host / args.steps:.2f} ms/step in the timed region (queue back-pressure included), "
              f"{host_free_ms:.2f} ms/step into an empty queue; wall {1e3 * dt / args.steps:.2f} ms/step", file=sys.stderr, flush=True)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    pairs_per_step = args.batch * world  # one text per study; N videos per study counted as one pair-study
    clips_per_step = args.batch * args.videos * world
    value = clips_per_step * args.steps / dt

    roofline = None
    if not args.no_roofline:
        K.TIMER = K.KernelTimer()
        runner._use_step_graph = False  # the event pairs bracket individual launches: instrumented pass always eager
        runner._wgrad_side = False      # ... and on the stream the timer records its events on
        for _ in range(args.steps):
            mi, _ = runner._train_step(**batch)
            mi["loss"]  # read like _run_epoch does: a per-step drain keeps the event pairs free of the launch-queue
            #             contention between streams that the free-running timed region has (events bracket launches)
        summ = K.TIMER.summary()
        if args.dump_kernels and rank == 0:
            for fam, lab, n, ms, units in K.TIMER.by_label()[: args.dump_kernels]:
                print(f"[kernels] {fam:9s} {ms / args.steps:8.3f} ms/step {n // args.steps:4d}x  {units / max(ms, 1e-9) / 1e9:8.1f} TFLOP/s  {lab}", file=sys.stderr)
        K.TIMER = None
        fam = "gemm_bf16" if args.dtype == "bf16" else "gemm_f32"
        if fam in summ and summ[fam]["ms"] > 0:
            s = summ[fam]
            secs = s["ms"] * 1e-3
            tflops = s["units"] / secs / 1e12
            gbs = s["bytes"] / secs / 1e9
            peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
            mfma_frac, hbm_frac = tflops / peak, gbs / HBM_PEAK_GBS
            # SURVEY §8(d) classifies this family as MFMA-bound: the headline `frac` is the MFMA fraction.  The HBM view stays beside it
            # (`hbm`): by the bytes THIS design moves (saved pre-activations, LayerNorm outputs, residual re-reads) the family sits
            # at a higher fraction of the HBM ceiling than of the MFMA one — `binding_by_bytes_moved` says which.
            bound = "mfma" if args.dtype == "bf16" else ("mfma" if mfma_frac >= hbm_frac else "hbm")
            binding = "mfma" if mfma_frac >= hbm_frac else "hbm"
            traffic = None  # HBM bytes per launch from the PMC passes (rocprofv3 cannot run inside this process)
            traffic_source = None
            import glob
            plain = args.gemm_dtype == "bf16" and not args.rope and args.loss == "contrastive" and not args.micro_batch
            for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gemm_traffic.json")), reverse=True):
                if world != 1 or not plain:
                    break
                rec = json.load(open(pmc))  # newest round first: the PMC pass of the kernels this tree actually ships
                if rec.get("workload") == [args.dtype, args.batch, args.videos, args.frames, args.size, args.pooling]:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = "committed PMC pass " + os.path.relpath(pmc, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; not observed in this process)"
                    break
            roofline = {"bound": bound, "kernel": "bf16 GEMM family (gemm_ring2_kernel / gemm_ring_kernel / gemm_wide_kernel / gemm_stream_kernel / gemm_p8_kernel / ln_mlp_kernel / gemm_kernel)"
                        if args.dtype == "bf16" else "fp32 GEMM family (gemm_kernel)",
                        "achieved": round(tflops if bound == "mfma" else gbs, 2), "peak": peak if bound == "mfma" else HBM_PEAK_GBS,
                        "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": round(mfma_frac if bound == "mfma" else hbm_frac, 4),
                        "traffic": traffic, "traffic_source": traffic_source, "binding_by_bytes_moved": binding,
                        "roofline_pass": "instrumented pass after the timed region: HIP events around every launch, per-step drain, side streams off",
                        "mfma": {"achieved_tflops": round(tflops, 2), "peak": peak, "frac": round(mfma_frac, 4)},
                        "hbm": {"achieved_gbs": round(gbs, 1), "peak": HBM_PEAK_GBS, "frac": round(hbm_frac, 4)},
                        "algorithmic_bytes_per_launch": round(s["bytes"] / max(1, s["launches"])),
                        "launches_per_step": s["launches"] // args.steps,
                        "avg_launch_us": round(1e3 * s["ms"] / s["launches"], 2),
                        "gflop_per_step": round(s["units"] / args.steps / 1e9, 1),
                        "ms_per_step_in_kernel": round(s["ms"] / args.steps, 2),
                        "other_families_ms_per_step": {k: round(v["ms"] / args.steps, 2) for k, v in summ.items() if k != fam}}
        # north_star: ">= 30 % of bf16 MFMA peak on the VideoEncoder forward" — the training-mode forward alone (autograd
        # on, activations saved, as inside the step), K back-to-back calls bracketed by one HIP-event pair on the launch
        # stream; algorithmic FLOPs = forward_macs (128.45 GFLOP per 16x224^2 clip) + proj + pooling heads
        from deepcoro_clip_amd.mvit import forward_macs
        ve = runner.video_encoder.module if hasattr(runner.video_encoder, "module") else runner.video_encoder
        fwd_videos = batch["videos"]
        if args.micro_batch and fwd_videos.shape[0] > args.micro_batch:
            # embedding-cache workloads never hold the whole batch's activations: time ONE micro-batch's forward (r03: this leg ran
            # all 512 clips of configs[3] at once and put 272.7 GB into the line's peak-memory field)
            fwd_videos = fwd_videos[:args.micro_batch]
        clips = fwd_videos.shape[0] * max(1, args.videos)
        tokens = ve.model.plan[-1]["q_thw"][0] * ve.model.plan[-1]["q_thw"][1] * ve.model.plan[-1]["q_thw"][2] + 1
        flop_clip = 2.0 * forward_macs(ve.model.plan)["total"] + 2.0 * tokens * 768 * 512
        if args.pooling == "attention":
            flop_clip += 2.0 * tokens * 512 * 1024 + 4.0 * tokens * 512  # K/V in-projection + 1-query attention
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(3, min(args.steps, 10))
        ve(fwd_videos)  # warm
        fwd_ms, host_s = float("inf"), 0.0
        for _ in range(2):   # two timed batches, the faster one counts (a cold host can starve the first)
            th0 = time.perf_counter()
            e0.record()
            for _ in range(reps):
                out_v = ve(fwd_videos)
                del out_v
            e1.record()
            th1 = time.perf_counter()
            torch.cuda.synchronize()
            if e0.elapsed_time(e1) / reps < fwd_ms:
                fwd_ms, host_s = e0.elapsed_time(e1) / reps, th1 - th0
        th0, th1 = 0.0, host_s
        fwd_tflops = flop_clip * clips / (fwd_ms * 1e-3) / 1e12
        if roofline is not None:
            roofline["video_fwd"] = {"ms": round(fwd_ms, 3), "gflop": round(flop_clip * clips / 1e9, 1), "tflops": round(fwd_tflops, 1),
                                     "host_enqueue_ms": round(1e3 * (th1 - th0) / reps, 3)}
            roofline["video_fwd_frac"] = round(fwd_tflops / (BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS), 4)
    barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        out = {
            "metric": "video-text clip-pairs/sec (whole node), contrastive training step",
            "value": round(value, 3), "unit": "clip-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"single-video CLIP step: MViT-v2-S {args.frames}x{args.size}x{args.size} + BERT-base "
                                   f"S={args.seq}, InfoNCE, AdamW, B={args.batch}/GPU x N={args.videos} videos/study"
                       if args.videos == 1 else
                       f"multi-video CLIP step: MViT-v2-S {args.frames}x{args.size}x{args.size} x N={args.videos} + "
                       f"{args.pooling} pooling + aggregator, BERT-base S={args.seq}, B={args.batch}/GPU"
                       + (", 3-D RoPE" if args.rope else "") + (", fp8 trunk GEMMs" if args.gemm_dtype == "fp8" else ""),
                       "extras": {"loss": args.loss, "rope": bool(args.rope), "gemm_dtype": args.gemm_dtype,
                                  "embedding_cache_micro_batch": args.micro_batch, "preset": args.config},
                       # `value` counts CLIPS (each with its study's text) per second; a study of N videos forms one
                       # video-text pair of the loss matrix: both conventions are recorded so rounds stay comparable
                       "value_convention": "clips/s = studies/s x videos per study",
                       "study_pairs_per_s": round(pairs_per_step * args.steps / dt, 3),
                       "global_batch": pairs_per_step, "clips_per_step": clips_per_step, "seq_len": args.seq,
                       "parallelism": f"dp{world}", "loss_first": loss0, "loss_last": m["loss"],
                       "step_graph": step_graph, "host_ms_per_step_unblocked": round(host_free_ms, 2),
                       "peak_hbm_gb": peak_hbm_gb,
                       "peak_hbm_gb_instrumented_pass": round(torch.cuda.max_memory_allocated() / 2**30, 1)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
