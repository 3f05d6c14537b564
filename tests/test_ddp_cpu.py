"""world_size-2 gloo tests of the data-parallel exchange steps: HipDDP's flat-arena bucketed gradient
averaging and GatherLayer's "gather forward / own-rows backward" rule, on CPU tensors (the communication code is
device-agnostic; the GPU path swaps gloo for RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                               torch.nn.Linear(64, 8))


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(5, 16, generator=g), torch.randn(5, 8, generator=g)


def _ddp_worker(rank, world, port, bucket_mb, q):
    try:
        _init(rank, world, port)
        from deepcoro_clip_amd.ddp import HipDDP

        m = _model(seed=rank)  # different init per rank: the wrapper must broadcast rank 0's weights
        w = HipDDP(m, bucket_mb=bucket_mb)
        ref = _model(seed=0)
        for p, r in zip(m.parameters(), ref.parameters()):
            assert torch.equal(p, r), "parameters were not broadcast from rank 0"
        nb = len(w._buckets)
        nparams = len(list(m.parameters()))
        calls = []
        for it in range(4):  # later iterations re-attach / re-zero the arena; from the second on: one hook per bucket
            if it == 3:      # the trainable set changes (update_freeze_ratio): the wrapper learns the order again
                m[0].weight.requires_grad_(False)
            w.attach_grads()
            x, y = _data(rank)
            c0 = w.hook_calls
            ((w(x) - y) ** 2).mean().backward()
            calls.append(w.hook_calls - c0)
            w.finish()
            # expected: mean over ranks of the per-rank gradient
            exp = [torch.zeros_like(p) for p in ref.parameters()]
            for r in range(world):
                ref.zero_grad()
                xr, yr = _data(r)
                ((ref(xr) - yr) ** 2).mean().backward()
                for e, p in zip(exp, ref.parameters()):
                    e += p.grad / world
            for p, e in zip(m.parameters(), exp):
                if not p.requires_grad:
                    continue
                assert p.grad.data_ptr() >= w._arena.data_ptr()  # a view into the flat arena
                assert torch.allclose(p.grad, e, atol=1e-6), (it, (p.grad - e).abs().max())
        # host cost: per-parameter hooks only while the arrival order is being learned
        assert calls[0] == nparams and calls[1] == nb and calls[2] == nb and calls[3] == nparams - 1, (calls, nparams, nb)
        q.put((rank, "ok", nb))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}", 0))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(worker, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, *args, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    return sorted(res)


@pytest.mark.parametrize("bucket_mb", [64.0, 0.004])
def test_hipddp_gradient_average_world2(bucket_mb):
    res = _run(_ddp_worker, bucket_mb)
    assert [r[1] for r in res] == ["ok", "ok"], res
    if bucket_mb < 1:
        assert res[0][2] > 1, "small bucket size must split the arena into several buckets"
    else:
        assert res[0][2] == 1


def _gather_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from deepcoro_clip_amd.losses import gather_all
        from oracle.losses import contrastive_loss

        B, D = 3, 32
        g = torch.Generator().manual_seed(7)
        V, T = torch.randn(world * B, D, generator=g), torch.randn(world * B, D, generator=g)
        v = V[rank * B:(rank + 1) * B].clone().requires_grad_(True)
        t = T[rank * B:(rank + 1) * B].clone().requires_grad_(True)
        lt = torch.log(torch.tensor(0.07))
        gv, gt = gather_all(v), gather_all(t)
        assert torch.equal(gv, V) and torch.equal(gt.detach(), T)
        loss = contrastive_loss(gv, gt, lt)  # global [Bg, Bg] problem evaluated on every rank
        loss.backward()
        Vr, Tr = V.clone().requires_grad_(True), T.clone().requires_grad_(True)
        ref = contrastive_loss(Vr, Tr, lt)
        ref.backward()
        assert abs(loss.item() - ref.item()) < 1e-6
        # reference convention (utils/loss/losses.py:85-93): each rank keeps its own rows of the global gradient
        assert torch.allclose(v.grad, Vr.grad[rank * B:(rank + 1) * B], atol=1e-7)
        assert torch.allclose(t.grad, Tr.grad[rank * B:(rank + 1) * B], atol=1e-7)
        # DistributedUtils helpers
        from deepcoro_clip_amd.ddp import DistributedUtils as DU

        assert DU.is_initialized() and DU.gather_object({"r": rank}, world) == [{"r": 0}, {"r": 1}]
        assert torch.equal(DU.gather_tensor(torch.full((2, 2), float(rank)), world)[::2, 0], torch.tensor([0.0, 1.0]))
        q.put((rank, "ok", 0))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}", 0))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_gather_layer_world2():
    res = _run(_gather_worker)
    assert [r[1] for r in res] == ["ok", "ok"], res


def test_gather_is_identity_without_process_group():
    from deepcoro_clip_amd.losses import gather_all

    x = torch.randn(4, 8, requires_grad=True)
    assert gather_all(x) is x


def _accum_worker(rank, world, port, q):
    """Gradient accumulation under HipDDP = torch DDP's no_sync(): accumulate-only micro-steps, ONE reduction from the
    hooks of the window's last backward; result = mean over ranks of the summed micro-batch gradients."""
    try:
        _init(rank, world, port)
        from deepcoro_clip_amd.ddp import HipDDP

        m = _model(seed=0)
        w = HipDDP(m, bucket_mb=0.004)  # several buckets
        ref = _model(seed=0)
        accum = 3

        def micro(r, i):
            g = torch.Generator().manual_seed(1000 + 10 * r + i)
            return torch.randn(4, 16, generator=g), torch.randn(4, 8, generator=g)

        for window in range(2):
            w.attach_grads()
            for i in range(accum):
                w.begin_backward(sync=(i == accum - 1))
                x, y = micro(rank, i + accum * window)
                (((w(x) - y) ** 2).mean() / accum).backward()
                if i < accum - 1:
                    assert not w._handles, "a non-final micro-step must not launch collectives"
            w.finish()
            exp = [torch.zeros_like(p) for p in ref.parameters()]
            for r in range(world):
                ref.zero_grad()
                for i in range(accum):
                    xr, yr = micro(r, i + accum * window)
                    (((ref(xr) - yr) ** 2).mean() / accum).backward()
                for e, p in zip(exp, ref.parameters()):
                    e += p.grad / world
            for p, e in zip(m.parameters(), exp):
                assert torch.allclose(p.grad, e, atol=1e-6), (window, (p.grad - e).abs().max())
        # ranks agree bit for bit (same reduced arena everywhere)
        flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        other = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert torch.equal(other[0], other[1])
        # a parameter outside every optimizer group keeps its history across attach_grads()
        keep = list(m.parameters())[-1]
        w.persistent_grad_ids = {id(keep)}
        before = keep.grad.clone()
        w.attach_grads()
        x, y = micro(rank, 99)
        ((w(x) - y) ** 2).mean().backward()
        w.finish()
        ref.zero_grad()
        new = torch.zeros_like(before)
        for r in range(world):
            xr, yr = micro(r, 99)
            ref.zero_grad()
            ((ref(xr) - yr) ** 2).mean().backward()
            new += list(ref.parameters())[-1].grad / world
        assert torch.allclose(keep.grad, before + new, atol=1e-6)
        q.put((rank, "ok", 0))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", 0))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_hipddp_gradient_accumulation_world2():
    res = _run(_accum_worker)
    assert [r[1] for r in res] == ["ok", "ok"], res


def _bf16_wire_worker(rank, world, port, q):
    """Optional bf16 bucket transport (half the xGMI bytes): the reduced gradients equal the mean of the per-rank gradients
    after ONE bf16 rounding of each rank's contribution (relative error <= 2^-8 per element), identical on every rank, and
    the fp32 default is untouched."""
    try:
        _init(rank, world, port)
        from deepcoro_clip_amd.ddp import HipDDP

        m, ref = _model(seed=0), _model(seed=0)
        w = HipDDP(m, bucket_mb=0.004, grad_wire_dtype=torch.bfloat16)
        for it in range(3):
            w.attach_grads()
            x, y = _data(rank)
            ((w(x) - y) ** 2).mean().backward()
            w.finish()
            per_rank = []
            for r in range(world):
                ref.zero_grad()
                xr, yr = _data(r)
                ((ref(xr) - yr) ** 2).mean().backward()
                per_rank.append([p.grad.clone() for p in ref.parameters()])
            for i, p in enumerate(m.parameters()):
                exact = sum(g[i] for g in per_rank) / world
                wire = sum(g[i].bfloat16().float() for g in per_rank).bfloat16().float() / world   # what the wire computes
                bound = sum(g[i].abs() for g in per_rank) / world * 2.0 ** -7 + 1e-12
                assert ((p.grad - exact).abs() <= bound).all(), (it, i, (p.grad - exact).abs().max())
                assert torch.allclose(p.grad, wire, rtol=2.0 ** -7, atol=1e-9)
                assert p.grad.dtype == torch.float32
        flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        other = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert torch.equal(other[0], other[1])
        q.put((rank, "ok", 0))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", 0))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_hipddp_bf16_bucket_transport_world2():
    res = _run(_bf16_wire_worker)
    assert [r[1] for r in res] == ["ok", "ok"], res


class _SlotLinear(torch.autograd.Function):
    """y = x W^T whose weight gradient is written straight into the parameter's arena slot (what the HIP weight-gradient GEMMs
    do through _grad_slots.take): autograd adopts the slot alias, so `grad.data_ptr() == view.data_ptr()` without any copy."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.w = w
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        from deepcoro_clip_amd import _grad_slots

        x, w = ctx.saved_tensors
        dw = dy.t() @ x
        slot = _grad_slots.take(ctx.w)
        if slot is not None:
            slot.copy_(dw)
            dw = slot
        return dy @ w, dw


class _Branchy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(16, 32)
        self.extra = torch.nn.Parameter(torch.randn(32, 32) * 0.1)   # unused unless use_extra
        self.b = torch.nn.Linear(32, 8)

    def forward(self, x, use_extra):
        h = torch.nn.functional.gelu(self.a(x))
        if use_extra:
            h = h + _SlotLinear.apply(h, self.extra)
        return self.b(h)


def _late_param_worker(rank, world, port, q):
    """a parameter without a gradient in the learning backward that gets one later — written straight
    into its arena slot — must not slip past the one-hook-per-bucket mode: its bucket is deferred to finish(), the values are
    right on every rank, and the wrapper learns the order again.  Same for a parameter that becomes `persistent`."""
    try:
        _init(rank, world, port)
        from deepcoro_clip_amd.ddp import HipDDP

        m, ref = _Branchy(), _Branchy()
        w = HipDDP(m, bucket_mb=0.002)   # several buckets; `extra` shares one with neighbours
        assert len(w._buckets) > 1
        modes = []
        for it, use in enumerate([False, False, True, True, True]):
            w.attach_grads()
            modes.append(w._mode)
            x, y = _data(rank)
            ((w(x, use) - y) ** 2).mean().backward()
            if it == 2:
                b = w._bucket_of[id(m.extra)]
                assert b in w._deferred and w._relearn, "the late gradient must defer its bucket"
                assert m.extra.grad.data_ptr() == w._views[id(m.extra)].data_ptr()  # it did land in the slot directly
            w.finish()
            exp = [torch.zeros_like(p) for p in ref.parameters()]
            for r in range(world):
                ref.zero_grad()
                xr, yr = _data(r)
                ((ref(xr, use) - yr) ** 2).mean().backward()
                for e, p in zip(exp, ref.parameters()):
                    if p.grad is not None:
                        e += p.grad / world
            for (n, p), e in zip(m.named_parameters(), exp):
                assert torch.allclose(p.grad, e, atol=1e-6), (it, n, (p.grad - e).abs().max())
        assert modes == ["learn", "bucket", "bucket", "learn", "bucket"], modes
        # a parameter nobody zeroes, declared while in bucket mode: its .grad is never None, so only its arrival hook can say
        # that this backward's contribution is there; the reduced value = history + mean of the new gradients
        keep = m.b.weight
        w.persistent_grad_ids = {id(keep)}
        hist = keep.grad.clone()
        for it in range(3):
            w.attach_grads()
            x, y = _data(rank)
            ((w(x, True) - y) ** 2).mean().backward()
            w.finish()
            new = torch.zeros_like(hist)
            for r in range(world):
                ref.zero_grad()
                xr, yr = _data(r)
                ((ref(xr, True) - yr) ** 2).mean().backward()
                new += ref.b.weight.grad / world
            hist = hist + new
            assert torch.allclose(keep.grad, hist, atol=1e-5), (it, (keep.grad - hist).abs().max())
        assert w._mode == "bucket"
        q.put((rank, "ok", 0))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", 0))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_hipddp_late_and_persistent_gradients_in_bucket_mode_world2():
    res = _run(_late_param_worker)
    assert [r[1] for r in res] == ["ok", "ok"], res


def _val_epoch_worker(rank, world, port, q):
    """_run_epoch('val') on two ranks: every rank issues the same collectives with the same key sets (retrieval metrics are
    computed on the reference rank only, the other rank reports placeholders under identical keys) and all ranks end up
    with rank 0's retrieval values and the rank-mean of the per-batch metrics.  The encoders are stubbed (no GPU here):
    this test is about the exchange pattern of runner.py, not the kernels."""
    try:
        _init(rank, world, port)
        import deepcoro_clip_amd.runner as R

        class Cfg:
            recall_k, ndcg_k = [1, 5], [5]
            device, world_size, is_ref_device = "cpu", world, rank == 0
            gradient_accumulation_steps, max_grad_norm, loss_name, scheduler_name, temperature = 1, 5.0, "contrastive", "", 0.07

        enc = torch.nn.Linear(2, 2)
        nb = 2 + rank  # different batch counts / sizes per rank: the gathers must cope with ragged shards
        loader = [{"videos": torch.zeros(3 + rank, 1), "reports": [f"r{rank}b{b}s{i}" for i in range(3 + rank)]} for b in range(nb)]
        runner = R.VideoContrastiveLearningRunner(config=Cfg(), train_loader=None, val_loader=loader, video_encoder=enc,
                                                  text_encoder=enc, optimizer=None, log_temp=torch.zeros(1), loss_fn=None)
        g = torch.Generator().manual_seed(5 + rank)

        def fake_val_step(videos, **kw):
            n = videos.shape[0]
            return ({"loss": 1.0 + rank, "alignment_score": 0.5}, {"video_embeddings": torch.randn(n, 8, generator=g),
                                                                   "text_embeddings": torch.randn(n, 8, generator=g)})

        runner._val_step = fake_val_step
        runner._preprocess_inputs = lambda batch: ({"videos": batch["videos"]}, [])
        R.compute_similarity_matrix = lambda v, t: torch.nn.functional.normalize(v, dim=1) @ torch.nn.functional.normalize(t, dim=1).t()
        calls = []

        def fake_metrics(sim, gt):
            calls.append(tuple(sim.shape))
            out = {f"Recall@{k}": 0.25 * k for k in Cfg.recall_k}
            out.update({"MRR_V2T": 0.5, "NDCG@5_V2T": 0.75, "MAP": 0.4, "MedianRank_V2T": 3})
            return out

        runner._retrieval_metrics = fake_metrics
        out = runner._run_epoch("val", 0)
        total = sum((3 + r) * (2 + r) for r in range(world))
        assert (calls == [(total, total)]) if rank == 0 else (calls == []), calls
        assert abs(out["val/loss"] - 1.5) < 1e-6 and abs(out["val/alignment_score"] - 0.5) < 1e-6
        assert out["val/Recall@5"] == 1.25 and out["val/MRR_V2T"] == 0.5 and out["val/MedianRank_V2T"] == 3.0, out
        q.put((rank, "ok", sorted(out)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", 0))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_run_epoch_val_collectives_world2():
    res = _run(_val_epoch_worker)
    assert [r[1] for r in res] == ["ok", "ok"], res
    assert res[0][2] == res[1][2], "every rank must report the same metric keys"


def _val_epoch_no_text_worker(rank, world, port, q):
    """A validation epoch whose steps produce no text embeddings (and no tokenizer): nothing can be ranked, so the retrieval
    keys must be ABSENT on every rank (they used to be reported — and checkpointed — as real 0.0), the text gather
    still runs on every rank (a rank-local `dim() == 2` used to guard that collective), and the other metrics still reduce."""
    try:
        _init(rank, world, port)
        import deepcoro_clip_amd.runner as R

        class Cfg:
            recall_k, ndcg_k = [1, 5], [5]
            device, world_size, is_ref_device = "cpu", world, rank == 0
            gradient_accumulation_steps, max_grad_norm, loss_name, scheduler_name, temperature = 1, 5.0, "contrastive", "", 0.07

        enc = torch.nn.Linear(2, 2)
        loader = [{"videos": torch.zeros(2 + rank, 1)} for _ in range(2)]
        runner = R.VideoContrastiveLearningRunner(config=Cfg(), train_loader=None, val_loader=loader, video_encoder=enc,
                                                  text_encoder=enc, optimizer=None, log_temp=torch.zeros(1), loss_fn=None)
        runner._val_step = lambda videos, **kw: ({"loss": 2.0 + rank}, {"video_embeddings": torch.ones(videos.shape[0], 8),
                                                                         "text_embeddings": torch.empty(0)})
        runner._preprocess_inputs = lambda batch: ({"videos": batch["videos"]}, [])
        out = runner._run_epoch("val", 0)
        assert abs(out["val/loss"] - 2.5) < 1e-6, out
        assert not any("Recall" in k or "MRR" in k or "NDCG" in k or "MAP" in k or "MedianRank" in k for k in out), out
        q.put((rank, "ok", sorted(out)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", 0))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_run_epoch_val_without_texts_drops_retrieval_keys_world2():
    res = _run(_val_epoch_no_text_worker)
    assert [r[1] for r in res] == ["ok", "ok"], res
    assert res[0][2] == res[1][2]
