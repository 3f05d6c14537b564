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
        for it in range(2):  # second iteration re-attaches / re-zeroes the arena
            w.attach_grads()
            x, y = _data(rank)
            ((w(x) - y) ** 2).mean().backward()
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
                assert p.grad.data_ptr() >= w._arena.data_ptr()  # a view into the flat arena
                assert torch.allclose(p.grad, e, atol=1e-6), (it, (p.grad - e).abs().max())
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
