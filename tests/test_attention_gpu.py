"""Flash attention kernels (csrc/attention.hip) vs a plain PyTorch fp32/fp64 reference."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, scale, kbias, residual, res_start):
    q = q.double().detach().requires_grad_(True)
    k = k.double().detach().requires_grad_(True)
    v = v.double().detach().requires_grad_(True)
    z = scale * torch.einsum("bqhd,bkhd->bhqk", q, k)
    if kbias is not None:
        z = z + kbias.double()[:, None, None, :]
    p = z.softmax(-1)
    o = torch.einsum("bhqk,bkhd->bqhd", p, v)
    if residual:
        mask = (torch.arange(q.shape[1], device=q.device) >= res_start).double()[None, :, None, None]
        o = o + q[..., : v.shape[-1]] * mask
    lse = torch.logsumexp(z, -1)
    return q, k, v, o, lse


CASES = [
    # B, H, Nq, Nk, DQK, DV, kbias, residual
    (2, 2, 70, 45, 96, 96, False, True),
    (1, 1, 300, 393, 128, 96, False, True),
    (2, 4, 33, 33, 160, 96, False, False),
    (3, 12, 32, 32, 64, 64, True, False),
    (2, 8, 1, 393, 64, 64, False, False),
    (2, 4, 8, 8, 128, 128, False, False),
    # long query ranges: the software-pipelined dK/dV kernel (blocks streaming >= 768 queries), even and odd tile counts
    (1, 2, 1000, 200, 128, 96, False, True),
    (1, 1, 1030, 393, 160, 96, False, False),
    (1, 1, 801, 40, 128, 96, True, True),
    (1, 2, 900, 300, 96, 96, False, False),   # the RoPE configuration's head width (no rel-pos columns)
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Nq,Nk,DQK,DV,use_bias,residual", CASES)
def test_attention_fwd_bwd(dtype, B, H, Nq, Nk, DQK, DV, use_bias, residual):
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Nq * 10 + Nk)
    # heads interleaved in memory like the qkv GEMM output: [B, N, H, D]
    q = torch.randn(B, Nq, H, DQK, generator=g).to("cuda", dtype)
    k = torch.randn(B, Nk, H, DQK, generator=g).to("cuda", dtype)
    v = torch.randn(B, Nk, H, DV, generator=g).to("cuda", dtype)
    do = torch.randn(B, Nq, H, DV, generator=g).to("cuda", dtype)
    kbias = None
    if use_bias:
        kbias = torch.zeros(B, Nk)
        kbias[:, Nk // 2 + 3:] = -1e30  # padding-mask style
        kbias[0] = 0
        kbias = kbias.cuda()
    scale = 1.0 / math.sqrt(96.0)
    o, lse = Kx.attn_fwd(q, k, v, scale=scale, kbias=kbias, residual=residual, res_start=1)
    qr, kr, vr, oref, lseref = _ref(q, k, v, scale, kbias, residual, 1)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (o.double() - oref).abs().max().item() < tol * max(1.0, oref.abs().max().item())
    assert (lse.double() - lseref).abs().max().item() < (1e-4 if dtype == torch.float32 else 2e-2)

    for qsplit in (1, 3):
        dq, dk, dv = Kx.attn_bwd(q, k, v, o, do, lse, scale=scale, kbias=kbias, residual=residual, res_start=1,
                                 qsplit=qsplit)
        oref.backward(do.double(), retain_graph=True)
        dq_ref = qr.grad.clone()
        # with `residual` the kernel's dq includes the shortcut's gradient, like autograd's
        gtol = 5e-5 if dtype == torch.float32 else 4e-2
        for name, got, ref in (("dq", dq, dq_ref), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
            err = (got.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
            assert err < gtol, (name, qsplit, err)
        qr.grad = None; kr.grad = None; vr.grad = None
