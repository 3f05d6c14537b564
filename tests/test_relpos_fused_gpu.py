"""Fused rel-pos kernels (dcc_relpos_aug_fwd_mfma / dcc_relpos_aug_bwd) against the stacked-table GEMM + gather / scatter path they
replace and against a torch fp32 restatement of torchvision mvit._add_rel_pos's inner products (reference call site
models/video_encoder.py:354-364)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tables(q_thw, k_thw, D, g):
    from deepcoro_clip_amd import functional as Fn

    idx, inv, tabs = [], [], []
    for a in (1, 2, 0):  # h, w, t
        qs, ks = q_thw[a], k_thw[a]
        d = 2 * max(qs, ks) - 1
        ix = Fn.relpos_index(qs, ks)
        iv = torch.full((qs, d), -1, dtype=torch.int32)
        iv.scatter_(1, ix.long(), torch.arange(ks, dtype=torch.int32).expand(qs, ks).contiguous())
        idx.append(ix.cuda())
        inv.append(iv.contiguous().cuda())
        tabs.append((torch.randn(d, D, generator=g) * 0.5).bfloat16())
    return idx, inv, tabs


@pytest.mark.parametrize("B,H,q_thw,k_thw", [(2, 4, (8, 14, 14), (8, 7, 7)), (1, 2, (8, 28, 28), (8, 14, 14)),
                                             (1, 1, (8, 56, 56), (8, 7, 7)), (2, 8, (8, 7, 7), (8, 7, 7)),
                                             (3, 2, (4, 14, 14), (4, 7, 7))])
def test_fused_relpos_matches_gemm_path_and_reference(B, H, q_thw, k_thw):
    from deepcoro_clip_amd import functional as Fn
    from deepcoro_clip_amd import kernels as K

    D = 96
    g = torch.Generator().manual_seed(5)
    A = Fn.relpos_aug_width(k_thw)
    DQK = D + A
    Nq = q_thw[0] * q_thw[1] * q_thw[2] + 1
    idx, inv, tabs = _tables(q_thw, k_thw, D, g)
    lens = tuple(t.shape[0] for t in tabs)
    Lp = 8 * ((sum(lens) + 7) // 8)
    Rcat = torch.zeros(Lp, D, dtype=torch.bfloat16)
    Rcat[:sum(lens)] = torch.cat(tabs, 0)
    Rcat = Rcat.cuda()
    scale = 1.0 / D ** 0.5
    dims = K.relpos_dims(B, H, D, A, Lp, q_thw, k_thw, lens)
    q = (torch.randn(B, Nq, H, D, generator=g) * 0.5).bfloat16()
    Qa = torch.full((B, Nq, H, DQK), 3.0, dtype=torch.bfloat16).cuda()
    Qa[..., :D] = q.cuda()
    Qb = Qa.clone()
    # ---------------------------------------------------------------- forward
    assert K.relpos_aug_fwd_mfma(Qa, Rcat, idx, dims, scale)
    P = K.gemm(Qb.view(-1, DQK)[:, :D], Rcat)
    K.relpos_gather(Qb, P, idx, dims, scale)
    assert torch.equal(Qa[..., :D], Qb[..., :D])
    # fp32 reference of the used entries: [h | w | t] columns, zero padding, zero class-token rows
    qf = q.float()[:, 1:].reshape(B, *q_thw, H, D)
    Rh, Rw, Rt = (t.float() for t in tabs)
    ih, iw, it = (i.cpu().long() for i in idx)
    ref = torch.zeros(B, Nq, H, A)
    r = ref[:, 1:].view(B, *q_thw, H, A)
    kh, kw, kt = k_thw[1], k_thw[2], k_thw[0]
    r[..., :kh] = torch.einsum("bthwnd,hkd->bthwnk", qf, Rh[ih])
    r[..., kh:kh + kw] = torch.einsum("bthwnd,wkd->bthwnk", qf, Rw[iw])
    r[..., kh + kw:kh + kw + kt] = torch.einsum("bthwnd,tkd->bthwnk", qf, Rt[it])
    ref /= scale
    got = Qa[..., D:].float().cpu()
    den = ref.abs().max().item()
    assert (got - ref).abs().max().item() / den < 2e-2           # P rounded to bf16, then the scaled entry
    assert (Qb[..., D:].float().cpu() - ref).abs().max().item() / den < 2e-2   # the old path rounds twice
    assert torch.all(got[:, 0] == 0) and torch.all(got[..., kh + kw + kt:] == 0)


@pytest.mark.parametrize("B,H,q_thw,k_thw", [(2, 4, (8, 14, 14), (8, 7, 7)), (1, 2, (8, 28, 28), (8, 14, 14)),
                                             (3, 8, (8, 7, 7), (8, 14, 14)), (1, 1, (8, 56, 56), (8, 7, 7)),
                                             (2, 8, (8, 7, 7), (8, 7, 7)), (1, 4, (4, 14, 14), (4, 14, 14))])
def test_fused_relpos_backward_matches_gemm_path_and_reference(B, H, q_thw, k_thw):
    """dcc_relpos_aug_bwd (dP built per tile in LDS, both products on the matrix cores) against scatter + two GEMMs and against
    an fp32 restatement: dq = dQf[:, :D] + sum_j g_j R[l_j] / scale, dR[l] = sum over (row, j -> l) g_j q_row / scale."""
    from deepcoro_clip_amd import functional as Fn
    from deepcoro_clip_amd import kernels as K

    D = 96
    g = torch.Generator().manual_seed(9)
    A = Fn.relpos_aug_width(k_thw)
    DQK = D + A
    Nq = q_thw[0] * q_thw[1] * q_thw[2] + 1
    idx, inv, tabs = _tables(q_thw, k_thw, D, g)
    lens = tuple(t.shape[0] for t in tabs)
    Lp = 8 * ((sum(lens) + 7) // 8)
    Rcat = torch.zeros(Lp, D, dtype=torch.bfloat16)
    Rcat[:sum(lens)] = torch.cat(tabs, 0)
    Rcat = Rcat.cuda()
    scale = 1.0 / D ** 0.5
    dims = K.relpos_dims(B, H, D, A, Lp, q_thw, k_thw, lens)
    Qf = (torch.randn(B, Nq, H, DQK, generator=g) * 0.5).bfloat16().cuda()
    dQf = (torch.randn(B, Nq, H, DQK, generator=g) * 0.5).bfloat16().cuda()
    dR = torch.zeros(Lp, D, device="cuda")
    dq = K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, scale, dR)
    assert dq is not None
    dP = torch.empty(B * Nq * H, Lp, device="cuda", dtype=torch.bfloat16)
    K.relpos_scatter(dQf, dP, inv, dims, scale)
    dR_old = Fn.wgrad(dP, Qf.view(-1, DQK)[:, :D])
    dq_old = K.gemm(dP, Rcat, b_km=True, resid=dQf.view(-1, DQK)[:, :D]).view(B, Nq, H, D)
    # fp32 reference from the bf16-rounded scaled gradients (what both paths feed the matrix cores)
    kh, kw, kt = k_thw[1], k_thw[2], k_thw[0]
    Rh, Rw, Rt = (t.float() for t in tabs)
    ih, iw, it = (i.cpu().long() for i in idx)
    gaug = (dQf[..., D:].float().cpu() / scale).bfloat16().float()
    gv = gaug[:, 1:].reshape(B, *q_thw, H, A)
    qf = Qf[..., :D].float().cpu()[:, 1:].reshape(B, *q_thw, H, D)
    dq_ref = dQf[..., :D].float().cpu().clone()
    dqv = dq_ref[:, 1:].view(B, *q_thw, H, D)
    dqv += torch.einsum("bthwnk,hkd->bthwnd", gv[..., :kh], Rh[ih])
    dqv += torch.einsum("bthwnk,wkd->bthwnd", gv[..., kh:kh + kw], Rw[iw])
    dqv += torch.einsum("bthwnk,tkd->bthwnd", gv[..., kh + kw:kh + kw + kt], Rt[it])
    dRh = torch.zeros_like(Rh).index_put_((ih.reshape(-1),), torch.einsum("bthwnk,bthwnd->hkd", gv[..., :kh], qf).reshape(-1, D), accumulate=True)
    dRw = torch.zeros_like(Rw).index_put_((iw.reshape(-1),), torch.einsum("bthwnk,bthwnd->wkd", gv[..., kh:kh + kw], qf).reshape(-1, D), accumulate=True)
    dRt = torch.zeros_like(Rt).index_put_((it.reshape(-1),), torch.einsum("bthwnk,bthwnd->tkd", gv[..., kh + kw:kh + kw + kt], qf).reshape(-1, D), accumulate=True)
    dR_ref = torch.cat([dRh, dRw, dRt], 0)
    n = sum(lens)
    rel = lambda a, b: (a - b).abs().max().item() / b.abs().max().item()
    assert rel(dq.float().cpu(), dq_ref) < 1e-2
    assert rel(dq_old.float().cpu(), dq_ref) < 1e-2
    assert rel(dR[:n].cpu(), dR_ref) < 1e-3                       # exact bf16 products, fp32 accumulation
    assert rel(dR_old[:n].float().cpu(), dR_ref) < 1e-3
    assert torch.all(dR[n:] == 0)
    dR2 = dR.clone()
    K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, scale, dR2)        # accumulates
    assert rel(dR2[:n].cpu(), 2 * dR_ref) < 1e-3


@pytest.mark.parametrize("B,H,q_thw,k_thw", [(2, 4, (8, 14, 14), (8, 7, 7)), (1, 2, (8, 28, 28), (8, 14, 14)),
                                             (3, 8, (8, 7, 7), (8, 14, 14)), (1, 1, (8, 56, 56), (8, 7, 7)),
                                             (1, 4, (4, 14, 14), (4, 14, 14))])
def test_mfma_relpos_forward_is_bit_identical_to_gemm_plus_gather(B, H, q_thw, k_thw):
    from deepcoro_clip_amd import functional as Fn
    from deepcoro_clip_amd import kernels as K

    D = 96
    g = torch.Generator().manual_seed(13)
    A = Fn.relpos_aug_width(k_thw)
    DQK = D + A
    Nq = q_thw[0] * q_thw[1] * q_thw[2] + 1
    idx, inv, tabs = _tables(q_thw, k_thw, D, g)
    lens = tuple(t.shape[0] for t in tabs)
    Lp = 8 * ((sum(lens) + 7) // 8)
    Rcat = torch.zeros(Lp, D, dtype=torch.bfloat16)
    Rcat[:sum(lens)] = torch.cat(tabs, 0)
    Rcat = Rcat.cuda()
    scale = 1.0 / D ** 0.5
    dims = K.relpos_dims(B, H, D, A, Lp, q_thw, k_thw, lens)
    Qa = torch.full((B, Nq, H, DQK), 3.0, dtype=torch.bfloat16).cuda()
    Qa[..., :D] = (torch.randn(B, Nq, H, D, generator=g) * 0.5).bfloat16().cuda()
    Qb = Qa.clone()
    assert K.relpos_aug_fwd_mfma(Qa, Rcat, idx, dims, scale)
    P = K.gemm(Qb.view(-1, DQK)[:, :D], Rcat)
    K.relpos_gather(Qb, P, idx, dims, scale)
    assert torch.equal(Qa, Qb)


@pytest.mark.parametrize("B,H,q_thw,k_thw", [(2, 4, (8, 14, 14), (8, 7, 7)), (1, 2, (8, 28, 28), (8, 14, 14)),
                                             (3, 8, (8, 7, 7), (8, 14, 14)), (1, 1, (8, 56, 56), (8, 7, 7)),
                                             (2, 8, (8, 7, 7), (8, 7, 7)), (1, 4, (4, 14, 14), (4, 14, 14))])
def test_fused_relpos_backward_with_layernorm_epilogue(B, H, q_thw, k_thw):
    """dcc_relpos_aug_bwd_ln against dcc_relpos_aug_bwd followed by dcc_layernorm_bwd on the same inputs: the epilogue rounds dq
    to bf16 before use, so dconv may differ only by the summation order of the two row means (<= 1 bf16 step), dgamma / dbeta and
    the table gradient by fp32 summation order; and against an fp64 LayerNorm backward of the two-kernel dq."""
    from deepcoro_clip_amd import functional as Fn
    from deepcoro_clip_amd import kernels as K

    D = 96
    g = torch.Generator().manual_seed(11)
    A = Fn.relpos_aug_width(k_thw)
    DQK = D + A
    Nq = q_thw[0] * q_thw[1] * q_thw[2] + 1
    idx, inv, tabs = _tables(q_thw, k_thw, D, g)
    lens = tuple(t.shape[0] for t in tabs)
    Lp = 8 * ((sum(lens) + 7) // 8)
    Rcat = torch.zeros(Lp, D, dtype=torch.bfloat16)
    Rcat[:sum(lens)] = torch.cat(tabs, 0)
    Rcat = Rcat.cuda()
    scale = 1.0 / D ** 0.5
    dims = K.relpos_dims(B, H, D, A, Lp, q_thw, k_thw, lens)
    rows = B * Nq * H
    Qf = (torch.randn(B, Nq, H, DQK, generator=g) * 0.5).bfloat16().cuda()
    dQf = (torch.randn(B, Nq, H, DQK, generator=g) * 0.5).bfloat16().cuda()
    conv = (torch.randn(B, Nq, H, D, generator=g) * 1.5 + 0.3).bfloat16().cuda()
    gamma = (1.0 + 0.2 * torch.randn(D, generator=g)).cuda()
    beta = torch.zeros(D).cuda()
    _, mean, rstd = K.layernorm_fwd(conv.view(rows, D), gamma, beta, 1e-6)
    # two kernels
    dR0 = torch.zeros(Lp, D, device="cuda")
    dq = K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, scale, dR0)
    dg0, db0 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dconv0 = K.layernorm_bwd(dq.view(rows, D), conv.view(rows, D), mean, rstd, gamma, dg0, db0)
    # one kernel; dgamma / dbeta / dR are accumulated into: start them off non-zero
    dR1 = torch.full((Lp, D), 0.5, device="cuda")
    dg1, db1 = torch.full((D,), 2.0, device="cuda"), torch.full((D,), -3.0, device="cuda")
    dconv1 = K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, scale, dR1, ln=(conv, mean, rstd, gamma, dg1, db1))
    assert dconv1 is not None and dconv1.shape == (B, Nq, H, D)
    torch.cuda.synchronize()
    a, b = dconv1.view(rows, D).float(), dconv0.float()
    # one bf16 step at the value's binade (upper bound), plus fp32 summation-order noise of the terms that cancel into it
    step = b.abs() * 2.0 ** -7 + 1e-6 * b.abs().max()
    bad = (a - b).abs() > step
    assert not bad.any(), (int(bad.sum()), (a - b).abs().max().item(), a[bad][:4].tolist(), b[bad][:4].tolist())
    assert (a != b).float().mean().item() < 2e-2                # and nearly all elements identical
    rel = lambda x, y: (x - y).abs().max().item() / y.abs().max().item()
    assert rel(dg1 - 2.0, dg0) < 1e-4 and rel(db1 + 3.0, db0) < 1e-4
    assert rel(dR1 - 0.5, dR0) < 1e-5
    # fp64 LayerNorm backward of the bf16 dq
    d = dq.view(rows, D).double()
    x = conv.view(rows, D).double()
    xh = (x - mean.double()[:, None]) * rstd.double()[:, None]
    gg = d * gamma.double()
    ref = rstd.double()[:, None] * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True))
    assert rel(a.double(), ref) < 6e-3
    assert rel((dg1 - 2.0).double(), (d * xh).sum(0)) < 1e-4 and rel((db1 + 3.0).double(), d.sum(0)) < 1e-4
