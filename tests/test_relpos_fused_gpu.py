"""Fused rel-pos forward (dcc_relpos_aug_fwd) against the stacked-table GEMM + gather path it replaces and against a torch fp32
restatement of torchvision mvit._add_rel_pos's inner products (reference call site models/video_encoder.py:354-364)."""
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


@pytest.mark.parametrize("B,H,q_thw,k_thw", [(2, 4, (8, 14, 14), (8, 7, 7)), (1, 2, (8, 28, 28), (8, 7, 7)),
                                             (1, 1, (8, 56, 56), (8, 7, 7)), (2, 8, (8, 7, 7), (8, 7, 7)),
                                             (3, 2, (4, 14, 14), (4, 7, 7)), (1, 1, (16, 56, 56), (16, 7, 7))])
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
    assert K.relpos_fused_supported(torch.empty(1, dtype=torch.bfloat16), D, A, k_thw, q_thw, Lp)
    q = (torch.randn(B, Nq, H, D, generator=g) * 0.5).bfloat16()
    Qa = torch.full((B, Nq, H, DQK), 3.0, dtype=torch.bfloat16).cuda()
    Qa[..., :D] = q.cuda()
    Qb = Qa.clone()
    # ---------------------------------------------------------------- forward
    K.relpos_aug_fwd(Qa, Rcat, idx, dims, scale)
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
    assert (got - ref).abs().max().item() / den < 1e-2           # one bf16 rounding of fp32 dot products
    assert (Qb[..., D:].float().cpu() - ref).abs().max().item() / den < 2e-2   # the old path rounds twice
    assert torch.all(got[:, 0] == 0) and torch.all(got[..., kh + kw + kt:] == 0)
