"""Bandwidth-bound HIP kernels (LayerNorm, depthwise/max pooling, patch embedding, RoPE, embeddings, token mean,
elementwise) through the C ABI against plain PyTorch fp32 references of the same op, plus the reference-generated
RoPE golden (tests/golden/rope.npz).  fp32: 1e-5-class; bf16: 1e-2-class (tolerances stated per test)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def _gen(seed=0):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("rows,C", [(1, 96), (393, 96), (1000, 192), (37, 384), (257, 768), (64, 512)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
def test_layernorm_fwd_bwd(rows, C, dtype, tol):
    from deepcoro_clip_amd import functional as Fn

    g = _gen(rows + C)
    x0 = torch.randn(rows, C, generator=g) * 2 + 0.5
    gam0, bet0, dy0 = torch.randn(C, generator=g), torch.randn(C, generator=g), torch.randn(rows, C, generator=g)
    x0 = x0.to(dtype).float()  # both sides start from the same (rounded) activations
    xr, gr, br = x0.clone().requires_grad_(True), gam0.clone().requires_grad_(True), bet0.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gr, br, 1e-6)
    yr.backward(dy0.to(dtype).float())
    x = x0.cuda().to(dtype).requires_grad_(True)
    gam, bet = gam0.cuda().requires_grad_(True), bet0.cuda().requires_grad_(True)
    y = Fn.layernorm(x, gam, bet, 1e-6)
    y.backward(dy0.cuda().to(dtype))
    assert y.dtype == dtype and _rel(y.float(), yr) < tol
    assert _rel(x.grad.float(), xr.grad) < tol
    assert _rel(gam.grad, gr.grad) < tol and _rel(bet.grad, br.grad) < tol


@pytest.mark.parametrize("rows,C", [(5, 96), (4099, 96), (1000, 384), (70000, 96)])
def test_layernorm_bwd_workspace_and_atomic_paths_agree(rows, C):
    """dcc_layernorm_bwd folds dgamma/dbeta either through block partials in a caller workspace (+ fold kernel) or, with a
    null workspace, through atomics; both accumulate INTO the gradients and must agree; a ragged last row group and
    strided rows (ldx > C) go through the zero-filling buffer loads."""
    from deepcoro_clip_amd import _lib as L, kernels as K

    g = _gen(rows)
    ld = C + 32
    xb = torch.randn(rows, ld, generator=g).cuda().bfloat16()
    x = xb[:, :C]
    dy = torch.randn(rows, C, generator=g).cuda().bfloat16()
    gam, bet = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    _, mean, rstd = K.layernorm_fwd(x, gam, bet, 1e-6)
    outs = []
    for use_ws in (True, False):
        dg, db = torch.full((C,), 0.5, device="cuda"), torch.full((C,), -0.25, device="cuda")
        dx = torch.empty(rows, C, device="cuda", dtype=torch.bfloat16)
        ws = torch.empty(1 << 20, device="cuda") if use_ws else None
        L.call("dcc_layernorm_bwd", L.ptr(dy), C, L.ptr(x), ld, L.ptr(mean), L.ptr(rstd), L.ptr(gam), L.ptr(dx), C, L.ptr(dg),
               L.ptr(db), None, L.ptr(ws), rows, C, L.dt(x), L.dt(dy), L.stream())
        outs.append((dx.float(), dg, db))
    xr = x.float().clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    F.layer_norm(xr, (C,), gr, br, 1e-6).backward(dy.float())
    for dx, dg, db in outs:
        assert _rel(dx, xr.grad) < 1.5e-2
        assert _rel(dg - 0.5, gr.grad) < 2e-3 and _rel(db + 0.25, br.grad) < 2e-3
    assert torch.equal(outs[0][0], outs[1][0])


def test_layernorm_rejects_unsupported_width():
    from deepcoro_clip_amd import kernels as K
    from deepcoro_clip_amd._lib import DccError

    x = torch.randn(4, 3072, device="cuda")
    with pytest.raises(DccError):  # wider than any LayerNorm on the path (<= 768): loud, no fallback
        K.layernorm_fwd(x, torch.ones(3072, device="cuda"), torch.zeros(3072, device="cuda"), 1e-6)
    with pytest.raises(DccError):
        K.layernorm_fwd(x.cpu(), torch.ones(3072), torch.zeros(3072), 1e-6)


@pytest.mark.parametrize("stride,thw,H,C,B", [((1, 1, 1), (4, 8, 8), 2, 96, 2), ((1, 2, 2), (4, 8, 8), 1, 96, 2),
                                              ((1, 4, 4), (2, 16, 16), 2, 96, 2), ((1, 8, 8), (2, 16, 16), 1, 96, 2),
                                              ((1, 2, 2), (3, 7, 7), 4, 96, 2), ((1, 1, 1), (2, 5, 3), 8, 96, 2),
                                              # runs of 7 columns (grids 56/28/14/7) incl. the stride-2 row-run gradient
                                              ((1, 2, 2), (2, 14, 14), 2, 96, 3), ((1, 1, 1), (3, 14, 14), 4, 96, 33),
                                              ((1, 2, 2), (4, 56, 56), 2, 96, 13), ((1, 4, 4), (2, 28, 28), 1, 96, 2)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_depthwise_pool_fwd_bwd(stride, thw, H, C, B, dtype, tol):
    """Pool's Conv3d(groups=C, k=3, p=1, stride) on [B, N, H, C] token views with class-token bypass (R4)."""
    from deepcoro_clip_amd import kernels as K

    g = _gen(7)
    geo = K.PoolGeom(B, H, C, thw, (3, 3, 3), stride, (1, 1, 1), True)
    x0 = torch.randn(B, geo.n_in, H, C, generator=g).to(dtype).float()
    w0 = torch.randn(C, 1, 3, 3, 3, generator=g) * 0.2
    dy0 = torch.randn(B, geo.n_out, H, C, generator=g).to(dtype).float()

    def ref(x, w):
        cls, tok = x[:, :1], x[:, 1:]
        t = tok.permute(0, 2, 3, 1).reshape(B * H, C, *thw)
        o = F.conv3d(t, w, None, stride, (1, 1, 1), groups=C)
        o = o.reshape(B, H, C, -1).permute(0, 3, 1, 2)
        return torch.cat([cls, o], 1)

    xr, wr = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    yr = ref(xr, wr)
    yr.backward(dy0)
    x = x0.cuda().to(dtype)
    y, _ = K.pool_fwd(x, geo, w0.cuda().reshape(C, 27))
    assert _rel(y.float(), yr) < tol
    dx = torch.empty_like(x)
    K.pool_bwd_data(dy0.cuda().to(dtype), dx, geo, w0.cuda().reshape(C, 27))
    assert _rel(dx.float(), xr.grad) < tol
    dw = torch.zeros(C, 27, device="cuda")
    K.dwconv_bwd_weight(dy0.cuda().to(dtype), x, geo, dw)
    assert _rel(dw.view_as(w0), wr.grad) < tol
    K.dwconv_bwd_weight(dy0.cuda().to(dtype), x, geo, dw)  # accumulates
    assert _rel(dw.view_as(w0), 2 * wr.grad) < tol


@pytest.mark.parametrize("stride,thw,H,B", [((1, 1, 1), (4, 14, 14), 4, 3), ((1, 1, 1), (2, 5, 3), 8, 2),
                                           ((1, 2, 2), (3, 14, 14), 2, 3), ((1, 2, 2), (3, 7, 7), 4, 2),
                                           ((1, 4, 4), (2, 28, 28), 2, 2), ((1, 8, 8), (2, 56, 56), 1, 2),
                                           ((1, 1, 1), (8, 7, 7), 8, 5), ((1, 2, 2), (4, 56, 56), 2, 5)])
@pytest.mark.parametrize("ldo", [96, 128])
def test_fused_pool_layernorm_fwd(stride, thw, H, B, ldo):
    """mvit.Pool = depthwise Conv3d + LayerNorm(head_dim) in one launch (dcc_pool_ln_fwd): against the torch fp32 reference of
    both ops on the bf16-rounded conv output, and against the two-kernel path (conv bit-identical; the LayerNorm differs only
    by the order of its row sums).  x is the strided q/k/v slice of a qkv buffer, out a column slice of a wider operand."""
    from deepcoro_clip_amd import kernels as K

    C = 96
    g = _gen(11)
    geo = K.PoolGeom(B, H, C, thw, (3, 3, 3), stride, (1, 1, 1), True)
    qkv = (torch.randn(B, geo.n_in, 3, H, C, generator=g) * 0.7).bfloat16().cuda()
    x = qkv[:, :, 1]
    assert K.pool_ln_supported(x, geo)
    w = (torch.randn(C, 27, generator=g) * 0.2).cuda()
    gam = (torch.rand(C, generator=g) + 0.5).cuda()
    bet = (torch.randn(C, generator=g) * 0.1).cuda()
    rows = B * geo.n_out * H
    wide = torch.full((rows, ldo), 7.0, device="cuda", dtype=torch.bfloat16)
    out = wide[:, :C]
    conv, mean, rstd = K.pool_ln_fwd(x, geo, w, gam, bet, 1e-6, out)
    # two-kernel path
    conv2, _ = K.pool_fwd(x, geo, w)
    y2, mean2, rstd2 = K.layernorm_fwd(conv2.view(rows, C), gam, bet, 1e-6)
    assert torch.equal(conv, conv2)
    assert (out.float() - y2.float()).abs().max().item() <= 2.0 ** -5  # one bf16 ulp at |y| < 4
    assert _rel(mean, mean2) < 1e-5 and _rel(rstd, rstd2) < 1e-5
    if ldo > C:
        assert torch.all(wide[:, C:] == 7.0)  # the columns beyond the head dimension are not touched
    # torch reference
    xf = x.float().cpu()
    cls, tok = xf[:, :1], xf[:, 1:]
    t = tok.permute(0, 2, 3, 1).reshape(B * H, C, *thw)
    o = F.conv3d(t, w.cpu().view(C, 1, 3, 3, 3), None, stride, (1, 1, 1), groups=C)
    o = torch.cat([cls, o.reshape(B, H, C, -1).permute(0, 3, 1, 2)], 1)
    assert _rel(conv.float().cpu(), o) < 1e-2
    ref = F.layer_norm(conv.float().cpu(), (C,), gam.cpu(), bet.cpu(), 1e-6).reshape(rows, C)
    assert _rel(out.float().cpu(), ref) < 1e-2
    assert _rel(mean.cpu(), conv.float().cpu().mean(-1).reshape(-1)) < 1e-4


@pytest.mark.parametrize("stride,thw,H,B,A", [((1, 2, 2), (8, 14, 14), 4, 2, 32), ((1, 1, 1), (8, 14, 14), 8, 2, 64),
                                             ((1, 8, 8), (8, 56, 56), 1, 1, 32), ((1, 4, 4), (8, 56, 56), 2, 1, 64)])
def test_k_pool_writes_the_one_hot_key_augmentation(stride, thw, H, B, A):
    """The k pool's fused kernel also writes the rel-pos one-hot columns behind the key (dcc_relpos_kaug's job): identical bytes."""
    from deepcoro_clip_amd import kernels as K

    C = 96
    g = _gen(17)
    geo = K.PoolGeom(B, H, C, thw, (3, 3, 3), stride, (1, 1, 1), True)
    qkv = (torch.randn(B, geo.n_in, 3, H, C, generator=g) * 0.7).bfloat16().cuda()
    w = (torch.randn(C, 27, generator=g) * 0.2).cuda()
    gam = (torch.rand(C, generator=g) + 0.5).cuda(); bet = (torch.randn(C, generator=g) * 0.1).cuda()
    k_thw = geo.out_thw
    assert sum(k_thw) <= A
    Kf = torch.full((B, geo.n_out, H, C + A), 7.0, device="cuda", dtype=torch.bfloat16)
    K.pool_ln_fwd(qkv[:, :, 1], geo, w, gam, bet, 1e-6, Kf.view(-1, C + A)[:, :C], aug=A)
    ref = torch.full_like(Kf, 7.0)
    ref[..., :C] = Kf[..., :C]
    dims = K.relpos_dims(B, H, C, A, 8, (1, 1, 1), k_thw, (1, 1, 1))   # only the key grid matters to kaug
    K.relpos_kaug(ref, dims)
    assert torch.equal(Kf, ref)


def test_fused_k_pool_declines_augmentations_wider_than_64():
    """k_t + k_h + k_w > 64 (e.g. 32 frames x 448^2 in block 1) needs a 96-wide one-hot key augmentation: outside the fused
    kernel's domain, so dcc_pool_ln_fwd must answer DCC_ERR_UNSUPPORTED (wrapper: None -> the caller takes conv + LayerNorm +
    relpos_kaug), not an argument error; an augmentation that is not a multiple of 32 stays an argument error."""
    from deepcoro_clip_amd import kernels as K
    from deepcoro_clip_amd._lib import DccError

    C, A = 96, 96
    g = _gen(23)
    geo = K.PoolGeom(1, 2, C, (20, 28, 28), (3, 3, 3), (1, 1, 1), (1, 1, 1), True)
    qkv = (torch.randn(1, geo.n_in, 3, 2, C, generator=g) * 0.7).bfloat16().cuda()
    w = (torch.randn(C, 27, generator=g) * 0.2).cuda()
    gam = (torch.rand(C, generator=g) + 0.5).cuda(); bet = (torch.randn(C, generator=g) * 0.1).cuda()
    assert sum(geo.out_thw) > 64
    Kf = torch.full((1, geo.n_out, 2, C + A), 7.0, device="cuda", dtype=torch.bfloat16)
    assert K.pool_ln_fwd(qkv[:, :, 1], geo, w, gam, bet, 1e-6, Kf.view(-1, C + A)[:, :C], aug=A) is None
    assert torch.all(Kf == 7.0)   # nothing was launched
    # the unfused path the caller falls back to writes all 96 augmentation columns
    conv, _ = K.pool_fwd(qkv[:, :, 1], geo, w)
    K.layernorm_fwd(conv.view(-1, C), gam, bet, 1e-6, out=Kf.view(-1, C + A)[:, :C])
    K.relpos_kaug(Kf, K.relpos_dims(1, 2, C, A, 8, (1, 1, 1), geo.out_thw, (1, 1, 1)))
    aug = Kf[0, 1:, 0, C:].float()
    assert torch.all(aug.sum(-1) == 3.0) and torch.all(Kf[0, 0, :, C:] == 0)   # one hot per axis; class token: none
    with pytest.raises(DccError):
        K.pool_ln_fwd(qkv[:, :, 1], geo, w, gam, bet, 1e-6, Kf.view(-1, C + A)[:, :C], aug=40)


@pytest.mark.parametrize("stride,thw,H,B", [((1, 2, 2), (8, 14, 14), 4, 3), ((1, 8, 8), (8, 56, 56), 1, 2), ((1, 1, 1), (4, 7, 7), 8, 2)])
def test_k_and_v_pools_in_one_launch_equal_two_launches(stride, thw, H, B):
    from deepcoro_clip_amd import kernels as K

    C, A = 96, 32
    g = _gen(19)
    geo = K.PoolGeom(B, H, C, thw, (3, 3, 3), stride, (1, 1, 1), True)
    qkv = (torch.randn(B, geo.n_in, 3, H, C, generator=g) * 0.7).bfloat16().cuda()
    ws = [(torch.randn(C, 27, generator=g) * 0.2).cuda() for _ in range(2)]
    gs = [(torch.rand(C, generator=g) + 0.5).cuda() for _ in range(2)]
    bs = [(torch.randn(C, generator=g) * 0.1).cuda() for _ in range(2)]
    rows = B * geo.n_out * H
    Kf = torch.full((rows, C + A), 7.0, device="cuda", dtype=torch.bfloat16); Vn = torch.empty(rows, C, device="cuda", dtype=torch.bfloat16)
    Kf2 = Kf.clone(); Vn2 = Vn.clone()
    pair = K.pool_ln_fwd_pair((qkv[:, :, 1], qkv[:, :, 2]), geo, ws, gs, bs, 1e-6, (Kf[:, :C], Vn), augs=(A, 0))
    assert pair is not None
    one_k = K.pool_ln_fwd(qkv[:, :, 1], geo, ws[0], gs[0], bs[0], 1e-6, Kf2[:, :C], aug=A)
    one_v = K.pool_ln_fwd(qkv[:, :, 2], geo, ws[1], gs[1], bs[1], 1e-6, Vn2)
    assert torch.equal(Kf, Kf2) and torch.equal(Vn, Vn2)
    for (c, m, r), (c2, m2, r2) in zip(pair, (one_k, one_v)):
        assert torch.equal(c, c2) and torch.equal(m, m2) and torch.equal(r, r2)


def test_specialised_entries_report_unsupported_without_raising():
    """The fused pool + LayerNorm entry has a general HIP counterpart: outside its domain it returns DCC_ERR_UNSUPPORTED, which
    `try_call` turns into False (the caller then launches conv + LayerNorm); argument errors still raise."""
    from deepcoro_clip_amd import _lib as L
    from deepcoro_clip_amd import kernels as K
    from deepcoro_clip_amd._lib import DccError

    geo = K.PoolGeom(1, 1, 96, (2, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1), True)
    x = torch.randn(1, geo.n_in, 1, 96, device="cuda")          # fp32: not in the fused kernel's domain
    y = torch.empty(1, geo.n_out, 1, 96, device="cuda")
    o = torch.empty(geo.n_out, 96, device="cuda")
    w = torch.randn(96, 27, device="cuda"); g_ = torch.ones(96, device="cuda"); b_ = torch.zeros(96, device="cuda")
    m = torch.empty(geo.n_out, device="cuda"); r = torch.empty(geo.n_out, device="cuda")
    args = (L.ptr(x), L.ptr(y), L.ptr(o), 96, L.ptr(w), L.ptr(g_), L.ptr(b_), L.ptr(m), L.ptr(r), geo.dims(),
            K._pool_strides(x, y), 1e-6, 0, L.dt(x), L.stream())
    assert L.try_call("dcc_pool_ln_fwd", *args) is False
    with pytest.raises(DccError):
        L.try_call("dcc_pool_ln_fwd", None, *args[1:])


@pytest.mark.parametrize("C", [192, 768])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_skip_fwd_bwd(dtype, C):
    """pool_skip: MaxPool3d k(1,3,3) s(1,2,2) p(0,1,1) on the non-class tokens.  Max pooling selects, so results are
    bit-exact in either dtype."""
    from deepcoro_clip_amd import functional as Fn

    B, thw = 2, (2, 7, 9)
    g = _gen(3)
    n = thw[0] * thw[1] * thw[2] + 1
    x0 = torch.randn(B * n, C, generator=g).to(dtype).float()
    xr = x0.clone().requires_grad_(True)
    tok = xr.view(B, n, C)[:, 1:].permute(0, 2, 1).reshape(B, C, *thw)
    o = F.max_pool3d(tok, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    yr = torch.cat([xr.view(B, n, C)[:, :1], o.flatten(2).transpose(1, 2)], 1).reshape(-1, C)
    dy0 = torch.randn(yr.shape, generator=g).to(dtype).float()
    yr.backward(dy0)
    x = x0.cuda().to(dtype).requires_grad_(True)
    y = Fn.pool_skip(x, B, thw, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    y.backward(dy0.cuda().to(dtype))
    assert torch.equal(y.float().cpu(), yr.detach())
    assert _rel(x.grad.float(), xr.grad) < (1e-6 if dtype == torch.float32 else 1e-2)  # overlapping windows add


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,T,S", [(2, 4, 32), (1, 2, 20), (3, 6, 28)])
def test_patch_embed_fwd_bwd(dtype, tol, B, T, S):
    """conv_proj + class token as im2col + MFMA GEMM vs F.conv3d on the NDHWC video (models/video_encoder.py:237-241)."""
    from deepcoro_clip_amd import functional as Fn

    g = _gen(11)
    v0 = torch.randn(B, T, S, S, 3, generator=g)
    w0 = torch.randn(96, 3, 3, 7, 7, generator=g) * 0.05
    b0, c0 = torch.randn(96, generator=g), torch.randn(96, generator=g)
    wr, br, cr = (t.clone().requires_grad_(True) for t in (w0, b0, c0))
    o = F.conv3d(v0.permute(0, 4, 1, 2, 3), wr, br, (2, 4, 4), (1, 3, 3)).flatten(2).transpose(1, 2)
    yr = torch.cat([cr.expand(B, 1, 96), o], 1)
    dy0 = torch.randn(yr.shape, generator=g)
    yr.backward(dy0)
    w, b, c = (t.cuda().requires_grad_(True) for t in (w0, b0, c0))
    y = Fn.patch_embed(v0.cuda(), w, b, c, dtype)
    assert y.shape == (B * yr.shape[1], 96)
    y.backward(dy0.cuda().to(dtype).view(-1, 96))
    assert _rel(y.float().view_as(yr), yr) < tol
    assert _rel(w.grad, wr.grad) < tol and _rel(b.grad, br.grad) < tol and _rel(c.grad, cr.grad) < tol


@pytest.mark.parametrize("B,T,S", [(2, 16, 224), (3, 4, 112), (1, 6, 56)])
@pytest.mark.parametrize("u8", [False, True])
def test_im2col_rows_are_the_rounded_patches_exactly(B, T, S, u8):
    """conv_proj operand (models/video_encoder.py:237 Conv3d k(3,7,7) s(2,4,4) p(1,3,3) as im2col): the bf16 column rows are the
    input values rounded once, zero in the padding — exact equality with a torch unfold of the padded clip, for fp32 frames and
    for uint8 frames normalised inside the gather (the LDS-staged row kernel and the chunk kernel must agree with it bit for bit)."""
    from deepcoro_clip_amd import kernels as K

    g = _gen(21)
    kt, kh, kw, st, sh, sw, pt, ph, pw = 3, 7, 7, 2, 4, 4, 1, 3, 3
    if u8:
        raw = torch.randint(0, 256, (B, T, S, S, 3), generator=g, dtype=torch.uint8)
        mean = torch.tensor([110.2, 98.7, 120.1]); std = torch.tensor([58.4, 61.2, 57.3])
        vid = (raw.float() - mean) * (1.0 / std)
        col = K.im2col3d(raw.cuda(), (kt, kh, kw), (st, sh, sw), (pt, ph, pw), 448, torch.bfloat16, norm=(mean.cuda(), std.cuda()))
    else:
        vid = torch.randn(B, T, S, S, 3, generator=g)
        col = K.im2col3d(vid.cuda(), (kt, kh, kw), (st, sh, sw), (pt, ph, pw), 448, torch.bfloat16)
    x = F.pad(vid.permute(0, 4, 1, 2, 3), (pw, pw, ph, ph, pt, pt))            # [B, 3, T+2, S+6, S+6]
    p = x.unfold(2, kt, st).unfold(3, kh, sh).unfold(4, kw, sw)                # [B, 3, To, Ho, Wo, kt, kh, kw]
    p = p.permute(0, 2, 3, 4, 5, 6, 7, 1).reshape(-1, kt * kh * kw * 3)        # (kt, kh, kw, c) column order
    ref = torch.zeros(p.shape[0], 448)
    ref[:, :441] = p
    assert torch.equal(col.float().cpu(), ref.bfloat16().float())


def test_rope_kernel_matches_reference_golden():
    """dcc_rope with the host-built tables vs models/rope_3d.py outputs recorded in tests/golden/rope.npz
    (a subset `tok` of the 393 tokens of an 8x7x7 grid, class token first)."""
    from deepcoro_clip_amd import kernels as K
    from deepcoro_clip_amd.mvit import rope_tables

    d = np.load(os.path.join(G, "rope.npz"))
    cos, sin = rope_tables(8, 7, 7, 96, n_special=1)
    assert np.allclose(cos.numpy(), d["cos"], atol=1e-6) and np.allclose(sin.numpy(), d["sin"], atol=1e-6)
    tok = torch.from_numpy(d["tok"]).long()
    cos_t, sin_t = cos[tok].contiguous().cuda(), sin[tok].contiguous().cuda()
    for name in ("q", "k"):
        x = torch.from_numpy(d[name]).permute(0, 2, 1, 3).contiguous().cuda()  # [B, N, h, D]
        y = torch.empty_like(x)
        K.rope(x, y, cos_t, sin_t)
        ref = torch.from_numpy(d[name + "_rot"]).permute(0, 2, 1, 3)
        assert (y.cpu() - ref).abs().max().item() < 1e-5
        back = torch.empty_like(x)
        K.rope(y, back, cos_t, sin_t, backward=True)  # a rotation: the adjoint is the inverse
        assert (back - x).abs().max().item() < 1e-5
        xb = x.bfloat16()
        yb = torch.empty_like(xb)
        K.rope(xb, yb, cos_t, sin_t)
        assert _rel(yb.float().cpu(), ref) < 1e-2
    c2, s2 = rope_tables(8, 14, 14, 96, n_special=1)
    sel = torch.from_numpy(d["sel"]).long()
    x = torch.from_numpy(d["q2_sel"]).permute(0, 2, 1, 3).contiguous().cuda()
    y = torch.empty_like(x)
    K.rope(x, y, c2[sel].contiguous().cuda(), s2[sel].contiguous().cuda())
    assert (y.cpu() - torch.from_numpy(d["q2_rot_sel"]).permute(0, 2, 1, 3)).abs().max().item() < 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_embedding_fwd_bwd(dtype, tol):
    """BERT embeddings: word + position + token-type(0) sum and its scatter-add gradient."""
    from deepcoro_clip_amd import functional as Fn

    g = _gen(5)
    V, P, D, B, S = 300, 40, 768, 3, 17
    ids = torch.randint(0, V, (B, S), generator=g)
    ids[0, :4] = 7  # repeated ids -> colliding atomic adds
    w0, p0, t0 = torch.randn(V, D, generator=g), torch.randn(P, D, generator=g), torch.randn(2, D, generator=g)
    wr, pr, tr = (t.clone().requires_grad_(True) for t in (w0, p0, t0))
    yr = F.embedding(ids, wr) + pr[:S][None] + tr[0][None, None]
    dy0 = torch.randn(yr.shape, generator=g).to(dtype).float()
    yr.backward(dy0)
    w, p, t = (x.cuda().requires_grad_(True) for x in (w0, p0, t0))
    y = Fn.embed(ids.cuda(), w, p, t, dtype)
    y.backward(dy0.cuda().to(dtype).view(y.shape))
    assert _rel(y.float().view_as(yr), yr) < tol
    assert _rel(w.grad, wr.grad) < tol and _rel(p.grad, pr.grad) < tol and _rel(t.grad, tr.grad) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_token_mean_qpool_add_cast(dtype, tol):
    from deepcoro_clip_amd import functional as Fn

    g = _gen(9)
    x0 = torch.randn(3, 393, 512, generator=g).to(dtype).float()
    xr = x0.clone().requires_grad_(True)
    xr.mean(1).backward(torch.ones(3, 512))
    x = x0.cuda().to(dtype).requires_grad_(True)
    y = Fn.token_mean(x)
    y.backward(torch.ones(3, 512, device="cuda", dtype=dtype))
    assert _rel(y.float(), x0.mean(1)) < tol and _rel(x.grad.float(), xr.grad) < tol
    # aggregator pooling: softmax(q . x^T) x over the N videos of a study (unscaled, video_aggregator.py:146-158)
    z0, q0 = torch.randn(4, 5, 512, generator=g) * 0.1, torch.randn(512, generator=g) * 0.1
    zr, qr = z0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
    wts = torch.softmax(zr @ qr, dim=1)
    outr = (wts[..., None] * zr).sum(1)
    go = torch.randn(4, 512, generator=g)
    outr.backward(go)
    z, q = z0.cuda().requires_grad_(True), q0.cuda().requires_grad_(True)
    out = Fn.qpool(z, q)
    out.backward(go.cuda())
    assert _rel(out, outr) < 1e-5 and _rel(z.grad, zr.grad) < 1e-5 and _rel(q.grad, qr.grad) < 1e-5
    a, b = torch.randn(1000, 7, generator=g), torch.randn(1000, 7, generator=g)
    assert _rel(Fn.add(a.cuda().to(dtype), b.cuda().to(dtype)).float(), a.to(dtype).float() + b.to(dtype).float()) < tol
    c = Fn.cast(a.cuda(), torch.bfloat16)
    assert torch.equal(c.cpu(), a.bfloat16())  # round-to-nearest-even like torch


def test_nan_to_num_and_dropout():
    from deepcoro_clip_amd import kernels as K

    x = torch.tensor([1.0, float("nan"), float("inf"), -float("inf"), -2.0] * 1001, device="cuda")
    K.nan_to_num_(x, 0.0, 1e4, -1e4)
    assert torch.equal(x.cpu(), torch.tensor([1.0, 0.0, 1e4, -1e4, -2.0] * 1001))
    y = torch.ones(1 << 20, device="cuda")
    m = K.dropout(y, 0.25, seed=123)
    kept = (m != 0).float().mean().item()
    assert abs(kept - 0.75) < 5e-3
    assert torch.allclose(m[m != 0], torch.tensor(1 / 0.75, device="cuda"))
    assert torch.equal(m, K.dropout(y, 0.25, seed=123)) and not torch.equal(m, K.dropout(y, 0.25, seed=124))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
def test_layernorm_skip_sums_both_gradients(dtype, tol):
    """(LN(x), x): the gradient of the residual use of x is added inside the LayerNorm-backward kernel."""
    from deepcoro_clip_amd import functional as Fn

    g = _gen(21)
    rows, C = 777, 384
    x0 = (torch.randn(rows, C, generator=g) * 1.5).to(dtype).float()
    gam0, bet0 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    d1, d2 = torch.randn(rows, C, generator=g).to(dtype).float(), torch.randn(rows, C, generator=g).to(dtype).float()
    xr, gr, br = x0.clone().requires_grad_(True), gam0.clone().requires_grad_(True), bet0.clone().requires_grad_(True)
    (F.layer_norm(xr, (C,), gr, br, 1e-6) * d1 + xr * d2).sum().backward()
    x = x0.cuda().to(dtype).requires_grad_(True)
    gam, bet = gam0.cuda().requires_grad_(True), bet0.cuda().requires_grad_(True)
    y, xs = Fn.layernorm_skip(x, gam, bet, 1e-6)
    torch.autograd.backward([y, xs], [d1.cuda().to(dtype), d2.cuda().to(dtype)])
    assert torch.equal(xs, x)
    assert _rel(x.grad.float(), xr.grad) < tol
    assert _rel(gam.grad, gr.grad) < tol and _rel(bet.grad, br.grad) < tol
    # only one of the two outputs used downstream: the missing gradient counts as zero
    x2 = x0.cuda().to(dtype).requires_grad_(True)
    y2, _ = Fn.layernorm_skip(x2, gam, bet, 1e-6)
    y2.backward(d1.cuda().to(dtype))
    xr2 = x0.clone().requires_grad_(True)
    F.layer_norm(xr2, (C,), gam0, bet0, 1e-6).backward(d1)
    assert _rel(x2.grad.float(), xr2.grad) < tol


def test_u8_normalize_matches_host_normalisation_bit_exact():
    """uint8 frames normalised on the GPU == float(x).sub(mean).div(std) of utils/video.py (v2.Normalize), bit for bit."""
    from deepcoro_clip_amd import kernels as K

    g = _gen(17)
    frames = torch.randint(0, 256, (2, 3, 20, 28, 3), generator=g, dtype=torch.uint8)
    mean, std = torch.tensor([104.5, 99.25, 97.0]), torch.tensor([53.1, 51.7, 50.9])
    ref = (frames.float() - mean) / std
    out = K.u8_normalize(frames.cuda(), mean.cuda(), std.cuda())
    assert out.dtype == torch.float32 and torch.equal(out.cpu(), ref)


def test_bf16_gelu_pair_against_erf():
    """The bf16 GEMM epilogues evaluate GELU as x * sigmoid(x (1.6003 + 0.0694 x^2)) (csrc/common.h), the reference
    (torchvision MViT MLP, HF BERT: nn.GELU) uses erf.  Bound the difference at kernel level over [-8, 8]:
    forward through a GEMM epilogue (A = [x | 0], B = [1 | 0]: the product is x exactly), derivative through dcc_dgelu_mul."""
    from deepcoro_clip_amd import kernels as K

    M = 8192
    x = torch.linspace(-8.0, 8.0, M).bfloat16()          # the bf16 grid points the kernel sees
    a = torch.zeros(M, 32, dtype=torch.bfloat16); a[:, 0] = x
    b = torch.zeros(128, 32, dtype=torch.bfloat16); b[:, 0] = 1.0
    pre = torch.empty(M, 128, dtype=torch.bfloat16, device="cuda")
    y = K.gemm(a.cuda(), b.cuda(), gelu=True, save_pre=True, aux=pre)
    assert torch.equal(pre[:, 0].cpu(), x) and torch.equal(pre[:, 5].cpu(), x)
    xd = x.double()
    ref = 0.5 * xd * (1.0 + torch.erf(xd / math.sqrt(2.0)))
    err = (y[:, 0].double().cpu() - ref).abs()
    bound = 2.8e-4 + ref.abs() * 2.0 ** -8          # fit error + one bf16 rounding of the result
    assert torch.all(err <= bound), (float(err.max()), float(xd[(err - bound).argmax()]))
    # derivative: dy = 1 -> dgelu(pre)
    d = K.dgelu_mul(torch.ones(M, device="cuda", dtype=torch.bfloat16), x.cuda())
    dref = 0.5 * (1.0 + torch.erf(xd / math.sqrt(2.0))) + xd * torch.exp(-0.5 * xd * xd) / math.sqrt(2.0 * math.pi)
    derr = (d.double().cpu() - dref).abs()
    dbound = 9.0e-4 + dref.abs() * 2.0 ** -8
    assert torch.all(derr <= dbound), (float(derr.max()), float(xd[(derr - dbound).argmax()]))
    # forward / backward consistency: the kernel's derivative is the derivative of the kernel's forward (central difference
    # of the closed form in fp64)
    g = lambda t: t / (1.0 + torch.exp(-t * (1.60031417 + 0.06940178 * t * t)))  # noqa: E731
    num = (g(xd + 1e-5) - g(xd - 1e-5)) / 2e-5
    assert torch.all((d.double().cpu() - num).abs() <= 1e-5 + num.abs() * 2.0 ** -8)
    print(f"[gelu] max |gelu_bf16 - erf form| {float(err.max()):.2e}, max |dgelu_bf16 - erf form| {float(derr.max()):.2e}")
