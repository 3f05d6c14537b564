"""Fused LayerNorm + MLP forward (csrc/mlp_fused.hip, `dcc_ln_mlp_fwd`) through the C ABI.

torchvision MultiscaleBlock's `x + mlp(norm2(x))` (models/video_encoder.py:232-253) as one launch.  Checked
  * EXACTLY on integer-valued problems (balanced +-1 rows make LayerNorm exact, a large fc1 bias makes GELU the identity): every
    index of the packed weight image, the accumulator-as-operand k permutation, bias / residual initialisation and all five
    output tensors, on every channel count of the domain, ragged row counts and more rows than one pass of the grid;
  * against the unfused HIP path (LayerNorm kernel + two GEMMs: <= 1-2 bf16 steps) and a torch fp32 reference on random data;
  * as an autograd node against the unfused node pair (same backward kernels: gradients agree to bf16 noise).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

CHANNELS = [96, 192, 384]


def _int_problem(M, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    H = 4 * C

    def sparse_pm1(rows, cols, nnz):  # exactly nnz entries of +-1 per row
        idx = torch.rand(rows, cols, generator=g).argsort(1)[:, :nnz]
        w = torch.zeros(rows, cols)
        w.scatter_(1, idx, torch.randint(0, 2, (rows, nnz), generator=g).float() * 2 - 1)
        return w

    # rows of +-1 with zero mean: LayerNorm(x) = +-1 / sqrt(1 + eps) -> +-1 exactly after the bf16 rounding
    x = torch.where(torch.rand(M, C, generator=g).argsort(1) < C // 2, 1.0, -1.0)
    w1 = sparse_pm1(H, C, 2)                                                        # fc1 in {-2, 0, 2}
    b1 = torch.full((H,), 10.0) + torch.randint(0, 3, (H,), generator=g).float()   # pre in [8, 14]: gelu(pre) == pre in bf16
    w2 = sparse_pm1(C, H, 8)
    b2 = torch.randint(-3, 4, (C,), generator=g).float()
    return x, w1, b1, w2, b2


@pytest.mark.parametrize("C", CHANNELS)
@pytest.mark.parametrize("M", [32, 40, 1000, 70017])
@pytest.mark.parametrize("residual", [True, False])
def test_integer_problem_is_exact(C, M, residual):
    from deepcoro_clip_amd import kernels as K

    if M > 2000 and not residual:
        pytest.skip("one large case per channel count is enough")
    x, w1, b1, w2, b2 = _int_problem(M, C, seed=C + M)
    gamma, beta = torch.ones(C), torch.zeros(C)
    pre_ref = x @ w1.t() + b1
    assert pre_ref.min() >= 8 and pre_ref.max() <= 16
    y_ref = pre_ref @ w2.t() + b2 + (x if residual else 0)
    assert y_ref.abs().max() <= 256
    packed = K.mlp_pack(w1.cuda().bfloat16(), w2.cuda().bfloat16())
    y, xn, mean, rstd, pre, act = K.ln_mlp_fwd(x.cuda().bfloat16(), gamma.cuda(), beta.cuda(), 1e-6, packed, b1.cuda(), b2.cuda(),
                                               residual=residual)
    torch.cuda.synchronize()
    assert torch.equal(xn.float().cpu(), x)
    assert torch.equal(pre.float().cpu(), pre_ref)
    assert torch.equal(act.float().cpu(), pre_ref)
    assert torch.equal(y.float().cpu(), y_ref)
    assert mean.abs().max().item() < 1e-6 and (rstd - 1.0).abs().max().item() < 1e-5


@pytest.mark.parametrize("C", CHANNELS)
def test_random_data_against_unfused_path_and_fp32_reference(C):
    from deepcoro_clip_amd import kernels as K

    M = 4099
    g = torch.Generator().manual_seed(C)
    H = 4 * C
    x = (torch.randn(M, C, generator=g) * 1.5 + 0.3).bfloat16()
    gamma, beta = 1.0 + 0.1 * torch.randn(C, generator=g), 0.05 * torch.randn(C, generator=g)
    w1, b1 = (torch.randn(H, C, generator=g) * C ** -0.5).bfloat16(), 0.1 * torch.randn(H, generator=g)
    w2, b2 = (torch.randn(C, H, generator=g) * H ** -0.5).bfloat16(), 0.1 * torch.randn(C, generator=g)
    xd, w1d, w2d = x.cuda(), w1.cuda(), w2.cuda()
    y, xn, mean, rstd, pre, act = K.ln_mlp_fwd(xd, gamma.cuda(), beta.cuda(), 1e-6, K.mlp_pack(w1d, w2d), b1.cuda(), b2.cuda())
    # the unfused HIP path: same rounding points (LayerNorm output, pre-activation, activation in bf16)
    xn_u, mean_u, rstd_u = K.layernorm_fwd(xd, gamma.cuda(), beta.cuda(), 1e-6)
    pre_u = torch.empty((M, H), device="cuda", dtype=torch.bfloat16)
    act_u = K.gemm(xn_u, w1d, bias=b1.cuda(), gelu=True, save_pre=True, aux=pre_u)
    y_u = K.gemm(act_u, w2d, bias=b2.cuda(), resid=xd)
    torch.cuda.synchronize()

    def ulps(a, b):  # max difference in units of the bf16 spacing at the larger magnitude
        a, b = a.float(), b.float()
        step = torch.maximum(a.abs(), b.abs()).clamp_min(2.0 ** -6) * 2.0 ** -7
        return ((a - b).abs() / step).max().item()

    assert torch.allclose(mean, mean_u, atol=1e-5) and torch.allclose(rstd, rstd_u, rtol=1e-5)
    assert ulps(xn, xn_u) <= 1.01
    # a one-step difference in xn moves pre by ~2^-8 |w|: compare pre / act / y as tensors, not element for element
    for a, b, tol in ((pre, pre_u, 4e-3), (act, act_u, 4e-3), (y, y_u, 4e-3)):
        rel = ((a.float() - b.float()).norm() / b.float().norm()).item()
        assert rel < tol, rel
    # torch fp32 reference of the same bf16 inputs (erf GELU: the bf16 epilogue's sigmoid form is within 2.8e-4 of it)
    xf = x.float()
    ref_xn = torch.nn.functional.layer_norm(xf, (C,), gamma, beta, 1e-6)
    ref_pre = ref_xn @ w1.float().t() + b1
    ref_y = xf + torch.nn.functional.gelu(ref_pre) @ w2.float().t() + b2
    assert ((y.float().cpu() - ref_y).norm() / ref_y.norm()).item() < 6e-3
    assert ((pre.float().cpu() - ref_pre).norm() / ref_pre.norm()).item() < 6e-3
    assert ((xn.float().cpu() - ref_xn).norm() / ref_xn.norm()).item() < 4e-3


@pytest.mark.parametrize("C", CHANNELS)
def test_inference_mode_saves_nothing_and_matches(C):
    from deepcoro_clip_amd import kernels as K

    M = 777
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, C, generator=g).bfloat16().cuda()
    w1 = (torch.randn(4 * C, C, generator=g) * 0.05).bfloat16().cuda()
    w2 = (torch.randn(C, 4 * C, generator=g) * 0.05).bfloat16().cuda()
    args = (torch.ones(C).cuda(), torch.zeros(C).cuda(), 1e-6, K.mlp_pack(w1, w2), torch.zeros(4 * C).cuda(), torch.zeros(C).cuda())
    full = K.ln_mlp_fwd(x, *args)
    lean = K.ln_mlp_fwd(x, *args, save=False)
    pre_only = K.ln_mlp_fwd(x, *args, save_act=False)
    assert all(t is None for t in lean[1:]) and pre_only[5] is None
    assert torch.equal(full[0], lean[0]) and torch.equal(full[0], pre_only[0]) and torch.equal(full[4], pre_only[4])


@pytest.mark.parametrize("C", CHANNELS)
@pytest.mark.parametrize("residual", [True, False])
def test_autograd_node_matches_the_unfused_nodes(C, residual):
    from deepcoro_clip_amd import functional as Fn

    M = 3000 + C
    g = torch.Generator().manual_seed(7 + C)
    H = 4 * C

    def params():
        gg = torch.Generator().manual_seed(11)
        return [t.cuda().requires_grad_(True) for t in (
            1.0 + 0.1 * torch.randn(C, generator=gg), 0.05 * torch.randn(C, generator=gg),
            torch.randn(H, C, generator=gg) * C ** -0.5, 0.1 * torch.randn(H, generator=gg),
            torch.randn(C, H, generator=gg) * H ** -0.5, 0.1 * torch.randn(C, generator=gg))]

    x0 = torch.randn(M, C, generator=g).bfloat16()
    dy = torch.randn(M, C, generator=g).bfloat16().cuda()
    xa, xb = x0.cuda().requires_grad_(True), x0.cuda().requires_grad_(True)
    pa, pb = params(), params()
    assert Fn.ln_mlp_ok(xa, pa[2], pa[4]) == (C in Fn.FUSED_MLP_CHANNELS)
    ya = Fn.ln_mlp(xa, pa[0], pa[1], pa[2], pa[3], pa[4], pa[5], 1e-6, residual=residual)
    ya.backward(dy)
    xn, xs = Fn.layernorm_skip(xb, pb[0], pb[1], 1e-6)
    yb = Fn.mlp(xn, pb[2], pb[3], pb[4], pb[5], residual=xs if residual else None)
    if not residual:
        yb = yb + 0.0 * xs
    yb.backward(dy)
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()  # noqa: E731
    assert rel(ya, yb) < 4e-3
    assert rel(xa.grad, xb.grad) < 1e-2
    for a, b, name in zip(pa, pb, ("gamma", "beta", "w1", "b1", "w2", "b2")):
        assert a.grad is not None and rel(a.grad, b.grad) < 1e-2, (name, rel(a.grad, b.grad))


def test_weight_image_follows_the_optimizer():
    """The packed image is cached per (parameter versions, optimizer epoch): an in-place update must re-pack."""
    from deepcoro_clip_amd import functional as Fn

    C = 96
    w1 = torch.nn.Parameter(torch.randn(4 * C, C).cuda() * 0.05)
    w2 = torch.nn.Parameter(torch.randn(C, 4 * C).cuda() * 0.05)
    p0 = Fn.wpack_mlp(w1, w2)
    assert Fn.wpack_mlp(w1, w2) is p0
    with torch.no_grad():
        w2.mul_(2.0)
    p1 = Fn.wpack_mlp(w1, w2)
    assert p1 is not p0 and not torch.equal(p0, p1)
    Fn.invalidate_weight_cache()
    assert Fn.wpack_mlp(w1, w2) is not p1


def test_c384_is_fused_only_when_no_backward_can_follow():
    """C = 384 loses with the saved tensors and wins without them: the dispatch follows autograd's state."""
    from deepcoro_clip_amd import functional as Fn

    if Fn._FUSED_MLP_ENV != "1":
        pytest.skip("DCC_FUSED_MLP overrides the default dispatch")
    C = 384
    x = torch.randn(2, 40, C, device="cuda").bfloat16()
    w1 = torch.nn.Parameter(torch.randn(4 * C, C, device="cuda") * 0.05)
    w2 = torch.nn.Parameter(torch.randn(C, 4 * C, device="cuda") * 0.05)
    assert not Fn.ln_mlp_ok(x, w1, w2)          # training: parameters require grad
    with torch.no_grad():
        assert Fn.ln_mlp_ok(x, w1, w2)          # validation / inference()
    assert Fn.ln_mlp_ok(x, w1.detach(), w2.detach())   # frozen weights, input without grad: nothing to save either
    assert Fn.ln_mlp_ok(x[..., :192].contiguous(), torch.nn.Parameter(w1[:768, :192].detach().clone()), torch.nn.Parameter(w2[:192, :768].detach().clone()))
