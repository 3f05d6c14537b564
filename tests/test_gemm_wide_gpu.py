"""Wide-tile GEMM kernel (csrc/gemm_wide.hip, 256 x 256 / 256 x 128 tiles, 8 waves): exact integer products in all four
operand layouts, ragged edges and K tails, fused bf16 epilogues, split-K fp32 accumulation with the fused bias column sums."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[8, 9], ids=["bn256", "bn128"])
def wide(request):
    from deepcoro_clip_amd import kernels as K

    old = K.GEMM_VARIANT
    K.GEMM_VARIANT = request.param << 8
    yield K
    K.GEMM_VARIANT = old


@pytest.mark.parametrize("a_km,b_km", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,Kd", [(256, 256, 128), (520, 392, 200), (1000, 136, 72), (136, 640, 1048)])
def test_wide_integer_exact_all_layouts(wide, a_km, b_km, M, N, Kd):
    K = wide
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randint(-3, 4, (M, Kd), generator=g).float()
    b = torch.randint(-3, 4, (N, Kd), generator=g).float()
    ref = a @ b.t()
    A = (a.t().contiguous() if a_km else a).bfloat16().cuda()
    Bm = (b.t().contiguous() if b_km else b).bfloat16().cuda()
    out = K.gemm(A, Bm, a_km=a_km, b_km=b_km, out_dtype=torch.float32)
    assert torch.equal(out.cpu(), ref), (out.cpu() - ref).abs().max()
    out16 = K.gemm(A, Bm, a_km=a_km, b_km=b_km)
    assert torch.equal(out16.float().cpu(), ref.bfloat16().float())


def test_wide_bf16_epilogues(wide):
    K = wide
    g = torch.Generator().manual_seed(3)
    M, N, Kd = 777, 384, 392
    a = torch.randn(M, Kd, generator=g).bfloat16().cuda()
    w = (torch.randn(N, Kd, generator=g) * 0.05).bfloat16().cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).bfloat16().cuda()
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = K.gemm(a, w, bias=bias, gelu=True, save_pre=True, aux=pre, resid=res)
    ref_pre = a.double() @ w.double().t() + bias.double()
    rel = lambda x, y: ((x.double() - y).abs().max() / y.abs().max()).item()  # noqa: E731
    assert rel(pre, ref_pre) < 6e-3
    assert rel(out, torch.nn.functional.gelu(pre.double()) + res.double()) < 8e-3
    out2 = K.gemm(a, w, dgelu=True, aux=pre, alpha=0.5)
    x = pre.double().clone().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert rel(out2, 0.5 * (a.double() @ w.double().t()) * x.grad) < 8e-3


def test_wide_weight_gradient_splitk_and_colsum(wide):
    K = wide
    g = torch.Generator().manual_seed(4)
    rows, N, Kd = 6000, 392, 264
    dy = torch.randint(-2, 3, (rows, N), generator=g).float()
    x = torch.randint(-2, 3, (rows, Kd), generator=g).float()
    ref = dy.t() @ x
    for sk in (1, 5, 16):
        out = torch.zeros(N, Kd, device="cuda")
        db = torch.zeros(N, device="cuda")
        K.gemm(dy.bfloat16().cuda(), x.bfloat16().cuda(), a_km=True, b_km=True, out=out, accumulate=sk > 1, splitk=sk, colsum=db)
        assert torch.equal(out.cpu(), ref), (sk, (out.cpu() - ref).abs().max())
        assert torch.equal(db.cpu(), dy.sum(0)), sk
