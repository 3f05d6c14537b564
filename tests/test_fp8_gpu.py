"""fp8 GEMM path of BASELINE configs[4]: quantisers against torch's float8 conversions, the block-scaled MFMA kernel
against exact integer products and against the dequantised fp32 product, its epilogues.  The VideoEncoder with
gemm_dtype="fp8" is compared with the fp64 oracle in tests/test_full_geometry_gpu.py (it shares that file's oracle cache)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _deq(q, scale, fmt):
    dt = torch.float8_e4m3fn if fmt == 0 else torch.float8_e5m2
    return q.view(dt).float() * scale


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fp8_quantize_matches_torch_float8(fmt, dtype):
    from deepcoro_clip_amd import kernels as K

    g = torch.Generator().manual_seed(fmt)
    x = (torch.randn(300, 256, generator=g) * 3).to(dtype).cuda()
    x[5, 7] = 37.5  # amax
    q, scale = K.fp8_quantize(x, fmt)
    fmax = 448.0 if fmt == 0 else 57344.0
    assert abs(scale.item() - 37.5 / fmax) < 1e-6 * 37.5 / fmax
    dt = torch.float8_e4m3fn if fmt == 0 else torch.float8_e5m2
    ref = (x.float() * (fmax / 37.5)).to(dt).view(torch.uint8)
    same = (q == ref).float().mean().item()
    assert same > 0.999, same  # round-to-nearest-even on both sides; ties on the scaled value may differ by the fp32 product
    d = (_deq(q, scale, fmt) - x.float()).abs() / x.float().abs().clamp_min(1e-3)
    assert d.max() < (0.08 if fmt == 0 else 0.15)
    qt, st = K.fp8_quantize(x[:, :128].contiguous(), fmt, transpose=True)
    q2, _ = K.fp8_quantize(x[:, :128].contiguous(), fmt)
    assert qt.shape == (128, 300) and torch.equal(qt.t().contiguous(), q2)
    # a strided view takes the generic kernels, a contiguous bf16 tensor the 16-byte streaming ones: identical bytes
    qs, ss = K.fp8_quantize(x[:, 64:192], fmt)
    qc, sc = K.fp8_quantize(x[:, 64:192].contiguous(), fmt)
    assert torch.equal(qs, qc) and torch.equal(ss, sc)
    z, sz = K.fp8_quantize(torch.zeros(8, 16, device="cuda", dtype=dtype), fmt)
    assert sz.item() == 1.0 and int(z.max()) == 0


@pytest.mark.parametrize("M,N,K_", [(256, 256, 256), (300, 136, 384), (1000, 384, 1536), (129, 8, 144)])
def test_fp8_gemm_exact_on_integer_data(M, N, K_):
    """Small integers are exact in e4m3 and their dot products exact in the fp32 accumulator: any operand-layout or
    K-tail mistake shows as an O(1) difference."""
    from deepcoro_clip_amd import kernels as K

    g = torch.Generator().manual_seed(M + N)
    a = torch.randint(-4, 5, (M, K_), generator=g).float()
    b = torch.randint(-4, 5, (N, K_), generator=g).float()
    a8 = a.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    b8 = b.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    one = torch.ones(1, device="cuda")
    out = K.gemm_fp8(a8, b8, one, one)
    ref = (a.double() @ b.double().t())
    assert out.shape == (M, N) and out.dtype == torch.bfloat16
    # integers up to 2^8 are exact in bf16; larger sums round to 8 significant bits
    assert torch.equal(out.float().cpu(), ref.float().bfloat16().float())
    # mixed formats: A in e5m2 (gradients), B in e4m3
    a5 = a.to(torch.float8_e5m2).view(torch.uint8).cuda()
    out5 = K.gemm_fp8(a5, b8, one, one, fmt_a=1)
    assert torch.equal(out5.float().cpu(), ref.float().bfloat16().float())


def test_fp8_gemm_scales_and_epilogues_vs_dequantised_product():
    from deepcoro_clip_amd import kernels as K

    g = torch.Generator().manual_seed(0)
    M, N, K_ = 640, 384, 512
    x = torch.randn(M, K_, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K_, generator=g) * 0.05).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).bfloat16().cuda()
    x8, sx = K.fp8_quantize(x, 0)
    w8, sw = K.fp8_quantize(w, 0)
    xd, wd = _deq(x8, sx, 0).double(), _deq(w8, sw, 0).double()
    pre_ref = xd @ wd.t() + bias.double()
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = K.gemm_fp8(x8, w8, sx, sw, bias=bias, gelu=True, save_pre=True, aux=pre, resid=res)
    rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()  # noqa: E731
    assert rel(pre, pre_ref) < 6e-3  # bf16 rounding of the output only
    ref = torch.nn.functional.gelu(pre.double()) + res.double()
    assert rel(out, ref) < 8e-3
    # and the quantisation itself stays within fp8's resolution of the true product
    assert rel(pre, x.double() @ w.double().t() + bias.double()) < 5e-2
    # dgelu epilogue with e5m2 gradients
    gy = torch.randn(M, K_, generator=g).bfloat16().cuda()
    g8, sg = K.fp8_quantize(gy, 1)
    out2 = K.gemm_fp8(g8, w8, sg, sw, fmt_a=1, dgelu=True, aux=pre)
    xx = pre.double().clone().requires_grad_(True)
    torch.nn.functional.gelu(xx).sum().backward()
    ref2 = (_deq(g8, sg, 1).double() @ wd.t()) * xx.grad
    assert rel(out2, ref2) < 8e-3
