"""Streaming panel GEMM (csrc/gemm_stream.hip: weight resident in LDS, every wave on its own 32-row tiles; the early MViT
blocks' short-reduction products): exact integer products for every reduction length it is built for, ragged row tails, column
chunking, every fused epilogue against the tiled kernels (bit-identical: same rounding contract) and against torch; the
data-gradient path through transposed weight shadows incl. their invalidation by an optimizer step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def whole_domain():
    """Variant 12 sends every shape of the streaming kernel's domain to it (the default dispatch keeps only the shapes it wins:
    plain wide outputs, csrc/gemm.hip), for the forward products and — through functional._dx_gemm — the data gradients."""
    from deepcoro_clip_amd import kernels as K

    old = K.GEMM_VARIANT
    K.GEMM_VARIANT = 12 << 8
    yield
    K.GEMM_VARIANT = old


def _tiled(K):
    """Context: the same call on the tiled ring kernel (variant 10: neither the wide nor the streaming kernel is chosen)."""
    class _Ctx:
        def __enter__(self):
            self.old = K.GEMM_VARIANT
            K.GEMM_VARIANT = 10 << 8

        def __exit__(self, *exc):
            K.GEMM_VARIANT = self.old
            return False
    return _Ctx()


# (M, N, K): K = every template instance; N: one chunk, 2 and 3 chunks (K = 192: 384 columns per chunk); M: ragged tails
SHAPES = [(16384, 288, 96), (16417, 96, 96), (20001, 384, 96), (16384, 768, 96), (16450, 576, 192), (16384, 768, 192),
          (16390, 1152, 192), (16384, 96, 288), (16399, 96, 384), (16384, 192, 384), (16384, 96, 576), (16384, 192, 576)]


@pytest.mark.parametrize("M,N,Kd", SHAPES)
def test_stream_integer_exact(M, N, Kd):
    from deepcoro_clip_amd import kernels as K

    assert K.stream_gemm_shape(M, N, Kd, torch.bfloat16)
    g = torch.Generator().manual_seed(M + 3 * N + Kd)
    a = torch.randint(-3, 4, (M, Kd), generator=g).float()
    w = torch.randint(-3, 4, (N, Kd), generator=g).float()
    # asymmetric weights (a transposed or permuted tile must not pass): column n also carries n % 5 on k = n % Kd
    w[torch.arange(N), torch.arange(N) % Kd] += (torch.arange(N) % 5).float()
    ref = (a @ w.t()).cuda()   # exact in fp32: |sum| < 2^24
    out = K.gemm(a.bfloat16().cuda(), w.bfloat16().cuda())
    assert out.dtype == torch.bfloat16 and torch.equal(out.float(), ref.bfloat16().float()), (out.float() - ref).abs().max()
    # strided operands / output: A a column slice of a wider buffer, C a column slice (row pitches != K, N)
    wide_a = torch.zeros(M, Kd + 64, dtype=torch.bfloat16, device="cuda")
    wide_a[:, 32:32 + Kd] = a.bfloat16().cuda()
    wide_c = torch.full((M, N + 32), 7.0, dtype=torch.bfloat16, device="cuda")
    K.gemm(wide_a[:, 32:32 + Kd], w.bfloat16().cuda(), out=wide_c[:, :N])
    assert torch.equal(wide_c[:, :N].float(), ref.bfloat16().float()) and torch.all(wide_c[:, N:] == 7.0)


@pytest.mark.parametrize("M,N,Kd", [(16401, 384, 96), (16384, 96, 384), (16390, 768, 192)])
def test_stream_epilogues_equal_tiled_kernels_and_torch(M, N, Kd):
    """bias / GELU + saved pre-activation / residual / gelu' x saved pre-activation: the streaming kernel keeps the tiled kernels'
    rounding contract (round once after alpha * acc + bias, ops on the rounded value), so the outputs agree to the last bf16 ulp
    (bit-identical where the tiled path is the mailbox kernel; reported); plus a torch fp64 check of the values themselves."""
    from deepcoro_clip_amd import kernels as K

    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, Kd, generator=g).bfloat16().cuda()
    w = (torch.randn(N, Kd, generator=g) * 0.08).bfloat16().cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).bfloat16().cuda()

    def run():
        pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        o1 = K.gemm(a, w, bias=bias, gelu=True, save_pre=True, aux=pre)
        o2 = K.gemm(a, w, bias=bias, resid=res)
        o3 = K.gemm(a, w, dgelu=True, aux=pre, alpha=0.5)
        o4 = K.gemm(a, w, bias=bias, gelu=True, save_pre=True, aux=pre, resid=res)
        return pre, o1, o2, o3, o4

    got = run()
    with _tiled(K):
        want = run()
    same = []
    for i, (x, y) in enumerate(zip(got, want)):
        same.append(bool(torch.equal(x, y)))
        d = (x.float() - y.float()).abs()
        assert torch.all(d <= 2.0 ** -7 * y.float().abs().clamp_min(2.0 ** -6)), (i, float(d.max()))   # at most one bf16 ulp apart
    print(f"[stream gemm] M{M} N{N} K{Kd}: bit-identical to the tiled kernel per epilogue (pre, gelu, resid, dgelu, all): {same}")
    pre, o1, o2, o3, _ = got
    ref_pre = a.double() @ w.double().t() + bias.double()
    rel = lambda x, y: ((x.double() - y).abs().max() / y.abs().max()).item()  # noqa: E731
    assert rel(pre, ref_pre) < 6e-3
    assert rel(o1, torch.nn.functional.gelu(pre.double())) < 8e-3
    assert rel(o2, ref_pre + res.double()) < 6e-3
    x = pre.double().clone().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert rel(o3, 0.5 * (a.double() @ w.double().t()) * x.grad) < 8e-3


def test_stream_data_gradient_through_transposed_shadow_and_its_invalidation():
    """LinearFn / MlpFn backward hand dX = dY W to the streaming kernel with a transposed bf16 weight shadow (functional.wcast_t);
    the shadow must follow the weight through an optimizer step (HipAdamW updates parameters through raw pointers)."""
    from deepcoro_clip_amd import functional as Fn
    from deepcoro_clip_amd.optim import HipAdamW

    rows, C = 16416, 96
    g = torch.Generator().manual_seed(5)
    lin1, lin2 = torch.nn.Linear(C, 4 * C).cuda(), torch.nn.Linear(4 * C, C).cuda()
    opt = HipAdamW([{"params": list(lin1.parameters()) + list(lin2.parameters()), "lr": 0.05, "weight_decay": 0.0}], lr=0.05)
    x0 = torch.randn(rows, C, generator=g).bfloat16().cuda()
    up = torch.randn(rows, C, generator=g).bfloat16().cuda()
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = Fn.mlp(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias, residual=x)
        y.backward(up)
        xr = x0.double().clone().requires_grad_(True)
        w1, b1, w2, b2 = (t.detach().double() for t in (lin1.weight, lin1.bias, lin2.weight, lin2.bias))
        yr = torch.nn.functional.gelu(xr @ w1.bfloat16().double().t() + b1) @ w2.bfloat16().double().t() + b2 + xr
        yr.backward(up.double())
        rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()  # noqa: E731
        assert rel(y.detach(), yr.detach()) < 1e-2, (step, rel(y.detach(), yr.detach()))
        assert rel(x.grad, xr.grad) < 2e-2, (step, rel(x.grad, xr.grad))   # wrong (stale) weights would be O(1) off after the step
        opt.step()   # lr 0.05: the weights move by ~0.05 per element, far beyond bf16 noise
    lin = torch.nn.Linear(C, C).cuda()
    x = x0.clone().requires_grad_(True)
    Fn.linear(x, lin.weight, lin.bias).backward(up)
    want = up.double() @ lin.weight.detach().bfloat16().double()
    assert ((x.grad.double() - want).norm() / want.norm()).item() < 1e-2
