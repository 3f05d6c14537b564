"""Eight-phase 256 x 256 GEMM (csrc/gemm_p8.hip, tuning variant 13): exact integer products over ragged M / N tails, strided
operands and every reduction depth class (2 .. 24 K-tiles, one and several tiles per workgroup), repeated launches (the kernel's
LDS-DMA / barrier schedule has no slack for a mis-counted wait: a race shows as a wrong tile in SOME run), and every fused epilogue
against the ring kernels' rounding contract and torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def p8_variant():
    from deepcoro_clip_amd import kernels as K

    old = K.GEMM_VARIANT
    K.GEMM_VARIANT = 13 << 8
    yield
    K.GEMM_VARIANT = old


def _tiled(K):
    class _Ctx:
        def __enter__(self):
            self.old = K.GEMM_VARIANT
            K.GEMM_VARIANT = 10 << 8

        def __exit__(self, *exc):
            K.GEMM_VARIANT = self.old
            return False
    return _Ctx()


def _int_operands(M, N, Kd, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(-3, 4, (M, Kd), generator=g).float()
    w = torch.randint(-3, 4, (N, Kd), generator=g).float()
    # asymmetric: a transposed, permuted or half-swapped tile must not pass
    w[torch.arange(N), torch.arange(N) % Kd] += (torch.arange(N) % 5).float()
    a[torch.arange(M), torch.arange(M) % Kd] += (torch.arange(M) % 3).float()
    return a, w


# one tile; ragged rows / columns; several tiles per workgroup (> 256 tiles); N < 256; deep and shallow reductions
SHAPES = [(256, 256, 128), (256, 256, 384), (300, 520, 128), (1000, 1152, 384), (777, 384, 1536), (4096, 1536, 192 + 64),
          (50208, 1536, 384), (50208, 384, 1536), (8192, 4096, 1024), (513, 8, 128), (31, 264, 3072), (70000, 1152, 128)]


@pytest.mark.parametrize("M,N,Kd", SHAPES)
def test_p8_integer_exact(M, N, Kd):
    from deepcoro_clip_amd import kernels as K

    a, w = _int_operands(M, N, Kd, M + 3 * N + Kd)
    ref = (a.cuda() @ w.cuda().t()).bfloat16()   # exact in fp32: |sum| < 2^24
    ab, wb = a.bfloat16().cuda(), w.bfloat16().cuda()
    for rep in range(3):
        out = torch.full((M, N), 5.0, device="cuda", dtype=torch.bfloat16)
        K.gemm(ab, wb, out=out)
        bad = out != ref
        assert not bad.any(), (rep, int(bad.sum()), bad.nonzero()[:4].tolist())
    # strided operands / output (row pitches != K, N; nothing outside the output columns is written)
    wide_a = torch.zeros(M, Kd + 64, dtype=torch.bfloat16, device="cuda")
    wide_a[:, 32:32 + Kd] = ab
    wide_c = torch.full((M, N + 40), 7.0, dtype=torch.bfloat16, device="cuda")
    K.gemm(wide_a[:, 32:32 + Kd], wb, out=wide_c[:, :N])
    assert torch.equal(wide_c[:, :N], ref) and torch.all(wide_c[:, N:] == 7.0)


def test_p8_repeated_launches_on_a_busy_chip():
    """60 launches back to back on alternating operand buffers (every launch finds L2 / MALL in a different state), each checked."""
    from deepcoro_clip_amd import kernels as K

    M, N, Kd = 25000, 1536, 384
    sets = []
    for seed in (1, 2, 3):
        a, w = _int_operands(M, N, Kd, seed)
        sets.append((a.bfloat16().cuda(), w.bfloat16().cuda(), (a.cuda() @ w.cuda().t()).bfloat16()))
    outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(60)]
    for i, o in enumerate(outs):
        ab, wb, _ = sets[i % 3]
        K.gemm(ab, wb, out=o)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        bad = o != sets[i % 3][2]
        assert not bad.any(), (i, int(bad.sum()), bad.nonzero()[:4].tolist())


@pytest.mark.parametrize("M,N,Kd", [(2100, 1536, 384), (1111, 384, 256), (4096, 1152, 768)])
def test_p8_epilogues_equal_ring_kernels_and_torch(M, N, Kd):
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
    print(f"[p8 gemm] M{M} N{N} K{Kd}: bit-identical to the ring kernel per epilogue (pre, gelu, resid, dgelu, all): {same}")
    pre, o1, o2, o3, _ = got
    ref_pre = a.double() @ w.double().t() + bias.double()
    rel = lambda x, y: ((x.double() - y).abs().max() / y.abs().max()).item()  # noqa: E731
    assert rel(pre, ref_pre) < 6e-3
    assert rel(o1, torch.nn.functional.gelu(pre.double())) < 8e-3
    assert rel(o2, ref_pre + res.double()) < 6e-3
    x = pre.double().clone().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert rel(o3, 0.5 * (a.double() @ w.double().t()) * x.grad) < 8e-3
